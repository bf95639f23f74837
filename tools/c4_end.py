"""Developer tool: pin the END state of the C4 solve (4096 x 8192; the oracle needs CPU-days for the whole pivot sequence).
  gpu:    python tools/c4_end.py gpu     (GPU box)  -> gpurun_out/c4_end_basis.npy: the positional final basis of the GPU solve
  oracle: python tools/c4_end.py oracle  (build container, ~minutes) -> tests/golden/lp_C4_end.npz: the oracle started FROM that basis
          (lp.Simplex's initialBasic, simplex.go:147-161): it must find the basis optimal at once (0 pivots) and returns the x / z bits
          of its own gonum-order solve of that basis."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import synth
m, seed = synth.CONFIGS["C4"]
c, A, b = synth.dense_lp_standard_form(m, seed)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "gpu":
    from gomilp_amd import lp
    cx = lp.Context(); p = cx.upload(c, A, b); r = p.solve(0.0); cx.close()
    assert r.status == lp.OK
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(root, "gpurun_out", "c4_end_basis.npy"), np.asarray(r.basis, dtype=np.int64))
    np.save(os.path.join(root, "gpurun_out", "c4_end_x.npy"), r.x)
    print("C4 GPU solve: pivots", r.stats["pivots_phase2"], "z %.17g" % r.z)
else:
    import time
    from oracle import oracle as O
    O.set_threads(8)
    basis = np.load(os.path.join(root, "gpurun_out", "c4_end_basis.npy"))
    t0 = time.time()
    o = O.simplex(c, A, b, 0.0, basis, trace=True)
    print("oracle from the GPU's final basis: status", o.status, "pivots", o.pivots_phase1, o.pivots_phase2, "z %.17g" % o.z, "%.0f s" % (time.time() - t0))
    assert o.status == 0 and o.pivots_phase2 == 0 and np.array_equal(o.basis, basis)
    np.savez_compressed(os.path.join(root, "tests", "golden", "lp_C4_end.npz"), m=m, seed=seed, basis=basis, x=o.x, z=o.z)
    gx = np.load(os.path.join(root, "gpurun_out", "c4_end_x.npy"))
    print("GPU x bits equal the oracle's:", bool(np.array_equal(gx, o.x)))
