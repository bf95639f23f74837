"""Developer tool (diagnostic flavour: GOMILP_DEBUG_BUILD=1): cycles per segment of a pivot of the register-resident kernel (res_kernels.hip),
on the heaviest child of the C5 wave.  usage: GOMILP_DEBUG_BUILD=1 python tools/res_stamps.py"""
import sys, os, ctypes, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, 8)
pool = lp.FrontierPool(workers=4, batch_res=1); pool.set_root(c, A, b)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); res = pool.solve(children[:1]); best = min(best, time.perf_counter() - t0)
buf = (ctypes.c_ulonglong * 128)()
lp.lib().gomilp_debug_res_stamps(buf)
a = np.array(buf[:], dtype=np.float64).reshape(4, 32)
nx, npv = a[0, 31], a[0, 30]
names = ["dump", "barrier", "wg pick", "ratio rest", "record", "poll", "pick", "row", "update", "rest", "argmin", "col+quot", "wave min", "row out", "u"]
print("heaviest child: %.3f ms per solve (batch %.3f ms); pivots %d exchanges %d (%.2f per pivot), same-XCD mode in %d, missing-record polls %.2f per exchange" % (
    1e3 * best, 1e3 * res.stats["seconds_batch"], npv, nx, nx / max(npv, 1), a[0, 28], a[0, 29] / max(nx, 1)))
for w in range(4):
    print("wave", w, " ".join("%s %.0f" % (names[i], a[w, i] / max(npv, 1)) for i in range(15)), "| sum %.0f cycles per pivot" % (a[w, :15].sum() / max(npv, 1)))
pool.close()
