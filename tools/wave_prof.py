"""Developer tool: batched C5 waves only (for rocprofv3 --kernel-trace --stats, or warm-up patterns).
usage: wave_prof.py [waves] [workers] [gap_s] [branch_vars: 8 = 256 children, 11 = 2048]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
children = synth.frontier_children(root.x, mask, int(sys.argv[4]) if len(sys.argv) > 4 else 8)
pool = lp.FrontierPool(workers=int(sys.argv[2]) if len(sys.argv) > 2 else 16, batched=1)
pool.set_root(c, A, b)
gap = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
packed = lp.pack_children(children)   # (as bench.py does: the packing of 2048 children is not the wave)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if gap: time.sleep(gap)
    t0 = time.perf_counter(); res = pool.solve(packed); dt = time.perf_counter() - t0
    print("wave %.2f ms | C side %.2f ms batch %.2f ms supersteps %d fallbacks %d busy %.2f ms feasible %d" % (1e3 * dt, 1e3 * res.stats["seconds_total"], 1e3 * res.stats["seconds_batch"], res.stats["supersteps"], res.stats["host_fallbacks"],
                                                                                                         1e3 * res.stats["seconds_busy_sum"], int((res.status == 0).sum())), flush=True)
pool.close()
