"""Developer tool: the C5 frontier wave several times in one process (variance of the single timed wave of bench.py)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from gomilp_amd import lp, synth
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m, seed = synth.CONFIGS['C5']
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(); r0 = cx.upload(c, A, b).solve(0.0); cx.close()
mask = synth.integrality_mask(m, m)
children = synth.frontier_children(r0.x, mask, 8)
pool = lp.FrontierPool(workers=workers); pool.set_root(c, A, b)
pool.solve(children[:32])
ts = []
for rep in range(8):
    t = time.perf_counter(); r = pool.solve(children); ts.append(time.perf_counter() - t)
print('workers', workers, 'wave ms', [round(x * 1e3, 1) for x in ts], 'median relax/s', round(len(children) / sorted(ts)[len(ts) // 2]))
pool.close()
