"""Developer tool: final-solve schedules side by side (bitwise equality + time).  gpurun -- python tools/lu_check.py"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from gomilp_amd import lp, synth
for name in ('C3', 'C2', 'M', 'C4'):
    m, seed = synth.CONFIGS[name]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    ref = None
    for mode in (2, 1):
        cx = lp.Context(lu_blocked=mode); rl = cx.upload(c, A, b)
        r = rl.solve(0.0); r = rl.solve(0.0)
        s = r.stats
        same = None if ref is None else bool(np.array_equal(ref.x, r.x) and ref.z == r.z)
        ref = ref or r
        print(name, m, 'mode', mode, lp.STATUS_NAMES[r.status], 'dense', s['lu_dense_steps'], 'rounds', s['lu_rounds'],
              'final dev %.2f ms host %.2f ms total %.2f ms' % (s['seconds_final_device'] * 1e3, s['seconds_final_host'] * 1e3, s['seconds_total'] * 1e3),
              'same_bits', same, flush=True)
        cx.close()
