"""Developer tool: N complete solves of a named config (for rocprofv3 --kernel-trace --stats).
   cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof -- python3 /root/repo/tools/solve_n.py M 5"""
import sys; sys.path.insert(0, '/root/repo')
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else 'M'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(); rl = cx.upload(c, A, b)
for i in range(reps):
    r = rl.solve(0.0)
s = r.stats
print(name, lp.STATUS_NAMES[r.status], 'pivots', s['pivots_phase2'], 'total %.2f ms loop %.2f final %.2f (dev %.2f host %.2f) rounds %d dense %d' % (
    s['seconds_total'] * 1e3, s['seconds_pivot_loop'] * 1e3, s['seconds_final_solve'] * 1e3, s['seconds_final_device'] * 1e3, s['seconds_final_host'] * 1e3, s['lu_rounds'], s['lu_dense_steps']))
cx.close()
