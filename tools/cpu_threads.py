"""Developer tool: CPU oracle pivots/s at the metric size for several OpenMP thread counts."""
import sys, time, os; sys.path.insert(0, '/root/repo')
from gomilp_amd import synth
from oracle import oracle as O
m, seed = synth.CONFIGS['M']
c, A, b = synth.dense_lp_standard_form(m, seed)
for t in (8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 1): break
    O.set_threads(t)
    t0 = time.perf_counter()
    r = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, stop_after_pivots=2)
    print(t, 'threads', r.pivots_phase2 / r.seconds_loop, 'pivots/s', time.perf_counter() - t0, 's wall', flush=True)
