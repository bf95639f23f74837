"""Developer tool: wide waves again and again — 256-, 2048-wide waves of the C5 root on one pool (virtual first block, two schedules, persistent
launches), quiet and with a second pool solving waves on the same GPU from another thread: every wave's status / z / x must be bit-identical to
the first one's.  Says how many relaxations were handed to a worker (a persistent launch that gave up a wait): same bits, counted.
usage: soak_wave.py [rounds]"""
import sys, os, time, threading; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m, seed = synth.CONFIGS["C5"]
c, A, b = synth.dense_lp_standard_form(m, seed)
mask = synth.integrality_mask(m, m)
cx = lp.Context(); root = cx.upload(c, A, b).solve(0.0); cx.close()
waves = {nv: lp.pack_children(synth.frontier_children(root.x, mask, nv)) for nv in (8, 11)}
pool = lp.FrontierPool(workers=4); pool.set_root(c, A, b)
ref = {nv: pool.solve(w) for nv, w in waves.items()}
bad = fb = 0
def check(nv, r):
    global bad, fb
    o = ref[nv]
    ok = o.status == 0
    same = np.array_equal(r.status, o.status) and np.array_equal(r.z[ok], o.z[ok]) and np.array_equal(r.x[ok], o.x[ok])
    bad += 0 if same else 1
    fb += r.stats["host_fallbacks"]
t0 = time.perf_counter()
for i in range(rounds):
    for nv in (8, 8, 11):
        check(nv, pool.solve(waves[nv]))
print("quiet: %d waves, %.1f ms per round, mismatching waves %d, relaxations handed to a worker %d" % (3 * rounds, 1e3 * (time.perf_counter() - t0) / rounds, bad, fb), flush=True)
stop = False
other = lp.FrontierPool(workers=2); other.set_root(c, A, b)
def noise():
    while not stop:
        other.solve(waves[8])
th = threading.Thread(target=noise); th.start()
bad0, fb0 = bad, fb
t0 = time.perf_counter()
for i in range(rounds):
    for nv in (8, 8, 11):
        check(nv, pool.solve(waves[nv]))
stop = True; th.join()
print("beside a second pool on the same GPU: %d waves, %.1f ms per round, mismatching waves %d, relaxations handed to a worker %d" % (3 * rounds, 1e3 * (time.perf_counter() - t0) / rounds, bad - bad0, fb - fb0), flush=True)
other.close(); pool.close()
sys.exit(1 if bad else 0)
