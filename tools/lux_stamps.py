"""Developer tool (diagnostic flavour: GOMILP_DEBUG_BUILD=1): cycles per segment of a dense step of the cross-workgroup LU panel (lu_cross.hip).
usage: GOMILP_DEBUG_BUILD=1 python tools/lux_stamps.py C3"""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
cx = lp.Context(lu_cross=1, lu_blocked=2); p = cx.upload(c, A, b)
for _ in range(3):
    r = p.solve(0.0)
buf = (ctypes.c_ulonglong * 64)()
lp.lib().gomilp_debug_lux_stamps(buf)
a = np.array(buf[:], dtype=np.float64).reshape(4, 16)
steps = a[0, 15]
names = ["wave0 top", "barrier1", "retire", "own search", "barrier2", "local pick+post+poll", "pick", "bookkeeping", "elimination"]
print(name, "dense steps", int(steps), "final solve %.3f ms, rounds %d" % (1e3 * r.stats["seconds_final_solve"], r.stats["lu_rounds"]))
for w in range(4):
    print("wave", w, " ".join("%s %.0f" % (names[i], a[w, i] / steps) for i in range(9)), "| sum %.0f cycles per dense step" % (a[w, :9].sum() / steps))
cx.close()
