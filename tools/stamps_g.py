"""Developer tool: cycles per pivot segment of the multi-workgroup block kernel (diagnostic build: knobs bt_groups + bt_stamps).
usage: python tools/stamps_g.py [M|C2 ...] [G]   (run with 2>&1: the engine prints the sums on stderr)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "M"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print("segments: 0 local-min A | 1 barrier | 2 block-min + post | 3 poll | 4 pick | 5 barrier || 6 column + ratios + local-min B | 7 barrier | 8 post | 9 poll | 10 pick | 11 barrier || 12 row + updates")
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
knobs = dict((k, int(v)) for k, v in (a.split("=") for a in sys.argv[3:]))
print("loop kernel (default at M): 0 local-min A | 1 wait+barrier | 2 combine+post | 3 poll | 4 pick | 11 column loads | 5 corrections+ratios+local-min B | 6 wait+barrier | 7 combine+post | 8 poll | 9 pick | 12 row loads | 10 rest of the row phase")
cx = lp.Context(bt_stamps=1, bt_groups=G, **knobs)
r = cx.upload(c, A, b).solve(0.0)
print(name, "G", G, "status", r.status, "pivots", r.stats["pivots_phase2"], "loop_ms %.2f" % (1e3 * r.stats["seconds_pivot_loop"]), flush=True)
cx.close()
