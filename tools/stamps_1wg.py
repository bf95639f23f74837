"""Developer tool: cycles per pivot segment of the single-workgroup block kernel k_bt_inner2 (diagnostic instance, knob bt_stamps) on the
512-row C5 root — the kernel the batched schedule and k_b_loop's pivot role are made of.
usage: python tools/stamps_1wg.py [config]   (2>&1: the engine prints the sums on stderr)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gomilp_amd import lp, synth
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
m, seed = synth.CONFIGS[name]
c, A, b = synth.dense_lp_standard_form(m, seed)
print("segments: 0 r + wave first-min + payload | 1 barrier A | 2 block first-min A | 3 column load | 4 column corrections | 5 ratios + wave first-min + payload | 6 barrier B | 7 block first-min B | 8 rest of the row-load latency | 9 row corrections, reduced costs, v' store | 10 row loads issued, u terms, x_B, u store")
cx = lp.Context(bt_stamps=1, bt_groups=-1, bt_lag=0)
r = cx.upload(c, A, b).solve(0.0)
print(name, "status", r.status, "pivots", r.stats["pivots_phase2"], "loop_ms %.2f" % (1e3 * r.stats["seconds_pivot_loop"]), flush=True)
cx.close()
