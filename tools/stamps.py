"""Developer tool: where a pivot of the block kernel spends its cycles (diagnostic build, context knob bt_stamps).
usage: python tools/stamps.py [M|C2|C3 ...]   -> one JSON line per solve on stderr (engine) + a summary table"""
import sys, json, os, subprocess; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth
SEG = ["r+wavemin A", "barrier A", "blockmin A", "col load", "col fma", "ratio+wavemin B", "barrier B", "blockmin B", "row load", "row fma/r/V", "u/xb/U/commit"]
for name in (sys.argv[1:] or ["M", "C2", "C3"]):
    for nt in (0, 1024):
        m, seed = synth.CONFIGS[name]
        c, A, b = synth.dense_lp_standard_form(m, seed)
        cx = lp.Context(bt_stamps=1, bt_nt=nt)
        p = cx.upload(c, A, b)
        r = p.solve(0.0)
        print(name, "nt", nt, "status", r.status, "pivots", r.stats["pivots_phase2"], "loop_ms %.2f" % (1e3 * r.stats["seconds_pivot_loop"]), flush=True)
        cx.close()
