"""Developer tool: the loop kernel's pivot role with replicated reduced costs (btr_kernels.hip k_bt_loopR, knob loop_rep) against the
oracle fixtures of the metric LP and of C2, and what it costs beside the two-exchange role (k_bt_loop).
usage: rep_check.py [M] [C2] [reps] [key=value,key=value ...]   (each such argument is one configuration; default: loop_rep=1 and loop_rep=0)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp, synth

HERE = os.path.dirname(os.path.abspath(__file__))
names = [a for a in sys.argv[1:] if not a.isdigit() and "=" not in a] or ["M", "C2"]
cfgs = [dict((k, int(v)) for k, v in (kv.split("=") for kv in a.split(","))) for a in sys.argv[1:] if "=" in a] or [{"loop_rep": 1}, {"loop_rep": 0}]
reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 4)
bad = 0
for name in names:
    fx = np.load(os.path.join(os.path.dirname(HERE), "tests", "golden", "lp_%s.npz" % name))
    m, seed = synth.CONFIGS[name]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    for cfg in cfgs:
        cx = lp.Context(**cfg)
        p = cx.upload(c, A, b)
        r = p.solve(0.0, trace=True)
        got = np.array(r.pivots, dtype=np.int64).reshape(-1, 6)[:, [0, 2, 3, 4, 5]]
        want = fx["trace"][:, [0, 2, 3, 4, 5]].astype(np.int64)
        same = got.shape == want.shape and np.array_equal(got, want)
        first = -1 if same else (int(np.argmax((got[: min(len(got), len(want))] != want[: min(len(got), len(want))]).any(axis=1))) if len(got) and len(want) else 0)
        bits = r.status == 0 and np.array_equal(r.x, fx["x"]) and r.z == float(fx["z"])
        ts, loops = [], []
        for _ in range(reps):
            t0 = time.perf_counter(); r2 = p.solve(0.0); ts.append(time.perf_counter() - t0); loops.append(r2.stats["seconds_pivot_loop"])
            bits = bits and np.array_equal(r2.x, fx["x"])
        if not (same and bits): bad += 1
        print("%s %s: status %d pivots %d (oracle %d) trace %s%s, x / z bits %s | solve best %.3f ms, pivot loop best %.3f ms = %.3f us per pivot, retries %d" % (
            name, cfg, r.status, len(got), len(want), "identical" if same else "DIFFERENT", "" if same else " (first difference at pivot %d)" % first,
            "identical" if bits else "DIFFERENT", 1e3 * min(ts), 1e3 * min(loops), 1e6 * min(loops) / max(1, len(want)), r.stats.get("device_retries", 0)), flush=True)
        cx.close()
print("mismatches %d" % bad)
sys.exit(1 if bad else 0)
