"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, see profiles/README.md) into
profiles/<tag>_pmc_traffic.json + a merged per-kernel CSV.   python tools/pmc_traffic.py <fetch_csv> <write_csv> <tag>"""
import csv, json, sys, collections
fetch_csv, write_csv, tag = sys.argv[1:4]
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        big = [x for x in v if x > 0.5 * max(v)] or v   # launches behind the device's `done` flag exit at once: not samples
        out[k] = (sum(big) / len(big), len(big))
    return out
f = per_kernel(fetch_csv, "FETCH_SIZE"); w = per_kernel(write_csv, "WRITE_SIZE")
rows = []
for k in sorted(set(f) | set(w)):
    rows.append((k, f.get(k, (0, 0))[1], f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]))
with open("profiles/%s_pmc_counters.csv" % tag, "w") as out:
    out.write("kernel,launches,FETCH_SIZE_mean_KB_raw,WRITE_SIZE_mean_KB\n")
    for k, n, fk, wk in rows: out.write('"%s",%d,%.3f,%.3f\n' % (k, n, fk, wk))
upd = [r for r in rows if "k_bt_update_tiled" in r[0]][0]
cal = [r for r in rows if "k_transpose_in" in r[0]]
m, nn = 2048, 2048
doc = {"workload": "M: 2048x4096 dense LP seed 2, pipeline blocked (K=8), T in 4x4 tiles",
       "kernel": "k_bt_update_tiled",
       "launches": upd[1],
       "FETCH_SIZE_mean_KB_raw": upd[2], "WRITE_SIZE_mean_KB": upd[3],
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact; separate --pmc passes",
       "traffic_bytes_per_launch": (2 * upd[2] + upd[3]) * 1024.0,
       "algorithmic_bytes_per_launch": 16.0 * m * nn}
if cal:
    doc["calibration"] = "k_transpose_in reads the 67.1 MB A once: FETCH_SIZE %.0f KB raw -> %.1f MB doubled" % (cal[0][2], 2 * cal[0][2] * 1024 / 1e6)
json.dump(doc, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(doc, indent=1))
