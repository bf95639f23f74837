"""Turn the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, see profiles/README.md; optionally an MFMA pass)
into profiles/<tag>_pmc_traffic.json (one entry per kernel of interest) + a merged per-kernel CSV.
    python tools/pmc_traffic.py <fetch_csv> <write_csv> <tag> [<mfma_csv>]"""
import csv, json, sys, collections
fetch_csv, write_csv, tag = sys.argv[1:4]
mfma_csv = sys.argv[4] if len(sys.argv) > 4 else None

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        big = [x for x in v if x > 0.5 * max(v)] or v   # launches behind the device's `done` flag exit at once: not samples
        out[k] = (sum(big) / len(big), len(big), sum(v), len(v))
    return out

f = per_kernel(fetch_csv, "FETCH_SIZE"); w = per_kernel(write_csv, "WRITE_SIZE")
rows = []
for k in sorted(set(f) | set(w)):
    rows.append((k, f.get(k, (0, 0, 0, 0))[1], f.get(k, (0, 0, 0, 0))[0], w.get(k, (0, 0, 0, 0))[0]))
with open("profiles/%s_pmc_counters.csv" % tag, "w") as out:
    out.write("kernel,launches,FETCH_SIZE_mean_KB_raw,WRITE_SIZE_mean_KB\n")
    for k, n, fk, wk in rows: out.write('"%s",%d,%.3f,%.3f\n' % (k, n, fk, wk))
m, nn = 2048, 2048
alg = {"k_bt_loop": 64 * (16.0 * m * nn + 8 * 16.0 * (m + nn) + 24.0 * (m + nn)), "k_bt_update_mfma16": 16.0 * m * nn, "k_bt_innerG<8, 256": 16 * 16.0 * (m + nn) + 24.0 * (m + nn)}
docs = []
for key, note in (("k_bt_loop", "persistent loop kernel, mean over the FULL launches (64 blocks of 8 pivots: the rank-8 update reads and writes the 33.6 MB tableau once "
                                "per block, 16-byte agent-scope accesses; the pivot workgroups add one column + one row per pivot and the exchange records).  The tableau "
                                "pair (67 MB) lives in the 256 MB Infinity Cache between blocks, so FETCH_SIZE / WRITE_SIZE (traffic at the fabric) can sit below the algorithmic bytes"),
                  ("k_bt_update_mfma16", "streaming rank-16 update on the matrix cores: 8-byte loads / stores, 512 contiguous bytes per instruction"),
                  ("k_bt_innerG<8, 256", "block kernel, 8 workgroups of one XCD: 8-byte reads of 128-byte tile lines (one column + one row of T per pivot) + the exchange records; FETCH_SIZE is "
                                       "UNCALIBRATED for this shape (MI355X_MICROARCH.md: only wide coalesced reads are known to report 1/2) — the x2 figure is an upper bound"),
                  ("k_bt_inner2_batch", "batched block kernel (C5 wave): mean over launches with 1..256 active relaxations"),
                  ("k_bt_update_tiled_batch", "batched rank-8 update (C5 wave): mean over launches with 1..256 active relaxations")):
    hit = [r for r in rows if key in r[0] and "batch" not in r[0].replace(key, "")] if "batch" not in key else [r for r in rows if key in r[0]]
    if not hit:
        continue
    r = hit[0]
    d = {"kernel": r[0], "launches": r[1], "FETCH_SIZE_mean_KB_raw": r[2], "WRITE_SIZE_mean_KB": r[3],
         "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact; separate --pmc passes",
         "traffic_bytes_per_launch": (2 * r[2] + r[3]) * 1024.0, "note": note}
    if key in alg:
        d["algorithmic_bytes_per_launch"] = alg[key]
    docs.append(d)
cal = [r for r in rows if "k_transpose_in" in r[0]]
if cal and docs:
    docs[0]["calibration"] = "k_transpose_in reads the 67.1 MB A of the metric LP once: FETCH_SIZE %.0f KB raw -> %.1f MB doubled" % (cal[0][2], 2 * cal[0][2] * 1024 / 1e6)
if mfma_csv:
    tot = collections.defaultdict(float)
    perk = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(mfma_csv)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        perk[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    docs.append({"kernel": "(all kernels of the run)", "mfma_counters_sum": dict(tot)})
    for k, cs in perk.items():
        if "k_bt_update_mfma16" in k or "k_bt_loop" in k:
            big = {c: [x for x in v if x > 0.5 * max(v)] or v for c, v in cs.items()}
            docs.append({"kernel": k, "mfma_counters_mean_per_launch": {c: sum(v) / len(v) for c, v in big.items()},
                         "note": "rank-16 update of T: (m/16) * ((n-m)/16) * 4 v_mfma_f64_16x16x4_f64 per launch" if "mfma16" in k else
                                 "rank-8 update per block: (m/16) * ((n-m)/16) * 2 v_mfma_f64_16x16x4_f64, 64 blocks per full launch"})
json.dump(docs, open("profiles/%s_pmc_traffic.json" % tag, "w"), indent=1)
print(json.dumps(docs, indent=1))
