"""Turn the rocprofv3 --pmc passes of tools/wave_prof.py (FETCH_SIZE, WRITE_SIZE: separate runs of the C5 wave ALONE, tools/prof_cmds.sh step 4)
into profiles/<tag>_pmc_traffic_C5.json: bytes at the fabric per kernel and WAVE (sum over the launches of the pass / waves of the pass).
    python tools/pmc_traffic_wave.py <fetch_csv> <write_csv> <tag> <waves>"""
import csv, json, sys, collections
fetch_csv, write_csv, tag, waves = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

f = per_kernel(fetch_csv, "FETCH_SIZE"); w = per_kernel(write_csv, "WRITE_SIZE")
docs = []
for k in sorted(set(f) | set(w), key=lambda k: -(2 * sum(f.get(k, [0])) + sum(w.get(k, [0])))):
    fk, wk = f.get(k, []), w.get(k, [])
    tot = (2 * sum(fk) + sum(wk)) * 1024.0
    if tot < 1e6:
        continue
    docs.append({"kernel": k, "launches_per_wave": max(len(fk), len(wk)) / waves, "FETCH_SIZE_KB_raw_per_wave": sum(fk) / waves, "WRITE_SIZE_KB_per_wave": sum(wk) / waves,
                 "traffic_bytes_per_wave": tot / waves,
                 "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact; separate --pmc passes"})
json.dump({"workload": "C5 wave (256 children of the 512x1024 root), %d waves per pass, tools/wave_prof.py" % waves, "kernels": docs}, open("profiles/%s_pmc_traffic_C5.json" % tag, "w"), indent=1)
for d in docs[:12]:
    print("%-70s %6.1f launches/wave %9.1f MB/wave" % (d["kernel"][:70], d["launches_per_wave"], d["traffic_bytes_per_wave"] / 1e6))
