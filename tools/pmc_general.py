import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in acc:
    if "gsb" in k: print(k, "launches", n[k], {c: v for c, v in acc[k].items()})
