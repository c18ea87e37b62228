import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "expand_kernel" in r["Kernel_Name"]]
    byc = collections.defaultdict(list)
    for r in rows:
        byc[r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for c, v in byc.items():
        tot = sum(x[1] for x in v)
        wsum = sum(g * x for g, x in v) / max(1, sum(g for g, _ in v))
        top = sorted(v, reverse=True)[:6]
        print("%-34s n=%d sum=%.6g grid-weighted-mean=%.6g top:" % (c, len(v), tot, wsum), " ".join("%d:%.5g" % t for t in top))
