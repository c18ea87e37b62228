import csv, glob, sys, collections, json
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-72:]
        key = (k, row["Counter_Name"])
        agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
    out = {"%s|%s" % k: {"sum": v[0], "n": v[1]} for k, v in agg.items()}
    print(json.dumps(out))
