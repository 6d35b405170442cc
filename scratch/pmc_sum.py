import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-44s %.6g   (dispatches %d)" % (c, v, n[(k, c)]))
