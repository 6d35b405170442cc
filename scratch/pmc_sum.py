"""Sums the counters of one rocprofv3 --pmc pass per kernel family: usage pmc_sum.py <pass dir> [--json]"""
import csv, glob, json, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)


def family(k):
    m = re.match(r"(?:void )?(?:mer::)?([A-Za-z_0-9]+)", k)
    return m.group(1) if m else k[:40]


import os
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
for f in files[-1:]:          # one pass = one process; a directory merged from several gpurun calls keeps older files: newest only
    for r in csv.DictReader(open(f)):
        k = family(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
if "--json" in sys.argv:
    print(json.dumps({k: {c: {"sum": v, "dispatches": n[(k, c)]} for c, v in cs.items()} for k, cs in acc.items()}))
else:
    for k in sorted(acc):
        print(k)
        for c, v in sorted(acc[k].items()):
            print("   %-44s %.6g   (dispatches %d)" % (c, v, n[(k, c)]))
