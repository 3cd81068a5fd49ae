"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py DIR [name-substring]."""
import csv, glob, sys, collections, json
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub and sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print(k[:100])
    for c in sorted(acc[k]): print("   %-32s %16.0f  (per launch, %d launches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
