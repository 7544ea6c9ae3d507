"""Per-kernel sums of the counters in a rocprofv3 --pmc counter_collection.csv.  Usage: pmc_summary.py file.csv [name filter]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
