"""Per-kernel sums of the counters in rocprofv3 --pmc counter_collection.csv files.
Usage: pmc_summary.py file.csv [file.csv ...] [--filter name]"""
import collections, csv, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = None
if "--filter" in sys.argv:
    flt = sys.argv[sys.argv.index("--filter") + 1]
    args = [a for a in args if a != flt]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for fn in args:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        if flt and flt not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((fn, r["Dispatch_Id"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        print(f"    {c:34s} {d[c]:.6g}")
