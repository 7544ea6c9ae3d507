"""Timeline of the last `--steps` repetitions in a rocprofv3 kernel trace: per kernel (in launch order within a
step) its duration and the idle gap in front of it on the device.  Usage: trace_gaps.py trace.csv anchor_kernel"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2]
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
lo, hi = idx[-3], idx[-2]                       # one full step, between two launches of the anchor
end_prev = max(int(r["End_Timestamp"]) for r in rows[:lo])
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - end_prev) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:80]}")
    end_prev = max(end_prev, e)
print("step span", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, "us")
