"""Device busy / idle summary of a rocprofv3 kernel trace: span of the last `phase` (the kernels between the last
two gaps longer than --gap ms), summed kernel time per name, idle time inside it and the largest gaps.
Usage: trace_busy.py trace.csv [--gap 20]"""
import collections, csv, sys
gap_ms = float(sys.argv[sys.argv.index("--gap") + 1]) if "--gap" in sys.argv else 20.0
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# phases: split where the device idles longer than gap_ms
phases, cur, end = [], [], None
for s, e, n in ev:
    if end is not None and s - end > gap_ms * 1e6:
        phases.append(cur)
        cur = []
    cur.append((s, e, n))
    end = e if end is None else max(end, e)
phases.append(cur)
for ph in phases[-3:]:
    t0, t1 = ph[0][0], max(e for _, e, _ in ph)
    busy, last, gaps = 0, t0, []
    for s, e, n in ph:
        if s > last:
            gaps.append(((s - last) / 1e6, n))
        if e > last:
            busy += e - max(s, last)
            last = e
    per = collections.defaultdict(float)
    for s, e, n in ph:
        per[n[:60]] += (e - s) / 1e6
    print(f"phase: {len(ph)} kernels, span {(t1 - t0) / 1e6:.1f} ms, device busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms")
    for n, t in sorted(per.items(), key=lambda kv: -kv[1])[:6]:
        print(f"    {t:8.1f} ms  {n}")
    print("    largest gaps (ms, before kernel):", [(round(g, 2), n[:30]) for g, n in sorted(gaps, reverse=True)[:6]])
