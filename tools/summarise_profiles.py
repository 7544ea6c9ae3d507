"""Copy what tools/collect_profiles.sh produced (gpurun_out/collect) into profiles/:
bench line, kernel-stats CSVs, per-kernel PMC traffic and profiles/hbm_traffic.json
(the figure bench.py reports as roofline.traffic).  Usage: python tools/summarise_profiles.py r01b"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "collect")
dst = os.path.join(root, "profiles")


def copy(pattern, name):
    hits = glob.glob(os.path.join(src, pattern))
    if hits:
        shutil.copy(hits[0], os.path.join(dst, name))
        print("copied", name)


copy(f"{tag}_bench.json", f"{tag}_bench.json")
copy(f"{tag}_configs_3_4_5_single_gpu.jsonl", f"{tag}_configs_3_4_5_single_gpu.jsonl")
copy("stats/*kernel_stats.csv", f"{tag}_kernel_stats.csv")
copy("stats3/*kernel_stats.csv", f"{tag}_cfg3_kernel_stats.csv")
copy("stats4/*kernel_stats.csv", f"{tag}_cfg4_kernel_stats.csv")
copy("stats6/*kernel_stats.csv", f"{tag}_cfg6_kernel_stats.csv")

pmc = collections.defaultdict(dict)
for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in glob.glob(os.path.join(src, d, "*counter_collection.csv")):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"])
                cnt[r["Kernel_Name"]] += 1
        for k in acc:
            pmc[k][f"{counter}_KB_per_launch"] = acc[k] / cnt[k]
            pmc[k]["launches"] = cnt[k]
if pmc:
    json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
    boot = [k for k in pmc if "project_boot_reg_kernel" in k and "false" in k] or \
           [k for k in pmc if "project_kernel<3, 1" in k]
    perm = [k for k in pmc if "project_perm_reg_kernel" in k]
    if boot:
        b = pmc[boot[0]]
        fetch = b.get("FETCH_SIZE_KB_per_launch", 0.0) * 1024
        write = b.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024
        json.dump({
            "note": "rocprofv3 --pmc, separate passes for FETCH_SIZE and WRITE_SIZE (" + tag + "). Counters are KB "
                    "per launch; the kernel's loads are 8 B/lane, so the guide's x2 FETCH_SIZE correction for "
                    "16 B/lane streams is NOT applied (true read bytes lie between 1x and 2x of the figure).",
            "kernel": boot[0],
            "boot_project_fetch_bytes": fetch,
            "boot_project_write_bytes": write,
            "boot_project_bytes_per_launch": fetch + write,
            **({"perm_kernel": perm[0],
                "perm_project_bytes_per_launch": (pmc[perm[0]].get("FETCH_SIZE_KB_per_launch", 0.0) +
                                                  pmc[perm[0]].get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024}
               if perm else {}),
        }, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
        print("hbm_traffic.json", fetch + write)

# per-launch durations of the projection kernels from the kernel trace (the stats
# CSV averages warm-up launches in; the steady-state launches are what bench.py's
# hipEvents time)
tr = glob.glob(os.path.join(src, "stats", "*kernel_trace.csv"))
if tr:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        if "project_" in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    line = [l for l in open(os.path.join(src, "stats.log")) if l.startswith("{")]
    out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 5 --no-cpu  (8 spin-up + 5 warm-up + 40 timed launches per kernel)",
           "per_launch_ms_in_launch_order": per,
           "bench_line_of_the_same_run": json.loads(line[-1]) if line else None}
    json.dump(out, open(os.path.join(dst, f"{tag}_project_kernel_launches.json"), "w"), indent=1)
    print("wrote per-launch durations")
