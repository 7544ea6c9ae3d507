"""Summarise tools/collect_cfg3.sh output (gpurun_out/cfg3_<tag>) into profiles/:
<tag>_cfg3_kernel_stats.csv and <tag>_cfg3_pmc_summary.json (per-kernel sums of the SQ counters and
the derived MFMA-busy / wait fractions).  Usage: python tools/summarise_cfg3.py r02a"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"cfg3_{tag}")
dst = os.path.join(root, "profiles")
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_cfg3_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, f"{tag}_cfg3.json")):
    shutil.copy(f, os.path.join(dst, f"{tag}_cfg3.json"))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
out = {}
for k, d in acc.items():
    if not any(t in k for t in ("item_", "latent_", "project_", "gram_")):
        continue
    e = dict(d)
    e["dispatches"] = len(cnt[k])
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    bc = d.get("SQ_BUSY_CYCLES", 0.0)
    if wc:
        e["wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / wc
        e["wait_inst_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0.0) / wc
        e["active_inst_any_frac_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    if d.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if bc:
        # SQ_BUSY_CYCLES is summed over the SEs/XCDs that report it; SQ_VALU_MFMA_BUSY_CYCLES over SIMDs.
        # The ratio below is a RELATIVE figure between kernels / versions of one kernel.
        e["mfma_busy_over_sq_busy"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / bc
    out[k] = e
json.dump({"note": "rocprofv3 --pmc (SQ counters only, own pass, program directly after --); sums over all "
                   "dispatches of bench_configs.py --config 3 --count 250 (both timed runs + cold run)",
           "kernels": out}, open(os.path.join(dst, f"{tag}_cfg3_pmc_summary.json"), "w"), indent=1)
print("wrote", len(out), "kernels")
