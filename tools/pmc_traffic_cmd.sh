#!/bin/bash
# Runs ON the GPU box: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, KB per launch at the L2's fabric
# side) of every kernel of one command.  Usage: bash tools/pmc_traffic_cmd.sh <outdir> <python script> [args...]
set -o pipefail
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o t -- python3 "$@" > "$out/$c.log" 2>&1 || { echo "$c pass failed"; tail -3 "$out/$c.log"; exit 1; }
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]] += 1
        for k in acc:
            res[k][c + "_MB_per_launch"] = round(acc[k] / cnt[k] * 1024 / 1e6, 2)
            res[k]["launches"] = cnt[k]
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -sum(x for n, x in kv[1].items() if n != "launches"))[:10]:
    print(f"{v.get('FETCH_SIZE_MB_per_launch', 0):12.1f} {v.get('WRITE_SIZE_MB_per_launch', 0):12.1f} {v['launches']:4d}  {k[:100]}")
PY
