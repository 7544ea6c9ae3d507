#!/bin/bash
# Runs ON the GPU box: LDS conflict counters + durations of bench.py's kernels for the shipped library and the
# developer builds given as arguments (paths of .so files).  Usage: bash tools/pmc_lds.sh <outdir> [lib.so ...]
set -o pipefail
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
for lib in "" "$@"; do
  tag=$(basename "${lib:-shipped}" .so)
  [ -n "$lib" ] && export PLSR_LIB="$root/$lib" || unset PLSR_LIB
  timeout -k 10 200 python3 bench.py --steps 60 --no-cpu --no-pls-call --no-ceiling > "$out/$tag.bench.log" 2>&1 || { echo "bench failed for $tag"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d "$out/$tag" -o p -- \
      python3 bench.py --steps 5 --warmup 1 --no-cpu --no-pls-call --no-ceiling > "$out/$tag.pmc.log" 2>&1 || { echo "pmc failed for $tag"; tail -3 "$out/$tag.pmc.log"; exit 1; }
  echo "== $tag"
  python3 - "$out/$tag.bench.log" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("   boot launch %.3f ms  perm launch %.3f ms  step %.3f ms" % (d["roofline"]["avg_launch_ms"], d["roofline"]["perm_kernel"]["avg_launch_ms"], d["ms_per_step"]))
PY
  python3 tools/pmc_summary.py $(find "$out/$tag" -name "*counter_collection.csv") --filter project_boot_reg | sed 's/^/   /'
done
