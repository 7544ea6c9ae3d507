#!/bin/bash
# Runs ON the GPU box: kernel trace + SQ counters of the rb bootstrap kernels (config 3).
# Usage: gpurun -- 'bash tools/collect_cfg3.sh r02a'   then  python tools/summarise_cfg3.py r02a
set -o pipefail
tag=${1:-rXX}
count=${2:-1000}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/cfg3_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
echo "[cfg3] plain run"
timeout -k 10 300 python3 bench_configs.py --config 3 --count $count > "$out/run.log" 2>&1 || exit 1
grep '^{' "$out/run.log" | tail -1 > "$out/${tag}_cfg3.json"
echo "[cfg3] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o c3 -- \
    python3 bench_configs.py --config 3 --count $count > "$out/stats.log" 2>&1 || exit 1
echo "[cfg3] SQ counters"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq" -o c3 -- \
    python3 bench_configs.py --config 3 --count 250 > "$out/pmc_sq.log" 2>&1 || exit 1
echo "[cfg3] done"
