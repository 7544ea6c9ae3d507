#!/bin/bash
# Runs ON the GPU box (through gpurun): collects everything DESIGN.md / bench.py cite
# under gpurun_out/collect/.  tools/summarise_profiles.py then copies the summaries
# into profiles/ (tracked).  Usage:  gpurun -- 'bash tools/collect_profiles.sh r01b'
set -o pipefail
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/collect
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
echo "[collect] bench with cpu baseline"
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > "$out/bench.log" 2>&1 || exit 1
grep '^{' "$out/bench.log" | tail -1 > "$out/${tag}_bench.json"
echo "[collect] kernel trace + stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu --no-pls-call --no-ceiling > "$out/stats.log" 2>&1 || exit 1
echo "[collect] PMC pass 1 (FETCH_SIZE)"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pls-call --no-ceiling > "$out/pmc_fetch.log" 2>&1 || exit 1
echo "[collect] PMC pass 2 (WRITE_SIZE)"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o bench -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pls-call --no-ceiling > "$out/pmc_write.log" 2>&1 || exit 1
for c in 3 4 5 6; do
  echo "[collect] config $c"
  timeout -k 10 400 python3 bench_configs.py --config $c > "$out/cfg$c.log" 2>&1 || exit 1
  grep '^{' "$out/cfg$c.log" | tail -1 >> "$out/${tag}_configs_3_4_5_single_gpu.jsonl"
done
echo "[collect] config 3 kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats3" -o c3 -- \
    python3 bench_configs.py --config 3 > "$out/stats3.log" 2>&1 || exit 1
echo "[collect] config 4 kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats4" -o c4 -- \
    python3 bench_configs.py --config 4 > "$out/stats4.log" 2>&1 || exit 1
echo "[collect] config 6 kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats6" -o c6 -- \
    python3 bench_configs.py --config 6 > "$out/stats6.log" 2>&1 || exit 1
echo "[collect] done"
