#!/bin/bash
# Runs ON the GPU box: SQ counters of one command's kernels, in passes of at most eight counters
# (program directly after --, every pass under its own timeout).  Exit status 1 if a pass failed.
# Usage: bash tools/pmc_kernel.sh <outdir> <python script> [args...]
set -o pipefail
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
CMD=("$@")
n=0
failed=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU"; do
  n=$((n+1))
  echo "[pmc] pass $n: $set"
  if ! timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pmc$n" -o p -- python3 "${CMD[@]}" > "$out/pmc$n.log" 2>&1; then
    echo "[pmc] pass $n failed"; tail -3 "$out/pmc$n.log"; failed=1
  fi
done
python3 tools/pmc_summary.py "$out"/pmc*/p_counter_collection.csv > "$out/summary.txt" || failed=1
echo "[pmc] done (failed=$failed)"
exit $failed
