#!/bin/bash
# Counter-based matrix-pipe utilisation of the two dominant kernels INSIDE the pipeline (one batch of bench.py's default workload):
#   scripts/pmc_mfma_util.sh r04
# Two --pmc passes (8 SQ slots each; GRBM is independent), never combined with sys / hip / hsa trace domains.
# Writes gpurun_out/pmc_<tag>/pmc_mfma_util.{txt,json}; copy both to profiles/<tag>_pmc_mfma_util.* and commit.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
grep -o "SQ_[A-Z0-9_]*" "$OUT/counters_available.txt" | sort -u > "$OUT/sq_counter_names.txt"
WORK=${PMC_WORK:-"python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs"}
run() { echo "== $*" >&2; "$@"; }
run rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --kernel-trace -d "$OUT" -o pass1 -- $WORK > "$OUT/pass1.log" 2>&1
run rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --kernel-trace -d "$OUT" -o pass2 -- $WORK > "$OUT/pass2.log" 2>&1
python "$R/scripts/rocpd_mfma_util.py" "$OUT/pmc_mfma_util.json" "$OUT/pass1_results.db" "$OUT/pass2_results.db" > "$OUT/pmc_mfma_util.txt" 2>&1
rm -f "$OUT"/*_results.db
tail -5 "$OUT"/pass1.log "$OUT"/pass2.log >&2
cat "$OUT/pmc_mfma_util.txt" >&2
