#!/bin/bash
# Tile-walk / MFMA-shape study of gemm_bf16_t256:  scripts/pmc_gemm_walk.sh <tag>     (env: SHAPES GROUP_MS MFMAS STORE_NT LAUNCHES; not GROUPS: bash keeps that name for itself)
# one un-profiled interleaved timing pass, then ONE --pmc pass (FETCH_SIZE + GRBM_GUI_ACTIVE, --kernel-trace only).
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/gemm_walk_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TIMED=1 python "$R/scripts/pmc_gemm_walk.py" "$OUT/plan.json" 2>&1 | grep -v amdgpu.ids > "$OUT/timed.txt"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace -d "$OUT" -o walk -- python "$R/scripts/pmc_gemm_walk.py" "$OUT/plan.json" > "$OUT/pass.log" 2>&1
python "$R/scripts/rocpd_gemm_walk.py" "$OUT/walk_results.db" "$OUT/plan.json" "$OUT/walk.json" > "$OUT/walk.txt" 2>&1
rm -f "$OUT"/*_results.db
tail -3 "$OUT/pass.log"; cat "$OUT/timed.txt" "$OUT/walk.txt"
