#!/bin/bash
# Profiles of one round, on the GPU box:  scripts/profile_round.sh r02
# Writes gpurun_out/prof_<tag>/ (merged back by gpurun); the summaries (*.txt, *.json) are then copied into profiles/ and committed.
#   kernel_stats_bench_default.txt     rocprofv3 --kernel-trace of `bench.py` (generate workload), per-kernel calls / avg / share
#   kernel_stats_bench_retrieval.txt   the same for `bench.py --workload retrieval` (ip_scan_kernel, select_kernel, the f32 CLIP tower)
#   pmc_traffic.json + pmc_passes_summary.txt   FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md §HBM), FETCH doubled
# PMC passes never carry sys / hip / hsa trace domains (gpurun refuses that combination).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*" >&2; "$@"; }
run rocprofv3 --kernel-trace -d "$OUT" -o bench -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-side-configs > "$OUT/bench_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_stats.py" "$OUT/bench_results.db" > "$OUT/kernel_stats_bench_default.txt" 2>&1
# the timed region alone (after the warm-up batch's last kernel): what runs PER BATCH, without model construction
python "$R/scripts/rocpd_stats.py" "$OUT/bench_results.db" image_postprocess 1 > "$OUT/kernel_stats_bench_default_timed_region.txt" 2>&1
run rocprofv3 --kernel-trace -d "$OUT" -o retr -- python "$R/bench.py" --workload retrieval --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/retrieval_under_rocprof.log" 2>&1
python "$R/scripts/rocpd_stats.py" "$OUT/retr_results.db" > "$OUT/kernel_stats_bench_retrieval.txt" 2>&1
run rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o fetch -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs > "$OUT/pmc_fetch.log" 2>&1
run rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT" -o write -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-side-configs > "$OUT/pmc_write.log" 2>&1
python "$R/scripts/rocpd_pmc.py" "$OUT/fetch_results.db" "$OUT/write_results.db" "$OUT/pmc_traffic.json" > "$OUT/pmc_passes_summary.txt" 2>&1
# retrieval scan: FETCH_SIZE calibrated on the kernel's own access pattern (guide: "calibrate on a known byte count in your own access
# pattern"): one pass over scan-only launches at N = 1 000 000 (2.05 GB per launch, 8x the Infinity Cache) gives the factor, the passes
# over the N = 118 287 shape are scaled by it.  One shape per pass (round 2 averaged three corpus sizes into one figure).
run rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o cfetch -- python "$R/scripts/pmc_scan_shape.py" 1000000 1 scan > "$OUT/pmc_cfetch.log" 2>&1
run rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o rfetch -- python "$R/scripts/pmc_scan_shape.py" 118287 16 scan > "$OUT/pmc_rfetch.log" 2>&1
run rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT" -o rwrite -- python "$R/scripts/pmc_scan_shape.py" 118287 16 scan > "$OUT/pmc_rwrite.log" 2>&1
python "$R/scripts/rocpd_pmc_calib.py" "$OUT/cfetch_results.db" 1000000 "$OUT/rfetch_results.db" "$OUT/rwrite_results.db" 118287 16 "$OUT/pmc_traffic_retrieval.json" > "$OUT/pmc_passes_summary_retrieval.txt" 2>&1
# the same for the Q = 64 one-pass scan (four query tiles per row range: HBM traffic must stay one corpus pass)
run rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o r64fetch -- python "$R/scripts/pmc_scan_shape.py" 1000000 64 scan > "$OUT/pmc_r64fetch.log" 2>&1
run rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT" -o r64write -- python "$R/scripts/pmc_scan_shape.py" 1000000 64 scan > "$OUT/pmc_r64write.log" 2>&1
python "$R/scripts/rocpd_pmc_calib.py" "$OUT/cfetch_results.db" 1000000 "$OUT/r64fetch_results.db" "$OUT/r64write_results.db" 1000000 64 "$OUT/pmc_q64_one_pass.json" > "$OUT/pmc_q64_one_pass.txt" 2>&1
# the databases are tens of MB each and gpurun merges at most 64 MiB back: keep the summaries, drop the raw traces
rm -f "$OUT"/*_results.db
ls -la "$OUT" >&2
tail -3 "$OUT"/*.txt >&2
