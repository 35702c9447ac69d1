#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the headline command            -> $OUT/trace/
#   2. three separate --pmc passes (SQ/GRBM; FETCH_SIZE; WRITE_SIZE) of a short run -> $OUT/pmc/
# then tools/kernel_stats_summary.py and tools/pmc_summary.py turn the CSVs into the markdown kept under profiles/, and
# pmc_summary.py writes bench_pmc_traffic.json (HBM bytes per launch per kernel + a hash of csrc/): bench.py's roofline.traffic.
# Usage: bash tools/profile_bench.sh gpurun_out/prof_v9 [tag]
set -e -o pipefail
OUT=${1:-gpurun_out/prof}
TAG=${2:-r1}
ROOT=$PWD
mkdir -p "$OUT/trace" "$OUT/pmc"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- \
  python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --plain-only > "$OUT/trace.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  -d "$OUT/pmc" -o sq --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --plain-only > "$OUT/pmc/sq.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc" -o fetch --output-format csv -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --plain-only > "$OUT/pmc/fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc" -o write --output-format csv -- \
  python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --plain-only > "$OUT/pmc/write.log" 2>&1
python3 tools/kernel_stats_summary.py "$OUT/trace" 7 > "$OUT/${TAG}_bench_summary.md"
python3 tools/pmc_summary.py "$OUT/pmc" 3 --json bench_pmc_traffic.json --workload fwd_b32_l64 --profile "profiles/${TAG}_pmc_per_kernel.md" \
  > "$OUT/${TAG}_pmc_per_kernel.md"
cp bench_pmc_traffic.json "$OUT/"   # gpurun merges only gpurun_out/ back: copy it to the repo root afterwards and commit it
grep -h '"metric"' "$OUT/trace.log" | cut -c1-400
