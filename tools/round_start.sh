#!/bin/bash
# First GPU call of a round: everything round 2 could not measure after gpurun closed (DESIGN.md sections 8.0 and 9), in one
# call, every step under its own time limit, stopping at the first failure.  Run on the GPU box from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/round_start.sh gpurun_out/r3_start'
# Leaves under $OUT: gpu_tests.log, bench.json.log (with the fold A/B and the HBM-kernel rows inside the line), bench_nofold,
# attention / norm / train bench tables, the soffset hardware probe, and the rocprofv3 kernel trace + PMC summaries.
set -e -o pipefail
OUT=${1:-gpurun_out/round_start}
mkdir -p "$OUT"
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 tools/probes/soffset_range_check.hip -o /tmp/soff_probe > "$OUT/probe_build.log" 2>&1
timeout -k 10 60 /tmp/soff_probe | tee "$OUT/soffset_probe.txt"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
tail -3 "$OUT/gpu_tests.log"
timeout -k 10 300 python -u bench.py > "$OUT/bench.json.log" 2>&1 || { tail -20 "$OUT/bench.json.log"; exit 1; }
grep -h '"metric"' "$OUT/bench.json.log" | cut -c1-600
LDM_ATTN_V1=1 LDM_BENCH_AB=0 timeout -k 10 300 python -u bench.py --no-cpu-baseline > "$OUT/bench_attn_v1.json.log" 2>&1
timeout -k 10 300 python -u tools/attention_bench.py > "$OUT/attention_bench.txt" 2>&1
timeout -k 10 300 python -u tools/norm_bench.py > "$OUT/norm_bench.txt" 2>&1
timeout -k 10 600 python -u bench.py --train > "$OUT/bench_train.json.log" 2>&1
timeout -k 10 600 python -u bench.py --strong > "$OUT/bench_strong.json.log" 2>&1
bash tools/profile_bench.sh "$OUT/prof" r3
