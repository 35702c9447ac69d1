#!/bin/bash
# The whole evidence pass of a finished build, one GPU call: tools/round_start.sh (tests, bench, A/B leg, side benches, rocprofv3
# trace + PMC passes -> bench_pmc_traffic.json), then the bench lines again WITH that traffic file in place (so the committed lines
# carry roofline.traffic of the build they ran on), the backward attention bench and the training-step kernel trace.
#   gpurun --timeout 1200 -- 'bash tools/final_evidence.sh gpurun_out/r3_final8'
set -o pipefail
OUT=${1:-gpurun_out/final}
bash tools/round_start.sh "$OUT" all || exit 1
cp "$OUT/prof/bench_pmc_traffic.json" bench_pmc_traffic.json || exit 1
R="$OUT/lines"; mkdir -p "$R"
timeout -k 10 300 python -u bench.py > "$R/bench.json.log" 2>&1 || { tail "$R/bench.json.log"; exit 1; }
LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline > "$R/bench_again.json.log" 2>&1 || exit 1
LDM_BENCH_AB=0 LDM_GEMM_PINGPONG=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline > "$R/bench_plain_ring.json.log" 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --train > "$R/bench_train.json.log" 2>&1 || exit 1
timeout -k 10 300 python -u bench.py --strong > "$R/bench_strong.json.log" 2>&1 || exit 1
timeout -k 10 200 python -u tools/attention_bench.py --backward > "$R/attention_backward_bench.txt" 2>&1 || exit 1
grep -h '"metric"' "$R"/*.log | cut -c1-180
export TMPDIR=/tmp
T="$OUT/train_trace"; mkdir -p "$T"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$T" -o t --output-format csv -- python3 "$PWD/bench.py" --train --steps 3 --warmup 2 --no-cpu-baseline --plain-only > "$T/log.txt" 2>&1 || { tail "$T/log.txt"; exit 1; }
python3 tools/trace_step_summary.py "$T" > "$T/summary.md" && rm -f "$T/t_kernel_trace.csv"
head -12 "$T/summary.md"
