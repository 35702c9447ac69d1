#!/bin/bash
# First GPU call of a round: the whole -m gpu suite, the headline bench (its line carries the LayerNorm-fold A/B and the
# HBM-kernel rows), a same-box A/B leg of the GEMM ring form (LDM_GEMM_PINGPONG=0: the plain ring), the side benches, then the rocprofv3 evidence
# (tools/profile_bench.sh: kernel trace + three PMC passes, bench_pmc_traffic.json).  Every GPU step has its own time limit;
# the script stops at the first step that ends in anything but success (a failed ASSERTION in the test suite -- pytest exit
# status 1 -- is reported and the run goes on; an abort, a fault or a time limit ends it).  On the GPU box, from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/round_start.sh gpurun_out/r3_start'
set -o pipefail
OUT=${1:-gpurun_out/round_start}
STAGE=${2:-all}     # "a": tests + headline bench + A/B legs; "b": side benches + rocprofv3; "all": both (about 20 minutes)
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {  # step <seconds> <logfile> <command...>
  local lim=$1 log=$2; shift 2
  timeout -k 10 "$lim" "$@" > "$log" 2>&1
  local rc=$?
  echo "[$(date +%H:%M:%S)] rc=$rc  $*  -> $log"
  return $rc
}
if [ "$STAGE" != b ]; then
step 900 "$OUT/gpu_tests.log" python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -x
rc=$?
tail -4 "$OUT/gpu_tests.log"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "test run ended with status $rc: stopping"; exit $rc; fi
step 300 "$OUT/bench.json.log" python -u bench.py || { tail -20 "$OUT/bench.json.log"; exit 1; }
grep -h '"metric"' "$OUT/bench.json.log" | cut -c1-700
LDM_BENCH_AB=0 LDM_GEMM_PINGPONG=0 step 200 "$OUT/bench_plain_ring.json.log" python -u bench.py --no-cpu-baseline || exit 1
LDM_BENCH_AB=0 step 200 "$OUT/bench_again.json.log" python -u bench.py --no-cpu-baseline || exit 1
for f in bench_plain_ring bench_again; do echo "$f: $(grep -h '"metric"' "$OUT/$f.json.log" | cut -c1-260)"; done
fi
if [ "$STAGE" = a ]; then exit 0; fi
step 300 "$OUT/attention_bench.txt" python -u tools/attention_bench.py || exit 1
step 300 "$OUT/norm_bench.txt" python -u tools/norm_bench.py || exit 1
step 600 "$OUT/bench_train.json.log" python -u bench.py --train || exit 1
step 600 "$OUT/bench_strong.json.log" python -u bench.py --strong || exit 1
bash tools/profile_bench.sh "$OUT/prof" r3
