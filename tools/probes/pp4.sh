set -o pipefail
O=gpurun_out/pp4; mkdir -p $O
for pp in 0 2 1 0 2; do
  LDM_BENCH_AB=0 LDM_GEMM_PINGPONG=$pp timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_pp$pp.log 2>&1 || { tail $O/bench_pp$pp.log; exit 1; }
  echo "pp=$pp $(grep -h '"metric"' $O/bench_pp$pp.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"])')"
done
