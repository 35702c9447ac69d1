set -o pipefail
O=gpurun_out/pp8; mkdir -p $O
for rep in 1 2; do
  timeout -k 10 300 python -u tools/gemm_bench.py "geglu" --tile 256,128 --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for pp in 0 2 0 2; do
  LDM_BENCH_AB=0 LDM_GEMM_PINGPONG=$pp timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_pp$pp.log 2>&1 || { tail $O/bench_pp$pp.log; exit 1; }
  echo "pp=$pp $(grep -h '"metric"' $O/bench_pp$pp.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"])')"
done
