set -o pipefail
O=gpurun_out/pp11; mkdir -p $O
LDM_GEMM_PINGPONG=3 timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 120 --timeout-method thread -k "conv or gemm or tile or split or layernorm_fold" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python -u tools/gemm_bench.py "conv3x3" "L0 ffout" "L0 proj" "L0 qkv" --pp --iters 30 2>&1 | grep -v amdgpu > $O/out.txt || { tail $O/out.txt; exit 1; }
cat $O/out.txt
for pp in 2 3 2 3; do
  LDM_BENCH_AB=0 LDM_GEMM_PINGPONG=$pp timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_pp$pp.log 2>&1 || { tail $O/bench_pp$pp.log; exit 1; }
  echo "pp=$pp $(grep -h '"metric"' $O/bench_pp$pp.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["avg_launch_us"])')"
done
