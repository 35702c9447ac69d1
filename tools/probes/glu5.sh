set -o pipefail
O=gpurun_out/glu5; mkdir -p $O
for v in 0 3 0 3; do
  if [ $v != 0 ]; then export LDM_GEMM_GLU_BK32=$v; else unset LDM_GEMM_GLU_BK32; fi
  echo "== BK32 mode $v" >> $O/out.txt
  timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "L0 qkv" "L0 proj" "L0 geglu" "L0 ff1" --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
for v in 0 2 0 2; do
  if [ $v != 0 ]; then export LDM_GEMM_GLU_BK32=$v; else unset LDM_GEMM_GLU_BK32; fi
  LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$v.log 2>&1 || { tail $O/bench_$v.log; exit 1; }
  echo "mode=$v $(grep -h '"metric"' $O/bench_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
