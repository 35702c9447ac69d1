set -o pipefail
O=gpurun_out/pp7; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for pp in 0 2; do
  echo "== pp $pp" >> $O/out.txt
  LDM_GEMM_PINGPONG=$pp timeout -k 10 300 python -u tools/gemm_bench.py "geglu" "ff1 plain" --tile 256,128 --iters 30 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
for v in 0 1 0 1; do
  LDM_TRAIN_GLU_FUSED=$v timeout -k 10 400 python -u bench.py --train --no-cpu-baseline > $O/train_$v.log 2>&1 || { tail $O/train_$v.log; exit 1; }
  echo "glu_fused=$v $(grep -h '"metric"' $O/train_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
