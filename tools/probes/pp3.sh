set -o pipefail
O=gpurun_out/pp3; mkdir -p $O
LDM_GEMM_PINGPONG=2 timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 120 --timeout-method thread -k "conv or gemm or tile or split or layernorm_fold" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for pp in 0 1 2; do
  LDM_GEMM_PINGPONG=$pp timeout -k 10 300 python -u tools/gemm_bench.py --iters 30 > $O/gemm_pp$pp.txt 2>&1 || { tail $O/gemm_pp$pp.txt; exit 1; }
done
paste -d'|' <(cut -c1-60,70-100 $O/gemm_pp0.txt) <(cut -c70-100 $O/gemm_pp1.txt) <(cut -c70-100 $O/gemm_pp2.txt)
for pp in 0 2; do
  echo "== stamps PP=$pp" >> $O/stamps.txt
  LDM_GEMM_PINGPONG=$pp timeout -k 10 200 python -u tools/gemm_stamps.py "L0 conv3x3 320" "L1 conv3x3" "L2 conv3x3" "L0 proj" "L0 ffout" 2>&1 | grep -v amdgpu >> $O/stamps.txt || exit 1
done
cat $O/stamps.txt
