set -o pipefail
O=gpurun_out/pp5; mkdir -p $O
for mode in "" "--ones" "--gap"; do
  echo "== mode '$mode'" >> $O/out.txt
  timeout -k 10 300 python -u tools/gemm_bench.py "conv3x3" "L0 ffout" "L0 proj" --pp --iters 30 $mode 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
