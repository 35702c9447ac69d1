set -o pipefail
O=gpurun_out/stag; mkdir -p $O
for st in 0 10 20 40 0; do
  echo "== stagger $st" >> $O/out.txt
  LDM_GEMM_STAGGER=$st timeout -k 10 300 python -u tools/gemm_bench.py "geglu" "L0 qkv" "L0 proj" "L0 ffout" "L0 conv3x3 320" --iters 30 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
