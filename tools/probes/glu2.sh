set -o pipefail
O=gpurun_out/glu2; mkdir -p $O
for rep in 1 2; do
timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "geglu" "ff1 plain" --tile 256,128 --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
