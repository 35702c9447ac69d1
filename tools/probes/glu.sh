set -o pipefail
O=gpurun_out/glu; mkdir -p $O
timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "glu" --iters 40 2>&1 | grep -v amdgpu > $O/out.txt || { tail $O/out.txt; exit 1; }
timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "glu" --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
cat $O/out.txt
