set -o pipefail
O=gpurun_out/glu3; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -q -x --timeout 300 --timeout-method thread -k "glu or gated or transformer or headline or nonlinear" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "glu" --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
done
cat $O/out.txt
