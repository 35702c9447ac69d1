set -o pipefail
O=gpurun_out/pp9; mkdir -p $O
timeout -k 10 300 python -u tools/gemm_bench.py "L0 conv3x3 320" "geglu" --tile 256,128 --iters 40 2>&1 | grep -v amdgpu >> $O/out.txt || { tail $O/out.txt; exit 1; }
cat $O/out.txt
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 300 --timeout-method thread -k "glu or tile" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python -u tools/gemm_bench.py "L0 proj" "L0 qkv" "L0 ffout" --sweep --iters 40 2>&1 | grep -v amdgpu > $O/sweep.txt || { tail $O/sweep.txt; exit 1; }
cat $O/sweep.txt
