set -o pipefail
mkdir -p gpurun_out/pp1
LDM_GEMM_PINGPONG=1 timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 120 --timeout-method thread -k "test_every_bf16_tile or test_split_k or layernorm_fold" > gpurun_out/pp1/tests.log 2>&1 || { tail -30 gpurun_out/pp1/tests.log; exit 1; }
tail -2 gpurun_out/pp1/tests.log
timeout -k 10 300 python -u tools/gemm_bench.py --pp --iters 30 > gpurun_out/pp1/gemm_pp.txt 2>&1 || { tail gpurun_out/pp1/gemm_pp.txt; exit 1; }
cat gpurun_out/pp1/gemm_pp.txt
LDM_GEMM_PINGPONG=0 timeout -k 10 200 python -u tools/gemm_stamps.py "L0 conv3x3 320" "L1 conv3x3" "L2 conv3x3" "L0 proj" "L0 qkv" > gpurun_out/pp1/stamps_pp0.txt 2>&1 || { tail gpurun_out/pp1/stamps_pp0.txt; exit 1; }
LDM_GEMM_PINGPONG=1 timeout -k 10 200 python -u tools/gemm_stamps.py "L0 conv3x3 320" "L1 conv3x3" "L2 conv3x3" "L0 proj" "L0 qkv" > gpurun_out/pp1/stamps_pp1.txt 2>&1 || { tail gpurun_out/pp1/stamps_pp1.txt; exit 1; }
cat gpurun_out/pp1/stamps_pp0.txt gpurun_out/pp1/stamps_pp1.txt
