set -o pipefail
LDM_GEMM_TILE=256,160 timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 300 --timeout-method thread -k "gated or glu" 2>&1 | tail -1 || exit 1
for i in 1 2; do
timeout -k 10 200 python -u tools/gemm_bench.py --tile 256,160 "L0 conv3x3 320->320" "L0 geglu" "L1 geglu" "L2 geglu" 2>&1 | grep -v amdgpu.ids | grep geglu | cut -c1-110 || exit 1
done
