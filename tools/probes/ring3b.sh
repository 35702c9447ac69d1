set -o pipefail
for i in 1 2; do for k in 320 640 1280; do
echo "== MAXK=$k"; LDM_GEMM_GLU_BK32_MAXK=$k timeout -k 10 200 python -u tools/gemm_bench.py "L0 conv3x3 320->320" "L1 geglu" "L2 geglu" 2>&1 | grep -v amdgpu.ids | grep geglu | cut -c1-110 || exit 1
done; done
