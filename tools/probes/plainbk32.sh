set -o pipefail
for i in 1 2; do for k in 0 320 1280; do
echo "MAXK=$k $(LDM_GEMM_PLAIN_BK32_MAXK=$k timeout -k 10 200 python -u tools/gemm_bench.py "L0 conv3x3 320->320" "L0 proj" "L0 qkv" "L0 ffout" 2>&1 | grep "proj\|qkv\|ffout" | awk '{printf "%s %s us | ", $2" "$3, $(NF-3)}')" || exit 1
done; done
