# timing-only ablations of the 64x64 level's short-K layers on the stamps build: 16 = no output store, 1 = no LDS-DMA in the ring loop
set -o pipefail
for i in 1 2; do for a in 0 16 1 17; do
echo "ABLATE=$a $(LDM_AMD_LIB=latent-diffusion_amd/lib/libldm_amd_stamps.so LDM_GEMM_ABLATE=$a timeout -k 10 200 python -u tools/gemm_bench.py "L0 conv3x3 320->320" "L0 proj" "L0 qkv" "L0 ffout" 2>&1 | grep "proj\|qkv\|ffout" | awk '{printf "%s %s us | ", $2" "$3, $(NF-3)}')" || exit 1
done; done
