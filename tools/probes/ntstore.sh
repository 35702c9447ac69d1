set -o pipefail
B=latent-diffusion_amd/build/prev
for i in 1 2; do for v in prev nt; do
export LDM_AMD_LIB=$B/libldm_amd_$v.so
echo "== $v"; timeout -k 10 200 python -u tools/gemm_bench.py "L0 conv3x3 320->320" "L0 proj" "L0 qkv" "L0 ffout" "L2 conv3x3" 2>&1 | grep -v amdgpu.ids | cut -c1-110 || exit 1
LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c60-175 || exit 1
done; done
