set -o pipefail
B=latent-diffusion_amd/build/prev
for i in 1 2 3; do for v in prev gnnt; do
echo "$v $(LDM_AMD_LIB=$B/libldm_amd_$v.so LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c60-175)" || exit 1
done; done
LDM_AMD_LIB=$B/libldm_amd_prev.so timeout -k 10 100 python -u tools/norm_bench.py 2>&1 | grep -v amdgpu | head -12
LDM_AMD_LIB=$B/libldm_amd_gnnt.so timeout -k 10 100 python -u tools/norm_bench.py 2>&1 | grep -v amdgpu | head -12
