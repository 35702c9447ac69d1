set -o pipefail
B=latent-diffusion_amd/build/prev
for i in 1 2; do
for lib in default $B/$1; do
if [ $lib = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$lib; fi
timeout -k 10 200 python -u tools/attention_bench.py --backward --only "d40" 2>&1 | grep -v amdgpu.ids | cut -c1-140 || exit 1
done
done
