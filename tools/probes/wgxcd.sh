set -o pipefail
B=latent-diffusion_amd/build/prev
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread -k "wgrad or train or backward" 2>&1 | tail -1 || exit 1
for i in 1 2; do for v in prev default; do
if [ $v = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$B/libldm_amd_$v.so; fi
echo "== $v"; timeout -k 10 200 python -u tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-100 || exit 1
done; done
