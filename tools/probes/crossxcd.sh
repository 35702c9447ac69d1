set -o pipefail
B=latent-diffusion_amd/build/prev
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py tests/test_unet_gpu.py -q -x --timeout 300 --timeout-method thread -k "attention or transformer or unet" 2>&1 | tail -1 || exit 1
for i in 1 2; do for v in prev default; do
if [ $v = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$B/libldm_amd_$v.so; fi
echo "== $v"; timeout -k 10 200 python -u tools/attention_bench.py 2>&1 | grep "cross\|L3" | cut -c1-70 || exit 1
timeout -k 10 200 python -u tools/attention_bench.py --backward 2>&1 | grep "cross\|L3" | cut -c1-70 || exit 1
done; done
