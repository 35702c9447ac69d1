set -o pipefail
B=latent-diffusion_amd/build/prev
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread -k "layernorm or transformer or train" 2>&1 | tail -1 || exit 1
for lib in $B/libldm_amd_prev.so default $B/libldm_amd_prev.so default $B/libldm_amd_prev.so default; do
  if [ $lib = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$lib; fi
  echo "$lib $(timeout -k 10 400 python -u bench.py --train --no-cpu-baseline 2>&1 | grep '"metric"' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" || exit 1
done
