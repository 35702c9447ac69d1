set -o pipefail
O=gpurun_out/trainab; mkdir -p $O
B=latent-diffusion_amd/build/prev
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 200 python -u tools/attention_bench.py --backward 2>&1 | grep -v amdgpu.ids | cut -c1-140 || exit 1
for lib in $B/${1:-libldm_amd_prev.so} default $B/${1:-libldm_amd_prev.so} default; do
  if [ $lib = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$lib; fi
  timeout -k 10 400 python -u bench.py --train --no-cpu-baseline > $O/train.log 2>&1 || { tail $O/train.log; exit 1; }
  echo "$lib $(grep -h '"metric"' $O/train.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
