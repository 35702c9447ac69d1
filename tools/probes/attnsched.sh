set -o pipefail
O=gpurun_out/attnsched; mkdir -p $O
B=latent-diffusion_amd/build/prev
for i in 1 2; do
for lib in default $B/libldm_amd_sched.so; do
if [ $lib = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$lib; fi
timeout -k 10 200 python -u tools/attention_bench.py --backward --only "L0 self" 2>&1 | grep -v amdgpu.ids | cut -c1-140 || exit 1
done
done
for lib in $B/libldm_amd_fold9.so default $B/libldm_amd_fold9.so default; do
  if [ $lib = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$lib; fi
  timeout -k 10 400 python -u bench.py --train --no-cpu-baseline > $O/train.log 2>&1 || { tail $O/train.log; exit 1; }
  echo "$lib $(grep -h '"metric"' $O/train.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
