set -o pipefail
O=gpurun_out/attnbwd; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
PREV=latent-diffusion_amd/build/prev/libldm_amd_prev.so
LDM_AMD_LIB=$PREV timeout -k 10 200 python -u tools/attention_bench.py --backward > $O/bwd_prev.txt 2>&1 || { tail $O/bwd_prev.txt; exit 1; }
timeout -k 10 200 python -u tools/attention_bench.py --backward > $O/bwd_new.txt 2>&1 || { tail $O/bwd_new.txt; exit 1; }
cat $O/bwd_prev.txt $O/bwd_new.txt | grep -v amdgpu.ids
for lib in prev new prev new; do
  if [ $lib = prev ]; then export LDM_AMD_LIB=$PREV; else unset LDM_AMD_LIB; fi
  timeout -k 10 400 python -u bench.py --train --no-cpu-baseline > $O/train_$lib.log 2>&1 || { tail $O/train_$lib.log; exit 1; }
  echo "$lib $(grep -h '"metric"' $O/train_$lib.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
