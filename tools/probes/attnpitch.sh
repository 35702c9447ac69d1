set -o pipefail
O=gpurun_out/attnpitch; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -q -x --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
PREV=latent-diffusion_amd/build/prev/libldm_amd_prev.so
for i in 1 2; do
for mode in "" "--backward"; do
LDM_AMD_LIB=$PREV timeout -k 10 200 python -u tools/attention_bench.py $mode > $O/prev$i$mode.txt 2>&1 || { tail $O/prev$i$mode.txt; exit 1; }
timeout -k 10 200 python -u tools/attention_bench.py $mode > $O/new$i$mode.txt 2>&1 || { tail $O/new$i$mode.txt; exit 1; }
cat $O/prev$i$mode.txt $O/new$i$mode.txt | grep -v amdgpu.ids | cut -c1-64
done
done
