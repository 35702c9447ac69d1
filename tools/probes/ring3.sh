set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -q -x --timeout 300 --timeout-method thread -k "gated or conv or glu or transformer or bf16" 2>&1 | tail -1 || exit 1
for i in 1 2 3; do for r in 0 1; do
echo "RING3=$r $(LDM_GEMM_128_RING3=$r LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c60-175)" || exit 1
done; done
