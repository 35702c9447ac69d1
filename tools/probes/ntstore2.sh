set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 300 --timeout-method thread -k "conv or gated or nontemporal or layernorm_folded" 2>&1 | tail -1 || exit 1
for i in 1 2 3; do for mb in 0 64 100 128 200; do
echo "LDM_GEMM_NT_MB=$mb $(LDM_GEMM_NT_MB=$mb LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c60-175)" || exit 1
done; done
