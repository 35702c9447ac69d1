set -o pipefail
B=latent-diffusion_amd/build/prev
timeout -k 10 400 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 300 --timeout-method thread -k "conv or ring or split or gemm or layernorm_folded or linear" 2>&1 | tail -1 || exit 1
for i in 1 2; do for v in prev default; do
if [ $v = default ]; then unset LDM_AMD_LIB; else export LDM_AMD_LIB=$B/libldm_amd_$v.so; fi
echo "== $v $(timeout -k 10 200 python -u tools/gemm_bench.py "L0 conv3x3 320->320" "L0 proj" "L0 qkv" "L0 ffout" "L2 conv3x3" 2>&1 | grep "proj\|qkv\|ffout\|conv3x3" | awk '{printf "%s %s | ", $2" "$3, $(NF-3)}')" || exit 1
echo "   $(LDM_BENCH_AB=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c60-175)" || exit 1
done; done
