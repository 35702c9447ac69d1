set -o pipefail
B=latent-diffusion_amd/build/prev
for i in 1 2; do for v in prev abl1 abl3 abl4; do
echo "$v $(LDM_AMD_LIB=$B/libldm_amd_$v.so timeout -k 10 200 python -u tools/attention_bench.py --only "L0 self" 2>&1 | grep "L0 self")" || exit 1
done; done
