# timing-only ablations of the d = 40 dK/dV kernel (results are wrong by construction): rocprofv3 per-kernel averages
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/attnabl; mkdir -p $O
B=$GRAFT_REPO_ROOT/latent-diffusion_amd/build/prev
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-prev abl16 abl32}; do
LDM_AMD_LIB=$B/libldm_amd_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o $v --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attention_bench.py --backward --only "L0 self" --iters 4 > $O/$v.log 2>&1 || { tail $O/$v.log; exit 1; }
echo "$v $(grep -h "attn_bwd_dkv\|attn_bwd_dq" $O/${v}_kernel_stats.csv | awk -F, '{gsub(/"/,""); printf "%s calls %s avg_ns %s | ", substr($1,1,28), $2, $4}')"
done
