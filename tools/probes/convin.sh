set -o pipefail
B=$GRAFT_REPO_ROOT/latent-diffusion_amd/build/prev
O=$GRAFT_REPO_ROOT/gpurun_out/convin; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 1 2 3 4; do
LDM_AMD_LIB=$B/libldm_amd_cin$m.so LDM_BENCH_AB=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o m$m --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/m$m.log 2>&1 || { tail $O/m$m.log; exit 1; }
echo "minb=$m $(grep -h "conv_in4\|conv_out_lanes4" $O/m${m}_kernel_stats.csv | awk -F, '{gsub(/"/,""); printf "%s avg_ns %s | ", substr($1,11,22), $4}')"
done
