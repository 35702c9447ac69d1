set -o pipefail
O=gpurun_out/attnbwd2; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -q -x --timeout 300 --timeout-method thread -k "attention" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
PREV=latent-diffusion_amd/build/prev/libldm_amd_prev.so
for i in 1 2; do
LDM_AMD_LIB=$PREV timeout -k 10 200 python -u tools/attention_bench.py --backward --only d40 > $O/bwd_prev$i.txt 2>&1 || { tail $O/bwd_prev$i.txt; exit 1; }
timeout -k 10 200 python -u tools/attention_bench.py --backward --only d40 > $O/bwd_new$i.txt 2>&1 || { tail $O/bwd_new$i.txt; exit 1; }
cat $O/bwd_prev$i.txt $O/bwd_new$i.txt | grep -v amdgpu.ids
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o new -- python3 $GRAFT_REPO_ROOT/tools/attention_bench.py --backward --only "L0 self" > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 || { tail $GRAFT_REPO_ROOT/$O/prof.log; exit 1; }
LDM_AMD_LIB=$GRAFT_REPO_ROOT/$PREV timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o prev -- python3 $GRAFT_REPO_ROOT/tools/attention_bench.py --backward --only "L0 self" > $GRAFT_REPO_ROOT/$O/prof_prev.log 2>&1 || { tail $GRAFT_REPO_ROOT/$O/prof_prev.log; exit 1; }
cd $GRAFT_REPO_ROOT && for f in $(find $O/prof -name "*kernel_stats.csv"); do echo $f; head -5 $f | cut -d, -f1-4 | cut -c1-120; done
