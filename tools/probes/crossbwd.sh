set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/crossbwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "L0 cross" "L1 cross" "L2 cross"; do
n=$(echo $c | tr -d ' ')
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o $n --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attention_bench.py --backward --only "$c" --iters 6 > $O/$n.log 2>&1 || { tail $O/$n.log; exit 1; }
echo "== $c"; python3 - $O/${n}_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_bwd' in r['Name'] or 'attention' in r['Name']:
        print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1000,1))
PY
done
