set -e -o pipefail
O=gpurun_out/r3_train_trace2; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $PWD/bench.py --train --steps 3 --warmup 2 --no-cpu-baseline --plain-only > $O/log.txt 2>&1
python3 tools/trace_step_summary.py $O > $O/summary.md
rm -f $O/t_kernel_trace.csv
head -60 $O/summary.md
