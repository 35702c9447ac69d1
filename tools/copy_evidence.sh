#!/bin/bash
# After tools/final_evidence.sh: copy what the judge reads from gpurun_out/<dir> into profiles/ and the repo root.
#   bash tools/copy_evidence.sh gpurun_out/r3_final8 r3
S=${1:?evidence dir}; TAG=${2:-r3}
cp "$S/prof/bench_pmc_traffic.json" bench_pmc_traffic.json
for f in bench bench_again bench_plain_ring bench_train bench_strong; do grep -v amdgpu.ids "$S/lines/$f.json.log" > "profiles/${TAG}_$f.json.log"; done
cp "$S/prof/${TAG}_bench_summary.md" "$S/prof/${TAG}_pmc_per_kernel.md" profiles/
cp "$S"/prof/trace/*kernel_stats.csv "profiles/${TAG}_bench_kernel_stats.csv"
grep -v amdgpu.ids "$S/attention_bench.txt" > "profiles/${TAG}_attention_bench.txt"
grep -v amdgpu.ids "$S/lines/attention_backward_bench.txt" > "profiles/${TAG}_attention_backward_bench.txt"
grep -v amdgpu.ids "$S/norm_bench.txt" > "profiles/${TAG}_norm_bench.txt"
tail -3 "$S/gpu_tests.log" > "profiles/${TAG}_gpu_tests_tail.txt"
cp "$S/train_trace/summary.md" "profiles/${TAG}_train_trace_summary_final.md"
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
print("traffic file hash", json.load(open("bench_pmc_traffic.json")).get("source_hash"), "| csrc hash", bench.kernel_source_hash())
PY
