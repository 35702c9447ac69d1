#!/bin/bash
# PMC passes over the d = 40 self-attention launch (tools/attention_bench.py --only "L0 self"); run on the GPU box.
# Usage: bash tools/attn_pmc.sh gpurun_out/attn_pmc [--backward]      (--backward: the dQ and dK/dV kernels instead)
set -o pipefail
OUT=${1:-gpurun_out/attn_pmc}
MODE=${2:-}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT" -o "$name" --output-format csv -- \
    python3 "$ROOT/tools/attention_bench.py" --iters 2 --only "L0 self" $MODE > "$OUT/$name.log" 2>&1 || return 1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE || exit 1
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob, os
out = sys.argv[1]
for name in ("sq1", "sq2"):
    files = glob.glob(os.path.join(out, "**", f"{name}_counter_collection.csv"), recursive=True)
    if not files:
        print("no csv for", name); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        if "attention" not in k and "attn_bwd_d" not in k:
            continue
        k = k[:44]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    for k, c in per.items():
        print(k, "dispatches", len(n[k]))
        for cn, v in sorted(c.items()):
            print(f"   {cn:28s} {v / len(n[k]):16.0f} per dispatch")
PY
