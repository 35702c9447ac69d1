#!/bin/bash
# HBM bytes of single GEMM shapes (tools/gemm_bench.py cases) by rocprofv3 --pmc: FETCH_SIZE and WRITE_SIZE in separate passes.
# Usage on the GPU box: bash tools/pmc_gemm.sh gpurun_out/<dir> "<tile or empty>" case-substring...
OUT=$1; TILE=$2; shift 2
mkdir -p "$OUT"; export TMPDIR=/tmp
[ -n "$TILE" ] && export LDM_GEMM_TILE=$TILE
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d "$OUT" -o "$c" --output-format csv -- python3 tools/gemm_bench.py --iters 3 "$@" > "$OUT/$c.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'PY'
import csv, sys, collections
d = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/{c}_counter_collection.csv")):
        if "conv_gemm" in r["Kernel_Name"]:
            per[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in per.items():
        kb = sum(v) / len(v)
        print(c, k, f"{len(v)} launches, {kb * 1024 * (2 if c == 'FETCH_SIZE' else 1) / 1e6:.1f} MB per launch" + (" (x2 gfx950 correction)" if c == "FETCH_SIZE" else ""))
PY
