#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the memory-bound rows bench.py reports in config.hbm_kernels, per launch,
# beside their algorithmic bytes.  On the GPU box from the repo root: bash tools/hbm_rows_pmc.sh gpurun_out/hbm_rows
set -e -o pipefail
OUT=${1:-gpurun_out/hbm_rows}
ROOT=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 python3 "$ROOT/tools/hbm_rows.py" > "$OUT/rows.jsonl" 2> "$OUT/rows.err"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o fetch --output-format csv -- python3 "$ROOT/tools/hbm_rows.py" > "$OUT/fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o write --output-format csv -- python3 "$ROOT/tools/hbm_rows.py" > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/hbm_rows_pmc.md"
import csv, json, sys, collections
out = sys.argv[1]
def load(path, ctr):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path, newline="")):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}
fe = load(f"{out}/fetch_counter_collection.csv", "FETCH_SIZE")
wr = load(f"{out}/write_counter_collection.csv", "WRITE_SIZE")
rows = [json.loads(l) for l in open(f"{out}/rows.jsonl") if l.startswith("{")]
want = {"ddpm_step": rows[0], "ddpm_add_noise": rows[1], "gn_fused_kernel": rows[2], "layernorm_kernel": rows[3]}
print("| kernel (as launched by bench.py's hbm_kernels rows) | launches counted | FETCH_SIZE x 2 (gfx950) MB / launch | WRITE_SIZE MB / launch | "
      "algorithmic MB / launch | counted / algorithmic | us / launch (un-profiled) | algorithmic GB/s |")
print("|---|---|---|---|---|---|---|---|")
for key, row in want.items():
    ks = [k for k in fe if key in k]
    f = sum(fe[k][0] for k in ks) * 2 * 1024; nf = sum(fe[k][1] for k in ks)
    w = sum(wr[k][0] for k in wr if key in k) * 1024; nw = sum(wr[k][1] for k in wr if key in k)
    if not nf or not nw:
        continue
    alg = row["algorithmic_bytes"]
    print(f"| `{key}` {row['shape']} | {nf} | {f / nf / 1e6:.1f} | {w / nw / 1e6:.1f} | {alg / 1e6:.1f} | "
          f"{(f / nf + w / nw) / alg:.2f} | {row['us_per_launch']} | {row['GB_per_s']} |")
PY
