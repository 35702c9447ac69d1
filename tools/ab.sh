#!/bin/bash
# Same-box A/B of two builds of the library: bench.py (headline workload, with its LayerNorm-fold A/B legs) and the norm bench,
# new / previous / new.  The previous build is latent-diffusion_amd/build/prev/libldm_amd_prev.so (git-ignored; copy it there
# before changing a kernel).  Usage on the GPU box: bash tools/ab.sh gpurun_out/<dir> [extra bench args]
set -o pipefail
OUT=${1:-gpurun_out/ab}; shift
PREV=$PWD/latent-diffusion_amd/build/prev/libldm_amd_prev.so
mkdir -p "$OUT"
run() { # run <tag> <lib or ""> cmd...
  local tag=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then LDM_AMD_LIB=$lib timeout -k 10 300 "$@" > "$OUT/$tag.log" 2>&1; else timeout -k 10 300 "$@" > "$OUT/$tag.log" 2>&1; fi
  local rc=$?
  echo "rc=$rc $tag"; return $rc
}
summ() { python3 - "$1" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        d = json.loads(line); c = d["config"]; ab = c.get("layernorm_fold_ab", {})
        o = ab.get("other", {})
        print(f'   {d["ms_per_step"]} ms (fold {c.get("layernorm_fold_active")}); other leg: {o.get("ms_per_step")} ms (fold {o.get("fold")})')
        print("   this:", {k: (v["avg_us"], v["launches"]) for k, v in d["roofline"]["all_gemm_kernels"].items()})
        if o.get("gemm_kernels"): print("   other:", {k: (v["avg_us"], v["launches"]) for k, v in o["gemm_kernels"].items()})
PY
}
run new1 "" python -u bench.py --no-cpu-baseline "$@" || exit 1; summ "$OUT/new1.log"
run prev "$PREV" python -u bench.py --no-cpu-baseline "$@" || exit 1; summ "$OUT/prev.log"
run new2 "" python -u bench.py --no-cpu-baseline "$@" || exit 1; summ "$OUT/new2.log"
run norm_new "" python -u tools/norm_bench.py || exit 1
run norm_prev "$PREV" python -u tools/norm_bench.py || exit 1
paste -d'|' <(grep -h "groupnorm\|layernorm" "$OUT/norm_new.log" | cut -c1-75) <(grep -h "groupnorm\|layernorm" "$OUT/norm_prev.log" | cut -c38-75)
