"""Per-kernel summary of the three rocprofv3 --pmc passes over bench.py (SQ/GRBM counters, FETCH_SIZE, WRITE_SIZE).

Usage: python tools/pmc_summary.py <dir with sq_/fetch_/write_counter_collection.csv> [forwards]
Conventions (MI355X_MICROARCH.md):
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs -> busy cycles of the dispatch = GUI/8;
  * MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GUI/8 * 256 CUs * 4 SIMDs);
  * FETCH_SIZE is reported in KiB and counts 128-B requests as 64 B on gfx950 -> doubled; WRITE_SIZE (KiB) is exact for
    16-byte-per-lane stores; Infinity-Cache hits are included in both, so "HBM bytes" is an upper bound of DRAM traffic;
  * durations come from the dispatch timestamps of the same (serialised) profiled run: a little slower than un-profiled.
"""
import csv
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?(?:ldm::)?([A-Za-z0-9_:]+(?:<[^(]*>)?)", name)
    s = m.group(1) if m else name
    s = re.sub(r"unsigned short", "bf16", s)
    return s[:60]


def full_name(k):
    """bench.py's spelling of a GEMM kernel: conv_gemm_bf16_kernel<BM,BN,BK> (the trace has the wave layout and stage count too)."""
    m = re.match(r"(conv_gemm_(?:bf16|f32)_kernel)<(\d+), ?(\d+), ?(\d+)", k)
    return f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)}>" if m else k


def load(path):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    calls = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = r["Dispatch_Id"]
            calls[k].add(d)
            if d not in seen:
                seen.add(d)
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return per, {k: len(v) for k, v in calls.items()}, dur


def main():
    d = sys.argv[1]
    fwd = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 3
    json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None      # bench_pmc_traffic.json
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "fwd_b32_l64"
    profile = sys.argv[sys.argv.index("--profile") + 1] if "--profile" in sys.argv else d
    sq, calls, dur = load(os.path.join(d, "sq_counter_collection.csv"))
    fe, _, fdur = load(os.path.join(d, "fetch_counter_collection.csv"))
    wr, _, wdur = load(os.path.join(d, "write_counter_collection.csv"))
    rows = []
    for k in sq:
        if not (k.startswith("conv_") or k.startswith("attention") or k.startswith("gn_") or k.startswith("layernorm")
                or k.startswith("ddpm") or k.startswith("splitk")):
            continue
        c = sq[k]
        gui = c["GRBM_GUI_ACTIVE"] / 8.0
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4) if gui else 0.0
        rd = 2.0 * fe[k]["FETCH_SIZE"] * 1024
        wb = wr[k]["WRITE_SIZE"] * 1024
        t = dur[k]
        rows.append((t, k, calls[k], util, c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                     c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1),
                     gui / 2.4e9 * 0 + (gui / t / 1e9 if t else 0), rd, wb, (rd + wb) / ((fdur[k] + wdur[k]) / 2) / 1e9 if fdur[k] else 0))
    rows.sort(reverse=True)
    print(f"| kernel | launches/fwd | ms/fwd (profiled) | MFMA busy | wave-parked | issue-stall | issuing | clock GHz | "
          f"read MB/launch | write MB/launch | (rd+wr) GB/s |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for t, k, n, util, wany, winst, act, clk, rd, wb, bw in rows:
        print(f"| `{k}` | {n / fwd:.0f} | {t / fwd * 1e3:.3f} | {100 * util:.1f}% | {100 * wany:.0f}% | {100 * winst:.0f}% | "
              f"{100 * act:.0f}% | {clk:.2f} | {rd / n / 1e6:.1f} | {wb / n / 1e6:.1f} | {bw:.0f} |")
    if json_out:
        # what bench.py's roofline.traffic reads: HBM(+Infinity Cache) bytes per launch per kernel, valid for the kernel sources
        # that were profiled (hash of latent-diffusion_amd/csrc, as bench.kernel_source_hash computes it)
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import kernel_source_hash
        # instantiations of one tile (the ring forms of the 256-row GEMM tiles differ in their last template argument) are
        # one kernel to bench.py, which names a GEMM by its tile: bytes and launches are summed over them
        agg = defaultdict(lambda: [0.0, 0])
        for t, k, n, util, wany, winst, act, clk, rd, wb, bw in rows:
            agg[full_name(k)][0] += rd + wb
            agg[full_name(k)][1] += n
        rec = {"source_hash": kernel_source_hash(), "profile": profile,
               "workloads": {workload: {k: {"hbm_bytes_per_launch": b / n, "launches_per_forward": n / fwd}
                                        for k, (b, n) in agg.items() if n}}}
        with open(json_out, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
