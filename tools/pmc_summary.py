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
    fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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


if __name__ == "__main__":
    main()
