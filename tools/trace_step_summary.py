"""Per-kernel table of the LAST steady-state step of a `rocprofv3 --kernel-trace` run: the launches between the last two
occurrences of a marker kernel (default: opt_adamw_kernel, one per training step; ldm::conv_in for the inference bench).
Usage: python tools/trace_step_summary.py <trace dir> [marker substring]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?(?:ldm::|at::native::)?([A-Za-z0-9_:]+(?:<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:84]


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "opt_adamw_kernel"
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    seg = rows[idx[-2] + 1:idx[-1] + 1]
    wall = int(seg[-1]["End_Timestamp"]) - int(rows[idx[-2]]["End_Timestamp"])
    per = defaultdict(lambda: [0, 0])
    for r in seg:
        e = per[short(r["Kernel_Name"])]
        e[0] += 1
        e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy = sum(v[1] for v in per.values())
    print(f"Last steady-state step: {len(seg)} launches, {wall / 1e6:.2f} ms wall (end of marker to end of marker), "
          f"{busy / 1e6:.2f} ms inside kernels.\n")
    print("| ms / step | calls | avg us | share | kernel |")
    print("|---:|---:|---:|---:|---|")
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"| {t / 1e6:.3f} | {n} | {t / n / 1e3:.1f} | {100.0 * t / busy:.2f}% | `{k}` |")


if __name__ == "__main__":
    main()
