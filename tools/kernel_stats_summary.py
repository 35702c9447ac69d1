"""Per-kernel table (calls, ms per forward, average duration, share) + inter-kernel idle time from a
`rocprofv3 --kernel-trace --stats` run of bench.py.  Usage: python tools/kernel_stats_summary.py <dir> <forwards>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, fwd = sys.argv[1], int(sys.argv[2])
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path, newline="")))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = defaultdict(lambda: [0, 0])
    for r in rows:
        e = per[r["Kernel_Name"]]
        e[0] += 1
        e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # the last complete forward: between the last two conv_in launches
    idx = [i for i, r in enumerate(rows) if "ldm::conv_in" in r["Kernel_Name"]]
    seg = rows[idx[-2]:idx[-1]]
    wall = int(rows[idx[-1]]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    total = sum(v[1] for v in per.values())
    print(f"Kernel time per forward (all {fwd} forwards of the run, one-time weight packing / initialisation included): "
          f"{total / fwd / 1e6:.2f} ms.")
    print(f"Last steady-state step: {len(seg)} kernels, {wall / 1e6:.2f} ms from first launch to the next step's first launch, "
          f"{busy / 1e6:.2f} ms inside kernels, {(wall - busy) / 1e6:.2f} ms between them "
          f"({(wall - busy) / max(len(seg), 1) / 1e3:.1f} us per kernel boundary).\n")
    print("| kernel | calls | ms per forward | avg us | share |")
    print("|---|---|---|---|---|")
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"| `{k[:96]}` | {n} | {t / fwd / 1e6:.3f} | {t / n / 1e3:.1f} | {100.0 * t / total:.1f}% |")


if __name__ == "__main__":
    main()
