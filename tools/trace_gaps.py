#!/usr/bin/env python3
"""Launch-gap analysis of a rocprofv3 --kernel-trace CSV: how much of the wall clock between the first and the last
kernel is spent INSIDE kernels and how much between them (dispatch gaps), per kernel name.
    python tools/trace_gaps.py <dir or *_kernel_trace.csv> [skip_first_n_kernels]
"""
import collections
import csv
import glob
import os
import sys


def find(path):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not c:
        raise SystemExit(f"no *kernel_trace.csv under {path}")
    return c[0]


def main():
    rows = list(csv.DictReader(open(find(sys.argv[1]))))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows))[skip:]
    busy = collections.defaultdict(lambda: [0, 0.0, 0.0])     # name -> [n, kernel us, gap-before us]
    tot_k = tot_g = 0.0
    for i, (s, e, n) in enumerate(ev):
        d = (e - s) / 1e3
        g = max(0.0, (s - ev[i - 1][1]) / 1e3) if i else 0.0
        if g > 200.0:            # host-side pause (between timing loops), not a dispatch gap
            g = 0.0
        b = busy[n]
        b[0] += 1; b[1] += d; b[2] += g
        tot_k += d; tot_g += g
    span = (ev[-1][1] - ev[0][0]) / 1e3
    print(f"{len(ev)} kernels, span {span/1e3:.2f} ms, in kernels {tot_k/1e3:.2f} ms, in gaps {tot_g/1e3:.2f} ms "
          f"(avg gap {tot_g/len(ev):.2f} us)")
    print(f"{'kernel':64s} {'n':>6s} {'avg us':>8s} {'gap us':>7s} {'kern ms':>8s} {'gap ms':>7s}")
    for n, (c, k, g) in sorted(busy.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:40]:
        print(f"{n[:64]:64s} {c:6d} {k/c:8.2f} {g/c:7.2f} {k/1e3:8.2f} {g/1e3:7.2f}")


if __name__ == "__main__":
    main()
