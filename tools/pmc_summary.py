#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir with *_counter_collection.csv> ..."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            dur = defaultdict(list)
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].split("::")[-1]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            print("==", f)
            for k in acc:
                n = len(dur[k])
                print(f"{k:28s} n={n:5d} dur_us={sum(dur[k]) / n / 1000:8.1f}  " +
                      "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(acc[k].items())))


if __name__ == "__main__":
    main()
