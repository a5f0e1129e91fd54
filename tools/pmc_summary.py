#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per dispatch for kernels whose name contains a substring.

    python tools/pmc_summary.py <dir with p*_counter_collection.csv> <kernel substring>
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, sub = sys.argv[1], sys.argv[2]
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
        per = defaultdict(float)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if sub not in row["Kernel_Name"]:
                    continue
                per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, name), v in per.items():
            acc[name].append(v)
    out = {}
    for name, vals in sorted(acc.items()):
        out[name] = sum(vals) / len(vals)
        print(f"{name:32s} {out[name]:18.1f}   (n={len(vals)})")
    return out


if __name__ == "__main__":
    main()
