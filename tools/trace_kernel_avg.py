#!/usr/bin/env python3
"""Average duration (us) of the launches of kernels whose name contains a substring, over every
*kernel_trace.csv under a directory (the launches of rocprofv3 counter passes run slower than
unprofiled ones: use these durations only together with the counters of the same passes)."""
import csv
import glob
import os
import sys


def main():
    d, sub = sys.argv[1], sys.argv[2]
    durs = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if sub in row["Kernel_Name"]:
                    durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if durs:
        durs.sort()
        print(f"launches {len(durs)}  mean_us {sum(durs) / len(durs):.2f}  median_us {durs[len(durs) // 2]:.2f}  min_us {durs[0]:.2f}")
    else:
        print("no launches found")


if __name__ == "__main__":
    main()
