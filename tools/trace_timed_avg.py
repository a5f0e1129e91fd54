#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --steps K`, the average duration of the LAST K
launches of the headline kernel -- the launches the bench timed (the ones before them are its
untimed priming and warm-up launches) -- next to the average over all launches.

    python tools/trace_timed_avg.py <kernel_trace.csv> <K> [kernel-name substring]
"""
import csv
import sys

path, k = sys.argv[1], int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "attn_fwd_"
rows = [r for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
last = dur[-k:]
print(f"kernel {name}: {len(dur)} launches, average {sum(dur) / len(dur):.1f} us; "
      f"the last {len(last)} (the timed ones): average {sum(last) / len(last):.1f} us, min {min(last):.1f}, max {max(last):.1f}")
