#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-counter mean over the
dispatches of the frame kernel.   python tools/pmc_summary.py gpurun_out/pmc*/*/*_counter_collection.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            if "wofdm_frames_kernel" not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
