#!/usr/bin/env python3
"""Kernel timeline of the LAST pass in a rocprofv3 --kernel-trace csv: name, start and end (ms from the pass's first kernel).
usage: python tools/pass_timeline.py <kernel_trace.csv> [kernels per pass to show=12]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_init starts the last pass
starts = [i for i, r in enumerate(rows) if "k_init" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + (int(sys.argv[2]) if len(sys.argv) > 2 else 12)]:
    name = r["Kernel_Name"].split("(")[0][-60:]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} -> {(int(r['End_Timestamp']) - t0) / 1e6:8.3f} ms  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
