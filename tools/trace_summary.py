#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 kernel trace (csv): python3 tools/trace_summary.py <dir> [last_n_frames]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if "render_kernel" not in n and "levels_" not in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t0 = t0 or s
    short = "render<%s>" % n.split("render_kernel<")[1].split(">")[0] if "render_kernel" in n else n.split("::")[-1][:30]
    print("%-32s start %10.1f us  dur %8.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3))
