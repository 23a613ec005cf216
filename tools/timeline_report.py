#!/usr/bin/env python3
"""Print the launch timeline (start, duration, queue) of the last few bench steps from a rocprofv3
--kernel-trace csv: shows which matcher kernels co-run with which extraction kernels.
usage: timeline_report.py <dir with *_kernel_trace.csv> [n_last_kernels]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
queues = {}
for r in rows:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0][-40:]
    print("%9.1f %9.1f %8.1f us  q%d %s%s" % (s, e, e - s, q, "    " * q, name))
