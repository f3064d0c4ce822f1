#!/usr/bin/env python3
"""one step of a rocprofv3 kernel trace as a timeline: python tools/trace_step.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_propagate' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0, i1 = idx[-k - 1], idx[-k]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print("%8.1f %8.1f %7.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Queue_Id'), r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]))
