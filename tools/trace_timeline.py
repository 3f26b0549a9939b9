#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run (csv): the last N kernels with start, duration and queue -- to see what overlaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print(f"{(s - t0) / 1e3:9.2f} +{(e - s) / 1e3:8.2f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
