#!/usr/bin/env python3
"""Timeline of the last N operations of a rocprofv3 --kernel-trace --memory-copy-trace run: kernels and copies in one list, by start time."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ops = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s' % r.get('Queue_Id', '?'), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', '%s %s bytes' % (r.get('Direction', '?'), r.get('Size', r.get('Bytes', '?')))))
ops.sort()
ops = ops[-n:]
t0 = ops[0][0]
for s, e, kind, name in ops:
    print(f"{(s - t0) / 1e3:9.2f} +{(e - s) / 1e3:8.2f}  {kind:6s} {name}")
