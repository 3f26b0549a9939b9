#!/usr/bin/env python3
"""Per-kernel durations of a rocprofv3 --kernel-trace run (csv): calls, mean, min, max in us, for kernels whose name holds any of the given substrings."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
pats = sys.argv[2:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if not pats or any(p in n for p in pats):
        d[n[:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:70s} calls {len(v):6d} mean {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f} total {sum(v)/1e3:8.2f} ms")
