#!/usr/bin/env python3
"""Steady-state frames of a sequence from a rocprofv3 --kernel-trace --memory-copy-trace directory: every kernel AND every host-to-device
copy of the last few frames, in start order, with its queue (copies: the engine's agent pair), start offset, duration and the gap to the
previous item of its queue; then busy time per queue and frame.   python tools/trace_frames.py <dir> [frames=3] [skip=1]"""
import csv
import glob
import sys

d = sys.argv[1]
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
items = []
for r in csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])):
    items.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + r['Queue_Id'], r['Kernel_Name'].replace('(anonymous namespace)::', '')[:46]))
cf = glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True)
if cf:
    for r in csv.DictReader(open(cf[0])):
        if 'HOST_TO_DEVICE' in r.get('Direction', '') or 'HtoD' in r.get('Direction', '') or r.get('Direction', '') == 'MEMORY_COPY_HOST_TO_DEVICE':
            items.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy', 'H2D copy'))
items.sort()
marks = [i for i, it in enumerate(items) if 'seed_fill' in it[3]]
a, b = marks[-nfr - 1 - skip], marks[-1 - skip]
t0 = items[a][0]
last = {}
for it in items[:a]:
    last[it[2]] = it[1]
busy = {}
for s, e, q, name in items[a:b]:
    g = s - last.get(q, s)
    busy[q] = busy.get(q, 0) + e - s
    if e - s > 15000 or q == 'copy':
        print("%9.1f  %-5s gap %7.1f dur %7.1f  %s" % ((s - t0) / 1e3, q, g / 1e3, (e - s) / 1e3, name))
    last[q] = e
span = (items[b][0] - t0) / 1e3
print("frames %d, span %.1f us = %.1f us per frame" % (nfr, span, span / nfr))
for q in sorted(busy):
    print("  %-5s busy %.1f us per frame" % (q, busy[q] / 1e3 / nfr))
