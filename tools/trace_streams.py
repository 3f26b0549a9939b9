#!/usr/bin/env python3
"""Per-queue timeline of steady-state frames of a sequence (rocprofv3 --kernel-trace output directory): for the frames between the last
few seed_fill kernels, every kernel with its queue, start offset, duration and the gap to the previous kernel OF ITS QUEUE; then, per
queue, busy time / gaps per frame.   python tools/trace_streams.py <dir> [frames=3] [skip=0: frames to leave out at the end of the run,
e.g. bench.py's instrumented passes]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'seed_fill' in r['Kernel_Name']]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
a, b = marks[-nfr - 2 - skip], marks[-2 - skip]
t0 = int(rows[a]['Start_Timestamp'])
last_end = {}
for r in rows[:a]:
    last_end[r['Queue_Id']] = int(r['End_Timestamp'])
busy, gaps = {}, {}
for r in rows[a:b]:
    q = r['Queue_Id']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = s - last_end.get(q, s)
    busy[q] = busy.get(q, 0) + e - s
    gaps.setdefault(q, []).append(g)
    print("%9.2f  q%-3s gap %7.2f dur %7.2f  %s" % ((s - t0) / 1e3, q, g / 1e3, (e - s) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::', '')[:50]))
    last_end[q] = e
span = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
print("frames %d, span %.1f us = %.1f us per frame" % (nfr, span, span / nfr))
for q in busy:
    small = [g for g in gaps[q] if g < 20000]
    print("queue %s: busy %.1f us per frame, %d kernels per frame, gaps < 20 us: sum %.1f us per frame (median %.2f)" %
          (q, busy[q] / 1e3 / nfr, len(gaps[q]) / nfr, sum(small) / 1e3 / nfr, sorted(small)[len(small) // 2] / 1e3 if small else 0))
