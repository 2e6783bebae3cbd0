#!/usr/bin/env python3
"""The last video of a rocprofv3 --kernel-trace --memory-copy-trace run as a timeline: per stream (queue) the kernels of each
transfer group collapsed into one span, and the device-to-host copies (dev aid: where does a short video lose time?).
    python tools/handoff_timeline.py kernel_trace.csv memory_copy_trace.csv"""
import csv
import sys

k = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0], r.get('Queue_Id', '?'))
     for r in csv.DictReader(open(sys.argv[1]))]
c = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %.1f MB' % (r.get('Direction', ''), int(r.get('Size', 0) or 0) / 1e6), 'dma')
     for r in csv.DictReader(open(sys.argv[2]))] if len(sys.argv) > 2 else []
ev = sorted(k + c)
# the last burst: everything after the last gap longer than 5 ms
start = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 8):i]) > 5_000_000:
        start = i
ev = ev[start:]
t0 = ev[0][0]
for s, e, name, q in ev:
    if name.startswith('COPY') or name in ('k_turn',) or (e - s) > 20_000:
        print('%9.1f .. %9.1f us  (%7.1f)  q%-4s %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
print('span %.1f us' % ((max(e[1] for e in ev) - t0) / 1e3))
