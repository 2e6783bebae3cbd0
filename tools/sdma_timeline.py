#!/usr/bin/env python3
"""The device-side timeline of the LAST pass in a rocprofv3 --kernel-trace --memory-copy-trace run of tools/short_pass.py (dev aid):
every kernel and every copy with its start and end relative to the pass's first kernel, and how long the link sat idle.
    python tools/sdma_timeline.py t_kernel_trace.csv t_memory_copy_trace.csv"""
import csv
import sys

kernels = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0], r.get('Queue_Id', '?'))
           for r in csv.DictReader(open(sys.argv[1]))]
copies = []
for r in csv.DictReader(open(sys.argv[2])):
    copies.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', r.get('Name', '?')), int(r.get('Bytes', 0) or 0)))
kernels.sort()
copies.sort()
frames = [k for k in kernels if k[2].startswith('k_')]
# passes are separated by host-side pauses of >= 1 ms: split the frame kernels at gaps of > 0.5 ms
last_start = frames[-1][0]
for a, b in zip(reversed(frames[:-1]), reversed(frames[1:])):
    if b[0] - a[1] > 500000:
        last_start = b[0]
        break
    last_start = a[0]
t0 = last_start
ev = [(s, e, 'kernel ' + n + ' (queue ' + str(q) + ')') for s, e, n, q in kernels if s >= t0] + [(s, e, 'COPY %s %.1f MB' % (d, b / 1e6)) for s, e, d, b in copies if s >= t0 - 1000]
ev.sort()
busy, last_end, idle = 0, None, 0
for s, e, what in ev:
    print('%9.1f .. %9.1f us (%7.1f)  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, what))
    if what.startswith('COPY') and 'HOST' in what.upper():
        if last_end is not None and s > last_end:
            idle += s - last_end
        last_end = e if last_end is None else max(last_end, e)
        busy += e - s
cp = [(s, e) for s, e, w in ev if w.startswith('COPY')]
if cp:
    print('first copy starts at %.1f us, last ends at %.1f us; the link busy %.1f us, idle between copies %.1f us' % ((cp[0][0] - t0) / 1e3, (cp[-1][1] - t0) / 1e3, busy / 1e3, idle / 1e3))
