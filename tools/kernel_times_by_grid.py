#!/usr/bin/env python3
"""Mean / median duration per (kernel, grid) from a rocprofv3 kernel trace CSV: tells launches of the same kernel with
different numbers of frames apart (dev aid).   python tools/kernel_times_by_grid.py out/t_kernel_trace.csv [name filter ...]"""
import collections
import csv
import statistics
import sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if len(sys.argv) > 2 and not any(f in name for f in sys.argv[2:]):
        continue
    wg = int(r.get('Workgroup_Size_X', 0) or 0)
    gy = int(r.get('Grid_Size_Y', 1) or 1)
    gx = int(r.get('Grid_Size_X', 0) or 0)
    per[(name, gx // max(wg, 1), gy)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (name, gx, gy), v in sorted(per.items()):
    print('%-28s blocks %7d x %d  n=%4d  mean %7.2f  median %7.2f  min %7.2f us' % (name, gx, gy, len(v), statistics.mean(v), statistics.median(v), min(v)))
