#!/usr/bin/env python3
"""Median per-kernel durations of the frame loop from a rocprofv3 kernel trace CSV (dev aid).
Usage on the GPU box:
  cd /tmp && rocprofv3 --kernel-trace -d out -o t --output-format csv -- python $REPO/tools/frame_once.py
  python $REPO/tools/kernel_times.py out/t_kernel_trace.csv
"""
import collections
import csv
import statistics
import sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    per[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
total = 0.0
for key in ('k_project', 'k_tiles', 'k_frame', 'k_fill_holes', 'k_crop_resize_u8'):
    for name, v in per.items():
        if key + '(' in name:
            total += statistics.median(v)
            print('%-18s n=%3d median %6.1f us  min %6.1f  max %6.1f' % (key, len(v), statistics.median(v), min(v), max(v)))
print('sum of medians %.1f us' % total)
