#!/usr/bin/env python3
"""Per-kernel medians and overlap figures of the frame loop from a rocprofv3 kernel trace CSV (dev aid):
how long each kernel runs, what fraction of the span some k_deliver (PCIe hand-off) is running, and per-stream idle gaps."""
import collections
import csv
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
names = ('k_project', 'k_tiles', 'k_frame', 'k_fill_holes', 'k_crop_resize_u8', 'k_deliver')
ev = []
for r in rows:
    for key in names:
        if r['Kernel_Name'].startswith(key + '(') or (' ' + key + '(') in r['Kernel_Name'] or ('::' + key + '(') in r['Kernel_Name']:
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), key, r.get('Queue_Id', '?')))
ev.sort()
# drop everything before the last big gap (set-up) -- keep the final burst of frames
per = collections.defaultdict(list)
for s, e, k, q in ev:
    per[k].append((e - s) / 1e3)
for k in names:
    if per[k]:
        v = per[k]
        print('%-18s n=%4d median %7.1f us  min %7.1f  max %7.1f' % (k, len(v), statistics.median(v), min(v), max(v)))
d = [(s, e) for s, e, k, q in ev if k == 'k_deliver']
if d:
    n = len(d)
    d = d[skip:]
    span = d[-1][1] - d[0][0]
    busy, cur_s, cur_e = 0, d[0][0], d[0][1]
    for s, e in d[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('deliveries: %d over %.1f us -> %.1f us/frame; some k_deliver running %.0f %% of that span' % (len(d), span / 1e3, span / 1e3 / len(d), 100.0 * busy / span))
