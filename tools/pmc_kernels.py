#!/usr/bin/env python3
"""Mean counter value per launch for the frame kernels, from rocprofv3 --pmc counter_collection CSVs (dev aid).

    python tools/pmc_kernels.py a_counter_collection.csv [b_counter_collection.csv ...]
"""
import collections
import csv
import sys

KEYS = ('k_project', 'k_tiles', 'k_place', 'k_frame', 'k_frame_ahead', 'k_frame_group_ahead', 'k_fill_holes', 'k_crop_resize_u8', 'k_deliver')
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        for k in KEYS:
            if k + '(' in r['Kernel_Name']:
                per[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in KEYS:
    if k in per:
        print(k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(per[k].items())))
