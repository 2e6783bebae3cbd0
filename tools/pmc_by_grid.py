#!/usr/bin/env python3
"""Mean counter value per launch per (kernel, grid y) from rocprofv3 --pmc counter_collection CSVs (dev aid).
    python tools/pmc_by_grid.py a_counter_collection.csv [name filter ...]"""
import collections
import csv
import sys

per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if len(sys.argv) > 2 and not any(f in name for f in sys.argv[2:]):
        continue
    per[(name, int(r.get('Grid_Size_Y', 1) or 1))][r['Counter_Name']].append(float(r['Counter_Value']))
for (name, gy), c in sorted(per.items()):
    print('%s x%d' % (name, gy), ' '.join('%s=%.4g' % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
