#!/usr/bin/env python3
"""Mean counter value per launch per (kernel, grid size) from rocprofv3 --pmc counter_collection CSVs (dev aid); with the
dispatches' own timestamps: the mean duration under the counters and the clock GRBM_GUI_ACTIVE implies.
    python tools/pmc_by_grid.py a_counter_collection.csv [name filter ...]      (--json: one JSON object instead of text)"""
import collections
import csv
import json
import sys

args = [a for a in sys.argv[1:] if a != '--json']
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(args[0])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if len(args) > 1 and not any(f in name for f in args[1:]):
        continue
    key = (name, int(r.get('Grid_Size', 0) or 0), int(r.get('Workgroup_Size', 0) or 0))
    per[key][r['Counter_Name']].append(float(r['Counter_Value']))
    if r.get('Start_Timestamp') and r.get('End_Timestamp'):
        dur[key][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
out = {}
for key, c in sorted(per.items()):
    name, grid, wg = key
    row = {k: sum(v) / len(v) for k, v in sorted(c.items())}
    row['launches'] = max(len(v) for v in c.values())
    if dur[key]:
        row['us'] = sum(dur[key].values()) / len(dur[key])
        if 'GRBM_GUI_ACTIVE' in row:
            row['clock_GHz'] = row['GRBM_GUI_ACTIVE'] / row['us'] * 1e-3
    out['%s grid=%d wg=%d' % (name, grid, wg)] = row
if '--json' in sys.argv:
    print(json.dumps(out, indent=1))
else:
    for k, row in out.items():
        print(k, ' '.join('%s=%.4g' % kv for kv in row.items()))
