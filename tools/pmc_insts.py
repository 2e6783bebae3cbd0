#!/usr/bin/env python3
"""Per-launch wave-level instruction counts of the scatter kernels from rocprofv3 --pmc counter_collection CSVs of ONE-frame
launches (KBE_LANES=1 KBE_FILL_GROUP=1 tools/frame_once.py, once per route) -> JSON for bench.py's VALU-issue roofline.
    python tools/pmc_insts.py fused_counter_collection.csv bucket_counter_collection.csv > profiles/rNN_scatter_insts.json"""
import collections
import csv
import json
import os
import sys

KEYS = ('k_place', 'k_frame', 'k_frame_ahead', 'k_project', 'k_tiles')
per = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        if name in KEYS:
            per[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {'note': 'mean per launch (one frame per launch, 1024^2 bench workload); wave-level instructions', 'kernels': {}}
for k in KEYS:
    if k in per:
        c = per[k]
        mean = lambda n: sum(c[n]) / len(c[n]) if c[n] else None     # noqa: E731
        out['kernels'][k] = {'launches': len(c['SQ_INSTS_VALU']), 'valu': mean('SQ_INSTS_VALU'), 'salu': mean('SQ_INSTS_SALU'), 'lds': mean('SQ_INSTS_LDS'),
                             'waves': mean('SQ_WAVES')}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out['sources_sha16'] = bench.kernel_sources_stamp()        # what bench.py checks before it quotes these figures
print(json.dumps(out, indent=1))
