#!/usr/bin/env python3
"""Per-launch HBM traffic from two rocprofv3 counter CSVs (FETCH_SIZE pass, WRITE_SIZE pass).

Usage: pmc_report.py fetch_counter_collection.csv write_counter_collection.csv
Calibrates both counters on the known-byte-count kernels of tools/pmc_traffic.py (k_fill_u32 writes
384 MiB; k_zkeys_decode reads and writes 384 MiB with 4 B/lane) as MI355X_MICROARCH.md asks, then
prints corrected bytes per launch for the frame kernels.
"""
import collections
import csv
import json
import os
import sys

CAL_BYTES = 96 * 1024 * 1024 * 4


def load(path, counter):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            per[r['Kernel_Name']].append(float(r['Counter_Value']))
    return per


def pick(per, key):
    for name, vals in per.items():
        if key + '(' in name or (key + '<') in name:
            return vals
    return []


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
cal_f = CAL_BYTES / (sum(pick(fetch, 'k_zkeys_decode')) / len(pick(fetch, 'k_zkeys_decode')))
# WRITE_SIZE: calibrated on k_zkeys_decode's 4-B/lane full-line stream (the pattern of the frame kernels'
# stores).  The constant fill k_fill_u32 reads ~3x fewer units for the same byte count on this stack
# (identical dwords seem to be merged before the fabric counter); it is reported, not used.
cal_w = CAL_BYTES / (sum(pick(write, 'k_zkeys_decode')) / len(pick(write, 'k_zkeys_decode')))
cal_w_fill = CAL_BYTES / (sum(pick(write, 'k_fill_u32')[-3:]) / 3)
out = {'bytes_per_FETCH_SIZE_unit': cal_f, 'bytes_per_WRITE_SIZE_unit': cal_w, 'bytes_per_WRITE_SIZE_unit_constant_fill': cal_w_fill,
       'kernels': {}}
for key in ('k_project', 'k_tiles', 'k_place', 'k_frame', 'k_frame_ahead', 'k_fill_holes', 'k_crop_resize_u8'):
    f, w = pick(fetch, key), pick(write, key)
    if f and w:
        out['kernels'][key] = {'launches': len(f), 'fetch_bytes': cal_f * sum(f) / len(f), 'write_bytes': cal_w * sum(w) / len(w)}
        out['kernels'][key]['hbm_bytes'] = out['kernels'][key]['fetch_bytes'] + out['kernels'][key]['write_bytes']
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
out['sources_sha16'] = bench.kernel_sources_stamp()        # what bench.py checks before it quotes these figures
print(json.dumps(out, indent=1))
