#!/usr/bin/env python3
"""configs[4]'s HBM traffic per FRAME and kernel as JSON, from the per-launch figures tools/gpu_profile.sh step 3c writes
(config4_traffic.txt: `name grid=... wg=256 FETCH_SIZE=... launches=... us=...`, FETCH_SIZE / WRITE_SIZE in the units
tools/pmc_report.py calibrated: its hbm_traffic.json, second argument).  bench.py reads the result for that workload's `roofline.traffic`.
    python tools/pmc_config4_report.py config4_traffic.txt hbm_traffic.json > hbm_traffic_config4.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

SIZE, POINTS = 2048, 4 * 2048 * 2048
TILES = (SIZE // 32) * (SIZE // 16)
cal = json.load(open(sys.argv[2]))
unit = {'FETCH_SIZE': cal['bytes_per_FETCH_SIZE_unit'], 'WRITE_SIZE': cal['bytes_per_WRITE_SIZE_unit']}
# what bench.py calls the launches of a frame
names = {'k_frame_group_ahead_dense': 'k_frame_ahead', 'k_frame_group_ahead': 'k_frame_ahead', 'k_frame_group': 'k_frame', 'k_place': 'k_place',
         'k_project_group': 'k_project', 'k_tiles_group': 'k_tiles'}
out = {}
for line in open(sys.argv[1]):
    m = re.match(r'(\w+) grid=(\d+) wg=(\d+) (FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+) launches=(\d+) us=([0-9.]+)', line.strip())
    if not m or m.group(1) not in names:
        continue
    kernel, grid, wg, counter, value, launches, us = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), float(m.group(5)), int(m.group(6)), float(m.group(7))
    per_frame_threads = POINTS if kernel == 'k_place' or kernel == 'k_project_group' else TILES * wg
    frames = max(1, round(grid / per_frame_threads))
    e = out.setdefault(names[kernel], {'kernel': kernel, 'frames_per_launch': frames, 'launches': launches, 'fetch_bytes': 0.0, 'write_bytes': 0.0, 'us_per_frame_under_the_counters': round(us / frames, 1)})
    e['fetch_bytes' if counter == 'FETCH_SIZE' else 'write_bytes'] = value * unit[counter] / frames
for e in out.values():
    e['hbm_bytes'] = e['fetch_bytes'] + e['write_bytes']
    e['bytes_per_point'] = round(e['hbm_bytes'] / POINTS, 1)
print(json.dumps({'workload': {'size': SIZE, 'upsample': 2, 'points': POINTS}, 'note': 'HBM bytes per FRAME (a launch holds frames_per_launch of them); '
                  'units calibrated by tools/pmc_report.py', 'bytes_per_FETCH_SIZE_unit': unit['FETCH_SIZE'], 'bytes_per_WRITE_SIZE_unit': unit['WRITE_SIZE'],
                  'kernels': out, 'sources_sha16': bench.kernel_sources_stamp()}, indent=1))
