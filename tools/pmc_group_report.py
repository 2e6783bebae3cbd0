#!/usr/bin/env python3
"""The scatter's GROUP launches under the counters: three rocprofv3 --pmc passes over tools/ahead_time.py (instruction counts,
FETCH_SIZE, WRITE_SIZE; each reduced by tools/pmc_by_grid.py --json) -> per-launch means of k_frame_group_ahead by frames per
launch, merged into the profile's hbm_traffic.json and scatter_insts.json as `by_frames_per_launch` (bench.py prices the launch
the timed region uses with these: the one-frame launch's figures miss what the frames of a group share).
    python tools/pmc_group_report.py insts.json fetch.json write.json hbm_traffic.json scatter_insts.json [tiles]"""
import json
import re
import sys

insts, fetch, write = (json.load(open(p)) for p in sys.argv[1:4])
traffic_path, insts_path = sys.argv[4], sys.argv[5]
tiles = int(sys.argv[6]) if len(sys.argv) > 6 else 2048         # 1024^2 in 32 x 16 tiles
traffic, counts = json.load(open(traffic_path)), json.load(open(insts_path))
cal_f, cal_w = traffic['bytes_per_FETCH_SIZE_unit'], traffic['bytes_per_WRITE_SIZE_unit']


def by_frames(table, kernel):
    out = {}
    for key, row in table.items():
        m = re.match(r'(\S+) grid=(\d+) wg=(\d+)', key)
        if m and m.group(1) == kernel and int(m.group(2)) % (tiles * int(m.group(3))) == 0:
            out[int(m.group(2)) // (tiles * int(m.group(3)))] = row
    return out


t_out, i_out = {}, {}
for kernel in ('k_frame_group_ahead', 'k_frame_group'):
    f, w, n = by_frames(fetch, kernel), by_frames(write, kernel), by_frames(insts, kernel)
    for frames in sorted(set(f) & set(w)):
        fb, wb = cal_f * f[frames]['FETCH_SIZE'], cal_w * w[frames]['WRITE_SIZE']
        t_out.setdefault(kernel, {})[str(frames)] = {'launches': int(min(f[frames]['launches'], w[frames]['launches'])), 'fetch_bytes': fb, 'write_bytes': wb,
                                                      'hbm_bytes': fb + wb, 'hbm_bytes_per_frame': (fb + wb) / frames}
    for frames in sorted(n):
        r = n[frames]
        i_out.setdefault(kernel, {})[str(frames)] = {'launches': int(r['launches']), 'valu': r.get('SQ_INSTS_VALU'), 'salu': r.get('SQ_INSTS_SALU'), 'lds': r.get('SQ_INSTS_LDS'),
                                                      'waves': r.get('SQ_WAVES'), 'valu_per_frame': r.get('SQ_INSTS_VALU', 0.0) / frames}
note = 'mean per launch of the scatter alone on a stream (tools/ahead_time.py), by frames per launch'
traffic['by_frames_per_launch'] = dict(t_out, note=note)
counts['by_frames_per_launch'] = dict(i_out, note=note)
json.dump(traffic, open(traffic_path, 'w'), indent=1)
json.dump(counts, open(insts_path, 'w'), indent=1)
for k, v in t_out.items():
    for fr, row in v.items():
        print('%s, %s frames per launch: %.1f MB per frame (%.1f in, %.1f out)' % (k, fr, row['hbm_bytes_per_frame'] / 1e6, row['fetch_bytes'] / int(fr) / 1e6, row['write_bytes'] / int(fr) / 1e6))
for k, v in i_out.items():
    for fr, row in v.items():
        print('%s, %s frames per launch: %.3f M VALU per frame' % (k, fr, row['valu_per_frame'] / 1e6))
