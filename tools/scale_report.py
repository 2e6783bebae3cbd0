#!/usr/bin/env python3
"""ONE command for the day a multi-GPU node appears: `bench.py --gpus N` for N = 1, 2, 4, 8 --
weak scaling (K frames per rank) and strong scaling (`--video-frames 128`: one video sharded over the ranks) -- each N launched the
way the driver launches it (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1), and a report:

  per N     frames/s of the whole job, ms per pass, `ranks_seen`, the collectives' backend, `cloud_broadcast_ms`;
  per rank  its device, the NUMA node it bound to or WHY it did not (sharding.NUMA_BIND), its own ms per pass and what that is on
            its PCIe link (a rank whose link or socket lags shows here, not in the max over ranks);
  the curve frames/s(N) / frames/s(1) -- printed ONLY when every line says `scaling_valid` (ranks on devices of their own, RCCL
            up): a run where ranks shared devices, or where the collectives fell back to gloo, is a functional check and is
            reported as such, without a curve.

    python tools/scale_report.py                         # N = 1, 2, 4, 8 as far as the box has GPUs
    python tools/scale_report.py --gpus 1,2 --dry-run    # ranks share the GPUs there are, collectives on gloo: does every path run?
    python tools/scale_report.py --out profiles/r06_scale.json

--dry-run sets KBE_DIST_BACKEND=gloo (bench.py then lets ranks share devices and marks its line `scaling_valid: false`): what a 1-GPU
box can check -- launcher, broadcast, sharding, per-rank accounting -- and what tests/test_bench_launcher.py checks (the curve's refusal on CPU, the dry run itself on the GPU box)."""
import argparse
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_bench(n, extra, env, timeout):
    """One bench.py run on n ranks -> its JSON line (dict), or {'error': ...}."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--no-cpu-baseline'] + extra
    if n > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', str(free_port())] + cmd[1:]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {'error': 'timed out after %d s' % timeout, 'cmd': ' '.join(cmd)}
    for text in reversed(p.stdout.strip().splitlines()):
        try:
            doc = json.loads(text)
            if isinstance(doc, dict) and 'metric' in doc:
                return doc
        except ValueError:
            continue
    return {'error': 'no JSON line (exit %d)' % p.returncode, 'stderr': p.stderr[-1500:], 'cmd': ' '.join(cmd)}


def summarise(n, mode, doc):
    if 'error' in doc:
        return {'n_gpus': n, 'mode': mode, 'error': doc['error'], 'detail': doc.get('stderr', '')}
    cfg = doc.get('config', {})
    return {'n_gpus': n, 'mode': mode, 'value': doc.get('value'), 'unit': doc.get('unit'), 'ms_per_step': doc.get('ms_per_step'), 'scaling': doc.get('scaling'),
            'device_only': (doc.get('device_only') or {}).get('value'), 'ranks_seen': cfg.get('ranks_seen', 1), 'collectives': cfg.get('collectives'),
            'scaling_valid': doc.get('scaling_valid', n == 1), 'scaling_note': doc.get('scaling_note'), 'cloud_broadcast_ms': doc.get('cloud_broadcast_ms'),
            'pcie_rank0_GBs': (doc.get('pcie') or {}).get('achieved'), 'ranks': doc.get('ranks'), 'frames_check': doc.get('frames_check')}


def curve_of(rows, mode):
    """frames/s(N) / frames/s(1) of the mode's rows -- ONLY from lines that are scaling measurements (every one `scaling_valid`, none
    failed) and only against an N = 1 line of the same mode.  -> (curve, None) or (None, why not)."""
    mine = [r for r in rows if r['mode'] == mode]
    failed = [r['n_gpus'] for r in mine if 'error' in r]
    if failed:
        return None, 'N = %s failed' % ', '.join(map(str, failed))
    invalid = [r['n_gpus'] for r in mine if not r['scaling_valid']]
    if invalid:
        return None, 'N = %s ran with scaling_valid false (ranks sharing devices, or collectives not on RCCL)' % ', '.join(map(str, invalid))
    base = next((r for r in mine if r['n_gpus'] == 1), None)
    if base is None or not base['value']:
        return None, 'no N = 1 line to divide by'
    return {str(r['n_gpus']): round(r['value'] / base['value'], 3) for r in mine}, None


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--gpus', default='1,2,4,8', help='rank counts to run (comma-separated)')
    ap.add_argument('--steps', type=int, default=75, help='weak scaling: frames per rank and pass')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--video-frames', type=int, default=128, help='strong scaling: frames of the one video (0: skip)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dry-run', action='store_true', help='ranks may share devices, collectives on gloo: a functional check, never a curve')
    ap.add_argument('--timeout', type=int, default=900)
    ap.add_argument('--out', default=None, help='write the report as JSON here as well')
    args = ap.parse_args()

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if args.dry_run:
        env['KBE_DIST_BACKEND'] = 'gloo'
    try:
        import torch
        n_devices = torch.cuda.device_count()
    except Exception:                               # noqa: BLE001
        n_devices = 0
    counts = [int(v) for v in args.gpus.split(',') if v]
    if not args.dry_run:
        skipped = [n for n in counts if n > n_devices]
        counts = [n for n in counts if n <= n_devices]
        if skipped:
            print('this box has %d GPU(s): N = %s not run (use --dry-run for a functional check with shared devices)' % (n_devices, ', '.join(map(str, skipped))))
    rows = []
    for mode, extra in (('weak', ['--steps', str(args.steps), '--warmup', str(args.warmup), '--size', str(args.size)]),
                        ('strong', ['--video-frames', str(args.video_frames), '--warmup', str(args.warmup), '--size', str(args.size)])):
        if mode == 'strong' and args.video_frames <= 0:
            continue
        for n in counts:
            row = summarise(n, mode, run_bench(n, extra, env, args.timeout))
            rows.append(row)
            if 'error' in row:
                print('%-6s N = %d: FAILED -- %s\n%s' % (mode, n, row['error'], row['detail']))
                continue
            print('%-6s N = %d: %10.1f %s  (%.3f ms per pass; left in HBM %s)  ranks seen %d over %s%s%s' % (
                mode, n, row['value'] or 0.0, row['unit'], row['ms_per_step'] or 0.0, '%.0f' % row['device_only'] if row['device_only'] else '-',
                row['ranks_seen'], row['collectives'] or 'no collectives', '' if row['cloud_broadcast_ms'] is None else ', cloud broadcast %.2f ms' % row['cloud_broadcast_ms'],
                '' if row['scaling_valid'] else '   ** NOT a scaling measurement: %s' % (row['scaling_note'] or 'scaling_valid false')))
            for r in row['ranks'] or []:
                print('         rank %d on device %d: NUMA node %s (%s); %d frames, %.3f ms per pass%s' % (
                    r['rank'], r['device'], r['numa_node'], r['numa'], r['frames'], r['ms_per_pass'], '' if r['pcie_GBs'] is None else ', %.1f GB/s on its link' % r['pcie_GBs']))
    report = {'rows': rows, 'dry_run': args.dry_run, 'devices': n_devices}
    for mode in ('weak', 'strong'):
        if not any(r['mode'] == mode for r in rows):
            continue
        curve, why = curve_of(rows, mode)
        report[mode + '_curve'] = curve
        if curve is None:
            print('%s scaling: NO CURVE -- %s' % (mode, why))
        else:
            print('%s scaling, frames/s(N) / frames/s(1): %s' % (mode, '  '.join('N=%s: %.2fx' % kv for kv in curve.items())))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(report, f, indent=1)
    return 0 if all('error' not in r for r in rows) else 1


if __name__ == '__main__':
    sys.exit(main())
