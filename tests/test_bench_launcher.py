"""bench.py's multi-rank start-up, on CPU: `python bench.py --gpus 2` must start its own ranks (the driver's first command
shape), and the driver's second shape (`python -m torch.distributed.run ... bench.py --gpus 2`) must be accepted as it is.
KBE_BENCH_LAUNCH_ONLY=1 stops after the rendezvous and the all-reduce of ones (no GPU here); the collectives then run on gloo
and the line says so ("scaling_valid": false)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, KBE_BENCH_LAUNCH_ONLY='1', OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1'], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line['launcher'] == 'ok' and line['world_size'] == 2 and line['ranks_seen'] == 2
    assert line['collectives'].startswith('gloo') and line['scaling_valid'] is False       # never mistaken for xGMI


def test_bench_under_the_drivers_launcher():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1'],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)['ranks_seen'] == 2


def test_bench_rejects_a_world_size_that_does_not_match():
    env = dict(_env(), WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=3' in r.stderr


def test_single_rank_launch_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')], env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout) == {'launcher': 'ok', 'world_size': 1, 'ranks_seen': 1, 'collectives': None, 'scaling_valid': True}


def test_roofline_groups_hold_consecutive_cameras_of_the_path():
    """bench.consecutive_groups: what the scatter's launch is priced on -- every group n consecutive cameras of the path, the last group the
    path's last n (no group wraps from the last camera to the first: no video does), a path shorter than a launch one group of all of it."""
    import bench
    for steps, n in ((75, 12), (20, 12), (1024, 12), (24, 12), (12, 12), (5, 12), (13, 4)):
        path = list(range(steps))
        groups = bench.consecutive_groups(path, n)
        assert all(g == list(range(g[0], g[0] + len(g))) for g in groups), (steps, n)            # consecutive, no wrap
        assert all(len(g) == min(n, steps) for g in groups), (steps, n)
        assert sorted(set(c for g in groups for c in g)) == path, (steps, n)                       # every camera of the path is in some group
        assert len(groups) == max(1, -(-steps // n)) and groups[-1][-1] == steps - 1, (steps, n)


def _scale_report():
    import importlib.util
    spec = importlib.util.spec_from_file_location('scale_report', os.path.join(ROOT, 'tools', 'scale_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scale_report_prints_a_curve_only_from_scaling_measurements():
    """tools/scale_report.py: the curve frames/s(N) / frames/s(1) exists only when every line of the mode is a
    scaling measurement -- ranks on devices of their own, RCCL up (`scaling_valid`) -- none failed, and there is an N = 1 line."""
    sr = _scale_report()
    row = lambda n, v, ok=True, mode='weak': {'n_gpus': n, 'mode': mode, 'value': v, 'scaling_valid': ok}     # noqa: E731
    curve, why = sr.curve_of([row(1, 100.0), row(2, 190.0), row(8, 720.0), row(1, 50.0, mode='strong')], 'weak')
    assert curve == {'1': 1.0, '2': 1.9, '8': 7.2} and why is None
    curve, why = sr.curve_of([row(1, 100.0), row(2, 190.0, ok=False)], 'weak')
    assert curve is None and 'scaling_valid false' in why and 'N = 2' in why
    curve, why = sr.curve_of([row(2, 190.0), row(4, 380.0)], 'weak')
    assert curve is None and 'N = 1' in why
    curve, why = sr.curve_of([row(1, 100.0), {'n_gpus': 2, 'mode': 'weak', 'error': 'timed out'}], 'weak')
    assert curve is None and 'failed' in why
    # a bench line as rank 0 prints it -> the report's row (per-rank accounts, the NUMA node or why there is none)
    doc = {'metric': 'm', 'value': 31000.0, 'unit': 'frames/s', 'ms_per_step': 4.8, 'scaling': 'weak', 'scaling_valid': True, 'cloud_broadcast_ms': 0.4,
           'config': {'ranks_seen': 2, 'collectives': 'nccl'}, 'pcie': {'achieved': 50.1},
           'ranks': [{'rank': 0, 'device': 0, 'numa_node': 0, 'numa': 'bound to 96 CPUs of node 0', 'frames': 75, 'ms_per_pass': 4.7, 'pcie_GBs': 50.1},
                     {'rank': 1, 'device': 1, 'numa_node': None, 'numa': 'the kernel reports no NUMA node for 0000:0b:00.0', 'frames': 75, 'ms_per_pass': 4.8, 'pcie_GBs': 49.2}]}
    r = sr.summarise(2, 'weak', doc)
    assert r['ranks_seen'] == 2 and r['collectives'] == 'nccl' and r['scaling_valid'] and r['ranks'][1]['numa_node'] is None and r['cloud_broadcast_ms'] == 0.4
    assert sr.summarise(2, 'weak', {'error': 'no JSON line (exit 1)', 'stderr': 'x'})['error']


def test_numa_binding_says_what_it_did(monkeypatch):
    """sharding.bind_to_gpu_numa_node never passes silently: NUMA_BIND holds the node, or the reason there is none."""
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import sharding
    assert sharding.bind_to_gpu_numa_node(0) is None             # no GPU here: no PCI address
    assert sharding.NUMA_BIND['node'] is None and sharding.NUMA_BIND['code'] == 2 and 'PCI address' in sharding.NUMA_BIND['reason']
    assert sharding.NUMA_CODES[sharding.NUMA_BIND['code']] in sharding.NUMA_BIND['reason']


def test_pmc_figures_are_quoted_only_from_files_measured_on_these_kernel_sources(tmp_path, monkeypatch, capsys):
    """VERDICT r5 "weak" 7: `roofline.traffic` was a constant read from a committed PMC file and would go stale silently when the kernel
    changed.  The report tools now stamp their JSON with a hash of the kernel sources' CODE (comments and white space taken out), bench.py
    quotes a file only while its stamp is the tree's, and says on stderr when it leaves one out."""
    import json
    import shutil
    import bench
    src = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc')
    fake = tmp_path / 'repo'
    (fake / 'ken-burns-effect_amd' / 'csrc').mkdir(parents=True)
    (fake / 'profiles').mkdir()
    for name in bench.KERNEL_SOURCES:
        shutil.copy(os.path.join(src, name), str(fake / 'ken-burns-effect_amd' / 'csrc' / name))
    monkeypatch.setattr(bench, 'ROOT', str(fake))
    stamp = bench.kernel_sources_stamp()
    assert stamp == bench.kernel_sources_stamp() and len(stamp) == 16
    target = fake / 'ken-burns-effect_amd' / 'csrc' / 'kbe_tiles.h'
    text = target.read_text()
    target.write_text('// a reworded comment\n/* and a block\n   of them */\n' + text.replace('\n', '\n   \n', 3))
    assert bench.kernel_sources_stamp() == stamp, 'comments and white space are not code'
    doc = {'kernels': {'k_frame': {'hbm_bytes': 1.0}}, 'by_frames_per_launch': {}, 'sources_sha16': stamp}
    (fake / 'profiles' / 'r09_hbm_traffic.json').write_text(json.dumps(doc))
    per, where = bench.measured_traffic()
    assert per['k_frame'] == 1.0 and where.endswith('r09_hbm_traffic.json')
    target.write_text(text.replace('constexpr int CNT_STRIDE = 32;', 'constexpr int CNT_STRIDE = 64;'))
    assert bench.kernel_sources_stamp() != stamp, 'a changed constant is code'
    assert bench.measured_traffic() == ({}, None) and 'left out of the line' in capsys.readouterr().err
    (fake / 'profiles' / 'r09_hbm_traffic.json').write_text(json.dumps({'kernels': {'k_frame': {'hbm_bytes': 1.0}}}))        # a file from before the stamp existed
    assert bench.measured_traffic() == ({}, None)
