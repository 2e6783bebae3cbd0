"""CPU checks behind the table-driven hole fill of ken-burns-effect_amd/csrc/kbe_holes.hip (k_hole_dist, struct Axis), as C
restatements of its arithmetic against brute force (no GPU, no oracle):
  * tools/advance_check.c -- m fp32 additions of a fill direction taken on the integer mantissa (axis_jump,
    axis_catch_up, advance_exact) against the additions one at a time (common.py:876-889): bits and pixels identical;
  * tools/strip_proto.c -- the strip test (build_strips) against brute-force walks on a mask: no direction that
    completes (both ends reach a valid pixel before leaving the image, common.py:880-896) is ever skipped."""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-o', exe, os.path.join(ROOT, 'tools', name + '.c'), '-lm'])
    return exe


def test_many_fp32_additions_at_once_equal_the_additions_one_at_a_time(tmp_path):
    out = subprocess.run([_build(tmp_path, 'advance_check')], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    m = re.search(r'walks (\d+), advances (\d+) \(catch-ups (\d+)\), mismatches (\d+)', out.stdout)
    assert m and int(m.group(4)) == 0 and int(m.group(2)) > 1000000 and int(m.group(3)) > 100000, out.stdout[-500:]


def test_strip_test_never_skips_a_direction_that_completes(tmp_path):
    """A zoomed-out frame in miniature: a trapezoid with a ragged, speckled rim and a tower beside it."""
    rng = np.random.default_rng(5)
    H = W = 1024
    yy, xx = np.mgrid[0:H, 0:W]
    half = 120 + (yy - 300) * 0.45
    mask = (yy >= 300) & (yy < 880) & (np.abs(xx - 500) < half)
    mask |= (xx >= 720) & (xx < 880) & (yy < 270)
    rim = (yy >= 290) & (yy < 900) & (np.abs(np.abs(xx - 500) - half) < 25)
    mask = np.where(rim, rng.random((H, W)) < 0.3, mask)
    mask[400:420, 380:520] = False                                   # a hole inside: directions complete here
    path = str(tmp_path / 'mask.u8')
    mask.astype(np.uint8).tofile(path)
    out = subprocess.run([_build(tmp_path, 'strip_proto'), path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    m = re.search(r'complete (\d+) .*survive strip test (\d+) .*false kills (\d+)', out.stdout)
    assert m, out.stdout[-500:]
    complete, survive, false_kills = (int(g) for g in m.groups())
    assert false_kills == 0 and complete > 100000 and survive >= complete
    pairs = int(re.search(r'pairs (\d+)', out.stdout).group(1))
    assert survive < 0.6 * pairs, 'the test skips a good share of the directions at once'
