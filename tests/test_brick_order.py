"""The logical brick order of the persistent Winograd kernels (elektronn3_amd/csrc/brick_order.h): its device functions compiled as host code
(tests/brick_order_check.cpp, g++) -- the decode is a bijection onto the brick grid, a cursor advanced by the step's digits equals the decode of the
advanced index for every grid shape (ragged brick counts, several column tiles and samples, steps that are no power of two: a CU reserve), and the set
of bricks an XCD works on at a time is one compact block."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('brick_order') / 'brick_order_check')
    subprocess.run(['g++', '-O1', '-o', exe, os.path.join(ROOT, 'tests', 'brick_order_check.cpp')], check=True)
    return exe


# (ntiles, tilesW, tilesH, tilesD, N, step, concurrent, w_run) -> expected log2 block (w, h, d)
CASES = {
    'cfg1 level 0 (32 -> 32 ch, 128^3)': ((1, 8, 32, 32, 1, 32, 32, 64), (2, 1, 2)),
    'cfg1 level 0, runs of 32 voxels': ((1, 8, 32, 32, 1, 32, 32, 32), (1, 2, 2)),
    'cfg1 level 1 (64 ch, 64^3)': ((2, 4, 16, 16, 1, 32, 32, 64), (2, 1, 1)),
    'cfg1 level 2 (128 ch, 32^3), batch 2': ((4, 2, 8, 8, 2, 32, 32, 64), (1, 1, 1)),
    'cfg5 tile level 0 (96 x 160 x 160)': ((1, 10, 40, 24, 1, 32, 32, 64), (1, 2, 2)),
    'odd brick counts: the plain order': ((1, 5, 7, 3, 2, 32, 32, 64), (0, 0, 0)),
    'three column tiles, CU reserve': ((3, 6, 4, 2, 1, 29, 29, 32), (1, 1, 1)),
    'level 0 with a CU reserve of 32': ((1, 8, 32, 32, 1, 28, 28, 64), (2, 1, 1)),
    'one brick per workgroup': ((2, 3, 2, 5, 1, 1, 7, 32), (0, 1, 0)),
    'step 0 (grid == bricks in conv3_wino_pkernel)': ((1, 4, 4, 4, 1, 0, 8, 32), (1, 1, 1)),
}


@pytest.mark.parametrize('name', list(CASES))
def test_brick_order(checker, name):
    args, block = CASES[name]
    r = subprocess.run([checker, *map(str, args)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert tuple(int(v) for v in r.stdout.split()[1:]) == block, r.stdout


def test_plain_order_switch(checker):
    r = subprocess.run([checker, '1', '8', '32', '32', '1', '32', '32', '64'], capture_output=True, text=True, env=dict(os.environ, E3_WINO_BLOCK='0'))
    assert r.returncode == 0 and r.stdout.split()[1:] == ['0', '0', '0'], r.stdout
    r = subprocess.run([checker, '1', '8', '32', '32', '1', '32', '32', '64'], capture_output=True, text=True, env=dict(os.environ, E3_WINO_BLOCK='3,0,2'))
    assert r.returncode == 0 and r.stdout.split()[1:] == ['3', '0', '2'], r.stdout
