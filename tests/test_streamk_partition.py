"""The partition of the cross-layer stream-K weight-gradient launch (csrc/kernels.h WSkPart, csrc/wgrad_wino.hip wgrad_sk_partition) -- host arithmetic, CPU tier:
every unit of work is taken exactly once, a (block, tile pair) cell meets at most two workgroups (what the reduction relies on), no two segments share a slab,
and the slab pool sized at plan time is large enough.  The library's own (internal, C++-mangled) host function is called through ctypes; the unit decode of the
kernels (wsk_unit, a header-inline function) is restated here in ten lines."""
import ctypes
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAXL = 16


class Layer(ctypes.Structure):
    _fields_ = [('dw', ctypes.c_void_p), ('Cin', ctypes.c_int), ('Cout', ctypes.c_int), ('ci_tiles', ctypes.c_int), ('tps', ctypes.c_int), ('nbricks', ctypes.c_int),
                ('B', ctypes.c_int), ('nblocks', ctypes.c_int), ('g0', ctypes.c_uint), ('t0', ctypes.c_uint), ('c0', ctypes.c_uint)]


class Part(ctypes.Structure):
    _fields_ = [('L', Layer * MAXL), ('n', ctypes.c_int), ('total', ctypes.c_uint), ('q', ctypes.c_uint), ('r', ctypes.c_uint), ('ntp', ctypes.c_uint),
                ('ncells', ctypes.c_uint), ('nwg', ctypes.c_uint), ('slab', ctypes.c_void_p)]


@pytest.fixture(scope='module')
def lib():
    from elektronn3_amd.build import build
    so = ctypes.CDLL(build())
    try:
        fn = so._Z18wgrad_sk_partitionR7WSkPartiPKiS2_S2_PKPfiS3_m
        slabs = so._Z20wgrad_sk_slab_floatsii
    except AttributeError:
        pytest.skip('internal symbol names changed (another compiler ABI)')
    fn.restype = ctypes.c_int
    slabs.restype = ctypes.c_size_t
    return fn, slabs


def unit(L, rel):
    """kernels.h wsk_unit: unit `rel` of a layer -> (block, tile pair, first brick, bricks left in the cell)"""
    per = L.B * L.tps
    b = min(rel // per, L.nblocks - 1)
    bsz = L.nbricks - b * L.B if b + 1 == L.nblocks else L.B
    r2 = rel - b * per
    tp, off = divmod(r2, bsz)
    return b, tp, b * L.B + off, bsz - off


def walk(p):
    """every workgroup's segments: (workgroup, layer, block, tile pair, brick0, nbricks, slab id)"""
    start = lambda i: i * p.q + min(i, p.r)
    for wg in range(p.nwg):
        g, gend, l = start(wg), start(wg + 1), 0
        while g < gend:
            while l + 1 < p.n and p.L[l + 1].g0 <= g:
                l += 1
            L = p.L[l]
            b, tp, brick, left = unit(L, g - L.g0)
            nb = min(gend - g, left)
            yield wg, l, b, tp, brick, nb, wg + L.c0 + b * L.tps + tp
            g += nb


@pytest.mark.parametrize('nwg', [256, 512])
def test_partition_covers_every_unit_once_and_never_shares_a_slab(lib, nwg):
    fn, slab_floats = lib
    rng = random.Random(7 + nwg)
    cases = [[(32, 32, 16384), (64, 32, 16384), (32, 32, 16384), (32, 64, 2048), (64, 64, 2048), (128, 64, 2048), (64, 64, 2048),
              (64, 128, 256), (128, 128, 256), (256, 128, 256), (128, 128, 256), (128, 256, 32), (256, 256, 32)],      # cfg 2, every layer
             [(8, 8, 1)], [(40, 24, 3), (8, 72, 5)], [(256, 256, 7)] * 16]
    for _ in range(30):
        cases.append([(8 * rng.randint(1, 40), 8 * rng.randint(1, 40), rng.choice([1, 2, 5, 31, 64, 257, 1000, 4096, 20000])) for _ in range(rng.randint(1, MAXL))])
    for layers in cases:
        n = len(layers)
        Cin = (ctypes.c_int * n)(*[c[0] for c in layers]); Cout = (ctypes.c_int * n)(*[c[1] for c in layers]); nbr = (ctypes.c_int * n)(*[c[2] for c in layers])
        dw = (ctypes.c_void_p * n)()
        tile_pairs = sum(-(-ci // 32) * -(-co // 32) for ci, co, _ in layers)
        pool = slab_floats(tile_pairs, nwg)
        p = Part()
        assert fn(ctypes.byref(p), n, Cin, Cout, nbr, dw, nwg, None, ctypes.c_size_t(pool)) == 0, layers
        assert p.n == n and p.nwg == nwg and p.total == sum(-(-ci // 32) * -(-co // 32) * nb for ci, co, nb in layers) and p.ntp == tile_pairs
        seen = {}           # (layer, tile pair) -> bricks covered
        slabs = set()
        cell_wgs = {}       # (layer, block, tile pair) -> workgroups
        for wg, l, b, tp, brick, nb, sid in walk(p):
            L = p.L[l]
            assert 0 <= tp < L.tps and 0 <= b < L.nblocks and nb > 0 and brick + nb <= min(L.nbricks, (b + 1) * L.B), (layers, wg, l, b, tp, brick, nb)
            cov = seen.setdefault((l, tp), [])
            cov.append((brick, brick + nb))
            assert sid not in slabs and sid * 27648 + 27648 <= pool, (layers, sid)
            slabs.add(sid)
            cell_wgs.setdefault((l, b, tp), []).append(wg)
        for l in range(n):
            L = p.L[l]
            for tp in range(L.tps):
                iv = sorted(seen[(l, tp)])
                assert iv[0][0] == 0 and iv[-1][1] == L.nbricks and all(a[1] == b_[0] for a, b_ in zip(iv, iv[1:])), (layers, l, tp, iv[:4])
        for key, wgs in cell_wgs.items():
            assert len(wgs) <= 2 and (len(wgs) == 1 or wgs[1] == wgs[0] + 1), (layers, key, wgs)     # one workgroup boundary at most: wgrad_sk_reduce_kernel reads slabs w0, w0 + 1
