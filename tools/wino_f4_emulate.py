#!/usr/bin/env python3
"""Whole-network parity room for Winograd F(2x2x4, 3x3x3) bricks -- CPU, torch fp32, no GPU.

The train step of BASELINE configs[1] (UNet(1, 2, n_blocks=4, start_filts=32), batch 2 of 64x128x128) is run with the reference's ATen op sequence
(oracle/torch_ref.py::unet_forward) in fp32, except that the forward and the data gradient of the 3x3x3 convs with >= 32 input channels at levels 0 and 1
(the shapes the persistent Winograd kernel serves: /root/reference/elektronn3/models/unet.py:131-149) are computed by an fp32 EMULATION of nested
Winograd tiles: explicit fp32 transform passes (one rounded multiply / add per coefficient, no fma), the channel contraction as an fp32 chain over
8-channel chunks (the kernels' k-loop), weights transformed in fp64 and rounded once (the packer).  The result is judged by EXACTLY the assertions of
tests/test_unet_gpu.py::test_full_size_against_the_reference_digest against tests/golden/cfg2_digest.npz (digest of the reference's own fp32 / fp64 runs).

    python tools/wino_f4_emulate.py direct f222 f224 f224:0,1,-1,1/2,-2 [--levels 0,1] [--out profiles/r05_f224_emulation.md]

`f224-dgrad` = forward on today's F(2x2x2) tiles, only the data gradient on F(2x2x4) (ReLU / arg-max decisions are taken in the forward pass).
`direct` = ATen's conv everywhere (calibration: the reference's fp32 run itself), `f222` = the tiles conv3_wino_pkernel computes today, `f224[:points]` =
F(4,3) along W with the given interpolation points (default 0,1,-1,2,-2; infinity is always the last point).
"""
import os
import sys
import time
from collections import OrderedDict
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


# ------------------------------------------------------------------------------------------------ Toom-Cook matrices for F(m, 3) from points
def toom_cook(m, points, r=3):
    """(AT [m x n], G [n x r], BT [n x n]) as Fractions for F(m, r) with the finite points given (+ infinity), n = m + r - 1."""
    n = m + r - 1
    a = [Fraction(p) for p in points]
    assert len(a) == n - 1 and len(set(a)) == n - 1
    AT = [[(a[j] ** i if j < n - 1 else Fraction(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for i in range(n - 1):
        N = Fraction(1)
        for k in range(n - 1):
            if k != i:
                N *= (a[i] - a[k])
        G.append([a[i] ** j / N for j in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    # BT from the identity  sum_k AT[i][k] G[k][j] BT[k][l] = [l == i + j]  (exact Gaussian elimination per column l)
    rows = [(i, j) for i in range(m) for j in range(r)]
    M = [[AT[i][k] * G[k][j] for k in range(n)] for (i, j) in rows]
    BT = [[Fraction(0)] * n for _ in range(n)]
    for l in range(n):
        rhs = [Fraction(int(l == i + j)) for (i, j) in rows]
        sol = _solve(M, rhs, n)
        for k in range(n):
            BT[k][l] = sol[k]
    return AT, G, BT


def _solve(M, rhs, n):
    A = [row[:] + [b] for row, b in zip(M, rhs)]
    piv = []
    r = 0
    for c in range(n):
        p = next((i for i in range(r, len(A)) if A[i][c] != 0), None)
        assert p is not None, 'singular'
        A[r], A[p] = A[p], A[r]
        s = A[r][c]
        A[r] = [v / s for v in A[r]]
        for i in range(len(A)):
            if i != r and A[i][c] != 0:
                f = A[i][c]
                A[i] = [vi - f * vr for vi, vr in zip(A[i], A[r])]
        piv.append(c)
        r += 1
    assert all(all(v == 0 for v in row) for row in A[n:]), 'inconsistent'
    return [A[i][n] for i in range(n)]


def rescale(AT, G, BT, scales):
    """Row k of G times s_k, row k of BT divided by s_k (the identity is unchanged): moves constants between the packer (free) and the kernel."""
    G = [[v * Fraction(s) for v in row] for row, s in zip(G, scales)]
    BT = [[v / Fraction(s) for v in row] for row, s in zip(BT, scales)]
    return AT, G, BT


def mats(m, points=None, scales=None):
    if m == 2:
        points = points or (0, 1, -1)
    else:
        points = points or (0, 1, -1, 2, -2)
    AT, G, BT = toom_cook(m, points)
    if scales is not None:
        AT, G, BT = rescale(AT, G, BT, scales)
    f = lambda M: np.array([[float(v) for v in row] for row in M], np.float64)
    return f(AT), f(G), f(BT)


# ------------------------------------------------------------------------------------------------ fp32 emulation of one nested-Winograd conv
def _apply(T, parts):
    """out_i = sum_j T[i][j] * parts[j] in fp32, one rounding per multiply and per add, zeros skipped, +-1 without a multiply."""
    out = []
    for row in T:
        acc = None
        for c, p in zip(row, parts):
            if c == 0:
                continue
            c32 = np.float32(c)
            term = p if c32 == 1 else (-p if c32 == -1 else p * float(c32))
            acc = term if acc is None else acc + term
        out.append(acc)
    return out


def wino_conv3(x, w, ms, mat, chunk=8):
    """'same' 3x3x3 conv of x [N, Ci, D, H, W] with w [Co, Ci, 3, 3, 3] by Winograd tiles ms = (mD, mH, mW); fp32 emulation."""
    N, Ci, D, H, W = x.shape
    Co = w.shape[0]
    tiles = [-(-s // m) for s, m in zip((D, H, W), ms)]
    pad = [t * m - s for t, m, s in zip(tiles, ms, (D, H, W))]
    U = w.double()
    for ax, m in enumerate(ms):          # packer: G w G^T in fp64, rounded once
        Gm = torch.from_numpy(mat[m][1])
        U = torch.movedim(torch.tensordot(Gm, U, dims=([1], [ax + 2])), 0, ax + 2)
    a = [m + 2 for m in ms]
    P = a[0] * a[1] * a[2]
    U = U.float().permute(2, 3, 4, 1, 0).reshape(P, Ci, Co).contiguous()          # [pos, ci, co]
    outs = []
    for n in range(N):
        xp = F.pad(x[n], (1, 1 + pad[2], 1, 1 + pad[1], 1, 1 + pad[0]))
        V = xp.unfold(1, a[0], ms[0]).unfold(2, a[1], ms[1]).unfold(3, a[2], ms[2])      # [Ci, tD, tH, tW, aD, aH, aW] (view)
        for ax, m in enumerate(ms):
            dim = 4 + ax
            parts = _apply(mat[m][2], list(V.unbind(dim)))
            V = torch.stack(parts, dim)
        T = tiles[0] * tiles[1] * tiles[2]
        V = V.permute(4, 5, 6, 1, 2, 3, 0).reshape(P, T, Ci)      # [pos, tile, ci]
        M = None
        for c0 in range(0, Ci, chunk):       # fp32 chain over 8-channel chunks
            part = torch.bmm(V[:, :, c0:c0 + chunk], U[:, c0:c0 + chunk])
            M = part if M is None else M.add_(part)
        del V
        M = M.reshape(a[0], a[1], a[2], tiles[0], tiles[1], tiles[2], Co)
        for ax, m in enumerate(ms):
            parts = _apply(mat[m][0], list(M.unbind(ax)))
            M = torch.stack(parts, ax)
        # [mD, mH, mW, tD, tH, tW, Co] -> [Co, D, H, W]
        y = M.permute(6, 3, 0, 4, 1, 5, 2).reshape(Co, tiles[0] * ms[0], tiles[1] * ms[1], tiles[2] * ms[2])[:, :D, :H, :W]
        outs.append(y)
    return torch.stack(outs, 0)


class WinoConv(torch.autograd.Function):
    """conv3d(x, w, b, padding=1): forward and data gradient on emulated Winograd tiles, weight gradient by ATen (the weight-gradient kernel does not change)."""
    @staticmethod
    def forward(ctx, x, w, b, ms, mat, ms_fwd=None):
        ctx.save_for_backward(x, w)
        ctx.ms, ctx.mat = ms, mat
        return wino_conv3(x, w, ms_fwd or ms, mat) + b.view(1, -1, 1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
        dx = wino_conv3(dy.contiguous(), wt, ctx.ms, ctx.mat)
        dw = torch.nn.grad.conv3d_weight(x, w.shape, dy, padding=1)
        return dx, dw, dy.sum((0, 2, 3, 4)), None, None, None


# ------------------------------------------------------------------------------------------------ the digest's assertions
def evaluate(case, mode_conv, levels, log):
    from helpers import load_npz, digest_state_dict, digest_inputs, digest_of, is_prebn_bias, sub
    import oracle.torch_ref as tr
    g = load_npz(case)
    seed = int(g['seed'])
    shapes = OrderedDict((str(k), tuple(int(i) for i in str(sh).split(',')) if str(sh) else ()) for k, sh in zip(g['names'], g['shapes']))
    sd0 = digest_state_dict(shapes, seed)
    nb = int(g['cfg.n_blocks'])
    x_np, t_np = digest_inputs(int(g['batch']), tuple(int(v) for v in g['shape']), seed)
    x, t = torch.from_numpy(x_np), torch.from_numpy(t_np)
    sd = {k: (torch.from_numpy(v).clone().requires_grad_(v.dtype != np.int64 and 'running' not in k)) for k, v in sd0.items()}
    full = tuple(int(v) for v in g['shape'])
    orig_conv = tr._conv

    def conv(xx, sdd, name):
        w = sdd[name + '.weight']
        if mode_conv is not None and w.dim() == 5 and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[1] >= 32:
            lvl = int(round(np.log2(full[0] / xx.shape[2])))
            if lvl in levels:
                return mode_conv(xx, w, sdd[name + '.bias'])
        return orig_conv(xx, sdd, name)
    tr._conv = conv
    try:
        t0 = time.time()
        out = tr.unet_forward(sd, x, nb, (), training=True)
        loss = tr.combined_loss(out, t)
        loss.backward()
        log(f'  step took {time.time() - t0:.0f} s, loss {float(loss):.7f} (fp64 reference {float(g["loss64"]):.7f}, fp32 reference {float(g["loss32"]):.7f})')
    finally:
        tr._conv = orig_conv
    res = OrderedDict()
    samp = out.detach()[:, :, ::8, ::8, ::8].numpy()
    err_ref = float(g['logits_err_ref'])
    e64 = float(np.abs(samp - g['logits64']).max())
    res['logits'] = dict(err=e64, err_ref=err_ref, bound=max(3 * err_ref, 2e-5), ok=e64 <= max(3 * err_ref, 2e-5),
                         ok32=bool(np.allclose(samp, g['logits32'], rtol=1e-4, atol=1e-4)))
    res['loss'] = dict(err=abs(float(loss.detach()) - float(g['loss64'])), bound=2e-5, ok=abs(float(loss.detach()) - float(g['loss64'])) < 2e-5)
    worst_rs = 0.0
    for k, v in sub(g, 'sd1').items():
        if 'running' in k and k in sd:
            a = sd[k].detach().numpy()
            worst_rs = max(worst_rs, float(np.max(np.abs(a - v) / (1e-6 + 1e-5 * np.abs(v)))))
    res['running_stats'] = dict(err=worst_rs, bound=1.0, ok=worst_rs <= 1.0)
    names = {k for k in sd if sd[k].requires_grad}
    gnorm = np.sqrt(sum(float(g['g/' + k][0]) ** 2 for k in names))
    rows = []
    for k in shapes:
        if k not in names:
            continue
        rec = g['g/' + k]
        n64, err_own = float(rec[0]), float(rec[2])
        p64 = rec[3:7]
        ns = int(rec[11]); s64 = rec[12:12 + ns]
        gr = sd[k].grad.detach().numpy()
        if is_prebn_bias(k, names):
            rows.append(dict(name=k, prebn=True, est=float(np.abs(gr).max()), bound=1e-5 * gnorm, ok=float(np.abs(gr).max()) <= 1e-5 * gnorm, err_own=0.0))
            continue
        _, s_h, p_h = digest_of(k, gr, seed)
        est = float(np.sqrt(np.mean((p_h - p64) ** 2))) / max(n64, 1e-30)
        smp = float(np.linalg.norm(s_h - s64) / max(np.linalg.norm(s64), 1e-30))
        bound = 2 * max(3 * err_own, 1e-4)
        ok_hard = est <= 1e-2 and (smp <= 1e-2 or np.linalg.norm(s64) < 1e-3 * n64)
        rows.append(dict(name=k, prebn=False, est=est, smp=smp, err_own=err_own, bound=bound, ok_hard=ok_hard, ok_tight=est <= bound, ok=ok_hard and (est <= bound or est <= 4e-3)))
    res['grads'] = rows
    return res


def parse_mode(spec):
    if spec == 'direct':
        return None, 'direct (ATen fp32 everywhere)'
    name, _, pts = spec.partition(':')
    dgrad_only = name.endswith('-dgrad')       # forward on the F(2x2x2) tiles of today's kernel, only the DATA GRADIENT on the larger tiles
    name = name[:-6] if dgrad_only else name
    ms = {'f222': (2, 2, 2), 'f224': (2, 2, 4), 'f244': (2, 4, 4), 'f444': (4, 4, 4)}[name]
    mat = {2: mats(2)}
    if 4 in ms:
        scales = None
        if '@' in pts:
            pts, _, sc = pts.partition('@')
            scales = [Fraction(s) for s in sc.split(',')]
        mat[4] = mats(4, [Fraction(p) for p in pts.split(',')] if pts else None, scales)
    return (lambda x, w, b: WinoConv.apply(x, w, b, ms, mat, (2, 2, 2) if dgrad_only else None)), f'{name} tiles' + (' for the data gradient only (forward: F(2x2x2))' if dgrad_only else '') + (f', F(4,3) points {pts or "0,1,-1,2,-2"}' if 4 in ms else '')


def main():
    argv = sys.argv[1:]
    levels, outp, case = (0, 1), None, 'cfg2_digest.npz'
    modes = []
    i = 0
    while i < len(argv):
        if argv[i] == '--levels':
            levels = tuple(int(v) for v in argv[i + 1].split(',')); i += 2
        elif argv[i] == '--out':
            outp = argv[i + 1]; i += 2
        elif argv[i] == '--case':
            case = argv[i + 1]; i += 2
        else:
            modes.append(argv[i]); i += 1
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)
    torch.manual_seed(0)
    log(f'# Winograd tile emulation against {case} (tools/wino_f4_emulate.py; levels {levels}; torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads)')
    summary = []
    for spec in modes:
        fn, desc = parse_mode(spec)
        log(f'\n## {spec}: {desc}')
        r = evaluate(case, fn, levels, log)
        lg = r['logits']
        log(f'  logits: max |emu - fp64| {lg["err"]:.3e}  (reference fp32 itself {lg["err_ref"]:.3e}, bound {lg["bound"]:.3e}) {"PASS" if lg["ok"] else "FAIL"}; allclose to the fp32 reference at 1e-4: {"PASS" if lg["ok32"] else "FAIL"}')
        log(f'  loss: |emu - fp64| {r["loss"]["err"]:.2e} (bound 2e-5) {"PASS" if r["loss"]["ok"] else "FAIL"}; running statistics (rtol 1e-5, atol 1e-6): worst {r["running_stats"]["err"]:.2f} of the tolerance {"PASS" if r["running_stats"]["ok"] else "FAIL"}')
        log('  | gradient tensor | est. rel-L2 vs fp64 | reference fp32 itself | ratio | tight bound 2*max(3*own,1e-4) | tight | test (tight or <= 4e-3, and <= 1e-2) |')
        log('  |---|---|---|---|---|---|---|')
        nfail = ntight = 0
        worst = (0.0, '')
        for row in r['grads']:
            if row['prebn']:
                if not row['ok']:
                    nfail += 1
                    log(f'  | {row["name"]} (pre-BN bias) | max {row["est"]:.2e} | - | - | {row["bound"]:.2e} | - | FAIL |')
                continue
            ratio = row['est'] / max(row['err_own'], 1e-30)
            if ratio > worst[0]:
                worst = (ratio, row['name'])
            nfail += not row['ok']; ntight += not row['ok_tight']
            log(f'  | {row["name"]} | {row["est"]:.2e} | {row["err_own"]:.2e} | {ratio:.2f} | {row["bound"]:.2e} | {"ok" if row["ok_tight"] else "over"} | {"PASS" if row["ok"] else "FAIL"} |')
        allok = lg['ok'] and lg['ok32'] and r['loss']['ok'] and r['running_stats']['ok'] and nfail == 0
        log(f'  => {"ALL ASSERTIONS PASS" if allok else "ASSERTIONS FAIL"}: {nfail} gradient tensors fail the test, {ntight} exceed the tight bound (pass through the unconditional 4e-3); worst ratio to the reference\'s own error {worst[0]:.2f} at {worst[1]}')
        summary.append((spec, lg['err'], lg['err_ref'], nfail, ntight, worst[0], allok))
    log('\n## summary')
    log('| mode | logits err vs fp64 | reference fp32 | failing grads | over tight bound | worst est / own | verdict |')
    log('|---|---|---|---|---|---|---|')
    for s in summary:
        log(f'| {s[0]} | {s[1]:.2e} | {s[2]:.2e} | {s[3]} | {s[4]} | {s[5]:.2f} | {"pass" if s[6] else "FAIL"} |')
    if outp:
        with open(outp, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
