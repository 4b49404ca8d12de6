#!/usr/bin/env python3
"""Randomised inference forwards: the default tensor layouts (channel-chunked conv1 -> conv2 / concat / pooled tensors, store boxes and transposed-conv boxes
of a needed region) against plain rows and whole tensors (E3_NO_CHUNKED_FWD=1 E3_NO_STORE_BOX=1) -- a change of layout and of what is stored only, so the
logits, the softmax output and the needed-region forward inside its region must agree BIT FOR BIT.  The switches are read once per process: two children run
the same seeded cases and the parent compares.  Both children run with E3_WINO4_MIN=1 (the F(2x2x4) kernel on every grid that has a brick).
Usage: python tools/fuzz_eval_layouts.py [n_cases] [seed]"""
import os, subprocess, sys, tempfile, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(n_cases, seed, path):
    from elektronn3_amd.unet import UNet
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = {}
    for case in range(n_cases):
        nb = ri(2, 4); sf = (16, 32, 32, 64)[ri(0, 3)]; inc = ri(1, 2); outc = ri(2, 4)
        mult = 2 ** (nb - 1)
        big = 3 if os.environ.get('FUZZ_BIG') else 1        # (FUZZ_BIG=1: grids on which the kernels are the default choice, not forced)
        D = ri(1, 4) * big * mult + (ri(0, 3) if ri(0, 1) else 0); H = ri(2, 6) * big * mult + ri(0, 5); W = ri(2, 8) * big * mult + ri(0, 5)
        N = ri(1, 2)
        kw = {}
        v = ri(0, 9)
        if v == 0: kw = dict(normalization='none')
        elif v == 1: kw = dict(full_norm=False)
        elif v == 2: kw = dict(merge_mode='add')
        elif v == 3: kw = dict(up_mode=('resizeconv_nearest', 'resizeconv_linear', 'resizeconv_nearest1')[ri(0, 2)])
        elif v == 4: kw = dict(attention=True)
        elif v == 5: kw = dict(activation=('leaky', 'prelu')[ri(0, 1)])
        elif v == 6: kw = dict(planar_blocks=(0,))
        res = (ri(0, 2), ri(0, 2)) if ri(0, 5) == 0 and v in (7, 8, 9) else None
        torch.manual_seed(1000 + case)
        desc = f'case {case}: nb={nb} sf={sf} in={inc} out={outc} {kw} res={res} N={N} shape={(D, H, W)}'
        try:
            if res is not None:
                from elektronn3_amd.resunet import UNet as ResUNet
                m = ResUNet(in_channels=inc, out_channels=outc, n_blocks=nb, start_filts=sf, enc_res_blocks=res[0], dec_res_blocks=res[1], **kw).cuda().train()
            else:
                m = UNet(in_channels=inc, out_channels=outc, n_blocks=nb, start_filts=sf, **kw).cuda().train()
        except Exception as e:
            print('skip (ctor):', desc, e, flush=True); continue
        x = torch.randn(N, inc, D, H, W, device='cuda')
        with torch.no_grad():
            m(x)                               # running statistics away from their initial values
            m.eval()
            y = m(x)
            sm = m.forward_softmax(x)
            d0, h0, w0 = ri(0, D // 3), ri(0, H // 3), ri(0, W // 3)
            d1, h1, w1 = ri(d0 + 1, D), ri(h0 + 1, H), ri(w0 + 1, W)
            r = m.forward_roi(x[:1], ((d0, d1), (h0, h1), (w0, w1)), softmax=True)[:, :, d0:d1, h0:h1, w0:w1]
        print(desc, 'ok', flush=True)
        out[desc] = (y.cpu(), sm.cpu(), r.cpu())
    torch.save(out, path)


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    res = []
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (('default', {}), ('rows', {'E3_NO_CHUNKED_FWD': '1', 'E3_NO_STORE_BOX': '1'})):
            f = os.path.join(td, tag + '.pt')
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(n), str(seed), f], env={**os.environ, 'E3_WINO4_MIN': '1', **extra},
                               capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stdout[-3000:], r.stderr[-3000:]); sys.exit(1)
            if tag == 'default': print(r.stdout)
            res.append(torch.load(f))
    bad = 0
    for k in res[0]:
        for name, a, b in zip(('logits', 'softmax', 'roi'), res[0][k], res[1][k]):
            if not (torch.isfinite(a).all() and torch.equal(a, b)):
                bad += 1; print('MISMATCH', k, name, float((a - b).abs().max()))
    print(f'{len(res[0])} cases, {bad} mismatches')
    sys.exit(1 if bad else 0)
