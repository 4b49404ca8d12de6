#!/usr/bin/env python3
"""Renders profiles/r06_parity_margins.md from the per-tensor tables that tests/test_unet_gpu.py::test_full_size_against_the_reference_digest
writes to gpurun_out/parity_margins_*.json on the GPU box: the distance of every gradient tensor of the full-size train steps (cfg 2, cfg 4) to the
reference's fp64 run, beside the reference's OWN fp32-vs-fp64 error and the two bounds of the test."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(out=os.path.join(ROOT, 'profiles', 'r06_parity_margins.md')):
    files = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'parity_margins_*.json')))
    if not files:
        raise SystemExit('no gpurun_out/parity_margins_*.json: run pytest -m gpu -k reference_digest on the GPU box first')
    L = ['# Parity margins of the full-size train steps (round 6)', '',
         'Source: `tests/test_unet_gpu.py::test_full_size_against_the_reference_digest` on one MI355X; `est` = estimated rel-L2 of (HIP gradient - the reference\'s fp64 gradient) from',
         'four Gaussian projections, `sampled` = the same on the strided sample, `own` = the reference\'s own fp32-vs-fp64 error (from the digest), `tight` = 2 x max(3 x own, 1e-4),',
         'hard cap 1e-2 (SURVEY 8c).  `est / own` says how the HIP path compares with the reference\'s fp32 run; `cap margin` = 1e-2 / est.', '']
    for f in files:
        d = json.load(open(f))
        L += [f'## {d["case"]}  (near ties in the fp64 run: {d["near_ties"]})', '',
              f'logits: max abs error vs fp64 {d["logits_err"]:.2e} (reference fp32: {d["logits_err_ref"]:.2e}, bound {d["logits_bound"]:.2e}); loss error {d["loss_err"]:.1e}', '',
              '| tensor | est | sampled | own (reference fp32) | est / own | tight bound | within tight | cap margin |', '|---|---|---|---|---|---|---|---|']
        for t in sorted(d['tensors'], key=lambda t: -t['est']):
            L.append(f'| {t["tensor"]} | {t["est"]:.2e} | {t["sampled"]:.2e} | {t["err_own"]:.2e} | {t["est"] / max(t["err_own"], 1e-30):.2f} | {t["tight_bound"]:.1e} | '
                     f'{"yes" if t["est"] <= t["tight_bound"] else "no (near-tie allowance 4e-3)"} | {1e-2 / max(t["est"], 1e-30):.1f}x |')
        w = max(d['tensors'], key=lambda t: t['est'])
        L += ['', f'worst: {w["tensor"]} at {w["est"]:.2e} = {w["est"] / 1e-2:.0%} of the hard cap; {sum(t["est"] <= t["tight_bound"] for t in d["tensors"])} of {len(d["tensors"])} tensors inside the tight bound.', '']
    open(out, 'w').write('\n'.join(L) + '\n')
    print(out)


if __name__ == '__main__':
    main(*sys.argv[1:])
