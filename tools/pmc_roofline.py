#!/usr/bin/env python3
"""HBM traffic per launch of bench.py's roofline kernel from rocprofv3 PMC passes -> profiles/r02_pmc_roofline.json (read by bench.py).

    python tools/pmc_roofline.py --dtype bf16 --kernel conv_b16_kernel --fetch DIR --write DIR [--busy DIR] --steps 3 -o profiles/r02_pmc_roofline.json

One pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  The dispatches of the
kernel are grouped by their position inside a step (every step launches the same sequence); the roofline layer (forward of
up_convs.2.conv1, 64->32 at full resolution) is the position with the largest read volume.  Bytes: FETCH_SIZE [KiB] x 1024 x 2 (gfx950:
FETCH_SIZE reports half of a wide coalesced read) + WRITE_SIZE [KiB] x 1024.
"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics


def per_position(d, kernel, counter, steps):
    cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(cc)):
        if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
            rows[int(r['Dispatch_Id'])] = rows.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
    vals = [rows[k] for k in sorted(rows)]
    per = len(vals) // steps
    assert per * steps == len(vals), (len(vals), steps)
    return [statistics.median(vals[s * per + i] for s in range(steps)) for i in range(per)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', required=True)
    ap.add_argument('--layer', default='up_convs.2.conv1')
    ap.add_argument('--kernel', required=True)
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--busy')
    ap.add_argument('--steps', type=int, required=True, help='steps in the profiled run incl. warm-up and the extra dgrad/wgrad timing steps')
    ap.add_argument('-o', '--out', required=True)
    ap.add_argument('--command', default='')
    a = ap.parse_args()
    rd = per_position(a.fetch, a.kernel, 'FETCH_SIZE', a.steps)
    wr = per_position(a.write, a.kernel, 'WRITE_SIZE', a.steps)
    pos = max(range(len(rd)), key=lambda i: rd[i])
    ent = {'kernel': a.kernel, 'position_in_step': pos, 'launches_per_step': len(rd),
           'fetch_bytes': rd[pos] * 1024 * 2, 'write_bytes': wr[pos] * 1024,
           'hbm_bytes_per_launch': rd[pos] * 1024 * 2 + wr[pos] * 1024,
           'all_positions_MB': [[round(r * 2048 / 1e6, 1), round(w * 1024 / 1e6, 1)] for r, w in zip(rd, wr)],
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes with --kernel-trace; FETCH_SIZE doubled (gfx950 wide-read correction); median over steps',
           'command': a.command}
    if a.busy:
        try:
            ent['mfma_busy_cycles'] = per_position(a.busy, a.kernel, 'SQ_VALU_MFMA_BUSY_CYCLES', a.steps)[pos]
        except Exception as e:  # noqa: BLE001
            ent['mfma_busy_cycles'] = f'n/a ({e})'
    data = json.load(open(a.out)) if os.path.exists(a.out) else {}
    data.setdefault(a.dtype, {})[a.layer] = ent
    os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
    json.dump(data, open(a.out, 'w'), indent=1)
    print(json.dumps(ent, indent=1))


if __name__ == '__main__':
    main()
