#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection + kernel_trace) per kernel.

    python tools/pmc_summary.py gpurun_out/pmcA [gpurun_out/pmcB ...] -o profiles/r01_pmc_conv.md

Derived columns (MI355X_MICROARCH.md): clock = GRBM_GUI_ACTIVE/8 per XCD / duration; MFMA pipe utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); waves/SIMD = 4*SQ_WAVE_CYCLES / (cycles*1024) (SQ_* count quad-cycles);
HBM bytes: FETCH_SIZE [KiB] * 1024 * 2 (gfx950: FETCH_SIZE reports half of a wide coalesced read) , WRITE_SIZE [KiB] * 1024.
"""
import argparse
import collections
import csv
import glob
import os
import re
import statistics


def short(n):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    return re.sub(r'\(.*', '', n)[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dirs', nargs='+')
    ap.add_argument('-o', '--out')
    ap.add_argument('--filter', default='conv3_wino|wgrad_wino|conv3_v3|conv_mfma|wgrad_conv|wgrad_point|conv_small|conv_final|bn_')
    a = ap.parse_args()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.dirs:
        cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
        kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
        if not cc or not kt:
            continue
        ktr = {r['Dispatch_Id']: r for r in csv.DictReader(open(kt[0]))}
        agg = collections.defaultdict(dict)
        grids = {}
        for r in csv.DictReader(open(cc[0])):
            agg[r['Dispatch_Id']][r['Counter_Name']] = agg[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            grids[r['Dispatch_Id']] = int(r.get('Grid_Size', 0)) // max(int(r.get('Workgroup_Size', 1)), 1)
        for disp, c in agg.items():
            k = ktr[disp]
            name = short(k['Kernel_Name'])
            if not re.search(a.filter, name):
                continue
            key = (name, grids.get(disp, 0))
            dur = int(k['End_Timestamp']) - int(k['Start_Timestamp'])
            per[key]['dur_us'].append(dur / 1e3)
            for cn, v in c.items():
                per[key][cn].append(v)
    lines = ['| kernel | workgroups | n | dur us | clock GHz | MFMA util | waves/SIMD | wait_any | wait_inst | HBM read MB | HBM write MB |', '|---|---|---|---|---|---|---|---|---|---|---|']
    for (name, grid), c in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]['dur_us'])):
        med = lambda k: statistics.median(c[k]) if k in c and c[k] else None
        dur = med('dur_us')
        cyc = med('GRBM_GUI_ACTIVE') / 8 if med('GRBM_GUI_ACTIVE') else None
        f = lambda x, p=2: '' if x is None else f'{x:.{p}f}'
        clock = cyc / (dur * 1e3) if cyc else None
        util = med('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024) if cyc and med('SQ_VALU_MFMA_BUSY_CYCLES') is not None else None
        wps = 4 * med('SQ_WAVE_CYCLES') / (cyc * 1024) if cyc and med('SQ_WAVE_CYCLES') else None
        wa = med('SQ_WAIT_ANY') / med('SQ_WAVE_CYCLES') if med('SQ_WAVE_CYCLES') and med('SQ_WAIT_ANY') is not None else None
        wi = med('SQ_WAIT_INST_ANY') / med('SQ_WAVE_CYCLES') if med('SQ_WAVE_CYCLES') and med('SQ_WAIT_INST_ANY') is not None else None
        rd = med('FETCH_SIZE') * 1024 * 2 / 1e6 if med('FETCH_SIZE') is not None else None
        wr = med('WRITE_SIZE') * 1024 / 1e6 if med('WRITE_SIZE') is not None else None
        lines.append(f'| `{name}` | {grid} | {len(c["dur_us"])} | {f(dur, 1)} | {f(clock)} | {f(util, 3)} | {f(wps)} | {f(wa)} | {f(wi)} | {f(rd, 1)} | {f(wr, 1)} |')
    # columns without a single value (their counters were not in any of the passes) are dropped instead of printed empty
    rows = [[c.strip() for c in l.strip('|').split('|')] for l in lines]
    keep = [j for j in range(len(rows[0])) if j < 4 or any(r[j] for r in rows[2:])]
    lines = ['| ' + ' | '.join(r[j] for j in keep) + ' |' for r in rows]
    text = '\n'.join(lines) + '\n'
    if a.out:
        hdr = ('# PMC counters (rocprofv3 --pmc, separate passes for SQ / FETCH_SIZE / WRITE_SIZE)\n\n'
               'FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; WRITE_SIZE is uncalibrated.\n'
               'Durations under the profiler are a few % longer than unprofiled ones.\n\n')
        open(a.out, 'w').write(hdr + text)
    print(text)


if __name__ == '__main__':
    main()
