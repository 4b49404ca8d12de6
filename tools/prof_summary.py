#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (or directory) as a per-kernel table (markdown).

    python tools/prof_summary.py gpurun_out/prof_r1a/bench_results.db [-o profiles/r01_kernel_stats.md] [--skip-first N]
"""
import argparse
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((?:[^()]|\([^()]*\))*\)\s*(?:\[clone .*\])?$', '', name)
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('-o', '--out')
    ap.add_argument('--title', default='')
    ap.add_argument('--by-position', default='', help='kernel-name substring: also list its dispatches by position within a step')
    ap.add_argument('--steps', type=int, default=0, help='number of identical steps in the trace (for --by-position)')
    a = ap.parse_args()
    path = a.db
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    db = sqlite3.connect(path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), '
                      'max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    lines = []
    if a.title:
        lines.append(f'# {a.title}\n')
    lines.append(f'source: `{a.db}`; total GPU kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n')
    lines.append('| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | LDS B | VGPR | AGPR | SGPR |')
    lines.append('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for n, c, tot, avg, mn, mx, g, wg, lds, v, ag, sg in rows:
        lines.append(f'| `{short(n)}` | {c} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} | {g} | {wg} | {lds} | {v} | {ag} | {sg} |')
    if a.by_position and a.steps:
        # one kernel name serves many layers: split its dispatches by their position inside a step (every step launches the same
        # sequence), so that a single layer's launch can be compared with the HIP-event timing bench.py reports for it
        d = db.execute('select duration from kernels where name like ? order by start', (f'%{a.by_position}%',)).fetchall()
        per = len(d) // a.steps
        lines.append(f'\n`{a.by_position}` by position within a step ({len(d)} dispatches = {a.steps} steps x {per}):\n')
        lines.append('| position | avg us | min us | max us |')
        lines.append('|---|---|---|---|')
        for i in range(per):
            v = [d[k * per + i][0] / 1e3 for k in range(a.steps)]
            lines.append(f'| {i} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |')
    text = '\n'.join(lines) + '\n'
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        open(a.out, 'w').write(text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main()
