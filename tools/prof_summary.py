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
    text = '\n'.join(lines) + '\n'
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        open(a.out, 'w').write(text)
    sys.stdout.write(text)


if __name__ == '__main__':
    main()
