#!/usr/bin/env python3
"""Timeline of the LAST training step in a rocprofv3 --kernel-trace database: every dispatch in start order with its queue, start offset and
duration -- to see what runs beside a resident foreign kernel (tools/probe_foreign_waves.py <us> <reserve> <blocks>).
    python tools/dp_timeline.py <dir-or-db> [marker-kernel-substring = conv_small_fwd]"""
import glob
import os
import re
import sqlite3
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'conv_small_fwd'
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
db = sqlite3.connect(path)
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute(f'select name, start, end, grid_x, workgroup_x{", " + qcol if qcol else ""} from kernels order by start').fetchall()
starts = [i for i, r in enumerate(rows) if marker in r[0]]
lo = starts[-1]
t0 = rows[lo][1]


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)[:60]


prev_end = t0
for r in rows[lo:]:
    gap = (r[1] - prev_end) / 1e3
    print(f'{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:8.1f} us  gap {gap:7.1f}  q={r[5] if qcol else "?"}  grid {r[3] // max(r[4], 1):6d}  {short(r[0])}')
    if (qcol is None) or True:
        prev_end = max(prev_end, r[2])
