#!/bin/bash
# per-layer forward / data-gradient / weight-gradient times (tools/layer_table.py) under the plain and the block-structured brick order, three alternating runs, medians
for i in 1 2 3; do
for v in plain block; do
  unset E3_WINO_BLOCK; [ $v = plain ] && export E3_WINO_BLOCK=0
  python tools/layer_table.py 6 2>/dev/null > /tmp/lt_${v}_$i.md
done; done
python - <<'PY'
import re
def load(p):
    rows = {}
    for l in open(p):
        c = [x.strip() for x in l.split('|')]
        if len(c) > 14 and ('conv' in c[1]):
            f = lambda s: float(s) if re.match(r'^[0-9.]+$', s) else None
            rows[c[1]] = (f(c[7]), f(c[11]), f(c[14]))
    return rows
import statistics
P = [load(f'/tmp/lt_plain_{i}.md') for i in (1, 2, 3)]; B = [load(f'/tmp/lt_block_{i}.md') for i in (1, 2, 3)]
tot = [0, 0]
for k in P[0]:
    out = [k]
    for j, name in enumerate(('fwd', 'dgrad', 'wgrad')):
        if P[0][k][j] is None: continue
        p = statistics.median(x[k][j] for x in P); b = statistics.median(x[k][j] for x in B)
        out.append(f'{name} {p:.0f} -> {b:.0f}')
        if j < 2: tot[0] += p; tot[1] += b
    print('  '.join(out))
print('sum fwd + dgrad: plain %.0f block %.0f us' % tuple(tot))
PY
