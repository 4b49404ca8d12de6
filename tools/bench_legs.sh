#!/bin/bash
# which leg of the default bench.py run the Predictor leg reacts to: the default run with one leg removed at a time (profiles/r06_predictor_modes.md)
show() { python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); p=d['predictor']; t=p['timing']
print('$2: step %.3f ms; Predictor %.1f MVox/s rows %.3f s power %.0f W' % (d['ms_per_step'], p['value'], t['rows_s'], p['sensors']['socket_power_w_min_mean_max'][1]))"; }
python bench.py --no-live-traffic > /tmp/a.json 2>/dev/null; show /tmp/a.json no-live-traffic
python bench.py --no-cpu-baseline > /tmp/b.json 2>/dev/null; show /tmp/b.json no-cpu-baseline
python bench.py --no-extra-legs > /tmp/c.json 2>/dev/null; show /tmp/c.json no-extra-legs
python bench.py > /tmp/d.json 2>/dev/null; show /tmp/d.json default
python bench.py --no-live-traffic --no-cpu-baseline > /tmp/e.json 2>/dev/null; show /tmp/e.json no-live-no-cpu
