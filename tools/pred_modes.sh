#!/bin/bash
# bench.py with the Predictor leg on the full cfg-5 volume, several processes in a row: throughput, rows of tiles, host time in front of the first tile (profiles/r06_predictor_modes.md)
for i in 1 2 3 4 5 6; do
  python bench.py --no-cpu-baseline --no-extra-legs --no-live-traffic --predictor-volume full --steps 5 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['predictor']; t=p['timing']
print('run $i: Predictor %.1f MVox/s (%.3f s) rows %.3f s' % (p['value'], p['seconds'], t['rows_s']), {k: t[k] for k in ('to_uploads_submitted_s','alloc_output_s','fill_output_s','first_piece_wait_s','to_first_tile_s','issue_s','wall_s','upload_worker_s','download_worker_s','slowest_rows') if k in t})"
done
