for i in 1 2 3; do
  for which in r05 r06; do
    if [ $which = r05 ]; then export E3_LIB_PATH=$PWD/tools/_bin/libe3unet_r05.so; else unset E3_LIB_PATH; fi
    p=$(python bench.py --no-cpu-baseline --no-extra-legs --predictor-volume full --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['predictor']; t=d['timing']
print('%.1f MVox/s (%.3f s; compute stream %.3f s, issue %.3f s)' % (d['value'], d['seconds'], t['compute_stream_s'], t['issue_s']), d.get('needed_region_ab'))")
    echo "round $i $which: Predictor 512x2048x2048 $p"
  done
done
