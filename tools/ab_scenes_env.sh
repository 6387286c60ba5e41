#!/bin/bash
# A/B helper over environment settings for tools/run_all_scenes.py.   usage: tools/ab_scenes_env.sh REPS "<scene names>" "VAR=value ..." ...
REPS="$1"; SC="$2"; shift 2
for rep in $(seq 1 "$REPS"); do
  for setting in "$@"; do
    echo -n "[$setting] "
    env $setting timeout -k 10 280 python tools/run_all_scenes.py 800 256 $SC 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%s: %.0f M trav %.1f shade %.1f' % (d['scene'][:14], d['Mpaths_per_s'], d['trav_ms'], d['shade_ms']), end=' | ')
print()" || exit 1
  done
done
