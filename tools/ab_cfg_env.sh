#!/bin/bash
# A/B helper over environment settings for tools/run_configs.py.   usage: tools/ab_cfg_env.sh REPS "<configs>" "VAR=value ..." ...
# (an empty string stands for the default environment; CAUSTIC_HIP_LIB=path selects another build of the library)
REPS="$1"; CFG="$2"; shift 2
for rep in $(seq 1 "$REPS"); do
  for setting in "$@"; do
    echo -n "[$setting] "
    env $setting timeout -k 10 280 python tools/run_configs.py $CFG 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%s: %.1f Mpaths/s frame %.3f s trav %.1f shade %.1f' % (d['config'][:10], d['path_samples_per_s']/1e6, d['frame_s'], d['traversal_ms'], d['bin_shade_ms']), end=' | ')
print()" || exit 1
  done
done
