#!/bin/bash
# A/B helper over environment settings: runs bench.py REPS times per setting (each in a process of its own) and
# prints a one-line digest per run.   usage: tools/ab_env.sh REPS "<bench args>" "VAR=value ..." "VAR=value ..." ...
# (an empty string stands for the default environment)
REPS="$1"; ARGS="$2"; shift 2
for rep in $(seq 1 "$REPS"); do
  for setting in "$@"; do
    echo -n "[$setting] "
    env $setting timeout -k 10 180 python bench.py $ARGS --no-cpu-baseline --no-s5 --no-rank-share 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['device_ms_per_step']
print('%.1f Mpaths/s  wall %.1f ms/step | trav %.1f  bin+shade %.1f' % (d['value']/1e6, d['ms_per_step'], m['traversal_summed_over_streams'], m['bin_and_shade_summed_over_streams']))" || exit 1
  done
done
