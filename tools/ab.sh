#!/bin/bash
# A/B helper: runs bench.py once per library variant given on the command line and prints a one-line digest.
# usage: tools/ab.sh "<bench args>" lib1.so lib2.so ...
ARGS="$1"; shift
for lib in "$@"; do
  echo -n "$(basename $lib): "
  CAUSTIC_HIP_LIB=$PWD/$lib timeout -k 10 180 python bench.py $ARGS --no-cpu-baseline --no-s5 --no-rank-share 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['device_ms_per_step']
print('%.1f Mpaths/s  wall %.1f ms/step | device %.1f  trav %.1f  bin+shade %.1f | extend avg %.3f ms' % (d['value']/1e6, d['ms_per_step'], m['render_call'], m['traversal_summed_over_streams'], m['bin_and_shade_summed_over_streams'], r['timed_region_two_streams']['avg_launch_ms']), '| 1-stream frame %.1f trav %.1f shade %.1f' % (r['exclusive']['frame_ms'], r['exclusive']['traversal_ms'], r['exclusive']['bin_and_shade_ms']))"
done
