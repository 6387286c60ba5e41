#!/bin/bash
# A/B helper for the other configurations: runs tools/run_configs.py once per library variant and prints a digest.
# usage: tools/ab_configs.sh "<run_configs args>" lib1.so lib2.so ...
ARGS="$1"; shift
for lib in "$@"; do
  echo "== $(basename $lib)"
  CAUSTIC_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/run_configs.py $ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    c=json.loads(l)
    print('  %-44s %7.1f Mpaths/s  visits %5.2f prim %4.2f  trav %6.1f ms  shade %6.1f ms  frame %.3f s' % (c['config'], c['path_samples_per_s']/1e6, c['bvh_visits_per_segment'], c['prim_tests_per_segment'], c['traversal_ms'], c['bin_shade_ms'], c['frame_s']))"
done
