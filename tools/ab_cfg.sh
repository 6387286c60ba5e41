#!/bin/bash
# A/B helper: runs tools/run_configs.py once per library variant.  usage: tools/ab_cfg.sh "<configs>" lib1.so lib2.so ...
CFG="$1"; shift
for lib in "$@"; do
  echo -n "$(basename $lib): "
  CAUSTIC_HIP_LIB=$PWD/$lib timeout -k 10 280 python tools/run_configs.py $CFG 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%s: %.1f Mpaths/s frame %.3f s trav %.1f shade %.1f' % (d['config'][:10], d['path_samples_per_s']/1e6, d['frame_s'], d['traversal_ms'], d['bin_shade_ms']), end=' | ')
print()"
done
