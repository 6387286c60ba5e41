# usage: tools/ab_streams_cfg.sh <run_configs args>: the configuration with 1, 2 and 3 batches in flight
for s in 1 2 3; do
  echo -n "streams=$s: "
  CHIP_STREAMS=$s timeout -k 10 280 python tools/run_configs.py "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%s: %.1f Mpaths/s frame %.3f s trav %.1f shade %.1f' % (d['config'][:12], d['path_samples_per_s']/1e6, d['frame_s'], d['traversal_ms'], d['bin_shade_ms']), end=' | ')
print()"
done
