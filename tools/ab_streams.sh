for s in 1 2 3 1 2; do
  echo -n "streams=$s: "
  CHIP_STREAMS=$s timeout -k 10 180 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-s5 --no-rank-share 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f Mpaths/s  wall %.1f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done
