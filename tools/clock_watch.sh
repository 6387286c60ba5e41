#!/bin/bash
# Samples the GPU's shader clock and power while a command runs (is the chip power-limited under this load?).
# usage: tools/clock_watch.sh <outfile> -- cmd...
OUT="$1"; shift; shift
"$@" &
PID=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" >> "$OUT"
  sleep 0.7
  kill -0 $PID 2>/dev/null || break
done
wait $PID
