#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE; one rocprofv3 run per counter group) of the instanced-sphere lattice render.
# usage: tools/pmc_lattice.sh <out-subdir-under-gpurun_out> [n]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
N="${2:-100}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$GRAFT_REPO_ROOT/tools/run_configs.py" "lattice:$N" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" && grep -E "k_extend|k_shade" "$OUT/summary.txt"
