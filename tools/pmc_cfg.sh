#!/bin/bash
# One rocprofv3 --pmc pass (the SQ group of tools/collect_roofline_profile.sh) over one configuration of tools/run_configs.py;
# prints the per-kernel sums.  Run on the GPU box:  tools/pmc_cfg.sh <config> [out-name]
set -u
ROOT="$GRAFT_REPO_ROOT"
CFG="${1:-cornell}"
NAME="${2:-$CFG}"
OUT="$ROOT/gpurun_out/r03/pmc_$NAME"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/raw" -o p -- python3 "$ROOT/tools/run_configs.py" "$CFG" > "$OUT/run.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/raw" > "$OUT/../pmc_${NAME}_summary.txt"
rm -rf "$OUT/raw"
grep -v " sum=0 " "$OUT/../pmc_${NAME}_summary.txt"
tail -2 "$OUT/run.log"
