#!/bin/bash
# One rocprofv3 --pmc pass of the LDS counters over one headline frame; prints the per-kernel sums.  Run on the GPU box.
#   tools/pmc_lds.sh [out-name]
set -u
ROOT="$GRAFT_REPO_ROOT"
NAME="${1:-lds}"
OUT="$ROOT/gpurun_out/r03/pmc_$NAME"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/raw" -o p -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-s5 --no-rank-share --no-exclusive > "$OUT/run.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/run.log"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT/raw" > "$OUT/../pmc_${NAME}_summary.txt"
rm -rf "$OUT/raw"
cat "$OUT/../pmc_${NAME}_summary.txt"
