#!/bin/bash
# Memory-pipeline counter passes over one configuration of tools/run_configs.py, summarised per kernel.
# Run on the GPU box:  tools/pmc_config_mem.sh <config> <outname>
set -u
CFG="$1"; NAME="$2"
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/$NAME"
rm -rf "$OUT"; mkdir -p "$OUT/p"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/p/$name" -o p -- python3 "$ROOT/tools/run_configs.py" $CFG > "$OUT/p/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/p/$name.log"; return 1; }
}
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum &&
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM &&
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
python3 "$ROOT/tools/pmc_summary.py" "$OUT/p" > "$OUT/summary.txt"
rm -rf "$OUT/p/ta" "$OUT/p/tcp" "$OUT/p/lds" "$OUT/p/fetch" "$OUT/p/write"
grep "k_shade" "$OUT/summary.txt" | cut -c1-150
