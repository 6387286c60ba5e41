#!/bin/bash
# Memory-pipeline PMC passes (texture addresser, vector L1, L2, HBM) of the instanced-sphere lattice render, one rocprofv3 run per pass.
# usage: tools/pmc_lattice_mem.sh <out-subdir-under-gpurun_out> [n]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"
N="${2:-100}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$GRAFT_REPO_ROOT/tools/run_configs.py" "lattice:$N" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
run_pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES &&
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum &&
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" && grep -E "k_extend<35u, true" "$OUT/summary.txt"
rm -rf "$OUT"/ta "$OUT"/tcp "$OUT"/fetch "$OUT"/write
