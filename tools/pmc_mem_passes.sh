#!/bin/bash
# Memory-pipeline PMC passes (texture addresser, vector L1, L2) for a short bench.py run, one rocprofv3 run per pass.
# usage: tools/pmc_mem_passes.sh <out-subdir-under-gpurun_out>
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
BENCH_ARGS="${BENCH_ARGS:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline}"
run_pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES &&
run_pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum &&
run_pass tcc TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_ATOMIC_sum &&
run_pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum &&
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" && grep -E "k_extend<0|k_shade_tiles" "$OUT/summary.txt"
