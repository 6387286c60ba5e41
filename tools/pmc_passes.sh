#!/bin/bash
# Collects the PMC passes for a short bench.py run, one rocprofv3 run per pass (slot limits: 8 SQ, 4 TCC).
# usage: tools/pmc_passes.sh <out-subdir-under-gpurun_out> [bench args...]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
BENCH_ARGS="${BENCH_ARGS:---spp 16 --steps 1 --warmup 0 --no-cpu-baseline}"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run_pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 &&
run_pass fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.txt" && grep -E "k_extend|k_shade<0" "$OUT/summary.txt"
