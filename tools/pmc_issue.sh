#!/bin/bash
# Instruction-issue counters (instruction mix by pipe, fetch, busy cycles) over one configuration of
# tools/run_configs.py, one rocprofv3 run per pass, summarised per kernel.
# Run on the GPU box:  tools/pmc_issue.sh <config> <outname>
set -u
CFG="$1"; NAME="$2"
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/$NAME"
rm -rf "$OUT"; mkdir -p "$OUT/p"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout -s KILL 300 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$OUT/p/$name" -o p -- python3 "$@" > "$OUT/p/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/p/$name.log"; return 1; }
}
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_CYCLES SQ_BUSY_CU_CYCLES -- $ROOT/tools/run_configs.py $CFG &&
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_IFETCH_LEVEL -- $ROOT/tools/run_configs.py $CFG &&
pass c SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 -- $ROOT/tools/run_configs.py $CFG &&
pass d SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU -- $ROOT/tools/run_configs.py $CFG &&
pass e GRBM_GUI_ACTIVE -- $ROOT/tools/run_configs.py $CFG &&
python3 "$ROOT/tools/pmc_summary.py" "$OUT/p" > "$OUT/summary.txt" && grep -E "k_extend<0u, true|k_shade" "$OUT/summary.txt"
rm -rf "$OUT/p/a" "$OUT/p/b" "$OUT/p/c" "$OUT/p/d" "$OUT/p/e"
