#!/bin/bash
# One SQ counter pass + one LDS/instruction-mix pass over one configuration of tools/run_configs.py, summarised per
# kernel.  Run on the GPU box:  tools/pmc_config.sh <config> <outname>
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
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $ROOT/tools/run_configs.py $CFG &&
pass mix SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA -- $ROOT/tools/run_configs.py $CFG &&
python3 "$ROOT/tools/pmc_summary.py" "$OUT/p" > "$OUT/summary.txt" && cat "$OUT/summary.txt"
rm -rf "$OUT/p/sq" "$OUT/p/mix"
