#!/bin/bash
# Collects the PMC passes behind bench.py's roofline objects (one rocprofv3 run per pass; slot limits: 8 SQ, 4 TCC)
# and writes profiles/$ROUND/roofline_profile.json + the per-kernel counter summaries.  Run on the GPU box:
#   tools/collect_roofline_profile.sh <git-hash> [round]
set -u
GIT="${1:-unknown}"
ROUND="${2:-r03}"
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/roofline_profile"
rm -rf "$OUT"; mkdir -p "$OUT/s2" "$OUT/s5"
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <subdir> <name> <counters...> -- <program args...>
  sub=$1; name=$2; shift 2
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout -s KILL 300 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$OUT/$sub/$name" -o p -- python3 "$@" > "$OUT/$sub/$name.log" 2>&1 || { echo "pass $sub/$name failed"; tail -3 "$OUT/$sub/$name.log"; return 1; }
}
S2="$ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-s5 --no-rank-share --no-exclusive"
S5="$ROOT/tools/run_configs.py lattice:100"
pass s2 sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $S2 &&
pass s2 fetch FETCH_SIZE GRBM_GUI_ACTIVE -- $S2 &&
pass s2 write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $S2 &&
pass s2 mix SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT -- $S2 &&
pass s5 sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -- $S5 &&
pass s5 fetch FETCH_SIZE GRBM_GUI_ACTIVE -- $S5 &&
pass s5 write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $S5 &&
pass s5 mix SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT -- $S5 &&
{ timeout -k 5 200 "$ROOT/tools/micro/valu_clock" > "$OUT/valu_issue_cost.txt" 2> "$OUT/valu_issue_cost.err" || { echo "valu_clock failed"; exit 1; }; } &&
python3 "$ROOT/tools/pmc_summary.py" "$OUT/s2" > "$OUT/pmc_s2_summary.txt" &&
python3 "$ROOT/tools/pmc_summary.py" "$OUT/s5" > "$OUT/pmc_s5_summary.txt" &&
python3 "$ROOT/tools/make_roofline_profile.py" "$OUT" "$GIT" > "$OUT/roofline_profile.json.tmp" && mv "$OUT/roofline_profile.json.tmp" "$OUT/roofline_profile.json" && cat "$OUT/roofline_profile.json"
RC=$?
rm -rf "$OUT"/s2/sq "$OUT"/s2/fetch "$OUT"/s2/write "$OUT"/s2/mix "$OUT"/s5/sq "$OUT"/s5/fetch "$OUT"/s5/write "$OUT"/s5/mix
exit $RC
