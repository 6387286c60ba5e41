#!/bin/bash
# Everything profiles/$ROUND holds, collected in one call on the GPU box:  tools/collect_round.sh <git-hash> [round, default r03]
#   1. the PMC passes behind bench.py's roofline objects (tools/collect_roofline_profile.sh),
#   2. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations),
#   3. the bench line itself (default flags), the other configurations and a sweep over every scene of scenes.py,
#   4. the co-run experiment (tools/micro/corun.py).
# Results land in gpurun_out/$ROUND/, to be copied into profiles/$ROUND/.
set -u
GIT="${1:-unknown}"
ROUND="${2:-r03}"
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/$ROUND"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 "$ROOT/tools/collect_roofline_profile.sh" "$GIT" "$ROUND" > "$OUT/collect_roofline.log" 2>&1 || { echo "roofline passes failed"; tail -5 "$OUT/collect_roofline.log"; exit 1; }
cp "$ROOT/gpurun_out/roofline_profile/roofline_profile.json" "$ROOT/gpurun_out/roofline_profile/pmc_s2_summary.txt" "$ROOT/gpurun_out/roofline_profile/pmc_s5_summary.txt" "$ROOT/gpurun_out/roofline_profile/valu_issue_cost.txt" "$OUT/"
mkdir -p "$ROOT/profiles/$ROUND" && cp "$OUT/roofline_profile.json" "$ROOT/profiles/$ROUND/"   # (the box's copy: what the bench runs below read)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-s5 --no-rank-share > "$OUT/bench_under_kernel_trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/bench_under_kernel_trace.log"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
rm -rf "$OUT/kt"
cd "$ROOT"
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
timeout -k 10 400 python3 tools/run_configs.py rtiow final cornell lattice:100 > "$OUT/configs.jsonl" 2> "$OUT/configs.err" || { echo "configs failed"; tail -5 "$OUT/configs.err"; exit 1; }
timeout -k 10 300 python3 tools/run_configs.py scad >> "$OUT/configs.jsonl" 2>> "$OUT/configs.err" || { echo "scad config failed"; exit 1; }
timeout -k 10 400 python3 tools/run_all_scenes.py 800 256 > "$OUT/all_scenes.jsonl" 2> "$OUT/all_scenes.err" || { echo "scene sweep failed"; tail -5 "$OUT/all_scenes.err"; exit 1; }
# what the traversal loses and a streaming kernel gets when they share the device (and the copy rate the shade kernels are measured against)
timeout -k 10 200 python3 tools/micro/corun.py 256 > "$OUT/corun_traversal_beside_copy.txt" 2> "$OUT/corun.err" || { echo "corun failed"; tail -3 "$OUT/corun.err"; }
cut -c1-600 "$OUT/bench.json"
