#!/bin/bash
# Everything profiles/r02 holds, collected in one call on the GPU box:  tools/collect_round.sh <git-hash>
#   1. the PMC passes behind bench.py's roofline objects (tools/collect_roofline_profile.sh),
#   2. rocprofv3 --kernel-trace --stats of the bench command (per-kernel durations),
#   3. the bench line itself (default flags), the other configurations and a sweep over every scene of scenes.py.
# Results land in gpurun_out/r02/, to be copied into profiles/r02/.
set -u
GIT="${1:-unknown}"
ROOT="$GRAFT_REPO_ROOT"
OUT="$ROOT/gpurun_out/r02"
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 "$ROOT/tools/collect_roofline_profile.sh" "$GIT" > "$OUT/collect_roofline.log" 2>&1 || { echo "roofline passes failed"; tail -5 "$OUT/collect_roofline.log"; exit 1; }
cp "$ROOT/gpurun_out/roofline_profile/roofline_profile.json" "$ROOT/gpurun_out/roofline_profile/pmc_s2_summary.txt" "$ROOT/gpurun_out/roofline_profile/pmc_s5_summary.txt" "$ROOT/gpurun_out/roofline_profile/valu_issue_cost.txt" "$OUT/"
mkdir -p "$ROOT/profiles/r02" && cp "$OUT/roofline_profile.json" "$ROOT/profiles/r02/"   # (the box's copy: what the bench runs below read)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-s5 --no-rank-share > "$OUT/bench_under_kernel_trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$OUT/bench_under_kernel_trace.log"; exit 1; }
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
rm -rf "$OUT/kt"
cd "$ROOT"
timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
timeout -k 10 400 python3 tools/run_configs.py rtiow final cornell lattice:100 > "$OUT/configs.jsonl" 2> "$OUT/configs.err" || { echo "configs failed"; tail -5 "$OUT/configs.err"; exit 1; }
timeout -k 10 300 python3 tools/run_configs.py scad >> "$OUT/configs.jsonl" 2>> "$OUT/configs.err" || { echo "scad config failed"; exit 1; }
timeout -k 10 400 python3 tools/run_all_scenes.py 800 256 > "$OUT/all_scenes.jsonl" 2> "$OUT/all_scenes.err" || { echo "scene sweep failed"; tail -5 "$OUT/all_scenes.err"; exit 1; }
cut -c1-600 "$OUT/bench.json"
