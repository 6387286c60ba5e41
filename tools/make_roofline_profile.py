#!/usr/bin/env python3
"""Turns the PMC summaries collected by tools/collect_roofline_profile.sh into profiles/r02/roofline_profile.json, the
per-ray-segment figures bench.py multiplies with its live segment counts and launch durations.
usage: make_roofline_profile.py <dir with pmc_s2_summary.txt, pmc_s5_summary.txt, s2/*.log, s5/*.log> <git hash>"""
import datetime
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    for f in ("kernels.hpp", "device_scene.h"):
        with open(os.path.join(ROOT, "rust-raytracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def counters(path, kernel_prefix):
    """Sums over every instantiation whose name starts with kernel_prefix: {counter: (sum, dispatches)}."""
    acc = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\S+)\s+sum=(\S+) dispatches=(\d+)", line)
        if not m or not m.group(1).startswith(kernel_prefix):
            continue
        s, n = acc.get(m.group(2), (0.0, 0))
        acc[m.group(2)] = (s + float(m.group(3)), n + int(m.group(4)))
    return acc


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def figures(c, segments):
    gui = c["GRBM_GUI_ACTIVE"][0]      # summed over the 8 XCDs and the dispatches
    simd_slots = gui / 8 / 4 * 1024     # issue slots (4 cycles each) of the chip's 1024 SIMDs while the kernel ran
    lane_ops = c["SQ_THREAD_CYCLES_VALU"][0]
    active = c["SQ_ACTIVE_INST_VALU"][0]
    return {
        "segments": segments,
        "launches": c["SQ_THREAD_CYCLES_VALU"][1],
        "valu_lane_ops_per_segment": lane_ops / segments,
        "valu_instructions_per_segment_per_lane": c["SQ_INSTS_VALU"][0] * 64 / segments,
        "lane_utilisation": lane_ops / (64.0 * active),
        "valu_busy": active / simd_slots,
        "frac_of_valu_peak_alone": lane_ops / (64.0 * simd_slots),
        "wave_wait_share": c["SQ_WAIT_ANY"][0] / c["SQ_WAVE_CYCLES"][0],
        "avg_launch_cycles_alone": gui / 8 / c["GRBM_GUI_ACTIVE"][1],
        "hbm_bytes_per_segment": (2 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024 / segments,
        "fetch_kb_as_reported": c["FETCH_SIZE"][0],
        "write_kb": c["WRITE_SIZE"][0],
        "l2_hit_rate": c["TCC_HIT_sum"][0] / max(1.0, c["TCC_HIT_sum"][0] + c["TCC_MISS_sum"][0]),
    }


def main():
    d, git = sys.argv[1], sys.argv[2]
    b = last_json(os.path.join(d, "s2", "sq.log"))
    seg2 = b["roofline"]["segments_per_path"] * b["config"]["path_samples_per_step"] * b["steps"]
    s2 = figures(counters(os.path.join(d, "pmc_s2_summary.txt"), "k_extend<0u, true, true"), seg2)
    s2["workload"] = b["config"]["workload"] + " (one frame under rocprofv3 --pmc, kernels serialised)"
    l = last_json(os.path.join(d, "s5", "sq.log"))
    seg5 = 2 * l["segments_per_path"] * 1920 * 1080 * l["spp"]  # run_configs renders the frame twice
    s5 = figures(counters(os.path.join(d, "pmc_s5_summary.txt"), "k_extend<35u, true, true"), seg5)
    s5["workload"] = l["config"] + f", {l['spp']} spp, two renders under rocprofv3 --pmc"
    out = {
        "git": git,
        "kernel_source_sha256_16": source_hash(),
        "collected": datetime.date.today().isoformat(),
        "how": "tools/collect_roofline_profile.sh: separate rocprofv3 --pmc runs for the SQ, FETCH_SIZE and WRITE_SIZE groups; "
               "counters summed over the traversal kernel's instantiations (bounce 0 and later bounces); FETCH_SIZE doubled "
               "(gfx950 reports half of a wide read stream); SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU count 4-cycle issue slots",
        "s2": s2,
        "s5": s5,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
