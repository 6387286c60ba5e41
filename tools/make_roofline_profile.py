#!/usr/bin/env python3
"""Turns the PMC summaries collected by tools/collect_roofline_profile.sh into profiles/<round>/roofline_profile.json, the
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


def counters(path, kernel_prefixes):
    """Sums over every instantiation whose name starts with one of kernel_prefixes: {counter: (sum, dispatches)}."""
    acc = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\S+)\s+sum=(\S+) dispatches=(\d+)", line)
        if not m or not m.group(1).startswith(tuple(kernel_prefixes)):
            continue
        s, n = acc.get(m.group(2), (0.0, 0))
        acc[m.group(2)] = (s + float(m.group(3)), n + int(m.group(4)))
    return acc


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def issue_costs(path):
    """Cycles per wave64 instruction per SIMD at 4 waves per SIMD, by instruction kind, from tools/micro/valu_clock."""
    cost = {}
    for line in open(path):
        m = re.match(r"(v_\S+)\s.*4 wave/SIMD:\s+([0-9.]+) cyc/instr", line)
        if m:
            cost[m.group(1)] = float(m.group(2))
    return cost


def figures(c, segments, cost):
    gui = c["GRBM_GUI_ACTIVE"][0]      # summed over the 8 XCDs and the dispatches
    simd_cycles = gui / 8 * 1024        # cycles of the chip's 1024 SIMDs while the kernel ran
    simd_slots = simd_cycles / 4        # 4-cycle issue slots (the unit of the SQ_*_VALU counters)
    lane_ops = c["SQ_THREAD_CYCLES_VALU"][0]
    active = c["SQ_ACTIVE_INST_VALU"][0]
    insts = c["SQ_INSTS_VALU"][0]
    # Issue cost of the kernel's own instruction mix: plain f32 multiply-adds, adds and multiplies issue at the fast rate
    # the microbenchmark measures for v_fma_f32; everything else (f64, min / max / max3, selects, compares,
    # conversions, 3-operand integer forms) at the slow one.  32-bit integer adds and logic also issue fast but the
    # counters do not separate them from the slow integer forms, so they are priced as slow: the mix cost is an upper
    # bound and the occupancy figure below a lower bound.
    fast = c["SQ_INSTS_VALU_FMA_F32"][0] + c["SQ_INSTS_VALU_ADD_F32"][0] + c["SQ_INSTS_VALU_MUL_F32"][0]
    # (the mix pass is a run of its own; its total is not collected, so shares are taken against the first run's count)
    fast_share = min(1.0, fast / insts)
    c_fast = cost["v_fma_f32"]
    c_slow = sum(cost[k] for k in ("v_fma_f64", "v_max3_f32", "v_min_u32", "v_cndmask_b32")) / 4.0
    mix_cost = fast_share * c_fast + (1.0 - fast_share) * c_slow
    cycles_per_inst = simd_cycles / insts
    return {
        "valu_issue": {
            "instructions": insts,
            "share_f32_fma_add_mul": fast_share,
            "measured_cost_cycles": {"fast (v_fma_f32)": c_fast, "slow (mean of v_fma_f64, v_max3_f32, v_min_u32, v_cndmask_b32)": c_slow},
            "mix_cost_cycles_per_instruction": mix_cost,
            "simd_cycles_per_instruction_alone": cycles_per_inst,
            "issue_occupancy_alone": mix_cost / cycles_per_inst,
        },
        "frac_of_nominal_valu_peak_alone": lane_ops / (64.0 * simd_cycles / 2.0),
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
    cost = issue_costs(os.path.join(d, "valu_issue_cost.txt"))
    # the kernels that trace the frame's segments: bounce 0 in k_first_small (which also shades its hits), later bounces in
    # k_extend_small (k_extend<0, true, true, *> when the scene is too large for the compact form).  The last bounces of a
    # batch run in k_tail_small (under 1 % of the frame's segments, shading included); its counters are left out (serialised
    # under the counters, a launch of it that takes a bounce runs on a nearly empty device for a long time, which says
    # nothing about the kernels that do the work), its segments stay in the count the others' operations are divided by.
    s2 = figures(counters(os.path.join(d, "pmc_s2_summary.txt"), ("k_first_small<", "k_extend_small<", "k_extend<0u, true, true")), seg2, cost)
    s2["workload"] = b["config"]["workload"] + " (one frame under rocprofv3 --pmc, kernels serialised)"
    s2["kernels"] = ("k_first_small<true> (bounce 0: camera ray, closest hit and the lean material code) + k_extend_small<false> (closest hit of later bounces); "
                     "k_tail_small<true> (the last bounces of a batch, under 1 % of the segments) is not in the counters")
    # the same for the later bounces' kernel alone (closest hit only)
    c_later = counters(os.path.join(d, "pmc_s2_summary.txt"), ("k_extend_small<false", "k_extend<0u, true, true, false"))
    # every path has one segment at bounce 0 (the batch tails' segments, which k_tail_small traces, stay in this count: under 1 %)
    seg_later = seg2 - b["config"]["path_samples_per_step"] * b["steps"]
    if c_later and seg_later > 0:
        s2["later_bounces_only"] = figures(c_later, seg_later, cost)
    l = last_json(os.path.join(d, "s5", "sq.log"))
    seg5 = 2 * l["segments_per_path"] * 1920 * 1080 * l["spp"]  # run_configs renders the frame twice
    s5 = figures(counters(os.path.join(d, "pmc_s5_summary.txt"), ("k_extend<35u, true, true",)), seg5, cost)
    s5["workload"] = l["config"] + f", {l['spp']} spp, two renders under rocprofv3 --pmc"
    out = {
        "git": git,
        "kernel_source_sha256_16": source_hash(),
        "collected": datetime.date.today().isoformat(),
        "how": "tools/collect_roofline_profile.sh: separate rocprofv3 --pmc runs for the SQ, FETCH_SIZE and WRITE_SIZE groups; "
               "counters summed over the traversal kernel's instantiations (bounce 0 and later bounces); FETCH_SIZE doubled "
               "(gfx950 reports half of a wide read stream); SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU count 4-cycle issue slots; "
               "instruction-kind counters in a run of their own; issue costs per instruction kind from tools/micro/valu_clock "
               "(valu_issue_cost.txt: cycles per wave64 instruction per SIMD with 4 waves per SIMD on every CU)",
        "valu_issue_cost_cycles_4_waves_per_simd": cost,
        "s2": s2,
        "s5": s5,
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
