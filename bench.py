#!/usr/bin/env python3
"""Benchmark of the GPU render path on the RTIOW final scene.

A step is one full frame: every (pixel, stratum) path sample of the 1920x1080, 1024 spp, depth 50
RTIOW final random-spheres scene.  With N GPUs (one process per GPU, launched by torch.distributed.run)
the strata are dealt round-robin over the ranks, every rank renders its share into an f32 radiance
sum on its GPU, one RCCL reduce gathers the sums on rank 0, which resolves the image
(`rust-raytracer_amd/distributed.py: frame_step`).  The scene is committed (flattened and uploaded) before
the timed region, so inputs are resident in HBM.

Prints ONE JSON line on rank 0 (metric = whole-job path-samples/s).  Besides the contract's fields:
  roofline      the dominant kernel (closest-hit traversal) against the resource that bounds it on this workload,
                vector-ALU issue: lane-operations per second against the nominal peak (the rate of plain f32
                multiply-adds), and under `issue` the occupancy of the issue port once every instruction is priced at
                the cost measured for its kind.  The work per ray segment comes from a committed
                rocprofv3 PMC summary (profiles/r03/roofline_profile.json, with the git hash it was taken at and the
                hash of the kernel source it describes); the segments and the launch durations are measured live.
  roofline_s5   the same kernel on the one configuration whose scene tables do not fit the caches (the 100^3
                OpenSCAD-shaped instance lattice), against HBM bandwidth: algorithmic node / instance bytes per
                launch over the live launch duration, with the PMC-measured traffic beside it.
  rank_share    one rank's share of an 8-GPU job timed on this GPU (every 8th stratum, no reduce).
  cpu_baseline  the CPU oracle timed on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import ctypes as C
import hashlib
import importlib
import json
import os
import sys
import time

import torch  # first: the library below must share torch's HIP runtime
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rust-raytracer_amd")

# Device record sizes that define the algorithmic bytes of one ray segment of the traversal kernel.
BYTES_WIDE_NODE = 112   # DWNode: f32 boxes of four children (6 x 16 B) + four child references (16 B) of a 128-B record
BYTES_SPHERE = 64       # the part of a sphere record a test reads: centre, motion, radius (7 f64) + material word
BYTES_INSTANCE = 128    # DTranslate with its sphere inlined: one record per instanced-sphere test
BYTES_SEGMENT_STATE = 64 + 16  # PathRay record read (not at bounce 0, whose rays are computed), PathHit record written
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
# Vector-ALU peak: a CDNA4 SIMD is 32 lanes wide, so a wave64 instruction of the fast kind (plain f32 multiply-add, add,
# multiply; 32-bit integer add and logic) issues over 2 cycles: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz (the data
# sheet's 157 TFLOP/s fp32 is this rate times 2 for multiply-add).  Every other kind of instruction (f64, min / max,
# selects, compares, conversions, 3-operand integer forms), i.e. four fifths of what the traversal executes, issues at
# little more than half that rate (tools/micro/valu_clock measures both; the profile carries the figures), so the
# kernel's own instruction mix cannot reach this peak: `issue` below prices the mix.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PROFILE_JSON = os.path.join(ROOT, "profiles", "r03", "roofline_profile.json")
KERNEL_SOURCES = [os.path.join(ROOT, "rust-raytracer_amd", "csrc", f) for f in ("kernels.hpp", "device_scene.h")]


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_profile():
    """The committed PMC-derived figures; None (and the reason) when absent or taken on other kernel sources."""
    if not os.path.exists(PROFILE_JSON):
        return None, "no profile file"
    try:
        with open(PROFILE_JSON) as fh:
            prof = json.load(fh)
    except ValueError:
        return None, "profile file is not valid JSON"
    if prof.get("kernel_source_sha256_16") != kernel_source_hash():
        return prof, "profile was taken on other kernel sources (stale)"
    return prof, None


def cpu_baseline(width, depth, seed):
    """The CPU oracle (a port of the reference's algorithm, crates/cli/src/main.rs:83-182 block loop)
    timed on a bounded sample of the same workload: same scene, same resolution, 64 spp."""
    oracle_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(oracle_so):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    orc = pkg.bindings.Backend(C.CDLL(oracle_so), "orc_", "oracle", sum_dtype="float64")
    # Threads: every core this process may run on (the reference starts num_cpus workers, crates/cli/src/main.rs:120),
    # within the share of the host the job was given when the machine is shared (the launcher exports it as
    # OMP_NUM_THREADS; os.cpu_count() names the whole machine there).
    host_cores = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    share = os.environ.get("OMP_NUM_THREADS", "")
    threads = max(1, min(usable, int(share)) if share.isdigit() and int(share) > 0 else usable)
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    spp = 64
    so = pkg.scenes.build(orc, "random_spheres", width=width, spp=spp, depth=depth)
    t0 = time.perf_counter()
    _, _, st = so.render_frame(seed=seed, flags=(threads << 8))
    dt = time.perf_counter() - t0
    return {
        "value": st["paths"] / dt,
        "unit": "path-samples/s",
        "cores": threads,
        "host_cores": host_cores,
        "cpu_model": cpu_model,
        "threads_note": "one worker thread per core of this job's share of the host (OMP_NUM_THREADS when set, else the affinity mask); "
                        "host_cores is what the whole machine reports",
        "kind": "port",
        "sample": f"same scene, {so.image_size()[0]}x{so.image_size()[1]}, {spp} spp (8x8 strata), depth {depth}: {st['paths']} path-samples in {dt:.1f} s",
    }


def traversal_sums(stats):
    seg = sum(s["segments"] for s in stats)
    return {
        "seg": seg,
        "bvh": sum(s["bvh_visits"] for s in stats),
        "prim": sum(s["prim_tests"] for s in stats),
        "launches": sum(s["trace_launches"] for s in stats),
        "ms_trace": sum(s["ms_trace"] for s in stats),
        "ms_total": sum(s["ms_total"] for s in stats),
        "ms_shade": sum(s["ms_shade"] for s in stats),
        "paths": sum(s["paths"] for s in stats),
    }


def s5_roofline(gpu, dev, seed, n, spp, prof, prof_note):
    """The instance lattice (top-level hierarchy over n^3 Translate(Group[Sphere]), what crates/openscad emits for nested
    loops): the configuration whose culling nodes and instances come from beyond the caches."""
    t0 = time.perf_counter()
    sc = gpu.new_scene()
    pkg.scenes.scad_lattice(sc, n=n, width=1920, spp=spp, depth=50)
    sc.commit()
    t_build = time.perf_counter() - t0
    w, h = sc.image_size()
    d_sum = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    sc.render_accumulate_device(d_sum.data_ptr(), seed=seed, stream=stream)  # warm-up (allocates the path pools)
    torch.cuda.synchronize()
    stats = []
    t0 = time.perf_counter()
    for _ in range(2):
        stats.append(sc.render_accumulate_device(d_sum.data_ptr(), seed=seed, stream=stream, flags=pkg.bindings.RENDER_TIME_KERNELS))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = traversal_sums(stats)
    alg = t["bvh"] * BYTES_WIDE_NODE + t["prim"] * BYTES_INSTANCE + t["seg"] * BYTES_SEGMENT_STATE
    achieved = alg / (t["ms_trace"] * 1e-3) / 1e9
    p5 = (prof or {}).get("s5") or {}
    traffic = p5.get("hbm_bytes_per_segment")
    traffic_per_launch = traffic * t["seg"] / t["launches"] if (traffic and not prof_note) else None
    avg_launch_s = t["ms_trace"] * 1e-3 / t["launches"]
    frac = achieved / HBM_PEAK_GBS
    out = {
        "workload": f"OpenSCAD-shaped instance lattice {n}^3 (+ ground), {w}x{h}, {spp} spp, depth 50; built and committed in {t_build:.1f} s",
        "path_samples_per_s": t["paths"] / dt,
        "kernel": "k_extend<35, true, true, *> (instanced spheres: 4-wide culling nodes and fused instance records from memory)",
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": frac if frac <= 1.0 else None,
        "traffic": traffic_per_launch,
        # what actually crosses the L2 / memory boundary (PMC bytes per launch over the live launch duration): the figure
        # to read against the HBM peak.  The algorithmic rate above counts every record as fetched from memory.
        "hbm_side_GBps": traffic_per_launch / avg_launch_s / 1e9 if traffic_per_launch else None,
        "hbm_side_frac": traffic_per_launch / avg_launch_s / 1e9 / HBM_PEAK_GBS if traffic_per_launch else None,
        "limiter": "memory latency (dependent node fetches; waves wait most of their cycles), not bandwidth",
        "wave_wait_share": p5.get("wave_wait_share"),
        "algorithmic_bytes_per_launch": alg / t["launches"],
        "avg_launch_ms": t["ms_trace"] / t["launches"],
        "launches": t["launches"],
        "per_segment": {"wide_nodes": t["bvh"] / t["seg"], "instance_tests": t["prim"] / t["seg"], "bytes": alg / t["seg"]},
        "note": "achieved prices every node / instance record as a fetch from memory; the tables (180 MB) fit the 256 MB "
                "Infinity Cache, so `traffic` (PMC: 2 x FETCH_SIZE + WRITE_SIZE beyond L2, per launch) includes cache hits",
    }
    if frac > 1.0:
        out["note"] += f"; the modelled rate ({achieved:.0f} GB/s) exceeds the HBM peak because most records are cache hits: frac withheld"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-s5", action="store_true", help="skip the instance-lattice (HBM-bound) roofline leg")
    ap.add_argument("--s5-n", type=int, default=100)
    ap.add_argument("--s5-spp", type=int, default=64)
    ap.add_argument("--no-rank-share", action="store_true", help="skip timing one rank's share of an 8-GPU job")
    ap.add_argument("--no-exclusive", action="store_true", help="skip the extra one-stream frame that gives exclusive kernel times (counter runs: every frame rendered counts)")
    ap.add_argument("--simulate-rank-of", type=int, default=8, help="ranks of the job whose per-rank share is timed on this GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    gpu = pkg.load_gpu()
    sc = pkg.scenes.build(gpu, "random_spheres", width=args.width, spp=args.spp, depth=args.depth)
    w, h = sc.image_size()
    q = sc.sqrt_spp()
    n_paths_frame = w * h * q * q
    d_sum = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
    d_rgb8 = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    D = pkg.distributed

    def make_step(n_ranks, my_rank, reduce_world, flags=0, stats_out=None):
        def render_partial(offset, stride):
            d_sum.zero_()
            st = sc.render_accumulate_device(d_sum.data_ptr(), seed=args.seed, stratum_offset=offset, stratum_stride=stride, stream=stream, flags=flags)
            if stats_out is not None:
                stats_out.append(st)
            return d_sum

        def resolve_frame(total):
            sc.resolve_device(total.data_ptr(), w * h, d_rgb8_ptr=d_rgb8.data_ptr(), stream=stream)
            return d_rgb8

        if reduce_world:
            return lambda: D.frame_step(render_partial, resolve_frame, my_rank, n_ranks)
        return lambda: render_partial(my_rank, n_ranks)  # one rank's share alone: no reduce, no resolve

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = make_step(world, rank, True)
    for _ in range(args.warmup):
        warm()
    fence()
    stats = []
    step = make_step(world, rank, True, flags=pkg.bindings.RENDER_TIME_KERNELS, stats_out=stats)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_paths_frame * args.steps / dt
        default_workload = args.width == 1920 and args.spp == 1024 and args.depth == 50
        prof, prof_note = load_profile()
        # traversal-kernel roofline from this rank's launches in the timed steps
        tr = traversal_sums(stats)
        seg, launches, ms_trace = tr["seg"], tr["launches"], tr["ms_trace"]
        lds_bytes = tr["bvh"] * BYTES_WIDE_NODE + tr["prim"] * BYTES_SPHERE + seg * BYTES_SEGMENT_STATE
        p2 = (prof or {}).get("s2") or {}
        ops_per_seg = p2.get("valu_lane_ops_per_segment")
        usable = bool(ops_per_seg) and not prof_note and default_workload
        achieved = ops_per_seg * seg / (ms_trace * 1e-3) / 1e12 if (usable and ms_trace > 0) else None
        roofline = {
            "kernel": "k_first_small<true> (bounce 0: camera ray, closest hit and the lean material code in one kernel) + k_extend_small<false> (closest hit of "
                      "later bounces) + k_tail_small<true> (the last bounces of a batch in one kernel): ordered traversal, compact form, f32 culling of 4-wide nodes, "
                      "f64 sphere tests, scene tables in LDS",
            "bound": "valu",
            "achieved": achieved,
            "peak": VALU_PEAK_TLANEOPS,
            "unit": "Tlane-op/s",
            "frac": achieved / VALU_PEAK_TLANEOPS if achieved else None,
            "traffic": (p2.get("hbm_bytes_per_segment") * seg / launches) if (usable and p2.get("hbm_bytes_per_segment")) else None,
            "definition": "achieved = vector-ALU lane-operations per ray segment (SQ_THREAD_CYCLES_VALU: active lanes x instructions, from the "
                          "profile) x segments of the measured launches / their summed HIP-event durations; peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz, "
                          "the rate of plain f32 multiply-adds.",
            # what the kernel's own instruction mix allows: measured issue cost per instruction kind x the mix, against the SIMD
            # cycles one instruction took with the kernel alone on the device (all from the committed profile)
            "issue": p2.get("valu_issue") if usable else None,
            "profile": {
                "path": os.path.relpath(PROFILE_JSON, ROOT),
                "git": (prof or {}).get("git"),
                "kernel_source_sha256_16": (prof or {}).get("kernel_source_sha256_16"),
                "note": prof_note or ("figures apply to the default workload only" if not default_workload else None),
                "alone": {k: p2.get(k) for k in ("lane_utilisation", "frac_of_nominal_valu_peak_alone", "wave_wait_share")},
            },
            "avg_launch_ms": ms_trace / launches if launches else None,
            "launches": launches,
            "per_segment": {
                "wide_nodes": tr["bvh"] / seg if seg else None,
                "prim_tests": tr["prim"] / seg if seg else None,
                "valu_lane_ops": ops_per_seg if usable else None,
            },
            "segments_per_path": seg / tr["paths"] if tr["paths"] else None,
            "trace_share_of_device_time": ms_trace / tr["ms_total"] if tr["ms_total"] else None,
            # the north star's "traversal bandwidth": node / sphere / path-record bytes the traversal consumes per second,
            # nearly all of them served by LDS; a rate of its own, not a fraction of any memory's peak
            "lds_served_traversal_GBps": lds_bytes / (ms_trace * 1e-3) / 1e9 if ms_trace > 0 else None,
        }
        if roofline["frac"] is not None and roofline["frac"] > 1.0:
            roofline["note"] = "frac above 1: the work model (lane-operations per segment from the profile) does not fit these launches"
        # One more frame on a single stream (after the timed region): its kernels do not overlap, so the traversal and
        # shade durations are exclusive and add up to (at most) the frame.  The timed region above runs two batches at a
        # time on two streams, where a launch's event interval includes time it shared with the other batch's kernels.
        exclusive = None
        try:
            if args.no_exclusive:
                raise RuntimeError("skipped (--no-exclusive)")
            os.environ["CHIP_STREAMS"] = "1"
            ex_stats = []
            ex_step = make_step(world, rank, False, flags=pkg.bindings.RENDER_TIME_KERNELS, stats_out=ex_stats)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ex_step()
            torch.cuda.synchronize()
            ex_ms = 1e3 * (time.perf_counter() - t1)
            ex = traversal_sums(ex_stats)
            ex_ach = ops_per_seg * ex["seg"] / (ex["ms_trace"] * 1e-3) / 1e12 if (usable and ex["ms_trace"] > 0) else None
            exclusive = {
                "what": "one frame of this rank's share with one batch in flight (CHIP_STREAMS=1), outside the timed region",
                "frame_ms": ex_ms,
                "traversal_ms": ex["ms_trace"],
                "bin_and_shade_ms": ex["ms_shade"],
                "avg_launch_ms": ex["ms_trace"] / ex["launches"] if ex["launches"] else None,
                "achieved": ex_ach,
                "frac": ex_ach / VALU_PEAK_TLANEOPS if ex_ach else None,
            }
        except Exception as e:  # an optional leg must not lose the headline line
            exclusive = {"error": repr(e)}
        finally:
            os.environ.pop("CHIP_STREAMS", None)
        roofline["exclusive"] = exclusive
        # The figures at the top of the object are the exclusive ones when that frame was rendered: kernel time that adds up
        # to (at most) the frame.  The timed region's own figures, whose event intervals overlap across the two streams,
        # stay beside them under a name that says so.
        roofline["timed_region_two_streams"] = {
            "achieved": roofline["achieved"], "frac": roofline["frac"], "avg_launch_ms": roofline["avg_launch_ms"],
            "note": "HIP-event intervals of the timed steps' launches; with two batches in flight an interval includes time the launch "
                    "shared with the other batch's kernels, so the sum over launches exceeds the step",
        }
        if exclusive and exclusive.get("achieved"):
            roofline["achieved"], roofline["frac"], roofline["avg_launch_ms"] = exclusive["achieved"], exclusive["frac"], exclusive["avg_launch_ms"]
            roofline["measured_on"] = "one frame rendered with one batch in flight right after the timed region (HIP events on the render stream): exclusive kernel time"
        else:
            roofline["measured_on"] = "the timed region (two batches in flight: intervals overlap)"
        out = {
            "metric": "path-samples/sec (whole node), RTIOW final scene",
            "value": value,
            "unit": "path-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64 geometry (hit decisions, t, points, normals) + f32 culling boxes, radiance and framebuffer",
            "data": "synthetic",
            "config": {
                "workload": f"RTIOW final random-spheres scene (scene seed 1), {w}x{h}, {args.spp} spp ({q}x{q} strata), depth {args.depth}, one frame per step",
                "path_samples_per_step": n_paths_frame,
                "parallelism": f"strata round-robin over {world} GPU(s), one RCCL reduce per frame",
                "render_seed": args.seed,
            },
            "roofline": roofline,
            # HIP-event intervals of the timed steps.  With two batches in flight the traversal and shade figures are sums
            # over two streams whose kernels overlap in time, so together they exceed render_call; roofline.exclusive holds
            # the non-overlapping figures of a one-stream frame.
            "device_ms_per_step": {
                "render_call": tr["ms_total"] / args.steps,
                "traversal_summed_over_streams": ms_trace / args.steps,
                "bin_and_shade_summed_over_streams": tr["ms_shade"] / args.steps,
                "traversal_exclusive": (exclusive or {}).get("traversal_ms"),
                "bin_and_shade_exclusive": (exclusive or {}).get("bin_and_shade_ms"),
            },
            "shading_mode": "default (f32-seeded Newton reciprocals / f32 trigonometry in scatter directions and PDFs); checked against the oracle "
                            "statistically (tests/test_gpu_parity.py: values within 2e-3 on >= 99 % of pixels); the bit- and segment-exact "
                            "comparisons run CHIP_RENDER_EXACT_SHADING",
        }
        def optional(name, fn):
            try:
                out[name] = fn()
            except Exception as e:  # an optional leg must not lose the headline line
                out[name] = {"error": repr(e)}

        def rank_share_leg():
            # one rank's share of an N-GPU job on this GPU: every N-th stratum, no reduce
            n_sim = args.simulate_rank_of
            share = make_step(n_sim, 0, False)
            share()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                share()
            torch.cuda.synchronize()
            ms_share = 1e3 * (time.perf_counter() - t0) / 3
            return {
                "ranks": n_sim,
                "ms_per_share": ms_share,
                "ms_full_frame_one_gpu": ms_per_step,
                "implied_scaling_ceiling": ms_per_step / ms_share,
                "note": f"strata 0, {n_sim}, {2 * n_sim}, ... of every pixel on this GPU, no reduce (24.9 MB over xGMI, < 1 ms); "
                        f"the ceiling is T(1) / T(share), at most {n_sim}",
            }

        if world == 1 and not args.no_rank_share and args.simulate_rank_of > 1:
            optional("rank_share", rank_share_leg)
        if world == 1 and not args.no_s5:
            optional("roofline_s5", lambda: s5_roofline(gpu, dev, args.seed, args.s5_n, args.s5_spp, prof, prof_note))
        if world == 1 and not args.no_cpu_baseline:
            optional("cpu_baseline", lambda: cpu_baseline(args.width, args.depth, args.seed))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # (rank 0 renders one more frame after the timed region: the others wait for it before the group goes)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
