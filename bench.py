#!/usr/bin/env python3
"""Benchmark of the GPU render path on the RTIOW final scene.

A step is one full frame: every (pixel, stratum) path sample of the 1920x1080, 1024 spp, depth 50
RTIOW final random-spheres scene.  With N GPUs (one process per GPU, launched by torch.distributed.run)
the strata are dealt round-robin over the ranks, every rank renders its share into an f32 radiance
sum on its GPU, one RCCL reduce gathers the sums on rank 0, which resolves the image.  The scene is
committed (flattened and uploaded) before the timed region, so inputs are resident in HBM.

Prints ONE JSON line on rank 0 (metric = whole-job path-samples/s).
"""
import argparse
import ctypes as C
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
BYTES_BVH_NODE = 64    # DTNode: f32 boxes of both children (2 x 6 f32) + two child refs, one 64-B record per visit
BYTES_SPHERE = 64      # the part of DSphere a test reads: centre, motion, radius (7 f64) + material word
BYTES_SEGMENT_STATE = 64 + 16  # PathRay record read (it carries the path id; not at bounce 0, whose rays are computed), PathHit record write
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
# HBM bytes per launch of the traversal kernel on the default workload (1920x1080, 64-strata batches, depth 50),
# from rocprofv3 PMC passes of this build (2 x FETCH_SIZE + WRITE_SIZE, in KB, summed over the launches of one
# frame and divided by their number): profiles/r01/pmc_traffic_k_extend.json.  None for any other workload.
PMC_TRAFFIC_DEFAULT_WORKLOAD = 414.0e6


def cpu_baseline(width, depth, seed):
    """The CPU oracle (a port of the reference's algorithm, crates/cli/src/main.rs:83-182 block loop)
    timed on a bounded sample of the same workload: same scene, same resolution, 64 spp."""
    oracle_so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(oracle_so):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    orc = pkg.bindings.Backend(C.CDLL(oracle_so), "orc_", "oracle", sum_dtype="float64")
    threads = min(os.cpu_count() or 1, 16)
    spp = 64
    so = pkg.scenes.build(orc, "random_spheres", width=width, spp=spp, depth=depth)
    t0 = time.perf_counter()
    import numpy as np

    _, _, st = so.render_frame(seed=seed, sum_dtype=np.float64, flags=(threads << 8))
    dt = time.perf_counter() - t0
    return {
        "value": st["paths"] / dt,
        "unit": "path-samples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"same scene, {so.image_size()[0]}x{so.image_size()[1]}, {spp} spp (8x8 strata), depth {depth}: {st['paths']} path-samples in {dt:.1f} s",
    }


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

    def step(flags=0):
        d_sum.zero_()
        st = sc.render_accumulate_device(
            d_sum.data_ptr(), seed=args.seed, stratum_offset=rank, stratum_stride=world, stream=stream, flags=flags
        )
        if world > 1:
            dist.reduce(d_sum, dst=0)
        if rank == 0:
            sc.resolve_device(d_sum.data_ptr(), w * h, d_rgb8_ptr=d_rgb8.data_ptr(), stream=stream)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step(flags=pkg.bindings.RENDER_TIME_KERNELS))
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = n_paths_frame * args.steps / dt
        # traversal-kernel roofline from this rank's launches in the timed steps
        seg = sum(s["segments"] for s in stats)
        bvh = sum(s["bvh_visits"] for s in stats)
        prim = sum(s["prim_tests"] for s in stats)
        launches = sum(s["trace_launches"] for s in stats)
        ms_trace = sum(s["ms_trace"] for s in stats)
        alg_bytes = bvh * BYTES_BVH_NODE + prim * BYTES_SPHERE + seg * BYTES_SEGMENT_STATE
        achieved = alg_bytes / (ms_trace * 1e-3) / 1e9 if ms_trace > 0 else None
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
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"RTIOW final random-spheres scene (scene seed 1), {w}x{h}, {args.spp} spp ({q}x{q} strata), depth {args.depth}, one frame per step",
                "path_samples_per_step": n_paths_frame,
                "parallelism": f"strata round-robin over {world} GPU(s), one RCCL reduce per frame",
                "render_seed": args.seed,
            },
            "roofline": {
                "kernel": "k_extend<0, true, true, *> (ordered BVH traversal: f32 child-box culling, f64 sphere tests, LDS scene cache)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": PMC_TRAFFIC_DEFAULT_WORKLOAD if (world == 1 and args.width == 1920 and args.spp == 1024 and args.depth == 50) else None,
                "note": "achieved = algorithmic bytes (every node / sphere record counted as an HBM fetch) / kernel time; the scene "
                        "tables are LDS-resident, so HBM only carries the path records (traffic) and the kernel is VALU-bound",
                "algorithmic_bytes_per_launch": alg_bytes / launches if launches else None,
                "avg_launch_ms": ms_trace / launches if launches else None,
                "launches": launches,
                "per_segment": {
                    "bvh_nodes": bvh / seg if seg else None,
                    "prim_tests": prim / seg if seg else None,
                    "bytes": alg_bytes / seg if seg else None,
                },
                "segments_per_path": seg / (sum(s["paths"] for s in stats)) if stats else None,
                "trace_share_of_device_time": ms_trace / sum(s["ms_total"] for s in stats) if stats else None,
            },
            "device_ms_per_step": {
                "render_call": sum(s["ms_total"] for s in stats) / args.steps,
                "traversal": ms_trace / args.steps,
                "bin_and_shade": sum(s["ms_shade"] for s in stats) / args.steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.width, args.depth, args.seed)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
