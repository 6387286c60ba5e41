"""Multi-GPU frame assembly: one process per GPU, samples split by stratum, one reduce per frame.

Every (pixel, stratum) sample of crates/core/src/camera.rs:335-341 is independent, so rank r of W
renders the strata s with s % W == r into a radiance-sum tensor, the sums are reduced onto rank 0
(RCCL when the tensors live on GPUs, gloo on CPU), and rank 0 applies camera.rs:343-344.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def strata_of_rank(n_strata: int, rank: int, world: int):
    """Global stratum indices rendered by `rank` (round-robin, so every rank covers the pixel area evenly)."""
    return list(range(rank, n_strata, world))


def reduce_frame(partial_sum: torch.Tensor, world: int, dst: int = 0) -> torch.Tensor:
    """Sums the per-rank radiance sums onto rank `dst` (in place).  NaNs are preserved, as the reference
    only zeroes NaNs on the final per-pixel sum (camera.rs:343)."""
    if world > 1:
        dist.reduce(partial_sum, dst=dst, op=dist.ReduceOp.SUM)
    return partial_sum


def resolve(sum_rgb: torch.Tensor, sqrt_spp: int, quantise: bool = True) -> torch.Tensor:
    """camera.rs:343-344 (scale by 1/sqrt_spp^2, NaN -> 0, gamma = sqrt, clamp [0, 0.999]) and, when
    `quantise`, crates/cli/src/main.rs:193-198 (x 255.999, truncate to u8)."""
    s = sum_rgb.to(torch.float64)
    s = torch.where(torch.isnan(s), torch.zeros_like(s), s)
    v = s * (1.0 / float(sqrt_spp * sqrt_spp))
    g = torch.where(v > 0, torch.sqrt(torch.clamp(v, min=0.0)), torch.zeros_like(v))
    g = torch.clamp(g, 0.0, 0.999)
    if quantise:
        return (g * 255.999).to(torch.uint8)
    return g


def render_frame_sharded(render_partial, rank: int, world: int) -> torch.Tensor:
    """`render_partial(stratum_offset, stratum_stride)` returns this rank's radiance-sum tensor; the
    result on rank 0 is the whole frame's sum (other ranks keep their partial sums)."""
    t = render_partial(rank, world)
    return reduce_frame(t, world)


def frame_step(render_partial, resolve_frame, rank: int, world: int):
    """One whole frame of the N-rank job, the step bench.py times: every rank renders its strata, the sums are
    reduced onto rank 0, rank 0 resolves (`resolve_frame(sum)`; camera.rs:343-344 and the 8-bit quantisation).
    Returns rank 0's resolved frame (None on the other ranks)."""
    total = render_frame_sharded(render_partial, rank, world)
    return resolve_frame(total) if rank == 0 else None
