"""The N > 1 path (strata dealt round-robin over ranks, one reduce per frame, resolve on rank 0) run with
world_size 2 over gloo on the CPU; the per-rank renderer here is the oracle, the sharding / reduce /
resolve code is the product's: `distributed.frame_step`, the function bench.py's timed step calls (there with the
GPU library as renderer and resolver)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    import ctypes as C
    import importlib

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("rust-raytracer_amd")
    orc = pkg.bindings.Backend(C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")), "orc_", "oracle", sum_dtype="float64")
    sc = pkg.scenes.build(orc, "cornell_box", width=24, spp=9, depth=6)
    D = pkg.distributed

    def partial(offset, stride):
        img, _, _ = sc.render_frame(seed=4, stratum_offset=offset, stratum_stride=stride, sum_dtype=np.float64)
        return torch.from_numpy(img)

    kept = {}

    def resolve(total):
        kept["sum"] = total
        return D.resolve(total, sc.sqrt_spp())

    rgb8 = D.frame_step(partial, resolve, rank, world)
    if rank == 0:
        np.save(os.path.join(tmpdir, "sum.npy"), kept["sum"].numpy())
        np.save(os.path.join(tmpdir, "rgb8.npy"), rgb8.numpy())
    else:
        assert rgb8 is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_equals_single_rank(tmp_path, pkg, oracle):
    import importlib

    importlib.import_module("rust-raytracer_amd.distributed")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sum.npy")
    rgb8 = np.load(tmp_path / "rgb8.npy")
    sc = pkg.scenes.build(oracle, "cornell_box", width=24, spp=9, depth=6)
    want, want8, _ = sc.render_frame(seed=4, want_rgb8=True, sum_dtype=np.float64)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    assert (np.abs(rgb8.astype(int) - want8.astype(int)) > 0).mean() < 1e-3
    D = pkg.distributed
    assert D.strata_of_rank(9, 0, 2) == [0, 2, 4, 6, 8] and D.strata_of_rank(9, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((D.strata_of_rank(1024, r, 8) for r in range(8)), [])) == list(range(1024))


def test_resolve_matches_reference_rules(pkg):
    import importlib

    D = importlib.import_module("rust-raytracer_amd.distributed")
    s = torch.tensor([[4.0, float("nan"), -1.0], [0.25 * 4, 0.5 * 4, 1.0 * 4]])
    g = D.resolve(s, 2, quantise=False)
    assert g[0].tolist() == [0.999, 0.0, 0.0]  # clamp, NaN -> 0, negative -> 0
    assert abs(g[1, 0].item() - 0.5) < 1e-15 and abs(g[1, 1].item() - 0.7071067811865476) < 1e-15 and g[1, 2].item() == 0.999
    q = D.resolve(s, 2)
    assert q[1].tolist() == [int(0.5 * 255.999), int(0.7071067811865476 * 255.999), int(0.999 * 255.999)]
