"""Times one rank's share of the headline frame (every n-th stratum of every pixel, no reduce) for jobs of 8, 4 and 2 ranks on this GPU.
usage: [CHIP_STREAMS=..] [CHIP_POOL_PATHS=..] [CHIP_TAIL_PATHS=..] share_time.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("rust-raytracer_amd")
gpu = pkg.load_gpu()
sc = pkg.scenes.build(gpu, "random_spheres", width=1920, spp=1024, depth=50)
w, h = sc.image_size()
d_sum = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for n in (8, 4, 2):
    for _ in range(2):
        d_sum.zero_(); sc.render_accumulate_device(d_sum.data_ptr(), seed=1, stratum_offset=0, stratum_stride=n, stream=stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        d_sum.zero_(); sc.render_accumulate_device(d_sum.data_ptr(), seed=1, stratum_offset=0, stratum_stride=n, stream=stream)
    torch.cuda.synchronize(); print(os.environ.get("CHIP_STREAMS"), os.environ.get("CHIP_POOL_PATHS"), "ranks", n, "ms per share %.2f" % (1e3 * (time.perf_counter() - t0) / 3), flush=True)
