#!/usr/bin/env python3
"""What does the traversal kernel lose, and a streaming kernel get, when the two share the device?
Renders the RTIOW frame on one stream (one batch in flight) while a second stream copies a large buffer in a loop
(torch's copy kernel: no LDS, few registers, nothing but loads and stores), and compares with each running alone."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("rust-raytracer_amd")
os.environ["CHIP_STREAMS"] = "1"
dev = torch.device("cuda", 0)
gpu = pkg.load_gpu()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
# second argument: what the other stream runs: "copy" (loads and stores only) or "sin" (an elementwise kernel with about the
# shade kernels' ratio of vector instructions to bytes moved: some 25 instructions per 8 bytes)
op = sys.argv[2] if len(sys.argv) > 2 else "copy"
sc = pkg.scenes.build(gpu, "random_spheres", width=1920, spp=spp, depth=50)
w, h = sc.image_size()
d_sum = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
n = 1 << 30  # 4 GiB of f32 read + 4 GiB written per copy
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)

def side_op():
    if op == "sin":
        torch.sin(a, out=b)
    else:
        b.copy_(a)


def render():
    return sc.render_accumulate_device(d_sum.data_ptr(), seed=1, stream=main.cuda_stream, flags=pkg.bindings.RENDER_TIME_KERNELS)

render(); torch.cuda.synchronize()
t0 = time.perf_counter(); st = render(); torch.cuda.synchronize(); t_alone = time.perf_counter() - t0
print(f"render alone: {1e3 * t_alone:.1f} ms (trav {st['ms_trace']:.1f} shade {st['ms_shade']:.1f})")
with torch.cuda.stream(side):
    side_op()
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(side):
    for _ in range(8):
        side_op()
torch.cuda.synchronize()
t_copy = (time.perf_counter() - t0) / 8
print(f"{op} alone: {1e3 * t_copy:.2f} ms per 8 GiB moved = {8 * 2**30 / t_copy / 1e12:.2f} TB/s")
# together: keep the copy stream busy for the whole render
reps = int(t_alone / t_copy * 3) + 4
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    ev0.record(side)
    for _ in range(reps):
        side_op()
    ev1.record(side)
t0 = time.perf_counter(); st = render(); t_with = time.perf_counter() - t0
torch.cuda.synchronize()
t_side = ev0.elapsed_time(ev1) / 1e3
print(f"render beside {op}: {1e3 * t_with:.1f} ms (trav {st['ms_trace']:.1f} shade {st['ms_shade']:.1f}); {reps} copies took {1e3 * t_side:.1f} ms "
      f"= {reps * 8 * 2**30 / t_side / 1e12:.2f} TB/s over their whole span")
