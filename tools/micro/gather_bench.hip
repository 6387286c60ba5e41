// Micro-benchmark: achievable HBM throughput for gathered 64-byte records (the path pool's access pattern).
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip ; run: ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: record i; mode 1: record i*stride (ascending, sparse); mode 2: scrambled within windows of 1024 live entries
__device__ inline uint64_t rec_index(uint64_t i, int mode, uint32_t stride) {
  if (mode == 0) return i;
  if (mode == 1) return i * stride;
  uint64_t w = i >> 10, k = i & 1023;
  k = (k * 397u + 13u) & 1023;  // odd multiplier: a permutation of the window
  return (w * 1024 + k) * stride;
}

template <int RW>  // 0 read, 1 write, 2 read+write
__global__ void __launch_bounds__(256) k_gather(uint4* base, uint64_t n, int mode, uint32_t stride, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    uint4* p = base + rec_index(i, mode, stride) * 4;
    if (RW != 1) {
      uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (RW != 0) {
      uint4 v = make_uint4((uint32_t)i, acc, 2, 3);
      p[0] = v; p[1] = v; p[2] = v; p[3] = v;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const uint64_t bytes = 16ull << 30;
  uint4* buf; uint32_t* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t nrec = bytes / 64;
  const int grids[] = {256 * 8, 256 * 20};
  for (int rw = 0; rw < 3; rw++)
    for (int mode = 0; mode < 3; mode++)
      for (uint32_t stride : {1u, 2u, 4u, 8u}) {
        if (mode == 0 && stride != 1) continue;
        if (mode == 1 && stride == 1) continue;
        for (int g : grids) {
          const uint64_t n = nrec / stride;
          float best = 1e30f;
          for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            if (rw == 0) k_gather<0><<<g, 256>>>(buf, n, mode, stride, sink);
            else if (rw == 1) k_gather<1><<<g, 256>>>(buf, n, mode, stride, sink);
            else k_gather<2><<<g, 256>>>(buf, n, mode, stride, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double gb = (double)n * 64 * (rw == 2 ? 2 : 1) / 1e9;
          printf("%s mode=%d stride=%u grid=%d: %.2f ms, %.0f GB/s useful\n", rw == 0 ? "read " : rw == 1 ? "write" : "r+w  ", mode, stride, g, best, gb / (best * 1e-3));
          fflush(stdout);
        }
      }
  return 0;
}
