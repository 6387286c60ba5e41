// How fast does one SIMD of an MI355X issue vector-ALU instructions of several kinds, and how many waves does it take?
// Each wave runs a long loop of eight independent chains of one instruction kind; the host times the whole kernel and
// prints cycles per wave-instruction per SIMD at the 2.4 GHz shader clock.  A block is 64 x waves-per-SIMD x 4
// threads and asks for 84 KB of LDS, so exactly one block is resident per CU; the grid is eight rounds of 256 blocks.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_clock valu_clock.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
enum Kind { FMA32, FMA64, ADD64, IADD, MINU, CNDMASK, MAX3, PKFMA, MAX32, CMP32, AND32, ANDOR, MUL64, MADU24, CVT, ADD32, MUL32, LSHL, MAD64, MULHI, MULLO, RCP32, SQRT64, RCP64 };
template <int K>
__global__ void __launch_bounds__(1024) k_loop(double* out, int iters) {
  float f[8]; double d[8]; unsigned u[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { f[j] = threadIdx.x + j; d[j] = threadIdx.x + j; u[j] = threadIdx.x * 7u + j; }
  const float fm = 1.0000001f, fc = 1e-7f;
  const double dm = 1.0000001, dc = 1e-7;
  const unsigned long long msk = 0x5555aaaa3333ccccull + blockIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (K == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(fm), "v"(fc));
      if (K == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(dm), "v"(dc));
      if (K == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(dc));
      if (K == IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == MINU) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "s"(msk));
      if (K == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(fm), "v"(fc));
      if (K == MAX32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fm));
      if (K == CMP32) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(f[j]), "v"(fm) : "vcc");
      if (K == AND32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, 3" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[j]) : "v"(dm));
      if (K == MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == CVT) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j]));
      if (K == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fc));
      if (K == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j]) : "v"(fm));
      if (K == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[j]));
      if (K == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[j]) : "v"(u[j]), "v"(u[(j + 1) & 7]) : "vcc");
      if (K == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
      if (K == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[j]));
      if (K == SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[j]));
      if (K == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[j]));
      if (K == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[j]) : "v"(dm));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) s += f[j] + d[j] + u[j];
  extern __shared__ double pad[];
  if (iters < 0) pad[threadIdx.x] = s;  // (keeps the allocation)
  out[(blockIdx.x & 255) * blockDim.x + threadIdx.x] = s;
}
template <int K>
static void run(const char* name, double* out) {
  const int iters = 20000, blocks = 256 * 8;
  const size_t lds = 84 * 1024;
  hipFuncSetAttribute((const void*)k_loop<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  printf("%-18s", name);
  for (int wps = 1; wps <= 4; wps++) {
    const int threads = 256 * wps;
    hipLaunchKernelGGL(k_loop<K>, dim3(blocks), dim3(threads), lds, 0, out, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_loop<K>, dim3(blocks), dim3(threads), lds, 0, out, iters);
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const double instr_per_simd = (double)wps * 8.0 * iters * (blocks / 256);
    printf("  %d wave/SIMD: %5.2f cyc/instr", wps, us * 2400.0 / instr_per_simd);
  }
  printf("\n");
}
int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 1024);
  run<FMA32>("v_fma_f32", out);
  run<ADD32>("v_add_f32", out);
  run<MUL32>("v_mul_f32", out);
  run<LSHL>("v_lshlrev_b32", out);
  run<MAX3>("v_max3_f32", out);
  run<PKFMA>("v_pk_fma_f32", out);
  run<FMA64>("v_fma_f64", out);
  run<ADD64>("v_add_f64", out);
  run<IADD>("v_add_u32", out);
  run<MINU>("v_min_u32", out);
  run<CNDMASK>("v_cndmask_b32", out);
  run<MAX32>("v_max_f32", out);
  run<CMP32>("v_cmp_le_f32", out);
  run<AND32>("v_and_b32", out);
  run<ANDOR>("v_and_or_b32", out);
  run<MUL64>("v_mul_f64", out);
  run<MADU24>("v_mad_u32_u24", out);
  run<CVT>("v_cvt_f32_f64", out);
  run<MAD64>("v_mad_u64_u32", out);
  run<MULHI>("v_mul_hi_u32", out);
  run<MULLO>("v_mul_lo_u32", out);
  run<RCP32>("v_rcp_f32", out);
  run<SQRT64>("v_sqrt_f64", out);
  run<RCP64>("v_rcp_f64", out);
  return 0;
}
