// Micro-benchmark: does a wave64 VALU instruction on gfx950 (SIMD-32, 2 passes of 32 lanes) skip a pass whose 32 lanes
// are all masked off?  Times a dependent-free v_fma_f32 stream under different EXEC masks.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/exec_half.hip -o build/exec_half && build/exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) fma_stream(float* out, unsigned long long mask, int iters)
{
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  const unsigned lane = threadIdx.x & 63u;
  if ((mask >> lane) & 1ull) {
    for (int i = 0; i < iters; ++i) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
          "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
          "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
          "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b), "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
  const int blocks = 256 * 8, iters = 20000;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const struct { const char* name; unsigned long long mask; } cases[] = {
    { "all 64 lanes      ", ~0ull }, { "lanes 0-31        ", 0xFFFFFFFFull }, { "lanes 32-63       ", 0xFFFFFFFF00000000ull },
    { "even lanes        ", 0x5555555555555555ull }, { "lanes 0-15        ", 0xFFFFull }, { "lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull },
    { "lane 0            ", 1ull } };
  for (int rep = 0; rep < 2; ++rep)
    for (const auto& c : cases) {
      hipLaunchKernelGGL(fma_stream, dim3(blocks), dim3(256), 0, 0, out, c.mask, 100);
      hipEventRecord(e0);
      hipLaunchKernelGGL(fma_stream, dim3(blocks), dim3(256), 0, 0, out, c.mask, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)blocks * 4 * iters * 16;
      if (rep) printf("%s  %.3f ms  %.1f G wave-inst/s  (%.2f cycles per wave-inst per SIMD at 2.4 GHz)\n", c.name, ms,
                      insts / ms / 1e6, 1024.0 * 2.4e9 * ms * 1e-3 / insts);
    }
  return 0;
}
