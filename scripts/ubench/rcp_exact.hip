// Exhaustive check on the device: for every binary32 x in [1e-7f, 2^125] (both signs of nothing: det is positive),
// the 7-instruction sequence  v_rcp_f32 + 6 fma  equals the correctly rounded IEEE quotient 1.0f / x bit for bit.
// (It is the compiler's own division sequence minus v_div_scale / v_div_fmas' scaling / v_div_fixup, which are the
// identity for a numerator of 1 and a denominator in that range.)  Prints the number of mismatches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/rcp_exact.hip -o build/rcp_exact && build/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp7(float x)
{
  const float y0 = __builtin_amdgcn_rcpf(x);
  const float e0 = __builtin_fmaf(-x, y0, 1.0f);
  const float y1 = __builtin_fmaf(e0, y0, y0);
  const float r1 = __builtin_fmaf(-x, y1, 1.0f);
  const float q1 = __builtin_fmaf(r1, y1, y1);
  const float r2 = __builtin_fmaf(-x, q1, 1.0f);
  return __builtin_fmaf(r2, y1, q1);
}

__global__ void check(uint32_t first, uint32_t count, unsigned long long* bad, uint32_t* example)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t bits = first + i;
  const float x = __uint_as_float(bits);
  const float want = 1.0f / x;
  const float got = rcp7(x);
  if (__float_as_uint(want) != __float_as_uint(got)) {
    if (atomicAdd(bad, 1ull) == 0ull) *example = bits;
  }
}

int main()
{
  float lo = 1e-7f, hi = 4.2535296e37f;   // 2^125
  uint32_t blo, bhi;
  std::memcpy(&blo, &lo, 4); std::memcpy(&bhi, &hi, 4);
  unsigned long long* bad; uint32_t* ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 4); hipMemset(bad, 0, 8); hipMemset(ex, 0, 4);
  const uint32_t total = bhi - blo + 1u;
  for (uint32_t off = 0; off < total; off += 1u << 28) {
    const uint32_t n = total - off < (1u << 28) ? total - off : (1u << 28);
    hipLaunchKernelGGL(check, dim3((n + 255u) / 256u), dim3(256), 0, 0, blo + off, n, bad, ex);
  }
  hipDeviceSynchronize();
  unsigned long long h = 0; uint32_t e = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&e, ex, 4, hipMemcpyDeviceToHost);
  printf("checked %u values in [1e-7, 2^125]: %llu mismatches%s", total, h, h ? "" : "\n");
  if (h) printf(" (first: bits 0x%08x)\n", e);
  return h ? 1 : 0;
}
