// Micro-benchmark for the wide walk's node fetch: 64 lanes each need a different 128-byte record of a 32 MB table.
//   A  per-lane fetch: every lane issues 8 x global_load_dwordx4 for its own record (64 lines per instruction)
//   B  cooperative fetch: instruction i loads the records of lanes 8i..8i+7, lane l taking 16-byte chunk l & 7
//      (8 lines per instruction, 128 contiguous bytes per 8 lanes); the data would then go through LDS to its owner
//   C  per-lane fetch of a 64-byte record (4 loads): what a quantised node would cost
// Same bytes per wave in A and B.  Prints GB/s and records/us.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/gather_pattern.hip -o build/gather_pattern && build/gather_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ void __launch_bounds__(256) gather(const float4* table, uint32_t n_records, float* out, int iters)
{
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint32_t rec = lcg(seed) % n_records;      // this lane's record
    if (MODE == 0) {
      const float4* q = table + (size_t)rec * 8u;
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = q[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    } else if (MODE == 1) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t owner = 8u * i + (lane >> 3);
        const uint32_t r = (uint32_t)__shfl((int)rec, (int)owner, 64);
        v[i] = table[(size_t)r * 8u + (lane & 7u)];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    } else {
      const float4* q = table + (size_t)rec * 8u;
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = q[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].w;
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main()
{
  const uint32_t n_records = 262144;   // 32 MB of 128-byte records
  std::vector<float> h((size_t)n_records * 32, 1.0f);
  float4* table; float* out;
  hipMalloc(&table, h.size() * 4); hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int blocks = 256 * 4, iters = 2000;   // 16 waves per CU
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_records, out, iters);
      if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_records, out, iters);
      if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, n_records, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double recs = (double)blocks * 256 * iters;
      const double bytes = recs * (mode == 2 ? 64 : 128);
      if (rep) printf("%s  %.3f ms  %.1f records/us chip-wide  %.2f TB/s\n",
                      mode == 0 ? "A per-lane 8 x 16 B      " : (mode == 1 ? "B cooperative 8 x 16 B   " : "C per-lane 4 x 16 B (64 B)"), ms, recs / ms / 1e3, bytes / ms / 1e9);
    }
  }
  return 0;
}
