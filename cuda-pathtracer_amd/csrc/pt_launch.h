// pt_launch.h — host-callable launchers of the device code in pt_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptamd {

struct KParams;

// kind: 1 brute force, 2 BVH.  lds_bytes: dynamic LDS needed when lds_resident.
hipError_t launch_megakernel(const KParams& p, int kind, bool lds_resident, size_t lds_bytes, bool stats,
                             hipStream_t stream);
hipError_t persistent_blocks_per_cu(bool lds_resident, size_t lds_bytes, int* out);
hipError_t launch_megakernel_persistent(const KParams& p, bool lds_resident, size_t lds_bytes, bool stats,
                                        uint32_t n_blocks, hipStream_t stream);
hipError_t blockwise_blocks_per_cu(bool lds_resident, size_t scene_lds_bytes, int* out);
hipError_t launch_megakernel_blockwise(const KParams& p, bool lds_resident, size_t scene_lds_bytes, bool stats,
                                       uint32_t n_blocks, hipStream_t stream);
uint32_t split_shader_waves();
hipError_t split_blocks_per_cu(bool lds_resident, size_t scene_lds_bytes, int* out);
hipError_t launch_megakernel_split(const KParams& p, bool lds_resident, size_t scene_lds_bytes, bool stats,
                                   uint32_t n_blocks, hipStream_t stream);
uint32_t restart_threads(bool lds_resident);
uint32_t restart_treelet_region_bytes();   // != 0: the wide walk's LDS treelet is chunk-major in a region of this size (pt_kernels.hip: PT_TREELET_SOA)
uint32_t restart_wide_blocks_per_cu();
hipError_t restart_blocks_per_cu(bool lds_resident, size_t lds_bytes, int* out);
hipError_t launch_megakernel_restart(const KParams& p, bool lds_resident, size_t lds_bytes, bool stats,
                                     uint32_t n_blocks, hipStream_t stream);
hipError_t launch_resolve(const KParams& p, hipStream_t stream);
// gamma step of the tonemap as a table (pt_kernels.hip: gamma_byte): 258 floats, and its exhaustive check
hipError_t build_gamma_table(float* table_dev, hipStream_t stream);
hipError_t launch_gamma_selftest(const float* table_dev, uint32_t first, uint32_t count, unsigned long long* out_dev, hipStream_t stream);
// Resolves every kernel entry point of the code object (setupFunctionTables' role: fail early when the device image is unusable).
hipError_t resolve_kernels();
// walk-only kernel fed from a ray queue (pt_kernels.hip: pt_trace_queue_kernel): shape of a configuration, launch
void trace_queue_shape(uint32_t config, uint32_t* threads, uint32_t* plane_nodes, uint32_t* blocks_per_cu);
hipError_t launch_trace_queue(const KParams& p, uint32_t config, size_t lds_bytes, uint32_t n_blocks, const float* rays_dev, uint32_t n, int4* out_dev,
                              uint32_t* head_dev, int* resident_blocks_per_cu, hipStream_t stream);
hipError_t launch_trace_rays(const KParams& p, int kind, const float* rays_dev, uint32_t n, int4* out_dev,
                             hipStream_t stream);

} // namespace ptamd
