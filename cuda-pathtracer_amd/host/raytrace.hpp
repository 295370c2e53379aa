// raytrace.hpp — C++ host mirror of the reference's render boundary over the C-ABI.
//
// The reference declares (cuda_opengl/include/shaders/raytrace.h:9-17):
//
//   cudaError_t raytrace(cudaArray_const_t array, const scene::Scenes& scenes, unsigned int scene_id,
//                        const std::vector<scene::Cubemap>& cubemaps, int cubemap_id,
//                        const scene::Camera* const cam, const unsigned int width,
//                        const unsigned int height, cudaStream_t stream,
//                        float3* temporal_framebuffer, bool moved, unsigned int post_id);
//   void setupFunctionTables();
//
// Same names, same argument order and meaning.  What changes for a caller:
//   * `array` is a linear RGBA8 device buffer (width*height*4 bytes, row 0 = top) instead of
//     the cudaArray of a GL renderbuffer;
//   * `scenes` / `cubemaps` are the context that owns the uploaded tables (ptamd_upload_*),
//     where the reference passes the device pointer graph GPUProcessor built;
//   * the return value is a ptamd_status (0 = ok, like cudaSuccess) and the message is
//     available from ptamd_get_last_error(); like the reference the launch is asynchronous.
// Header-only; link with -lptamd.  No HIP headers are needed by the caller.
#pragma once

#include "ptamd.h"

namespace ptamd_host {

struct Scenes { ptamd_context* ctx; };   // scene::Scenes: device scene tables (owned by ctx)
struct Cubemaps { ptamd_context* ctx; }; // std::vector<scene::Cubemap>

inline int raytrace(void* array, const Scenes& scenes, unsigned int scene_id, const Cubemaps& cubemaps,
                    int cubemap_id, const ptamd_camera* const cam, const unsigned int width,
                    const unsigned int height, void* stream, float* temporal_framebuffer, bool moved,
                    unsigned int post_id)
{
  (void)cubemaps;
  return ptamd_raytrace(scenes.ctx, array, scene_id, (uint32_t)cubemap_id, cam, width, height, stream,
                        temporal_framebuffer, moved ? 1 : 0, post_id);
}

inline int setupFunctionTables(ptamd_context* ctx) { return ptamd_setup_function_tables(ctx); }

} // namespace ptamd_host
