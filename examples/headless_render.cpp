// headless_render.cpp — minimal C++ host over the C-ABI: what GPUProcessor::init()/render()
// (cuda_opengl/src/gpu_processor.cpp:271-386) do for the path, without a window.
//   g++ -std=c++17 -Iinclude -Icuda-pathtracer_amd/host examples/headless_render.cpp
//       -Lcuda-pathtracer_amd -lptamd -Wl,-rpath,$PWD/cuda-pathtracer_amd -o headless_render
//   ./headless_render assets/indoor.scene 960 540 64 out.ppm
#include "raytrace.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != PTAMD_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ptamd_get_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s SCENE WIDTH HEIGHT FRAMES OUT.ppm\n", argv[0]); return 2; }
  const unsigned w = (unsigned)std::atoi(argv[2]), h = (unsigned)std::atoi(argv[3]);
  const int frames = std::atoi(argv[4]);
  ptamd_host_scene* hs = nullptr;
  CHECK(ptamd_host_scene_load(argv[1], 0, &hs));
  ptamd_scene_desc desc;
  ptamd_camera cam;
  CHECK(ptamd_host_scene_desc(hs, &desc));
  CHECK(ptamd_host_scene_camera(hs, &cam));
  ptamd_context* ctx = nullptr;
  CHECK(ptamd_create(0, &ctx));
  uint32_t scene_id = 0, cubemap_id = 0;
  CHECK(ptamd_upload_scene(ctx, &desc, &scene_id));
  float cube[24];
  CHECK(ptamd_cubemap_from_color(0x131b23, cube)); // gpu_processor.cpp:75,128-132
  CHECK(ptamd_upload_cubemap(ctx, cube, 1, &cubemap_id));
  CHECK(ptamd_host::setupFunctionTables(ctx));
  void *surface = nullptr, *tfb = nullptr;
  CHECK(ptamd_device_alloc(ctx, (size_t)w * h * 4, &surface));
  CHECK(ptamd_device_alloc(ctx, (size_t)w * h * 12, &tfb));
  CHECK(ptamd_device_memset(ctx, tfb, 0, (size_t)w * h * 12, nullptr)); // cudaCalloc, gpu_processor.cpp:255
  ptamd_host::Scenes scenes{ ctx };
  ptamd_host::Cubemaps cubemaps{ ctx };
  for (int f = 0; f < frames; ++f) // main.cpp:172-206 without the window
    CHECK(ptamd_host::raytrace(surface, scenes, scene_id, cubemaps, (int)cubemap_id, &cam, w, h, nullptr,
                               static_cast<float*>(tfb), false, 0));
  std::vector<unsigned char> px((size_t)w * h * 4);
  CHECK(ptamd_device_to_host(ctx, px.data(), surface, px.size(), nullptr));
  FILE* out = std::fopen(argv[5], "wb");
  if (!out) return 1;
  std::fprintf(out, "P6 %u %u 255\n", w, h);
  for (size_t i = 0; i < (size_t)w * h; ++i) std::fwrite(&px[i * 4], 1, 3, out);
  std::fclose(out);
  ptamd_device_free(ctx, surface);
  ptamd_device_free(ctx, tfb);
  ptamd_destroy(ctx);
  ptamd_host_scene_free(hs);
  return 0;
}
