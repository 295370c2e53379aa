// headless_render.cpp — minimal C++ host over the C-ABI: what GPUProcessor::init()/render()
// (cuda_opengl/src/gpu_processor.cpp:271-386) do for the path, without a window.
//   g++ -std=c++17 -Iinclude -Icuda-pathtracer_amd/host examples/headless_render.cpp
//       -Lcuda-pathtracer_amd -lptamd -Wl,-rpath,$PWD/cuda-pathtracer_amd -o headless_render
//   ./headless_render assets/crate_land.scene 960 540 64 out.png      (.png or .ppm)
#include "interop.hpp"
#include "raytrace.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != PTAMD_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ptamd_get_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s SCENE WIDTH HEIGHT FRAMES OUT.(png|ppm)\n", argv[0]); return 2; }
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
  // uploadCubemap (gpu_processor.cpp:68-161): decode <asset folder>/<cubemap name>, cut the cross into six faces;
  // anything that fails ends in the 1x1 cubemap of colour 0x131b23
  std::vector<float> cube(24);
  uint32_t cube_size = 1;
  CHECK(ptamd_cubemap_from_color(0x131b23, cube.data()));
  {
    const std::string scene_path = argv[1], name = ptamd_host_scene_cubemap(hs);
    const size_t slash = scene_path.find_last_of('/');
    const std::string folder = slash == std::string::npos ? "." : scene_path.substr(0, slash);
    int32_t cw = 0, ch = 0, cc = 0;
    float* cross = nullptr;
    if (!name.empty() && ptamd_image_loadf((folder + "/" + name).c_str(), &cw, &ch, &cc, &cross) == PTAMD_OK) {
      std::vector<float> faces((size_t)6 * (cw / 4) * (cw / 4) * 4 + 24);
      uint32_t size = 0;
      if (ptamd_cubemap_from_cross(cross, (uint32_t)cw, (uint32_t)ch, (uint32_t)cc, faces.data(), &size) == PTAMD_OK) {
        cube.swap(faces);
        cube_size = size;
      }
      ptamd_image_free(cross);
    }
  }
  CHECK(ptamd_upload_cubemap(ctx, cube.data(), cube_size, &cubemap_id));
  CHECK(ptamd_host::setupFunctionTables(ctx));
  // driver::Interop stand-in: two RGBA8 surfaces; blit() hands the finished frame to the presenter (here: keep the last one)
  std::vector<unsigned char> px;
  ptamd_host::Interop interop(ctx, w, h, [&](const uint8_t* p, unsigned iw, unsigned ih) { px.assign(p, p + (size_t)iw * ih * 4); });
  void* tfb = nullptr;
  CHECK(ptamd_device_alloc(ctx, (size_t)w * h * 12, &tfb));
  CHECK(ptamd_device_memset(ctx, tfb, 0, (size_t)w * h * 12, nullptr)); // cudaCalloc, gpu_processor.cpp:255
  ptamd_host::Scenes scenes{ ctx };
  ptamd_host::Cubemaps cubemaps{ ctx };
  for (int f = 0; f < frames; ++f) { // main.cpp:172-206 without the window
    // GPUProcessor::render(), gpu_processor.cpp:365-386
    CHECK(interop.map(nullptr));
    CHECK(ptamd_host::raytrace(interop.getArray(), scenes, scene_id, cubemaps, (int)cubemap_id, &cam, interop.width(),
                               interop.height(), nullptr, static_cast<float*>(tfb), false, 0));
    CHECK(interop.unmap(nullptr));
    if (f + 1 == frames) CHECK(interop.blit(nullptr));   // main.cpp:198 blits every frame; one presentation is enough here
    interop.swap();                                        // main.cpp:200
  }
  std::vector<unsigned char> rgb((size_t)w * h * 3);      // the surface's alpha is 0 (raytrace.cu:232): write RGB
  for (size_t i = 0; i < (size_t)w * h; ++i) std::memcpy(&rgb[i * 3], &px[i * 4], 3);
  const size_t len = std::strlen(argv[5]);
  if (len > 4 && std::strcmp(argv[5] + len - 4, ".png") == 0) {
    CHECK(ptamd_image_save_png(argv[5], rgb.data(), (int32_t)w, (int32_t)h, 3));
  } else {
    FILE* out = std::fopen(argv[5], "wb");
    if (!out) return 1;
    std::fprintf(out, "P6 %u %u 255\n", w, h);
    std::fwrite(rgb.data(), 1, rgb.size(), out);
    std::fclose(out);
  }
  interop.clean();
  ptamd_device_free(ctx, tfb);
  ptamd_destroy(ctx);
  ptamd_host_scene_free(hs);
  return 0;
}
