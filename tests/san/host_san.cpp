// host_san.cpp — sanitizer harness of the library's HOST code (scene / OBJ / MTL parser, JPEG / PNG / HDR decoders, PNG writer,
// resampler, BVH builder and its four host walks).  Test infrastructure: built by `make san` with g++ -fsanitize=address,undefined
// from the same sources the product library is built from (the device half, csrc/, is not part of it: GPU sanitizers are not
// available on the pool) and run by tests/test_sanitizers.py.
//
//   host_san <assets dir> <scratch dir> <iterations> <seed>
//
// 1. every shipped asset through every loader entry point, every scene's faces through the builder and the four walks;
// 2. `iterations` damaged copies of every input file class (bit flips, truncation, doubled chunks, hostile numbers in the text
//    formats): the loaders may refuse a file, they may not read or write out of bounds, leak, or overflow.
// Exit code 0: no finding (ASan / UBSan abort the process otherwise).
#include "ptamd.h"
#include "ptamd_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <random>
#include <sstream>
#include <string>
#include <vector>

namespace ptamd {
// the two services the host sources take from csrc/ptamd_api.cpp
static std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* tuning_env(const char*) { return nullptr; }
} // namespace ptamd
extern "C" const char* ptamd_get_last_error(void) { return ptamd::g_err.c_str(); }

static std::vector<uint8_t> slurp(const std::string& path)
{
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const std::vector<uint8_t>& b)
{
  std::ofstream f(path, std::ios::binary | std::ios::trunc);
  f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
}

static unsigned long g_loaded = 0, g_refused = 0;

// one image file through both decoders, the resampler and the cube-cross splitter
static void exercise_image(const std::string& path, bool heavy)
{
  int32_t w = 0, h = 0, c = 0;
  uint8_t* px8 = nullptr;
  if (ptamd_image_load8(path.c_str(), &w, &h, &c, &px8) == PTAMD_OK) {
    ++g_loaded;
    volatile uint8_t sink = px8[(size_t)w * h * c - 1];   // the last byte the decoder promises
    (void)sink;
    ptamd_image_free(px8);
  } else ++g_refused;
  float* pxf = nullptr;
  if (ptamd_image_loadf(path.c_str(), &w, &h, &c, &pxf) == PTAMD_OK) {
    volatile float sink = pxf[(size_t)w * h * c - 1];
    (void)sink;
    if (heavy && w > 0 && h > 0) {
      std::vector<float> out((size_t)37 * 23 * c);
      ptamd_image_resize_float(pxf, w, h, out.data(), 37, 23, c);
      if (w % 4 == 0 && h % 3 == 0 && w / 4 == h / 3 && c >= 3) {
        const uint32_t size = (uint32_t)w / 4u;
        std::vector<float> faces((size_t)6 * size * size * 4);
        uint32_t out_size = 0;
        ptamd_cubemap_from_cross(pxf, (uint32_t)w, (uint32_t)h, (uint32_t)c, faces.data(), &out_size);
      }
    }
    ptamd_image_free(pxf);
  }
}

// one .scene through the loader; optionally its faces through the builder and the three host walks
static void exercise_scene(const std::string& path, uint32_t flags, bool walk, std::mt19937& rng)
{
  ptamd_host_scene* hs = nullptr;
  if (ptamd_host_scene_load(path.c_str(), flags, &hs) != PTAMD_OK) { ++g_refused; return; }
  ++g_loaded;
  ptamd_scene_desc d;
  ptamd_camera cam;
  ptamd_host_scene_desc(hs, &d);
  ptamd_host_scene_camera(hs, &cam);
  (void)ptamd_host_scene_cubemap(hs);
  for (uint32_t i = 0; i < ptamd_host_scene_unloaded_count(hs); ++i) (void)ptamd_host_scene_unloaded_name(hs, i);
  // touch everything the descriptor points at
  uint64_t sum = 0;
  for (uint32_t i = 0; i < d.n_faces; ++i) sum += reinterpret_cast<const uint8_t*>(d.faces + i)[sizeof(ptamd_face) - 1];
  for (uint32_t i = 0; i < d.n_materials; ++i) sum += reinterpret_cast<const uint8_t*>(d.materials + i)[sizeof(ptamd_material) - 1];
  for (uint32_t i = 0; i < d.n_lights; ++i) sum += reinterpret_cast<const uint8_t*>(d.lights + i)[sizeof(ptamd_light) - 1];
  for (uint32_t i = 0; i < d.n_textures; ++i) sum += reinterpret_cast<const uint8_t*>(d.textures + i)[sizeof(ptamd_texture_desc) - 1];
  if (d.n_texel_floats) sum += (uint64_t)(d.texels[d.n_texel_floats - 1] != 0.0f);
  volatile uint64_t sink = sum;
  (void)sink;
  if (walk && d.n_faces) {
    const uint32_t n = 1500;
    std::vector<float> rays((size_t)n * 6);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    for (uint32_t i = 0; i < n; ++i) {
      float* r = &rays[(size_t)i * 6];
      r[0] = u(rng); r[1] = u(rng); r[2] = u(rng);
      if (i % 97 == 0) r[i % 3] = 0.0f;           // axis-parallel directions (infinite reciprocals)
      if (i % 193 == 0) r[(i + 1) % 3] = -0.0f;
      r[3] = cam.position.x + 3.0f * u(rng); r[4] = cam.position.y + 3.0f * u(rng); r[5] = cam.position.z + 3.0f * u(rng);
    }
    std::vector<int32_t> a((size_t)n * 4), b((size_t)n * 4), c((size_t)n * 4), e((size_t)n * 4);
    uint64_t ca[2] = {0, 0}, cb[5] = {0, 0, 0, 0, 0}, cc[6] = {0, 0, 0, 0, 0, 0}, ce[5] = {0, 0, 0, 0, 0};
    if (ptamd_host_bvh_trace(d.faces, d.n_faces, rays.data(), n, a.data(), ca) != PTAMD_OK ||
        ptamd_host_bvh4_trace(d.faces, d.n_faces, rays.data(), n, b.data(), cb) != PTAMD_OK ||
        ptamd_host_bvh4q_trace(d.faces, d.n_faces, rays.data(), n, e.data(), ce) != PTAMD_OK ||
        ptamd_host_bvh8_trace(d.faces, d.n_faces, rays.data(), n, c.data(), cc) != PTAMD_OK) {
      std::fprintf(stderr, "host walk failed on %s: %s\n", path.c_str(), ptamd_get_last_error());
      std::exit(3);
    }
    if (std::memcmp(a.data(), b.data(), a.size() * 4) != 0 || std::memcmp(a.data(), c.data(), a.size() * 4) != 0 ||
        std::memcmp(a.data(), e.data(), a.size() * 4) != 0) {
      std::fprintf(stderr, "the four host walks disagree on %s\n", path.c_str());
      std::exit(4);
    }
  }
  ptamd_host_scene_free(hs);
}

// ---- damage
static void damage_binary(std::vector<uint8_t>& b, std::mt19937& rng)
{
  if (b.empty()) return;
  switch (rng() % 6u) {
  case 0: { const uint32_t k = 1u + rng() % 8u; for (uint32_t i = 0; i < k; ++i) b[rng() % b.size()] ^= (uint8_t)(1u << (rng() % 8u)); } break;
  case 1: { const uint32_t k = 1u + rng() % 16u; for (uint32_t i = 0; i < k; ++i) b[rng() % b.size()] = (uint8_t)rng(); } break;
  case 2: b.resize(rng() % b.size()); break;                                                  // truncated
  case 3: { const size_t at = rng() % b.size(), n = 1u + rng() % 64u;                          // a run of one byte value
            const uint8_t v = (rng() & 1u) ? 0xFFu : (uint8_t)0u; for (size_t i = at; i < at + n && i < b.size(); ++i) b[i] = v; } break;
  case 4: { const size_t at = rng() % b.size(), n = 1u + rng() % 256u;                         // a chunk doubled
            std::vector<uint8_t> chunk(b.begin() + (long)at, b.begin() + (long)std::min(b.size(), at + n));
            b.insert(b.begin() + (long)at, chunk.begin(), chunk.end()); } break;
  default: { const size_t at = rng() % b.size(), n = 1u + rng() % 256u;                        // a chunk removed
             b.erase(b.begin() + (long)at, b.begin() + (long)std::min(b.size(), at + n)); } break;
  }
  // headers are where the sizes live: half of the time also hit the first 1 KiB
  if ((rng() & 1u) && !b.empty()) b[rng() % std::min<size_t>(b.size(), 1024u)] = (uint8_t)rng();
}

static void damage_text(std::vector<uint8_t>& b, std::mt19937& rng)
{
  if (b.empty()) return;
  static const char* hostile[] = { "99999999999", "-1", "0", "4294967295", "4294967296", "-2147483649", "nan", "inf", "-inf", "1e39", "-1e39",
                                   "1e-46", "", " ", "//", "\\", "../../../../etc/passwd", "%s%s%n", "0x7fffffff", "1/2/3/4/5", "//3", "1//", "-0" };
  std::string s(b.begin(), b.end());
  switch (rng() % 5u) {
  case 0: {   // one number replaced
    std::vector<std::pair<size_t, size_t>> runs;
    for (size_t i = 0; i < s.size();) {
      if (std::isdigit((unsigned char)s[i]) || ((s[i] == '-' || s[i] == '.') && i + 1 < s.size() && std::isdigit((unsigned char)s[i + 1]))) {
        size_t j = i + 1;
        while (j < s.size() && (std::isdigit((unsigned char)s[j]) || s[j] == '.' || s[j] == 'e' || s[j] == '-')) ++j;
        runs.push_back({i, j - i});
        i = j;
      } else ++i;
    }
    if (!runs.empty()) { const auto r = runs[rng() % runs.size()]; s.replace(r.first, r.second, hostile[rng() % (sizeof(hostile) / sizeof(*hostile))]); }
  } break;
  case 1: {   // one line removed / doubled / cut short
    std::vector<size_t> starts{0};
    for (size_t i = 0; i + 1 < s.size(); ++i) if (s[i] == '\n') starts.push_back(i + 1);
    const size_t k = rng() % starts.size(), a = starts[k], e = k + 1 < starts.size() ? starts[k + 1] : s.size();
    const uint32_t how = rng() % 3u;
    if (how == 0) s.erase(a, e - a);
    else if (how == 1) s.insert(a, s.substr(a, e - a));
    else if (e - a > 2) s.erase(a + 1 + rng() % (e - a - 1), std::string::npos), s += "\n";
  } break;
  case 2: s.resize(rng() % s.size()); break;
  case 3: { const uint32_t k = 1u + rng() % 6u; for (uint32_t i = 0; i < k; ++i) s[rng() % s.size()] = (char)(rng() % 256u); } break;
  default: {  // a very long token
    const size_t at = rng() % s.size();
    s.insert(at, std::string(1u + rng() % 5000u, "9a /\\"[rng() % 5u]));
  } break;
  }
  b.assign(s.begin(), s.end());
}

int main(int argc, char** argv)
{
  if (argc < 5) { std::fprintf(stderr, "usage: host_san <assets> <scratch> <iterations> <seed>\n"); return 2; }
  const std::string assets = argv[1], scratch = argv[2];
  const int iterations = std::atoi(argv[3]);
  std::mt19937 rng((uint32_t)std::strtoul(argv[4], nullptr, 10));

  const char* scenes[] = { "color_sample", "cornell", "crate_land", "indoor", "island", "sss_crate" };
  const char* images[] = { "obj/textures/brickwall.jpg", "obj/textures/crack2.jpg", "obj/textures/metal_crate/albedo.jpg", "obj/textures/metal_crate/specular.jpg",
                           "obj/textures/water/normal.jpg", "obj/textures/wooden_planck/albedo_2.jpg", "obj/textures/parquet/normal_1.jpg",
                           "cubemap/field_with_house.jpg" };

  // ---- 1. the shipped inputs, whole
  for (const char* s : scenes)
    for (uint32_t flags = 0; flags < 4u; ++flags) exercise_scene(assets + "/" + s + ".scene", flags, flags == 2u, rng);
  for (const char* i : images) exercise_image(assets + "/" + i, true);
  // PNG: written by the library's own encoder, read back by its decoder (1..4 channels, odd sizes)
  std::vector<std::string> pngs;
  for (int c = 1; c <= 4; ++c) {
    const int w = 19 + 13 * c, h = 7 + 5 * c;
    std::vector<uint8_t> px((size_t)w * h * c);
    for (auto& v : px) v = (uint8_t)rng();
    const std::string path = scratch + "/san_" + std::to_string(c) + ".png";
    if (ptamd_image_save_png(path.c_str(), px.data(), w, h, c) != PTAMD_OK) { std::fprintf(stderr, "png save failed\n"); return 5; }
    int32_t rw = 0, rh = 0, rc = 0;
    uint8_t* back = nullptr;
    if (ptamd_image_load8(path.c_str(), &rw, &rh, &rc, &back) != PTAMD_OK || rw != w || rh != h || rc != c || std::memcmp(back, px.data(), px.size()) != 0) {
      std::fprintf(stderr, "png round trip failed (%d channels)\n", c);
      return 6;
    }
    ptamd_image_free(back);
    pngs.push_back(path);
  }
  const unsigned long whole_loaded = g_loaded, whole_refused = g_refused;

  // ---- 2. damaged copies.  The scene files name their OBJ / MTL / textures by relative path: the scratch directory holds a
  //         copy of the text inputs (obj/ below it), images are damaged one file at a time on their own.
  const std::string sdir = scratch + "/fuzz";
  if (std::system(("mkdir -p '" + sdir + "/obj'").c_str()) != 0) return 7;
  struct Text { std::string rel; std::vector<uint8_t> bytes; };
  std::vector<Text> texts;
  for (const char* s : scenes) {
    texts.push_back({ std::string(s) + ".scene", slurp(assets + "/" + s + ".scene") });
    if (std::string(s) != "cornell") {
      texts.push_back({ std::string("obj/") + s + ".obj", slurp(assets + "/obj/" + s + ".obj") });
      texts.push_back({ std::string("obj/") + s + ".mtl", slurp(assets + "/obj/" + s + ".mtl") });
    }
  }
  for (const Text& t : texts) spit(sdir + "/" + t.rel, t.bytes);
  for (int it = 0; it < iterations; ++it) {
    // text inputs: one file of one scene damaged, the scene loaded without images (the image decoders get their own turns)
    const Text& t = texts[rng() % texts.size()];
    std::vector<uint8_t> b = t.bytes;
    damage_text(b, rng);
    if (rng() % 4u == 0u) damage_binary(b, rng);
    spit(sdir + "/" + t.rel, b);
    std::string base = t.rel.substr(t.rel.find('/') == std::string::npos ? 0 : 4);
    base = base.substr(0, base.rfind('.'));
    exercise_scene(sdir + "/" + base + ".scene", PTAMD_LOAD_NO_IMAGES | (rng() & 1u), it % 16 == 0, rng);
    spit(sdir + "/" + t.rel, t.bytes);
  }
  {
    std::vector<std::vector<uint8_t>> originals;
    std::vector<std::string> ext;
    const char* small[] = { "obj/textures/crack2.jpg", "obj/textures/metal_crate/albedo.jpg", "obj/textures/water/normal.jpg" };
    for (const char* i : small) { originals.push_back(slurp(assets + "/" + i)); ext.push_back(".jpg"); }
    for (const std::string& p : pngs) { originals.push_back(slurp(p)); ext.push_back(".png"); }
    // a Radiance picture: header + flat RGBE pixels (the loader's third format)
    {
      std::string h = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 9 +X 11\n";
      std::vector<uint8_t> b(h.begin(), h.end());
      for (int i = 0; i < 9 * 11 * 4; ++i) b.push_back((uint8_t)(i % 4 == 3 ? 128 + (int)(rng() % 8u) : rng()));
      originals.push_back(b); ext.push_back(".hdr");
    }
    for (int it = 0; it < iterations; ++it) {
      const size_t k = rng() % originals.size();
      std::vector<uint8_t> b = originals[k];
      damage_binary(b, rng);
      if (rng() % 3u == 0u) damage_binary(b, rng);
      const std::string path = sdir + "/damaged" + ext[k];
      spit(path, b);
      exercise_image(path, it % 8 == 0);
    }
  }
  // ---- 3. hostile geometry straight into the builder: a soup of triangles with one coordinate far out, infinite or NaN (such a
  //         scene has boxes without a finite centre: round 3's slot assignment of the eight-wide nodes read an unset table entry
  //         for them), degenerate faces, duplicates; every leaf size.  The four walks must agree on every ray.
  {
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const float bad[] = { 3.0e9f, -3.0e9f, std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(),
                          std::numeric_limits<float>::quiet_NaN(), 3.0e38f, -3.0e38f, 1.0e-30f, -9.0e7f, 9.0e7f };
    for (int round = 0; round < 24; ++round) {
      const uint32_t n = 20u + (uint32_t)(rng() % 120u);
      std::vector<ptamd_face> faces(n);
      std::memset(faces.data(), 0, n * sizeof(ptamd_face));
      for (uint32_t i = 0; i < n; ++i) {
        const float cx = 1.2f * u(rng), cy = 1.2f * u(rng), cz = 1.2f * u(rng);
        for (int k = 0; k < 3; ++k) { faces[i].vertices[k].x = cx + 0.9f * u(rng); faces[i].vertices[k].y = cy + 0.9f * u(rng); faces[i].vertices[k].z = cz + 0.9f * u(rng); }
      }
      const uint32_t n_bad = 1u + (uint32_t)(rng() % 4u);
      for (uint32_t k = 0; k < n_bad; ++k) (&faces[rng() % n].vertices[rng() % 3u].x)[rng() % 3u] = bad[rng() % (sizeof bad / sizeof *bad)];
      if (round % 3 == 0) faces[rng() % n] = faces[rng() % n];                       // a duplicate
      if (round % 4 == 0) faces[rng() % n].vertices[2] = faces[rng() % n].vertices[1];   // (nearly) degenerate
      if (round % 5 == 0) for (int k = 0; k < 3; ++k) { ptamd_float3& v = faces[1].vertices[k]; v.x *= 1.0e9f; v.y *= 1.0e9f; v.z *= 1.0e9f; }
      for (uint32_t max_leaf : { 1u, 3u, 4u, 15u }) {
        ptamd::Bvh bvh;
        if (ptamd::build_bvh(faces.data(), n, 1e-3f, max_leaf, bvh) != PTAMD_OK) { std::fprintf(stderr, "build_bvh refused a soup\n"); return 8; }
      }
      const uint32_t nr = 400;
      std::vector<float> rays((size_t)nr * 6);
      for (auto& v : rays) v = 2.0f * u(rng);
      std::vector<int32_t> a((size_t)nr * 4), b4((size_t)nr * 4), q4((size_t)nr * 4), b8((size_t)nr * 4);
      if (ptamd_host_bvh_trace(faces.data(), n, rays.data(), nr, a.data(), nullptr) != PTAMD_OK ||
          ptamd_host_bvh4_trace(faces.data(), n, rays.data(), nr, b4.data(), nullptr) != PTAMD_OK ||
          ptamd_host_bvh4q_trace(faces.data(), n, rays.data(), nr, q4.data(), nullptr) != PTAMD_OK ||
          ptamd_host_bvh8_trace(faces.data(), n, rays.data(), nr, b8.data(), nullptr) != PTAMD_OK) return 9;
      // (the quantised forms are for finite coordinates within +-1e8: the library walks the float nodes beyond that)
      float extent = 0.0f;
      for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
          for (int ax = 0; ax < 3; ++ax) {
            const float v = std::fabs((&faces[i].vertices[k].x)[ax]);
            if (v <= std::numeric_limits<float>::max()) extent = std::max(extent, v);
          }
      const bool quantised_ok = extent <= 1.0e8f;
      if (a != b4 || (quantised_ok && (a != q4 || a != b8))) { std::fprintf(stderr, "the host walks disagree on a hostile soup (round %d)\n", round); return 10; }
    }
  }
  std::printf("host_san: shipped inputs %lu loaded / %lu refused; damaged inputs %lu loaded / %lu refused; no sanitizer finding\n",
              whole_loaded, whole_refused, g_loaded - whole_loaded, g_refused - whole_refused);
  return 0;
}
