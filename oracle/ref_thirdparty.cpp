// ref_thirdparty.cpp — TEST INFRASTRUCTURE ONLY (oracle/_ref).
//
// A C wrapper around the third-party libraries the REFERENCE itself vendors and calls on the
// input side of the hot path:
//   tinyobj::LoadObj            tiny_obj_loader.h 1.0.8   (called at scene.cpp:341)
//   stbi_loadf / stbi_load      stb_image.h 2.16          (material_loader.cpp:97, gpu_processor.cpp:99)
//   stbir_resize_float          stb_image_resize.h 0.95   (material_loader.cpp:358,367)
// The headers are compiled WHERE THEY LIE under /root/reference (-I, see oracle/Makefile); nothing
// of them is copied into this repository and the output goes to oracle/_ref/ only (git-ignored).
// They are single-header libraries, so a translation unit that defines the *_IMPLEMENTATION macros
// is how the reference itself builds them (scene.cpp:23, gpu_processor.cpp:9, material_loader.cpp:8).
//
// Used by tests/golden/make_ref_golden.py to generate fixtures and by tests/test_ref_thirdparty.py to
// check the from-scratch loader / image decoder of cuda-pathtracer_amd/host against the real thing.
// The product never links or loads this file.
#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>
#define STB_IMAGE_IMPLEMENTATION
#include <stb/stb_image.h>
#define STB_IMAGE_RESIZE_IMPLEMENTATION
#include <stb/stb_image_resize.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

void put_floats(std::string& s, const char* key, const float* p, size_t n) {
  s += "\""; s += key; s += "\":[";
  char buf[16];
  for (size_t i = 0; i < n; ++i) { std::snprintf(buf, sizeof buf, i ? ",%u" : "%u", bits(p[i])); s += buf; }
  s += "]";
}

void put_string(std::string& s, const char* key, const std::string& v) {
  s += "\""; s += key; s += "\":\"";
  for (unsigned char c : v) {
    char buf[8];
    if (c == '"' || c == '\\') { s += '\\'; s += (char)c; }
    else if (c < 0x20 || c >= 0x7f) { std::snprintf(buf, sizeof buf, "\\u%04x", c); s += buf; }
    else s += (char)c;
  }
  s += "\"";
}

} // namespace

extern "C" {

// Runs tinyobj::LoadObj exactly as scene.cpp:341 does (triangulate defaulted to true) and returns its
// raw output as a JSON document (floats as their uint32 bit patterns). Caller frees with ref_free.
// On failure returns NULL and writes tinyobj's error string into err.
char* ref_tinyobj_load(const char* obj_path, const char* mtl_dir, char* err, int err_len) {
  tinyobj::attrib_t attrib;
  std::vector<tinyobj::shape_t> shapes;
  std::vector<tinyobj::material_t> materials;
  std::string load_error;
  bool ok = tinyobj::LoadObj(&attrib, &shapes, &materials, &load_error, obj_path, mtl_dir);
  if (err && err_len > 0) { std::snprintf(err, (size_t)err_len, "%s", load_error.c_str()); }
  if (!ok) return nullptr;
  std::string s = "{";
  put_floats(s, "vertices", attrib.vertices.data(), attrib.vertices.size()); s += ",";
  put_floats(s, "normals", attrib.normals.data(), attrib.normals.size()); s += ",";
  put_floats(s, "texcoords", attrib.texcoords.data(), attrib.texcoords.size()); s += ",\"shapes\":[";
  for (size_t i = 0; i < shapes.size(); ++i) {
    const auto& m = shapes[i].mesh;
    if (i) s += ",";
    s += "{"; put_string(s, "name", shapes[i].name); s += ",\"indices\":[";
    char buf[48];
    for (size_t k = 0; k < m.indices.size(); ++k) {
      std::snprintf(buf, sizeof buf, k ? ",%d,%d,%d" : "%d,%d,%d", m.indices[k].vertex_index,
                    m.indices[k].normal_index, m.indices[k].texcoord_index);
      s += buf;
    }
    s += "],\"num_face_vertices\":[";
    for (size_t k = 0; k < m.num_face_vertices.size(); ++k) {
      std::snprintf(buf, sizeof buf, k ? ",%d" : "%d", (int)m.num_face_vertices[k]); s += buf;
    }
    s += "],\"material_ids\":[";
    for (size_t k = 0; k < m.material_ids.size(); ++k) {
      std::snprintf(buf, sizeof buf, k ? ",%d" : "%d", m.material_ids[k]); s += buf;
    }
    s += "]}";
  }
  s += "],\"materials\":[";
  for (size_t i = 0; i < materials.size(); ++i) {
    const auto& m = materials[i];
    if (i) s += ",";
    s += "{"; put_string(s, "name", m.name); s += ",";
    put_floats(s, "ambient", m.ambient, 3); s += ",";
    put_floats(s, "diffuse", m.diffuse, 3); s += ",";
    put_floats(s, "specular", m.specular, 3); s += ",";
    put_floats(s, "transmittance", m.transmittance, 3); s += ",";
    put_floats(s, "emission", m.emission, 3); s += ",";
    put_floats(s, "shininess", &m.shininess, 1); s += ",";
    put_floats(s, "ior", &m.ior, 1); s += ",";
    put_floats(s, "dissolve", &m.dissolve, 1); s += ",";
    char buf[32]; std::snprintf(buf, sizeof buf, "\"illum\":%d,", m.illum); s += buf;
    put_string(s, "ambient_texname", m.ambient_texname); s += ",";
    put_string(s, "diffuse_texname", m.diffuse_texname); s += ",";
    put_string(s, "specular_texname", m.specular_texname); s += ",";
    put_string(s, "specular_highlight_texname", m.specular_highlight_texname); s += ",";
    put_string(s, "bump_texname", m.bump_texname); s += ",";
    put_string(s, "displacement_texname", m.displacement_texname); s += ",";
    put_string(s, "alpha_texname", m.alpha_texname); s += ",";
    put_string(s, "normal_texname", m.normal_texname);
    s += "}";
  }
  s += "]}";
  char* out = (char*)std::malloc(s.size() + 1);
  if (out) std::memcpy(out, s.c_str(), s.size() + 1);
  return out;
}

// stbi_loadf(path, &w, &h, &n, STBI_default) — the call of material_loader.cpp:97 / gpu_processor.cpp:99.
float* ref_stbi_loadf(const char* path, int* w, int* h, int* n) { return stbi_loadf(path, w, h, n, STBI_default); }

// stbi_load(path, ..., STBI_default): the 8-bit image the float conversion starts from.
unsigned char* ref_stbi_load(const char* path, int* w, int* h, int* n) { return stbi_load(path, w, h, n, STBI_default); }

const char* ref_stbi_failure_reason(void) { return stbi_failure_reason(); }

// stbir_resize_float(in, iw, ih, 0, out, ow, oh, 0, channels) — material_loader.cpp:358-370.
int ref_stbir_resize_float(const float* in, int iw, int ih, float* out, int ow, int oh, int channels) {
  return stbir_resize_float(in, iw, ih, 0, out, ow, oh, 0, channels);
}

void ref_free(void* p) { std::free(p); }

} // extern "C"
