// ptamd_internal.h — shared declarations of libptamd's translation units (not installed).
#pragma once

#include "ptamd.h"

#include <string>
#include <vector>

namespace ptamd {

void set_error(const std::string& msg);

// Every tuning / A-B knob of the library (PTAMD_ROUND_MIN, PTAMD_OVERLAP, PTAMD_BVH_MAX_LEAF, ...) is an environment variable that
// is read ONLY when PTAMD_TUNING=1 is set too: a production host's environment cannot change how the library renders.
// Returns the variable's value, or nullptr when it is unset or tuning is off.
const char* tuning_env(const char* name);

// Output of the host loader: the flattened arrays ptamd_scene_desc points into.
struct HostScene {
  std::vector<ptamd_face> faces;
  std::vector<uint32_t> mesh_sizes;
  std::vector<ptamd_material> materials;
  std::vector<ptamd_light> lights;
  std::vector<ptamd_texture_desc> textures;
  std::vector<float> texels;
  std::vector<std::string> unloaded_textures; // image files named by the MTL that were not decoded
  ptamd_camera camera;
  std::string cubemap;
};

// Image decoding is injected by the host application (the role stb_image plays for the reference).
struct ImageProvider {
  ptamd_image_load_fn load;
  ptamd_image_free_fn release;
  void* user;
};

// provider == nullptr: the built-in decoder (image_decode.cpp) unless flags bit 1 (PTAMD_LOAD_NO_IMAGES) is set.
int load_host_scene(const char* scene_path, uint32_t flags, const ImageProvider* provider, HostScene*& out);

// ---- built-in image decoding (image_decode.cpp): JPEG, bit-identical to the reference's stb_image 2.16
struct Image8 { int w = 0, h = 0, c = 0; std::vector<uint8_t> px; };
bool decode_jpeg(const uint8_t* bytes, size_t n_bytes, Image8& img, std::string& err);
bool decode_png(const uint8_t* bytes, size_t n_bytes, Image8& img, std::string& err);   // image_png.cpp
bool encode_png(const uint8_t* pixels, int w, int h, int channels, std::vector<uint8_t>& out);
bool load_image8(const char* path, Image8& img, std::string& err);
// stbi_loadf(path, &w, &h, &c, STBI_default): *data is malloc'd, free with std::free / ptamd_image_free
bool load_image_float(const char* path, int* w, int* h, int* c, float** data, std::string& err);
const float* ldr_to_linear_table();
// stbir_resize_float(in, w, h, 0, out, W, H, 0, channels) restated (image_resize.cpp)
bool resize_float(const float* in, int in_w, int in_h, float* out, int out_w, int out_h, int channels);
const ImageProvider* builtin_image_provider();

// ---- BVH (bvh_builder.cpp) -----------------------------------------------------------
//
// Binary SAH BVH, laid out for a stackless, per-octant ORDERED threaded traversal:
//   node record = 64 bytes = 4 x float4
//     q0 = { lo.x, lo.y, lo.z, bits(first_tri | count << 24) }   (count == 0: interior)
//     q1 = { hi.x, hi.y, hi.z, bits(right_child | split_axis << 30) }   (left child = node + 1, DFS pre-order)
//     q2,q3 = miss[8]: for ray octant o (bit a set <=> dir[a] < 0), the node to test next when this box is
//             missed or its subtree is done (0xFFFFFFFF = end of the walk)
//   A ray visits the child nearer along the split axis first (right child when it runs against the axis).
//   The kernels' LDS copy re-encodes the links as 16-bit hit|miss address pairs (pt_kernels.hip: stage_scene).
// Triangles are re-ordered leaf-major; record = 48 bytes = 3 x float4
//     t0 = { e1.x, e1.y, e1.z, e2.x }  t1 = { e2.y, e2.z, v0.x, v0.y }
//     t2 = { v0.z, bits(global face index), 0, 0 }      with e1 = v1 - v0, e2 = v2 - v0
// (edges first: the determinant test of Moller-Trumbore needs only the first 24 bytes)
struct Bvh {
  std::vector<float> nodes;      // 16 floats per node
  std::vector<float> tris;       // 12 floats per triangle, leaf-major
  uint32_t n_nodes = 0, n_leaves = 0, max_leaf = 0, depth = 0;
  uint32_t n_tris = 0;           // triangle records (>= faces when references were split)
  // The same tree with four children per node, for scenes walked from L2/HBM with a per-lane stack (one 128-byte line
  // per node visit instead of one dependent 64-byte load per box test).  node record = 32 floats = 8 x float4:
  //   q0..q2 = centre.x[4], .y[4], .z[4]     q3..q5 = half extent .x[4], .y[4], .z[4]  (child boxes, one lane per child;
  //            slab distances of an axis = t(centre) -+ half * |1/d|: no min / max per axis)
  //   q6     = reference[4]: 0xFFFFFFFF empty | node index | 0x80000000 | count << 24 | first triangle (leaf)
  //   q7     = 8 halfwords, one per ray octant: nibble c = the children that octant visits AFTER child c
  // Nodes are numbered breadth-first (node 0 = root).  Leaves point into `tris`.
  std::vector<float> nodes4;
  uint32_t n_nodes4 = 0, depth4 = 0;
  // The same four-wide nodes (same numbering, same references) in 64 bytes: 16 dwords
  //   [0..2] origin (float)   [3] exponent bytes ex | ey << 8 | ez << 16 (scale[a] = 2^(e[a] - 127)), child count << 24
  //   [4..7] reference[4]     [8..10] low planes x[4] y[4] z[4], one byte per child   [11..13] high planes x[4] y[4] z[4]
  //          child box = origin + plane * scale, low planes rounded down, high planes up; an empty slot has low 255, high 0
  //   [14..15] halfword o (octants 0..3): nibble c = the children octant o visits AFTER child c; octant 7 - o visits them in
  //          the reverse order
  std::vector<uint32_t> nodes4q;
  // ... and with EIGHT children per node and quantised child boxes: the walk of scenes that do not fit in LDS.  node record =
  // 32 dwords = ONE 128-byte line:
  //   [0..2]   origin (float): the low corner of the node's (inflated) box      [3] exponent bytes ex | ey << 8 | ez << 16
  //            (scale[a] = 2^(e[a] - 127)), child count << 24
  //   [4..11]  reference[8]: 0xFFFFFFFF empty | node index | 0x80000000 | count << 24 | first triangle (leaf)
  //   [12..17] low planes  x[8] y[8] z[8], one byte per child       [18..23] high planes x[8] y[8] z[8]
  //            child box = origin + plane * scale, low planes rounded down, high planes up (the inflated child box is inside)
  //   [24..31] unused
  // Children sit in slots by direction (slot bits = signs of their offset from the node's centre), so that ascending
  // (slot ^ ray octant) is a front-to-back order.  Breadth-first numbering as for nodes4.
  std::vector<uint32_t> nodes8;
  uint32_t n_nodes8 = 0, depth8 = 0, max_leaf8 = 0;
  float extent = 0.0f;           // largest finite |coordinate| of the scene
  bool all_finite = true;        // no vertex coordinate is NaN or infinite
  float margin_floor = 0.0f;     // smallest inflation any box face received (absolute margin + extent * 2^-20)
};


// margin: absolute inflation added to every box face (see DESIGN.md "Conservative boxes")
// forms: which of the knob-only node forms to build beside the binary tree and the four-wide float nodes (the eight-wide quantised
// nodes take a dynamic programme over the whole tree; a production upload builds neither)
constexpr uint32_t kBvhForm8 = 1u, kBvhForm4q = 2u;
int build_bvh(const ptamd_face* faces, uint32_t n_faces, float margin, uint32_t max_leaf, Bvh& out, uint32_t forms = kBvhForm8 | kBvhForm4q);

// Host traversal with the same structure the kernel uses (tests + stats cross-check).
struct HostHit { int32_t kind; int32_t index; float t; float u, v; };
void bvh_trace_host(const Bvh& bvh, const ptamd_face* faces, const float dir[3], const float origin[3],
                    HostHit& out, uint64_t* nodes_visited, uint64_t* tris_tested);
void bvh4_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                     uint64_t* tris_tested);
void bvh8_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                     uint64_t* tris_tested);
void bvh4q_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                      uint64_t* tris_tested);

} // namespace ptamd
