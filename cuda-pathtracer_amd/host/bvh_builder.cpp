// bvh_builder.cpp — host SAH BVH builder + the host mirror of the device traversal.
//
// The reference has NO acceleration structure: its intersect() is a double loop over
// every face (cuda_opengl/include/shaders/intersection.cuh:179-196).  The BVH is a
// build-side accelerator whose contract is: for every ray, return exactly the record the
// brute-force loop returns, i.e. the lexicographic minimum of (t, global face index) over
// the faces whose Moller-Trumbore test passes with t > 0 (first-wins strict `<` in storage
// order == that minimum).  Culling must therefore be conservative; see DESIGN.md
// "Conservative boxes" for the margin argument and the measured miss rate.
//
// Layout (documented in ptamd_internal.h): 64-byte nodes with per-octant miss links so a
// ray walks the tree front-to-back without a stack; triangles re-ordered leaf-major as
// 48-byte {v0, e1, e2, global index} records.
#include "ptamd_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace ptamd {

namespace {

struct Box {
  float lo[3], hi[3];
  void reset()
  {
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::max(); hi[a] = -std::numeric_limits<float>::max(); }
  }
  void grow(const Box& b)
  {
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
  }
  float half_area() const
  {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0.f || dy < 0.f || dz < 0.f) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Prim { Box box; float c[3]; uint32_t face; };

struct BuildNode {
  Box box;
  int left = -1, right = -1; // build-node indices; leaf when left < 0
  uint32_t first = 0, count = 0;
  int axis = 0;
};

constexpr float kTravCost = 1.0f;
float g_isect_cost = 1.6f; // SAH cost of one triangle test relative to one box test (build_bvh may override)
constexpr int kBins = 32;
uint32_t g_sweep_limit = 1u << 30;   // nodes with more primitives than this are split by binning (build_bvh may override)

struct Builder {
  std::vector<Prim> prims;
  std::vector<BuildNode> nodes;
  uint32_t max_leaf = 4;
  uint32_t depth = 0;

  int build(uint32_t first, uint32_t count, uint32_t level)
  {
    depth = std::max(depth, level + 1);
    int id = (int)nodes.size();
    nodes.emplace_back();
    Box box, cbox;
    box.reset(); cbox.reset();
    for (uint32_t i = first; i < first + count; ++i) {
      box.grow(prims[i].box);
      for (int a = 0; a < 3; ++a) {
        cbox.lo[a] = std::min(cbox.lo[a], prims[i].c[a]);
        cbox.hi[a] = std::max(cbox.hi[a], prims[i].c[a]);
      }
    }
    nodes[id].box = box;
    nodes[id].first = first;
    nodes[id].count = count;
    if (count == 1) return id;

    // best split
    float best_cost = std::numeric_limits<float>::max();
    int best_axis = -1;
    uint32_t best_mid = 0;   // sweep: split position in sorted order
    float best_plane = 0.f;  // binned: centroid threshold
    bool best_binned = false;
    const float inv_area = 1.0f / std::max(box.half_area(), 1e-30f);

    if (count <= g_sweep_limit) {
      std::vector<float> right_area(count);
      for (int a = 0; a < 3; ++a) {
        if (!(cbox.hi[a] > cbox.lo[a])) continue;
        std::sort(prims.begin() + first, prims.begin() + first + count,
                  [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.face < y.face); });
        Box acc;
        acc.reset();
        for (uint32_t i = count; i-- > 1;) {
          acc.grow(prims[first + i].box);
          right_area[i] = acc.half_area();
        }
        acc.reset();
        for (uint32_t i = 1; i < count; ++i) {
          acc.grow(prims[first + i - 1].box);
          float cost = kTravCost + g_isect_cost * inv_area * (acc.half_area() * (float)i + right_area[i] * (float)(count - i));
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_mid = i; best_binned = false; }
        }
      }
    } else {
      for (int a = 0; a < 3; ++a) {
        float ext = cbox.hi[a] - cbox.lo[a];
        if (!(ext > 0.f)) continue;
        Box bbox[kBins];
        uint32_t bcnt[kBins];
        for (int b = 0; b < kBins; ++b) { bbox[b].reset(); bcnt[b] = 0; }
        float scale = (float)kBins / ext;
        for (uint32_t i = first; i < first + count; ++i) {
          int b = std::min(kBins - 1, std::max(0, (int)((prims[i].c[a] - cbox.lo[a]) * scale)));
          bbox[b].grow(prims[i].box);
          bcnt[b]++;
        }
        float rarea[kBins];
        uint32_t rcnt[kBins];
        Box acc;
        acc.reset();
        uint32_t n = 0;
        for (int b = kBins - 1; b >= 1; --b) {
          acc.grow(bbox[b]); n += bcnt[b];
          rarea[b] = acc.half_area(); rcnt[b] = n;
        }
        acc.reset();
        n = 0;
        for (int b = 1; b < kBins; ++b) {
          acc.grow(bbox[b - 1]); n += bcnt[b - 1];
          if (n == 0 || rcnt[b] == 0) continue;
          float cost = kTravCost + g_isect_cost * inv_area * (acc.half_area() * (float)n + rarea[b] * (float)rcnt[b]);
          if (cost < best_cost) {
            best_cost = cost; best_axis = a; best_binned = true;
            best_plane = cbox.lo[a] + (float)b / scale;
          }
        }
      }
    }

    const float leaf_cost = g_isect_cost * (float)count;
    if (count <= max_leaf && (best_axis < 0 || leaf_cost <= best_cost)) return id; // leaf

    uint32_t mid;
    if (best_axis < 0) {
      // all centroids coincide: split by face index halves
      std::sort(prims.begin() + first, prims.begin() + first + count,
                [](const Prim& x, const Prim& y) { return x.face < y.face; });
      mid = count / 2;
      best_axis = 0;
    } else if (best_binned) {
      const int a = best_axis;
      const float plane = best_plane;
      auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
                               [a, plane](const Prim& p) { return p.c[a] < plane; });
      mid = (uint32_t)(it - (prims.begin() + first));
      if (mid == 0 || mid == count) {
        std::sort(prims.begin() + first, prims.begin() + first + count,
                  [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.face < y.face); });
        mid = count / 2;
      }
    } else {
      const int a = best_axis;
      std::sort(prims.begin() + first, prims.begin() + first + count,
                [a](const Prim& x, const Prim& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.face < y.face); });
      mid = best_mid;
    }
    int l = build(first, mid, level + 1);
    int r = build(first + mid, count - mid, level + 1);
    nodes[id].left = l;
    nodes[id].right = r;
    nodes[id].axis = best_axis;
    return id;
  }
};

// Clips a convex polygon (double coordinates) against the half-space  x[axis] <= pos  (keep_low)
// or  x[axis] >= pos.
void clip_polygon(const std::vector<double>& in, int axis, double pos, bool keep_low, std::vector<double>& out)
{
  out.clear();
  const size_t n = in.size() / 3;
  for (size_t i = 0; i < n; ++i) {
    const double* a = &in[i * 3];
    const double* b = &in[((i + 1) % n) * 3];
    const bool ia = keep_low ? a[axis] <= pos : a[axis] >= pos;
    const bool ib = keep_low ? b[axis] <= pos : b[axis] >= pos;
    if (ia) out.insert(out.end(), a, a + 3);
    if (ia != ib) {
      const double t = (pos - a[axis]) / (b[axis] - a[axis]);
      double c[3] = { a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]), a[2] + t * (b[2] - a[2]) };
      c[axis] = pos;
      out.insert(out.end(), c, c + 3);
    }
  }
}

// alpha: a reference is split while the half-area of its box exceeds alpha * (half-area of the
// scene box); budget: at most this many extra references.
void split_references(std::vector<Prim>& prims, const ptamd_face* faces, float alpha, uint32_t budget)
{
  if (!(alpha > 0.f) || budget == 0 || prims.empty()) return;
  Box scene;
  scene.reset();
  for (const Prim& p : prims) scene.grow(p.box);
  const float threshold = alpha * scene.half_area();
  struct Item { Prim prim; std::vector<double> poly; };
  std::vector<Item> work;
  std::vector<Prim> done;
  for (const Prim& p : prims) {
    if (p.box.half_area() > threshold) {
      Item it;
      it.prim = p;
      for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a) it.poly.push_back((double)(&faces[p.face].vertices[k].x)[a]);
      work.push_back(std::move(it));
    } else {
      done.push_back(p);
    }
  }
  uint32_t extra = 0;
  std::vector<double> lo_poly, hi_poly;
  while (!work.empty()) {
    // largest first
    size_t big = 0;
    for (size_t i = 1; i < work.size(); ++i)
      if (work[i].prim.box.half_area() > work[big].prim.box.half_area()) big = i;
    Item it = std::move(work[big]);
    work.erase(work.begin() + (long)big);
    if (extra >= budget || !(it.prim.box.half_area() > threshold)) { done.push_back(it.prim); continue; }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
      if (it.prim.box.hi[a] - it.prim.box.lo[a] > it.prim.box.hi[axis] - it.prim.box.lo[axis]) axis = a;
    const double pos = 0.5 * ((double)it.prim.box.lo[axis] + (double)it.prim.box.hi[axis]);
    clip_polygon(it.poly, axis, pos, true, lo_poly);
    clip_polygon(it.poly, axis, pos, false, hi_poly);
    if (lo_poly.size() < 9 || hi_poly.size() < 9) { done.push_back(it.prim); continue; } // degenerate: keep whole
    const std::vector<double>* halves[2] = { &lo_poly, &hi_poly };
    for (int h = 0; h < 2; ++h) {
      Item c;
      c.prim.face = it.prim.face;
      c.prim.box.reset();
      c.poly = *halves[h];
      for (size_t i = 0; i < c.poly.size(); i += 3)
        for (int a = 0; a < 3; ++a) {
          // round outwards when narrowing to float; the clipped box never exceeds the parent's
          const float f = (float)c.poly[i + a];
          const float flo = (double)f > c.poly[i + a] ? std::nextafter(f, -std::numeric_limits<float>::infinity()) : f;
          const float fhi = (double)f < c.poly[i + a] ? std::nextafter(f, std::numeric_limits<float>::infinity()) : f;
          c.prim.box.lo[a] = std::max(std::min(c.prim.box.lo[a], flo), it.prim.box.lo[a]);
          c.prim.box.hi[a] = std::min(std::max(c.prim.box.hi[a], fhi), it.prim.box.hi[a]);
        }
      for (int a = 0; a < 3; ++a) c.prim.c[a] = 0.5f * c.prim.box.lo[a] + 0.5f * c.prim.box.hi[a];
      work.push_back(std::move(c));
    }
    ++extra;
  }
  prims.swap(done);
}

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

} // namespace

int build_bvh(const ptamd_face* faces, uint32_t n_faces, float margin, uint32_t max_leaf, Bvh& out, uint32_t forms)
{
  out = Bvh();
  float split_alpha = 0.0f;   // pre-splitting off unless asked for (tuning knobs)
  uint32_t split_budget = n_faces / 2u + 16u;
  if (const char* e = tuning_env("PTAMD_BVH_SPLIT_ALPHA")) split_alpha = (float)std::atof(e);
  if (const char* e = tuning_env("PTAMD_BVH_SPLIT_BUDGET")) split_budget = (uint32_t)std::atoi(e);
  if (n_faces == 0) return PTAMD_OK;
  if (n_faces >= (1u << 24)) { set_error("build_bvh: more than 2^24 faces"); return PTAMD_ERR_LIMIT; }
  if (const char* e = tuning_env("PTAMD_BVH_MAX_LEAF")) max_leaf = (uint32_t)std::atoi(e);   // tuning knobs
  // (process-wide builder constants: back to their defaults on every build, so that a knob set for one build does not outlive it)
  g_isect_cost = 1.6f; g_sweep_limit = 1u << 30;
  if (const char* e = tuning_env("PTAMD_BVH_ISECT_COST")) g_isect_cost = (float)std::atof(e);
  if (const char* e = tuning_env("PTAMD_BVH_SWEEP_LIMIT")) g_sweep_limit = (uint32_t)std::atoi(e);   // (2048: round 2's builder)
  if (max_leaf < 1) max_leaf = 1;
  if (max_leaf > 15) max_leaf = 15;
  Builder b;
  b.max_leaf = max_leaf;
  b.prims.resize(n_faces);
  float extent = 0.0f;   // largest finite |coordinate| of the scene
  bool all_finite = true;
  for (uint32_t i = 0; i < n_faces; ++i) {
    Prim& p = b.prims[i];
    p.face = i;
    p.box.reset();
    for (int k = 0; k < 3; ++k) {
      const float* v = &faces[i].vertices[k].x;
      for (int a = 0; a < 3; ++a) {
        // NaN and infinite coordinates would poison every ancestor box (an infinite one gives the quantised nodes an origin of
        // -inf, and every plane of such a node decodes to NaN): keep them out of the bounds.  Such a face can never pass
        // Moller-Trumbore either way — an infinite edge makes the determinant +-inf or NaN, and with 1 / det = 0 or NaN the hit
        // distance comes out NaN, which `t > 0` rejects (tests/test_gpu_parity.py: every kernel == the brute-force oracle on them).
        if (std::fabs(v[a]) <= std::numeric_limits<float>::max()) { p.box.lo[a] = std::min(p.box.lo[a], v[a]); p.box.hi[a] = std::max(p.box.hi[a], v[a]); }
        if (std::fabs(v[a]) <= std::numeric_limits<float>::max()) extent = std::max(extent, std::fabs(v[a]));
        else all_finite = false;
      }
    }
    for (int a = 0; a < 3; ++a) {
      if (p.box.lo[a] > p.box.hi[a]) { p.box.lo[a] = 0.f; p.box.hi[a] = 0.f; }
      p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
    }
  }
  // ---- reference pre-splitting ("early split clipping"): a few huge faces (walls, floor) would
  // otherwise bloat every ancestor box.  Such a face is represented by several REFERENCES, each
  // with the tight box of (face clipped to a sub-box); all of them point at the same triangle
  // record, so Moller-Trumbore and the (t, index) minimum are untouched — a face tested twice
  // yields the same candidate twice.
  split_references(b.prims, faces, split_alpha, split_budget);
  const uint32_t n_refs = (uint32_t)b.prims.size();
  b.nodes.reserve(2 * n_refs);
  b.build(0, n_refs, 0);

  // ---- flatten: DFS pre-order, left child first
  const uint32_t n_nodes = (uint32_t)b.nodes.size();
  std::vector<uint32_t> order(n_nodes), pos(n_nodes);
  {
    std::vector<int> stack;
    stack.push_back(0);
    uint32_t k = 0;
    while (!stack.empty()) {
      int id = stack.back();
      stack.pop_back();
      pos[id] = k;
      order[k++] = (uint32_t)id;
      if (b.nodes[id].left >= 0) {
        stack.push_back(b.nodes[id].right);
        stack.push_back(b.nodes[id].left);
      }
    }
  }
  out.n_nodes = n_nodes;
  out.depth = b.depth;
  out.nodes.assign((size_t)n_nodes * 16, 0.0f);
  out.tris.assign((size_t)n_refs * 12, 0.0f); // upper bound; trimmed after the leaves are written

  // per-octant miss links: top-down.  miss[o] of the root is END.
  std::vector<uint32_t> miss((size_t)n_nodes * 8, 0xFFFFFFFFu);
  for (uint32_t k = 0; k < n_nodes; ++k) {
    const BuildNode& bn = b.nodes[order[k]];
    if (bn.left < 0) continue;
    const uint32_t l = pos[bn.left], r = pos[bn.right];
    for (int o = 0; o < 8; ++o) {
      const bool right_first = (o >> bn.axis) & 1; // direction negative along the split axis
      const uint32_t first = right_first ? r : l, second = right_first ? l : r;
      miss[(size_t)first * 8 + o] = second;
      miss[(size_t)second * 8 + o] = miss[(size_t)k * 8 + o];
    }
  }

  // The slab tests of the LDS loop and of the four-wide float walk form an axis' distances from the box's centre and half
  // extent: tc = fma(c, 1/d, -(o/d)), then fma(-+h, |1/d|, tc).  Expressed as a displacement of the plane along the axis the
  // roundings add up to: the reciprocal (v_rcp_f32, 1 ulp) (|o| + |p|) 2^-23, -(o/d) |o| 2^-24, tc (|o| + |c|) 2^-24 and the
  // final fma (|o| + |p|) 2^-24 — at most (|o| + |p|) 2^-22 + |o| 2^-24 with an exact reciprocal, about 1.75 (|o| + |p|) 2^-22
  // with a 2-ulp one.  (The box [c - h, c + h] itself contains [lo, hi] exactly: h is rounded up where the record is formed.)
  // Bounce rays start on the scene's surfaces (|o| <= extent), so every box also gets extent * 2^-20 — 2.3x the worst case at
  // |o| = |p| = extent — and |p| * 1e-6 on top; a camera much farther out than the scene is the launcher's business
  // (ptamd_api.cpp: far-origin check, (|camera| + extent) * 2^-21 against Bvh::margin_floor).
  const float origin_margin = extent * (1.0f / 1048576.0f);
  out.extent = extent;
  out.all_finite = all_finite;
  out.margin_floor = margin + origin_margin;
  uint32_t tri_cursor = 0;
  std::vector<uint32_t> leaf_info(b.nodes.size(), 0u);   // build-node id -> first_tri | count << 24 (leaves only)
  for (uint32_t k = 0; k < n_nodes; ++k) {
    const BuildNode& bn = b.nodes[order[k]];
    float* q = &out.nodes[(size_t)k * 16];
    for (int a = 0; a < 3; ++a) {
      float lo = bn.box.lo[a], hi = bn.box.hi[a];
      q[a] = lo - (margin + origin_margin + std::fabs(lo) * 1e-6f);
      q[4 + a] = hi + (margin + origin_margin + std::fabs(hi) * 1e-6f);
    }
    uint32_t info = 0, child = 0;
    if (bn.left < 0) {
      out.n_leaves++;
      // triangles of a leaf in ascending global face index
      std::vector<uint32_t> ids;
      for (uint32_t i = 0; i < bn.count; ++i) ids.push_back(b.prims[bn.first + i].face);
      std::sort(ids.begin(), ids.end());
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end()); // two references of one face in one leaf
      info = tri_cursor | ((uint32_t)ids.size() << 24);
      leaf_info[order[k]] = info;
      out.max_leaf = std::max(out.max_leaf, (uint32_t)ids.size());
      for (uint32_t fi : ids) {
        const ptamd_face& f = faces[fi];
        float* t = &out.tris[(size_t)tri_cursor * 12];
        // e1/e2 are the reference's v0v1/v0v2 (intersection.cuh:106-107): same subtraction.  Record order e1, e2, v0,
        // index: the determinant test needs only the first 24 bytes, v0 and the index come with the second read
        t[0] = f.vertices[1].x - f.vertices[0].x;
        t[1] = f.vertices[1].y - f.vertices[0].y;
        t[2] = f.vertices[1].z - f.vertices[0].z;
        t[3] = f.vertices[2].x - f.vertices[0].x;
        t[4] = f.vertices[2].y - f.vertices[0].y;
        t[5] = f.vertices[2].z - f.vertices[0].z;
        t[6] = f.vertices[0].x; t[7] = f.vertices[0].y; t[8] = f.vertices[0].z;
        t[9] = u2f(fi);
        tri_cursor++;
      }
    } else {
      child = pos[bn.right] | ((uint32_t)bn.axis << 30);
    }
    q[3] = u2f(info);
    q[7] = u2f(child);
    for (int o = 0; o < 8; ++o) q[8 + o] = u2f(miss[(size_t)k * 8 + o]);
  }
  out.tris.resize((size_t)tri_cursor * 12);
  out.n_tris = tri_cursor;

  // ---- the same tree, collapsed to four children per node (layout: ptamd_internal.h).  A node's children start as the
  // two children of a binary node; the interior child with the largest box is replaced by its own two children until
  // there are four (or only leaves are left).  Leaves keep their triangle ranges in `tris`.
  {
    struct Wide { int child[4]; int n; };
    std::vector<Wide> wide;
    std::vector<int> wide_root;       // build-node id each wide node was made from
    std::vector<uint32_t> wide_depth;
    wide_root.push_back(0);
    wide_depth.push_back(1);
    for (size_t w = 0; w < wide_root.size(); ++w) {      // breadth-first: the top of the tree is contiguous
      Wide wn;
      wn.n = 0;
      const BuildNode& root = b.nodes[(size_t)wide_root[w]];
      if (root.left < 0) { wn.child[wn.n++] = wide_root[w]; }          // a one-leaf tree: the root node holds that leaf
      else { wn.child[wn.n++] = root.left; wn.child[wn.n++] = root.right; }
      while (wn.n < 4) {
        int pick = -1;
        float area = -1.0f;
        for (int i = 0; i < wn.n; ++i) {
          const BuildNode& c = b.nodes[(size_t)wn.child[i]];
          if (c.left >= 0 && c.box.half_area() > area) { area = c.box.half_area(); pick = i; }
        }
        if (pick < 0) break;
        const BuildNode& c = b.nodes[(size_t)wn.child[pick]];
        wn.child[pick] = c.left;
        wn.child[wn.n++] = c.right;
      }
      for (int i = wn.n; i < 4; ++i) wn.child[i] = -1;
      wide.push_back(wn);
      out.depth4 = std::max(out.depth4, wide_depth[w]);
      for (int i = 0; i < wn.n; ++i)
        if (b.nodes[(size_t)wn.child[i]].left >= 0) {
          // interior child: becomes a wide node of its own; remember where (negative marker resolved below)
          wide_root.push_back(wn.child[i]);
          wide_depth.push_back(wide_depth[w] + 1);
        }
    }
    // wide node index of every build node that became one (in push order)
    std::vector<int> wide_of(b.nodes.size(), -1);
    for (size_t w = 0; w < wide_root.size(); ++w) wide_of[(size_t)wide_root[w]] = (int)w;
    out.n_nodes4 = (uint32_t)wide.size();
    out.nodes4.assign((size_t)out.n_nodes4 * 32, 0.0f);
    for (size_t w = 0; w < wide.size(); ++w) {
      float* q = &out.nodes4[w * 32];
      float key[4][8];
      for (int c = 0; c < 4; ++c) {
        if (wide[w].child[c] < 0) {
          // empty slot: a point box far beyond MAX_DIST (no ray reaches it: its slab distances are +-huge, never within
          // [0, best <= 1e5]), reference 0xFFFFFFFF
          for (int a = 0; a < 3; ++a) { q[a * 4 + c] = 3.0e38f; q[12 + a * 4 + c] = 0.0f; }   // (centre, half extent)
          q[24 + c] = u2f(0xFFFFFFFFu);
          continue;
        }
        const BuildNode& cn = b.nodes[(size_t)wide[w].child[c]];
        for (int a = 0; a < 3; ++a) {
          const float lo = cn.box.lo[a], hi = cn.box.hi[a];
          // stored as centre and half extent (the walk then needs no min / max per axis: t(centre) -+ half * |1/d|); the half extent
          // is rounded up, so [centre - half, centre + half] contains the inflated box; a box that is not finite becomes "everything"
          const float blo = lo - (margin + origin_margin + std::fabs(lo) * 1e-6f), bhi = hi + (margin + origin_margin + std::fabs(hi) * 1e-6f);
          float ctr = 0.5f * blo + 0.5f * bhi;
          float half = std::max(bhi - ctr, ctr - blo) * 1.00000024f;
          if (!(std::fabs(blo) <= 3.0e38f && std::fabs(bhi) <= 3.0e38f)) { ctr = 0.0f; half = 3.0e38f; }
          q[a * 4 + c] = ctr;
          q[12 + a * 4 + c] = half;
        }
        uint32_t ref;
        if (cn.left < 0) {
          const uint32_t info = leaf_info[(size_t)wide[w].child[c]];
          ref = 0x80000000u | ((info >> 24) << 24) | (info & 0xFFFFFFu);   // leaf: count in bits 24..30, first triangle below
        } else {
          ref = (uint32_t)wide_of[(size_t)wide[w].child[c]];
        }
        q[24 + c] = u2f(ref);
        // traversal order of octant o: children sorted by the centre of their box along (+-1, +-1, +-1)
        for (int o = 0; o < 8; ++o) {
          float k = 0.0f;
          for (int a = 0; a < 3; ++a) {
            const float ctr = 0.5f * cn.box.lo[a] + 0.5f * cn.box.hi[a];
            k += ((o >> a) & 1) ? -ctr : ctr;
          }
          key[c][o] = k;
        }
      }
      // halfword o of q[28..31]: nibble c = the children a ray of octant o visits AFTER child c (farther ones)
      uint32_t words[4] = { 0, 0, 0, 0 };
      for (int o = 0; o < 8; ++o) {
        uint32_t half = 0;
        for (int c = 0; c < 4; ++c) {
          if (wide[w].child[c] < 0) continue;
          uint32_t farther = 0;
          for (int d = 0; d < 4; ++d) {
            if (d == c || wide[w].child[d] < 0) continue;
            if (key[d][o] > key[c][o] || (key[d][o] == key[c][o] && d > c)) farther |= 1u << d;
          }
          half |= farther << (4 * c);
        }
        words[o >> 1] |= half << (16 * (o & 1));
      }
      for (int i = 0; i < 4; ++i) q[28 + i] = u2f(words[i]);
    }

    // ---- ... and the same four-wide nodes in 64 bytes (Bvh::nodes4q): child boxes as 8-bit planes on a per-node grid (float
    // origin, one power-of-two scale per axis), rounded outward.  Half the bytes and half the load instructions per visit.
    // The visiting order of an octant and of its opposite are each other's reverse: only octants 0..3 are stored.
    if (forms & kBvhForm4q) out.nodes4q.assign((size_t)out.n_nodes4 * 16, 0u);
    const float mq = margin + 2.0f * origin_margin;   // (two more roundings in the slab arithmetic than the float form: as for nodes8)
    for (size_t w = 0; (forms & kBvhForm4q) && w < wide.size(); ++w) {
      uint32_t* q = &out.nodes4q[w * 16];
      float lo_c[4][3], hi_c[4][3];
      float nlo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, nhi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
      for (int c = 0; c < 4; ++c) {
        if (wide[w].child[c] < 0) continue;
        const BuildNode& cn = b.nodes[(size_t)wide[w].child[c]];
        for (int a = 0; a < 3; ++a) {
          // (an infinite face is moved to the largest finite value: the node's origin and grid must stay finite)
          lo_c[c][a] = std::max(-3.0e38f, std::min(3.0e38f, cn.box.lo[a] - (mq + std::fabs(cn.box.lo[a]) * 1e-6f)));
          hi_c[c][a] = std::max(-3.0e38f, std::min(3.0e38f, cn.box.hi[a] + (mq + std::fabs(cn.box.hi[a]) * 1e-6f)));
          nlo[a] = std::min(nlo[a], lo_c[c][a]);
          nhi[a] = std::max(nhi[a], hi_c[c][a]);
        }
      }
      uint32_t expo[3];
      float scale[3];
      for (int a = 0; a < 3; ++a) {
        int e = 0;
        const float ext = nhi[a] - nlo[a];
        if (ext > 0.0f && ext <= std::numeric_limits<float>::max()) { (void)std::frexp(ext / 255.0f, &e); }   // 2^e >= ext / 255
        else if (!(ext <= std::numeric_limits<float>::max())) e = 120;
        else e = -120;
        e = std::max(-120, std::min(120, e));
        expo[a] = (uint32_t)(e + 127);
        scale[a] = std::ldexp(1.0f, e);
        q[a] = f2u(nlo[a]);
      }
      q[3] = expo[0] | (expo[1] << 8) | (expo[2] << 16) | ((uint32_t)wide[w].n << 24);
      uint8_t qlo[3][4], qhi[3][4];
      float key[4][4];
      for (int c = 0; c < 4; ++c) {
        uint32_t ref = 0xFFFFFFFFu;
        if (wide[w].child[c] < 0) {
          for (int a = 0; a < 3; ++a) { qlo[a][c] = 255; qhi[a][c] = 0; }   // inverted: no ray enters before it leaves
        } else {
          const BuildNode& cn = b.nodes[(size_t)wide[w].child[c]];
          for (int a = 0; a < 3; ++a) {
            // rounded outward, then checked with the device's own decode: fma(q, scale, origin) must enclose the child
            int l = (int)std::floor((lo_c[c][a] - nlo[a]) / scale[a]);
            int h = (int)std::ceil((hi_c[c][a] - nlo[a]) / scale[a]);
            if (!(lo_c[c][a] == lo_c[c][a]) || !(hi_c[c][a] == hi_c[c][a])) { l = 0; h = 255; }
            l = std::max(0, std::min(255, l));
            h = std::max(0, std::min(255, h));
            while (l > 0 && !(std::fma((float)l, scale[a], nlo[a]) <= lo_c[c][a])) --l;
            while (h < 255 && !(std::fma((float)h, scale[a], nlo[a]) >= hi_c[c][a])) ++h;
            qlo[a][c] = (uint8_t)l;
            qhi[a][c] = (uint8_t)h;
          }
          if (cn.left < 0) {
            const uint32_t info = leaf_info[(size_t)wide[w].child[c]];
            ref = 0x80000000u | ((info >> 24) << 24) | (info & 0xFFFFFFu);
          } else {
            ref = (uint32_t)wide_of[(size_t)wide[w].child[c]];
          }
          for (int o = 0; o < 4; ++o) {
            float k = 0.0f;
            for (int a = 0; a < 3; ++a) {
              const float ctr = 0.5f * cn.box.lo[a] + 0.5f * cn.box.hi[a];
              k += ((o >> a) & 1) ? -ctr : ctr;
            }
            key[c][o] = k;
          }
        }
        q[4 + c] = ref;
      }
      for (int a = 0; a < 3; ++a) {
        std::memcpy(&q[8 + a], qlo[a], 4);
        std::memcpy(&q[11 + a], qhi[a], 4);
      }
      uint32_t words[2] = { 0, 0 };
      for (int o = 0; o < 4; ++o) {
        uint32_t half = 0;
        for (int c = 0; c < 4; ++c) {
          if (wide[w].child[c] < 0) continue;
          uint32_t farther = 0;
          for (int d = 0; d < 4; ++d) {
            if (d == c || wide[w].child[d] < 0) continue;
            if (key[d][o] > key[c][o] || (key[d][o] == key[c][o] && d > c)) farther |= 1u << d;
          }
          half |= farther << (4 * c);
        }
        words[o >> 1] |= half << (16 * (o & 1));
      }
      q[14] = words[0];
      q[15] = words[1];
    }
  }

  // ---- the same tree once more, collapsed to EIGHT children per node with quantised child boxes (layout: ptamd_internal.h,
  // Bvh::nodes8): one 128-byte line per node again, half the node visits of the four-wide form.
  if (forms & kBvhForm8) {
    // Which binary nodes become the children of a wide node is decided by the dynamic programme of Ylitie, Karras, Laine
    // 2017 (section 3.1) instead of round 2's greedy "open the largest child" rule, which left the bottom of the tree full of
    // wide nodes with two or three leaves (3.1 children per node on the atrium): cost[n][i] = cheapest SAH cost of
    // representing the subtree of binary node n with at most i child slots of its parent —
    //   one slot:  a leaf holding all its triangles (if they are at most max_leaf), or a wide node of its own:
    //              area(n) * c_node + distribute(n, 8);
    //   i slots:   min(cost[n][i - 1], distribute(n, i)),  distribute(n, j) = min_k cost[left][k] + cost[right][j - k].
    // A child is a LEAF (its subtree's triangle records are contiguous: leaves were written in depth-first order) or a wide node.
    const float c_node = 1.0f, c_prim = 0.4f;   // a wide visit ~ 230 VALU + one line, a triangle test ~ 70 VALU + one record
    const size_t nb = b.nodes.size();
    std::vector<float> cost(nb * 8, 0.0f);            // [n][i - 1]
    std::vector<uint8_t> dec(nb * 8, 0);              // i = 1: 0 leaf, 1 wide node; i > 1: 0 = as with i - 1 slots, k = left gets k
    std::vector<uint8_t> dec8(nb, 0);                 // split of the 8 slots of n's own wide node
    std::vector<uint32_t> sub_first(nb, 0), sub_count(nb, 0);   // triangle records of the subtree (contiguous)
    {
      // post-order over build nodes: children have larger indices than their parent (nodes are appended while recursing)
      for (size_t i = nb; i-- > 0;) {
        const BuildNode& bn = b.nodes[i];
        const float area = bn.box.half_area();
        float* c = &cost[i * 8];
        if (bn.left < 0) {
          sub_first[i] = leaf_info[i] & 0xFFFFFFu;
          sub_count[i] = leaf_info[i] >> 24;
          for (int k = 0; k < 8; ++k) c[k] = area * (float)sub_count[i] * c_prim;
          continue;
        }
        const size_t l = (size_t)bn.left, r = (size_t)bn.right;
        sub_first[i] = std::min(sub_first[l], sub_first[r]);
        sub_count[i] = sub_count[l] + sub_count[r];
        float dist[9];
        uint8_t dk[9];
        for (int j = 2; j <= 8; ++j) {
          dist[j] = std::numeric_limits<float>::infinity();
          dk[j] = 1;
          for (int k = 1; k < j; ++k) {
            const float v = cost[l * 8 + (size_t)(k - 1)] + cost[r * 8 + (size_t)(j - k - 1)];
            if (v < dist[j]) { dist[j] = v; dk[j] = (uint8_t)k; }
          }
        }
        dec8[i] = dk[8];
        const bool contiguous = sub_first[l] + sub_count[l] == sub_first[r] || sub_first[r] + sub_count[r] == sub_first[l];
        const float as_leaf = (sub_count[i] <= max_leaf && contiguous) ? area * (float)sub_count[i] * c_prim : std::numeric_limits<float>::infinity();
        const float as_node = area * c_node + dist[8];
        c[0] = std::min(as_leaf, as_node);
        dec[i * 8] = as_leaf <= as_node ? 0 : 1;
        for (int j = 2; j <= 8; ++j) {
          if (dist[j] < c[j - 2]) { c[j - 1] = dist[j]; dec[i * 8 + (size_t)(j - 1)] = dk[j]; }
          else { c[j - 1] = c[j - 2]; dec[i * 8 + (size_t)(j - 1)] = 0; }
        }
      }
    }
    struct Wide8 { int child[8]; int n; };   // child: build-node id; is_leaf says whether its whole subtree is one leaf
    std::vector<Wide8> wide;
    std::vector<uint8_t> child_is_leaf;      // [wide node * 8 + slot]
    std::vector<int> wide_root;
    std::vector<uint32_t> wide_depth;
    wide_root.push_back(0);
    wide_depth.push_back(1);
    for (size_t w = 0; w < wide_root.size(); ++w) {      // breadth-first: the top of the tree is contiguous
      Wide8 wn;
      wn.n = 0;
      bool leaf_flag[8] = { false, false, false, false, false, false, false, false };
      // hand `slots` child slots to the subtree of build node `id`
      struct Item { int id; int slots; };
      std::vector<Item> todo;
      const BuildNode& root = b.nodes[(size_t)wide_root[w]];
      if (root.left < 0) { leaf_flag[wn.n] = true; wn.child[wn.n++] = wide_root[w]; }
      else {
        const int k = dec8[(size_t)wide_root[w]];
        todo.push_back({ root.right, 8 - k });
        todo.push_back({ root.left, k });
      }
      while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        const BuildNode& bn = b.nodes[(size_t)it.id];
        if (bn.left < 0) { leaf_flag[wn.n] = true; wn.child[wn.n++] = it.id; continue; }
        int slots = it.slots;
        while (slots > 1 && dec[(size_t)it.id * 8 + (size_t)(slots - 1)] == 0) --slots;   // "as with one slot fewer"
        if (slots == 1) {
          leaf_flag[wn.n] = dec[(size_t)it.id * 8] == 0;     // merged leaf, or a wide node of its own
          wn.child[wn.n++] = it.id;
          continue;
        }
        const int k = dec[(size_t)it.id * 8 + (size_t)(slots - 1)];
        todo.push_back({ bn.right, slots - k });
        todo.push_back({ bn.left, k });
      }
      for (int i = wn.n; i < 8; ++i) wn.child[i] = -1;
      // Slot assignment (after Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide
      // BVHs", 2017): slot s stands for the diagonal direction (+-1, +-1, +-1) whose sign bits are s; a child goes to the slot
      // whose direction matches its offset from the node's centre best, so that for a ray of octant o (bit a: dir[a] < 0)
      // ascending (slot ^ o) is a front-to-back order — no per-octant table in the node.  Exact assignment by dynamic
      // programming over slot subsets (8 x 256 states).
      {
        Box nb;
        nb.reset();
        for (int i = 0; i < wn.n; ++i) nb.grow(b.nodes[(size_t)wn.child[i]].box);
        float cost[8][8];
        for (int i = 0; i < 8; ++i)
          for (int sl = 0; sl < 8; ++sl) {
            float v = 0.0f;
            if (i < wn.n) {
              const Box& cb = b.nodes[(size_t)wn.child[i]].box;
              for (int a = 0; a < 3; ++a) {
                float off = (0.5f * cb.lo[a] + 0.5f * cb.hi[a]) - (0.5f * nb.lo[a] + 0.5f * nb.hi[a]);
                // (a box with an infinite or NaN face has no meaningful offset: it takes whatever slot is left.  A NaN here
                // would win no comparison below and leave the assignment undefined.)
                if (!(std::fabs(off) <= std::numeric_limits<float>::max())) off = 0.0f;
                v += ((sl >> a) & 1) ? off : -off;
              }
            }
            cost[i][sl] = v;
          }
        float best[256];
        int8_t from[8][256];
        std::memset(from, -1, sizeof from);
        for (int m = 0; m < 256; ++m) best[m] = -std::numeric_limits<float>::infinity();
        best[0] = 0.0f;
        // children are placed in index order: after i children the used-slot mask has i bits
        for (int m = 0; m < 256; ++m) {
          const int i = __builtin_popcount((unsigned)m);
          if (i >= 8 || best[m] == -std::numeric_limits<float>::infinity()) continue;
          for (int sl = 0; sl < 8; ++sl) {
            if ((m >> sl) & 1) continue;
            const float v = best[m] + cost[i][sl];
            const int m2 = m | (1 << sl);
            if (v > best[m2]) { best[m2] = v; from[i][m2] = (int8_t)sl; }
          }
        }
        int slot_of[8];
        for (int i = 7, m = 255; i >= 0; --i) {
          int sl = from[i][m];
          if (sl < 0 || !((m >> sl) & 1)) sl = __builtin_ctz((unsigned)m);   // (cannot happen with finite costs: any slot still free)
          slot_of[i] = sl;
          m &= ~(1 << sl);
        }
        int placed[8];
        bool placed_leaf[8];
        for (int sl = 0; sl < 8; ++sl) { placed[sl] = -1; placed_leaf[sl] = false; }
        for (int i = 0; i < wn.n; ++i) { placed[slot_of[i]] = wn.child[i]; placed_leaf[slot_of[i]] = leaf_flag[i]; }
        for (int sl = 0; sl < 8; ++sl) { wn.child[sl] = placed[sl]; leaf_flag[sl] = placed_leaf[sl]; }
      }
      wide.push_back(wn);
      for (int sl = 0; sl < 8; ++sl) child_is_leaf.push_back(leaf_flag[sl] ? 1 : 0);
      out.depth8 = std::max(out.depth8, wide_depth[w]);
      for (int i = 0; i < 8; ++i)
        if (wn.child[i] >= 0 && !leaf_flag[i]) {
          wide_root.push_back(wn.child[i]);
          wide_depth.push_back(wide_depth[w] + 1);
        }
    }
    std::vector<int> wide_of(b.nodes.size(), -1);
    for (size_t w = 0; w < wide_root.size(); ++w) wide_of[(size_t)wide_root[w]] = (int)w;
    out.n_nodes8 = (uint32_t)wide.size();
    out.nodes8.assign((size_t)out.n_nodes8 * 32, 0u);
    // the 8-bit planes cost up to one quantisation step of slack per face; the slab arithmetic has two more roundings than
    // the plain form (scale * 1/d, (origin - o) / d): the boxes are inflated by twice the origin-dependent margin first
    const float m8 = margin + 2.0f * origin_margin;
    for (size_t w = 0; w < wide.size(); ++w) {
      uint32_t* q = &out.nodes8[w * 32];
      float lo_c[8][3], hi_c[8][3];
      float nlo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, nhi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
      for (int c = 0; c < 8; ++c) {
        if (wide[w].child[c] < 0) continue;
        const BuildNode& cn = b.nodes[(size_t)wide[w].child[c]];
        for (int a = 0; a < 3; ++a) {
          // (an infinite face is moved to the largest finite value: the node's origin and grid must stay finite — with an
          // origin of -inf every plane of the node decodes to NaN and the whole subtree is missed)
          lo_c[c][a] = std::max(-3.0e38f, std::min(3.0e38f, cn.box.lo[a] - (m8 + std::fabs(cn.box.lo[a]) * 1e-6f)));
          hi_c[c][a] = std::max(-3.0e38f, std::min(3.0e38f, cn.box.hi[a] + (m8 + std::fabs(cn.box.hi[a]) * 1e-6f)));
          nlo[a] = std::min(nlo[a], lo_c[c][a]);
          nhi[a] = std::max(nhi[a], hi_c[c][a]);
        }
      }
      uint32_t expo[3];
      float scale[3];
      for (int a = 0; a < 3; ++a) {
        // smallest power of two with (extent / scale) <= 255, kept inside the normal range; NaN / inf extents: the largest
        int e = 0;
        const float ext = nhi[a] - nlo[a];
        if (ext > 0.0f && ext <= std::numeric_limits<float>::max()) { (void)std::frexp(ext / 255.0f, &e); }   // ext / 255 = m * 2^e, m in [0.5, 1) -> 2^e >= ext / 255
        else if (!(ext <= std::numeric_limits<float>::max())) e = 120;
        else e = -120;
        e = std::max(-120, std::min(120, e));
        expo[a] = (uint32_t)(e + 127);
        scale[a] = std::ldexp(1.0f, e);
        q[a] = f2u(nlo[a]);
      }
      q[3] = expo[0] | (expo[1] << 8) | (expo[2] << 16) | ((uint32_t)wide[w].n << 24);
      uint8_t qlo[3][8], qhi[3][8];
      for (int c = 0; c < 8; ++c) {
        uint32_t ref = 0xFFFFFFFFu;
        if (wide[w].child[c] < 0) {
          for (int a = 0; a < 3; ++a) { qlo[a][c] = 255; qhi[a][c] = 0; }   // inverted: no ray enters before it leaves
        } else {
          for (int a = 0; a < 3; ++a) {
            // rounded outward, then checked with the device's own decode: fma(q, scale, origin) must enclose the child
            int l = (int)std::floor((lo_c[c][a] - nlo[a]) / scale[a]);
            int h = (int)std::ceil((hi_c[c][a] - nlo[a]) / scale[a]);
            l = std::max(0, std::min(255, l));
            h = std::max(0, std::min(255, h));
            while (l > 0 && !(std::fma((float)l, scale[a], nlo[a]) <= lo_c[c][a])) --l;
            while (h < 255 && !(std::fma((float)h, scale[a], nlo[a]) >= hi_c[c][a])) ++h;
            qlo[a][c] = (uint8_t)l;
            qhi[a][c] = (uint8_t)h;
          }
          if (child_is_leaf[w * 8 + (size_t)c]) {
            // (possibly several binary leaves merged: their triangle records follow each other)
            const size_t id = (size_t)wide[w].child[c];
            ref = 0x80000000u | (sub_count[id] << 24) | sub_first[id];
            out.max_leaf8 = std::max(out.max_leaf8, sub_count[id]);
          } else {
            ref = (uint32_t)wide_of[(size_t)wide[w].child[c]];
          }
        }
        q[4 + c] = ref;
      }
      for (int a = 0; a < 3; ++a) {
        std::memcpy(&q[12 + a * 2], qlo[a], 8);
        std::memcpy(&q[18 + a * 2], qhi[a], 8);
      }
    }
  }
  return PTAMD_OK;
}

// Mirror of the device's four-wide walk (csrc/pt_kernels.hip: walk4_*): a stack of (reference, entry distance); a node's
// hit children are pushed farthest first in the node's order for the ray's octant; entries whose entry distance lies
// beyond the best hit are dropped when popped.  Result contract as for the binary walk.
static void bvh4_trace_impl(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                            uint64_t* tris_tested, bool quantised)
{
  const float MAX_DIST = 100000.0f;
  float best_t = MAX_DIST, best_u = 0.f, best_v = 0.f;
  uint32_t best_idx = 0xFFFFFFFFu;
  const int oct = (dir[0] < 0.f ? 1 : 0) | (dir[1] < 0.f ? 2 : 0) | (dir[2] < 0.f ? 4 : 0);
  // the quantised form takes the octant from the SIGN BITS (-0.0 counts as negative: its stand-in below is -1e-30, so the ray
  // enters through the high plane)
  const uint32_t soct = (std::signbit(dir[0]) ? 1u : 0u) | (std::signbit(dir[1]) ? 2u : 0u) | (std::signbit(dir[2]) ? 4u : 0u);
  float inv[3], noi[3];
  for (int a = 0; a < 3; ++a) {
    const float da = std::fabs(dir[a]) < 1e-30f ? std::copysign(1e-30f, dir[a]) : dir[a];
    inv[a] = 1.0f / da;
    noi[a] = -(origin[a] * inv[a]);
  }
  struct Entry { uint32_t ref; float tnear; };
  std::vector<Entry> stack;
  if (bvh.n_nodes4) stack.push_back({ 0u, 0.0f });
  while (!stack.empty()) {
    const Entry e = stack.back();
    stack.pop_back();
    if (!(e.tnear <= best_t)) continue;
    if (e.ref & 0x80000000u) {
      const uint32_t first = e.ref & 0xFFFFFFu, count = (e.ref >> 24) & 0x7Fu;
      for (uint32_t k = 0; k < count; ++k) {
        const float* t = &bvh.tris[(size_t)(first + k) * 12];
        if (tris_tested) ++*tris_tested;
        const float e1[3] = { t[0], t[1], t[2] }, e2[3] = { t[3], t[4], t[5] }, v0[3] = { t[6], t[7], t[8] };
        const float p[3] = { dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0] };
        const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if (det < 1e-7f) continue;
        const float inv_det = 1.0f / det;
        const float tv[3] = { origin[0] - v0[0], origin[1] - v0[1], origin[2] - v0[2] };
        const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv_det;
        if (u < 0 || u > 1) continue;
        const float q[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
        const float v = (dir[0] * q[0] + dir[1] * q[1] + dir[2] * q[2]) * inv_det;
        if (v < 0 || u + v > 1) continue;
        const float tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv_det;
        uint32_t idx;
        std::memcpy(&idx, &t[9], 4);
        if (tt > 0.0f && (tt < best_t || (tt == best_t && idx < best_idx && best_idx != 0xFFFFFFFFu))) {
          best_t = tt; best_u = u; best_v = v; best_idx = idx;
        }
      }
      continue;
    }
    if (nodes_visited) {
      ++*nodes_visited;
      if (e.ref < 85u) ++nodes_visited[3];    // counters[3], [4] of ptamd_host_bvh4_trace: visits to the first 85 / 341 nodes
      if (e.ref < 341u) ++nodes_visited[4];   // (breadth-first numbering: the top four / five levels of a full tree)
    }
    uint32_t hit = 0;
    float tn[4];
    uint32_t order, refs[4];
    int self_counted = 0;   // quantised form, octants 4..7: the stored order is the opposite octant's, read inverted
    if (!quantised) {
      const float* q = &bvh.nodes4[(size_t)e.ref * 32];
      for (int c = 0; c < 4; ++c) {
        float tnear = -std::numeric_limits<float>::infinity(), tfar = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
          const float tc = std::fma(q[a * 4 + c], inv[a], noi[a]), ai = std::fabs(inv[a]);
          tnear = std::max(tnear, std::fma(-q[12 + a * 4 + c], ai, tc));
          tfar = std::min(tfar, std::fma(q[12 + a * 4 + c], ai, tc));
        }
        tn[c] = std::max(tnear, 0.0f);
        if (tn[c] <= std::min(tfar, best_t)) hit |= 1u << c;
        std::memcpy(&refs[c], &q[24 + c], 4);
      }
      uint32_t w;
      std::memcpy(&w, &q[28 + (oct >> 1)], 4);
      order = (w >> (16 * (oct & 1))) & 0xFFFFu;
    } else {
      // the device's operations (pt_kernels.hip: walk4q_visit): A = scale / d, B = fma(origin, 1/d, -o/d), t = fma(plane, A, B)
      const uint32_t* q = &bvh.nodes4q[(size_t)e.ref * 16];
      float A[3], B[3];
      for (int a = 0; a < 3; ++a) {
        float sc, org;
        const uint32_t sb = ((q[3] >> (8 * a)) & 0xFFu) << 23;
        std::memcpy(&sc, &sb, 4);
        std::memcpy(&org, &q[a], 4);
        A[a] = sc * inv[a];
        B[a] = std::fma(org, inv[a], noi[a]);
      }
      for (int c = 0; c < 4; ++c) {
        float tnear = -std::numeric_limits<float>::infinity(), tfar = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; ++a) {
          const uint32_t lo = (q[8 + a] >> (8 * c)) & 0xFFu, hi = (q[11 + a] >> (8 * c)) & 0xFFu;
          const bool neg = ((soct >> a) & 1u) != 0u;
          tnear = std::max(tnear, std::fma((float)(neg ? hi : lo), A[a], B[a]));
          tfar = std::min(tfar, std::fma((float)(neg ? lo : hi), A[a], B[a]));
        }
        tn[c] = std::max(tnear, 0.0f);
        if (tn[c] <= std::min(tfar, best_t)) hit |= 1u << c;
        refs[c] = q[4 + c];
      }
      const uint32_t h = (soct & 4u) ? (~soct & 3u) : (soct & 3u);
      order = (q[14 + (h >> 1)] >> (16 * (h & 1u))) & 0xFFFFu;
      if (soct & 4u) { order = ~order & 0xFFFFu; self_counted = 1; }
    }
    // farthest first: a child goes below every child that is nearer than it
    Entry pushed[4];
    const int nhit = __builtin_popcount(hit);
    for (int c = 0; c < 4; ++c) {
      if (!((hit >> c) & 1u)) continue;
      const int rank = __builtin_popcount(hit & ((order >> (4 * c)) & 0xFu)) - self_counted;   // hit children farther than c
      pushed[nhit - 1 - rank] = { refs[c], tn[c] };     // nearest last = on top
    }
    // pushed[] is in stack order: index 0 deepest (farthest)
    for (int i = 0; i < nhit; ++i) stack.push_back(pushed[i]);
  }
  out.kind = best_idx == 0xFFFFFFFFu ? 0 : 1;
  out.index = best_idx == 0xFFFFFFFFu ? -1 : (int32_t)best_idx;
  out.t = best_t; out.u = best_u; out.v = best_v;
}

void bvh4_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                     uint64_t* tris_tested)
{
  bvh4_trace_impl(bvh, dir, origin, out, nodes_visited, tris_tested, false);
}
// ... over the 64-byte quantised nodes (Bvh::nodes4q)
void bvh4q_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                      uint64_t* tris_tested)
{
  bvh4_trace_impl(bvh, dir, origin, out, nodes_visited, tris_tested, true);
}

// Mirror of the device's eight-wide walk (csrc/pt_kernels.hip: walk8_*): child boxes decoded from the node's origin, per-axis
// power-of-two scale and 8-bit planes with the device's operations (A = scale / d, B = fma(origin, 1/d, -o/d),
// t = fma(plane, A, B)); hit children stacked farthest first in ascending (slot ^ octant) order; entries beyond the best hit
// dropped when popped.  Result contract as for the other walks.
void bvh8_trace_host(const Bvh& bvh, const float dir[3], const float origin[3], HostHit& out, uint64_t* nodes_visited,
                     uint64_t* tris_tested)
{
  const float MAX_DIST = 100000.0f;
  float best_t = MAX_DIST, best_u = 0.f, best_v = 0.f;
  uint32_t best_idx = 0xFFFFFFFFu;
  // octant from the SIGN BITS (-0.0 counts as negative: its stand-in below is -1e-30, so the ray enters through the high plane)
  const uint32_t oct = (std::signbit(dir[0]) ? 1u : 0u) | (std::signbit(dir[1]) ? 2u : 0u) | (std::signbit(dir[2]) ? 4u : 0u);
  float inv[3], noi[3];
  for (int a = 0; a < 3; ++a) {
    const float da = std::fabs(dir[a]) < 1e-30f ? std::copysign(1e-30f, dir[a]) : dir[a];
    inv[a] = 1.0f / da;
    noi[a] = -(origin[a] * inv[a]);
  }
  struct Entry { uint32_t ref; float tnear; };
  std::vector<Entry> stack;
  if (bvh.n_nodes8) stack.push_back({ 0u, 0.0f });
  while (!stack.empty()) {
    const Entry e = stack.back();
    stack.pop_back();
    if (!(e.tnear <= best_t)) continue;
    if (e.ref & 0x80000000u) {
      const uint32_t first = e.ref & 0xFFFFFFu, count = (e.ref >> 24) & 0x7Fu;
      for (uint32_t k = 0; k < count; ++k) {
        const float* t = &bvh.tris[(size_t)(first + k) * 12];
        if (tris_tested) ++*tris_tested;
        const float e1[3] = { t[0], t[1], t[2] }, e2[3] = { t[3], t[4], t[5] }, v0[3] = { t[6], t[7], t[8] };
        const float p[3] = { dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0] };
        const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if (det < 1e-7f) continue;
        const float inv_det = 1.0f / det;
        const float tv[3] = { origin[0] - v0[0], origin[1] - v0[1], origin[2] - v0[2] };
        const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv_det;
        if (u < 0 || u > 1) continue;
        const float qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
        const float v = (dir[0] * qv[0] + dir[1] * qv[1] + dir[2] * qv[2]) * inv_det;
        if (v < 0 || u + v > 1) continue;
        const float tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * inv_det;
        uint32_t idx;
        std::memcpy(&idx, &t[9], 4);
        if (tt > 0.0f && (tt < best_t || (tt == best_t && idx < best_idx && best_idx != 0xFFFFFFFFu))) {
          best_t = tt; best_u = u; best_v = v; best_idx = idx;
        }
      }
      continue;
    }
    const uint32_t* q = &bvh.nodes8[(size_t)e.ref * 32];
    if (nodes_visited) {
      ++*nodes_visited;
      if (e.ref < 73u) ++nodes_visited[3];    // counters[3], [4]: visits to the first 73 / 585 nodes (three / four full levels)
      if (e.ref < 585u) ++nodes_visited[4];
    }
    float A[3], B[3];
    for (int a = 0; a < 3; ++a) {
      const float scale = u2f(((q[3] >> (8 * a)) & 0xFFu) << 23);
      A[a] = scale * inv[a];
      B[a] = std::fma(u2f(q[a]), inv[a], noi[a]);
    }
    const uint8_t* planes = reinterpret_cast<const uint8_t*>(q + 12);   // lo.x[8] lo.y[8] lo.z[8] hi.x[8] hi.y[8] hi.z[8]
    uint32_t hit = 0;
    float tn[8];
    for (int c = 0; c < 8; ++c) {
      float tnear = 0.0f, tfar = best_t;
      for (int a = 0; a < 3; ++a) {
        const float tl = std::fma((float)planes[a * 8 + c], A[a], B[a]), th = std::fma((float)planes[24 + a * 8 + c], A[a], B[a]);
        const bool neg = (oct >> a) & 1u;     // the ray runs against this axis: it enters through the high plane
        tnear = std::max(tnear, neg ? th : tl);
        tfar = std::min(tfar, neg ? tl : th);
      }
      tn[c] = tnear;
      if (tnear <= tfar) hit |= 1u << c;
    }
    // stack order: farthest first = descending (slot ^ octant); the nearest hit child ends on top
    for (int f = 7; f >= 0; --f) {
      const int c = f ^ (int)oct;
      if ((hit >> c) & 1u) stack.push_back({ q[4 + c], tn[c] });
    }
  }
  out.kind = best_idx == 0xFFFFFFFFu ? 0 : 1;
  out.index = best_idx == 0xFFFFFFFFu ? -1 : (int32_t)best_idx;
  out.t = best_t; out.u = best_u; out.v = best_v;
}

// Mirror of the device traversal (csrc/pt_kernels.hip: traverse_bvh); float ops in the same
// order.  Only the final (kind, index, t) has to agree with brute force — the set of
// visited nodes is an implementation detail.
void bvh_trace_host(const Bvh& bvh, const ptamd_face* faces, const float dir[3], const float origin[3],
                    HostHit& out, uint64_t* nodes_visited, uint64_t* tris_tested)
{
  (void)faces;
  const float MAX_DIST = 100000.0f;
  float best_t = MAX_DIST, best_u = 0.f, best_v = 0.f;
  uint32_t best_idx = 0xFFFFFFFFu;
  const int oct = (dir[0] < 0.f ? 1 : 0) | (dir[1] < 0.f ? 2 : 0) | (dir[2] < 0.f ? 4 : 0);
  // same slab formulation as the kernel (fma of lo/hi with 1/d and -o/d; zero components replaced
  // by a tiny stand-in).  The device uses v_rcp_f32 (1 ulp) where this uses an exact division;
  // the visited set may differ by a node, the result may not (conservative boxes).
  float inv[3], noi[3];
  for (int a = 0; a < 3; ++a) {
    const float da = std::fabs(dir[a]) < 1e-30f ? std::copysign(1e-30f, dir[a]) : dir[a];
    inv[a] = 1.0f / da;
    noi[a] = -(origin[a] * inv[a]);
  }
  uint32_t node = bvh.n_nodes ? 0u : 0xFFFFFFFFu;
  while (node != 0xFFFFFFFFu) {
    const float* q = &bvh.nodes[(size_t)node * 16];
    if (nodes_visited) ++*nodes_visited;
    float tnear = -std::numeric_limits<float>::infinity(), tfar = std::numeric_limits<float>::infinity();
    for (int a = 0; a < 3; ++a) {
      float t0 = std::fma(q[a], inv[a], noi[a]);
      float t1 = std::fma(q[4 + a], inv[a], noi[a]);
      tnear = std::fmax(tnear, std::fmin(t0, t1));
      tfar = std::fmin(tfar, std::fmax(t0, t1));
    }
    const bool hit = tnear <= tfar && tfar >= 0.0f && tnear <= best_t;
    const uint32_t info = f2u(q[3]);
    const uint32_t miss = f2u(q[8 + oct]);
    if (!hit) { node = miss; continue; }
    const uint32_t count = info >> 24;
    if (count == 0) {
      const uint32_t child = f2u(q[7]);
      const uint32_t right = child & 0x3FFFFFFFu, axis = child >> 30;
      node = ((oct >> axis) & 1) ? right : node + 1;
      continue;
    }
    const uint32_t first = info & 0xFFFFFFu;
    for (uint32_t k = 0; k < count; ++k) {
      const float* t = &bvh.tris[(size_t)(first + k) * 12];
      if (tris_tested) ++*tris_tested;
      // intersection.cuh:102-135, same operation order
      const float e1x = t[0], e1y = t[1], e1z = t[2], e2x = t[3], e2y = t[4], e2z = t[5];
      const float px = dir[1] * e2z - dir[2] * e2y;
      const float py = dir[2] * e2x - dir[0] * e2z;
      const float pz = dir[0] * e2y - dir[1] * e2x;
      const float det = e1x * px + e1y * py + e1z * pz;
      if (det < 1e-7f) continue; // == (double)det < 0.0000001 (intersection.cuh:110), see mt_test
      const float inv_det = 1.0f / det;
      const float tx = origin[0] - t[6], ty = origin[1] - t[7], tz = origin[2] - t[8];
      const float u = (tx * px + ty * py + tz * pz) * inv_det;
      if (u < 0 || u > 1) continue;
      const float qx = ty * e1z - tz * e1y;
      const float qy = tz * e1x - tx * e1z;
      const float qz = tx * e1y - ty * e1x;
      const float v = (dir[0] * qx + dir[1] * qy + dir[2] * qz) * inv_det;
      if (v < 0 || u + v > 1) continue;
      const float tt = (e2x * qx + e2y * qy + e2z * qz) * inv_det;
      const uint32_t idx = f2u(t[9]);
      if (tt > 0.0f && (tt < best_t || (tt == best_t && idx < best_idx && best_idx != 0xFFFFFFFFu))) {
        best_t = tt; best_idx = idx; best_u = u; best_v = v;
      }
    }
    node = miss;
  }
  out.t = best_t;
  out.u = best_u;
  out.v = best_v;
  if (best_idx == 0xFFFFFFFFu) { out.kind = 0; out.index = -1; }
  else { out.kind = 1; out.index = (int32_t)best_idx; }
}

} // namespace ptamd

extern "C" int ptamd_host_bvh8_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                                     int32_t* out, uint64_t* counters)
{
  if ((n_faces && !faces) || (n && (!rays || !out))) { ptamd::set_error("ptamd_host_bvh8_trace: null argument"); return PTAMD_ERR_ARG; }
  ptamd::Bvh bvh;
  int rc = ptamd::build_bvh(faces, n_faces, 1e-3f, 2, bvh);
  if (rc != PTAMD_OK) return rc;
  for (uint32_t i = 0; i < n; ++i) {
    ptamd::HostHit h;
    ptamd::bvh8_trace_host(bvh, rays + (size_t)i * 6, rays + (size_t)i * 6 + 3, h,
                           counters ? &counters[0] : nullptr, counters ? &counters[1] : nullptr);
    out[i * 4 + 0] = h.kind;
    out[i * 4 + 1] = h.index;
    std::memcpy(&out[i * 4 + 2], &h.t, 4);
    out[i * 4 + 3] = 0;
  }
  if (counters) { counters[2] = bvh.depth8; counters[5] = bvh.n_nodes8; }
  return PTAMD_OK;
}

extern "C" int ptamd_host_bvh4_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                                     int32_t* out, uint64_t* counters)
{
  if ((n_faces && !faces) || (n && (!rays || !out))) { ptamd::set_error("ptamd_host_bvh4_trace: null argument"); return PTAMD_ERR_ARG; }
  ptamd::Bvh bvh;
  int rc = ptamd::build_bvh(faces, n_faces, 1e-3f, 4, bvh);
  if (rc != PTAMD_OK) return rc;
  for (uint32_t i = 0; i < n; ++i) {
    ptamd::HostHit h;
    ptamd::bvh4_trace_host(bvh, rays + (size_t)i * 6, rays + (size_t)i * 6 + 3, h,
                           counters ? &counters[0] : nullptr, counters ? &counters[1] : nullptr);
    out[i * 4 + 0] = h.kind;
    out[i * 4 + 1] = h.index;
    std::memcpy(&out[i * 4 + 2], &h.t, 4);
    out[i * 4 + 3] = 0;
  }
  if (counters) counters[2] = bvh.depth4;
  return PTAMD_OK;
}

extern "C" int ptamd_host_bvh4q_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                                      int32_t* out, uint64_t* counters)
{
  if ((n_faces && !faces) || (n && (!rays || !out))) { ptamd::set_error("ptamd_host_bvh4q_trace: null argument"); return PTAMD_ERR_ARG; }
  ptamd::Bvh bvh;
  int rc = ptamd::build_bvh(faces, n_faces, 1e-3f, 2, bvh);
  if (rc != PTAMD_OK) return rc;
  for (uint32_t i = 0; i < n; ++i) {
    ptamd::HostHit h;
    ptamd::bvh4q_trace_host(bvh, rays + (size_t)i * 6, rays + (size_t)i * 6 + 3, h,
                            counters ? &counters[0] : nullptr, counters ? &counters[1] : nullptr);
    out[i * 4 + 0] = h.kind;
    out[i * 4 + 1] = h.index;
    std::memcpy(&out[i * 4 + 2], &h.t, 4);
    out[i * 4 + 3] = 0;
  }
  if (counters) counters[2] = bvh.depth4;
  return PTAMD_OK;
}

extern "C" int ptamd_host_bvh_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                                    int32_t* out, uint64_t* counters)
{
  if ((n_faces && !faces) || (n && (!rays || !out))) { ptamd::set_error("ptamd_host_bvh_trace: null argument"); return PTAMD_ERR_ARG; }
  ptamd::Bvh bvh;
  int rc = ptamd::build_bvh(faces, n_faces, 1e-3f, 4, bvh);
  if (rc != PTAMD_OK) return rc;
  for (uint32_t i = 0; i < n; ++i) {
    ptamd::HostHit h;
    ptamd::bvh_trace_host(bvh, faces, rays + (size_t)i * 6, rays + (size_t)i * 6 + 3, h,
                          counters ? &counters[0] : nullptr, counters ? &counters[1] : nullptr);
    out[i * 4 + 0] = h.kind;
    out[i * 4 + 1] = h.index;
    std::memcpy(&out[i * 4 + 2], &h.t, 4);
    out[i * 4 + 3] = 0;
  }
  return PTAMD_OK;
}
