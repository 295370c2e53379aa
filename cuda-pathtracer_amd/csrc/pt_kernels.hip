// pt_kernels.hip — the gfx950 per-pixel path-tracing megakernel and its launchers.
//
// Replaces the reference's `__global__ kernel` (cuda_opengl/src/shaders/raytrace.cu:212-271)
// together with radiance() (:41-210) and intersect() (intersection.cuh:161-246).
//
// Shape of the kernel (MI355X-first, not a translation of the 16x16-thread CUDA launch):
//   * one wave64 owns an 8x8 pixel tile, four waves (a 16x16 tile) form a workgroup;
//   * at workgroup start the whole traversal working set — 64-byte BVH nodes and 48-byte
//     {v0,e1,e2} triangle records — is staged into LDS with 16-byte coalesced loads, so
//     the hot loop never touches HBM/L2 (scenes that do not fit fall back to L2-resident
//     global reads through the same code);
//   * nearest hit = stackless, per-octant ORDERED threaded BVH walk ("while-while": all
//     lanes walk boxes until each has a leaf, then all lanes test triangles), conservative
//     boxes, (t, global face index) lexicographic minimum == the reference's first-wins
//     brute-force loop;  the brute-force loop itself is kept as a second variant with
//     wave-uniform LDS broadcast reads (zero traversal divergence);
//   * shading data (normals/uvs/material) is fetched from L2 only for the winning face;
//   * a ray that missed cannot hit later (the reference keeps looping with the same ray,
//     raytrace.cu:194-199), so its remaining iterations skip the walk — exact.
#include "pt_device.h"
#include "pt_launch.h"

namespace ptamd {

#define PT_MAX_DIST 100000.0f
#define PT_END 0xFFFFFFFFu

// ---------------------------------------------------------------- nearest hit

struct Best { float t, u, v; uint32_t idx; };

// intersection.cuh:102-135 on a {v0,e1,e2} record; identical operation order.
// Accept rule: reference `t < best && t > 0` in storage order == lexicographic (t, idx).
template <bool ORDERED>
PT_DEV void mt_test(float4 a, float4 b, float4 c, f3 o, f3 d, Best& best)
{
  const f3 v0 = mk3(a.x, a.y, a.z);
  const f3 e1 = mk3(a.w, b.x, b.y);
  const f3 e2 = mk3(b.z, b.w, c.x);
  const f3 p_vec = cross(d, e2);
  const float det = dot(e1, p_vec);
  if ((double)det < 0.0000001) return;
  const float inv_det = 1.0f / det;
  const f3 t_vec = o - v0;
  const float u = dot(t_vec, p_vec) * inv_det;
  if (u < 0 || u > 1) return;
  const f3 qvec = cross(t_vec, e1);
  const float v = dot(d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return;
  const float t = dot(e2, qvec) * inv_det;
  const uint32_t idx = f_as_u(c.y);
  bool take;
  if (ORDERED) take = t < best.t && t > 0.0f;
  else take = t > 0.0f && (t < best.t || (t == best.t && idx < best.idx && best.idx != PT_END));
  if (take) { best.t = t; best.u = u; best.v = v; best.idx = idx; }
}

// The reference algorithm: every face in storage order.  `tris` is LDS (or global); the
// index is wave-uniform so LDS reads are broadcasts.
template <bool STATS>
PT_DEV void traverse_brute(const float4* tris, uint32_t n_faces, f3 o, f3 d, Best& best,
                           uint32_t& n_tris)
{
  for (uint32_t i = 0; i < n_faces; ++i) {
    const float4 a = tris[i * 3 + 0], b = tris[i * 3 + 1], c = tris[i * 3 + 2];
    mt_test<true>(a, b, c, o, d, best);
  }
  if (STATS) n_tris += n_faces;
}

// Stackless ordered threaded BVH walk (layout: host/ptamd_internal.h).
template <bool STATS>
PT_DEV void traverse_bvh(const float4* nodes, const float4* tris, uint32_t n_nodes, f3 o, f3 d,
                         Best& best, uint32_t& n_nodes_visited, uint32_t& n_tris)
{
  const uint32_t oct = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
  const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const float* links = reinterpret_cast<const float*>(nodes) + 8 + oct;
  uint32_t node = n_nodes ? 0u : PT_END;
  for (;;) {
    uint32_t leaf_first = 0, leaf_count = 0;
    while (node != PT_END) {
      const float4 q0 = nodes[node * 4 + 0];
      const float4 q1 = nodes[node * 4 + 1];
      const uint32_t miss = f_as_u(links[node * 16]);
      if (STATS) ++n_nodes_visited;
      const float t0x = (q0.x - o.x) * inv.x, t1x = (q1.x - o.x) * inv.x;
      const float t0y = (q0.y - o.y) * inv.y, t1y = (q1.y - o.y) * inv.y;
      const float t0z = (q0.z - o.z) * inv.z, t1z = (q1.z - o.z) * inv.z;
      const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                          __builtin_fminf(t0z, t1z));
      const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                         __builtin_fmaxf(t0z, t1z));
      const bool hit = tnear <= tfar * 1.0000005f && tfar >= 0.0f && tnear <= best.t;
      const uint32_t info = f_as_u(q0.w);
      const uint32_t count = info >> 24;
      if (hit && count) {
        leaf_first = info & 0xFFFFFFu;
        leaf_count = count;
        node = miss;
        break;
      }
      const uint32_t child = f_as_u(q1.w);
      const uint32_t down = ((oct >> (child >> 30)) & 1u) ? (child & 0x3FFFFFFFu) : node + 1u;
      node = hit ? down : miss;
    }
    if (leaf_count == 0) break;
    for (uint32_t k = 0; k < leaf_count; ++k) {
      const uint32_t ti = (leaf_first + k) * 3;
      mt_test<false>(tris[ti], tris[ti + 1], tris[ti + 2], o, d, best);
    }
    if (STATS) n_tris += leaf_count;
  }
}

// intersection.cuh:140-155 (see the oracle's note on the discarded conditional at :152)
PT_DEV bool intersect_sphere(f3 o, f3 d, f3 center, float radius, float& t)
{
  const float epsilon = 0.01f;
  const f3 op = center - o;
  const float b = dot(op, d);
  float disc = b * b - dot(op, op) + radius * radius;
  if (disc < 0.0f) return false;
  disc = __builtin_sqrtf(disc);
  t = b - disc;
  if (!(t > epsilon)) t = b + disc;
  return t != 0.0f;
}

// intersection.cuh:20-26
PT_DEV int texture_idx(const TexDesc& tex, float uvx, float uvy)
{
  int x = (int)(uvx * (float)(tex.w - 1));
  int y = (int)(uvy * (float)(tex.h - 1));
  return (y * tex.w + x) * tex.nb_chan;
}

struct Counters { uint32_t rays, nodes, tris, mesh_hits, nmap_hits; };

// intersection.cuh:161-246.  KIND: 1 brute force, 2 BVH.
template <int KIND, bool STATS>
PT_DEV bool intersect(const KParams& p, const float4* s_nodes, const float4* s_tris, f3 o, f3 d,
                      Hit& hit, Counters& cnt)
{
  Best best;
  best.t = PT_MAX_DIST; best.u = 0.f; best.v = 0.f; best.idx = PT_END;
  hit.dist = PT_MAX_DIST;
  if (STATS) cnt.rays++;
  if (KIND == 1) traverse_brute<STATS>(s_tris, p.n_faces, o, d, best, cnt.tris);
  else traverse_bvh<STATS>(s_nodes, s_tris, p.n_nodes, o, d, best, cnt.nodes, cnt.tris);

  const bool mesh_hit = best.idx != PT_END;
  f3 surface_normal = mk3(0.f), tangent = mk3(0.f);
  float uvx = 0.f, uvy = 0.f;
  int mat_id = -1;
  if (mesh_hit) {
    // deferred part of intersectTriangle (intersection.cuh:124-131) for the winner only
    const float4* sh = p.shade + (size_t)best.idx * 5;
    const float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3], s4 = sh[4];
    const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
    const float u = best.u, v = best.v;
    const float w = 1.0f - u - v;
    const f3 nrm = w * n0 + u * n1 + v * n2;
    const float ux = w * s2.y + u * s2.w + v * s3.y;
    const float uy = w * s2.z + u * s3.x + v * s3.z;
    uvx = ux - __builtin_floorf(ux / 1.0f); // mod(uv, 1.0): cutils_math.h:1728-1737
    uvy = uy - __builtin_floorf(uy / 1.0f);
    tangent = mk3(s3.w, s4.x, s4.y);
    mat_id = (int)f_as_u(s4.z);
    hit.normal = nrm;
    surface_normal = nrm;
    hit.dist = best.t;
    hit.light = -1;
  }

  for (uint32_t l = 0; l < p.n_lights; ++l) {
    const float4 la = p.lights[l * 2 + 0], lb = p.lights[l * 2 + 1];
    const f3 center = mk3(la.w, lb.x, lb.y);
    float t;
    if (intersect_sphere(o, d, center, lb.w, t) && t < hit.dist && t >= 0.0f) {
      hit.light = (int)l;
      hit.dist = t;
      hit.diffuse_col = mk3(la.x, la.y, la.z);
      hit.normal = normalize(center - (t * d)); // intersection.cuh:208: origin ignored
      mat_id = -1;
    }
  }

  if (mat_id >= 0) {
    const int4 m = p.materials[mat_id];
    hit.ior = u_as_f((uint32_t)m.z);
    const TexDesc tex = p.textures[m.x];
    const float* texel = p.texels + tex.offset + texture_idx(tex, uvx, uvy);
    hit.diffuse_col = mk3(texel[0], texel[1], texel[2]);
    hit.specular_col = texel[3];
    if (STATS) cnt.mesh_hits++;
    if (m.y >= 0) {
      const TexDesc nt = p.textures[m.y];
      const float* nx = p.texels + nt.offset + texture_idx(nt, uvx, uvy);
      const f3 n = normalize((mk3(nx[0], nx[1], nx[2]) * 2.0f) - 1.0f);
      const f3 binormal = normalize(cross(tangent, surface_normal));
      const f3 tx = tangent, ty = -binormal, tz = surface_normal;
      hit.normal = mk3(tx.x * n.x + ty.x * n.y + tz.x * n.z,
                       tx.y * n.x + ty.y * n.y + tz.y * n.z,
                       tx.z * n.x + ty.z * n.y + tz.z * n.z);
      if (STATS) cnt.nmap_hits++;
    }
  }
  return hit.dist < PT_MAX_DIST;
}

// texCubemap restated (see oracle/pt_oracle.c: or_tex_cubemap for the definition)
PT_DEV f3 env_lookup(const KParams& p, f3 dir)
{
  const float x = dir.x, y = dir.y, z = -dir.z; // raytrace.cu:60,197
  const uint32_t n = p.cubemap_size;
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), az = __builtin_fabsf(z);
  int face;
  float m, s, t;
  if (ax >= ay && ax >= az) {
    m = ax;
    if (x >= 0.0f) { face = 0; s = -z; t = -y; } else { face = 1; s = z; t = -y; }
  } else if (ay >= az) {
    m = ay;
    if (y >= 0.0f) { face = 2; s = x; t = z; } else { face = 3; s = x; t = -z; }
  } else {
    m = az;
    if (z >= 0.0f) { face = 4; s = x; t = -y; } else { face = 5; s = -x; t = -y; }
  }
  const float4* base = p.cubemap + (size_t)face * n * n;
  if (n == 1) { const float4 c = base[0]; return mk3(c.x, c.y, c.z); }
  const float u = (s / m + 1.0f) * 0.5f;
  const float v = (t / m + 1.0f) * 0.5f;
  const float xb = u * (float)n - 0.5f;
  const float yb = v * (float)n - 0.5f;
  const float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb);
  float a = __builtin_floorf((xb - fx) * 256.0f) * (1.0f / 256.0f);
  float b = __builtin_floorf((yb - fy) * 256.0f) * (1.0f / 256.0f);
  int i0 = (int)fx, j0 = (int)fy, i1 = i0 + 1, j1 = j0 + 1;
  const int hi = (int)n - 1;
  if (!(xb == xb)) { i0 = i1 = 0; a = 0.0f; }
  if (!(yb == yb)) { j0 = j1 = 0; b = 0.0f; }
  i0 = i0 < 0 ? 0 : (i0 > hi ? hi : i0);
  i1 = i1 < 0 ? 0 : (i1 > hi ? hi : i1);
  j0 = j0 < 0 ? 0 : (j0 > hi ? hi : j0);
  j1 = j1 < 0 ? 0 : (j1 > hi ? hi : j1);
  const float4 t00 = base[(size_t)j0 * n + i0], t10 = base[(size_t)j0 * n + i1];
  const float4 t01 = base[(size_t)j1 * n + i0], t11 = base[(size_t)j1 * n + i1];
  const f3 top = mk3(t00.x, t00.y, t00.z) * (1.0f - a) + mk3(t10.x, t10.y, t10.z) * a;
  const f3 bot = mk3(t01.x, t01.y, t01.z) * (1.0f - a) + mk3(t11.x, t11.y, t11.z) * a;
  return top * (1.0f - b) + bot * b;
}

// ---------------------------------------------------------------- radiance (raytrace.cu:41-210)

template <int KIND, bool STATS>
PT_DEV f3 radiance(const KParams& p, const float4* s_nodes, const float4* s_tris, f3 o, f3 d,
                   Xorwow& rng, Counters& cnt)
{
  f3 acc = mk3(0.0f);
  f3 throughput = mk3(1.0f);
  Hit inter; // value-initialised + carried over
  inter.normal = mk3(0.f); inter.diffuse_col = mk3(0.f);
  inter.dist = 0.f; inter.specular_col = 0.f; inter.ior = 0.f; inter.light = -1;

  if (!p.is_static) { // raytrace.cu:54-62
    if (intersect<KIND, STATS>(p, s_nodes, s_tris, o, d, inter, cnt)) return inter.diffuse_col;
    return env_lookup(p, d);
  }

  bool missed = false;
  const int max_bounces = p.bounces;
  for (int b = 0; b < max_bounces; b++) {
    const float r1 = xorwow_uniform(rng);
    bool found = false;
    if (!missed) {
      found = intersect<KIND, STATS>(p, s_nodes, s_tris, o, d, inter, cnt);
      missed = !found;
    } else if (STATS) {
      cnt.rays++; // the reference issues this (futile) intersect() call; count it as a ray
    }
    if (found) {
      const float cos_theta = dot(inter.normal, d);
      f3 oriented_normal = inter.normal;
      const f3 spec = normalize(reflect(d, inter.normal));
      const f3 direct_light = inter.diffuse_col / 0.5f; // brdf_lambert / pdf_lambert (brdf.cuh:14-31)
      if (inter.ior == 1.0f || inter.light >= 0) {
        if (inter.light >= 0) {
          const float4 la = p.lights[inter.light * 2 + 0], lb = p.lights[inter.light * 2 + 1];
          acc = acc + (mk3(la.x, la.y, la.z) * lb.z) * throughput;
        }
        const float phi = (float)((double)2.0f * 3.14159265358979323846 * (double)xorwow_uniform(rng));
        const float sin_t = __builtin_sqrtf(r1);
        const float cos_t = __builtin_sqrtf(1.f - r1);
        const f3 axis = ((double)__builtin_fabsf(oriented_normal.x) > .1) ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
        const f3 u = normalize(cross(axis, oriented_normal));
        const f3 v = cross(oriented_normal, u);
        float sphi, cphi;
        pt_sincosf(phi, sphi, cphi);
        const f3 dd = normalize(v * sin_t * cphi + u * sphi * sin_t + oriented_normal * cos_t);
        o = o + d * inter.dist;
        d = mix(dd, spec, inter.specular_col);
        o = o + d * 0.03f;
        throughput = throughput * direct_light;
      } else {
        const float n1 = 1.0f;
        const float n2 = inter.ior;
        oriented_normal = cos_theta < 0 ? inter.normal : inter.normal * -1.0f;
        const float c1 = dot(oriented_normal, d);
        const bool entering = dot(inter.normal, oriented_normal) > 0;
        const float eta = entering ? n1 / n2 : n2 / n1;
        const float eta_2 = eta * eta;
        const float c2_term = 1.0f - eta_2 * (1.0f - c1 * c1);
        if (c2_term < 0.0f) {
          o = o + oriented_normal * inter.dist / 100.f;
          d = spec;
        } else {
          float R0 = (n2 - n1) / (n1 + n2);
          R0 *= R0;
          const float c2 = __builtin_sqrtf(c2_term);
          const f3 T = normalize(eta * d + (eta * c1 - c2) * oriented_normal);
          const float f_cos_theta = pt_powf(cos_theta, 5.0f);
          const float f_r = R0 + (1.0f - R0) * f_cos_theta;
          if (xorwow_uniform(rng) < 0.25f) {
            throughput = throughput * (f_r * direct_light);
            o = o + oriented_normal * inter.dist / 100.f;
            d = spec;
          } else {
            const float f_t = 1.0f - f_r;
            throughput = throughput * (f_t * direct_light);
            o = o + oriented_normal * inter.dist / 10000.f;
            d = T;
          }
        }
      }
    } else {
      acc = acc + env_lookup(p, d) * throughput;
    }
    const float pmax = __builtin_fmaxf(throughput.x, __builtin_fmaxf(throughput.y, throughput.z));
    if (r1 > pmax && b > 1) return acc;
    throughput = throughput * (1.0f / pmax);
  }
  return acc;
}

// ---------------------------------------------------------------- post process

PT_DEV f3 uncharted_tonemap(f3 x) // post_process.cuh:14-25
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}

PT_DEV f3 exposure(f3 color) // post_process.cuh:31-41
{
  const float exposure_bias = 2.0f;
  const f3 curr = uncharted_tonemap(exposure_bias * color);
  const f3 W = mk3(11.2f);
  const f3 white_scale = 1.0f / uncharted_tonemap(W);
  return curr * white_scale;
}

PT_DEV f3 post_process(uint32_t id, f3 c) // raytrace.cu:327-352
{
  if (id == 1) {
    const float gray = (float)((double)c.x * 0.3 + (double)c.y * 0.59 + (double)c.z * 0.11);
    return mk3(gray, gray, gray);
  }
  if (id == 2)
    return mk3((float)((double)c.x * 0.393 + (double)c.y * 0.769 + (double)c.z * 0.189),
               (float)((double)c.x * 0.349 + (double)c.y * 0.686 + (double)c.z * 0.168),
               (float)((double)c.x * 0.272 + (double)c.y * 0.534 + (double)c.z * 0.131));
  if (id == 3)
    return mk3((float)(1.0 - (double)c.x), (float)(1.0 - (double)c.y), (float)(1.0 - (double)c.z));
  return c;
}

// ---------------------------------------------------------------- LDS staging

// Copies n16 16-byte words global -> LDS with the whole workgroup, coalesced.
PT_DEV void stage_to_lds(float4* dst, const float4* src, uint32_t n16)
{
  for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------- the megakernel

// Workgroup = 256 threads = 4 waves; wave w owns the 8x8 tile (w&1, w>>1) of a 16x16 block.
template <int KIND, bool LDS_RESIDENT, bool STATS>
__global__ void __launch_bounds__(256) pt_megakernel(const KParams p)
{
  extern __shared__ float4 s_mem[];
  const float4* s_nodes;
  const float4* s_tris;
  if (LDS_RESIDENT) {
    if (KIND == 1) {
      stage_to_lds(s_mem, p.tris_brute, p.n_faces * 3);
      s_nodes = nullptr;
      s_tris = s_mem;
    } else {
      stage_to_lds(s_mem, p.nodes, p.n_nodes * 4);
      stage_to_lds(s_mem + p.n_nodes * 4, p.tris_bvh, p.n_faces * 3);
      s_nodes = s_mem;
      s_tris = s_mem + p.n_nodes * 4;
    }
    __syncthreads();
  } else {
    s_nodes = p.nodes;
    s_tris = KIND == 1 ? p.tris_brute : p.tris_bvh;
  }

  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t x = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t y = p.row_begin + blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
  Counters cnt;
  cnt.rays = cnt.nodes = cnt.tris = cnt.mesh_hits = cnt.nmap_hits = 0;
  const bool active = x < p.width && y < p.row_end;

  if (active) {
    // raytrace.cu:227-229 with the reference's launch geometry (16x16 blocks, padded grid)
    const uint32_t grid_x = p.width / 16u + 1u;
    const uint32_t tid = ((x >> 4) + (y >> 4) * grid_x) * 256u + (y & 15u) * 16u + (x & 15u);
    Xorwow rng;
    xorwow_init(rng, p.hash_seed + tid);

    // generateRay (intersection.cuh:75-97), pixel-invariant terms precomputed on the host
    const int half_w = (int)(p.width / 2u), half_h = (int)(p.height / 2u);
    const f3 screen_pos = (p.cam_p0 + (p.cam_u * (float)((int)x - half_w))) + (p.cam_v * (float)((int)y - half_h));
    f3 dir = normalize(screen_pos - p.cam_pos);
    f3 origin = p.cam_pos;

    // camera_dof (post_process.cuh:49-67)
    {
      const f3 focal_point = p.focus_dist * dir;
      const float random_angle = (float)((double)(xorwow_uniform(rng) * 2.0f) * 3.14159265358979323846);
      const float random_radius = xorwow_uniform(rng) * p.aperture;
      float sn, cs;
      pt_sincosf(random_angle, sn, cs);
      const f3 ap = (cs * p.cam_u + sn * p.cam_v) * random_radius;
      dir = normalize(focal_point - ap);
      origin = origin + ap;
    }

    f3 rad = radiance<KIND, STATS>(p, s_nodes, s_tris, origin, dir, rng, cnt);
    rad = mk3(clamp01(rad.x), clamp01(rad.y), clamp01(rad.z));

    // temporal accumulation (raytrace.cu:250-258), row-flipped index
    const size_t i = (size_t)(p.height - y - 1u - p.tfb_row0) * p.width + x;
    float* tp = p.tfb + i * 3;
    f3 t = mk3(tp[0], tp[1], tp[2]);
    t = t * (float)p.is_static;
    t = t + rad;
    tp[0] = t.x; tp[1] = t.y; tp[2] = t.z;
    rad = t / p.frame_nb_f;

    rad = exposure(rad);
    const float g = 1.0f / 2.2f;
    rad = mk3(pt_powf(rad.x, g), pt_powf(rad.y, g), pt_powf(rad.z, g));
    rad = post_process(p.post_id, rad);
    const uint32_t px = (pt_f2u(rad.x * 255.0f) & 0xffu) | ((pt_f2u(rad.y * 255.0f) & 0xffu) << 8) |
                        ((pt_f2u(rad.z * 255.0f) & 0xffu) << 16);
    p.surface[(size_t)(y - p.surf_row0) * p.width + x] = px;
  }

  if (STATS) {
    // wave reduction then one atomic per wave and counter
    unsigned long long v[6] = { cnt.rays, cnt.nodes, cnt.tris, cnt.mesh_hits, cnt.nmap_hits, active ? 1ull : 0ull };
    for (int k = 0; k < 6; ++k) {
      unsigned long long s = v[k];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0 && s) atomicAdd(&p.stats[k], s);
    }
  }
}

// Nearest-hit query on explicit rays (tests: BVH vs brute force on the device).
template <int KIND>
__global__ void __launch_bounds__(256) pt_trace_rays_kernel(const KParams p, const float* rays, uint32_t n, int4* out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f3 d = mk3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
  const f3 o = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
  Hit h;
  h.normal = mk3(0.f); h.diffuse_col = mk3(0.f); h.dist = 0.f; h.specular_col = 0.f; h.ior = 0.f; h.light = -1;
  Counters cnt;
  cnt.rays = cnt.nodes = cnt.tris = cnt.mesh_hits = cnt.nmap_hits = 0;
  Best best;
  best.t = PT_MAX_DIST; best.u = 0.f; best.v = 0.f; best.idx = PT_END;
  uint32_t a = 0, b = 0;
  if (KIND == 1) traverse_brute<false>(p.tris_brute, p.n_faces, o, d, best, b);
  else traverse_bvh<false>(p.nodes, p.tris_bvh, p.n_nodes, o, d, best, a, b);
  int kind = best.idx != PT_END ? 1 : 0;
  int index = kind ? (int)best.idx : -1;
  float t = best.t;
  for (uint32_t l = 0; l < p.n_lights; ++l) {
    const float4 la = p.lights[l * 2 + 0], lb = p.lights[l * 2 + 1];
    float ts;
    if (intersect_sphere(o, d, mk3(la.w, lb.x, lb.y), lb.w, ts) && ts < t && ts >= 0.0f) {
      kind = 2; index = (int)l; t = ts;
    }
  }
  out[i] = make_int4(kind, index, (int)f_as_u(t), 0);
}

// ---------------------------------------------------------------- launchers

template <int KIND, bool LDS_RES, bool STATS>
static hipError_t launch_variant(const KParams& p, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
  auto kern = pt_megakernel<KIND, LDS_RES, STATS>;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, stream, p);
  return hipGetLastError();
}

hipError_t launch_megakernel(const KParams& p, int kind, bool lds_resident, size_t lds_bytes, bool stats,
                             hipStream_t stream)
{
  const uint32_t rows = p.row_end - p.row_begin;
  if (rows == 0 || p.width == 0) return hipSuccess;
  dim3 grid((p.width + 15u) / 16u, (rows + 15u) / 16u);
  if (!lds_resident) lds_bytes = 0;
#define PT_DISPATCH(K, L, S) return launch_variant<K, L, S>(p, grid, lds_bytes, stream)
  if (kind == 1) {
    if (lds_resident) { if (stats) PT_DISPATCH(1, true, true); else PT_DISPATCH(1, true, false); }
    else { if (stats) PT_DISPATCH(1, false, true); else PT_DISPATCH(1, false, false); }
  } else {
    if (lds_resident) { if (stats) PT_DISPATCH(2, true, true); else PT_DISPATCH(2, true, false); }
    else { if (stats) PT_DISPATCH(2, false, true); else PT_DISPATCH(2, false, false); }
  }
#undef PT_DISPATCH
}

hipError_t launch_trace_rays(const KParams& p, int kind, const float* rays_dev, uint32_t n, int4* out_dev,
                             hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  dim3 grid((n + 255u) / 256u);
  if (kind == 1) hipLaunchKernelGGL(pt_trace_rays_kernel<1>, grid, dim3(256), 0, stream, p, rays_dev, n, out_dev);
  else hipLaunchKernelGGL(pt_trace_rays_kernel<2>, grid, dim3(256), 0, stream, p, rays_dev, n, out_dev);
  return hipGetLastError();
}

} // namespace ptamd
