// pt_device.h — device-side building blocks of the gfx950 path-tracing megakernel.
//
// Everything here executes the reference's per-pixel arithmetic
// (cuda_opengl/src/shaders/raytrace.cu:41-271, include/shaders/{intersection,post_process,
// brdf}.cuh, cutils_math.h) as IEEE binary32 operations in the reference's evaluation
// order; the translation unit is built with -ffp-contract=off so nothing is fused unless
// written as an explicit fma.  What differs from the reference is everything AROUND the
// arithmetic: geometry comes from LDS-staged 48-byte {e1,e2,v0} records, the nearest hit is
// found by an ordered stackless BVH walk, camera constants are hoisted to the host, the
// RNG/transcendentals are the build's defined functions (DESIGN.md "Defined arithmetic").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptamd {

struct f3 { float x, y, z; };

#define PT_DEV __device__ __forceinline__

PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 mk3(float s) { return mk3(s, s, s); }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator+(f3 a, float b) { return mk3(a.x + b, a.y + b, a.z + b); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator-(f3 a, float b) { return mk3(a.x - b, a.y - b, a.z - b); }
PT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
PT_DEV f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
PT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
PT_DEV f3 operator/(float b, f3 a) { return mk3(b / a.x, b / a.y, b / a.z); }
PT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// cutils_math.h:70-74,1557: v * (1.0f / sqrtf(dot))
PT_DEV f3 normalize(f3 v) { float inv_len = 1.0f / __builtin_sqrtf(dot(v, v)); return v * inv_len; }

#ifndef PT_SHORT_MATH
#define PT_SHORT_MATH 1   /* 0: the compiler's full sqrt / division everywhere (A/B: scripts/build_variants.sh x="-DPT_SHORT_MATH=0" + scripts/gpu_ab.sh) */
#endif
// ---- short forms of the two IEEE operations the shading code is full of.  Each is the compiler's own expansion minus
// the steps that are the identity inside a range; scripts/ubench/sqrt_exact.hip and rcp_exact.hip compare them with the
// full forms on ALL 2^32 binary32 patterns on the device (0 mismatches inside the ranges stated here).
// 1.0f / x, correctly rounded, for 2^-95 <= |x| <= 2^125: no v_div_scale / v_div_fixup (7 VALU instead of 11).
PT_DEV float rcp_exact_in_range(float x)
{
  const float y0 = __builtin_amdgcn_rcpf(x);
  const float e0 = __builtin_fmaf(-x, y0, 1.0f);
  const float y1 = __builtin_fmaf(e0, y0, y0);
  const float r1 = __builtin_fmaf(-x, y1, 1.0f);
  const float q1 = __builtin_fmaf(r1, y1, y1);
  const float r2 = __builtin_fmaf(-x, q1, 1.0f);
  return __builtin_fmaf(r2, y1, q1);
}
// sqrtf(x), correctly rounded, for x == 0, 2^-96 <= x <= +inf and NaN: v_sqrt_f32 and the "is a neighbour better" step,
// without the 2^32 pre-scaling of small inputs and the 0 / inf fix-up (9 VALU instead of 15).
PT_DEV float sqrt_exact_in_range(float x)
{
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
  s = rd <= 0.0f ? sd : s;
  s = ru > 0.0f ? su : s;
  return s;
}
// true when no active lane of the wave violates c: the short forms are taken by whole waves only
PT_DEV bool wave_all(bool c) { return __builtin_amdgcn_ballot_w64(!c) == 0ull; }
// normalize() for the hot spots of path_post: same value, 16 VALU instead of 26 for the reciprocal length whenever
// every lane's squared length is in [2^-96, inf)
PT_DEV f3 normalize_hot(f3 v)
{
  const float d = dot(v, v);
  float inv_len;
  if (PT_SHORT_MATH && wave_all(d >= 0x1p-96f && d < __builtin_inff())) inv_len = rcp_exact_in_range(sqrt_exact_in_range(d));
  else inv_len = 1.0f / __builtin_sqrtf(d);
  return v * inv_len;
}
// 1.0f / x where x is usually, not always, inside rcp_exact_in_range's domain
PT_DEV float rcp_hot(float x)
{
  if (PT_SHORT_MATH && wave_all(x >= 0x1p-95f && x <= 0x1p125f)) return rcp_exact_in_range(x);
  return 1.0f / x;
}
// sqrtf(x) for a non-negative (or NaN) x that is usually 0 or >= 2^-96
PT_DEV float sqrt_checked(float x)
{
  if (PT_SHORT_MATH && wave_all(!(x < 0x1p-96f) || x == 0.0f)) return sqrt_exact_in_range(x);
  return __builtin_sqrtf(x);
}
// sqrtf(x) for an x known to be 0 or >= 2^-96 (a uniform variate, 1 - a uniform variate)
PT_DEV float sqrt_hot(float x) { return PT_SHORT_MATH ? sqrt_exact_in_range(x) : __builtin_sqrtf(x); }
// cutils_math.h:1678
PT_DEV f3 reflect(f3 i, f3 n) { return i - (2.0f * n) * dot(n, i); }
// cutils_math.h:1722
PT_DEV f3 mix(f3 x, f3 y, float a) { return (x * (1.0f - a)) + y * a; }
// cutils_math.h:44-56,1357: NaN-propagating-to-1 clamp
PT_DEV float clamp01(float f) { float m = f < 1.0f ? f : 1.0f; return 0.0f > m ? 0.0f : m; }

PT_DEV uint32_t f_as_u(float f) { return __float_as_uint(f); }
PT_DEV float u_as_f(uint32_t u) { return __uint_as_float(u); }

// ---------------------------------------------------------------- defined math
// Same definitions as the test oracle states (DESIGN.md "Defined arithmetic"): the
// reference's cosf/sinf/__cosf/__sinf/powf are CUDA-toolkit code outside its tree.

PT_DEV void pt_sincosf(float x, float& s, float& c)
{
  const float two_over_pi = 0x1.45f306p-1f;
  const float pio2_hi = 0x1.921fb6p+0f;
  const float pio2_lo = -0x1.777a5cp-25f;
  float k = __builtin_rintf(x * two_over_pi);
  float r = __builtin_fmaf(-k, pio2_hi, x);
  r = __builtin_fmaf(-k, pio2_lo, r);
  float r2 = r * r;
  float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  float sn = __builtin_fmaf(r * r2, ps, r);
  float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  float cs = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(-0.5f, r2, 1.0f));
  int q = (int)k & 3;
  float so = (q & 1) ? cs : sn;
  float co = (q & 1) ? sn : cs;
  if (q == 1 || q == 2) co = -co;
  if (q >= 2) so = -so;
  s = so;
  c = co;
}

// Polynomial coefficients of pt_powf live in constant memory: they are fetched with scalar
// loads into SGPR pairs.  As inline 64-bit literals the compiler materialised each of them in a
// VGPR pair, hoisted all of them out of the bounce loop and spilled them (112 B of scratch per
// lane, 232 MB of extra HBM writes per 1080p launch in the rocprofv3 WRITE_SIZE counter).
__constant__ double kPowLog[9] = { 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 1.0 };
__constant__ double kPowExp[14] = { 1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0,
                                    1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0,
                                    1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0, 1.0 };

// powf evaluated in binary64: 2^(y*log2 x)
__device__ __noinline__ float pt_powf(float xf, float yf)
{
  if (yf == 0.0f || xf == 1.0f) return 1.0f;
  if (xf != xf || yf != yf) return __builtin_nanf("");
  double x = (double)xf, y = (double)yf;
  bool negate = false;
  if (xf < 0.0f) {
    if (__builtin_floorf(yf) != yf) return __builtin_nanf("");
    float h = __builtin_fabsf(yf) * 0.5f;
    negate = __builtin_floorf(h) != h;
    x = -x;
  }
  double res;
  const double inf = __builtin_inf();
  if (x == 0.0) {
    res = y > 0.0 ? 0.0 : inf;
  } else if (x == inf) {
    res = y > 0.0 ? inf : 0.0;
  } else if (y == inf || y == -inf) {
    res = ((x > 1.0) == (y > 0.0)) ? inf : 0.0;
  } else {
    uint64_t b = (uint64_t)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = kPowLog[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) p = __builtin_fma(p, s2, kPowLog[i]);
    double lnm = 2.0 * s * __builtin_fma(p, s2, kPowLog[8]);
    double log2x = (double)e + lnm * 0x1.71547652b82fep+0;
    double z = y * log2x;
    if (z > 1100.0) {
      res = inf;
    } else if (z < -1100.0) {
      res = 0.0;
    } else {
      double n = __builtin_rint(z);
      double t = (z - n) * 0x1.62e42fefa39efp-1;
      double q = kPowExp[0];
#pragma unroll
      for (int i = 1; i < 14; ++i) q = __builtin_fma(q, t, kPowExp[i]);
      int ni = (int)n;
      int n1 = ni / 2, n2 = ni - n1;
      double s1 = __longlong_as_double((long long)((uint64_t)(n1 + 1023) << 52));
      double s2b = __longlong_as_double((long long)((uint64_t)(n2 + 1023) << 52));
      res = q * s1 * s2b;
    }
  }
  return (float)(negate ? -res : res);
}

// cvt.rzi.u32.f32 semantics of the reference's bit-field stores (raytrace.cu:266-268)
PT_DEV uint32_t pt_f2u(float v)
{
  if (!(v > 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)v;
}

// ---------------------------------------------------------------- RNG (cuRAND XORWOW restated)

struct Xorwow { uint32_t v0, v1, v2, v3, v4, d; };

// raytrace.cu:275-285 (host function in the reference; batched launches evaluate it per sample)
PT_DEV uint32_t wang_hash(uint32_t a)
{
  a = (a ^ 61u) ^ (a >> 16);
  a = a + (a << 3);
  a = a ^ (a >> 4);
  a = a * 0x27d4eb2du;
  a = a ^ (a >> 15);
  return a;
}

PT_DEV void xorwow_init(Xorwow& st, uint32_t seed)
{
  uint32_t s0 = seed ^ 0xaad26b49u;
  uint32_t s1 = 0xf7dcefddu; // high word of the 64-bit seed is zero (raytrace.cu:235)
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  st.v0 = 123456789u + t0;
  st.v1 = 362436069u ^ t0;
  st.v2 = 521288629u + t1;
  st.v3 = 88675123u ^ t1;
  st.v4 = 5783321u + t0;
  st.d = 6615241u + t1 + t0;
}

PT_DEV float xorwow_uniform(Xorwow& st)
{
  uint32_t t = st.v0 ^ (st.v0 >> 2);
  st.v0 = st.v1; st.v1 = st.v2; st.v2 = st.v3; st.v3 = st.v4;
  st.v4 = (st.v4 ^ (st.v4 << 4)) ^ (t ^ (t << 1));
  st.d += 362437u;
  uint32_t x = st.v4 + st.d;
  return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// ---------------------------------------------------------------- kernel parameters

struct TexDesc { int32_t w, h, nb_chan; uint32_t pad; uint64_t offset; };

#define PT_HEAD_STRIDE 64u
// pixel tile of a wave in the persistent kernels: PT_TILE_W x (64 / PT_TILE_W), PT_TILE_W a power of two (8: square)
#ifndef PT_TILE_W_LOG2
#define PT_TILE_W_LOG2 3u
#endif
#define PT_TILE_W (1u << PT_TILE_W_LOG2)
#define PT_TILE_H (64u >> PT_TILE_W_LOG2)
struct KParams {
  // scene (device pointers)
  const float4* nodes;      // 4 float4 per BVH node
  const float4* tris_bvh;   // 3 float4 per triangle, leaf-major
  const float4* tris_brute; // 3 float4 per triangle, storage order
  const float4* shade;      // 7 float4 per face, storage order: n0 n1 n2 | uv0 uv1 uv2 | tangent | material id, ior | diffuse map desc | normal map desc
  const int4* materials;    // {diffuse_spec_map, normal_map, bits(ior), 0}
  const float4* lights;     // 2 float4 per light: {color.xyz, vec.x} {vec.y, vec.z, emission, radius^2}
  const TexDesc* textures;
  const float* texels;
  const float4* cubemap;    // 6 * size * size
  uint32_t cubemap_size;
  uint32_t n_faces, n_lights, n_nodes, n_bvh_tris;
  // camera, pixel-invariant part of generateRay hoisted to the host (same float ops)
  f3 cam_pos, cam_p0, cam_u, cam_v; // p0 = position + dir * screen_dist
  float focus_dist, aperture;
  // frame
  uint32_t width, height, row_begin, row_end;
  uint32_t hash_seed;
  float frame_nb_f;   // (float)frame_nb
  float frame_nb_inv; // 1 / frame_nb_f when frame_nb_f is a power of two (then x * frame_nb_inv is x / frame_nb_f for every x: the
                      // same real number rounded once), else 0: the epilogue's three divisions become multiplies (4, 8, 16 spp)
  int32_t is_static;
  int32_t bounces;
  uint32_t post_id;
  // outputs
  float* tfb;         // float3 per pixel
  uint32_t* surface;  // RGBA8 per pixel
  uint32_t tfb_row0;  // buffers hold frame rows starting here (0 for full-frame buffers)
  uint32_t tfb_reset; // != 0: treat the accumulator as zero on entry (ptamd_launch.reset_accumulation)
  uint32_t surf_row0;
  unsigned long long* stats; // 8 counters or nullptr
  unsigned long long* error_flag; // incremented when a bounded spin of the split kernel times out
  // persistent variant: 8x8 tiles over the row band, handed out by a ticket counter
  uint32_t* tile_counter;
  // persistent kernel: eight ticket heads, PT_HEAD_STRIDE dwords apart (one per XCD; a single head saturates at ~88
  // dequeues/us, MI355X_MICROARCH.md 'dequeue').  Head h hands out tickets n_static + 8 t + h; the first n_static
  // tickets are taken by the waves without an atomic.
  uint32_t* tile_heads;
  uint32_t n_static;
  uint32_t tiles_x, n_tiles, tiles_per_ticket;
  uint32_t refill_min; // idle lanes that trigger a refill (1..64)
  // batched frames (persistent variant): `sample_count` consecutive static frames in one launch.
  // Sample k of a pixel uses hash_seed = WangHash(frame_nb0 + k); its clamped radiance goes to
  // samples_out[k][band row][x] and pt_resolve_kernel applies them to the accumulator in frame order.
  uint32_t sample_count, frame_nb0;
  float* samples_out;
  // restart kernel: per-wave pools of fresh paths (192 float4 per wave) and the straggler threshold of a round
  float4* pool;
  uint32_t pool_lds_offset;   // != 0: the pools live in LDS this many bytes behind the start of the dynamic LDS (2 304 bytes per wave)
  uint32_t round_min, round_div;
  uint32_t round_div_m16;   // ceil(2^16 / round_div): x / round_div == (x * round_div_m16) >> 16 for x < 1024 (the waves divide lane counts: two scalar instructions instead of the fifteen of a 32-bit division)
  uint32_t walk_min;   // a box phase ends once fewer lanes than this are still walking (1: when none is)
  uint32_t small_det;  // != 0: the scene's coordinates are <= 1e8, so Moller-Trumbore determinants stay below 2^125
  // != 0: the environment is one colour (a 1x1 cubemap whose six texels are bit-identical — what the reference ends up
  // with when its cubemap image does not load): texCubemap returns env_color whatever the direction, no lookup
  uint32_t env_uniform;
  float env_r, env_g, env_b;
  // 258 floats built once per context (pt_build_gamma_table): T[j], j = 1..255, is the smallest x whose surface byte
  // pt_f2u(pt_powf(x, 1/2.2) * 255) reaches j, T[0] = 0, T[256] the smallest x whose value reaches 256 (from there on the
  // resolve pass evaluates pt_powf as before).  nullptr: no table, pt_powf everywhere.
  const float* gamma_table;
  // wide walk (scenes that do not fit in LDS): 8 float4 per node — Bvh::nodes4, or Bvh::nodes8 when `wide8` is set —, per-lane
  // stacks in LDS with a global continuation
  const float4* nodes4;
  uint32_t n_nodes4;
  uint32_t stack_lds_entries, stack_spill_entries;
  // interleaved bands of a multi-GPU split (restart kernel, band-local buffers): 0/1 = off
  uint32_t ilv_ranks, ilv_rank, ilv_rows;
  uint32_t y_limit;   // restart kernel: frame rows >= this are outside the launch (row_end; the frame height for interleaved bands)
  uint32_t treelet_nodes;   // the first nodes of nodes4 (top of the tree) are staged in LDS in front of the stacks
  uint2* stack_spill;
  uint32_t walk_min4;
  uint32_t wide8;                 // which form `nodes4` holds: 0 four-wide float nodes, 1 the eight-wide quantised form (PTAMD_WIDE8 knob), 2 four-wide 64-byte quantised nodes
  uint32_t far_table[16];         // eight-wide walk: for ray octant o, byte c of the pair [2 o], [2 o + 1] = the slots visited after slot c
  uint32_t xcd_regions;           // restart kernel: != 0: tickets map to tiles through XCD-local regions (pt_kernels.hip: region_tile); needs n_static % 8 == 0 and tiles_per_ticket == 1
  uint32_t brute_walk;            // restart kernel: the launch wants the instantiation that tests every triangle record instead of walking the tree (far origin)
  unsigned long long* timeline;   // restart kernel: != nullptr selects the instantiation that records 4 time stamps per wave (ptamd_set_timeline)
};
// LDS bytes of one wave's pool of fresh paths (restart kernel: 64 entries x 9 dwords)
#define PT_POOL_LDS_BYTES 2304u

// one intersect() result carried through radiance()
struct Hit {
  f3 normal;
  f3 diffuse_col;
  float dist;
  float specular_col; // carried over between iterations on light hits
  float ior;
  int light;          // -1 none
  float emission;     // light hits: the light's emission (its colour is diffuse_col)
};

} // namespace ptamd
