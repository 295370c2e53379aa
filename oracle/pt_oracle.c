/*
 * pt_oracle.c — CPU ORACLE (test infrastructure, see pt_oracle.h).  PARITY UNPINNED.
 *
 * Plain-C restatement of the reference megakernel, written to execute the SAME sequence
 * of IEEE-754 binary32 operations per pixel as the reference source text does
 * (evaluation order of every expression is kept; build with -ffp-contract=off so the
 * compiler fuses nothing).  Each function cites the reference lines it follows;
 * abbreviations: RT = cuda_opengl/src/shaders/raytrace.cu, IX = include/shaders/intersection.cuh,
 * PP = include/shaders/post_process.cuh, BR = include/shaders/brdf.cuh,
 * CM = include/shaders/cutils_math.h, SD = include/scene/scene_data.h.
 *
 * Arithmetic that lives OUTSIDE /root/reference and is restated from its published
 * definition (the choices are recorded in DESIGN.md "Defined arithmetic"):
 *   - cuRAND XORWOW (CUDA toolkit curand_kernel.h; the VS project pins CUDA 9.0):
 *     Marsaglia xorwow, Weyl increment 362437, seed scramble of curand_init, and
 *     curand_uniform = x * 2^-32 + 2^-33.
 *   - CUDA libdevice cosf/sinf/powf and the fast intrinsics __cosf/__sinf/__fdividef/
 *     __fsqrt_rn: restated as or_sincosf (<= 2 ulp on [0, 2pi]), or_powf (computed in
 *     binary64, < 0.51 ulp), IEEE division and IEEE sqrt.
 *   - the texture unit's cubemap fetch (texCubemap, linear filter): face selection and
 *     (s,t) mapping from the CUDA Programming Guide cubemap table, bilinear weights
 *     quantised to 8 fractional bits as the guide's "linear filtering" section states.
 *   - float->unsigned conversion of cvt.rzi.u32.f32 (NaN and negatives give 0).
 * Double-precision sub-expressions of the reference (literals without the f suffix,
 * M_PI from <math.h> on Linux) are kept in double.
 *
 * Reference behaviour that is undefined in its source and is DEFINED here:
 * `IntersectionData inter` (RT:52) is never initialised and several fields are not written
 * on light-sphere hits (IX:199-212); the oracle value-initialises it to zero and carries
 * fields over between iterations.
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

_Static_assert(sizeof(or_face) == 112, "Face is 112 B (SD:46-53)");
_Static_assert(sizeof(or_material) == 16, "Material is 16 B (SD:95-100)");
_Static_assert(sizeof(or_light) == 32, "LightProp is 32 B (SD:109-115)");
_Static_assert(sizeof(or_camera) == 64, "Camera is 64 B (SD:123-133)");

#define OR_PI_D 3.14159265358979323846 /* M_PI of <math.h> (double), RT:106, PP:55 */

/* ------------------------------------------------------------------ study instantiations (tests/test_tolerance_study.py)
 * The DEFAULT build (no macro, -ffp-contract=off) is the oracle.  The switches below build stand-ins for what the reference's
 * own binary may compute differently — it is built with `nvcc -O2` and nothing else (CMakeLists.txt:20-22), i.e. --fmad=true,
 * with libdevice's cosf/sinf/powf/tanf and the fast intrinsics __cosf/__sinf/__fdividef — so that the size of that gap can be
 * stated (scripts/tolerance_study.py).  None of them is "the reference": they bound how far two faithful
 * builds of this integrator drift apart.
 *   -ffp-contract=fast     (compiler flag) a*b+c contracted to fma wherever the compiler likes, as nvcc's default permits
 *   -DOR_STUDY_LIBM        glibc's sinf/cosf/powf in place of or_sincosf/or_powf (another <= 2 ulp implementation, as libdevice is)
 *   -DOR_STUDY_FASTINTR    RT:111-112,122's __cosf/__sinf as values with an absolute error of up to 2^-22 (documented bound of
 *                          sin.approx.f32 on [-pi, pi]: 2^-21.41), and __fdividef(a, b) (IX:113, RT:206) as a * (1 / b)
 *                          (two roundings; documented bound 2 ulp) */
#ifdef OR_STUDY_FASTINTR
static inline float or_fast_trig(double v) { return (float)(rint(v * 2097152.0) / 2097152.0); }   /* multiples of 2^-21 */
#define OR_FDIVIDEF(a, b) ((a) * (1.0f / (b)))
#else
#define OR_FDIVIDEF(a, b) ((a) / (b))
#endif

/* ------------------------------------------------------------------ CM.h float3 ops */
static inline or_f3 f3(float x, float y, float z) { or_f3 r = { x, y, z }; return r; }
static inline or_f3 f3s(float s) { return f3(s, s, s); }                                   /* CM:139 */
static inline or_f3 add(or_f3 a, or_f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); } /* CM:432 */
static inline or_f3 adds(or_f3 a, float b) { return f3(a.x + b, a.y + b, a.z + b); }      /* CM:444 */
static inline or_f3 sub(or_f3 a, or_f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); } /* CM:707 */
static inline or_f3 subs(or_f3 a, float b) { return f3(a.x - b, a.y - b, a.z - b); }      /* CM:719 */
static inline or_f3 mul(or_f3 a, or_f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); } /* CM:966 */
static inline or_f3 muls(or_f3 a, float b) { return f3(a.x * b, a.y * b, a.z * b); }      /* CM:977 */
static inline or_f3 smul(float b, or_f3 a) { return f3(b * a.x, b * a.y, b * a.z); }      /* CM:981 */
static inline or_f3 divv(or_f3 a, or_f3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); } /* CM:1174 */
static inline or_f3 divs(or_f3 a, float b) { return f3(a.x / b, a.y / b, a.z / b); }      /* CM:1186 */
static inline or_f3 sdiv(float b, or_f3 a) { return f3(b / a.x, b / a.y, b / a.z); }      /* CM:1198 */
static inline or_f3 neg(or_f3 a) { return f3(-a.x, -a.y, -a.z); }                         /* CM:323 */
static inline float dot(or_f3 a, or_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* CM:1484 */
static inline or_f3 cross(or_f3 a, or_f3 b)                                               /* CM:1688 */
{
  return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline or_f3 normalize(or_f3 v)                                                    /* CM:70-74,1557 */
{
  float inv_len = 1.0f / sqrtf(dot(v, v));
  return muls(v, inv_len);
}
static inline or_f3 reflect(or_f3 i, or_f3 n)                                             /* CM:1678 */
{
  return sub(i, muls(smul(2.0f, n), dot(n, i)));
}
static inline or_f3 mix(or_f3 x, or_f3 y, float a)                                        /* CM:1722 */
{
  return add(muls(x, 1.0f - a), muls(y, a));
}
static inline float fminf_d(float a, float b) { return a < b ? a : b; }                   /* CM:44 */
static inline float fmaxf_d(float a, float b) { return a > b ? a : b; }                   /* CM:50 */
static inline float clampf(float f, float a, float b) { return fmaxf_d(a, fminf_d(f, b)); } /* CM:1357 */

/* ------------------------------------------------------------------ defined math */

/* sin and cos of x for |x| <~ 100 (the path only passes [0, 2*pi]).  Cody-Waite
 * reduction by pi/2 with fma, then the Cephes single-precision minimax polynomials on
 * [-pi/4, pi/4].  Stands for CUDA's cosf/sinf (PP:60) and __cosf/__sinf (RT:122). */
void or_sincosf(float x, float* s, float* c)
{
#ifdef OR_STUDY_LIBM
  *s = sinf(x); *c = cosf(x);
  return;
#endif
  const float two_over_pi = 0x1.45f306p-1f;
  const float pio2_hi = 0x1.921fb6p+0f;
  const float pio2_lo = -0x1.777a5cp-25f;
  float k = rintf(x * two_over_pi);
  float r = fmaf(-k, pio2_hi, x);
  r = fmaf(-k, pio2_lo, r);
  float r2 = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  float sn = fmaf(r * r2, ps, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  float cs = fmaf(r2 * r2, pc, fmaf(-0.5f, r2, 1.0f));
  int q = (int)k & 3;
  float so = (q & 1) ? cs : sn;
  float co = (q & 1) ? sn : cs;
  if (q == 1 || q == 2) co = -co;
  if (q >= 2) so = -so;
  *s = so;
  *c = co;
}

static inline double or_bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t or_double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* powf computed in binary64: 2^(y*log2(x)).  Stands for CUDA's powf (RT:163, CM:1750). */
float or_powf(float xf, float yf)
{
#ifdef OR_STUDY_LIBM
  return powf(xf, yf);
#endif
  if (yf == 0.0f || xf == 1.0f) return 1.0f;
  if (xf != xf || yf != yf) return NAN;
  double x = (double)xf, y = (double)yf;
  int negate = 0;
  if (xf < 0.0f) {
    if (floorf(yf) != yf) return NAN;
    /* integer exponent: odd -> negative result */
    negate = fmodf(fabsf(yf), 2.0f) == 1.0f;
    x = -x;
  }
  double res;
  if (x == 0.0) {
    res = y > 0.0 ? 0.0 : INFINITY;
  } else if (isinf(x)) {
    res = y > 0.0 ? INFINITY : 0.0;
  } else if (isinf(y)) {
    res = ((x > 1.0) == (y > 0.0)) ? INFINITY : 0.0;
  } else {
    uint64_t b = or_double_to_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023; /* a float is never a binary64 subnormal */
    double m = or_bits_to_double((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 0x1.6a09e667f3bcdp+0) { m *= 0.5; e += 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 17.0;
    p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0);
    p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0);
    p = fma(p, s2, 1.0 / 3.0);
    double lnm = 2.0 * s * fma(p, s2, 1.0);
    double log2x = (double)e + lnm * 0x1.71547652b82fep+0;
    double z = y * log2x;
    if (z > 1100.0) {
      res = INFINITY;
    } else if (z < -1100.0) {
      res = 0.0;
    } else {
      double n = rint(z);
      double t = (z - n) * 0x1.62e42fefa39efp-1;
      double q = 1.0 / 6227020800.0;           /* 1/13! */
      q = fma(q, t, 1.0 / 479001600.0);
      q = fma(q, t, 1.0 / 39916800.0);
      q = fma(q, t, 1.0 / 3628800.0);
      q = fma(q, t, 1.0 / 362880.0);
      q = fma(q, t, 1.0 / 40320.0);
      q = fma(q, t, 1.0 / 5040.0);
      q = fma(q, t, 1.0 / 720.0);
      q = fma(q, t, 1.0 / 120.0);
      q = fma(q, t, 1.0 / 24.0);
      q = fma(q, t, 1.0 / 6.0);
      q = fma(q, t, 0.5);
      q = fma(q, t, 1.0);
      q = fma(q, t, 1.0);
      /* scale by 2^n in two exact steps so the exponent field never leaves [1,2046] */
      int ni = (int)n;
      int n1 = ni / 2, n2 = ni - n1;
      double s1 = or_bits_to_double((uint64_t)(n1 + 1023) << 52);
      double s2b = or_bits_to_double((uint64_t)(n2 + 1023) << 52);
      res = q * s1 * s2b;
    }
  }
  return (float)(negate ? -res : res);
}

/* cvt.rzi.u32.f32: NaN -> 0, negative -> 0, >= 2^32 -> 0xffffffff, else truncate. */
static inline uint32_t or_f2u(float v)
{
  if (!(v > 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)v;
}

/* ------------------------------------------------------------------ RNG */

/* RT:275-285 */
uint32_t or_wang_hash(uint32_t a)
{
  a = (a ^ 61u) ^ (a >> 16);
  a = a + (a << 3);
  a = a ^ (a >> 4);
  a = a * 0x27d4eb2du;
  a = a ^ (a >> 15);
  return a;
}

/* cuRAND XORWOW, curand_init(seed, subsequence=0, offset=0) (call site RT:235).  The seed
 * argument there is a 32-bit unsigned widened to 64 bits, so its high word is zero.
 * state = { v[0..4], d }. */
void or_xorwow_init(uint32_t seed, uint32_t st[6])
{
  uint32_t s0 = seed ^ 0xaad26b49u;
  uint32_t s1 = 0u ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  st[0] = 123456789u + t0;
  st[1] = 362436069u ^ t0;
  st[2] = 521288629u + t1;
  st[3] = 88675123u ^ t1;
  st[4] = 5783321u + t0;
  st[5] = 6615241u + t1 + t0;
}

uint32_t or_xorwow_next(uint32_t st[6])
{
  uint32_t t = st[0] ^ (st[0] >> 2);
  st[0] = st[1];
  st[1] = st[2];
  st[2] = st[3];
  st[3] = st[4];
  st[4] = (st[4] ^ (st[4] << 4)) ^ (t ^ (t << 1));
  st[5] += 362437u;
  return st[4] + st[5];
}

/* curand_uniform: (0, 1] (call sites RT:70,107,169; PP:55,58) */
float or_xorwow_uniform(uint32_t st[6])
{
  uint32_t x = or_xorwow_next(st);
  return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* ------------------------------------------------------------------ textures */

/* IX:20-65 nearest texel, no filtering */
static inline int texture_idx(const or_texture* tex, or_f2 uv)
{
  int x = (int)(uv.x * (float)(tex->w - 1));
  int y = (int)(uv.y * (float)(tex->h - 1));
  return (y * tex->w + x) * tex->nb_chan;
}

/* texCubemap(cubemap_ref, x, y, z) with cudaFilterModeLinear, normalised coordinates
 * (RT:22,60,197,305-309).  Face order +x,-x,+y,-y,+z,-z (GP.cpp:119-127).  A 1x1 cubemap
 * returns its texel exactly (every tap is the same texel). */
void or_tex_cubemap(const or_scene* sc, float x, float y, float z, float out[4])
{
  const uint32_t n = sc->cubemap_size;
  float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
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
  const float* base = sc->cubemap + (size_t)face * n * n * 4;
  if (n == 1) {
    out[0] = base[0]; out[1] = base[1]; out[2] = base[2]; out[3] = base[3];
    return;
  }
  float u = (s / m + 1.0f) * 0.5f;
  float v = (t / m + 1.0f) * 0.5f;
  float xb = u * (float)n - 0.5f;
  float yb = v * (float)n - 0.5f;
  float fx = floorf(xb), fy = floorf(yb);
  /* 8 fractional bits (CUDA Programming Guide, "Linear Filtering") */
  float a = floorf((xb - fx) * 256.0f) * (1.0f / 256.0f);
  float b = floorf((yb - fy) * 256.0f) * (1.0f / 256.0f);
  int i0 = (int)fx, j0 = (int)fy, i1 = i0 + 1, j1 = j0 + 1;
  const int hi = (int)n - 1;
  if (!(xb == xb)) { i0 = i1 = 0; a = 0.0f; }   /* NaN direction: defined as texel 0 */
  if (!(yb == yb)) { j0 = j1 = 0; b = 0.0f; }
  i0 = i0 < 0 ? 0 : (i0 > hi ? hi : i0);
  i1 = i1 < 0 ? 0 : (i1 > hi ? hi : i1);
  j0 = j0 < 0 ? 0 : (j0 > hi ? hi : j0);
  j1 = j1 < 0 ? 0 : (j1 > hi ? hi : j1);
  const float* t00 = base + ((size_t)j0 * n + i0) * 4;
  const float* t10 = base + ((size_t)j0 * n + i1) * 4;
  const float* t01 = base + ((size_t)j1 * n + i0) * 4;
  const float* t11 = base + ((size_t)j1 * n + i1) * 4;
  for (int c = 0; c < 4; ++c) {
    float top = t00[c] * (1.0f - a) + t10[c] * a;
    float bot = t01[c] * (1.0f - a) + t11[c] * a;
    out[c] = top * (1.0f - b) + bot * b;
  }
}

/* ------------------------------------------------------------------ ray generation */

typedef struct { or_f3 dir, origin; } or_ray; /* SD:140-144 */

/* IX:75-97.  Mutates cam->u / cam->v exactly like the reference (the host-supplied
 * u/v are ignored and recomputed; u is negated after v was derived from it). */
static or_ray generate_ray(int x, int y, int half_w, int half_h, or_camera* cam)
{
  float screen_dist = (float)half_w / tanf(cam->fov_x * 0.5f);
  or_ray ray;
  ray.origin = cam->position;
  cam->u = normalize(cross(cam->dir, f3(0.0f, -1.0f, 0.0f)));
  cam->v = normalize(cross(cam->u, cam->dir));
  cam->u = muls(cam->u, -1.0f);
  or_f3 screen_pos = add(add(add(cam->position, muls(cam->dir, screen_dist)),
                             muls(cam->u, (float)(x - half_w))),
                         muls(cam->v, (float)(y - half_h)));
  ray.dir = sub(screen_pos, cam->position);
  ray.dir = normalize(ray.dir);
  return ray;
}

void or_generate_ray(int x, int y, int half_w, int half_h, or_camera* cam, or_f3* dir, or_f3* origin)
{
  or_ray r = generate_ray(x, y, half_w, half_h, cam);
  *dir = r.dir;
  *origin = r.origin;
}

/* PP:49-67 */
static void camera_dof(or_ray* r, const or_camera* cam, uint32_t* rng)
{
  or_f3 focal_point = smul(cam->focus_dist, r->dir);
  float random_angle = (float)((double)(or_xorwow_uniform(rng) * 2.0f) * OR_PI_D);
  float random_radius = or_xorwow_uniform(rng) * cam->aperture;
  float sn, cs;
  or_sincosf(random_angle, &sn, &cs);
  or_f3 random_aperture_pos = muls(add(smul(cs, cam->u), smul(sn, cam->v)), random_radius);
  or_f3 final_ray_dir = normalize(sub(focal_point, random_aperture_pos));
  r->origin = add(r->origin, random_aperture_pos);
  r->dir = final_ray_dir;
}

/* ------------------------------------------------------------------ intersection */

/* IX:102-135.  bu/bv (barycentrics) are extra outputs for the equivalence tests. */
static int intersect_triangle(const or_face* face, or_f3* out_normal, or_f2* out_uv,
                              const or_ray* ray, float* t, float* bu, float* bv)
{
  or_f3 v0v1 = sub(face->vertices[1], face->vertices[0]);
  or_f3 v0v2 = sub(face->vertices[2], face->vertices[0]);
  or_f3 p_vec = cross(ray->dir, v0v2);
  float det = dot(v0v1, p_vec);
  if ((double)det < 0.0000001) /* IX:110: double literal */
    return 0;
  float inv_det = OR_FDIVIDEF(1.0f, det); /* __fdividef(1.f, det) */
  or_f3 t_vec = sub(ray->origin, face->vertices[0]);
  float u = dot(t_vec, p_vec) * inv_det;
  if (u < 0 || u > 1)
    return 0;
  or_f3 qvec = cross(t_vec, v0v1);
  float v = dot(ray->dir, qvec) * inv_det;
  if (v < 0 || u + v > 1)
    return 0;
  float w = 1.0f - u - v;
  *out_normal = add(add(smul(w, face->normals[0]), smul(u, face->normals[1])), smul(v, face->normals[2]));
  or_f2 uv;
  uv.x = w * face->texcoords[0].x + u * face->texcoords[1].x + v * face->texcoords[2].x;
  uv.y = w * face->texcoords[0].y + u * face->texcoords[1].y + v * face->texcoords[2].y;
  /* mod(out_uv, 1.0): CM:1728-1737 */
  out_uv->x = uv.x - floorf(uv.x / 1.0f);
  out_uv->y = uv.y - floorf(uv.y / 1.0f);
  *t = dot(v0v2, qvec) * inv_det;
  if (bu) *bu = u;
  if (bv) *bv = v;
  return 1;
}

int or_intersect_triangle(const or_face* f, const or_f3* dir, const or_f3* origin,
                          or_f3* n, or_f2* uv, float* t, float* bu, float* bv)
{
  or_ray r = { *dir, *origin };
  return intersect_triangle(f, n, uv, &r, t, bu, bv);
}

/* IX:140-155.  NOTE the statement at IX:152 discards the value of its conditional
 * expression: t ends up as b - disc when that exceeds epsilon, otherwise as b + disc
 * WHATEVER its sign (it is never set to 0). */
static int intersect_sphere(const or_ray* r, const or_light* light, float* t)
{
  const float epsilon = 0.01f;
  or_f3 op = sub(light->vec, r->origin);
  float b = dot(op, r->dir);
  float disc = b * b - dot(op, op) + light->radius * light->radius;
  if (disc < 0.0f)
    return 0;
  disc = sqrtf(disc); /* __fsqrt_rn */
  *t = b - disc;
  if (!(*t > epsilon))
    *t = b + disc;
  return *t != 0.0f;
}

int or_intersect_sphere(const or_f3* dir, const or_f3* origin, const or_light* l, float* t)
{
  or_ray r = { *dir, *origin };
  return intersect_sphere(&r, l, t);
}

/* IX:7-18 */
typedef struct {
  or_f3 normal, surface_normal, tangent, diffuse_col;
  or_f2 uv;
  const or_light* light;
  float dist, specular_col, ior;
} or_inter;

typedef struct { uint64_t calls, mesh_hits, nmap_hits; } or_stats;

/* IX:161-246 brute-force nearest hit.  face_index/bary are extra outputs. */
static int intersect(const or_ray* r, const or_scene* scene, or_inter* in, or_stats* st,
                     int* out_face, float* out_bu, float* out_bv)
{
  const float MAX_DIST = 100000.0f;
  float inter_dist = MAX_DIST;
  in->dist = MAX_DIST;
  or_f3 normal = f3s(0.0f);
  or_f2 uv = { 0.0f, 0.0f };
  const or_material* inter_mat = NULL;
  int face_index = -1, g = 0;
  float bu = 0.0f, bv = 0.0f, cu = 0.0f, cv = 0.0f;
  if (st) st->calls++;

  for (uint32_t m = 0; m < scene->n_meshes; ++m) {
    const or_mesh* mesh = &scene->meshes[m];
    for (uint32_t i = 0; i < mesh->size; ++i, ++g) {
      const or_face* face = &mesh->data[i];
      if (intersect_triangle(face, &normal, &uv, r, &inter_dist, &cu, &cv) &&
          inter_dist < in->dist && inter_dist > 0.0f) {
        inter_mat = &scene->materials[face->material_id];
        in->ior = inter_mat->ior;
        in->normal = normal;
        in->surface_normal = normal;
        in->tangent = face->tangent;
        in->uv = uv;
        in->dist = inter_dist;
        in->light = NULL;
        face_index = g; bu = cu; bv = cv;
      }
    }
  }

  for (uint32_t l = 0; l < scene->n_lights; ++l) {
    const or_light* light = &scene->lights[l];
    if (intersect_sphere(r, light, &inter_dist) && inter_dist < in->dist && inter_dist >= 0.0f) {
      in->light = light;
      in->dist = inter_dist;
      in->diffuse_col = f3(light->color.x, light->color.y, light->color.z);
      in->normal = normalize(sub(light->vec, smul(inter_dist, r->dir))); /* IX:208: origin ignored */
      inter_mat = NULL;
      face_index = -1;
    }
  }

  if (inter_mat) {
    const or_texture* tex = &scene->textures[inter_mat->diffuse_spec_map];
    int idx = texture_idx(tex, in->uv);
    in->diffuse_col.x = tex->data[idx];
    in->diffuse_col.y = tex->data[idx + 1];
    in->diffuse_col.z = tex->data[idx + 2];
    in->specular_col = tex->data[idx + 3];
    if (st) st->mesh_hits++;
    if (inter_mat->normal_map >= 0) {
      const or_texture* nt = &scene->textures[inter_mat->normal_map];
      int nidx = texture_idx(nt, in->uv);
      or_f3 nrm = f3(nt->data[nidx], nt->data[nidx + 1], nt->data[nidx + 2]);
      in->normal = normalize(subs(muls(nrm, 2.0f), 1.0f));
      or_f3 binormal = normalize(cross(in->tangent, in->surface_normal));
      or_f3 tx = in->tangent, ty = neg(binormal), tz = in->surface_normal;
      or_f3 a = in->normal;
      /* CM:1134-1139 mat3 * float3 */
      in->normal = f3(tx.x * a.x + ty.x * a.y + tz.x * a.z,
                      tx.y * a.x + ty.y * a.y + tz.y * a.z,
                      tx.z * a.x + ty.z * a.y + tz.z * a.z);
      if (st) st->nmap_hits++;
    }
  }
  if (out_face) *out_face = face_index;
  if (out_bu) *out_bu = bu;
  if (out_bv) *out_bv = bv;
  return in->dist < MAX_DIST;
}

void or_intersect(const or_scene* sc, const or_f3* dir, const or_f3* origin, or_hit* out)
{
  or_ray r = { *dir, *origin };
  or_inter in;
  memset(&in, 0, sizeof in);
  int face = -1;
  float bu = 0, bv = 0;
  int hit = intersect(&r, sc, &in, NULL, &face, &bu, &bv);
  out->t = in.dist;
  out->u = out->v = 0.0f;
  if (!hit) { out->kind = 0; out->index = -1; }
  else if (in.light) { out->kind = 2; out->index = (int32_t)(in.light - sc->lights); }
  else { out->kind = 1; out->index = face; out->u = bu; out->v = bv; }
}

/* rays: n * {dir.xyz, origin.xyz}; out: n * {kind, index, t bits, 0} */
void or_intersect_batch(const or_scene* sc, const float* rays, uint32_t n, int32_t* out)
{
  for (uint32_t i = 0; i < n; ++i) {
    or_f3 d = f3(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
    or_f3 o = f3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    or_hit h;
    or_intersect(sc, &d, &o, &h);
    out[i * 4 + 0] = h.kind;
    out[i * 4 + 1] = h.index;
    memcpy(&out[i * 4 + 2], &h.t, 4);
    out[i * 4 + 3] = 0;
  }
}

/* ------------------------------------------------------------------ radiance (RT:41-210) */

static or_f3 env_lookup(const or_scene* sc, or_f3 d)
{
  float v[4];
  or_tex_cubemap(sc, d.x, d.y, -d.z, v); /* RT:60,197 */
  return f3(v[0], v[1], v[2]);
}

static or_f3 radiance(or_ray* r, const or_scene* sc, uint32_t* rng, int is_static, int bounces, or_stats* st)
{
  or_f3 acc = f3s(0.0f);
  or_f3 throughput = f3s(1.0f);
  or_inter inter;
  memset(&inter, 0, sizeof inter); /* DEFINED: value-initialised (reference leaves it indeterminate) */

  if (!is_static) { /* RT:54-62: the dangling else makes this an unconditional return */
    if (intersect(r, sc, &inter, st, NULL, NULL, NULL))
      return inter.diffuse_col;
    return env_lookup(sc, r->dir);
  }

  const int max_bounces = bounces; /* RT:66 with static_samples = bounces - 2 */
  for (int b = 0; b < max_bounces; b++) {
    or_f3 oriented_normal;
    float r1 = or_xorwow_uniform(rng); /* drawn before tracing, also on misses */
    if (intersect(r, sc, &inter, st, NULL, NULL, NULL)) {
      float cos_theta = dot(inter.normal, r->dir);
      oriented_normal = inter.normal;
      or_f3 spec = normalize(reflect(r->dir, inter.normal));
      float PDF = 0.5f;                       /* BR:27-31 */
      or_f3 BRDF = inter.diffuse_col;         /* BR:14-18 */
      or_f3 direct_light = divs(BRDF, PDF);
      if (inter.ior == 1.0f || inter.light != NULL) {
        if (inter.light != NULL) {
          BRDF = f3(inter.light->color.x, inter.light->color.y, inter.light->color.z);
          acc = add(acc, mul(muls(BRDF, inter.light->emission), throughput));
        }
        float phi = (float)((double)2.0f * OR_PI_D * (double)or_xorwow_uniform(rng));
        float sin_t = sqrtf(r1);
        float cos_t = sqrtf(1.f - r1);
        or_f3 axis = ((double)fabsf(oriented_normal.x) > .1) ? f3(0.0f, 1.0f, 0.0f) : f3(1.0f, 0.0f, 0.0f);
        or_f3 u = normalize(cross(axis, oriented_normal));
        or_f3 v = cross(oriented_normal, u);
        float sphi, cphi;
#ifdef OR_STUDY_FASTINTR
        sphi = or_fast_trig(sin((double)phi)); cphi = or_fast_trig(cos((double)phi)); /* __sinf / __cosf (RT:111-112,122) */
#else
        or_sincosf(phi, &sphi, &cphi);
#endif
        or_f3 d = normalize(add(add(muls(muls(v, sin_t), cphi), muls(muls(u, sphi), sin_t)),
                                muls(oriented_normal, cos_t)));
        r->origin = add(r->origin, muls(r->dir, inter.dist));
        r->dir = mix(d, spec, inter.specular_col); /* not renormalised */
        r->origin = add(r->origin, muls(r->dir, 0.03f));
        throughput = mul(throughput, direct_light);
      } else {
        float n1 = 1.0f;
        float n2 = inter.ior;
        oriented_normal = cos_theta < 0 ? inter.normal : muls(inter.normal, -1.0f);
        float c1 = dot(oriented_normal, r->dir);
        int entering = dot(inter.normal, oriented_normal) > 0;
        float eta = entering ? n1 / n2 : n2 / n1;
        float eta_2 = eta * eta;
        float c2_term = 1.0f - eta_2 * (1.0f - c1 * c1);
        if (c2_term < 0.0f) {
          r->origin = add(r->origin, divs(muls(oriented_normal, inter.dist), 100.f));
          r->dir = spec;
        } else {
          float R0 = (n2 - n1) / (n1 + n2);
          R0 *= R0;
          float c2 = sqrtf(c2_term);
          or_f3 T = normalize(add(smul(eta, r->dir), smul(eta * c1 - c2, oriented_normal)));
          float f_cos_theta = or_powf(cos_theta, 5.0f); /* RT:162 is dead, RT:163 wins */
          float f_r = R0 + (1.0f - R0) * f_cos_theta;
          if (or_xorwow_uniform(rng) < 0.25f) {
            throughput = mul(throughput, smul(f_r, direct_light));
            r->origin = add(r->origin, divs(muls(oriented_normal, inter.dist), 100.f));
            r->dir = spec;
          } else {
            float f_t = 1.0f - f_r;
            throughput = mul(throughput, smul(f_t, direct_light));
            r->origin = add(r->origin, divs(muls(oriented_normal, inter.dist), 10000.f));
            r->dir = T;
          }
        }
      }
    } else {
      acc = add(acc, mul(env_lookup(sc, r->dir), throughput)); /* the loop continues */
    }
    float p = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)); /* CUDA fmaxf: NaN-ignoring */
    if (r1 > p && b > 1)
      return acc;
    throughput = muls(throughput, OR_FDIVIDEF(1.0f, p)); /* __fdividef(1.0f, p) */
  }
  return acc;
}

/* ------------------------------------------------------------------ post process */

/* PP:14-25 */
static or_f3 uncharted_tonemap(or_f3 x)
{
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  or_f3 num = adds(mul(x, adds(smul(A, x), C * B)), D * E);
  or_f3 den = adds(mul(x, adds(smul(A, x), B)), D * F);
  return subs(divv(num, den), E / F);
}

/* PP:31-41 */
static or_f3 exposure(or_f3 color)
{
  const float exposure_bias = 2.0f;
  or_f3 curr = uncharted_tonemap(smul(exposure_bias, color));
  or_f3 W = f3s(11.2f);
  or_f3 white_scale = sdiv(1.0f, uncharted_tonemap(W));
  return mul(curr, white_scale);
}

void or_exposure(const float in[3], float out[3])
{
  or_f3 r = exposure(f3(in[0], in[1], in[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* RT:327-352 (double literals) */
static or_f3 post_process(uint32_t id, or_f3 c)
{
  switch (id) {
  case 1: {
    float gray = (float)((double)c.x * 0.3 + (double)c.y * 0.59 + (double)c.z * 0.11);
    return f3(gray, gray, gray);
  }
  case 2:
    return f3((float)((double)c.x * 0.393 + (double)c.y * 0.769 + (double)c.z * 0.189),
              (float)((double)c.x * 0.349 + (double)c.y * 0.686 + (double)c.z * 0.168),
              (float)((double)c.x * 0.272 + (double)c.y * 0.534 + (double)c.z * 0.131));
  case 3:
    return f3((float)(1.0 - (double)c.x), (float)(1.0 - (double)c.y), (float)(1.0 - (double)c.z));
  default:
    return c;
  }
}

void or_post_process(uint32_t post_id, const float in[3], float out[3])
{
  or_f3 r = post_process(post_id, f3(in[0], in[1], in[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* RT:24-35,231-232,266-268: r is the low byte, alpha = 0 */
uint32_t or_pack_rgba(const float rad[3])
{
  uint32_t r = or_f2u(rad[0] * 255.0f) & 0xffu;
  uint32_t g = or_f2u(rad[1] * 255.0f) & 0xffu;
  uint32_t b = or_f2u(rad[2] * 255.0f) & 0xffu;
  return r | (g << 8) | (b << 16);
}

/* ------------------------------------------------------------------ kernel (RT:212-271) */

typedef struct {
  const or_scene* sc;
  or_camera cam;
  uint32_t width, height, y0, y1, hash_seed, post_id;
  int32_t frame_nb, moved, bounces;
  float* tfb;
  uint8_t* rgba;
  atomic_uint* next_row; /* shared row counter: rows differ a lot in cost, so threads pull them */
  or_stats stats;
} or_job;

static void render_pixel(or_job* j, int x, int y)
{
  const uint32_t width = j->width, height = j->height;
  const unsigned half_w = width / 2, half_h = height / 2;
  const unsigned grid_x = width / 16 + 1; /* RT:316: padded grid also when width % 16 == 0 */
  const unsigned tid = ((unsigned)(x >> 4) + (unsigned)(y >> 4) * grid_x) * 256u +
                       (unsigned)(y & 15) * 16u + (unsigned)(x & 15);
  uint32_t rng[6];
  or_xorwow_init(j->hash_seed + tid, rng);

  or_camera cam = j->cam; /* by-value kernel argument, mutated by generateRay */
  or_ray r = generate_ray(x, y, (int)half_w, (int)half_h, &cam);
  camera_dof(&r, &cam, rng);

  int is_static = !j->moved;
  or_f3 rad = radiance(&r, j->sc, rng, is_static, j->bounces, &j->stats);
  rad = f3(clampf(rad.x, 0.0f, 1.0f), clampf(rad.y, 0.0f, 1.0f), clampf(rad.z, 0.0f, 1.0f));

  size_t i = (size_t)(height - (unsigned)y - 1) * width + (unsigned)x;
  or_f3* tfb = (or_f3*)j->tfb;
  tfb[i] = muls(tfb[i], (float)is_static);
  tfb[i] = add(tfb[i], rad);
  rad = divs(tfb[i], (float)j->frame_nb);
  rad = exposure(rad);
  const float g = 1.0f / 2.2f;
  rad = f3(or_powf(rad.x, g), or_powf(rad.y, g), or_powf(rad.z, g));
  rad = post_process(j->post_id, rad);
  float rv[3] = { rad.x, rad.y, rad.z };
  uint32_t px = or_pack_rgba(rv);
  memcpy(j->rgba + ((size_t)y * width + (unsigned)x) * 4, &px, 4);
}

static void* render_band(void* arg)
{
  or_job* j = (or_job*)arg;
  for (;;) {
    uint32_t y = j->y0 + atomic_fetch_add(j->next_row, 1u);
    if (y >= j->y1) break;
    for (uint32_t x = 0; x < j->width; ++x)
      render_pixel(j, (int)x, (int)y);
  }
  return NULL;
}

static or_stats g_last_stats;

void or_last_stats(uint64_t out[3])
{
  out[0] = g_last_stats.calls;
  out[1] = g_last_stats.mesh_hits;
  out[2] = g_last_stats.nmap_hits;
}

int or_render(const or_scene* sc, const or_camera* cam, uint32_t width, uint32_t height,
              uint32_t y0, uint32_t y1, uint32_t hash_seed, int32_t frame_nb, int32_t moved,
              uint32_t post_id, int32_t bounces, float* tfb, uint8_t* rgba, int32_t nthreads)
{
  if (!sc || !cam || !tfb || !rgba || y1 > height || y0 > y1 || bounces < 1 || frame_nb < 1 || post_id > 3)
    return 1;
  if (nthreads < 1) nthreads = 1;
  uint32_t rows = y1 - y0;
  if ((uint32_t)nthreads > rows) nthreads = rows ? (int32_t)rows : 1;
  or_job* jobs = (or_job*)calloc((size_t)nthreads, sizeof(or_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  if (!jobs || !th) { free(jobs); free(th); return 2; }
  atomic_uint next_row;
  atomic_init(&next_row, 0u);
  for (int t = 0; t < nthreads; ++t) {
    or_job* j = &jobs[t];
    j->sc = sc; j->cam = *cam; j->width = width; j->height = height;
    j->y0 = y0; j->y1 = y1; j->next_row = &next_row;
    j->hash_seed = hash_seed; j->post_id = post_id; j->frame_nb = frame_nb;
    j->moved = moved; j->bounces = bounces; j->tfb = tfb; j->rgba = rgba;
  }
  if (nthreads == 1) {
    render_band(&jobs[0]);
  } else {
    for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, render_band, &jobs[t]);
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  }
  memset(&g_last_stats, 0, sizeof g_last_stats);
  for (int t = 0; t < nthreads; ++t) {
    g_last_stats.calls += jobs[t].stats.calls;
    g_last_stats.mesh_hits += jobs[t].stats.mesh_hits;
    g_last_stats.nmap_hits += jobs[t].stats.nmap_hits;
  }
  free(jobs);
  free(th);
  return 0;
}
