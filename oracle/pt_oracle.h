/*
 * pt_oracle.h — CPU ORACLE for the path-tracing megakernel.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's per-pixel megakernel
 * (cuda_opengl/src/shaders/raytrace.cu:41-271 and the headers it includes).  It exists
 * to CHECK the HIP product path; nothing under cuda-pathtracer_amd/ may include, link
 * or call it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures, it cannot
 * be built here (needs nvcc + the CUDA toolkit headers + cuRAND; stand-in headers are
 * not allowed), and three of its arithmetic dependencies live outside its tree (cuRAND
 * XORWOW, CUDA libdevice/fast-math intrinsics, the texture unit's cubemap filter).
 * Those are restated here from their published definitions (see pt_oracle.c) and the
 * restatement is pinned only by hand-derived known-answer tests (tests/test_oracle_kat.py).
 *
 * Layouts mirror cuda_opengl/include/scene/scene_data.h (sizes checked by
 * _Static_assert in pt_oracle.c): Face 112 B, Material 16 B, LightProp 32 B, Camera 64 B.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } or_f3;
typedef struct { float x, y; } or_f2;

/* scene_data.h:46-53 */
typedef struct {
  or_f3 vertices[3];
  or_f3 normals[3];
  or_f2 texcoords[3];
  or_f3 tangent;
  uint32_t material_id;
} or_face;

/* scene_data.h:95-100 (12 B of fields, __align__(8) -> 16 B) */
typedef struct {
  int32_t diffuse_spec_map;
  int32_t normal_map;
  float ior;
  int32_t _pad;
} or_material;

/* scene_data.h:109-115 */
typedef struct {
  or_f3 color;
  or_f3 vec;
  float emission;
  float radius;
} or_light;

/* scene_data.h:123-133 */
typedef struct {
  or_f3 position;
  or_f3 dir;
  or_f3 u;
  or_f3 v;
  float fov_x;
  float speed;
  float aperture;
  float focus_dist;
} or_camera;

/* scene_data.h:31-37 (pointer graph flattened: the oracle borrows host pointers) */
typedef struct {
  int32_t w, h, nb_chan, _pad;
  const float* data;
} or_texture;

/* scene_data.h:59-62 */
typedef struct {
  uint32_t size;
  uint32_t _pad;
  const or_face* data;
} or_mesh;

/* scene_data.h:71-87 merged: one SceneData + the global texture table + the bound cubemap */
typedef struct {
  const or_mesh* meshes;       uint32_t n_meshes;    uint32_t _p0;
  const or_material* materials; uint32_t n_materials; uint32_t _p1;
  const or_light* lights;      uint32_t n_lights;    uint32_t _p2;
  const or_texture* textures;  uint32_t n_textures;  uint32_t _p3;
  /* gpu_processor.cpp:68-161: 6 faces (+x,-x,+y,-y,+z,-z) of size*size float4 */
  const float* cubemap;        uint32_t cubemap_size; uint32_t _p4;
} or_scene;

/* result of one intersect() call, for the BVH-vs-brute-force equivalence tests */
typedef struct {
  int32_t kind;      /* 0 miss, 1 mesh face, 2 light sphere */
  int32_t index;     /* global face index (mesh-major storage order) or light index */
  float t;           /* intersection.dist */
  float u, v;        /* barycentrics of the accepted face (0 for lights/miss) */
} or_hit;

/* ---- primitives (known-answer tests) ---- */
uint32_t or_wang_hash(uint32_t a);                                   /* raytrace.cu:275-285 */
void     or_xorwow_init(uint32_t seed, uint32_t state[6]);           /* cuRAND curand_init(seed,0,0) */
uint32_t or_xorwow_next(uint32_t state[6]);                          /* cuRAND curand() */
float    or_xorwow_uniform(uint32_t state[6]);                       /* cuRAND curand_uniform() */
void     or_sincosf(float x, float* s, float* c);
float    or_powf(float x, float y);
void     or_generate_ray(int x, int y, int half_w, int half_h, or_camera* cam,
                         or_f3* dir, or_f3* origin);                 /* intersection.cuh:75-97 */
int      or_intersect_triangle(const or_face* f, const or_f3* dir, const or_f3* origin,
                               or_f3* n, or_f2* uv, float* t, float* bu, float* bv);  /* :102-135 */
int      or_intersect_sphere(const or_f3* dir, const or_f3* origin, const or_light* l, float* t); /* :140-155 */
void     or_intersect(const or_scene* sc, const or_f3* dir, const or_f3* origin, or_hit* out);     /* :161-246 */
void     or_intersect_batch(const or_scene* sc, const float* rays, uint32_t n, int32_t* out);
void     or_exposure(const float in[3], float out[3]);               /* post_process.cuh:14-41 */
void     or_tex_cubemap(const or_scene* sc, float x, float y, float z, float out[4]);
uint32_t or_pack_rgba(const float rad[3]);                           /* raytrace.cu:231-232,266-268 */
void     or_post_process(uint32_t post_id, const float in[3], float out[3]); /* raytrace.cu:327-352 */

/*
 * One reference raytrace()+kernel() launch (raytrace.cu:212-325) restricted to surface rows
 * [y0, y1).  hash_seed = WangHash(frame_nb) is computed by the caller exactly as
 * raytrace.cu:321 does.  `bounces` = iterations of the raytrace.cu:67 loop when static
 * (reference default 3 == static_samples 1).  tfb is the FULL frame accumulator
 * (float3[W*H], reference row-flipped index raytrace.cu:252); rgba is the full W*H*4
 * surface (row 0 = top).  Returns 0 on success.
 */
int or_render(const or_scene* sc, const or_camera* cam, uint32_t width, uint32_t height,
              uint32_t y0, uint32_t y1, uint32_t hash_seed, int32_t frame_nb, int32_t moved,
              uint32_t post_id, int32_t bounces, float* tfb, uint8_t* rgba, int32_t nthreads);

/* trace statistics of the last or_render call on this thread group (brute force):
 * out[0] = intersect() calls, out[1] = mesh hits, out[2] = normal-mapped hits */
void or_last_stats(uint64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
