/*
 * ptamd.h — C-ABI of the MI355X-native path-tracing megakernel (libptamd.so).
 *
 * This is the drop-in boundary for the ONE hot path of DavidPeicho/cuda-pathtracer:
 *     cudaError_t raytrace(...)   and   void setupFunctionTables()
 *     (reference: cuda_opengl/include/shaders/raytrace.h:9-17, implemented in
 *      cuda_opengl/src/shaders/raytrace.cu:287-375; called from
 *      cuda_opengl/src/gpu_processor.cpp:375-377 and cuda_opengl/src/main.cpp:170)
 * plus the device-side data the reference hands to it (scene upload:
 * cuda_opengl/src/scene/scene.cpp:202-283,370-391; textures/cubemaps:
 * cuda_opengl/src/gpu_processor.cpp:68-238).
 *
 * Conventions: plain C, plain pointers and sizes, no C++/STL/torch types.  Every
 * function returns 0 (PTAMD_OK) on success and a non-zero ptamd_status otherwise; it
 * never throws, never calls exit().  The message of the last failure on the calling
 * thread is returned by ptamd_get_last_error().  Calls on one context must be
 * serialised by the caller (the reference is single-threaded: gpu_processor.cpp:254).
 *
 * There is NO CPU fallback behind this API: compute entry points fail with
 * PTAMD_ERR_HIP when no gfx950 device is usable.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  PTAMD_OK = 0,
  PTAMD_ERR_ARG = 1,     /* null pointer / out-of-range id / bad size */
  PTAMD_ERR_HIP = 2,     /* a HIP runtime call failed (message has hipGetErrorString) */
  PTAMD_ERR_IO = 3,      /* loader: file missing / unparsable */
  PTAMD_ERR_LIMIT = 4    /* scene exceeds a documented capacity */
} ptamd_status;

/* ---- POD layouts: byte-compatible with cuda_opengl/include/scene/scene_data.h -------- */

typedef struct { float x, y, z; } ptamd_float3;
typedef struct { float x, y; } ptamd_float2;

/* scene::Face, scene_data.h:46-53 — 112 bytes */
typedef struct {
  ptamd_float3 vertices[3];
  ptamd_float3 normals[3];
  ptamd_float2 texcoords[3];
  ptamd_float3 tangent;
  uint32_t material_id;
} ptamd_face;

/* scene::Material, scene_data.h:95-100 — 16 bytes (12 + alignment pad) */
typedef struct {
  int32_t diffuse_spec_map;   /* id into the texture table (RGBA float: rgb = albedo, a = specular) */
  int32_t normal_map;         /* id into the texture table (RGB float) or -1 */
  float ior;
  int32_t _pad;
} ptamd_material;

/* scene::LightProp, scene_data.h:109-115 — 32 bytes */
typedef struct {
  ptamd_float3 color;
  ptamd_float3 vec;
  float emission;
  float radius;
} ptamd_light;

/* scene::Camera, scene_data.h:123-133 — 64 bytes.  u and v are ignored by the kernel
 * exactly as in the reference (generateRay recomputes them, intersection.cuh:84-87). */
typedef struct {
  ptamd_float3 position;
  ptamd_float3 dir;
  ptamd_float3 u;
  ptamd_float3 v;
  float fov_x;
  float speed;
  float aperture;
  float focus_dist;
} ptamd_camera;

/* scene::Texture, scene_data.h:31-37, with the data pointer replaced by an offset (in
 * floats) into one texel blob */
typedef struct {
  int32_t w, h, nb_chan;
  uint32_t _pad;
  uint64_t offset;
} ptamd_texture_desc;

/* Flattened scene::SceneData + the global texture table (scene_data.h:71-87).
 * Faces are mesh-major in storage order: mesh m owns faces
 * [sum(mesh_sizes[0..m)), +mesh_sizes[m]).  The global face index in that order is the
 * tie-break key of the nearest-hit search (intersection.cuh:179-196: first wins). */
typedef struct {
  const ptamd_face* faces;            uint32_t n_faces;
  const uint32_t* mesh_sizes;         uint32_t n_meshes;
  const ptamd_material* materials;    uint32_t n_materials;
  const ptamd_light* lights;          uint32_t n_lights;
  const ptamd_texture_desc* textures; uint32_t n_textures;
  const float* texels;                uint64_t n_texel_floats;
} ptamd_scene_desc;

/* ---- errors ------------------------------------------------------------------------ */

const char* ptamd_get_last_error(void);
const char* ptamd_version(void);
/* 16 hex digits: sha256 of the device sources (csrc/pt_kernels.hip, pt_device.h, pt_launch.h) and the compiler flags this
 * library's code object was built from.  Profiles carry it (profiles/pmc_latest.json), so that counters of one build are
 * never used to price another. */
const char* ptamd_build_id(void);

/* ---- host-side scene loader (replaces scene::Scene::upload's parsing half,
 *      scene.cpp:86-170,202-262,304-358 and material_loader.cpp:153-401) --------------- */

typedef struct ptamd_host_scene ptamd_host_scene;

/* Parses a .scene file and the OBJ/MTL it names, decoding the textures the MTL names with the
 * built-in decoder (ptamd_image_loadf below).  flags:
 *   PTAMD_LOAD_FIX_BACKSLASHES  normalise '\\' to '/' in MTL texture paths (default off = reference-
 *                               on-Linux behaviour: such textures fail to load and degrade to 1x1
 *                               constants, material_loader.cpp:97-104)
 *   PTAMD_LOAD_NO_IMAGES        do not open image files at all: every texture degrades to its 1x1 constant */
#define PTAMD_LOAD_FIX_BACKSLASHES 1u
#define PTAMD_LOAD_NO_IMAGES       2u
int  ptamd_host_scene_load(const char* scene_path, uint32_t flags, ptamd_host_scene** out);

/* stbi_loadf(path, &w, &h, &nb_chan, STBI_default) replacement (material_loader.cpp:97, gpu_processor.cpp:99):
 * decodes a JPEG (baseline / extended / progressive), a PNG (every colour type, bit depth and interlace mode) or a
 * Radiance .hdr (RGBE, flat or run-length coded) to w*h*nb_chan floats, nb_chan as stb_image reports it (JPEG: 1 for
 * grayscale files and 3 otherwise; PNG: 1..4; HDR: 3).  8-bit sources are linearised the way stbi_loadf does it
 * (colour channels pow(v/255, 2.2f) in single precision, the alpha of 2- and 4-channel images v/255); HDR pixels are
 * (r, g, b) * 2^(e-136), no gamma.  Pixels are bit-identical to stb_image 2.16's, the reference's decoder
 * (tests/test_ref_thirdparty.py).  Other formats: PTAMD_ERR_IO.  ptamd_image_load8 returns the 8-bit pixels of a
 * JPEG or PNG (stbi_load).  Free either buffer with ptamd_image_free. */
int  ptamd_image_loadf(const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, float** data);
int  ptamd_image_load8(const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, uint8_t** data);
void ptamd_image_free(void* data);
/* Writes 8-bit pixels (1 = gray, 2 = gray+alpha, 3 = RGB, 4 = RGBA; row 0 = top) as an uncompressed PNG: the
 * image-output helper for a headless host (the reference only ever presents through GL). */
int  ptamd_image_save_png(const char* path, const uint8_t* pixels, int32_t w, int32_t h, int32_t channels);
/* stbir_resize_float(in, in_w, in_h, 0, out, out_w, out_h, 0, channels) replacement (material_loader.cpp:358,367):
 * the rescale applied when a material's diffuse and specular maps differ in size.  Bit-identical to
 * stb_image_resize 0.95 (Catmull-Rom on growing axes, Mitchell otherwise, clamped edges). */
int  ptamd_image_resize_float(const float* in, int32_t in_w, int32_t in_h, float* out, int32_t out_w, int32_t out_h,
                              int32_t channels);

/* A host may inject its own decoder instead, as stb_image is for the reference (stbi_loadf):
 * `load` returns 0 and a w*h*nb_chan float buffer (already linearised the way stbi_loadf does it:
 * colour channels pow(v/255, 2.2), alpha v/255) or non-zero when the file cannot be decoded;
 * `release` frees that buffer.  With load == NULL this is ptamd_host_scene_load.  The loader then
 * applies material_loader.cpp:243-401 (diffuse rgb + specular a packed into one RGBA texture, 1x1
 * fallbacks, normal maps registered as they are, de-duplication by name). */
typedef int (*ptamd_image_load_fn)(void* user, const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, float** data);
typedef void (*ptamd_image_free_fn)(void* user, float* data);
int  ptamd_host_scene_load_ex(const char* scene_path, uint32_t flags, ptamd_image_load_fn load,
                              ptamd_image_free_fn release, void* user, ptamd_host_scene** out);
/* Names (as written in the MTL) of image files that could not be loaded. */
uint32_t ptamd_host_scene_unloaded_count(const ptamd_host_scene* s);
const char* ptamd_host_scene_unloaded_name(const ptamd_host_scene* s, uint32_t i);
void ptamd_host_scene_free(ptamd_host_scene* s);
/* Borrowed views into the loaded scene, valid until ptamd_host_scene_free. */
int  ptamd_host_scene_desc(const ptamd_host_scene* s, ptamd_scene_desc* out);
int  ptamd_host_scene_camera(const ptamd_host_scene* s, ptamd_camera* out);
/* "" when the .scene has no cubemap line; "0xRRGGBB" constant syntax is returned as is. */
const char* ptamd_host_scene_cubemap(const ptamd_host_scene* s);

/* Cubemap helpers (gpu_processor.cpp:37-57,68-132; texture_utils.cpp:5-52):
 * constant-colour 1x1x6 cubemap from 0xRRGGBB, and cube-cross -> 6 faces unpack
 * (+x,-x,+y,-y,+z,-z, float4 per texel, w = 0).  out must hold 6*size*size*4 floats. */
int ptamd_cubemap_from_color(uint32_t rgb, float out[24]);
int ptamd_cubemap_from_cross(const float* cross, uint32_t width, uint32_t height,
                             uint32_t nb_chan, float* out, uint32_t* out_size);

/* ---- device context ------------------------------------------------------------------ */

typedef struct ptamd_context ptamd_context;

int  ptamd_create(int32_t device_ordinal, ptamd_context** out);
void ptamd_destroy(ptamd_context* ctx);

/* Deep-copies a flattened scene to the device and builds the traversal structures
 * (replaces scene.cpp:177-283,370-391 + gpu_processor.cpp:178-238).  Host arrays are
 * copied; the caller keeps ownership.  Capacity: fewer than 2^32 texel floats in total (PTAMD_ERR_LIMIT). */
int ptamd_upload_scene(ptamd_context* ctx, const ptamd_scene_desc* scene, uint32_t* out_scene_id);
/* faces: 6*size*size float4 in +x,-x,+y,-y,+z,-z order (gpu_processor.cpp:134-153). */
int ptamd_upload_cubemap(ptamd_context* ctx, const float* faces, uint32_t size, uint32_t* out_cubemap_id);

/* setupFunctionTables() (raytrace.cu:360-375).  The reference copies four device
 * function pointers to the host; here post-process dispatch is a switch inside the
 * kernel, so this resolves every kernel entry point of the gfx950 code object
 * (hipFuncGetAttributes) and fails with PTAMD_ERR_HIP when the device image cannot be used. */
int ptamd_setup_function_tables(ptamd_context* ctx);

/* ---- the hot path ---------------------------------------------------------------------
 * ptamd_raytrace == one reference raytrace() call (raytrace.cu:287-325): 1 sample per
 * pixel, frame counter kept in the context (the reference's function-static `seed`,
 * raytrace.cu:296-300): moved -> counter = 0; ++counter; hash_seed = WangHash(counter).
 * Bounce count is the reference's hard-coded 3 (static_samples = 1, raytrace.cu:243).
 *   surface_rgba8 : device pointer, width*height*4 bytes, row 0 = top of the picture
 *                   (replaces the cudaArray of the GL renderbuffer, raytrace.cu:270)
 *   temporal_framebuffer : device pointer, float3[width*height], reference row-flipped
 *                   index (raytrace.cu:252); borrowed, like the reference's
 *   stream        : hipStream_t (NULL = default stream).  The launch is asynchronous.
 */
int ptamd_raytrace(ptamd_context* ctx, void* surface_rgba8, uint32_t scene_id, uint32_t cubemap_id,
                   const ptamd_camera* cam, uint32_t width, uint32_t height, void* stream,
                   float* temporal_framebuffer, int32_t moved, uint32_t post_id);

typedef enum {
  PTAMD_KERNEL_AUTO = 0,          /* the shipped default: PTAMD_KERNEL_BVH_RESTART (PTAMD_KERNEL_BVH_PERSISTENT for single-frame
                                     launches on scenes that fit in LDS: no resolve pass per launch) */
  PTAMD_KERNEL_BRUTE_FORCE = 1,   /* the reference algorithm: every face, LDS-staged, wave-uniform; 1 thread = 1 pixel */
  PTAMD_KERNEL_BVH = 2,           /* stackless ordered BVH walk, LDS-staged nodes + triangles; 1 thread = 1 pixel */
  PTAMD_KERNEL_BVH_PERSISTENT = 3, /* same walk in persistent waves with mid-path lane refill (ballot + mbcnt) */
  PTAMD_KERNEL_BVH_BLOCKWISE = 4, /* persistent workgroups; live rays of each bounce compacted + octant-sorted through LDS */
  PTAMD_KERNEL_BVH_SPLIT = 5,     /* shader waves own the paths, traverser waves pull their rays from LDS and restart lanes */
  PTAMD_KERNEL_BVH_RESTART = 6,   /* persistent waves, lanes asynchronous per walk: a round ends without waiting for its few
                                     stragglers (they keep their place in the tree), finished lanes shade, ended paths restart
                                     at once from a pool of fresh paths that is refilled a whole tile at a time */
  PTAMD_KERNEL_BVH_RESTART_FMA = 7 /* OPT-IN, NOT bit-exact: the same kernel compiled with floating-point contraction allowed (a * b + c
                                     fused wherever the compiler likes — what the reference's own build permits: nvcc's default
                                     --fmad=true, cuda_opengl/CMakeLists.txt:20-22).  Every other kernel kind executes the
                                     reference's operation sequence unfused and equals the CPU oracle bit for bit; this one is
                                     held to the measured drift between faithful builds of the integrator instead (BASELINE.md
                                     section 5: all but <= 1e-4 of the pixels of the headline scene identical, mean image within
                                     2e-6).  Same launch shapes as PTAMD_KERNEL_BVH_RESTART (batched frames, bands, pipelining);
                                     no counters (ptamd_raytrace_stats), never selected by PTAMD_KERNEL_AUTO */
} ptamd_kernel_kind;

/* Explicit form used by the bench, the tests and the multi-GPU row split. */
typedef struct {
  void* surface_rgba8;
  float* temporal_framebuffer;
  void* stream;
  ptamd_camera camera;
  uint32_t scene_id, cubemap_id;
  uint32_t width, height;      /* FULL frame size (seeds and ray generation use it) */
  uint32_t row_begin, row_end; /* surface rows [row_begin, row_end) rendered by this call */
  uint32_t frame_nb;           /* >= 1: the reference's `seed`; hash_seed = WangHash(frame_nb) */
  uint32_t bounces;            /* iterations of the raytrace.cu:67 loop when !moved (reference: 3) */
  int32_t moved;
  uint32_t post_id;            /* 0 none, 1 grayscale, 2 sepia, 3 invert (raytrace.cu:327-357) */
  uint32_t kernel;             /* ptamd_kernel_kind */
  uint32_t band_local_buffers; /* 0: buffers are full-frame; 1: they hold only the row band */
  uint32_t frame_count;        /* 0 or 1: one frame.  N > 1 (static frames, persistent kernels only): frames
                                  frame_nb .. frame_nb+N-1 in ONE launch — same accumulator and final surface
                                  as N consecutive calls, bit for bit; intermediate surfaces are not produced */
  uint32_t machine_share;      /* persistent kernels: 0 or 1 = size the grid to the whole GPU; k > 1 = to 1/k of it, so that
                                  k launches in flight (one per stream) co-reside instead of queueing behind each other —
                                  what a multi-GPU host does with its small per-GPU bands (results do not depend on it).
                                  Launches in flight on different streams may share one context: batched launches park
                                  their samples in a per-stream scratch owned by the context.  Calls on one context must
                                  still come from one host thread at a time (as for the reference's raytrace()) */
  /* Interleaved bands of a multi-GPU split (0 or 1 rank: off).  The frame is cut into bands of interleave_rows rows (a
   * multiple of 8); band j belongs to rank j % interleave_ranks; this launch renders the bands of interleave_rank and
   * stores them one after the other in band-local buffers (band_local_buffers must be 1, row_begin 0, row_end height,
   * kernel PTAMD_KERNEL_AUTO / _BVH_RESTART).  Expensive and cheap parts of the picture are thereby spread over all
   * ranks; pixels are the same as in any other split (seeds come from frame coordinates). */
  uint32_t interleave_ranks, interleave_rank, interleave_rows;
  /* != 0: the temporal framebuffer counts as zero on entry — the launch starts a new accumulation, exactly as if the
   * caller had cleared the rows it renders first (the reference clears once, gpu_processor.cpp:255, and restarts an
   * accumulation only through `moved`); saves the clear and the read.  With frame_count = N it applies to the first
   * frame of the batch. */
  uint32_t reset_accumulation;
  /* != 0: never pipeline this launch inside the library (below, "Back-to-back launches"): path-tracing kernel and resolve pass
   * both run on launch->stream, sized to the whole GPU (or to machine_share), whatever the state of the stream.  0: the library
   * decides per launch (it pipelines when the stream's previous launch is still running).  For hosts that want run-to-run
   * identical launch shapes, e.g. under a profiler. */
  uint32_t no_pipelining;
} ptamd_launch;

/* Asynchronous on launch->stream.  Once a configuration (frame size, frame_count, stream) has been launched once — the
 * first launch sizes that stream's sample scratch, which synchronises and allocates — later launches only enqueue
 * kernels and memsets, so a host may capture them into a hipGraph (hipStreamBeginCapture on launch->stream) and replay it:
 * tests/test_gpu_parity.py test_batched_launch_is_graph_capturable.  The captured launch keeps its frame_nb / seeds.
 * What a captured launch pins, and how the library enforces it: the stream's in-stream sample slab and the launch's ring slot of
 * tile-ticket heads are baked into the graph.  From the capture on (until ptamd_release_captured for that stream)
 *   - a launch on that stream that would need a LARGER slab returns PTAMD_ERR_LIMIT instead of reallocating it under the graph
 *     (the same or smaller configurations, eager or captured, are fine: stream order protects the slab; launches the library
 *     pipelines use slabs of their own);
 *   - the ring slot is taken out of the rotation, so no later launch of the context shares its ticket heads with a replay;
 *   - the stream's scratch survives the "17th stream" eviction (a context keeps at most 16 per-stream scratches; when all 16
 *     are pinned the launch that needs a 17th returns PTAMD_ERR_LIMIT);
 *   - a launch that would have to SIZE the slab inside the capture returns PTAMD_ERR_LIMIT (launch the configuration once eagerly).
 * Memory per stream: one slab of rows x width x 12 bytes x min(frame_count, 4) (longer batches are issued as consecutive launches
 * of four frames: same bits), plus three more of the same size once the library has pipelined launches of that stream.
 *
 * Back-to-back launches.  A host that issues launches of the default kernel on ONE stream without waiting for them (the
 * reference's render loop does not wait: gpu_processor.cpp:365-386) gets them pipelined by the library: when the stream's
 * previous launch has not finished, the new one is sized to half the GPU and its path-tracing kernel runs on an internal stream,
 * its accumulate / tonemap pass on the caller's stream behind an event — two launches are then resident side by side and
 * the tail of one is covered by the bulk of the next.  Stream semantics are unchanged: everything the launch writes that the
 * caller can see (accumulator, surface) is written on the caller's stream, in order.  Not applied when machine_share > 1 (the
 * caller runs its own pipeline), during graph capture, or for ptamd_raytrace_stats. */
int ptamd_raytrace_ex(ptamd_context* ctx, const ptamd_launch* launch);
/* Tells the library that the graphs captured on `stream` are gone (or will not be replayed any more): its sample slab may be
 * reallocated again and its pinned ring slots return to the rotation.  PTAMD_OK also when nothing was pinned. */
int ptamd_release_captured(ptamd_context* ctx, void* stream);
/* Rows an interleaved launch renders (= rows its band-local buffers must hold): the bands j = rank, rank + ranks, ... of
 * band_rows rows each, the last band of the frame possibly shorter. */
uint32_t ptamd_interleaved_rows(uint32_t height, uint32_t ranks, uint32_t rank, uint32_t band_rows);
int ptamd_reset_frame_counter(ptamd_context* ctx);
uint32_t ptamd_wang_hash(uint32_t a); /* raytrace.cu:275-285 */

/* ---- measurement / introspection ----------------------------------------------------- */

typedef struct {
  uint64_t rays;            /* intersect() calls */
  uint64_t nodes_visited;   /* BVH nodes whose box was tested */
  uint64_t tris_tested;     /* Moller-Trumbore tests issued */
  uint64_t mesh_hits;       /* intersect() calls won by a mesh face (one 16 B texel fetch each) */
  uint64_t nmap_hits;       /* of those, on normal-mapped materials (one 12 B fetch each) */
  uint64_t samples;         /* pixels rendered */
  uint64_t wave_node_iters; /* wave-level executions of the box-test loop body (lane utilisation = nodes_visited / (64 * this)) */
  uint64_t wave_tri_iters;  /* wave-level executions of the triangle-test loop body */
  uint64_t fetch_events;    /* split kernel: wave-level ray fetches by traverser waves */
  uint64_t fetch_rays;      /* split kernel: rays handed out by those fetches */
  /* idle lane-slots of the box-test loop (64 * wave_node_iters = nodes_visited + these three), tile kernel: */
  uint64_t idle_unstarted;  /* lane traces no ray in this call (path ended earlier, or outside the frame) */
  uint64_t idle_finished;   /* lane's walk is over, the wave is still walking */
  uint64_t idle_parked;     /* lane left the current box phase (holds a leaf / just finished) */
} ptamd_trace_stats;

/* Renders like ptamd_raytrace_ex with an instrumented build of the selected kernel and
 * returns exact traversal counts (synchronous; outputs are written as usual). */
int ptamd_raytrace_stats(ptamd_context* ctx, const ptamd_launch* launch, ptamd_trace_stats* out);

/* Where the waves of the last ptamd_raytrace_stats launch of PTAMD_KERNEL_BVH_RESTART spent their shader-clock cycles, summed
 * over the waves (instrumented build only): out[0] pool refill (tickets, path_begin), [1] box phases of the wide walk, [2] its
 * leaf phases, [3] r1 + light loop (light loop + shading + bookkeeping = [4] - [0] - [1] - [2]), [4] the whole round loop, [5] leaf phases entered, [6] node fetches of the four-wide walk (issue to data), [7] its visits as a whole
 * (fetch, box tests, pushes, pops), [8] shading record fetch and decode, [9] path_post ([8] included) + parking the sample, [10] path_post up to the BSDF sample ([8] + the misses' loop), [11] the BSDF sample.
 * Synchronises. */
int ptamd_phase_cycles(ptamd_context* ctx, uint64_t out[12]);

/* Where a launch's time goes (measurement hook of the default kernel; profiles/r03_tail_*).  After ptamd_set_timeline(ctx, n)
 * every launch of PTAMD_KERNEL_BVH_RESTART with at most n waves in its grid records four device time stamps per wave
 * (hipDeviceAttributeWallClockRate ticks): kernel entry, scene staged, the moment the wave found no tile ticket left, exit.
 * ptamd_read_timeline synchronises, copies 4 * n_waves words ([wave][stamp]; waves of workgroup b are b * waves_per_group ..;
 * words of waves a launch did not have stay 0) and clears the buffer.  n = 0 switches the recording off.  Costs two scalar
 * loads per wave when off. */
int ptamd_set_timeline(ptamd_context* ctx, uint32_t max_waves);
int ptamd_read_timeline(ptamd_context* ctx, uint64_t* out, uint32_t n_waves, uint32_t* clock_khz);

typedef struct {
  uint32_t n_faces, n_lights, n_nodes, n_leaves, max_leaf_size, depth;
  uint32_t node_bytes, tri_bytes, lds_bytes_bvh, lds_bytes_brute;
  uint32_t n_nodes4, depth4;   /* the four-wide form of the tree (128-byte nodes), walked when the scene does not fit in LDS */
} ptamd_scene_info;
int ptamd_scene_info_get(ptamd_context* ctx, uint32_t scene_id, ptamd_scene_info* out);

/* Synchronises the device and returns how many times a bounded spin of the split kernel's
 * producer/consumer protocol timed out since the context was created (always 0 unless there is a bug;
 * a non-zero count means frames rendered by PTAMD_KERNEL_BVH_SPLIT are incomplete). */
int ptamd_device_error_count(ptamd_context* ctx, uint64_t* out);

/* Test hook.  The resolve pass reads the byte the reference's gamma + store sequence (raytrace.cu:262-268: powf(c, 1/2.2),
 * c * 255 through a truncating conversion) produces off a 256-step table built once per context with that very sequence
 * (PTAMD_GAMMA_TABLE=0: no table).  This compares the table form with the sequence itself for every binary32 value the
 * table form is used for (all positive values below the 256th step) plus samples of the values it hands back to the
 * sequence, on the device: *out_mismatches must be 0. */
int ptamd_gamma_table_selftest(ptamd_context* ctx, uint64_t* out_checked, uint64_t* out_mismatches);

/* Nearest-hit query on explicit rays through the device traversal (tests: BVH vs brute
 * force equivalence).  kernel: PTAMD_KERNEL_BRUTE_FORCE, PTAMD_KERNEL_BVH (binary walk) or
 * PTAMD_KERNEL_BVH_RESTART (the four-wide stack walk).  rays: n * {dir.xyz, origin.xyz};
 * out: n * {kind, index, t bits, pad}. */
int ptamd_trace_rays(ptamd_context* ctx, uint32_t scene_id, uint32_t kernel,
                     const float* rays_host, uint32_t n, int32_t* out_host);

/* Measurement hook (round 4): the four-wide walk WITHOUT a path around it.  Persistent waves pull rays {dir.xyz, origin.xyz} from a
 * queue in device memory and write {kind, index, t bits, 0} records (as ptamd_trace_rays with PTAMD_KERNEL_BVH_RESTART), a lane
 * taking its next ray as soon as `refill_min` lanes of its wave are idle.  No path state in registers, so the same walk runs at
 * config 0: 16 waves per CU (4 per SIMD), 512-node LDS treelet; 1: 20 waves (5 per SIMD), two workgroups with 256 nodes each;
 * 2: 24 waves (6 per SIMD), 256 nodes each; 3: 16 waves, 256 nodes.  Asynchronous on `stream`; rays_dev / out_dev are device
 * pointers; *out_waves_per_cu = waves resident per CU (occupancy query).  The queue head and the stack continuation belong to the
 * context: calls on ONE context must be ordered (one stream, or an event between streams); n < 2^31.
 * scripts/gpu_trace_queue.py, profiles/r04_notes.md. */
int ptamd_trace_rays_queue(ptamd_context* ctx, uint32_t scene_id, const float* rays_dev, uint32_t n, int32_t* out_dev, uint32_t config,
                           uint32_t refill_min, void* stream, uint32_t* out_waves_per_cu);

/* Host mirror of the device BVH walk (same node/triangle records, same float operations),
 * so the builder's "equals brute force" contract can be tested without a GPU.  This is a
 * test hook for the acceleration structure only; it renders nothing.
 * rays: n * {dir.xyz, origin.xyz}; out: n * {kind, index, t bits, 0};
 * counters (optional): [0] += nodes visited, [1] += triangles tested. */
int ptamd_host_bvh_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                         int32_t* out, uint64_t* counters);

/* The same for the four-wide form of the tree that scenes too big for LDS are walked in (per-lane stack, children
 * visited in a per-octant order).  counters (optional, 5 words): [0] += wide nodes visited, [1] += triangles tested,
 * [2] = depth of the wide tree, [3] / [4] += visits to the first 85 / 341 nodes (the part of the tree the device keeps
 * in LDS is numbered first). */
int ptamd_host_bvh4_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                          int32_t* out, uint64_t* counters);
/* ... over the same four-wide nodes in their 64-byte form (8-bit child planes on a per-node grid), built with leaves of at most
 * two triangles as the device uses them.  counters as above.  The quantised forms (this one and the eight-wide one below) are
 * for scenes whose finite coordinates stay within +-1e8: the library walks the float nodes beyond that. */
int ptamd_host_bvh4q_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                           int32_t* out, uint64_t* counters);

/* ... and for the eight-wide form with quantised child boxes (one 128-byte line per node: origin, per-axis power-of-two
 * scale, 8-bit planes), built with leaves of at most two triangles as the device uses it.  counters (optional, 6 words):
 * [0] += nodes visited, [1] += triangles tested, [2] = depth, [3] / [4] += visits to the first 73 / 585 nodes, [5] = node count. */
int ptamd_host_bvh8_trace(const ptamd_face* faces, uint32_t n_faces, const float* rays, uint32_t n,
                          int32_t* out, uint64_t* counters);

/* Plain device-memory helpers so that C/C++ hosts need not link HIP themselves. */
int ptamd_device_alloc(ptamd_context* ctx, size_t bytes, void** out);
int ptamd_device_free(ptamd_context* ctx, void* p);
int ptamd_device_memset(ptamd_context* ctx, void* p, int value, size_t bytes, void* stream);
int ptamd_device_to_host(ptamd_context* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int ptamd_stream_synchronize(ptamd_context* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
