// ptamd_api.cpp — the C-ABI of libptamd.so (include/ptamd.h): device context, scene and
// cubemap upload, and the raytrace() replacement.
//
// Reference call path being replaced:
//   GPUProcessor::render  -> raytrace(...)           cuda_opengl/src/gpu_processor.cpp:375-377
//   raytrace()            -> kernel<<<...>>>(...)    cuda_opengl/src/shaders/raytrace.cu:287-325
// The frame counter that raytrace.cu keeps in a function-static (:296-300) lives in the
// context.  Pixel-invariant camera terms of generateRay (intersection.cuh:79-87) are
// computed here once per launch with the same float operations the kernel would do.
#include "../host/ptamd_internal.h"
#include "pt_device.h"
#include "pt_launch.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

namespace ptamd {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

const char* tuning_env(const char* name)
{
  const char* gate = std::getenv("PTAMD_TUNING");
  if (!gate || std::atoi(gate) != 1) return nullptr;
  return std::getenv(name);
}

namespace {

struct DeviceScene {
  float4* nodes = nullptr;
  float4* nodes4 = nullptr;   // the four-wide form of the same tree (8 float4 per node)
  float4* nodes8 = nullptr;   // ... and the eight-wide quantised form (8 float4 per node): walked instead when PTAMD_WIDE8=1 (tuning)
  float4* nodes4q = nullptr;  // ... and the four-wide form in 64-byte quantised nodes (4 float4 per node, numbered as nodes4)
  uint32_t n_nodes8 = 0, depth8 = 0;
  float4* tris_bvh = nullptr;
  float4* tris_brute = nullptr;
  float4* shade = nullptr;
  int4* materials = nullptr;
  float4* lights = nullptr;
  TexDesc* textures = nullptr;
  float* texels = nullptr;
  uint32_t n_faces = 0, n_lights = 0, n_nodes = 0, n_materials = 0, n_textures = 0;
  uint32_t n_bvh_tris = 0; // triangle records behind the BVH leaves (>= n_faces with split references)
  uint32_t n_nodes4 = 0, depth4 = 0;
  float extent = 0.0f;       // largest finite |coordinate| of the scene (bvh_builder.cpp)
  bool all_finite = true;    // no NaN or infinite vertex coordinate
  float margin_floor = 0.0f; // smallest inflation of any box face: what the slab test's rounding error must stay below
  ptamd_scene_info info{};
};

struct DeviceCubemap {
  float4* faces = nullptr;
  uint32_t size = 0;
  bool uniform = false;     // size 1 and the six texels' rgb bit-identical: every lookup returns color
  float color[3] = { 0.f, 0.f, 0.f };
};

} // namespace

} // namespace ptamd

struct ptamd_context {
  int device = 0;
  std::vector<ptamd::DeviceScene> scenes;
  std::vector<ptamd::DeviceCubemap> cubemaps;
  uint32_t frame_counter = 0; // raytrace.cu:296 `static unsigned int seed`
  unsigned long long* d_stats = nullptr;
  float* d_gamma = nullptr;               // 258 floats: the gamma step of the tonemap as a table (pt_kernels.hip: gamma_byte); null with PTAMD_GAMMA_TABLE=0
  // persistent variant: ring of tile ticket counters (one per in-flight launch) and grid sizing
  uint32_t* d_tickets = nullptr;
  uint32_t* d_heads = nullptr;   // kTicketRing sets of 8 ticket heads, PT_HEAD_STRIDE dwords apart (persistent kernel)
  uint32_t ticket_next = 0;
  std::vector<bool> heads_clean;   // per ring slot: its ticket heads are known to be zero (creation, or its last user's resolve pass)
  std::vector<bool> slot_pinned;   // per ring slot: baked into a captured graph (skipped by the rotation until ptamd_release_captured)
  int n_cus = 0;
  // resident workgroups per CU of the persistent kernels: depends on the scene's dynamic LDS bytes, so the cache is
  // keyed by them ([0] persistent, [1] blockwise, [2] split, [3] restart)
  struct Occupancy { size_t lds = ~(size_t)0; int blocks_per_cu = -1; } occupancy[5];   // ([4]: the contracted restart kernel)
  // parked samples of batched launches, one scratch per stream: launches on one stream are ordered, launches on
  // different streams of one context (frames in flight, ptamd_launch.machine_share) must not share a buffer
  // Four slabs per stream.  [0..2] are used in turn by pipelined launches (megakernel on an internal stream, below): the
  // megakernel of launch N+1 writes its samples while the resolve pass of launch N still reads its own, and with three of
  // them the megakernel of launch N+2 — next on launch N's internal stream — does not have to wait for that resolve pass
  // either (two slabs: a 60 us bubble per launch, two event hops and the pass itself); [3] belongs to launches that stay on
  // the caller's stream from start to end (one at a time, captured into a graph, instrumented, or part of the caller's own
  // pipeline): stream order alone protects it, also against replays of a captured launch.
  struct SampleScratch {
    void* stream = nullptr;
    float* buf[4] = { nullptr, nullptr, nullptr, nullptr };
    size_t bytes[4] = { 0, 0, 0, 0 };
    hipEvent_t mega_done[3] = { nullptr, nullptr, nullptr };   // megakernel of the last launch that used slab i has finished
    hipEvent_t resolved[3] = { nullptr, nullptr, nullptr };    // resolve pass of the last launch that used slab i has finished
    bool resolved_valid[3] = { false, false, false };
    hipEvent_t last_done = nullptr;                   // recorded behind every launch of this stream: is the host running ahead?
    uint32_t flip = 0;
    bool no_pipeline = false;     // the three pipelining slabs could not be allocated once: this stream's launches stay on the caller's stream
    // Graph capture (ptamd.h "What a captured launch pins"): a launch captured on this stream baked slab [3] and its ring slots of
    // ticket heads into a graph.  Until ptamd_release_captured the slab is not reallocated and the slots are not handed to anyone else.
    bool captured = false;
    std::vector<uint32_t> pinned_slots;
  };
  std::vector<SampleScratch> sample_scratch;
  // Consecutive launches on ONE caller stream overlap: the megakernel of a launch (which reads scene tables and writes only
  // the context's scratch) runs on one of two internal streams, its resolve pass (the only part that touches the caller's
  // accumulator and surface) on the caller's stream behind an event.  The tail of launch N — waves finishing the tiles
  // they hold at falling occupancy once the tickets are gone — is then filled by the first workgroups of launch N+1, for
  // a host that simply calls raytrace() again without synchronising (gpu_processor.cpp:365-386 does not).
  hipStream_t internal[2] = { nullptr, nullptr };
  bool overlap = true;                    // PTAMD_OVERLAP=0 (tuning): everything on the caller's stream
  bool wide4q = false;                    // PTAMD_WIDE4Q=1 (tuning): big scenes walk the 64-byte quantised four-wide nodes instead of the float ones (ahead by 2.8 % while the walk's LDS accesses went out as FLAT instructions, level since they are LDS instructions: profiles/r03_notes.md)
  bool wide8 = false;                     // PTAMD_WIDE8=1 (tuning): big scenes walk the eight-wide quantised nodes (measured 8 % slower: DESIGN.md §4)
  uint2* d_trace_spill = nullptr;             // ptamd_trace_rays_queue: global continuation of the walk-only kernel's stacks (grown on demand)
  struct { uint32_t config = ~0u; size_t lds = 0; int resident = 0; } trace_queue_cache;   // ... its last configuration: dynamic-LDS attribute set, blocks resident per CU
  size_t trace_spill_bytes = 0;
  unsigned long long* d_timeline = nullptr;   // ptamd_set_timeline: 4 time stamps per wave of the restart kernel
  uint32_t timeline_waves = 0;
  uint32_t default_kernel = PTAMD_KERNEL_BVH_RESTART; // what PTAMD_KERNEL_AUTO means
  bool default_kernel_is_builtin = true;              // false once PTAMD_DEFAULT_KERNEL pinned it
  uint32_t refill_min = 0; // 0 = choose per launch (see do_launch); PTAMD_REFILL_MIN pins it
  // restart kernel: a round of walks ends once fewer than min(round_min, entering lanes / round_div) lanes are unfinished
  // measured (round 2 sweep, 1080p x 4 spp x 4 bounces; re-run with scripts/gpu_ab.sh): round_min 16-32 and walk_min 4-6 are a flat optimum
  uint32_t round_min = 16, round_div = 4; // PTAMD_ROUND_MIN, PTAMD_ROUND_DIV
  uint32_t walk_min = 7;                  // restart kernel: a box phase ends once fewer lanes than this still walk (PTAMD_WALK_MIN; 4 / 5 / 7 / 8 / 10 / 12: 9331 / 9372 / 9405 / 9377 / 9338 / 9273 Msamples/s with the final shading code)
  bool pool_in_lds = true;                // restart kernel: pools of fresh paths in LDS when they fit (PTAMD_POOL_LDS=0: always the global slab)
  bool pool_in_lds_wide = false;          // ... also for scenes walked from L2 (PTAMD_POOL_LDS_WIDE=1).  Off since round 4: the 36 KB the pools took are four more LDS
                                          // entries of every lane's stack (7 -> 11: fewer pushes and pops through the global continuation, and the hand-scheduled visit
                                          // needs room for four entries in EVERY lane's LDS part): atrium 1 590 -> 1 727 Msamples/s, tessellated indoor 3 849 -> 3 865
  uint32_t treelet_nodes = 512;           // wide walk: nodes of the top of the tree staged in LDS (PTAMD_TREELET; with LDS pools 341 / 512 / 640: 1286 / 1291 / 1275)
  uint32_t walk_min4 = 16;                // the same threshold for the four-wide walk (PTAMD_WALK_MIN4; 1/4/8/16/24: 813/902/960/994/971 Msamples/s)
  bool short_rcp = true;                  // restart kernel: 7-instruction exact 1/det where the scene allows it (PTAMD_SHORT_RCP=0: always the full division)
  uint32_t tiles_per_ticket = 1;
  uint32_t xcd_regions = 0;               // restart kernel: XCD-local tile regions (0 never, 1 for scenes walked from L2, 2 always; PTAMD_XCD_REGIONS).  Off: measured -0.5 % on the atrium, -0.7 % on the headline (profiles/r04_notes.md)
};

namespace ptamd {
namespace {

constexpr size_t kLdsBudget = 64 * 1024;

// 1 / c for a positive power of two c (KParams::frame_nb_inv), else 0
float frame_nb_inverse(float c)
{
  uint32_t bits;
  std::memcpy(&bits, &c, 4);
  const uint32_t exponent = bits >> 23;   // sign bit included: negative values fail the range test
  if ((bits & 0x007FFFFFu) != 0u || exponent < 1u || exponent > 253u) return 0.0f;
  return 1.0f / c;
}
constexpr uint32_t kCompactMaxNodes = 896;   // 896 * 32 B = 28 KB of boxes below 0x8000 with 4 KB to spare for static LDS
constexpr float kQuantisedMaxExtent = 1.0e8f;   // largest |coordinate| of a scene walked over quantised nodes (nodes4q / nodes8): see do_launch
constexpr uint32_t kCompactMaxTris = 2047;   // a leaf's link code holds count << 11 | first triangle record in 15 bits (stage_scene)
constexpr size_t kShadeFloats = 28;   // 7 float4 per face (pt_kernels.hip: resolve_hit).  Round 4 re-measured on the atrium: a 128-byte stride (one line per record) -1.2 %, a 64-byte hot half + 64-byte cold half (one line, 17 MB instead of 30) level, -0.6 % on textured scenes (profiles/r04_notes.md)
constexpr float kBoxMargin = 1e-3f; // absolute box inflation, DESIGN.md "Conservative boxes"
constexpr uint32_t kMaxLeaf = 2;   // 2 / 3 / 4 = 10902 / 10839 / 10160 Msamples/s on the headline now that a box test costs 16 VALU and a triangle test ~67 (round 3, PTAMD_BVH_MAX_LEAF sweep: every bench configuration >= leaves of three)
constexpr uint32_t kTicketRing = 1024;
constexpr size_t kMaxScratchStreams = 16;   // sample scratches kept per context (one per stream that batches frames)
constexpr uint32_t kMaxFramesPerSlab = 4;   // a batched launch parks at most this many frames at a time: longer batches are issued as consecutive launches of <= 4 frames (the same bits by the contract of frame_count), so a stream's slab bytes do not depend on frame_count
#ifndef PT_PERSISTENT_THREADS
#define PT_PERSISTENT_THREADS 512
#endif
#ifndef PT_BW_THREADS
#define PT_BW_THREADS 512
#endif
constexpr uint32_t kPersistentThreads = PT_PERSISTENT_THREADS; // same macro as pt_kernels.hip

int hip_fail(const char* what, hipError_t e)
{
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  return PTAMD_ERR_HIP;
}

#define PT_HIP(call)                                         \
  do {                                                       \
    hipError_t _e = (call);                                  \
    if (_e != hipSuccess) return hip_fail(#call, _e);        \
  } while (0)

template <typename T>
int upload(T*& dst, const void* src, size_t bytes)
{
  dst = nullptr;
  if (bytes == 0) bytes = 16; // keep pointers valid for empty tables
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
  if (src) PT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  else PT_HIP(hipMemset(dst, 0, bytes));
  return PTAMD_OK;
}

// the same with `pad` zero bytes behind the table (reads that run past the last record stay inside the allocation)
template <typename T>
int upload_padded(T*& dst, const void* src, size_t bytes, size_t pad)
{
  dst = nullptr;
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&dst), bytes + pad));
  PT_HIP(hipMemset(reinterpret_cast<char*>(dst) + bytes, 0, pad));
  if (bytes) PT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return PTAMD_OK;
}

void free_scene(DeviceScene& s)
{
  void* ptrs[] = { s.nodes, s.nodes4, s.nodes8, s.nodes4q, s.tris_bvh, s.tris_brute, s.shade, s.materials, s.lights, s.textures, s.texels };
  for (void* q : ptrs) (void)hipFree(q);
  s = DeviceScene();
}

// KParams::far_table: children sit in slots by direction and a ray of octant o visits them in ascending (slot ^ o) order;
// byte c of the entry of octant o = the slots visited AFTER slot c
void fill_far_table(uint32_t t[16])
{
  for (uint32_t o = 0; o < 8; ++o) {
    uint64_t e = 0;
    for (uint32_t c = 0; c < 8; ++c) {
      uint32_t m = 0;
      for (uint32_t d = 0; d < 8; ++d) if ((d ^ o) > (c ^ o)) m |= 1u << d;
      e |= (uint64_t)m << (8 * c);
    }
    t[2 * o] = (uint32_t)e; t[2 * o + 1] = (uint32_t)(e >> 32);
  }
}

void free_scratch(ptamd_context::SampleScratch& c)
{
  for (int i = 0; i < 4; ++i) (void)hipFree(c.buf[i]);
  for (int i = 0; i < 3; ++i) {
    if (c.mega_done[i]) (void)hipEventDestroy(c.mega_done[i]);
    if (c.resolved[i]) (void)hipEventDestroy(c.resolved[i]);
  }
  if (c.last_done) (void)hipEventDestroy(c.last_done);
  c = ptamd_context::SampleScratch();
}

inline f3 hf3(ptamd_float3 v) { f3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }
inline f3 hadd(f3 a, f3 b) { f3 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
inline f3 hmuls(f3 a, float s) { f3 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
inline f3 hcross(f3 a, f3 b)
{
  f3 r;
  r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x;
  return r;
}
inline f3 hnormalize(f3 v)
{
  float inv_len = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  return hmuls(v, inv_len);
}

int validate_launch(const ptamd_context* ctx, const ptamd_launch* l)
{
  if (!ctx || !l) { set_error("ptamd_raytrace: null context or launch"); return PTAMD_ERR_ARG; }
  if (!l->surface_rgba8 || !l->temporal_framebuffer) { set_error("ptamd_raytrace: null output buffer"); return PTAMD_ERR_ARG; }
  if (l->scene_id >= ctx->scenes.size()) { set_error("ptamd_raytrace: scene_id out of range"); return PTAMD_ERR_ARG; }
  if (l->cubemap_id >= ctx->cubemaps.size()) { set_error("ptamd_raytrace: cubemap_id out of range"); return PTAMD_ERR_ARG; }
  if (l->post_id > 3) { set_error("ptamd_raytrace: post_id out of range (0..3)"); return PTAMD_ERR_ARG; }
  if (l->width == 0 || l->height == 0 || l->width > 65536 || l->height > 65536) { set_error("ptamd_raytrace: bad frame size"); return PTAMD_ERR_ARG; }
  if (l->row_begin > l->row_end || l->row_end > l->height) { set_error("ptamd_raytrace: bad row band"); return PTAMD_ERR_ARG; }
  if (l->frame_nb == 0) { set_error("ptamd_raytrace: frame_nb must be >= 1"); return PTAMD_ERR_ARG; }
  if (l->bounces == 0 || l->bounces > 1024) { set_error("ptamd_raytrace: bounces out of range (1..1024)"); return PTAMD_ERR_ARG; }
  if (l->frame_count > 4096) { set_error("ptamd_raytrace: frame_count out of range (<= 4096)"); return PTAMD_ERR_ARG; }
  if (l->frame_count > 1 && l->moved) { set_error("ptamd_raytrace: batched frames must be static (moved = 0)"); return PTAMD_ERR_ARG; }
  if (l->kernel > PTAMD_KERNEL_BVH_RESTART_FMA) { set_error("ptamd_raytrace: unknown kernel kind"); return PTAMD_ERR_ARG; }
  if (l->machine_share > 64) { set_error("ptamd_raytrace: machine_share out of range (<= 64)"); return PTAMD_ERR_ARG; }
  if (l->interleave_ranks > 1) {
    if (l->interleave_rank >= l->interleave_ranks || l->interleave_rows == 0 || l->interleave_rows % 8u != 0 || l->interleave_rows > 4096 ||
        !l->band_local_buffers || l->row_begin != 0 || l->row_end != l->height || l->moved ||
        (l->kernel != PTAMD_KERNEL_AUTO && l->kernel != PTAMD_KERNEL_BVH_RESTART && l->kernel != PTAMD_KERNEL_BVH_RESTART_FMA)) {
      set_error("ptamd_raytrace: interleaved bands need rank < ranks, rows a multiple of 8, band-local buffers, the whole frame as row range, "
                "a static frame and the default kernel");
      return PTAMD_ERR_ARG;
    }
  }
  return PTAMD_OK;
}

// the contracted instantiation of the restart kernel (pt_kernels_fma.hip)
extern "C" hipError_t ptamd_fma_restart_blocks_per_cu(int lds_resident, size_t lds_bytes, int* out);
extern "C" hipError_t ptamd_fma_launch_restart(const void* kparams, int lds_resident, size_t lds_bytes, uint32_t n_blocks, hipStream_t stream);

// later_chunk: the launch is the second or a later part of a batch the library cut into parts (kMaxFramesPerSlab): it follows
// its predecessor on the same stream by construction, so it is pipelined behind it whatever the caller's machine_share
int do_launch(ptamd_context* ctx, const ptamd_launch* l_in, bool stats, bool later_chunk = false)
{
  int rc = validate_launch(ctx, l_in);
  if (rc != PTAMD_OK) return rc;
  // PTAMD_KERNEL_BVH_RESTART_FMA: everything below treats the launch as one of the restart kernel; only the code object differs
  const bool fma = l_in->kernel == PTAMD_KERNEL_BVH_RESTART_FMA;
  ptamd_launch l_copy;
  const ptamd_launch* l = l_in;
  if (fma) {
    if (stats) { set_error("ptamd_raytrace_stats: the contracted kernel has no instrumented build"); return PTAMD_ERR_ARG; }
    l_copy = *l_in; l_copy.kernel = PTAMD_KERNEL_BVH_RESTART; l = &l_copy;
  }
  // Batches longer than kMaxFramesPerSlab: consecutive launches of at most that many frames.  frame_count = N is by contract the
  // same accumulator and final surface as N consecutive calls, so this changes no bit; what it bounds is the sample slab
  // (rows x width x 12 bytes x 4 frames whatever N: 0.4 GB at 4K instead of 1.6 GB at 16 spp, and four slabs per stream).
  if (l_in->frame_count > kMaxFramesPerSlab) {
    for (uint32_t k0 = 0; k0 < l_in->frame_count; k0 += kMaxFramesPerSlab) {
      ptamd_launch part = *l_in;
      part.frame_nb = l_in->frame_nb + k0;
      part.frame_count = l_in->frame_count - k0 < kMaxFramesPerSlab ? l_in->frame_count - k0 : kMaxFramesPerSlab;
      if (k0 > 0) part.reset_accumulation = 0;
      rc = do_launch(ctx, &part, stats, k0 > 0);
      if (rc != PTAMD_OK) return rc;
    }
    return PTAMD_OK;
  }
  PT_HIP(hipSetDevice(ctx->device));
  const DeviceScene& s = ctx->scenes[l->scene_id];
  const DeviceCubemap& cm = ctx->cubemaps[l->cubemap_id];

  KParams p;
  std::memset(&p, 0, sizeof p);
  p.nodes = s.nodes; p.tris_bvh = s.tris_bvh; p.tris_brute = s.tris_brute; p.shade = s.shade;
  p.materials = s.materials; p.lights = s.lights; p.textures = s.textures; p.texels = s.texels;
  p.cubemap = cm.faces; p.cubemap_size = cm.size;
  p.env_uniform = cm.uniform ? 1u : 0u; p.env_r = cm.color[0]; p.env_g = cm.color[1]; p.env_b = cm.color[2];
  p.gamma_table = ctx->d_gamma;
  p.n_faces = s.n_faces; p.n_lights = s.n_lights; p.n_nodes = s.n_nodes; p.n_bvh_tris = s.n_bvh_tris;
  p.nodes4 = s.nodes4; p.n_nodes4 = s.n_nodes4;
  fill_far_table(p.far_table);
  // finite edges of at most 2e8 per axis and unit directions: det = e1 . (dir x e2) is far below 2^125 (or NaN, which
  // both forms of the reciprocal pass on)
  p.small_det = ctx->short_rcp && s.all_finite && s.extent <= 1.0e8f ? 1u : 0u;

  // generateRay's pixel-invariant part (intersection.cuh:79-89)
  const ptamd_camera& cam = l->camera;
  const int half_w = (int)(l->width / 2u);
  const float screen_dist = (float)half_w / tanf(cam.fov_x * 0.5f);
  f3 down; down.x = 0.0f; down.y = -1.0f; down.z = 0.0f;
  f3 u = hnormalize(hcross(hf3(cam.dir), down));
  f3 v = hnormalize(hcross(u, hf3(cam.dir)));
  u = hmuls(u, -1.0f);
  p.cam_pos = hf3(cam.position);
  p.cam_p0 = hadd(hf3(cam.position), hmuls(hf3(cam.dir), screen_dist));
  p.cam_u = u; p.cam_v = v;
  p.focus_dist = cam.focus_dist; p.aperture = cam.aperture;

  p.width = l->width; p.height = l->height; p.row_begin = l->row_begin; p.row_end = l->row_end;
  p.hash_seed = ptamd_wang_hash(l->frame_nb);
  p.frame_nb_f = (float)(int)l->frame_nb;
  p.frame_nb_inv = frame_nb_inverse(p.frame_nb_f);
  p.is_static = l->moved ? 0 : 1;
  p.bounces = (int32_t)l->bounces;
  p.post_id = l->post_id;
  p.tfb = l->temporal_framebuffer;
  p.tfb_reset = l->reset_accumulation ? 1u : 0u;
  p.surface = static_cast<uint32_t*>(l->surface_rgba8);
  if (l->band_local_buffers) {
    p.tfb_row0 = l->height - l->row_end; // band covers accumulator rows [H-row_end, H-row_begin)
    p.surf_row0 = l->row_begin;
  }
  p.stats = stats ? ctx->d_stats : nullptr;
  p.error_flag = ctx->d_stats + 15;

  uint32_t which = l->kernel == PTAMD_KERNEL_AUTO ? ctx->default_kernel : l->kernel;
  // Box margins cover the slab test's rounding, at most 1.75 (|origin| + |plane|) * 2^-22, for origins inside the scene's extent
  // (bvh_builder.cpp).  A camera so far outside it that this bound exceeds the margin (e.g. 1e5 units from a
  // unit-sized scene) would need wider boxes: such launches test every face instead — the reference algorithm, exact
  // for any origin — inside the restart kernel (KParams::brute_walk: all its launch shapes keep working, interleaved
  // bands and batched frames included) or, for the other kernels, through the exhaustive tile kernel.
  const float cam_far = std::fmax(std::fabs(cam.position.x), std::fmax(std::fabs(cam.position.y), std::fabs(cam.position.z))) +
                        std::fabs(cam.aperture);
  // (2^-21, not the 2^-22 of a single fma: the centre / half-extent form rounds a slab distance twice — t(centre), then -+ half * |1/d| —
  // on top of the reciprocal's and -o/d's roundings: worst case about 1.75 (|origin| + |plane|) * 2^-22, bvh_builder.cpp)
  const bool far_origin = !((cam_far + s.extent) * (1.0f / 2097152.0f) <= s.margin_floor) && s.n_faces != 0;   // also true for NaN
  if (far_origin) {
    if (which == PTAMD_KERNEL_BVH_RESTART) p.brute_walk = 1u;
    else which = PTAMD_KERNEL_BRUTE_FORCE;
  }
  // only the restart kernel maps its tiles to the rows of interleaved bands; every other kernel would render the whole
  // frame into the band-local buffers (PTAMD_DEFAULT_KERNEL behind PTAMD_KERNEL_AUTO can ask for one)
  if (l->interleave_ranks > 1u && which != PTAMD_KERNEL_BVH_RESTART) {
    set_error("ptamd_raytrace: interleaved bands need the restart kernel (PTAMD_KERNEL_AUTO resolves to another one here)");
    return PTAMD_ERR_ARG;
  }
  const int kind = which == PTAMD_KERNEL_BRUTE_FORCE ? 1 : 2;
  const size_t lds = kind == 1 ? s.info.lds_bytes_brute : s.info.lds_bytes_bvh;
  // the LDS copy of a BVH addresses its boxes with 15 bits (pt_kernels.hip: stage_scene): 32 bytes per node, nodes first
  const bool resident = lds <= kLdsBudget && (kind == 1 || (s.n_nodes <= kCompactMaxNodes && s.n_bvh_tris <= kCompactMaxTris));
  hipStream_t stream = static_cast<hipStream_t>(l->stream);
  // Per-stream state of the persistent kernels: sample slabs, events (found or made here, once per launch)
  ptamd_context::SampleScratch* sc = nullptr;
  if (which == PTAMD_KERNEL_BVH_PERSISTENT || which == PTAMD_KERNEL_BVH_SPLIT || which == PTAMD_KERNEL_BVH_RESTART) {
    for (auto& c : ctx->sample_scratch) if (c.stream == l->stream) sc = &c;
    if (!sc) {
      if (ctx->sample_scratch.size() >= kMaxScratchStreams) {
        // a host cycling through short-lived streams: drop every scratch once nothing can be using them — except those a captured
        // graph has pinned (ptamd_release_captured frees them for this)
        size_t pinned = 0;
        for (auto& c : ctx->sample_scratch) pinned += c.captured ? 1u : 0u;
        if (pinned >= kMaxScratchStreams) {
          set_error("ptamd_raytrace: all 16 per-stream sample scratches of this context are pinned by captured graphs (ptamd_release_captured)");
          return PTAMD_ERR_LIMIT;
        }
        PT_HIP(hipDeviceSynchronize());
        std::vector<ptamd_context::SampleScratch> kept;
        for (auto& c : ctx->sample_scratch) { if (c.captured) kept.push_back(c); else free_scratch(c); }
        ctx->sample_scratch.swap(kept);
      }
      ctx->sample_scratch.emplace_back();
      sc = &ctx->sample_scratch.back();
      sc->stream = l->stream;
    }
  }
  // Launches of the restart kernel that a host issues back to back on one stream are pipelined by the library
  // (ptamd_context::internal): when the previous launch of this stream has not finished yet — the host is running ahead —
  // this one is sized to half the GPU and its megakernel goes to an internal stream, so that the two are resident side by
  // side and the tail of one is covered by the bulk of the other, exactly what a host gets from two streams and
  // machine_share = 2.  Not when the caller runs its own pipeline (machine_share > 1), captures a graph or wants counters;
  // a host that waits for every frame gets whole-GPU launches on its own stream as before.
  bool pipelined = ctx->overlap && which == PTAMD_KERNEL_BVH_RESTART && (l->machine_share <= 1u || later_chunk) && !stats;
  bool capturing = false;
  if (stream != nullptr) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) capturing = true;
  }
  if (capturing || (sc && sc->no_pipeline) || l->no_pipelining) pipelined = false;
  if (pipelined && !later_chunk) pipelined = sc->last_done != nullptr && hipEventQuery(sc->last_done) == hipErrorNotReady;
  // PTAMD_KERNEL_AUTO, one frame per launch (the reference's interactive loop, ptamd_raytrace) on an LDS-resident scene,
  // one launch at a time: the persistent kernel writes the surface itself, the restart kernel would add its resolve
  // pass to every launch (1080p, one launch per spp, one at a time: 6.03 vs 5.89 Gsamples/s).
  if (l->kernel == PTAMD_KERNEL_AUTO && which == PTAMD_KERNEL_BVH_RESTART && ctx->default_kernel_is_builtin && l->frame_count <= 1 &&
      resident && l->interleave_ranks <= 1 && l->machine_share <= 1 && !pipelined && !p.brute_walk)
    which = PTAMD_KERNEL_BVH_PERSISTENT;
  const bool overlap = pipelined;
  hipError_t e;
  if (far_origin && !p.brute_walk && l->frame_count > 1) {
    // batched frames == consecutive launches by contract: issue them that way
    for (uint32_t k = 0; k < l->frame_count; ++k) {
      ptamd_launch one = *l;
      one.frame_nb = l->frame_nb + k; one.frame_count = 1; one.kernel = PTAMD_KERNEL_BRUTE_FORCE;
      if (k > 0) one.reset_accumulation = 0;
      rc = do_launch(ctx, &one, stats);
      if (rc != PTAMD_OK) return rc;
    }
    return PTAMD_OK;
  }
  if (l->frame_count > 1 && which != PTAMD_KERNEL_BVH_PERSISTENT && which != PTAMD_KERNEL_BVH_SPLIT && which != PTAMD_KERNEL_BVH_RESTART) {
    set_error("ptamd_raytrace: frame_count > 1 needs a persistent kernel (PTAMD_KERNEL_AUTO, _BVH_PERSISTENT, _BVH_RESTART or _BVH_SPLIT)");
    return PTAMD_ERR_ARG;
  }
  if (which == PTAMD_KERNEL_BVH_BLOCKWISE) {
    // persistent workgroups over 32x16 super-tiles; tickets 0..n_blocks-1 are static
    const uint32_t rows = l->row_end - l->row_begin;
    const uint32_t st_rows = (PT_BW_THREADS / 64u) * 2u; // super-tile = 32 x (2 * waves) pixels
    p.tiles_x = (l->width + 31u) / 32u;
    p.n_tiles = p.tiles_x * ((rows + st_rows - 1u) / st_rows);
    if (p.n_tiles == 0) return PTAMD_OK;
    ptamd_context::Occupancy& occ = ctx->occupancy[1];
    const size_t occ_key = resident ? lds : 0;
    if (occ.blocks_per_cu < 0 || occ.lds != occ_key) {
      int q = -1;
      e = blockwise_blocks_per_cu(resident, lds, &q);
      if (e != hipSuccess || q < 1) { occ.blocks_per_cu = -1; return hip_fail("occupancy query of the blockwise kernel", e); }
      occ.blocks_per_cu = q; occ.lds = occ_key;
    }
    const int bpc = occ.blocks_per_cu;
    uint32_t n_blocks = (uint32_t)ctx->n_cus * (uint32_t)bpc;
    if (n_blocks > p.n_tiles) n_blocks = p.n_tiles;
    p.tile_counter = ctx->d_tickets + (ctx->ticket_next++ % kTicketRing);
    PT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.tile_counter), (int)n_blocks, 1, stream));
    e = launch_megakernel_blockwise(p, resident, lds, stats, n_blocks, stream);
  } else if (which == PTAMD_KERNEL_BVH_PERSISTENT || which == PTAMD_KERNEL_BVH_SPLIT || which == PTAMD_KERNEL_BVH_RESTART) {
    const bool split = which == PTAMD_KERNEL_BVH_SPLIT;
    const bool restart = which == PTAMD_KERNEL_BVH_RESTART;
    const uint32_t count = l->frame_count > 1 ? l->frame_count : 1u;

    uint32_t rows = l->row_end - l->row_begin;
    if (l->interleave_ranks > 1u) {
      if (!restart) { set_error("ptamd_raytrace: interleaved bands need the restart kernel behind PTAMD_KERNEL_AUTO"); return PTAMD_ERR_ARG; }
      rows = ptamd_interleaved_rows(l->height, l->interleave_ranks, l->interleave_rank, l->interleave_rows);
      p.ilv_ranks = l->interleave_ranks; p.ilv_rank = l->interleave_rank; p.ilv_rows = l->interleave_rows;
      // the launch's buffers hold `rows` rows: parked samples and the resolve pass address them as the band [0, rows)
      // with band-local buffers; only the restart kernel's tile -> frame-row map knows about the interleaving
      p.row_begin = 0; p.row_end = rows;
      p.tfb_row0 = l->height - rows;
      p.surf_row0 = 0;
    }
    p.y_limit = l->row_end;
    p.tiles_x = (l->width + PT_TILE_W - 1u) / PT_TILE_W;
    p.n_tiles = p.tiles_x * ((rows + PT_TILE_H - 1u) / PT_TILE_H);
    if (p.n_tiles == 0) return PTAMD_OK;
    if ((uint64_t)p.n_tiles * count >= (1ull << 31)) {   // (tile, frame) tickets are 32-bit
      set_error("ptamd_raytrace: rows x width x frame_count too large for one launch (split the batch)");
      return PTAMD_ERR_LIMIT;
    }
    // restart kernel on a scene that does not fit in LDS: the four-wide walk.  Its per-lane stack needs at most
    // 3 x (depth of the wide tree) entries of 8 bytes; as many as fit the workgroup's LDS share live in LDS
    // ([entry][lane], 512 bytes per entry and wave), the rest in a global slab.
    size_t launch_lds = lds;
    if (restart && !resident) {
      // the eight-wide quantised form instead of the four-wide one (knob; not for the instrumented / time-stamp / far-origin instantiations)
      // the quantised node forms decode a plane as fma(plane, scale / d, fma(origin, 1 / d, -o / d)): with the 1e30 that stands in
      // for 1 / 0 (axis-parallel rays) the inner fma stays finite for coordinates up to kQuantisedMaxExtent; beyond it the float
      // nodes are walked, whose planes overflow one by one (an infinite slab distance is still a correct one)
      const bool quantised_ok = s.extent <= kQuantisedMaxExtent;
      const bool wide8 = ctx->wide8 && quantised_ok && !stats && !fma && !p.brute_walk && !ctx->d_timeline && s.n_nodes8 != 0;
      if (wide8) { p.nodes4 = s.nodes8; p.n_nodes4 = s.n_nodes8; p.wide8 = 1u; }
      const bool wide4q = !wide8 && ctx->wide4q && quantised_ok && !stats && !fma && !p.brute_walk && !ctx->d_timeline && s.nodes4q != nullptr;
      if (wide4q) { p.nodes4 = s.nodes4q; p.wide8 = 2u; }
      const uint32_t node_bytes = wide4q ? 64u : 128u;
      const uint32_t need = (wide8 ? 7u * s.depth8 : 3u * s.depth4) + 1u;   // a visit stacks all hit children but the nearest
      const uint32_t waves = restart_threads(false) / 64u;
      const uint32_t share = 160u * 1024u / restart_wide_blocks_per_cu() - 256u;   // LDS bytes of one resident workgroup
      // the top of the tree (breadth-first numbering: nodes 0..340 are its first five levels when full) goes to LDS too:
      // 512 nodes = 64 KB of the one workgroup's 160 KB, then 7 stack entries per lane
      // ... and the waves' pools of fresh paths (PT_POOL_LDS_BYTES each), behind the stacks
      const uint32_t pools = (ctx->pool_in_lds && ctx->pool_in_lds_wide) ? waves * PT_POOL_LDS_BYTES : 0u;
      // (the same LDS bytes hold twice as many 64-byte nodes)
      // chunk-major treelet (pt_kernels.hip: PT_TREELET_SOA): a region of fixed size whatever the number of nodes staged
      const uint32_t region = restart_treelet_region_bytes();
      uint32_t treelet_want = ctx->treelet_nodes * (128u / node_bytes);
      if (region && treelet_want > region / node_bytes) treelet_want = region / node_bytes;
      uint32_t treelet = treelet_want < p.n_nodes4 ? treelet_want : p.n_nodes4;
      if (!region && treelet * node_bytes + waves * 512u * 4u + pools > share) treelet = (share - pools - waves * 512u * 4u) / node_bytes;   // keep >= 4 stack entries
      const uint32_t treelet_bytes = region ? (treelet ? region : 0u) : treelet * node_bytes;
      uint32_t fit = (share - pools - treelet_bytes) / (waves * 512u);
      if (const char* ev = tuning_env("PTAMD_STACK_LDS")) { int v = std::atoi(ev); if (v >= 1 && (uint32_t)v <= fit) fit = (uint32_t)v; }   // tuning knob
      p.treelet_nodes = treelet;
      p.stack_lds_entries = need < fit ? need : fit;
      p.stack_spill_entries = need - p.stack_lds_entries;
      launch_lds = (size_t)treelet_bytes + (size_t)p.stack_lds_entries * waves * 512u;
      if (pools) {
        p.pool_lds_offset = (uint32_t)launch_lds;
        if (!p.pool_lds_offset) p.pool_lds_offset = 16u;
        launch_lds = p.pool_lds_offset + pools;
      }
    }
    if (restart && resident) {
      // pools of fresh paths in LDS when two workgroups with their scene copies leave room for them (PT_POOL_LDS_BYTES
      // per wave); else in a global slab (3 KiB per wave, L2-resident)
      const uint32_t waves = restart_threads(true) / 64u;
      const size_t with_pools = ((lds + 15u) & ~(size_t)15u) + (size_t)waves * PT_POOL_LDS_BYTES;
      const size_t blocks_wanted = (24u + waves - 1u) / waves;             // 24 waves per CU
      if (ctx->pool_in_lds && with_pools * blocks_wanted + 1024u <= 160u * 1024u) {
        p.pool_lds_offset = (uint32_t)((lds + 15u) & ~(size_t)15u);
        if (p.pool_lds_offset == 0) p.pool_lds_offset = 16u;               // (an empty scene: keep the flag non-zero)
        launch_lds = p.pool_lds_offset + (size_t)waves * PT_POOL_LDS_BYTES;
      }
    }
    ptamd_context::Occupancy& occ = ctx->occupancy[split ? 2 : (restart ? (fma ? 4 : 3) : 0)];
    const size_t occ_key = resident ? (restart ? launch_lds : lds) : (restart ? launch_lds + 1u : 0);
    if (occ.blocks_per_cu < 0 || occ.lds != occ_key) {
      int q = -1;
      e = split ? split_blocks_per_cu(resident, lds, &q)
                : (restart ? (fma ? ptamd_fma_restart_blocks_per_cu(resident ? 1 : 0, launch_lds, &q) : restart_blocks_per_cu(resident, launch_lds, &q)) : persistent_blocks_per_cu(resident, lds, &q));
      if (e != hipSuccess || q < 1) { occ.blocks_per_cu = -1; return hip_fail("occupancy query of the persistent kernel", e); }
      occ.blocks_per_cu = q; occ.lds = occ_key;
    }
    const int bpc = occ.blocks_per_cu;
    // waves that take tile tickets: every wave of a persistent block, the shader waves of a split block
    const uint32_t waves_per_block = split ? split_shader_waves() : (restart ? restart_threads(resident) / 64u : kPersistentThreads / 64u);
    uint32_t n_blocks = (uint32_t)ctx->n_cus * (uint32_t)bpc;
    p.sample_count = count;
    p.frame_nb0 = l->frame_nb;
    // Mid-path lane refill pays once paths are long enough for dead lanes to dominate the box loop
    // (measured, batched 1080p: 4 bounces 4.65 vs 4.46 Gsamples/s without/with, 5: 3.90 vs 4.13,
    // 6: 3.44 vs 3.92, 8: 2.89 vs 3.71); below that, whole-wave refill keeps primary rays coherent.
    p.refill_min = ctx->refill_min ? ctx->refill_min : (l->bounces >= 5 ? 16u : 64u);
    const uint32_t tiles_per_ticket = ctx->tiles_per_ticket;
    const uint32_t share = overlap ? (l->machine_share > 2u ? l->machine_share : 2u) : l->machine_share;
    if (share > 1u) n_blocks = n_blocks / share > 0u ? n_blocks / share : 1u;
    const uint32_t n_tickets = (p.n_tiles * count + tiles_per_ticket - 1u) / tiles_per_ticket;
    const uint32_t useful = (n_tickets + waves_per_block - 1u) / waves_per_block;
    if (n_blocks > useful) n_blocks = useful;
    // XCD-local regions (pt_kernels.hip: region_tile): the ticket -> tile map that keeps every XCD on a compact part of the frame.
    // Needs whole groups of eight workgroups (one per XCD) and one tile per ticket.
    if (restart && tiles_per_ticket == 1u && n_blocks >= 8u && (uint64_t)p.n_tiles * count < (1ull << 28) &&
        (ctx->xcd_regions == 2u || (ctx->xcd_regions == 1u && !resident))) {
      n_blocks &= ~7u;
      p.xcd_regions = 1u;
    }
    // seeds of frames frame_nb+1.. are hashed on the device; the tonemap uses the last frame number
    if (count > 1) p.frame_nb_f = (float)(int)(l->frame_nb + count - 1u);
    p.frame_nb_inv = frame_nb_inverse(p.frame_nb_f);
    // the restart kernel parks every sample (also of a single frame: pt_resolve_kernel accumulates and tonemaps) and
    // keeps a 3 KiB pool of fresh paths per wave
    const bool parks = count > 1 || restart;
    uint32_t scratch_slab = 0;
    if (parks) {
      const size_t sample_bytes = ((size_t)count * rows * l->width * 3u * sizeof(float) + 255u) & ~(size_t)255u;
      const size_t pool_bytes = (restart && !p.pool_lds_offset) ? (size_t)n_blocks * waves_per_block * 192u * sizeof(float4) : 0u;
      const size_t spill_bytes = (size_t)n_blocks * waves_per_block * p.stack_spill_entries * 512u;
      const size_t need = sample_bytes + pool_bytes + spill_bytes + 16u;
      uint32_t& slab = scratch_slab;
      slab = overlap ? sc->flip % 3u : 3u;
      for (int i = 0; i < 3 && overlap; ++i) {
        if (!sc->mega_done[i]) PT_HIP(hipEventCreateWithFlags(&sc->mega_done[i], hipEventDisableTiming));
        if (!sc->resolved[i]) PT_HIP(hipEventCreateWithFlags(&sc->resolved[i], hipEventDisableTiming));
      }
      // Slab [3] belongs to launches that stay on the caller's stream; [0..2] exist only for streams whose launches the library
      // has actually pipelined (a host that waits for every frame pays for ONE slab, not four).  Growing synchronises
      // (the old buffer may be in use): it happens once per stream and configuration — for the pipelining slabs at the first launch
      // that finds its predecessor still running, which costs that launch its overlap and no more.
      auto grow = [&](uint32_t i) -> int {
        PT_HIP(hipStreamSynchronize(stream));
        for (hipStream_t is : ctx->internal) if (is) PT_HIP(hipStreamSynchronize(is));
        (void)hipFree(sc->buf[i]);
        sc->buf[i] = nullptr; sc->bytes[i] = 0;
        hipError_t me = hipMalloc(reinterpret_cast<void**>(&sc->buf[i]), need);
        if (me != hipSuccess) { sc->buf[i] = nullptr; (void)hipGetLastError(); return PTAMD_ERR_HIP; }
        sc->bytes[i] = need;
        return PTAMD_OK;
      };
      if (overlap) {
        bool ok = true;
        for (uint32_t i = 0; i < 3u && ok; ++i) if (need > sc->bytes[i]) ok = grow(i) == PTAMD_OK;
        if (!ok) {
          // no room for the pipelining slabs: this stream renders unpipelined from now on (slab [3] on the caller's stream)
          for (uint32_t i = 0; i < 3u; ++i) { (void)hipFree(sc->buf[i]); sc->buf[i] = nullptr; sc->bytes[i] = 0; }
          sc->no_pipeline = true;
          return do_launch(ctx, l_in, stats, later_chunk);
        }
      } else if (need > sc->bytes[3]) {
        if (capturing) {
          set_error("ptamd_raytrace: a launch cannot size its stream's sample slab inside a graph capture: issue this configuration once eagerly first");
          return PTAMD_ERR_LIMIT;
        }
        if (sc->captured) {
          set_error("ptamd_raytrace: a captured graph pins this stream's sample slab; a larger launch would reallocate it under the graph "
                    "(ptamd_release_captured(ctx, stream) once the graph is gone)");
          return PTAMD_ERR_LIMIT;
        }
        if (grow(3u) != PTAMD_OK) return hip_fail("hipMalloc of the sample slab", hipErrorOutOfMemory);
      }
      p.samples_out = sc->buf[slab];
      p.pool = reinterpret_cast<float4*>(reinterpret_cast<char*>(sc->buf[slab]) + sample_bytes);
      p.stack_spill = reinterpret_cast<uint2*>(reinterpret_cast<char*>(sc->buf[slab]) + sample_bytes + pool_bytes);
    }
    p.round_min = ctx->round_min;
    p.round_div = ctx->round_div;
    p.round_div_m16 = (65536u + ctx->round_div - 1u) / ctx->round_div;
    p.walk_min = ctx->walk_min;
    p.walk_min4 = ctx->walk_min4;
    p.tiles_per_ticket = tiles_per_ticket;
    // tickets 0..n_waves-1 are taken statically by the waves; the shared counter hands out the rest
    uint32_t slot = ctx->ticket_next++ % kTicketRing;
    for (uint32_t tries = 0; ctx->slot_pinned[slot]; ++tries) {   // slots baked into captured graphs are not handed out again
      if (tries >= kTicketRing) { set_error("ptamd_raytrace: every ring slot of ticket heads is pinned by captured graphs (ptamd_release_captured)"); return PTAMD_ERR_LIMIT; }
      slot = ctx->ticket_next++ % kTicketRing;
    }
    if (capturing && sc) { ctx->slot_pinned[slot] = true; sc->pinned_slots.push_back(slot); sc->captured = true; }
    p.tile_counter = ctx->d_tickets + slot;
    p.n_static = n_blocks * waves_per_block;
    if (split) PT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.tile_counter), (int)p.n_static, 1, stream));
    hipStream_t mega_stream = stream;
    if (overlap) {
      // the megakernel touches nothing of the caller's: it may start before earlier work on the caller's stream has finished,
      // as soon as the slab's previous reader (the resolve pass three launches back) is done
      mega_stream = ctx->internal[sc->flip++ & 1u];
      if (sc->resolved_valid[scratch_slab]) PT_HIP(hipStreamWaitEvent(mega_stream, sc->resolved[scratch_slab], 0));
    }
    if (!split) {
      p.tile_heads = ctx->d_heads + (size_t)slot * 8u * PT_HEAD_STRIDE;
      // the whole ring is zeroed at creation and a launch that parks its samples has its resolve pass zero its heads
      // again (pt_resolve_kernel); only slots whose last user did not get that far are cleared here
      if (!ctx->heads_clean[slot]) PT_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.tile_heads), 0, 8u * PT_HEAD_STRIDE, mega_stream));
      ctx->heads_clean[slot] = false;
    }
    if (restart && !fma && ctx->d_timeline && n_blocks * waves_per_block <= ctx->timeline_waves) p.timeline = ctx->d_timeline;
    if (split) { p.tiles_per_ticket = 1; e = launch_megakernel_split(p, resident, lds, stats, n_blocks, stream); }
    else if (restart) e = fma ? ptamd_fma_launch_restart(&p, resident ? 1 : 0, launch_lds, n_blocks, mega_stream) : launch_megakernel_restart(p, resident, launch_lds, stats, n_blocks, mega_stream);
    else e = launch_megakernel_persistent(p, resident, lds, stats, n_blocks, stream);
    if (e == hipSuccess && overlap) {
      PT_HIP(hipEventRecord(sc->mega_done[scratch_slab], mega_stream));
      PT_HIP(hipStreamWaitEvent(stream, sc->mega_done[scratch_slab], 0));
    }
    if (e == hipSuccess && parks) {
      e = launch_resolve(p, stream);
      if (e == hipSuccess && !split) ctx->heads_clean[slot] = true;
      if (e == hipSuccess && overlap) {
        // whoever writes this slab next (a megakernel on an internal stream) waits for this pass
        PT_HIP(hipEventRecord(sc->resolved[scratch_slab], stream));
        sc->resolved_valid[scratch_slab] = true;
      }
    }
    if (e == hipSuccess && !capturing && ctx->overlap && sc) {
      if (!sc->last_done) PT_HIP(hipEventCreateWithFlags(&sc->last_done, hipEventDisableTiming));
      PT_HIP(hipEventRecord(sc->last_done, stream));
    }
  } else {
    e = launch_megakernel(p, kind, resident, lds, stats, stream);
  }
  if (e != hipSuccess) return hip_fail("megakernel launch", e);
  return PTAMD_OK;
}

} // namespace
} // namespace ptamd

using namespace ptamd;

extern "C" {

const char* ptamd_get_last_error(void) { return g_last_error.c_str(); }
#ifndef PTAMD_BUILD_ID
#define PTAMD_BUILD_ID "unknown"
#endif
const char* ptamd_version(void) { return "ptamd 0.3 (gfx950) device code " PTAMD_BUILD_ID; }
const char* ptamd_build_id(void) { return PTAMD_BUILD_ID; }

uint32_t ptamd_interleaved_rows(uint32_t height, uint32_t ranks, uint32_t rank, uint32_t band_rows)
{
  if (ranks == 0 || rank >= ranks || band_rows == 0) return 0;
  uint32_t rows = 0;
  for (uint64_t y0 = (uint64_t)rank * band_rows; y0 < height; y0 += (uint64_t)ranks * band_rows)
    rows += (uint32_t)(y0 + band_rows <= height ? band_rows : height - y0);
  return rows;
}

uint32_t ptamd_wang_hash(uint32_t a)
{
  a = (a ^ 61u) ^ (a >> 16);
  a = a + (a << 3);
  a = a ^ (a >> 4);
  a = a * 0x27d4eb2du;
  a = a ^ (a >> 15);
  return a;
}

int ptamd_create(int32_t device_ordinal, ptamd_context** out)
{
  if (!out) { set_error("ptamd_create: null out"); return PTAMD_ERR_ARG; }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    set_error(std::string("ptamd_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
              "); this library has no CPU fallback");
    return PTAMD_ERR_HIP;
  }
  if (device_ordinal < 0 || device_ordinal >= n) { set_error("ptamd_create: device ordinal out of range"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(device_ordinal));
  std::unique_ptr<ptamd_context> ctx(new (std::nothrow) ptamd_context());
  if (!ctx) { set_error("ptamd_create: out of memory"); return PTAMD_ERR_ARG; }
  ctx->device = device_ordinal;
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_stats), 32 * sizeof(unsigned long long)));   // 0..12 counters, 14 self-test, 15 error flag, 16..21 phase cycles
  PT_HIP(hipMemset(ctx->d_stats, 0, 32 * sizeof(unsigned long long)));
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_tickets), kTicketRing * sizeof(uint32_t)));
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_heads), (size_t)kTicketRing * 8u * PT_HEAD_STRIDE * sizeof(uint32_t)));
  PT_HIP(hipMemset(ctx->d_heads, 0, (size_t)kTicketRing * 8u * PT_HEAD_STRIDE * sizeof(uint32_t)));
  ctx->heads_clean.assign(kTicketRing, true);
  ctx->slot_pinned.assign(kTicketRing, false);
  {
    const char* e = tuning_env("PTAMD_GAMMA_TABLE"); // tuning knob: 0 = pt_powf for every pixel
    if (!e || std::atoi(e) != 0) {
      PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_gamma), 258 * sizeof(float)));
      hipError_t ge = build_gamma_table(ctx->d_gamma, nullptr);
      if (ge == hipSuccess) ge = hipDeviceSynchronize();
      if (ge != hipSuccess) return hip_fail("ptamd_create: gamma table", ge);
    }
  }
  if (const char* e = tuning_env("PTAMD_OVERLAP")) ctx->overlap = std::atoi(e) != 0; // tuning knob
  if (ctx->overlap) {
    // the two internal streams of the launch pipeline, with their hardware queues brought up now (a stream's first
    // operation costs ~6 ms on this runtime: it would otherwise land in the frame where a host starts to run ahead)
    for (hipStream_t& is : ctx->internal) {
      PT_HIP(hipStreamCreateWithFlags(&is, hipStreamNonBlocking));
      PT_HIP(hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long), is));
      PT_HIP(hipStreamSynchronize(is));
    }
  }
  hipDeviceProp_t prop;
  PT_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  ctx->n_cus = prop.multiProcessorCount;
  if (const char* e = tuning_env("PTAMD_REFILL_MIN")) { // tuning knob
    int v = std::atoi(e);
    ctx->refill_min = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }
  if (const char* e = tuning_env("PTAMD_DEFAULT_KERNEL")) { // tuning knob: 1..6
    int v = std::atoi(e);
    if (v >= 1 && v <= 6) { ctx->default_kernel = (uint32_t)v; ctx->default_kernel_is_builtin = false; }
  }
  if (const char* e = tuning_env("PTAMD_ROUND_MIN")) { // tuning knob
    int v = std::atoi(e);
    ctx->round_min = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }
  if (const char* e = tuning_env("PTAMD_WALK_MIN")) { // tuning knob
    int v = std::atoi(e);
    ctx->walk_min = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }
  if (const char* e = tuning_env("PTAMD_WALK_MIN4")) { // tuning knob
    int v = std::atoi(e);
    ctx->walk_min4 = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }
  if (const char* e = tuning_env("PTAMD_SHORT_RCP")) ctx->short_rcp = std::atoi(e) != 0; // tuning knob
  if (const char* e = tuning_env("PTAMD_WIDE8")) ctx->wide8 = std::atoi(e) != 0; // tuning knob
  if (const char* e = tuning_env("PTAMD_WIDE4Q")) ctx->wide4q = std::atoi(e) != 0; // tuning knob
  if (const char* e = tuning_env("PTAMD_POOL_LDS")) ctx->pool_in_lds = std::atoi(e) != 0; // tuning knob
  if (const char* e = tuning_env("PTAMD_POOL_LDS_WIDE")) ctx->pool_in_lds_wide = std::atoi(e) != 0; // tuning knob
  if (const char* e = tuning_env("PTAMD_TREELET")) { // tuning knob
    int v = std::atoi(e);
    ctx->treelet_nodes = (uint32_t)(v < 0 ? 0 : (v > 1024 ? 1024 : v));
  }
  if (const char* e = tuning_env("PTAMD_ROUND_DIV")) { // tuning knob
    int v = std::atoi(e);
    ctx->round_div = (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
  }
  if (const char* e = tuning_env("PTAMD_XCD_REGIONS")) { // tuning knob
    int v = std::atoi(e);
    ctx->xcd_regions = (uint32_t)(v < 0 ? 0 : (v > 2 ? 2 : v));
  }
  if (const char* e = tuning_env("PTAMD_TILES_PER_TICKET")) {
    int v = std::atoi(e);
    ctx->tiles_per_ticket = (uint32_t)(v < 1 ? 1 : (v > 1024 ? 1024 : v));
  }
  *out = ctx.release();
  return PTAMD_OK;
}

void ptamd_destroy(ptamd_context* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // before anything is freed: megakernels may still be running on the non-blocking internal streams (they read the scene tables
  // and the ticket heads), resolve passes on the callers' streams
  (void)hipDeviceSynchronize();
  for (auto& s : ctx->scenes) free_scene(s);
  for (auto& c : ctx->cubemaps) (void)hipFree(c.faces);
  (void)hipFree(ctx->d_stats);
  (void)hipFree(ctx->d_gamma);
  (void)hipFree(ctx->d_tickets);
  (void)hipFree(ctx->d_heads);
  for (auto& c : ctx->sample_scratch) free_scratch(c);
  for (hipStream_t is : ctx->internal) if (is) (void)hipStreamDestroy(is);
  (void)hipFree(ctx->d_timeline);
  (void)hipFree(ctx->d_trace_spill);
  delete ctx;
}

int ptamd_upload_scene(ptamd_context* ctx, const ptamd_scene_desc* sc, uint32_t* out_scene_id)
{
  if (!ctx || !sc || !out_scene_id) { set_error("ptamd_upload_scene: null argument"); return PTAMD_ERR_ARG; }
  if ((sc->n_faces && !sc->faces) || (sc->n_materials && !sc->materials) || (sc->n_lights && !sc->lights) ||
      (sc->n_textures && !sc->textures) || (sc->n_texel_floats && !sc->texels) || (sc->n_meshes && !sc->mesh_sizes)) {
    set_error("ptamd_upload_scene: null table with non-zero count");
    return PTAMD_ERR_ARG;
  }
  if (sc->n_texel_floats >= (1ull << 32)) { set_error("ptamd_upload_scene: more than 2^32 texel floats"); return PTAMD_ERR_LIMIT; }
  uint64_t total = 0;
  for (uint32_t m = 0; m < sc->n_meshes; ++m) total += sc->mesh_sizes[m];
  if (total != sc->n_faces) { set_error("ptamd_upload_scene: mesh_sizes do not sum to n_faces"); return PTAMD_ERR_ARG; }
  for (uint32_t i = 0; i < sc->n_faces; ++i)
    if (sc->faces[i].material_id >= sc->n_materials) { set_error("ptamd_upload_scene: face material_id out of range"); return PTAMD_ERR_ARG; }
  for (uint32_t i = 0; i < sc->n_textures; ++i) {
    const ptamd_texture_desc& t = sc->textures[i];
    if (t.w < 1 || t.h < 1 || t.nb_chan < 1 || t.offset + (uint64_t)t.w * t.h * t.nb_chan > sc->n_texel_floats) {
      set_error("ptamd_upload_scene: texture descriptor out of the texel blob");
      return PTAMD_ERR_ARG;
    }
  }
  for (uint32_t i = 0; i < sc->n_materials; ++i) {
    const ptamd_material& m = sc->materials[i];
    if (m.diffuse_spec_map < 0 || (uint32_t)m.diffuse_spec_map >= sc->n_textures || sc->textures[m.diffuse_spec_map].nb_chan != 4 ||
        (m.normal_map >= 0 && ((uint32_t)m.normal_map >= sc->n_textures || sc->textures[m.normal_map].nb_chan < 3))) {
      set_error("ptamd_upload_scene: material texture id invalid (diffuse+spec must be 4-channel)");
      return PTAMD_ERR_ARG;
    }
  }

  Bvh bvh;
  // (the quantised node forms only where their tuning knob is set: nothing else can select them)
  int rc = build_bvh(sc->faces, sc->n_faces, kBoxMargin, kMaxLeaf, bvh, (ctx->wide8 ? kBvhForm8 : 0u) | (ctx->wide4q ? kBvhForm4q : 0u));
  if (rc != PTAMD_OK) return rc;

  // storage-order {e1,e2,v0,idx} records for the brute-force variant, and the shading records
  std::vector<float> brute((size_t)sc->n_faces * 12, 0.0f), shade((size_t)sc->n_faces * kShadeFloats, 0.0f);
  for (uint32_t i = 0; i < sc->n_faces; ++i) {
    const ptamd_face& f = sc->faces[i];
    float* t = &brute[(size_t)i * 12];
    t[0] = f.vertices[1].x - f.vertices[0].x; t[1] = f.vertices[1].y - f.vertices[0].y; t[2] = f.vertices[1].z - f.vertices[0].z;
    t[3] = f.vertices[2].x - f.vertices[0].x; t[4] = f.vertices[2].y - f.vertices[0].y; t[5] = f.vertices[2].z - f.vertices[0].z;
    t[6] = f.vertices[0].x; t[7] = f.vertices[0].y; t[8] = f.vertices[0].z;
    std::memcpy(&t[9], &i, 4);
    // self-contained shading record (one parallel burst of loads per hit instead of the dependent
    // face -> material -> texture descriptor -> texel chain of intersection.cuh:216-243): 28 floats =
    // n0 n1 n2 | uv0 uv1 uv2 | tangent | material id (sign bit: constant map) | ior | diffuse+spec map {w,h,nb_chan,offset}
    // or its one RGBA texel | normal map {..} (w = 0: none)
    float* s = &shade[(size_t)i * kShadeFloats];
    std::memcpy(s, f.normals, 36);
    std::memcpy(s + 9, f.texcoords, 24);
    std::memcpy(s + 15, &f.tangent, 12);
    std::memcpy(s + 18, &f.material_id, 4);
    const ptamd_material& m = sc->materials[f.material_id];
    std::memcpy(s + 19, &m.ior, 4);
    const ptamd_texture_desc& dt = sc->textures[m.diffuse_spec_map];
    if (dt.w == 1 && dt.h == 1) {
      // a 1x1 diffuse+specular map (every material of indoor.obj as the reference loads it on Linux): sampleTexture can
      // only ever return texel 0 (intersection.cuh:20-26: x = int(uv.x * 0)), so the record carries the texel itself
      // and the kernel skips the dependent texel load; flagged in the sign bit of the material id word
      std::memcpy(s + 20, sc->texels + dt.offset, 16);
      const uint32_t flagged = f.material_id | 0x80000000u;
      std::memcpy(s + 18, &flagged, 4);
    } else {
      const int32_t d4[4] = { dt.w, dt.h, dt.nb_chan, (int32_t)(uint32_t)dt.offset };
      std::memcpy(s + 20, d4, 16);
    }
    if (m.normal_map >= 0) {
      const ptamd_texture_desc& nt = sc->textures[m.normal_map];
      const int32_t n4[4] = { nt.w, nt.h, nt.nb_chan, (int32_t)(uint32_t)nt.offset };
      std::memcpy(s + 24, n4, 16);
      uint32_t word;
      std::memcpy(&word, s + 18, 4);
      word |= 0x40000000u;               // bit 30 of the material id word: the record's 7th float4 (normal map) is in use
      std::memcpy(s + 18, &word, 4);
    }
  }
  std::vector<int32_t> mats((size_t)sc->n_materials * 4, 0);
  for (uint32_t i = 0; i < sc->n_materials; ++i) {
    mats[i * 4 + 0] = sc->materials[i].diffuse_spec_map;
    mats[i * 4 + 1] = sc->materials[i].normal_map;
    std::memcpy(&mats[i * 4 + 2], &sc->materials[i].ior, 4);
  }
  std::vector<TexDesc> tex(sc->n_textures);
  for (uint32_t i = 0; i < sc->n_textures; ++i) {
    tex[i].w = sc->textures[i].w; tex[i].h = sc->textures[i].h; tex[i].nb_chan = sc->textures[i].nb_chan;
    tex[i].pad = 0; tex[i].offset = sc->textures[i].offset;
  }

  PT_HIP(hipSetDevice(ctx->device));
  DeviceScene d;
  d.n_faces = sc->n_faces; d.n_lights = sc->n_lights; d.n_nodes = bvh.n_nodes; d.n_bvh_tris = bvh.n_tris;
  d.extent = bvh.extent; d.all_finite = bvh.all_finite; d.margin_floor = bvh.margin_floor;
  d.n_nodes4 = bvh.n_nodes4; d.depth4 = bvh.depth4;
  d.n_nodes8 = bvh.n_nodes8; d.depth8 = bvh.depth8;
  d.n_materials = sc->n_materials; d.n_textures = sc->n_textures;
  // device copy of the lights: the radius only ever enters as radius * radius (intersection.cuh:147) — the same binary32
  // product whoever forms it — so the table carries the square in its place and every sphere test saves the multiply
  std::vector<ptamd_light> dev_lights(sc->lights, sc->lights + sc->n_lights);
  for (ptamd_light& dl : dev_lights) dl.radius = dl.radius * dl.radius;
  if ((rc = upload(d.nodes, bvh.nodes.data(), bvh.nodes.size() * 4)) ||
      (rc = upload(d.nodes4, bvh.nodes4.data(), bvh.nodes4.size() * 4)) ||
      (bvh.nodes8.empty() ? 0 : (rc = upload(d.nodes8, bvh.nodes8.data(), bvh.nodes8.size() * 4))) ||
      (bvh.nodes4q.empty() ? 0 : (rc = upload(d.nodes4q, bvh.nodes4q.data(), bvh.nodes4q.size() * 4))) ||
      (rc = upload_padded(d.tris_bvh, bvh.tris.data(), bvh.tris.size() * 4, 128)) ||   // (the merged wide walk reads eight 16-byte words from a leaf's first record)
      (rc = upload(d.tris_brute, brute.data(), brute.size() * 4)) ||
      (rc = upload(d.shade, shade.data(), shade.size() * 4)) ||
      (rc = upload(d.materials, mats.data(), mats.size() * 4)) ||
      (rc = upload(d.lights, dev_lights.data(), dev_lights.size() * sizeof(ptamd_light))) ||
      (rc = upload(d.textures, tex.data(), tex.size() * sizeof(TexDesc))) ||
      (rc = upload(d.texels, sc->texels, (size_t)sc->n_texel_floats * 4))) {
    free_scene(d);
    return rc;
  }
  d.info.n_faces = sc->n_faces; d.info.n_lights = sc->n_lights; d.info.n_nodes = bvh.n_nodes;
  d.info.n_leaves = bvh.n_leaves; d.info.max_leaf_size = bvh.max_leaf; d.info.depth = bvh.depth;
  d.info.node_bytes = 64; d.info.tri_bytes = 48;
  d.info.n_nodes4 = d.n_nodes4; d.info.depth4 = d.depth4;
  d.info.lds_bytes_bvh = bvh.n_nodes * 64u + bvh.n_tris * 48u;
  d.info.lds_bytes_brute = sc->n_faces * 48u;
  ctx->scenes.push_back(d);
  *out_scene_id = (uint32_t)ctx->scenes.size() - 1;
  return PTAMD_OK;
}

int ptamd_upload_cubemap(ptamd_context* ctx, const float* faces, uint32_t size, uint32_t* out_cubemap_id)
{
  if (!ctx || !faces || !out_cubemap_id || size == 0 || size > 16384) { set_error("ptamd_upload_cubemap: bad argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  DeviceCubemap c;
  c.size = size;
  if (size == 1) {
    c.uniform = true;
    for (int f = 1; f < 6; ++f) c.uniform = c.uniform && std::memcmp(faces + f * 4, faces, 12) == 0;
    std::memcpy(c.color, faces, 12);
  }
  int rc = upload(c.faces, faces, (size_t)6 * size * size * 16);
  if (rc != PTAMD_OK) return rc;
  ctx->cubemaps.push_back(c);
  *out_cubemap_id = (uint32_t)ctx->cubemaps.size() - 1;
  return PTAMD_OK;
}

int ptamd_setup_function_tables(ptamd_context* ctx)
{
  if (!ctx) { set_error("ptamd_setup_function_tables: null context"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  // resolves every kernel entry point in the gfx950 code object (hipFuncGetAttributes loads it on first use), so a
  // missing or mismatched device image fails here, as the reference's cudaMemcpyFromSymbol calls would (raytrace.cu:362-374)
  hipError_t e = resolve_kernels();
  if (e != hipSuccess) return hip_fail("ptamd_setup_function_tables: device code object", e);
  return PTAMD_OK;
}

int ptamd_reset_frame_counter(ptamd_context* ctx)
{
  if (!ctx) { set_error("ptamd_reset_frame_counter: null context"); return PTAMD_ERR_ARG; }
  ctx->frame_counter = 0;
  return PTAMD_OK;
}

int ptamd_raytrace(ptamd_context* ctx, void* surface_rgba8, uint32_t scene_id, uint32_t cubemap_id,
                   const ptamd_camera* cam, uint32_t width, uint32_t height, void* stream,
                   float* temporal_framebuffer, int32_t moved, uint32_t post_id)
{
  if (!ctx || !cam) { set_error("ptamd_raytrace: null argument"); return PTAMD_ERR_ARG; }
  // raytrace.cu:296-300
  uint32_t seed = ctx->frame_counter;
  if (moved) seed = 0;
  seed++;
  ptamd_launch l;
  std::memset(&l, 0, sizeof l);
  l.surface_rgba8 = surface_rgba8; l.temporal_framebuffer = temporal_framebuffer; l.stream = stream;
  l.camera = *cam; l.scene_id = scene_id; l.cubemap_id = cubemap_id;
  l.width = width; l.height = height; l.row_begin = 0; l.row_end = height;
  l.frame_nb = seed; l.bounces = 3; /* static_samples = 1 (raytrace.cu:243,66) */
  l.moved = moved; l.post_id = post_id; l.kernel = PTAMD_KERNEL_AUTO;
  int rc = do_launch(ctx, &l, false);
  if (rc == PTAMD_OK) ctx->frame_counter = seed;
  return rc;
}

int ptamd_raytrace_ex(ptamd_context* ctx, const ptamd_launch* launch) { return do_launch(ctx, launch, false); }

int ptamd_release_captured(ptamd_context* ctx, void* stream)
{
  if (!ctx) { set_error("ptamd_release_captured: null context"); return PTAMD_ERR_ARG; }
  for (auto& c : ctx->sample_scratch) {
    if (c.stream != stream) continue;
    for (uint32_t slot : c.pinned_slots) { ctx->slot_pinned[slot] = false; ctx->heads_clean[slot] = false; }   // (a replay may have been cut short: clear before reuse)
    c.pinned_slots.clear();
    c.captured = false;
  }
  return PTAMD_OK;
}

int ptamd_raytrace_stats(ptamd_context* ctx, const ptamd_launch* launch, ptamd_trace_stats* out)
{
  if (!ctx || !out) { set_error("ptamd_raytrace_stats: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  hipStream_t st = launch ? static_cast<hipStream_t>(launch->stream) : nullptr;
  PT_HIP(hipMemsetAsync(ctx->d_stats, 0, 13 * sizeof(unsigned long long), st));
  PT_HIP(hipMemsetAsync(ctx->d_stats + 16, 0, 12 * sizeof(unsigned long long), st));
  int rc = do_launch(ctx, launch, true);
  if (rc != PTAMD_OK) return rc;
  PT_HIP(hipStreamSynchronize(st));
  unsigned long long h[13];
  PT_HIP(hipMemcpy(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost));
  out->rays = h[0]; out->nodes_visited = h[1]; out->tris_tested = h[2];
  out->mesh_hits = h[3]; out->nmap_hits = h[4]; out->samples = h[5];
  out->wave_node_iters = h[6]; out->wave_tri_iters = h[7];
  out->fetch_events = h[8]; out->fetch_rays = h[9];
  out->idle_unstarted = h[10]; out->idle_finished = h[11]; out->idle_parked = h[12];
  return PTAMD_OK;
}

int ptamd_phase_cycles(ptamd_context* ctx, uint64_t out[12])
{
  if (!ctx || !out) { set_error("ptamd_phase_cycles: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipDeviceSynchronize());
  PT_HIP(hipMemcpy(out, ctx->d_stats + 16, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return PTAMD_OK;
}

int ptamd_device_error_count(ptamd_context* ctx, uint64_t* out)
{
  if (!ctx || !out) { set_error("ptamd_device_error_count: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipDeviceSynchronize());
  unsigned long long v = 0;
  PT_HIP(hipMemcpy(&v, ctx->d_stats + 15, sizeof v, hipMemcpyDeviceToHost));
  *out = v;
  return PTAMD_OK;
}

int ptamd_gamma_table_selftest(ptamd_context* ctx, uint64_t* out_checked, uint64_t* out_mismatches)
{
  if (!ctx || !out_checked || !out_mismatches) { set_error("ptamd_gamma_table_selftest: null argument"); return PTAMD_ERR_ARG; }
  *out_checked = 0; *out_mismatches = 0;
  if (!ctx->d_gamma) return PTAMD_OK;   // no table in use
  PT_HIP(hipSetDevice(ctx->device));
  float limit = 0.0f;                   // T[256]: the table form is used below it
  PT_HIP(hipMemcpy(&limit, ctx->d_gamma + 256, sizeof limit, hipMemcpyDeviceToHost));
  uint32_t limit_bits;
  std::memcpy(&limit_bits, &limit, 4);
  // every positive value below the limit, plus the 2^20 patterns from the limit on (those take the pt_powf form: must agree
  // trivially), plus the negative half's first 2^20 and the NaN patterns' first 2^20
  PT_HIP(hipMemsetAsync(ctx->d_stats + 14, 0, sizeof(unsigned long long), nullptr));
  const uint32_t ranges[3][2] = { { 0u, limit_bits + (1u << 20) }, { 0x80000000u, 1u << 20 }, { 0x7F800000u, 1u << 20 } };
  for (const auto& r : ranges) {
    hipError_t e = launch_gamma_selftest(ctx->d_gamma, r[0], r[1], ctx->d_stats + 14, nullptr);
    if (e != hipSuccess) return hip_fail("ptamd_gamma_table_selftest", e);
    *out_checked += r[1];
  }
  PT_HIP(hipDeviceSynchronize());
  unsigned long long bad = 0;
  PT_HIP(hipMemcpy(&bad, ctx->d_stats + 14, sizeof bad, hipMemcpyDeviceToHost));
  *out_mismatches = bad;
  return PTAMD_OK;
}

int ptamd_set_timeline(ptamd_context* ctx, uint32_t max_waves)
{
  if (!ctx) { set_error("ptamd_set_timeline: null context"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipDeviceSynchronize());
  (void)hipFree(ctx->d_timeline);
  ctx->d_timeline = nullptr; ctx->timeline_waves = 0;
  if (max_waves == 0) return PTAMD_OK;
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_timeline), (size_t)max_waves * 4u * sizeof(unsigned long long)));
  PT_HIP(hipMemset(ctx->d_timeline, 0, (size_t)max_waves * 4u * sizeof(unsigned long long)));
  ctx->timeline_waves = max_waves;
  return PTAMD_OK;
}

int ptamd_read_timeline(ptamd_context* ctx, uint64_t* out, uint32_t n_waves, uint32_t* clock_khz)
{
  if (!ctx || !out || n_waves > ctx->timeline_waves) { set_error("ptamd_read_timeline: bad argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipDeviceSynchronize());
  PT_HIP(hipMemcpy(out, ctx->d_timeline, (size_t)n_waves * 4u * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  PT_HIP(hipMemset(ctx->d_timeline, 0, (size_t)ctx->timeline_waves * 4u * sizeof(unsigned long long)));
  if (clock_khz) {
    int khz = 0;
    PT_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    *clock_khz = (uint32_t)khz;
  }
  return PTAMD_OK;
}

int ptamd_scene_info_get(ptamd_context* ctx, uint32_t scene_id, ptamd_scene_info* out)
{
  if (!ctx || !out || scene_id >= ctx->scenes.size()) { set_error("ptamd_scene_info_get: bad argument"); return PTAMD_ERR_ARG; }
  *out = ctx->scenes[scene_id].info;
  return PTAMD_OK;
}

int ptamd_trace_rays(ptamd_context* ctx, uint32_t scene_id, uint32_t kernel, const float* rays_host, uint32_t n,
                     int32_t* out_host)
{
  if (!ctx || scene_id >= ctx->scenes.size() || (n && (!rays_host || !out_host)) ||
      (kernel > PTAMD_KERNEL_BVH && kernel != PTAMD_KERNEL_BVH_RESTART)) {
    set_error("ptamd_trace_rays: bad argument");
    return PTAMD_ERR_ARG;
  }
  if (n == 0) return PTAMD_OK;
  PT_HIP(hipSetDevice(ctx->device));
  const DeviceScene& s = ctx->scenes[scene_id];
  KParams p;
  std::memset(&p, 0, sizeof p);
  p.nodes = s.nodes; p.tris_bvh = s.tris_bvh; p.tris_brute = s.tris_brute; p.lights = s.lights;
  p.n_faces = s.n_faces; p.n_lights = s.n_lights; p.n_nodes = s.n_nodes; p.n_bvh_tris = s.n_bvh_tris;
  p.nodes4 = s.nodes4; p.n_nodes4 = s.n_nodes4;
  fill_far_table(p.far_table);
  p.small_det = 0u;                           // caller-supplied directions need not be unit vectors
  p.stack_lds_entries = 3u * s.depth4 + 1u;   // PTAMD_KERNEL_BVH_RESTART: the wide walk, whole stack in LDS
  if (ctx->wide8 && s.extent <= kQuantisedMaxExtent && s.n_nodes8 != 0) { p.nodes4 = s.nodes8; p.n_nodes4 = s.n_nodes8; p.wide8 = 1u; p.stack_lds_entries = 7u * s.depth8 + 1u; }
  else if (ctx->wide4q && s.extent <= kQuantisedMaxExtent && s.nodes4q != nullptr) { p.nodes4 = s.nodes4q; p.wide8 = 2u; }
  float* d_rays = nullptr;
  int4* d_out = nullptr;
  PT_HIP(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)n * 24));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 16);
  if (e != hipSuccess) { (void)hipFree(d_rays); return hip_fail("hipMalloc", e); }
  int rc = PTAMD_OK;
  if ((e = hipMemcpy(d_rays, rays_host, (size_t)n * 24, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = launch_trace_rays(p, kernel == PTAMD_KERNEL_BRUTE_FORCE ? 1 : (kernel == PTAMD_KERNEL_BVH_RESTART ? 3 : 2), d_rays, n, d_out, nullptr)) != hipSuccess ||
      (e = hipDeviceSynchronize()) != hipSuccess ||
      (e = hipMemcpy(out_host, d_out, (size_t)n * 16, hipMemcpyDeviceToHost)) != hipSuccess)
    rc = hip_fail("ptamd_trace_rays", e);
  (void)hipFree(d_rays);
  (void)hipFree(d_out);
  return rc;
}

int ptamd_trace_rays_queue(ptamd_context* ctx, uint32_t scene_id, const float* rays_dev, uint32_t n, int32_t* out_dev, uint32_t config,
                           uint32_t refill_min, void* stream, uint32_t* out_waves_per_cu)
{
  if (!ctx || scene_id >= ctx->scenes.size() || (n && (!rays_dev || !out_dev)) || config > 3u || n >= 0x80000000u) { set_error("ptamd_trace_rays_queue: bad argument"); return PTAMD_ERR_ARG; }
  if (n == 0) return PTAMD_OK;
  PT_HIP(hipSetDevice(ctx->device));
  const DeviceScene& s = ctx->scenes[scene_id];
  if (s.n_nodes4 == 0) { set_error("ptamd_trace_rays_queue: the scene has no wide tree"); return PTAMD_ERR_ARG; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  KParams p;
  std::memset(&p, 0, sizeof p);
  p.nodes4 = s.nodes4; p.n_nodes4 = s.n_nodes4; p.tris_bvh = s.tris_bvh; p.n_bvh_tris = s.n_bvh_tris;
  p.lights = s.lights; p.n_lights = s.n_lights;
  p.refill_min = refill_min < 1u ? 1u : (refill_min > 64u ? 64u : refill_min);
  p.walk_min4 = ctx->walk_min4;
  uint32_t threads, plane, bpc;
  trace_queue_shape(config, &threads, &plane, &bpc);
  const uint32_t waves = threads / 64u;
  p.treelet_nodes = plane < s.n_nodes4 ? plane : s.n_nodes4;
  const uint32_t need = 3u * s.depth4 + 1u;
  const uint32_t share = 160u * 1024u / bpc - 512u;
  const uint32_t treelet_bytes = plane * 128u;
  uint32_t fit = (share - treelet_bytes) / (waves * 512u);
  uint32_t cap = 7u;        // (what the restart kernel's waves get next to their pools: the same stack traffic in every configuration)
  if (const char* ev = tuning_env("PTAMD_TRACE_STACK")) { int v = std::atoi(ev); if (v >= 1) cap = (uint32_t)v; }   // tuning knob
  if (fit > cap) fit = cap;
  p.stack_lds_entries = need < fit ? need : fit;
  p.stack_spill_entries = need - p.stack_lds_entries;
  const size_t lds = (size_t)treelet_bytes + (size_t)p.stack_lds_entries * waves * 512u;
  const uint32_t n_blocks = (uint32_t)ctx->n_cus * bpc;
  const size_t spill = (size_t)n_blocks * waves * p.stack_spill_entries * 512u + 16u;
  if (spill > ctx->trace_spill_bytes) {
    PT_HIP(hipDeviceSynchronize());
    (void)hipFree(ctx->d_trace_spill);
    ctx->d_trace_spill = nullptr; ctx->trace_spill_bytes = 0;
    PT_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace_spill), spill));
    ctx->trace_spill_bytes = spill;
  }
  p.stack_spill = ctx->d_trace_spill;
  uint32_t* head = reinterpret_cast<uint32_t*>(ctx->d_stats + 28);
  PT_HIP(hipMemsetAsync(head, 0, sizeof(uint32_t), st));
  // (the occupancy query costs the host a millisecond: once per configuration and LDS size)
  auto& cache = ctx->trace_queue_cache;   // (per context: the attribute and the answer belong to this context's device)
  const bool cached = cache.config == config && cache.lds == lds;
  int resident = cache.resident;
  hipError_t e = launch_trace_queue(p, config, lds, n_blocks, rays_dev, n, reinterpret_cast<int4*>(out_dev), head, cached ? nullptr : &resident, st);
  if (e != hipSuccess) return hip_fail("ptamd_trace_rays_queue", e);
  cache.config = config; cache.lds = lds; cache.resident = resident;
  if (out_waves_per_cu) *out_waves_per_cu = (uint32_t)(resident < (int)bpc ? resident : (int)bpc) * waves;
  return PTAMD_OK;
}

int ptamd_device_alloc(ptamd_context* ctx, size_t bytes, void** out)
{
  if (!ctx || !out) { set_error("ptamd_device_alloc: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipMalloc(out, bytes ? bytes : 16));
  return PTAMD_OK;
}

int ptamd_device_free(ptamd_context* ctx, void* p)
{
  if (!ctx) { set_error("ptamd_device_free: null context"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipFree(p));
  return PTAMD_OK;
}

int ptamd_device_memset(ptamd_context* ctx, void* p, int value, size_t bytes, void* stream)
{
  if (!ctx || !p) { set_error("ptamd_device_memset: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipMemsetAsync(p, value, bytes, static_cast<hipStream_t>(stream)));
  return PTAMD_OK;
}

int ptamd_device_to_host(ptamd_context* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
  if (!ctx || !dst_host || !src_dev) { set_error("ptamd_device_to_host: null argument"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  PT_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return PTAMD_OK;
}

int ptamd_stream_synchronize(ptamd_context* ctx, void* stream)
{
  if (!ctx) { set_error("ptamd_stream_synchronize: null context"); return PTAMD_ERR_ARG; }
  PT_HIP(hipSetDevice(ctx->device));
  PT_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return PTAMD_OK;
}

} // extern "C"
