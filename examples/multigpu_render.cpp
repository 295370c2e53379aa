// multigpu_render.cpp — C++ host for the N-GPU path (BASELINE.json configs[2]): pixel-row bands over the GPUs of one
// node, one RCCL all-gather of the finished RGBA8 bands per frame.  No Python, no torch: the C-ABI (include/ptamd.h)
// for rendering, HIP for the stream, RCCL (ncclAllGather) for the collective.
//
// The reference is single-GPU (cuda_opengl/src/driver/gpu_info.cpp:26-41); what this host keeps from it is the render
// call itself: every rank issues the same raytrace() work GPUProcessor::render() issues (gpu_processor.cpp:365-386),
// restricted to its rows, with per-pixel seeds taken from GLOBAL pixel coordinates (raytrace.cu:227-229 with the
// full-frame launch geometry), so the assembled frame equals the one-GPU frame bit for bit.
//
// One host thread per GPU inside one process (ncclCommInitAll); each thread owns a ptamd context on its device.
//
//   hipcc -std=c++17 -O2 -Iinclude examples/multigpu_render.cpp -Lcuda-pathtracer_amd -lptamd -lrccl \
//         -Wl,-rpath,$PWD/cuda-pathtracer_amd -o multigpu_render
//   ./multigpu_render assets/indoor.scene 1920 1080 4 4 out.png [--ranks N] [--frames F] [--interleave ROWS] [--check]
//        SCENE WIDTH HEIGHT SPP BOUNCES OUT   --ranks: GPUs to use (default: all)   --frames: timed frames (default 20)
//        --interleave: ranks own interleaved bands of ROWS rows (band j -> rank j % N; default 8, 0 = one contiguous band
//        each): the parts of a picture differ in cost, interleaving spreads them   --check: rank 0 also renders the whole
//        frame alone and compares (exit code 3 on any difference)
#include "ptamd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Options {
  std::string scene, out;
  unsigned width = 1920, height = 1080, spp = 4, bounces = 4;
  int ranks = 0, frames = 20;
  unsigned interleave = 8;
  bool check = false;
};

std::atomic<int> g_failed{0};

// A rank that fails must not leave its peers waiting inside a collective for ever: it raises g_failed and aborts every
// communicator (ncclCommAbort may be called from any thread and makes pending and later operations on it return), and the
// ranks also meet at host-side rendezvous points — before the first collective and after each timed phase — where a
// raised flag sends all of them home.
std::vector<ncclComm_t>* g_comms = nullptr;
std::once_flag g_abort_once;
void fail_all()
{
  g_failed = 1;
  std::call_once(g_abort_once, [] { if (g_comms) for (ncclComm_t c : *g_comms) if (c) (void)ncclCommAbort(c); });
}

class Rendezvous {
 public:
  explicit Rendezvous(int n) : n_(n) {}
  // returns false when any rank has failed (every rank then sees false at the same point)
  bool meet()
  {
    std::unique_lock<std::mutex> lk(m_);
    const unsigned long gen = gen_;
    if (++count_ == n_) { count_ = 0; ++gen_; verdict_ = g_failed.load() == 0; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen_ != gen; });
    return verdict_;
  }
  // a rank that returns early still has to be counted at the points its peers will reach
  void leave() { std::unique_lock<std::mutex> lk(m_); --n_; if (n_ > 0 && count_ == n_) { count_ = 0; ++gen_; verdict_ = false; cv_.notify_all(); } }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  unsigned long gen_ = 0;
  bool verdict_ = true;
};
Rendezvous* g_meet = nullptr;

#define PT_OK(call)                                                                                              \
  do {                                                                                                           \
    int rc_ = (call);                                                                                            \
    if (rc_ != PTAMD_OK) { std::fprintf(stderr, "[rank %d] %s failed (%d): %s\n", rank, #call, rc_, ptamd_get_last_error()); fail_all(); g_meet->leave(); return; } \
  } while (0)
#define HIP_OK(call)                                                                                             \
  do {                                                                                                           \
    hipError_t e_ = (call);                                                                                      \
    if (e_ != hipSuccess) { std::fprintf(stderr, "[rank %d] %s: %s\n", rank, #call, hipGetErrorString(e_)); fail_all(); g_meet->leave(); return; } \
  } while (0)
#define NCCL_OK(call)                                                                                            \
  do {                                                                                                           \
    ncclResult_t r_ = (call);                                                                                    \
    if (r_ != ncclSuccess) { std::fprintf(stderr, "[rank %d] %s: %s\n", rank, #call, ncclGetErrorString(r_)); fail_all(); g_meet->leave(); return; } \
  } while (0)

// Contiguous row bands, remainders to the last ranks (cuda_pathtracer_amd/tiles.py: row_bands).
void band_of(unsigned height, int world, int rank, unsigned* y0, unsigned* y1)
{
  const unsigned base = height / (unsigned)world, extra = height % (unsigned)world;
  unsigned start = 0;
  for (int r = 0; r <= rank; ++r) {
    const unsigned n = base + ((unsigned)r >= (unsigned)world - extra ? 1u : 0u);
    *y0 = start;
    *y1 = start + n;
    start += n;
  }
}

struct Shared {
  const Options* opt;
  const ptamd_scene_desc* desc;
  const ptamd_camera* cam;
  const std::vector<float>* cube;
  uint32_t cube_size;
  std::vector<ncclComm_t> comms;
  std::vector<unsigned char> frame;   // rank 0's gathered frame (host copy)
  double ms_per_frame = 0.0;
  bool check_ok = true;
};

void rank_main(int rank, int world, Shared* sh)
{
  const Options& o = *sh->opt;
  HIP_OK(hipSetDevice(rank));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  ptamd_context* ctx = nullptr;
  PT_OK(ptamd_create(rank, &ctx));
  PT_OK(ptamd_setup_function_tables(ctx));
  uint32_t scene_id = 0, cubemap_id = 0;
  PT_OK(ptamd_upload_scene(ctx, sh->desc, &scene_id));
  PT_OK(ptamd_upload_cubemap(ctx, sh->cube->data(), sh->cube_size, &cubemap_id));

  const bool ilv = o.interleave != 0 && world > 1;
  unsigned y0 = 0, y1 = 0;
  band_of(o.height, world, rank, &y0, &y1);
  if (ilv) { y0 = 0; y1 = ptamd_interleaved_rows(o.height, (uint32_t)world, (uint32_t)rank, o.interleave); }   // rows of this rank's buffers
  unsigned max_rows = 0;
  for (int r = 0; r < world; ++r) {
    unsigned a, b;
    band_of(o.height, world, r, &a, &b);
    const unsigned n = ilv ? ptamd_interleaved_rows(o.height, (uint32_t)world, (uint32_t)r, o.interleave) : b - a;
    if (n > max_rows) max_rows = n;
  }
  const size_t band_bytes = (size_t)max_rows * o.width * 4;   // bands are padded to the tallest one: one fixed-size collective
  void *surface = nullptr, *tfb = nullptr, *gathered = nullptr;
  // (from here on y1 - y0 = the rows this rank's buffers hold)
  PT_OK(ptamd_device_alloc(ctx, band_bytes, &surface));
  PT_OK(ptamd_device_alloc(ctx, (size_t)(y1 - y0) * o.width * 12, &tfb));
  PT_OK(ptamd_device_alloc(ctx, band_bytes * (size_t)world, &gathered));
  PT_OK(ptamd_device_memset(ctx, surface, 0, band_bytes, stream));

  ptamd_launch l;
  std::memset(&l, 0, sizeof l);
  l.surface_rgba8 = surface; l.temporal_framebuffer = static_cast<float*>(tfb); l.stream = stream;
  l.camera = *sh->cam; l.scene_id = scene_id; l.cubemap_id = cubemap_id;
  l.width = o.width; l.height = o.height; l.row_begin = y0; l.row_end = y1;
  l.frame_nb = 1; l.bounces = o.bounces; l.moved = 0; l.post_id = 0; l.kernel = PTAMD_KERNEL_AUTO;
  l.band_local_buffers = 1;
  if (ilv) { l.row_begin = 0; l.row_end = o.height; l.interleave_ranks = (uint32_t)world; l.interleave_rank = (uint32_t)rank; l.interleave_rows = o.interleave; }
  l.frame_count = o.spp;          // the spp static frames of one picture as ONE launch (== spp consecutive raytrace() calls)
  l.reset_accumulation = 1;       // every frame starts a new accumulation: no clear of the accumulator in front of the launch

  auto frame = [&]() -> bool {
    if (ptamd_raytrace_ex(ctx, &l) != PTAMD_OK) return false;
    return ncclAllGather(surface, gathered, band_bytes, ncclUint8, sh->comms[rank], stream) == ncclSuccess;
  };
  // every rank has its context, scene and buffers: only now may anyone enter a collective
  if (!g_meet->meet()) return;
  for (int i = 0; i < 3; ++i)
    if (!frame()) { std::fprintf(stderr, "[rank %d] warm-up frame failed: %s\n", rank, ptamd_get_last_error()); fail_all(); g_meet->leave(); return; }
  HIP_OK(hipStreamSynchronize(stream));
  if (!g_meet->meet()) return;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < o.frames; ++i)
    if (!frame()) { std::fprintf(stderr, "[rank %d] frame failed: %s\n", rank, ptamd_get_last_error()); fail_all(); g_meet->leave(); return; }
  HIP_OK(hipStreamSynchronize(stream));
  if (!g_meet->meet()) return;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / o.frames;

  if (rank == 0) {
    sh->ms_per_frame = ms;
    // drop the padding rows while copying the gathered bands to the host
    sh->frame.assign((size_t)o.width * o.height * 4, 0);
    for (int r = 0; r < world; ++r) {
      if (!ilv) {
        unsigned a, b;
        band_of(o.height, world, r, &a, &b);
        PT_OK(ptamd_device_to_host(ctx, sh->frame.data() + (size_t)a * o.width * 4,
                                   static_cast<char*>(gathered) + (size_t)r * band_bytes, (size_t)(b - a) * o.width * 4, stream));
        continue;
      }
      // rank r's buffer holds its bands r, r + world, ... one after the other
      size_t local = 0;
      for (unsigned a = (unsigned)r * o.interleave; a < o.height; a += (unsigned)world * o.interleave) {
        const unsigned b = a + o.interleave < o.height ? a + o.interleave : o.height;
        PT_OK(ptamd_device_to_host(ctx, sh->frame.data() + (size_t)a * o.width * 4,
                                   static_cast<char*>(gathered) + (size_t)r * band_bytes + local * o.width * 4,
                                   (size_t)(b - a) * o.width * 4, stream));
        local += b - a;
      }
    }
    if (o.check) {
      void *full_surface = nullptr, *full_tfb = nullptr;
      PT_OK(ptamd_device_alloc(ctx, (size_t)o.width * o.height * 4, &full_surface));
      PT_OK(ptamd_device_alloc(ctx, (size_t)o.width * o.height * 12, &full_tfb));
      PT_OK(ptamd_device_memset(ctx, full_tfb, 0, (size_t)o.width * o.height * 12, stream));
      ptamd_launch f = l;
      f.surface_rgba8 = full_surface; f.temporal_framebuffer = static_cast<float*>(full_tfb);
      f.row_begin = 0; f.row_end = o.height; f.band_local_buffers = 0;
      f.interleave_ranks = f.interleave_rank = f.interleave_rows = 0;
      PT_OK(ptamd_raytrace_ex(ctx, &f));
      std::vector<unsigned char> ref((size_t)o.width * o.height * 4);
      PT_OK(ptamd_device_to_host(ctx, ref.data(), full_surface, ref.size(), stream));
      sh->check_ok = ref == sh->frame;
      ptamd_device_free(ctx, full_surface);
      ptamd_device_free(ctx, full_tfb);
    }
  }
  ptamd_device_free(ctx, surface);
  ptamd_device_free(ctx, tfb);
  ptamd_device_free(ctx, gathered);
  ptamd_destroy(ctx);
  (void)hipStreamDestroy(stream);
}

} // namespace

int main(int argc, char** argv)
{
  Options o;
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s SCENE WIDTH HEIGHT SPP BOUNCES OUT.(png|ppm) [--ranks N] [--frames F] [--check]\n", argv[0]);
    return 2;
  }
  o.scene = argv[1];
  o.width = (unsigned)std::atoi(argv[2]); o.height = (unsigned)std::atoi(argv[3]);
  o.spp = (unsigned)std::atoi(argv[4]); o.bounces = (unsigned)std::atoi(argv[5]);
  o.out = argv[6];
  for (int i = 7; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) o.ranks = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) o.frames = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--interleave") && i + 1 < argc) o.interleave = (unsigned)std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--check")) o.check = true;
    else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  if (o.width == 0 || o.height == 0 || o.spp == 0 || o.bounces == 0 || o.frames < 1 || o.interleave % 8u != 0) {
    std::fprintf(stderr, "bad frame parameters (--interleave must be a multiple of 8)\n");
    return 2;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { std::fprintf(stderr, "no HIP device: this host has no CPU fallback\n"); return 1; }
  const int world = o.ranks > 0 ? o.ranks : n_dev;
  if (world > n_dev) { std::fprintf(stderr, "--ranks %d but only %d GPU(s) are visible\n", world, n_dev); return 1; }
  if ((unsigned)world > o.height) { std::fprintf(stderr, "more ranks than rows\n"); return 2; }

  ptamd_host_scene* hs = nullptr;
  if (ptamd_host_scene_load(o.scene.c_str(), 0, &hs) != PTAMD_OK) { std::fprintf(stderr, "scene: %s\n", ptamd_get_last_error()); return 1; }
  ptamd_scene_desc desc;
  ptamd_camera cam;
  if (ptamd_host_scene_desc(hs, &desc) != PTAMD_OK || ptamd_host_scene_camera(hs, &cam) != PTAMD_OK) return 1;
  std::vector<float> cube(24);
  uint32_t cube_size = 1;
  ptamd_cubemap_from_color(0x131b23, cube.data());   // fallback of gpu_processor.cpp:128-132
  {
    const std::string name = ptamd_host_scene_cubemap(hs);
    const size_t slash = o.scene.find_last_of('/');
    const std::string folder = slash == std::string::npos ? "." : o.scene.substr(0, slash);
    int32_t cw = 0, ch = 0, cc = 0;
    float* cross = nullptr;
    if (!name.empty() && ptamd_image_loadf((folder + "/" + name).c_str(), &cw, &ch, &cc, &cross) == PTAMD_OK) {
      std::vector<float> faces((size_t)6 * (cw / 4) * (cw / 4) * 4 + 24);
      uint32_t size = 0;
      if (ptamd_cubemap_from_cross(cross, (uint32_t)cw, (uint32_t)ch, (uint32_t)cc, faces.data(), &size) == PTAMD_OK) { cube.swap(faces); cube_size = size; }
      ptamd_image_free(cross);
    }
  }

  Shared sh;
  sh.opt = &o; sh.desc = &desc; sh.cam = &cam; sh.cube = &cube; sh.cube_size = cube_size;
  sh.comms.resize((size_t)world);
  std::vector<int> devs((size_t)world);
  for (int r = 0; r < world; ++r) devs[(size_t)r] = r;
  if (ncclCommInitAll(sh.comms.data(), world, devs.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
  Rendezvous meet(world);
  g_meet = &meet;
  g_comms = &sh.comms;
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r, world, &sh);
  for (auto& t : threads) t.join();
  if (!g_failed) for (auto c : sh.comms) ncclCommDestroy(c);   // (aborted communicators are gone already)
  ptamd_host_scene_free(hs);
  if (g_failed) return 1;

  std::vector<unsigned char> rgb((size_t)o.width * o.height * 3);   // the surface's alpha is 0 (raytrace.cu:232): write RGB
  for (size_t i = 0; i < (size_t)o.width * o.height; ++i) std::memcpy(&rgb[i * 3], &sh.frame[i * 4], 3);
  const size_t len = o.out.size();
  if (len > 4 && o.out.compare(len - 4, 4, ".png") == 0) {
    if (ptamd_image_save_png(o.out.c_str(), rgb.data(), (int32_t)o.width, (int32_t)o.height, 3) != PTAMD_OK) return 1;
  } else {
    FILE* f = std::fopen(o.out.c_str(), "wb");
    if (!f) return 1;
    std::fprintf(f, "P6 %u %u 255\n", o.width, o.height);
    std::fwrite(rgb.data(), 1, rgb.size(), f);
    std::fclose(f);
  }
  const double msamples = (double)o.width * o.height * o.spp / (sh.ms_per_frame * 1e-3) / 1e6;
  std::printf("{\"host\": \"c++\", \"n_gpus\": %d, \"frame\": \"%ux%u %u spp %u bounces\", \"ms_per_frame\": %.4f, \"msamples_per_s\": %.1f, "
              "\"bands\": \"%s\", \"gather\": \"ncclAllGather of RGBA8 row bands\"%s}\n",
              world, o.width, o.height, o.spp, o.bounces, sh.ms_per_frame, msamples,
              (o.interleave != 0 && world > 1) ? "interleaved" : "contiguous",
              o.check ? (sh.check_ok ? ", \"check\": \"equals the 1-GPU frame\"" : ", \"check\": \"DIFFERS from the 1-GPU frame\"") : "");
  return o.check && !sh.check_ok ? 3 : 0;
}
