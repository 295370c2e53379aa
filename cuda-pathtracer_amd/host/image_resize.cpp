// image_resize.cpp — float image rescale used when a material's diffuse and specular maps differ in size.
//
// The reference rescales the smaller map with stbir_resize_float(in, w, h, 0, out, W, H, 0, channels)
// (material_loader.cpp:350-375; stb_image_resize 0.95 defaults: Catmull-Rom when an axis grows, Mitchell
// when it does not, clamped edges, linear colour, no alpha handling).  The texels that come out of it are
// inputs of the render, so its float arithmetic is restated here operation for operation:
//   * per axis, scale = (float)out / in; an axis with scale > 1 GATHERS (each output sample sums its <= 4
//     clamped input neighbours in ascending order, weights normalised per output sample, zero weights at
//     either end dropped); any other axis SCATTERS (each input sample, margin included, adds into the
//     output samples it covers in ascending input order, weights = kernel * scale normalised per OUTPUT
//     sample afterwards)                                             stb_image_resize.h:1006-1230
//   * the horizontal pass runs first on every (clamped) input row, then the vertical pass combines rows;
//     every accumulation is `acc += value * weight` in float, starting from 0   stb_image_resize.h:1439-2198
// tests/test_ref_thirdparty.py compares the result with the real stbir_resize_float (oracle/_ref) bit for
// bit over growing / shrinking / mixed / equal-size axes, 1 and 3 channels.
#include "ptamd_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace ptamd {
namespace {

// Catmull-Rom and Mitchell-Netravali (B = C = 1/3) in the exact float expression order of
// stb_image_resize.h:808-834 (integer literals promote to float, nothing is double)
float catmull_rom(float x)
{
  x = (float)std::fabs(x);
  if (x < 1.0f) return 1 - x * x * (2.5f - 1.5f * x);
  if (x < 2.0f) return 2 - x * (4 + x * (0.5f * x - 2.5f));
  return 0.0f;
}

float mitchell(float x)
{
  x = (float)std::fabs(x);
  if (x < 1.0f) return (16 + x * x * (21 * x - 36)) / 18;
  if (x < 2.0f) return (32 + x * (-60 + x * (36 - 7 * x))) / 18;
  return 0.0f;
}

constexpr float kSupport = 2.0f;   // both kernels
constexpr int kTaps = 4;           // (int)ceil(support * 2): weights stored per contributor

struct Axis {
  bool gather = false;             // scale > 1
  float scale = 1.0f;
  int in_size = 0, out_size = 0;
  int margin = 0;                  // input samples consulted beyond each edge
  int n_contrib = 0;               // gather: one per output sample; scatter: one per input sample incl. margins
  std::vector<int> n0, n1;
  std::vector<float> w;            // kTaps per contributor, flat (a 5th weight spills into the next group, as in stb)

  float& weight(int n, int c) { return w[(size_t)kTaps * n + c]; }
};

void build_gather(Axis& a)
{
  const float radius = kSupport * a.scale;
  for (int n = 0; n < a.n_contrib; ++n) {
    const float centre = (float)n + 0.5f;
    const float lo = (centre - radius + 0.0f) / a.scale, hi = (centre + radius + 0.0f) / a.scale;
    const float in_centre = (centre + 0.0f) / a.scale;
    int first = (int)std::floor(lo + 0.5), last = (int)std::floor(hi - 0.5);
    a.n0[n] = first; a.n1[n] = last;
    float total = 0;
    float* g = &a.weight(n, 0);
    for (int i = 0; i <= last - first; ++i) {
      const float in_pixel_centre = (float)(i + first) + 0.5f;
      g[i] = catmull_rom(in_centre - in_pixel_centre);
      if (i == 0 && !g[i]) { a.n0[n] = ++first; --i; continue; }     // leading zero weight: start one sample later
      total += g[i];
    }
    const float norm = 1 / total;
    for (int i = 0; i <= last - first; ++i) g[i] *= norm;
    for (int i = last - first; i >= 0; --i) {                          // trailing zero weights
      if (g[i]) break;
      a.n1[n] = a.n0[n] + i - 1;
    }
  }
}

void build_scatter(Axis& a)
{
  const float radius = kSupport / a.scale;
  for (int n = 0; n < a.n_contrib; ++n) {
    const float centre = (float)(n - a.margin) + 0.5f;
    const float lo = (centre - radius) * a.scale - 0.0f, hi = (centre + radius) * a.scale - 0.0f;
    const float out_centre = centre * a.scale - 0.0f;
    const int first = (int)std::floor(lo + 0.5), last = (int)std::floor(hi - 0.5);
    a.n0[n] = first; a.n1[n] = last;
    float* g = &a.weight(n, 0);
    for (int i = 0; i <= last - first; ++i) {
      const float x = ((float)(i + first) + 0.5f) - out_centre;
      g[i] = mitchell(x) * a.scale;
    }
    for (int i = last - first; i >= 0; --i) {
      if (g[i]) break;
      a.n1[n] = a.n0[n] + i - 1;
    }
  }
  // every OUTPUT sample's incoming weights are made to sum to one (stb_image_resize.h:1115-1150)
  for (int i = 0; i < a.out_size; ++i) {
    float total = 0;
    for (int j = 0; j < a.n_contrib; ++j) {
      if (i >= a.n0[j] && i <= a.n1[j]) total += a.weight(j, i - a.n0[j]);
      else if (i < a.n0[j]) break;
    }
    const float norm = 1 / total;
    for (int j = 0; j < a.n_contrib; ++j) {
      if (i >= a.n0[j] && i <= a.n1[j]) a.weight(j, i - a.n0[j]) *= norm;
      else if (i < a.n0[j]) break;
    }
  }
  // then leading zero weights and output positions < 0 are dropped, the rest shifted down (:1152-1186)
  const size_t limit = a.w.size();
  for (int j = 0; j < a.n_contrib; ++j) {
    int skip = 0;
    while ((size_t)kTaps * j + skip < limit && a.weight(j, skip) == 0) ++skip;
    a.n0[j] += skip;
    while (a.n0[j] < 0) { ++a.n0[j]; ++skip; }
    const int range = a.n1[j] - a.n0[j] + 1;
    const int count = range < kTaps ? range : kTaps;
    for (int i = 0; i < count; ++i) {
      if (i + skip >= kTaps) break;
      a.weight(j, i) = a.weight(j, i + skip);
    }
  }
  for (int j = 0; j < a.n_contrib; ++j)
    if (a.n1[j] > a.out_size - 1) a.n1[j] = a.out_size - 1;
}

Axis build_axis(int in_size, int out_size)
{
  Axis a;
  a.in_size = in_size; a.out_size = out_size;
  a.scale = ((float)out_size / in_size) / (1.0f - 0.0f);
  a.gather = a.scale > 1;
  const int pixel_width = a.gather ? (int)std::ceil(kSupport * 2) : (int)std::ceil(kSupport * 2 / a.scale);
  a.margin = pixel_width / 2;
  a.n_contrib = a.gather ? out_size : in_size + a.margin * 2;
  a.n0.assign((size_t)a.n_contrib, 0);
  a.n1.assign((size_t)a.n_contrib, 0);
  a.w.assign((size_t)a.n_contrib * kTaps + 8, 0.0f);
  if (a.gather) build_gather(a); else build_scatter(a);
  return a;
}

inline int clampi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// one input row -> one row of out_w samples
void horizontal(Axis& h, const float* row, int channels, float* out)
{
  std::memset(out, 0, (size_t)h.out_size * channels * sizeof(float));
  if (h.gather) {
    for (int x = 0; x < h.out_size; ++x) {
      int c_idx = 0;
      for (int k = h.n0[x]; k <= h.n1[x]; ++k) {
        const float wgt = h.weight(x, c_idx++);
        const float* src = row + (size_t)clampi(k, h.in_size) * channels;
        for (int c = 0; c < channels; ++c) out[(size_t)x * channels + c] += src[c] * wgt;
      }
    }
  } else {
    for (int x = 0; x < h.n_contrib; ++x) {
      const float* src = row + (size_t)clampi(x - h.margin, h.in_size) * channels;
      for (int k = h.n0[x]; k <= h.n1[x]; ++k) {
        const float wgt = h.weight(x, k - h.n0[x]);
        for (int c = 0; c < channels; ++c) out[(size_t)k * channels + c] += src[c] * wgt;
      }
    }
  }
}

} // namespace

bool resize_float(const float* in, int in_w, int in_h, float* out, int out_w, int out_h, int channels)
{
  if (!in || !out || in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0 || channels <= 0 || channels > 64) return false;
  Axis h = build_axis(in_w, out_w), v = build_axis(in_h, out_h);
  const size_t row_len = (size_t)out_w * channels;
  std::vector<float> rows((size_t)in_h * row_len);      // horizontal pass of every input row
  for (int y = 0; y < in_h; ++y) horizontal(h, in + (size_t)y * in_w * channels, channels, rows.data() + (size_t)y * row_len);
  std::memset(out, 0, (size_t)out_h * row_len * sizeof(float));
  if (v.gather) {
    for (int y = 0; y < out_h; ++y) {
      float* dst = out + (size_t)y * row_len;
      int c_idx = 0;
      for (int k = v.n0[y]; k <= v.n1[y]; ++k) {
        const float wgt = v.weight(y, c_idx++);
        const float* src = rows.data() + (size_t)clampi(k, in_h) * row_len;
        for (size_t i = 0; i < row_len; ++i) dst[i] += src[i] * wgt;
      }
    }
  } else {
    const float radius = kSupport / v.scale;
    for (int y = -v.margin; y < in_h + v.margin; ++y) {
      const float centre = (float)y + 0.5f;
      const int first = (int)std::floor((centre - radius) * v.scale - 0.0f + 0.5);
      const int last = (int)std::floor((centre + radius) * v.scale - 0.0f - 0.5);
      if (last < 0 || first >= out_h) continue;         // this input row reaches no output row
      const int j = y + v.margin;
      const float* src = rows.data() + (size_t)clampi(y, in_h) * row_len;
      for (int k = v.n0[j]; k <= v.n1[j]; ++k) {
        const float wgt = v.weight(j, k - v.n0[j]);
        float* dst = out + (size_t)k * row_len;
        for (size_t i = 0; i < row_len; ++i) dst[i] += src[i] * wgt;
      }
    }
  }
  return true;
}

} // namespace ptamd

extern "C" int ptamd_image_resize_float(const float* in, int32_t in_w, int32_t in_h, float* out, int32_t out_w, int32_t out_h,
                                        int32_t channels)
{
  if (!ptamd::resize_float(in, in_w, in_h, out, out_w, out_h, channels)) {
    ptamd::set_error("ptamd_image_resize_float: bad argument");
    return PTAMD_ERR_ARG;
  }
  return PTAMD_OK;
}
