// image_decode.cpp — native image decoding for the scene loader: the role stb_image's stbi_loadf plays
// for the reference (material_loader.cpp:97, gpu_processor.cpp:99).
//
// Every texture and cubemap the reference ships is a JPEG (21 progressive 4:4:4 files, one baseline 4:2:0,
// one grayscale), so this file holds a JPEG decoder written from ITU-T T.81 (baseline, extended and
// progressive Huffman, 8-bit, 1/3/4 components, restart intervals) plus stbi_loadf's 8-bit -> float rule.
// T.81 leaves three things to the implementation; for those the reference's decoder (stb_image 2.16) defines
// the pixels, and this file restates its arithmetic so that decoded textures are bit-identical to the
// reference's:
//   * the inverse DCT: 12-bit fixed-point "islow" factorisation, column pass keeps 2 extra bits
//     ((x + 512) >> 10), row pass rounds with 65536 + (128 << 17) and shifts by 17 (stb_image.h:2129-2213);
//   * chroma upsampling: the (3 near + far) triangle filters per direction (stb_image.h:3124-3187), rows
//     paired as in stb_image.h:3548-3585;
//   * YCbCr -> RGB in 20-bit fixed point with the green cb term truncated to 16 bits (stb_image.h:3317-3343).
// tests/test_ref_thirdparty.py checks every shipped JPEG (and synthetic ones covering the other sampling
// layouts) against the real stb_image compiled from /root/reference into oracle/_ref; tests/golden holds the
// checksums for where /root/reference is absent.
//
// PNG is decoded by image_png.cpp, Radiance .hdr below (float path only).  The other formats stb_image reads (BMP,
// TGA, GIF, PSD, PIC, PNM) are reported as undecodable; a host that needs them passes its own decoder through ptamd_host_scene_load_ex.
#include "ptamd_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ptamd {
namespace {

// natural (row-major) position of the k-th coefficient of the zig-zag sequence (T.81 Figure A.6)
const uint8_t kNaturalOrder[64] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Fail { std::string why; };
[[noreturn]] void fail(const char* why) { throw Fail{ why }; }

// ---- Huffman tables (T.81 Annex C) with a 9-bit look-ahead
struct HuffTable {
  bool defined = false;
  uint8_t counts[17] = {};     // codes of each length 1..16
  uint8_t symbols[256] = {};
  uint16_t look[512] = {};     // (length << 8) | symbol for codes of <= 9 bits, 0 otherwise

  void build(const uint8_t bits[16], const uint8_t* vals, int n) {
    std::memset(look, 0, sizeof look);
    counts[0] = 0;
    for (int i = 0; i < 16; ++i) counts[i + 1] = bits[i];
    std::memcpy(symbols, vals, (size_t)n);
    uint32_t code = 0; int idx = 0;
    for (int len = 1; len <= 16; ++len) {
      for (int k = 0; k < counts[len]; ++k, ++idx, ++code) {
        if (code >= (1u << len)) fail("bad code lengths");
        if (len <= 9) {
          uint32_t first = code << (9 - len);
          for (uint32_t f = 0; f < (1u << (9 - len)); ++f) look[first + f] = (uint16_t)((len << 8) | symbols[idx]);
        }
      }
      code <<= 1;
    }
    defined = true;
  }
};

// ---- entropy-coded segment reader: MSB first, 0xFF00 unstuffing, stops feeding at a marker (zeros after it)
struct BitReader {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  uint32_t acc = 0;
  int n = 0;
  int marker = 0;    // marker code met inside the segment (0: none yet)

  int byte() { return p < end ? *p++ : 0; }
  void restart() { acc = 0; n = 0; marker = 0; }
  void fill() {
    while (n <= 24) {
      int b = 0;
      if (!marker) {
        b = byte();
        if (b == 0xFF) {
          int c = byte();
          while (c == 0xFF) c = byte();
          if (c != 0) { marker = c; b = 0; }
        }
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  uint32_t get(int k) {
    if (k == 0) return 0;
    if (n < k) fill();
    uint32_t v = acc >> (32 - k);
    acc <<= k; n -= k;
    return v;
  }
  int bit() { return (int)get(1); }
  int extend(int s) {                 // RECEIVE + EXTEND (T.81 F.2.2.1)
    if (s == 0) return 0;
    int v = (int)get(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
  }
  int symbol(const HuffTable& h) {
    if (n < 16) fill();
    uint32_t e = h.look[acc >> 23];
    if (e) { int len = (int)(e >> 8); acc <<= len; n -= len; return (int)(e & 0xFF); }
    uint32_t code = 0, first = 0; int idx = 0;
    uint32_t bits16 = acc >> 16;
    for (int len = 1; len <= 16; ++len) {
      code = (code << 1) | ((bits16 >> (16 - len)) & 1u);
      uint32_t cnt = h.counts[len];
      if (code - first < cnt) { acc <<= len; n -= len; return h.symbols[idx + (int)(code - first)]; }
      idx += (int)cnt;
      first = (first + cnt) << 1;
    }
    fail("bad huffman code");
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, dc_pred = 0;
  int x = 0, y = 0;            // samples that carry picture (non-interleaved scans cover only these)
  int w2 = 0, h2 = 0;          // plane size padded to whole MCUs
  std::vector<uint8_t> plane;
  std::vector<int16_t> coeff;  // progressive: 64 per block, blocks row-major, w2 / 8 per row
};

// ---- inverse DCT.  12-bit constants exactly as stb forms them: (int)(c_float * 4096 + 0.5)  (stb_image.h:2129)
constexpr int fix12(float c) { return (int)((double)(c * 4096) + 0.5); }

struct Idct1D { int e0, e1, e2, e3, o0, o1, o2, o3; };   // x[k] = e[k] + o[k], x[7 - k] = e[k] - o[k]

inline Idct1D idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
  // even part
  int z = (s2 + s6) * fix12(0.5411961f);
  int a2 = z + s6 * fix12(-1.847759065f);
  int a3 = z + s2 * fix12(0.765366865f);
  int a0 = (s0 + s4) * 4096, a1 = (s0 - s4) * 4096;
  Idct1D r;
  r.e0 = a0 + a3; r.e3 = a0 - a3; r.e1 = a1 + a2; r.e2 = a1 - a2;
  // odd part
  int z1 = s7 + s1, z2 = s5 + s3, z3 = s7 + s3, z4 = s5 + s1;
  int z5 = (z3 + z4) * fix12(1.175875602f);
  int b0 = s7 * fix12(0.298631336f), b1 = s5 * fix12(2.053119869f);
  int b2 = s3 * fix12(3.072711026f), b3 = s1 * fix12(1.501321110f);
  z1 = z5 + z1 * fix12(-0.899976223f);
  z2 = z5 + z2 * fix12(-2.562915447f);
  z3 = z3 * fix12(-1.961570560f);
  z4 = z4 * fix12(-0.390180644f);
  r.o0 = b3 + z1 + z4; r.o1 = b2 + z2 + z3; r.o2 = b1 + z2 + z4; r.o3 = b0 + z1 + z3;
  return r;
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void idct_block(const int16_t* c, uint8_t* out, int stride) {
  int mid[64];
  for (int i = 0; i < 8; ++i) {
    Idct1D r = idct_1d(c[i], c[8 + i], c[16 + i], c[24 + i], c[32 + i], c[40 + i], c[48 + i], c[56 + i]);
    const int rnd = 512;     // keep 2 extra bits: >> 10 of a 12-bit scaled value
    mid[i] = (r.e0 + rnd + r.o0) >> 10;       mid[56 + i] = (r.e0 + rnd - r.o0) >> 10;
    mid[8 + i] = (r.e1 + rnd + r.o1) >> 10;   mid[48 + i] = (r.e1 + rnd - r.o1) >> 10;
    mid[16 + i] = (r.e2 + rnd + r.o2) >> 10;  mid[40 + i] = (r.e2 + rnd - r.o2) >> 10;
    mid[24 + i] = (r.e3 + rnd + r.o3) >> 10;  mid[32 + i] = (r.e3 + rnd - r.o3) >> 10;
  }
  for (int i = 0; i < 8; ++i) {
    const int* m = mid + 8 * i;
    uint8_t* o = out + (size_t)stride * i;
    Idct1D r = idct_1d(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    const int rnd = 65536 + (128 << 17);      // 12 + 2 + 3 bits to drop, and the +128 level shift
    o[0] = clamp8((r.e0 + rnd + r.o0) >> 17); o[7] = clamp8((r.e0 + rnd - r.o0) >> 17);
    o[1] = clamp8((r.e1 + rnd + r.o1) >> 17); o[6] = clamp8((r.e1 + rnd - r.o1) >> 17);
    o[2] = clamp8((r.e2 + rnd + r.o2) >> 17); o[5] = clamp8((r.e2 + rnd - r.o2) >> 17);
    o[3] = clamp8((r.e3 + rnd + r.o3) >> 17); o[4] = clamp8((r.e3 + rnd - r.o3) >> 17);
  }
}

// ---- the decoder
struct Jpeg {
  const uint8_t* p; const uint8_t* end;
  HuffTable dc[4], ac[4];
  uint16_t quant[4][64] = {};
  Component comp[4];
  int n_comp = 0, width = 0, height = 0;
  int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
  bool progressive = false, have_frame = false;
  bool jfif = false; int adobe_transform = -1; int rgb_ids = 0;
  int restart_interval = 0;
  // scan state
  int scan_n = 0, order[4] = {}, ss = 0, se = 63, ah = 0, al = 0;
  int eob_run = 0, todo = 0;
  BitReader br;
  int pending_marker = 0;

  int u8() { return p < end ? *p++ : 0; }
  int u16() { int a = u8(); return (a << 8) | u8(); }
  void skip(int n) { if (n < 0) fail("bad segment length"); p = (end - p < n) ? end : p + n; }

  int next_marker() {
    if (pending_marker) { int m = pending_marker; pending_marker = 0; return m; }
    int x = u8();
    if (x != 0xFF) return 0;
    while (x == 0xFF) x = u8();
    return x;
  }

  void read_tables_or_app(int m) {
    switch (m) {
    case 0: fail("expected marker");
    case 0xDD: if (u16() != 4) fail("bad DRI len"); restart_interval = u16(); return;
    case 0xDB: {
      int len = u16() - 2;
      while (len > 0) {
        int q = u8(), wide = q >> 4, t = q & 15;
        if (wide > 1) fail("bad DQT type");
        if (t > 3) fail("bad DQT table");
        for (int i = 0; i < 64; ++i) quant[t][kNaturalOrder[i]] = (uint16_t)(wide ? u16() : u8());
        len -= wide ? 129 : 65;
      }
      if (len != 0) fail("bad DQT len");
      return;
    }
    case 0xC4: {
      int len = u16() - 2;
      while (len > 0) {
        int q = u8(), cls = q >> 4, th = q & 15;
        if (cls > 1 || th > 3) fail("bad DHT header");
        uint8_t bits[16], vals[256]; int n = 0;
        for (int i = 0; i < 16; ++i) { bits[i] = (uint8_t)u8(); n += bits[i]; }
        if (n > 256) fail("bad DHT header");
        for (int i = 0; i < n; ++i) vals[i] = (uint8_t)u8();
        (cls ? ac[th] : dc[th]).build(bits, vals, n);
        len -= 17 + n;
      }
      if (len != 0) fail("bad DHT len");
      return;
    }
    default: break;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
      int len = u16();
      if (len < 2) fail("bad APP/COM len");
      len -= 2;
      if (m == 0xE0 && len >= 5) {
        static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
        bool ok = true;
        for (int i = 0; i < 5; ++i) if (u8() != (uint8_t)tag[i]) ok = false;
        len -= 5;
        if (ok) jfif = true;
      } else if (m == 0xEE && len >= 12) {
        static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
        bool ok = true;
        for (int i = 0; i < 6; ++i) if (u8() != (uint8_t)tag[i]) ok = false;
        len -= 6;
        if (ok) { u8(); u16(); u16(); adobe_transform = u8(); len -= 6; }
      }
      skip(len);
      return;
    }
    fail("unknown marker");
  }

  void read_frame_header(int m) {
    progressive = (m == 0xC2);
    int len = u16(); if (len < 11) fail("bad SOF len");
    if (u8() != 8) fail("only 8-bit");
    height = u16(); if (height == 0) fail("no header height");
    width = u16(); if (width == 0) fail("0 width");
    n_comp = u8();
    if (n_comp != 1 && n_comp != 3 && n_comp != 4) fail("bad component count");
    if (len != 8 + 3 * n_comp) fail("bad SOF len");
    rgb_ids = 0;
    for (int i = 0; i < n_comp; ++i) {
      Component& c = comp[i];
      c.id = u8();
      if (n_comp == 3 && c.id == "RGB"[i]) ++rgb_ids;
      int q = u8();
      c.h = q >> 4; c.v = q & 15; c.tq = u8();
      if (c.h < 1 || c.h > 4) fail("bad H");
      if (c.v < 1 || c.v > 4) fail("bad V");
      if (c.tq > 3) fail("bad TQ");
    }
    if ((uint64_t)width * (uint64_t)height * (uint64_t)n_comp > (1ull << 30)) fail("too large");
    h_max = v_max = 1;
    for (int i = 0; i < n_comp; ++i) { if (comp[i].h > h_max) h_max = comp[i].h; if (comp[i].v > v_max) v_max = comp[i].v; }
    mcu_x = (width + 8 * h_max - 1) / (8 * h_max);
    mcu_y = (height + 8 * v_max - 1) / (8 * v_max);
    for (int i = 0; i < n_comp; ++i) {
      Component& c = comp[i];
      c.x = (width * c.h + h_max - 1) / h_max;
      c.y = (height * c.v + v_max - 1) / v_max;
      c.w2 = mcu_x * c.h * 8; c.h2 = mcu_y * c.v * 8;
      c.plane.assign((size_t)c.w2 * c.h2, 0);
      if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
    }
    have_frame = true;
  }

  void read_scan_header() {
    int len = u16();
    scan_n = u8();
    if (scan_n < 1 || scan_n > 4 || scan_n > n_comp) fail("bad SOS component count");
    if (len != 6 + 2 * scan_n) fail("bad SOS len");
    for (int i = 0; i < scan_n; ++i) {
      int id = u8(), q = u8(), which = 0;
      while (which < n_comp && comp[which].id != id) ++which;
      if (which == n_comp) fail("bad SOS component");
      comp[which].td = q >> 4; comp[which].ta = q & 15;
      if (comp[which].td > 3 || comp[which].ta > 3) fail("bad huffman table index");
      order[i] = which;
    }
    ss = u8(); se = u8();
    int a = u8(); ah = a >> 4; al = a & 15;
    if (progressive) {
      if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) fail("bad SOS");
    } else {
      if (ss != 0 || ah != 0 || al != 0) fail("bad SOS");
      se = 63;
    }
  }

  void reset_entropy() {
    br.restart();
    for (Component& c : comp) c.dc_pred = 0;
    eob_run = 0;
    todo = restart_interval ? restart_interval : 0x7fffffff;
  }

  // one sequential block: coefficients are dequantised as they are decoded, products truncated to 16 bits
  void block_sequential(Component& c, int16_t* d) {
    const HuffTable& hd = dc[c.td]; const HuffTable& ha = ac[c.ta];
    const uint16_t* q = quant[c.tq];
    std::memset(d, 0, 64 * sizeof(int16_t));
    int t = br.symbol(hd);
    if (t > 15) fail("bad huffman code");
    c.dc_pred += br.extend(t);
    d[0] = (int16_t)(c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
      int rs = br.symbol(ha), s = rs & 15, r = rs >> 4;
      if (s == 0) { if (rs != 0xF0) break; k += 16; continue; }
      k += r;
      if (k > 63) break;        // corrupt run: ignore the tail
      int z = kNaturalOrder[k++];
      d[z] = (int16_t)(br.extend(s) * q[z]);
    }
  }

  void block_prog_dc(Component& c, int16_t* d) {
    if (se != 0) fail("can't merge dc and ac");
    if (ah == 0) {
      std::memset(d, 0, 64 * sizeof(int16_t));
      int t = br.symbol(dc[c.td]);
      if (t > 15) fail("bad huffman code");
      c.dc_pred += br.extend(t);
      d[0] = (int16_t)(c.dc_pred * (1 << al));
    } else if (br.bit()) {
      d[0] = (int16_t)(d[0] + (int16_t)(1 << al));
    }
  }

  static void refine(BitReader& br, int16_t& v, int16_t bit) {     // correction bit of an already non-zero coefficient
    if (br.bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
  }

  void block_prog_ac(Component& c, int16_t* d) {
    if (ss == 0) fail("can't merge dc and ac");
    const HuffTable& ha = ac[c.ta];
    if (ah == 0) {                       // first pass over this band (T.81 G.1.2.2)
      if (eob_run) { --eob_run; return; }
      for (int k = ss; k <= se;) {
        int rs = br.symbol(ha), s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += (int)br.get(r); break; }
          k += 16;
        } else {
          k += r;
          if (k > 63) break;
          d[kNaturalOrder[k++]] = (int16_t)(br.extend(s) * (1 << al));
        }
      }
      return;
    }
    const int16_t bit = (int16_t)(1 << al);   // refinement pass (T.81 G.1.2.3)
    if (eob_run) {
      --eob_run;
      for (int k = ss; k <= se; ++k) { int16_t& v = d[kNaturalOrder[k]]; if (v != 0) refine(br, v, bit); }
      return;
    }
    int k = ss;
    do {
      int rs = br.symbol(ha), s = rs & 15, r = rs >> 4;
      int fresh = 0;
      if (s == 0) {
        if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += (int)br.get(r); r = 64; }
      } else {
        if (s != 1) fail("bad huffman code");
        fresh = br.bit() ? bit : -bit;
      }
      while (k <= se) {
        int16_t& v = d[kNaturalOrder[k++]];
        if (v != 0) refine(br, v, bit);
        else { if (r == 0) { v = (int16_t)fresh; break; } --r; }
      }
    } while (k <= se);
  }

  // returns false when the scan stops at something that is not a restart marker
  bool mcu_done() {
    if (--todo > 0) return true;
    if (br.n < 24) br.fill();
    if (br.marker < 0xD0 || br.marker > 0xD7) return false;
    reset_entropy();
    return true;
  }

  void decode_scan() {
    br.p = p; br.end = end;
    reset_entropy();
    int16_t tmp[64];
    if (scan_n == 1) {                   // non-interleaved: only the blocks that carry picture, raster order
      Component& c = comp[order[0]];
      int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, stride = c.w2 / 8;
      for (int j = 0; j < bh; ++j)
        for (int i = 0; i < bw; ++i) {
          if (!progressive) {
            block_sequential(c, tmp);
            idct_block(tmp, c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2);
          } else {
            int16_t* d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * stride);
            if (ss == 0) block_prog_dc(c, d); else block_prog_ac(c, d);
          }
          if (!mcu_done()) goto out;
        }
    } else {
      for (int j = 0; j < mcu_y; ++j)
        for (int i = 0; i < mcu_x; ++i) {
          for (int k = 0; k < scan_n; ++k) {
            Component& c = comp[order[k]];
            for (int y = 0; y < c.v; ++y)
              for (int x = 0; x < c.h; ++x) {
                int bx = i * c.h + x, by = j * c.v + y;
                if (!progressive) {
                  block_sequential(c, tmp);
                  idct_block(tmp, c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2);
                } else {
                  block_prog_dc(c, c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8)));
                }
              }
          }
          if (!mcu_done()) goto out;
        }
    }
  out:
    p = br.p;
    if (br.marker) { pending_marker = br.marker; return; }
    while (p < end) { if (u8() == 0xFF) { pending_marker = u8(); break; } }   // trailing padding after the scan
  }

  void finish_progressive() {
    for (int n = 0; n < n_comp; ++n) {
      Component& c = comp[n];
      int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, stride = c.w2 / 8;
      const uint16_t* q = quant[c.tq];
      for (int j = 0; j < bh; ++j)
        for (int i = 0; i < bw; ++i) {
          int16_t* d = c.coeff.data() + 64 * ((size_t)i + (size_t)j * stride);
          for (int k = 0; k < 64; ++k) d[k] = (int16_t)(d[k] * q[k]);
          idct_block(d, c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2);
        }
    }
  }

  void decode() {
    if (next_marker() != 0xD8) fail("no SOI");
    int m = next_marker();
    while (m != 0xC0 && m != 0xC1 && m != 0xC2) {
      read_tables_or_app(m);
      m = next_marker();
      while (m == 0) { if (p >= end) fail("no SOF"); m = next_marker(); }   // padding between segments
    }
    read_frame_header(m);
    m = next_marker();
    while (m != 0xD9) {
      if (m == 0xDA) { read_scan_header(); decode_scan(); }
      else if (m == 0xDC) { u16(); u16(); }
      else read_tables_or_app(m);
      m = next_marker();
    }
    if (progressive) finish_progressive();
  }
};

// ---- upsampling (stb_image.h:3124-3187,3305-3315): `near` is the sample row this output row lies in
const uint8_t* upsample_row(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) {
  if (hs == 1 && vs == 1) return near;
  if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2); return out; }
  if (hs == 2 && vs == 1) {
    if (w == 1) { out[0] = out[1] = near[0]; return out; }
    out[0] = near[0];
    out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
    for (int i = 1; i < w - 1; ++i) {
      int n = 3 * near[i] + 2;
      out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
      out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
    }
    out[2 * w - 2] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
    out[2 * w - 1] = near[w - 1];
    return out;
  }
  if (hs == 2 && vs == 2) {
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
    int cur = 3 * near[0] + far[0];
    out[0] = (uint8_t)((cur + 2) >> 2);
    for (int i = 1; i < w; ++i) {
      int prev = cur;
      cur = 3 * near[i] + far[i];
      out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
      out[2 * i] = (uint8_t)((3 * cur + prev + 8) >> 4);
    }
    out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
    return out;
  }
  for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];   // other ratios: replicate
  return out;
}

// ---- YCbCr -> RGB, 20-bit fixed point (stb_image.h:3317-3343)
constexpr int fix20(float c) { return ((int)(c * 4096.0f + 0.5f)) << 8; }

inline void ycc_to_rgb(uint8_t* out, int y, int cb, int cr) {
  int yf = (y << 20) + (1 << 19);
  cr -= 128; cb -= 128;
  int r = yf + cr * fix20(1.40200f);
  int g = (int)((uint32_t)yf + (uint32_t)(cr * -fix20(0.71414f)) + ((uint32_t)(cb * -fix20(0.34414f)) & 0xffff0000u));
  int b = yf + cb * fix20(1.77200f);
  out[0] = clamp8(r >> 20); out[1] = clamp8(g >> 20); out[2] = clamp8(b >> 20);
}

inline uint8_t mul8(int x, int y) { unsigned t = (unsigned)(x * y + 128); return (uint8_t)((t + (t >> 8)) >> 8); }

bool read_file(const char* path, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0) { std::fclose(f); return false; }
  out.resize((size_t)n);
  size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

} // namespace

// 8-bit decode with stbi_load(..., STBI_default) semantics: 1 channel for grayscale files, 3 otherwise.
bool decode_jpeg(const uint8_t* bytes, size_t n_bytes, Image8& img, std::string& err) {
  try {
    Jpeg j; j.p = bytes; j.end = bytes + n_bytes;
    j.decode();
    const int W = j.width, H = j.height, nc = j.n_comp >= 3 ? 3 : 1;
    const bool is_rgb = j.n_comp == 3 && (j.rgb_ids == 3 || (j.adobe_transform == 0 && !j.jfif));
    img.w = W; img.h = H; img.c = nc;
    img.px.assign((size_t)W * H * nc, 0);
    struct Row { int hs, vs, ystep, w_lores, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; };
    Row rows[4];
    const int planes = j.n_comp;
    for (int k = 0; k < planes; ++k) {
      Row& r = rows[k]; const Component& c = j.comp[k];
      r.hs = j.h_max / c.h; r.vs = j.v_max / c.v;
      r.ystep = r.vs >> 1;
      r.w_lores = (W + r.hs - 1) / r.hs;
      r.ypos = 0;
      r.line0 = r.line1 = c.plane.data();
      r.buf.assign((size_t)r.w_lores * r.hs + 8, 0);
    }
    const uint8_t* line[4] = {};
    for (int y = 0; y < H; ++y) {
      uint8_t* out = img.px.data() + (size_t)y * W * nc;
      for (int k = 0; k < planes; ++k) {
        Row& r = rows[k];
        bool bottom = r.ystep >= (r.vs >> 1);
        line[k] = upsample_row(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
        if (++r.ystep >= r.vs) {
          r.ystep = 0;
          r.line0 = r.line1;
          if (++r.ypos < j.comp[k].y) r.line1 += j.comp[k].w2;
        }
      }
      if (nc == 1) { std::memcpy(out, line[0], (size_t)W); continue; }
      for (int x = 0; x < W; ++x, out += 3) {
        if (j.n_comp == 3) {
          if (is_rgb) { out[0] = line[0][x]; out[1] = line[1][x]; out[2] = line[2][x]; }
          else ycc_to_rgb(out, line[0][x], line[1][x], line[2][x]);
        } else if (j.adobe_transform == 0) {          // CMYK
          int m = line[3][x];
          out[0] = mul8(line[0][x], m); out[1] = mul8(line[1][x], m); out[2] = mul8(line[2][x], m);
        } else if (j.adobe_transform == 2) {          // YCCK
          ycc_to_rgb(out, line[0][x], line[1][x], line[2][x]);
          int m = line[3][x];
          out[0] = mul8(255 - out[0], m); out[1] = mul8(255 - out[1], m); out[2] = mul8(255 - out[2], m);
        } else {
          ycc_to_rgb(out, line[0][x], line[1][x], line[2][x]);
        }
      }
    }
    return true;
  } catch (const Fail& f) {
    err = f.why;
    return false;
  } catch (const std::bad_alloc&) {
    err = "out of memory";
    return false;
  }
}

bool load_image8(const char* path, Image8& img, std::string& err) {
  std::vector<uint8_t> bytes;
  if (!read_file(path, bytes)) { err = "can't open file"; return false; }
  if (bytes.size() >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return decode_jpeg(bytes.data(), bytes.size(), img, err);
  if (bytes.size() >= 8 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G')
    return decode_png(bytes.data(), bytes.size(), img, err);
  err = "unsupported image format (the built-in decoder reads JPEG and PNG; pass a decoder through ptamd_host_scene_load_ex)";
  return false;
}

// stbi_loadf's 8-bit -> float rule (stb_image.h:1565-1583 with the default gamma 2.2, scale 1): colour channels
// pow(v / 255, 2.2) evaluated in single precision (a C++ translation unit resolves pow(float, float) to the float
// overload, as the reference's gpu_processor.cpp does); the last channel of 2- and 4-channel images is alpha, v / 255.
const float* ldr_to_linear_table() {
  struct Table { float v[256]; Table() { for (int i = 0; i < 256; ++i) v[i] = (float)(std::pow((float)i / 255.0f, 2.2f) * 1.0f); } };
  static const Table table;      // thread-safe one-time initialisation
  return table.v;
}

// Radiance RGBE (.hdr), the one format stbi_loadf returns WITHOUT the 8-bit detour and the gamma rule
// (stb_image.h:1209-1222, 6405-6590): header lines up to an empty one (FORMAT=32-bit_rle_rgbe required), "-Y h +X w",
// then flat RGBE pixels (width < 8 or >= 32768) or per-scanline, per-channel run-length coding.  A pixel with a
// non-zero exponent byte e is (r, g, b) * 2^(e - 136); three floats per pixel.  stb's oddity is kept: when a scanline
// of an RLE-sized image does not start with the RLE marker, that pixel becomes pixel 0 and the REST of the file is read
// as flat pixels from pixel 1 on.
bool decode_hdr(const uint8_t* bytes, size_t n_bytes, int& w, int& h, std::vector<float>& out, std::string& err)
{
  const uint8_t* p = bytes; const uint8_t* end = bytes + n_bytes;
  auto get8 = [&]() -> int { return p < end ? *p++ : 0; };
  auto token = [&]() -> std::string {          // one header line (stb_image.h:6427-6447; a byte that ends the file is dropped)
    std::string t;
    int c = get8();
    while (p < end && c != '\n') {
      t += (char)c;
      if (t.size() == 1023) { while (p < end && get8() != '\n') {} break; }
      c = get8();
    }
    return t;
  };
  const std::string id = token();
  if (id != "#?RADIANCE" && id != "#?RGBE") { err = "not HDR"; return false; }
  bool valid = false;
  for (;;) {
    const std::string t = token();
    if (t.empty()) break;
    if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
  }
  if (!valid) { err = "unsupported HDR format"; return false; }
  const std::string dims = token();
  if (dims.compare(0, 3, "-Y ") != 0) { err = "unsupported HDR data layout"; return false; }
  char* rest = nullptr;
  const long height = std::strtol(dims.c_str() + 3, &rest, 10);
  while (*rest == ' ') ++rest;
  if (std::strncmp(rest, "+X ", 3) != 0) { err = "unsupported HDR data layout"; return false; }
  const long width = std::strtol(rest + 3, nullptr, 10);
  if (width <= 0 || height <= 0 || (uint64_t)width * (uint64_t)height > (1ull << 28)) { err = "bad HDR size"; return false; }
  w = (int)width; h = (int)height;
  out.assign((size_t)w * h * 3, 0.0f);
  auto convert = [](float* o, const uint8_t* rgbe) {
    if (rgbe[3] != 0) {
      const float f = (float)std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
      o[0] = rgbe[0] * f; o[1] = rgbe[1] * f; o[2] = rgbe[2] * f;
    } else o[0] = o[1] = o[2] = 0.0f;
  };
  auto flat_from = [&](size_t first_pixel) {
    for (size_t i = first_pixel; i < (size_t)w * h; ++i) {
      uint8_t rgbe[4];
      for (int k = 0; k < 4; ++k) rgbe[k] = (uint8_t)get8();
      convert(&out[i * 3], rgbe);
    }
  };
  if (w < 8 || w >= 32768) { flat_from(0); return true; }
  std::vector<uint8_t> scan((size_t)w * 4);
  for (int j = 0; j < h; ++j) {
    const int c1 = get8(), c2 = get8(); int len = get8();
    if (c1 != 2 || c2 != 2 || (len & 0x80)) {
      const uint8_t rgbe[4] = { (uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)get8() };
      convert(&out[0], rgbe);
      flat_from(1);
      return true;
    }
    len = (len << 8) | get8();
    if (len != w) { err = "invalid decoded scanline length"; return false; }
    for (int k = 0; k < 4; ++k) {
      int i = 0, nleft;
      while ((nleft = w - i) > 0) {
        int count = get8();
        if (count > 128) {
          const uint8_t value = (uint8_t)get8();
          count -= 128;
          if (count > nleft) { err = "bad RLE data in HDR"; return false; }
          for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = value;
        } else {
          if (count > nleft) { err = "bad RLE data in HDR"; return false; }
          if (count == 0 && p >= end) { err = "truncated HDR"; return false; }     // (stb would spin on a truncated file)
          for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = (uint8_t)get8();
        }
      }
    }
    for (int i = 0; i < w; ++i) convert(&out[((size_t)j * w + i) * 3], &scan[(size_t)i * 4]);
  }
  return true;
}

bool load_image_float(const char* path, int* w, int* h, int* c, float** data, std::string& err) {
  {
    std::vector<uint8_t> bytes;
    if (!read_file(path, bytes)) { err = "can't open file"; return false; }
    if ((bytes.size() >= 11 && std::memcmp(bytes.data(), "#?RADIANCE\n", 11) == 0) ||
        (bytes.size() >= 7 && std::memcmp(bytes.data(), "#?RGBE\n", 7) == 0)) {
      std::vector<float> px; int iw = 0, ih = 0;
      if (!decode_hdr(bytes.data(), bytes.size(), iw, ih, px, err)) return false;
      float* out = (float*)std::malloc(px.size() * sizeof(float) + 16);
      if (!out) { err = "out of memory"; return false; }
      std::memcpy(out, px.data(), px.size() * sizeof(float));
      *w = iw; *h = ih; *c = 3; *data = out;
      return true;
    }
  }
  Image8 img;
  if (!load_image8(path, img, err)) return false;
  const size_t n = (size_t)img.w * img.h * img.c;
  float* out = (float*)std::malloc(n * sizeof(float) + 16);
  if (!out) { err = "out of memory"; return false; }
  const float* lut = ldr_to_linear_table();
  const int colour = (img.c & 1) ? img.c : img.c - 1;
  for (size_t i = 0; i < n; i += (size_t)img.c) {
    for (int k = 0; k < colour; ++k) out[i + k] = lut[img.px[i + k]];
    if (colour < img.c) out[i + colour] = (float)img.px[i + colour] / 255.0f;
  }
  *w = img.w; *h = img.h; *c = img.c; *data = out;
  return true;
}

} // namespace ptamd

// ------------------------------------------------------------------------ C-ABI

extern "C" {

int ptamd_image_loadf(const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, float** data)
{
  if (!path || !w || !h || !nb_chan || !data) { ptamd::set_error("ptamd_image_loadf: null argument"); return PTAMD_ERR_ARG; }
  std::string err; int iw = 0, ih = 0, ic = 0;
  *data = nullptr;
  if (!ptamd::load_image_float(path, &iw, &ih, &ic, data, err)) {
    ptamd::set_error(std::string("ptamd_image_loadf: ") + path + ": " + err);
    return PTAMD_ERR_IO;
  }
  *w = iw; *h = ih; *nb_chan = ic;
  return PTAMD_OK;
}

int ptamd_image_load8(const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, uint8_t** data)
{
  if (!path || !w || !h || !nb_chan || !data) { ptamd::set_error("ptamd_image_load8: null argument"); return PTAMD_ERR_ARG; }
  ptamd::Image8 img; std::string err;
  *data = nullptr;
  if (!ptamd::load_image8(path, img, err)) {
    ptamd::set_error(std::string("ptamd_image_load8: ") + path + ": " + err);
    return PTAMD_ERR_IO;
  }
  uint8_t* out = (uint8_t*)std::malloc(img.px.size() + 16);
  if (!out) { ptamd::set_error("ptamd_image_load8: out of memory"); return PTAMD_ERR_LIMIT; }
  std::memcpy(out, img.px.data(), img.px.size());
  *w = img.w; *h = img.h; *nb_chan = img.c; *data = out;
  return PTAMD_OK;
}

void ptamd_image_free(void* data) { std::free(data); }

} // extern "C"
