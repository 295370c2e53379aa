// image_png.cpp — PNG decoding for the scene loader (the second format a material library is likely to name;
// the reference would read it through stb_image's stbi_loadf, material_loader.cpp:97).
//
// Written from the PNG specification (ISO/IEC 15948) and RFC 1950/1951: chunk walk, zlib/deflate inflate,
// the five scanline filters, Adam7, bit depths 1/2/4/8/16, colour types 0/2/3/4/6, PLTE and tRNS.  Everything is
// integer work with one possible answer; where the specification leaves the 8-bit result open, stb_image 2.16's
// choices are taken so that the pixels equal the reference's:
//   * 1/2/4-bit gray is scaled by 0xff/0x55/0x11 (replication), palette indices are not   (stb_image.h:4405-4450)
//   * 16-bit samples keep their high byte                                              (stb_image.h:1001-1015)
//   * tRNS on gray/RGB adds an alpha channel that is 0 exactly on the key colour (compared at full depth), a
//     paletted image becomes RGB, or RGBA when a tRNS chunk is present                (stb_image.h:4535-4620,4745-4765)
//   * CRCs and the Adler-32 are not verified, an unknown critical chunk is an error    (stb_image.h:4820-4840)
// Channel count = 1 (gray), 2 (gray+alpha), 3, 4 as stbi_load(..., STBI_default) reports it.
// tests/test_ref_thirdparty.py compares against the real stb_image (oracle/_ref) on PNGs of every colour type,
// depth, filter and interlace mode.
#include "ptamd_internal.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace ptamd {
namespace {

struct Fail { const char* why; };
[[noreturn]] void fail(const char* why) { throw Fail{ why }; }

// ---- inflate (RFC 1951)
struct Huff {
  uint16_t count[16] = {};
  uint16_t symbol[288] = {};
  uint16_t fast[512] = {};       // 9-bit look-ahead: (len << 12) | symbol, 0 = longer code

  void build(const uint8_t* lens, int n) {
    std::memset(count, 0, sizeof count);
    std::memset(fast, 0, sizeof fast);
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) fail("bad code lengths"); }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    uint16_t next[16]; std::memcpy(next, offs, sizeof next);
    for (int i = 0; i < n; ++i) if (lens[i]) symbol[next[lens[i]]++] = (uint16_t)i;
    // look-ahead table indexed by the next 9 stream bits (LSB first => codes bit-reversed)
    int code = 0, idx = 0;
    for (int l = 1; l <= 9; ++l) {
      for (int k = 0; k < count[l]; ++k, ++code, ++idx) {
        int rev = 0;
        for (int b = 0; b < l; ++b) rev |= ((code >> b) & 1) << (l - 1 - b);
        for (int f = rev; f < 512; f += 1 << l) fast[f] = (uint16_t)((l << 12) | symbol[idx]);
      }
      code <<= 1;
    }
  }
};

struct Inflater {
  const uint8_t* p; const uint8_t* end;
  uint64_t acc = 0; int n = 0;
  int padded = 0;                // zero bytes fed after the end of the input
  std::vector<uint8_t>& out;
  Inflater(const uint8_t* b, const uint8_t* e, std::vector<uint8_t>& o) : p(b), end(e), out(o) {}

  void need(int k) {
    while (n < k) {
      uint64_t b = 0;
      if (p < end) b = *p++; else if (++padded > 8) fail("unexpected end of deflate data");
      acc |= b << n; n += 8;
    }
  }
  uint32_t bits(int k) { if (k == 0) return 0; need(k); uint32_t v = (uint32_t)(acc & ((1ull << k) - 1)); acc >>= k; n -= k; return v; }

  int decode(const Huff& h) {
    need(16);
    uint16_t e = h.fast[acc & 511];
    if (e) { int l = e >> 12; acc >>= l; n -= l; return e & 0xFFF; }
    int code = 0, first = 0, idx = 0;
    uint32_t look = (uint32_t)acc;
    for (int l = 1; l < 16; ++l) {
      code |= (int)(look & 1); look >>= 1;
      int cnt = h.count[l];
      if (code - cnt < first) { acc >>= l; n -= l; return h.symbol[idx + (code - first)]; }
      idx += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    fail("bad huffman code");
  }

  void block(const Huff& lit, const Huff& dist) {
    static const uint16_t lbase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    static const uint8_t lext[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    static const uint16_t dbase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                        8193, 12289, 16385, 24577 };
    static const uint8_t dext[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    for (;;) {
      int s = decode(lit);
      if (s < 256) { out.push_back((uint8_t)s); continue; }
      if (s == 256) return;
      s -= 257;
      if (s >= 29) fail("bad huffman code");
      int len = lbase[s] + (int)bits(lext[s]);
      int d = decode(dist);
      if (d >= 30) fail("bad huffman code");
      size_t back = dbase[d] + bits(dext[d]);
      if (back > out.size()) fail("bad dist");
      size_t from = out.size() - back;
      for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
    }
  }

  void run(bool zlib_header) {
    if (zlib_header) {
      int cmf = (int)bits(8), flg = (int)bits(8);
      if ((cmf * 256 + flg) % 31 != 0) fail("bad zlib header");
      if (flg & 32) fail("no preset dict");
      if ((cmf & 15) != 8) fail("bad compression");
    }
    int final;
    do {
      final = (int)bits(1);
      int type = (int)bits(2);
      if (type == 0) {
        acc >>= n & 7; n -= n & 7;                  // to the byte boundary
        uint32_t len = bits(16), nlen = bits(16);
        if ((len ^ 0xFFFF) != nlen) fail("zlib corrupt");
        for (uint32_t i = 0; i < len; ++i) out.push_back((uint8_t)bits(8));
      } else if (type == 1) {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        uint8_t d[32]; std::memset(d, 5, sizeof d);
        Huff lit, dist; lit.build(l, 288); dist.build(d, 32);
        block(lit, dist);
      } else if (type == 2) {
        static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
        int hlit = (int)bits(5) + 257, hdist = (int)bits(5) + 1, hclen = (int)bits(4) + 4;
        uint8_t cl[19] = {};
        for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)bits(3);
        Huff clh; clh.build(cl, 19);
        uint8_t lens[288 + 32 + 140] = {};
        int i = 0;
        while (i < hlit + hdist) {
          int c = decode(clh);
          if (c < 16) lens[i++] = (uint8_t)c;
          else {
            uint8_t fill = 0; int rep;
            if (c == 16) { if (i == 0) fail("bad codelengths"); fill = lens[i - 1]; rep = 3 + (int)bits(2); }
            else if (c == 17) rep = 3 + (int)bits(3);
            else rep = 11 + (int)bits(7);
            if (i + rep > hlit + hdist) fail("bad codelengths");
            while (rep--) lens[i++] = fill;
          }
        }
        Huff lit, dist; lit.build(lens, hlit); dist.build(lens + hlit, hdist);
        block(lit, dist);
      } else fail("zlib corrupt");
    } while (!final);
  }
};

inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

struct Png {
  uint32_t width = 0, height = 0;
  int depth = 0, colour = 0, interlace = 0;
  int img_n = 0;                 // samples per pixel in the stream
  int pal_n = 0;                 // 0, 3 or 4 (palette with tRNS)
  uint8_t palette[1024] = {};
  uint32_t pal_len = 0;
  bool has_trans = false; uint8_t tc[3] = {}; uint16_t tc16[3] = {};
  bool iphone = false;
  std::vector<uint8_t> idata;

  // unfilter a (sub)image of w x h pixels from raw into samples (bytes as stored: packed for depth < 8, big-endian for 16)
  void unfilter(const uint8_t*& raw, const uint8_t* raw_end, uint32_t w, uint32_t h, std::vector<uint8_t>& rows, uint32_t& row_bytes) {
    const int bpp = depth < 8 ? 1 : img_n * (depth / 8);
    row_bytes = (uint32_t)(((uint64_t)img_n * w * depth + 7) >> 3);
    if ((uint64_t)(raw_end - raw) < (uint64_t)(row_bytes + 1) * h) fail("not enough pixels");
    rows.assign((size_t)row_bytes * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
      const int ft = *raw++;
      if (ft > 4) fail("invalid filter");
      uint8_t* cur = rows.data() + (size_t)row_bytes * y;
      const uint8_t* up = y ? cur - row_bytes : nullptr;
      for (uint32_t i = 0; i < row_bytes; ++i) {
        const int a = i >= (uint32_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (uint32_t)bpp) ? up[i - bpp] : 0;
        int v = raw[i];
        switch (ft) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: break;
        }
        cur[i] = (uint8_t)v;
      }
      raw += row_bytes;
    }
  }

  // one row of stored samples -> one 16-bit value per sample (8-bit and lower: the expanded byte; 16: the full value)
  void expand_row(const uint8_t* src, uint32_t w, uint16_t* dst) const {
    static const uint8_t scale_table[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    const uint32_t count = w * img_n;
    if (depth == 8) { for (uint32_t i = 0; i < count; ++i) dst[i] = src[i]; return; }
    if (depth == 16) { for (uint32_t i = 0; i < count; ++i) dst[i] = (uint16_t)((src[2 * i] << 8) | src[2 * i + 1]); return; }
    const int scale = colour == 0 ? scale_table[depth] : 1;
    const int per = 8 / depth, mask = (1 << depth) - 1;
    for (uint32_t i = 0; i < count; ++i) {
      const int shift = (per - 1 - (int)(i % per)) * depth;
      dst[i] = (uint16_t)((scale * ((src[i / per] >> shift) & mask)) & 0xFF);
    }
  }

  void decode(const uint8_t* bytes, size_t n_bytes, Image8& img) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (n_bytes < 8 || std::memcmp(bytes, sig, 8) != 0) fail("bad png sig");
    const uint8_t* p = bytes + 8; const uint8_t* end = bytes + n_bytes;
    auto u8 = [&]() -> uint32_t { return p < end ? *p++ : 0; };
    auto u32 = [&]() -> uint32_t { uint32_t a = u8(), b = u8(), c = u8(), d = u8(); return (a << 24) | (b << 16) | (c << 8) | d; };
    bool first = true, have_idat = false;
    for (;;) {
      if (p >= end) fail("outofdata");
      const uint32_t len = u32(), type = u32();
      const uint8_t* body = p;
      if (len > (uint32_t)(end - p)) { if (type == 0x49444154u || type == 0x49484452u) fail("outofdata"); }
      switch (type) {
      case 0x43674249u: iphone = true; break;                                   // CgBI
      case 0x49484452u: {                                                       // IHDR
        if (!first) fail("multiple IHDR");
        first = false;
        if (len != 13) fail("bad IHDR len");
        width = u32(); height = u32();
        if (width > (1u << 24) || height > (1u << 24)) fail("too large");
        depth = (int)u8();
        if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) fail("1/2/4/8/16-bit only");
        colour = (int)u8();
        if (colour > 6) fail("bad ctype");
        if (colour == 3 && depth == 16) fail("bad ctype");
        if (colour == 3) pal_n = 3; else if (colour & 1) fail("bad ctype");
        if (u8()) fail("bad comp method");
        if (u8()) fail("bad filter method");
        interlace = (int)u8();
        if (interlace > 1) fail("bad interlace method");
        if (!width || !height) fail("0-pixel image");
        if (!pal_n) {
          img_n = ((colour & 2) ? 3 : 1) + ((colour & 4) ? 1 : 0);
          if ((1u << 30) / width / (uint32_t)img_n < height) fail("too large");
        } else {
          img_n = 1;
          if ((1u << 30) / width / 4 < height) fail("too large");
        }
        break;
      }
      case 0x504c5445u: {                                                       // PLTE
        if (first) fail("first not IHDR");
        if (len > 256 * 3) fail("invalid PLTE");
        pal_len = len / 3;
        if (pal_len * 3 != len) fail("invalid PLTE");
        for (uint32_t i = 0; i < pal_len; ++i) {
          palette[i * 4 + 0] = (uint8_t)u8(); palette[i * 4 + 1] = (uint8_t)u8(); palette[i * 4 + 2] = (uint8_t)u8();
          palette[i * 4 + 3] = 255;
        }
        break;
      }
      case 0x74524e53u: {                                                       // tRNS
        if (first) fail("first not IHDR");
        if (have_idat) fail("tRNS after IDAT");
        if (pal_n) {
          if (pal_len == 0) fail("tRNS before PLTE");
          if (len > pal_len) fail("bad tRNS len");
          pal_n = 4;
          for (uint32_t i = 0; i < len; ++i) palette[i * 4 + 3] = (uint8_t)u8();
        } else {
          static const uint8_t scale_table[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
          if (!(img_n & 1)) fail("tRNS with alpha");
          if (len != (uint32_t)img_n * 2) fail("bad tRNS len");
          has_trans = true;
          for (int k = 0; k < img_n; ++k) {
            const uint32_t v = (u8() << 8) | u8();
            if (depth == 16) tc16[k] = (uint16_t)v; else tc[k] = (uint8_t)((v & 255) * scale_table[depth]);
          }
        }
        break;
      }
      case 0x49444154u: {                                                       // IDAT
        if (first) fail("first not IHDR");
        if (pal_n && !pal_len) fail("no PLTE");
        have_idat = true;
        idata.insert(idata.end(), body, body + len);
        break;
      }
      case 0x49454e44u: {                                                       // IEND
        if (first) fail("first not IHDR");
        if (!have_idat) fail("no IDAT");
        finish(img);
        return;
      }
      default:
        if (first) fail("first not IHDR");
        if ((type & (1u << 29)) == 0) fail("unknown critical PNG chunk");
        break;
      }
      p = body + len;          // chunk data, then its CRC (not verified)
      if (p > end) p = end;
      u32();
    }
  }

  void finish(Image8& img) {
    // bytes the filtered stream must hold (all seven sub-images when interlaced)
    static const uint8_t xo[7] = { 0, 4, 0, 2, 0, 1, 0 }, yo[7] = { 0, 0, 4, 0, 2, 0, 1 };
    static const uint8_t xs[7] = { 8, 8, 4, 4, 2, 2, 1 }, ys[7] = { 8, 8, 8, 4, 4, 2, 2 };
    auto stream_bytes = [&](uint32_t w, uint32_t h) -> uint64_t { return ((((uint64_t)img_n * w * depth + 7) >> 3) + 1) * h; };
    uint64_t expected = 0;
    if (!interlace) expected = stream_bytes(width, height);
    else for (int k = 0; k < 7; ++k) {
      const uint32_t w = (width - xo[k] + xs[k] - 1) / xs[k], h = (height - yo[k] + ys[k] - 1) / ys[k];
      if (w && h) expected += stream_bytes(w, h);
    }
    std::vector<uint8_t> raw;
    const uint64_t most = (uint64_t)idata.size() * 1032 + 1024;            // deflate cannot expand further than this
    if (expected > most) fail("not enough pixels");
    raw.reserve((size_t)expected);
    Inflater inf(idata.data(), idata.data() + idata.size(), raw);
    inf.run(!iphone);
    if (raw.size() < expected) fail("not enough pixels");
    const int out_n = img_n + (has_trans ? 1 : 0);
    std::vector<uint16_t> px((size_t)width * height * out_n);   // full-depth samples, alpha 0xffff/0xff added for tRNS
    const uint16_t opaque = depth == 16 ? 0xFFFF : 0xFF;
    std::vector<uint8_t> rows; std::vector<uint16_t> line;
    const uint8_t* rp = raw.data(); const uint8_t* rend = raw.data() + raw.size();
    auto place = [&](uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, uint32_t dx, uint32_t dy) {
      uint32_t row_bytes = 0;
      unfilter(rp, rend, w, h, rows, row_bytes);
      line.resize((size_t)w * img_n);
      for (uint32_t y = 0; y < h; ++y) {
        expand_row(rows.data() + (size_t)row_bytes * y, w, line.data());
        for (uint32_t x = 0; x < w; ++x) {
          uint16_t* d = px.data() + ((size_t)(y0 + y * dy) * width + (x0 + x * dx)) * out_n;
          for (int c = 0; c < img_n; ++c) d[c] = line[(size_t)x * img_n + c];
          if (has_trans) d[img_n] = opaque;
        }
      }
    };
    if (!interlace) place(width, height, 0, 0, 1, 1);
    else {
      for (int k = 0; k < 7; ++k) {
        const uint32_t w = (width - xo[k] + xs[k] - 1) / xs[k], h = (height - yo[k] + ys[k] - 1) / ys[k];
        if (w && h) place(w, h, xo[k], yo[k], xs[k], ys[k]);
      }
    }
    if (has_trans) {
      const size_t count = (size_t)width * height;
      for (size_t i = 0; i < count; ++i) {
        uint16_t* d = px.data() + i * out_n;
        bool key = true;
        for (int c = 0; c < img_n; ++c) key = key && d[c] == (depth == 16 ? tc16[c] : (uint16_t)tc[c]);
        // gray: alpha is set either way; RGB: only cleared on the key (it already holds "opaque")
        if (img_n == 1) d[1] = key ? 0 : opaque; else if (key) d[3] = 0;
      }
    }
    const int final_n = pal_n ? pal_n : out_n;
    img.w = (int)width; img.h = (int)height; img.c = final_n;
    img.px.assign((size_t)width * height * final_n, 0);
    const size_t count = (size_t)width * height;
    if (pal_n) {
      for (size_t i = 0; i < count; ++i) {
        const uint8_t* e = palette + 4 * (px[i] & 0xFF);
        for (int c = 0; c < pal_n; ++c) img.px[i * pal_n + c] = e[c];
      }
    } else {
      for (size_t i = 0; i < count * out_n; ++i) img.px[i] = (uint8_t)(depth == 16 ? px[i] >> 8 : px[i]);
    }
  }
};

} // namespace

bool decode_png(const uint8_t* bytes, size_t n_bytes, Image8& img, std::string& err)
{
  try {
    Png png;
    png.decode(bytes, n_bytes, img);
    return true;
  } catch (const Fail& f) {
    err = f.why;
    return false;
  } catch (const std::bad_alloc&) {
    err = "out of memory";
    return false;
  }
}

// ---- writer: 8-bit gray / gray+alpha / RGB / RGBA, filter 0, stored (uncompressed) deflate blocks.  Output only needs to
// be a valid PNG any viewer opens ("PNG/PPM writer for the RGBA8 buffer, row 0 = top").
namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t* d, size_t n)
{
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
  return crc;
}

void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }

void put_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data)
{
  put32(out, (uint32_t)data.size());
  const size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  put32(out, crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}

} // namespace

bool encode_png(const uint8_t* pixels, int w, int h, int channels, std::vector<uint8_t>& out)
{
  if (!pixels || w <= 0 || h <= 0 || channels < 1 || channels > 4) return false;
  static const uint8_t colour_of[5] = { 0, 0, 4, 2, 6 };
  out.assign({ 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' });
  std::vector<uint8_t> d;
  put32(d, (uint32_t)w); put32(d, (uint32_t)h);
  d.push_back(8); d.push_back(colour_of[channels]); d.push_back(0); d.push_back(0); d.push_back(0);
  put_chunk(out, "IHDR", d);
  const size_t row = (size_t)w * channels;
  std::vector<uint8_t> raw;
  raw.reserve((row + 1) * h);
  for (int y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), pixels + row * y, pixels + row * (y + 1)); }
  d.clear();
  d.push_back(0x78); d.push_back(0x01);
  uint32_t a = 1, b = 0;
  for (size_t pos = 0; pos < raw.size() || pos == 0;) {
    const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
    d.push_back(pos + n >= raw.size() ? 1 : 0);
    d.push_back((uint8_t)n); d.push_back((uint8_t)(n >> 8)); d.push_back((uint8_t)~n); d.push_back((uint8_t)(~n >> 8));
    d.insert(d.end(), raw.begin() + pos, raw.begin() + pos + n);
    for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
    pos += n;
    if (n == 0) break;
  }
  put32(d, (b << 16) | a);
  put_chunk(out, "IDAT", d);
  put_chunk(out, "IEND", {});
  return true;
}

} // namespace ptamd

extern "C" int ptamd_image_save_png(const char* path, const uint8_t* pixels, int32_t w, int32_t h, int32_t channels)
{
  std::vector<uint8_t> bytes;
  if (!path || !ptamd::encode_png(pixels, w, h, channels, bytes)) { ptamd::set_error("ptamd_image_save_png: bad argument"); return PTAMD_ERR_ARG; }
  FILE* f = std::fopen(path, "wb");
  if (!f) { ptamd::set_error(std::string("ptamd_image_save_png: cannot open '") + path + "'"); return PTAMD_ERR_IO; }
  const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
  const int rc = std::fclose(f);
  if (n != bytes.size() || rc != 0) { ptamd::set_error(std::string("ptamd_image_save_png: short write to '") + path + "'"); return PTAMD_ERR_IO; }
  return PTAMD_OK;
}
