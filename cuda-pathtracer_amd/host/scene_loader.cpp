// scene_loader.cpp — host-side scene input for the megakernel: .scene / OBJ / MTL -> the
// flattened arrays of ptamd_scene_desc.
//
// Behavioural contract (what the device path receives must be what the reference would
// have uploaded on Linux):
//   .scene grammar ........ cuda_opengl/src/scene/scene.cpp:32-170
//   OBJ semantics ......... tinyobjloader 1.0.8 as the reference calls it
//                           (scene.cpp:341, LoadObj(..., triangulate = true)): shapes split
//                           on `o`/`g`, per-face material ids, fan triangulation, negative
//                           indices, file order preserved
//   Face flattening ....... scene.cpp:218-262 (AoS Face, per-face tangent)
//   Materials/textures .... material_loader.cpp:164-401 (diffuse rgb + specular a packed
//                           into one RGBA float texture, 1x1 fallbacks, global texture ids)
// Image decoding goes through an ImageProvider (default: the built-in decoder of image_decode.cpp);
// a file that cannot be opened "fails to load", which is exactly what happens to indoor.mtl's
// backslash paths in the reference on Linux (material_loader.cpp:97-104).
// The lexical rules that decide VALUES are tinyobj's, restated: its own decimal reader (not strtod),
// its line splitting, its texture-option grammar, its material flushing.  tests/test_ref_thirdparty.py
// checks this parser against the real tinyobj 1.0.8 compiled from /root/reference (oracle/_ref) on the
// shipped assets and on fuzzed OBJ/MTL text; the outputs must be bit-identical.
#include "ptamd_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace ptamd {

namespace {

constexpr double kPi = 3.14159265358979323846; // M_PI of <math.h> (scene.cpp:10-11,76,101)

inline ptamd_float3 f3(float x, float y, float z) { return ptamd_float3{ x, y, z }; }
inline ptamd_float3 sub3(ptamd_float3 a, ptamd_float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline ptamd_float3 cross3(ptamd_float3 a, ptamd_float3 b)
{
  return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline ptamd_float3 normalize3(ptamd_float3 v)
{
  float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  return f3(v.x * inv, v.y * inv, v.z * inv);
}

bool at_end(std::stringstream& s) { return s.peek() == std::char_traits<char>::eof(); }

// scene.cpp:44-59: three floats; fails when the stream ends before the third one starts
bool read_vec3(ptamd_float3& out, std::stringstream& s)
{
  if (at_end(s)) return false;
  s >> out.x;
  if (at_end(s)) return false;
  s >> out.y;
  if (at_end(s)) return false;
  s >> out.z;
  return true;
}

float read_float_or(std::stringstream& s, float fallback) // scene.cpp:32-42
{
  if (at_end(s)) return fallback;
  float v = 0.0f;
  s >> v;
  return v;
}

bool read_camera(ptamd_camera& cam, std::stringstream& s) // scene.cpp:61-84
{
  if (!read_vec3(cam.position, s)) return false;
  if (!read_vec3(cam.dir, s)) return false;
  cam.dir = normalize3(cam.dir);
  if (at_end(s)) return false;
  s >> cam.fov_x;
  cam.fov_x = (float)(((double)cam.fov_x * kPi) / 180.0);
  cam.focus_dist = read_float_or(s, 2.0f);
  cam.aperture = read_float_or(s, 0.125f);
  cam.speed = 1.4f;
  return true;
}

// ---------------------------------------------------------------- OBJ / MTL

struct MtlEntry {
  std::string name;
  float diffuse[3] = { 0.f, 0.f, 0.f };
  float specular[3] = { 0.f, 0.f, 0.f };
  float ior = 1.f;
  std::string diffuse_tex, specular_tex, bump_tex, normal_tex;
};

const char* skip_ws(const char* p)
{
  while (*p == ' ' || *p == '\t') ++p;
  return p;
}
bool is_space(char c) { return c == ' ' || c == '\t'; }
bool is_eol(char c) { return c == '\0' || c == '\r' || c == '\n'; }

bool is_digit(char c) { return (unsigned)(c - '0') < 10u; }

// tinyobj's own decimal reader (tiny_obj_loader.h:498-611), restated because it, not strtod, defines
// the vertex bits: sign, digits accumulated as m = m * 10 + d, fraction digits added as
// d * 10^-k (table for k < 8, pow beyond), optional exponent applied as ldexp(m * 5^e, e).
// Greedy: stops at the first character outside the grammar.  No leading '.', no inf/nan, no hex.
bool parse_decimal(const char* s, const char* end, double& out)
{
  static const double kNegPow10[8] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
  if (s >= end) return false;
  const char* c = s;
  bool negative = false;
  if (*c == '+' || *c == '-') { negative = (*c == '-'); ++c; }
  else if (!is_digit(*c)) return false;
  double mantissa = 0.0;
  int n_int = 0;
  for (; c != end && is_digit(*c); ++c, ++n_int) { mantissa *= 10; mantissa += (int)(*c - '0'); }
  if (n_int == 0) return false;
  int exp10 = 0;
  if (c != end && (*c == '.' || *c == 'e' || *c == 'E')) {
    if (*c == '.') {
      ++c;
      for (int k = 1; c != end && is_digit(*c); ++c, ++k)
        mantissa += (int)(*c - '0') * (k < 8 ? kNegPow10[k] : std::pow(10.0, (double)-k));
    }
    if (c != end && (*c == 'e' || *c == 'E')) {
      ++c;
      bool exp_negative = false;
      if (c != end && (*c == '+' || *c == '-')) { exp_negative = (*c == '-'); ++c; }
      else if (!is_digit(*c)) return false;          // "1e" / "1ex": the whole number is rejected
      int n_exp = 0;
      for (; c != end && is_digit(*c); ++c, ++n_exp) exp10 = exp10 * 10 + (int)(*c - '0');
      if (exp_negative) exp10 = -exp10;
      if (n_exp == 0) return false;
    }
  }
  const double v = exp10 ? std::ldexp(mantissa * std::pow(5.0, (double)exp10), exp10) : mantissa;
  out = negative ? -v : v;
  return true;
}

// parseReal (tiny_obj_loader.h:613-621): always consumes one blank-delimited token, parsed or not
float parse_real(const char*& p, double fallback = 0.0)
{
  p += std::strspn(p, " \t");
  const char* end = p + std::strcspn(p, " \t\r");
  double v = fallback;
  parse_decimal(p, end, v);
  p = end;
  return (float)v;
}

void skip_token(const char*& p)
{
  p += std::strspn(p, " \t");
  p += std::strcspn(p, " \t\r");
}

// safeGetline (tiny_obj_loader.h:392-423): lines end at "\n", "\r\n" or a lone "\r"
bool read_lines(const std::string& path, std::vector<std::string>& lines)
{
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  std::string text;        // a path that opens but cannot be read (a directory) reads as empty, as it does for tinyobj
  char buf[1 << 16];
  while (f.read(buf, sizeof buf) || f.gcount() > 0) text.append(buf, (size_t)f.gcount());
  std::string cur;
  for (size_t i = 0; i < text.size(); ++i) {
    const char c = text[i];
    if (c == '\n' || c == '\r') {
      if (c == '\r' && i + 1 < text.size() && text[i + 1] == '\n') ++i;
      lines.push_back(cur);
      cur.clear();
    } else cur += c;
  }
  if (!cur.empty()) lines.push_back(cur);
  return true;
}

bool option(const char* p, const char* kw)
{
  size_t n = std::strlen(kw);
  return std::strncmp(p, kw, n) == 0 && is_space(p[n]);
}

// ParseTextureNameAndOption (tiny_obj_loader.h:833-918).  Known options swallow a FIXED number of
// tokens whatever those tokens are (`-s 1 2 tex.jpg` eats the file name); anything else is a file name,
// and the last file name on the line wins.
std::string parse_texture_name(const char* p)
{
  std::string name;
  while (!is_eol(*p)) {
    p += std::strspn(p, " \t");
    int eat = -1;
    if (option(p, "-blendu") || option(p, "-blendv")) { p += 8; eat = 1; }
    else if (option(p, "-clamp") || option(p, "-boost")) { p += 7; eat = 1; }
    else if (option(p, "-bm") || option(p, "-mm")) { eat = (p[1] == 'm') ? 2 : 1; p += 4; }
    else if (option(p, "-o") || option(p, "-s") || option(p, "-t")) { p += 3; eat = 3; }
    else if (option(p, "-type")) { p += 5; eat = 1; }
    else if (option(p, "-imfchan")) { p += 9; eat = 1; }
    if (eat >= 0) {
      for (int i = 0; i < eat; ++i) skip_token(p);
      continue;
    }
    size_t n = std::strcspn(p, " \t\r");
    name.assign(p, n);
    p += n;
    p += std::strspn(p, " \t");
  }
  return name;
}

bool starts(const char* p, const char* kw)
{
  size_t n = std::strlen(kw);
  return std::strncmp(p, kw, n) == 0 && is_space(p[n]);
}

// LoadMtl (tiny_obj_loader.h:1010-1393).  A material is flushed when the next `newmtl` arrives and its name
// is not empty; the LAST one is flushed unconditionally, so a file without any `newmtl` (or an mtllib name
// that opens but yields no line, e.g. a directory) still contributes one default material named "".
bool load_mtl(const std::string& path, std::vector<MtlEntry>& out, std::map<std::string, int>& index)
{
  std::vector<std::string> lines;
  if (!read_lines(path, lines)) return false;
  MtlEntry cur;
  for (std::string& line : lines) {
    size_t last = line.find_last_not_of(" \t");
    line.erase(last == std::string::npos ? 0 : last + 1);
    if (line.empty()) continue;
    const char* p = skip_ws(line.c_str());
    if (*p == '\0' || *p == '#') continue;
    if (starts(p, "newmtl")) {
      if (!cur.name.empty()) {
        index.insert(std::make_pair(cur.name, (int)out.size()));
        out.push_back(cur);
      }
      cur = MtlEntry();
      cur.name = p + 7;
    } else if (p[0] == 'K' && p[1] == 'd' && is_space(p[2])) {
      p += 2;
      cur.diffuse[0] = parse_real(p); cur.diffuse[1] = parse_real(p); cur.diffuse[2] = parse_real(p);
    } else if (p[0] == 'K' && p[1] == 's' && is_space(p[2])) {
      p += 2;
      cur.specular[0] = parse_real(p); cur.specular[1] = parse_real(p); cur.specular[2] = parse_real(p);
    } else if (p[0] == 'N' && p[1] == 'i' && is_space(p[2])) {
      p += 2;
      cur.ior = parse_real(p);
    } else if (starts(p, "map_Kd")) {
      std::string n = parse_texture_name(p + 7);
      if (!n.empty()) cur.diffuse_tex = n;
    } else if (starts(p, "map_Ks")) {
      std::string n = parse_texture_name(p + 7);
      if (!n.empty()) cur.specular_tex = n;
    } else if (starts(p, "map_bump") || starts(p, "map_Bump")) {
      std::string n = parse_texture_name(p + 9);
      if (!n.empty()) cur.bump_tex = n;
    } else if (starts(p, "bump")) {
      std::string n = parse_texture_name(p + 5);
      if (!n.empty()) cur.bump_tex = n;
    } else if (starts(p, "norm")) {
      std::string n = parse_texture_name(p + 5);
      if (!n.empty()) cur.normal_tex = n;
    }
  }
  index.insert(std::make_pair(cur.name, (int)out.size()));
  out.push_back(cur);
  return true;
}

struct Corner { int v = -1, vt = -1, vn = -1; };

bool fix_index(int idx, int n, int& out)
{
  if (idx > 0) { out = idx - 1; return true; }
  if (idx == 0) return false;
  out = n + idx;
  return true;
}

// i, i/j, i//k, i/j/k
bool parse_corner(const char*& p, int nv, int nvn, int nvt, Corner& c)
{
  if (!fix_index(std::atoi(p), nv, c.v)) return false;
  p += std::strcspn(p, "/ \t\r");
  if (*p != '/') return true;
  ++p;
  if (*p == '/') {
    ++p;
    if (!fix_index(std::atoi(p), nvn, c.vn)) return false;
    p += std::strcspn(p, "/ \t\r");
    return true;
  }
  if (!fix_index(std::atoi(p), nvt, c.vt)) return false;
  p += std::strcspn(p, "/ \t\r");
  if (*p != '/') return true;
  ++p;
  if (!fix_index(std::atoi(p), nvn, c.vn)) return false;
  p += std::strcspn(p, "/ \t\r");
  return true;
}

struct ObjData {
  std::vector<float> v, vn, vt;
  // one entry per shape (tinyobj shape_t), triangles already fanned
  struct Tri { Corner c[3]; int material; };
  std::vector<std::vector<Tri>> shapes;
  std::vector<MtlEntry> materials;
};

bool load_obj(const std::string& path, const std::string& mtl_dir, ObjData& obj, std::string& err)
{
  std::vector<std::string> lines;
  if (!read_lines(path, lines)) {
    err = "cannot open OBJ '" + path + "'";
    return false;
  }
  std::map<std::string, int> mtl_index;
  std::vector<ObjData::Tri> shape;                   // triangles already exported to the open shape
  std::vector<std::vector<Corner>> group;            // pending polygons (one material)
  int material = -1;

  auto export_group = [&]() -> bool {
    if (group.empty()) return false;
    for (const auto& poly : group) {
      if (poly.size() < 3) continue;
      for (size_t k = 2; k < poly.size(); ++k) {
        ObjData::Tri t;
        t.c[0] = poly[0]; t.c[1] = poly[k - 1]; t.c[2] = poly[k];
        t.material = material;
        shape.push_back(t);
      }
    }
    return true;
  };

  for (const std::string& line : lines) {
    if (line.empty()) continue;
    const char* p = skip_ws(line.c_str());
    if (*p == '\0' || *p == '#') continue;
    if (p[0] == 'v' && is_space(p[1])) {
      p += 2;
      float x = parse_real(p), y = parse_real(p), z = parse_real(p);
      obj.v.push_back(x); obj.v.push_back(y); obj.v.push_back(z);
    } else if (p[0] == 'v' && p[1] == 'n' && is_space(p[2])) {
      p += 3;
      float x = parse_real(p), y = parse_real(p), z = parse_real(p);
      obj.vn.push_back(x); obj.vn.push_back(y); obj.vn.push_back(z);
    } else if (p[0] == 'v' && p[1] == 't' && is_space(p[2])) {
      p += 3;
      float x = parse_real(p), y = parse_real(p);
      obj.vt.push_back(x); obj.vt.push_back(y);
    } else if (p[0] == 'f' && is_space(p[1])) {
      p = skip_ws(p + 2);
      std::vector<Corner> poly;
      while (!is_eol(*p)) {
        Corner c;
        if (!parse_corner(p, (int)(obj.v.size() / 3), (int)(obj.vn.size() / 3), (int)(obj.vt.size() / 2), c)) {
          err = "bad face index in '" + path + "'";
          return false;
        }
        poly.push_back(c);
        p += std::strspn(p, " \t\r");
      }
      group.push_back(std::move(poly));
    } else if (starts(p, "usemtl")) {
      std::string name = p + 7;
      int id = -1;
      auto it = mtl_index.find(name);
      if (it != mtl_index.end()) id = it->second;
      if (id != material) {
        export_group();
        group.clear();
        material = id;
      }
    } else if (starts(p, "mtllib")) {
      // names split on single blanks, first file that opens wins (tiny_obj_loader.h:1617-1650); an empty
      // name (two blanks in a row) is tried too, as tinyobj does: it opens the MTL directory itself
      std::stringstream ss(std::string(p + 7));
      std::string fn;
      while (std::getline(ss, fn, ' '))
        if (load_mtl(mtl_dir + fn, obj.materials, mtl_index)) break;
    } else if (p[0] == 'g' && is_space(p[1])) {
      export_group();
      if (!shape.empty()) obj.shapes.push_back(shape);
      shape.clear();
      group.clear();
    } else if (p[0] == 'o' && is_space(p[1])) {
      bool any = export_group();
      if (any) obj.shapes.push_back(shape);
      shape.clear();
      group.clear();
    }
  }
  bool any = export_group();
  if (any || !shape.empty()) obj.shapes.push_back(shape);
  return true;
}

} // namespace

// ---------------------------------------------------------------- materials and textures
//
// material_loader.cpp:164-401.  Image decoding goes through an ImageProvider: the reference gets
// its float texels from stb_image's stbi_loadf (material_loader.cpp:97); here the default provider is
// the built-in decoder of image_decode.cpp (same pixels), and a host application may pass its own
// through ptamd_host_scene_load_ex.  A file that cannot be opened or decoded "fails to load", which
// is what the reference does with indoor.mtl's backslash paths on Linux.

namespace {

struct Image { int w = 0, h = 0, nb_chan = 0; std::vector<float> data; }; // nb_chan == 0: not loaded

struct TextureBook {
  const ImageProvider* provider = nullptr;
  std::string folder;
  bool fix_backslashes = false;
  std::map<std::string, Image> loaded;    // MaterialLoader::_loaded_tex
  std::map<std::string, int> packed;      // MaterialLoader::_packed_tex
  HostScene* hs = nullptr;

  int push_texture(int w, int h, int nb_chan, const float* data)
  {
    ptamd_texture_desc td{};
    td.w = w; td.h = h; td.nb_chan = nb_chan;
    td.offset = hs->texels.size();
    hs->texels.insert(hs->texels.end(), data, data + (size_t)w * h * nb_chan);
    hs->textures.push_back(td);
    return (int)hs->textures.size() - 1; // == MaterialLoader::_id++
  }

  int push_unit(const float rgba[4], int nb_chan) { return push_texture(1, 1, nb_chan, rgba); }

  // checkAndupload (material_loader.cpp:83-115)
  const Image& load(const std::string& name)
  {
    auto it = loaded.find(name);
    if (it != loaded.end()) return it->second;
    Image img;
    if (provider && provider->load) {
      std::string rel = name;
      if (fix_backslashes) for (char& c : rel) if (c == '\\') c = '/';
      const std::string full = folder + "/" + rel;
      int32_t w = 0, h = 0, c = 0;
      float* data = nullptr;
      if (provider->load(provider->user, full.c_str(), &w, &h, &c, &data) == 0 && data && w > 0 && h > 0 && c > 0) {
        img.w = w; img.h = h; img.nb_chan = c;
        img.data.assign(data, data + (size_t)w * h * c);
      }
      if (data && provider->release) provider->release(provider->user, data);
    }
    if (img.nb_chan == 0) hs->unloaded_textures.push_back(name);
    return loaded.emplace(name, std::move(img)).first->second;
  }

  // pack(rgb, a) / pack(tex, default) (material_loader.cpp:15-74)
  int push_packed(const Image* rgb, const Image* a, const float default_rgb[3], float default_a)
  {
    const Image& ref = rgb ? *rgb : *a;
    std::vector<float> out((size_t)ref.w * ref.h * 4);
    for (size_t i = 0; i < (size_t)ref.w * ref.h; ++i) {
      out[i * 4 + 0] = rgb ? rgb->data[i * 3 + 0] : default_rgb[0];
      out[i * 4 + 1] = rgb ? rgb->data[i * 3 + 1] : default_rgb[1];
      out[i * 4 + 2] = rgb ? rgb->data[i * 3 + 2] : default_rgb[2];
      out[i * 4 + 3] = a ? a->data[i] : default_a;
    }
    return push_texture(ref.w, ref.h, 4, out.data());
  }

  // getTextureId(tex_rgb, tex_a, default_rgb, default_a) (material_loader.cpp:243-382)
  int diffuse_spec(const std::string& tex_rgb, const std::string& tex_a, const float default_rgb[3], float default_a)
  {
    const float unit[4] = { default_rgb[0], default_rgb[1], default_rgb[2], default_a };
    if (tex_rgb.empty() && tex_a.empty()) return push_unit(unit, 4);          // CASE 1
    const Image* rgb = tex_rgb.empty() ? nullptr : &load(tex_rgb);
    const Image* a = tex_a.empty() ? nullptr : &load(tex_a);
    if (!rgb) return a->nb_chan == 1 ? push_packed(nullptr, a, default_rgb, default_a) : push_unit(unit, 4); // CASE 2
    if (!a) return rgb->nb_chan == 3 ? push_packed(rgb, nullptr, default_rgb, default_a) : push_unit(unit, 4);
    const std::string token = tex_rgb + tex_a;                                // CASE 3
    auto it = packed.find(token);
    if (it != packed.end()) return it->second;
    if (rgb->nb_chan != 3 && a->nb_chan != 1) return push_unit(unit, 4);      // both failed: not cached (ref. :315-322)
    int id;
    if (rgb->nb_chan != 3) id = push_packed(nullptr, a, default_rgb, default_a);
    else if (a->nb_chan != 1) id = push_packed(rgb, nullptr, default_rgb, default_a);
    else if (rgb->w == a->w && rgb->h == a->h) id = push_packed(rgb, a, default_rgb, default_a);
    else {
      // the smaller map is rescaled to the larger one's size with stbir_resize_float
      // (material_loader.cpp:350-375; image_resize.cpp restates it bit for bit)
      const bool a_bigger = (long)a->w * a->h > (long)rgb->w * rgb->h;
      const Image& big = a_bigger ? *a : *rgb;
      const Image& small = a_bigger ? *rgb : *a;
      Image scaled;
      scaled.w = big.w; scaled.h = big.h; scaled.nb_chan = small.nb_chan;
      scaled.data.resize((size_t)big.w * big.h * small.nb_chan);
      resize_float(small.data.data(), small.w, small.h, scaled.data.data(), big.w, big.h, small.nb_chan);
      id = a_bigger ? push_packed(&scaled, a, default_rgb, default_a) : push_packed(rgb, &scaled, default_rgb, default_a);
    }
    packed[token] = id;
    return id;
  }

  // getTextureId(tex_rgb) -> registerOrGet (material_loader.cpp:223-230,384-401)
  int normal_map(const std::string& name)
  {
    if (name.empty()) return -1;
    auto it = packed.find(name);
    if (it != packed.end()) return it->second;
    const Image& img = load(name);
    if (img.nb_chan == 0) return -1;
    // sampleTexture reads three consecutive floats (intersection.cuh:35-45): a map with fewer than
    // three channels would be read out of bounds by the reference; it is dropped instead.
    if (img.nb_chan < 3) return -1;
    const int id = push_texture(img.w, img.h, img.nb_chan, img.data.data());
    packed[name] = id;
    return id;
  }
};

} // namespace

static void build_materials(const ObjData& obj, HostScene& hs, const ImageProvider* provider,
                            const std::string& mtl_folder, bool fix_backslashes)
{
  TextureBook book;
  book.provider = provider;
  book.folder = mtl_folder;
  book.fix_backslashes = fix_backslashes;
  book.hs = &hs;
  for (const MtlEntry& m : obj.materials) {
    ptamd_material mat{};
    const float default_rgb[3] = { m.diffuse[0], m.diffuse[1], m.diffuse[2] };
    const float default_spec = (float)(((double)(m.specular[0] + m.specular[1] + m.specular[2])) / 3.0);
    mat.diffuse_spec_map = book.diffuse_spec(m.diffuse_tex, m.specular_tex, default_rgb, default_spec);
    mat.normal_map = book.normal_map(!m.bump_tex.empty() ? m.bump_tex : m.normal_tex);
    mat.ior = m.ior;
    hs.materials.push_back(mat);
  }
}

// scene.cpp:218-262
static bool build_faces(const ObjData& obj, HostScene& hs, std::string& err)
{
  const int nv = (int)(obj.v.size() / 3), nvn = (int)(obj.vn.size() / 3), nvt = (int)(obj.vt.size() / 2);
  for (const auto& shape : obj.shapes) {
    hs.mesh_sizes.push_back((uint32_t)shape.size());
    for (const auto& t : shape) {
      ptamd_face face{};
      for (int k = 0; k < 3; ++k) {
        const Corner& c = t.c[k];
        if (c.v < 0 || c.v >= nv) { err = "vertex index out of range"; return false; }
        face.vertices[k] = f3(obj.v[3 * c.v], obj.v[3 * c.v + 1], obj.v[3 * c.v + 2]);
        // the reference indexes normals[-3]/texcoords[-2] when a corner has none (undefined
        // behaviour, scene.cpp:241-247); defined here as zeros
        if (c.vn >= 0 && c.vn < nvn)
          face.normals[k] = f3(obj.vn[3 * c.vn], obj.vn[3 * c.vn + 1], obj.vn[3 * c.vn + 2]);
        if (c.vt >= 0 && c.vt < nvt)
          face.texcoords[k] = ptamd_float2{ obj.vt[2 * c.vt], obj.vt[2 * c.vt + 1] };
      }
      if (t.material < 0 || t.material >= (int)hs.materials.size()) {
        // reference: materials.data[-1] (out of bounds).  Refuse instead of reading garbage.
        err = "face without a valid material (usemtl missing or unknown)";
        return false;
      }
      face.material_id = (uint32_t)t.material;
      ptamd_float3 e1 = sub3(face.vertices[1], face.vertices[0]);
      ptamd_float3 e2 = sub3(face.vertices[2], face.vertices[0]);
      float du1 = face.texcoords[1].x - face.texcoords[0].x, dv1 = face.texcoords[1].y - face.texcoords[0].y;
      float du2 = face.texcoords[2].x - face.texcoords[0].x, dv2 = face.texcoords[2].y - face.texcoords[0].y;
      float f = 1.0f / (du1 * dv2 - du2 * dv1);
      face.tangent.x = f * (dv2 * e1.x - dv1 * e2.x);
      face.tangent.y = f * (dv2 * e1.y - dv1 * e2.y);
      face.tangent.z = f * (dv2 * e1.z - dv1 * e2.z);
      hs.faces.push_back(face);
    }
  }
  return true;
}

static int builtin_load(void*, const char* path, int32_t* w, int32_t* h, int32_t* nb_chan, float** data)
{
  std::string err; int iw = 0, ih = 0, ic = 0;
  if (!load_image_float(path, &iw, &ih, &ic, data, err)) return 1;
  *w = iw; *h = ih; *nb_chan = ic;
  return 0;
}
static void builtin_release(void*, float* data) { std::free(data); }

const ImageProvider* builtin_image_provider()
{
  static const ImageProvider prov{ builtin_load, builtin_release, nullptr };
  return &prov;
}

int load_host_scene(const char* scene_path, uint32_t flags, const ImageProvider* provider, HostScene*& out)
{
  if (!provider && !(flags & PTAMD_LOAD_NO_IMAGES)) provider = builtin_image_provider();
  std::ifstream file(scene_path);
  if (!file.is_open()) {
    set_error(std::string("cannot open scene file '") + scene_path + "'");
    return PTAMD_ERR_IO;
  }
  HostScene* hs = new HostScene();
  // default camera (scene.cpp:98-102); position/focus/aperture are indeterminate in the
  // reference when no camera line exists — defined here
  ptamd_camera& cam = hs->camera;
  std::memset(&cam, 0, sizeof cam);
  cam.u = f3(1.0f, 0.0f, 0.0f);
  cam.v = f3(0.0f, -1.0f, 0.0f);
  cam.fov_x = (float)((90.0 * kPi) / 180.0);
  cam.dir = cross3(cam.u, cam.v);
  cam.focus_dist = 2.0f;
  cam.aperture = 0.125f;
  cam.speed = 1.4f;

  std::string objfile, line, token;
  while (std::getline(file, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::stringstream iss(line);
    token.clear();
    iss >> token;
    if (token == "p_light") {
      ptamd_light l{};
      if (!read_vec3(l.vec, iss)) continue;
      if (!read_vec3(l.color, iss)) continue;
      if (at_end(iss)) continue;
      iss >> l.emission;
      if (at_end(iss)) continue;
      iss >> l.radius;
      hs->lights.push_back(l);
    } else if (token == "scene") {
      if (at_end(iss)) continue;
      iss >> objfile;
    } else if (token == "camera") {
      ptamd_camera c = cam;
      if (read_camera(c, iss)) cam = c;
      else cam = c; // the reference parses in place: a partial parse leaves partial values
    } else if (token == "cubemap") {
      if (!at_end(iss)) iss >> hs->cubemap;
    }
  }

  // scene.cpp:329-339
  std::string path(scene_path), base_dir, mtl_dir, full_obj;
  std::string::size_type pos = path.find_last_of('/');
  if (pos != std::string::npos) {
    base_dir = path.substr(0, pos) + "/";
    mtl_dir = base_dir;
    full_obj = base_dir + objfile;
  }
  pos = objfile.find_last_of('/');
  if (pos != std::string::npos) mtl_dir = base_dir + "/" + objfile.substr(0, pos) + "/";

  ObjData obj;
  std::string err;
  if (!load_obj(full_obj, mtl_dir, obj, err)) {
    set_error("scene '" + path + "': " + err);
    delete hs;
    return PTAMD_ERR_IO;
  }
  build_materials(obj, *hs, provider, mtl_dir, (flags & 1u) != 0);
  if (!build_faces(obj, *hs, err)) {
    set_error("scene '" + path + "': " + err);
    delete hs;
    return PTAMD_ERR_IO;
  }
  out = hs;
  return PTAMD_OK;
}

} // namespace ptamd

// ------------------------------------------------------------------------ C-ABI

extern "C" {

int ptamd_host_scene_load(const char* scene_path, uint32_t flags, ptamd_host_scene** out)
{
  if (!scene_path || !out) { ptamd::set_error("ptamd_host_scene_load: null argument"); return PTAMD_ERR_ARG; }
  ptamd::HostScene* hs = nullptr;
  int rc = ptamd::load_host_scene(scene_path, flags, nullptr, hs);
  if (rc != PTAMD_OK) return rc;
  *out = reinterpret_cast<ptamd_host_scene*>(hs);
  return PTAMD_OK;
}

int ptamd_host_scene_load_ex(const char* scene_path, uint32_t flags, ptamd_image_load_fn load, ptamd_image_free_fn release,
                             void* user, ptamd_host_scene** out)
{
  if (!scene_path || !out) { ptamd::set_error("ptamd_host_scene_load_ex: null argument"); return PTAMD_ERR_ARG; }
  ptamd::ImageProvider prov{ load, release, user };
  ptamd::HostScene* hs = nullptr;
  int rc = ptamd::load_host_scene(scene_path, flags, load ? &prov : nullptr, hs);
  if (rc != PTAMD_OK) return rc;
  *out = reinterpret_cast<ptamd_host_scene*>(hs);
  return PTAMD_OK;
}

uint32_t ptamd_host_scene_unloaded_count(const ptamd_host_scene* s)
{
  return s ? (uint32_t)reinterpret_cast<const ptamd::HostScene*>(s)->unloaded_textures.size() : 0u;
}

const char* ptamd_host_scene_unloaded_name(const ptamd_host_scene* s, uint32_t i)
{
  if (!s) return "";
  const auto& v = reinterpret_cast<const ptamd::HostScene*>(s)->unloaded_textures;
  return i < v.size() ? v[i].c_str() : "";
}

void ptamd_host_scene_free(ptamd_host_scene* s) { delete reinterpret_cast<ptamd::HostScene*>(s); }

int ptamd_host_scene_desc(const ptamd_host_scene* s, ptamd_scene_desc* out)
{
  if (!s || !out) { ptamd::set_error("ptamd_host_scene_desc: null argument"); return PTAMD_ERR_ARG; }
  const ptamd::HostScene* hs = reinterpret_cast<const ptamd::HostScene*>(s);
  std::memset(out, 0, sizeof *out);
  out->faces = hs->faces.data();           out->n_faces = (uint32_t)hs->faces.size();
  out->mesh_sizes = hs->mesh_sizes.data(); out->n_meshes = (uint32_t)hs->mesh_sizes.size();
  out->materials = hs->materials.data();   out->n_materials = (uint32_t)hs->materials.size();
  out->lights = hs->lights.data();         out->n_lights = (uint32_t)hs->lights.size();
  out->textures = hs->textures.data();     out->n_textures = (uint32_t)hs->textures.size();
  out->texels = hs->texels.data();         out->n_texel_floats = hs->texels.size();
  return PTAMD_OK;
}

int ptamd_host_scene_camera(const ptamd_host_scene* s, ptamd_camera* out)
{
  if (!s || !out) { ptamd::set_error("ptamd_host_scene_camera: null argument"); return PTAMD_ERR_ARG; }
  *out = reinterpret_cast<const ptamd::HostScene*>(s)->camera;
  return PTAMD_OK;
}

const char* ptamd_host_scene_cubemap(const ptamd_host_scene* s)
{
  if (!s) return "";
  return reinterpret_cast<const ptamd::HostScene*>(s)->cubemap.c_str();
}

// gpu_processor.cpp:37-57
int ptamd_cubemap_from_color(uint32_t rgb, float out[24])
{
  if (!out) { ptamd::set_error("ptamd_cubemap_from_color: null argument"); return PTAMD_ERR_ARG; }
  float r = (float)((rgb >> 16) & 0xFF) / 255.0f;
  float g = (float)((rgb >> 8) & 0xFF) / 255.0f;
  float b = (float)(rgb & 0xFF) / 255.0f;
  for (int f = 0; f < 6; ++f) {
    out[f * 4 + 0] = r; out[f * 4 + 1] = g; out[f * 4 + 2] = b; out[f * 4 + 3] = 0.0f;
  }
  return PTAMD_OK;
}

// gpu_processor.cpp:101-127 + texture_utils.cpp:5-52: a 4x3 cube cross; faces cut as
// +x=(col2,row1) -x=(col0,row1) +y=(col1,row0) -y=(col1,row2) +z=(col1,row1) -z=(col3,row1)
int ptamd_cubemap_from_cross(const float* cross, uint32_t width, uint32_t height, uint32_t nb_chan,
                             float* out, uint32_t* out_size)
{
  if (!cross || !out || !out_size || nb_chan < 3) {
    ptamd::set_error("ptamd_cubemap_from_cross: bad argument");
    return PTAMD_ERR_ARG;
  }
  uint32_t size = width / 4;
  if (size == 0 || size != height / 3) { ptamd::set_error("cubemap: width and height are not the same"); return PTAMD_ERR_ARG; }
  if (size & (size - 1)) { ptamd::set_error("cubemap: size should be a power of 2"); return PTAMD_ERR_ARG; }
  static const uint32_t col[6] = { 2, 0, 1, 1, 1, 3 };
  static const uint32_t row[6] = { 1, 1, 0, 2, 1, 1 };
  float* dst = out;
  for (int f = 0; f < 6; ++f)
    for (uint32_t y = 0; y < size; ++y)
      for (uint32_t x = 0; x < size; ++x) {
        const float* src = cross + ((size_t)(y + row[f] * size) * width + (x + col[f] * size)) * nb_chan;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = 0.0f;
        dst += 4;
      }
  *out_size = size;
  return PTAMD_OK;
}

} // extern "C"
