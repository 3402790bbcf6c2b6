// ImageTexture on the host: what load_image (renderprocess.rs:532-561) hands to MIPMap::create, and MIPMap::create
// itself (mipmap.rs:270-382) over the BlockedArray of memory.rs:24-98.
//
// The reference decodes through the `image` crate (0.23.14, Cargo.lock) and `.into_rgb8()`. That crate is not in
// /root/reference; restated here is the PNG path for 8-bit-or-less samples (png 0.16.8 with Transformations::EXPAND:
// palette -> RGB, grey bit depths 1/2/4 scaled to 8 bits, alpha dropped by into_rgb8). 16-bit PNGs, interlaced PNGs
// and the crate's other formats are reported as "unsupported" (RRT_EUNSUP where a material uses the texture):
// their conversion rules cannot be checked against anything in this image.
#include <zlib.h>

#include <cmath>
#include <cstring>
#include <fstream>

#include "scene.hpp"

namespace rrt {

namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}
}  // namespace

// returns 0 = decoded, 1 = not a decodable image (the reference's `load_image` Err: texture not registered),
// 2 = an image the `image` crate decodes but this restatement does not
int decode_png_rgb8(const std::string& path, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgb, std::string* why) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { *why = "cannot open"; return 1; }
  // ImageReader::open takes the format from the path's extension alone (image 0.23 io/reader.rs, ImageFormat::from_path);
  // no / unknown extension: decode() fails with an unsupported-format error and the texture is skipped
  std::string ext;
  {
    const size_t dot = path.find_last_of('.'), slash = path.find_last_of('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ext = path.substr(dot + 1);
    for (auto& ch : ext) ch = (char)std::tolower((unsigned char)ch);
  }
  if (ext != "png") {
    static const char* other[] = {"jpg", "jpeg", "gif", "webp", "tif", "tiff", "tga", "dds", "bmp", "ico", "hdr", "pbm", "pam", "ppm", "pgm", "ff", "avif"};
    for (const char* o : other) if (ext == o) { *why = std::string("a .") + o + " file (only the PNG decoder of the image crate is restated)"; return 2; }
    *why = "no image format for extension '" + ext + "'";
    return 1;
  }
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) { *why = "bad PNG signature"; return 1; }
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> plte, idat;
  bool seen_ihdr = false, seen_iend = false;
  size_t pos = 8;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    if (pos + 12 + (size_t)len > buf.size()) { *why = "truncated chunk"; return 1; }
    const uint8_t* type = &buf[pos + 4];
    const uint8_t* data = &buf[pos + 8];
    const uint32_t crc = be32(&buf[pos + 8 + len]);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) { *why = "chunk CRC mismatch"; return 1; }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) { *why = "bad IHDR"; return 1; }
      width = be32(data); height = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) { *why = "bad IHDR methods"; return 1; }
      seen_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) { seen_iend = true; break; }
    pos += 12 + (size_t)len;
  }
  if (!seen_ihdr || !seen_iend || width == 0 || height == 0) { *why = "missing IHDR / IEND"; return 1; }
  // IHDR is file content: 2^32 - 1 squared wraps the buffer sizes below and a crafted file would index past them (or ask for
  // terabytes). 65 536 texels per side is far beyond any texture the scene format is used with.
  if (width > 65536u || height > 65536u) { *why = "image larger than 65536 x 65536"; return 2; }
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: *why = "bad colour type"; return 1;
  }
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!depth_ok) { *why = "bad bit depth"; return 1; }
  if (depth == 16) { *why = "16-bit PNG (the image crate's 16 -> 8 bit rule is not restated)"; return 2; }
  if (interlace != 0) { *why = "interlaced PNG (Adam7 is not restated)"; return 2; }
  if (ctype == 3 && plte.empty()) { *why = "palette image without PLTE"; return 1; }
  const size_t bpp_bits = (size_t)channels * (size_t)depth, stride = ((size_t)width * bpp_bits + 7) / 8, fbpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
  std::vector<uint8_t> raw((stride + 1) * (size_t)height);
  uLongf raw_len = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || raw_len != raw.size()) { *why = "bad zlib stream"; return 1; }
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  rgb->assign((size_t)width * height * 3, 0);
  for (uint32_t y = 0; y < height; y++) {
    const uint8_t* line = &raw[(stride + 1) * (size_t)y];
    const int ft = line[0];
    if (ft > 4) { *why = "bad filter type"; return 1; }
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= fbpp ? cur[i - fbpp] : 0, b = prev[i], c = i >= fbpp ? prev[i - fbpp] : 0;
      int v = line[1 + i];
      if (ft == 1) v += a; else if (ft == 2) v += b; else if (ft == 3) v += (a + b) / 2; else if (ft == 4) v += paeth(a, b, c);
      cur[i] = (uint8_t)v;
    }
    for (uint32_t x = 0; x < width; x++) {
      uint8_t* o = &(*rgb)[3 * ((size_t)y * width + x)];
      auto sample = [&](size_t k) -> int {   // k-th sample of the line
        if (depth == 8) return cur[k];
        const size_t bit = k * (size_t)depth;
        return (cur[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1 << depth) - 1);
      };
      if (ctype == 0) {   // EXPAND scales grey samples to 8 bits by bit replication: 1 bit x255, 2 bits x85, 4 bits x17
        int g = sample(x);
        if (depth < 8) g = g * (255 / ((1 << depth) - 1));
        o[0] = o[1] = o[2] = (uint8_t)g;
      } else if (ctype == 3) {
        const size_t idx = (size_t)sample(x);
        if (3 * idx + 2 >= plte.size()) { *why = "palette index out of range"; return 1; }
        o[0] = plte[3 * idx]; o[1] = plte[3 * idx + 1]; o[2] = plte[3 * idx + 2];
      } else if (ctype == 4) {
        o[0] = o[1] = o[2] = cur[2 * (size_t)x];
      } else {
        const uint8_t* px = &cur[(size_t)channels * x];
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
      }
    }
    prev.swap(cur);
  }
  *w = width; *h = height;
  return 0;
}

namespace {

// BlockedArray index memory.rs:76-85 (see include/rrt.h)
inline size_t ba_index(size_t u_blocks, size_t u, size_t v) { return 16 * (u_blocks * (v & 3) + (u & 3)) + 4 * (v >> 2) + (u >> 2); }
inline size_t ba_round_up(size_t x) { return (x + 3) & ~(size_t)3; }

struct Level {
  size_t u_res = 0, v_res = 0, u_blocks = 0;
  std::vector<double> data;   // RGB triples
  void init(size_t u, size_t v) { u_res = u; v_res = v; u_blocks = ba_round_up(u) >> 2; data.assign(3 * ba_round_up(u) * ba_round_up(v), 0.0); }
  double* at(size_t u, size_t v) {
    const size_t i = ba_index(u_blocks, u, v);
    // images narrower than 16 texels (after the power-of-two resampling) index past the vector in BlockedArray::new
    if (3 * i + 2 >= data.size()) throw Panic("memory.rs:84/96 BlockedArray index out of bounds (" + std::to_string(u_res) + " x " + std::to_string(v_res) + " level)");
    return &data[3 * i];
  }
};

double lanczos(double x, double tau) {   // texture/mod.rs:191-204
  x = std::fabs(x);
  if (x < 1e-5) return 1.0;
  if (x > 1.0) return 0.0;
  x *= M_PI;
  const double s = std::sin(x * tau) / (x * tau);
  return s * (std::sin(x) / x);
}
size_t f2usize(double v) {   // Rust `as usize`: saturating, NaN -> 0
  if (!(v > 0.0)) return 0;
  if (v >= 18446744073709551615.0) return (size_t)-1;
  return (size_t)v;
}
struct ResampleWeight { size_t first_texel; double weight[4]; };
std::vector<ResampleWeight> resample_weights(size_t old_res, size_t new_res) {   // mipmap.rs:26-47
  std::vector<ResampleWeight> wt(new_res);
  const double filter_width = 2.0;
  for (size_t i = 0; i < new_res; i++) {
    const double center = ((double)i + 0.5) * (double)old_res / (double)new_res;
    wt[i].first_texel = f2usize(std::floor(center - filter_width + 0.5));
    for (int j = 0; j < 4; j++) {
      const double pos = (double)(wt[i].first_texel + (size_t)j) + 0.5;
      wt[i].weight[j] = lanczos((pos - center) / filter_width, 2.0);
    }
    const double inv = 1.0 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
    for (int j = 0; j < 4; j++) wt[i].weight[j] *= inv;
  }
  return wt;
}
size_t mod_usize(size_t a, size_t b) { return a - (a / b) * b; }
size_t clamp_usize(size_t v, size_t lo, size_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
size_t round_up_pow2(size_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }   // misc.rs:318-330 (no >> 32)

// MIPMap::texel mipmap.rs:107-131
void texel(Level& l, int wrap, size_t s, size_t t, double out[3]) {
  size_t ts = 0, tt = 0;
  if (wrap == RRT_WRAP_REPEAT) { ts = mod_usize(s, l.u_res); tt = mod_usize(t, l.v_res); }
  else if (wrap == RRT_WRAP_BLACK) { if (s >= l.u_res || t >= l.v_res) { out[0] = out[1] = out[2] = 0.0; return; } }   // (in range: texel (0, 0), :119-123)
  else { ts = clamp_usize(s, 0, l.u_res); tt = clamp_usize(t, 0, l.v_res); }
  const double* p = l.at(ts, tt);
  out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}

}  // namespace

// load_image's pixel preparation + MIPMap::create; appends one rrt_image and its texels to the scene
int build_mipmap(SceneData& s, uint32_t w, uint32_t h, const std::vector<uint8_t>& rgb8, bool do_trilinear, double max_aniso, int wrap) {
  const size_t rx = w, ry = h;
  std::vector<double> img(3 * rx * ry);
  for (size_t i = 0; i < rx * ry * 3; i++) img[i] = (double)rgb8[i] / 255.0;
  for (size_t y = 0; y < ry / 2; y++)   // vertical flip renderprocess.rs:546-552
    for (size_t x = 0; x < rx; x++)
      for (int c = 0; c < 3; c++) std::swap(img[3 * (y * rx + x) + c], img[3 * ((ry - 1 - y) * rx + x) + c]);
  size_t res[2] = {rx, ry};
  std::vector<double> resampled;
  auto pow2 = [](size_t v) { return v != 0 && (v & (v - 1)) == 0; };
  if (!pow2(rx) || !pow2(ry)) {   // :279-331
    const size_t px = round_up_pow2(rx), py = round_up_pow2(ry);
    const auto sw = resample_weights(rx, px);
    resampled.assign(3 * px * py, 0.0);
    for (size_t t = 0; t < ry; t++)
      for (size_t x = 0; x < px; x++) {
        double* dst = &resampled[3 * (t * px + x)];
        dst[0] = dst[1] = dst[2] = 0.0;
        for (int j = 0; j < 4; j++) {
          size_t os = sw[x].first_texel + (size_t)j;
          if (wrap == RRT_WRAP_REPEAT) os = mod_usize(os, rx);
          else if (wrap == RRT_WRAP_CLAMP) os = clamp_usize(os, 0, rx - 1);
          if (os < rx) for (int c = 0; c < 3; c++) dst[c] += img[3 * (t * rx + os) + c] * sw[x].weight[j];
        }
      }
    const auto tw = resample_weights(ry, py);
    std::vector<double> work(3 * py);
    for (size_t x = 0; x < px; x++) {
      std::fill(work.begin(), work.end(), 0.0);
      for (size_t t = 0; t < py; t++)
        for (int j = 0; j < 4; j++) {
          size_t off = tw[t].first_texel + (size_t)j;
          if (wrap == RRT_WRAP_REPEAT) off = mod_usize(off, ry);
          else if (wrap == RRT_WRAP_CLAMP) off = clamp_usize(off, 0, ry - 1);
          if (off < ry) for (int c = 0; c < 3; c++) work[3 * t + c] += resampled[3 * (off * px + x) + c] * tw[t].weight[j];
        }
      for (size_t t = 0; t < py; t++)
        for (int c = 0; c < 3; c++) { const double v = work[3 * t + c]; resampled[3 * (t * px + x) + c] = v < 0.0 ? 0.0 : v; }   // clamp(0, inf)
    }
    res[0] = px; res[1] = py;
  }
  const std::vector<double>& base = resampled.empty() ? img : resampled;
  const size_t n_levels = 1 + f2usize(std::log2((double)std::max(res[0], res[1])));
  std::vector<Level> pyr(1);
  pyr[0].init(res[0], res[1]);
  for (size_t u = 0; u < res[0]; u++)      // BlockedArray::new memory.rs:41-47: u outer, v inner
    for (size_t v = 0; v < res[1]; v++)
      for (int c = 0; c < 3; c++) pyr[0].at(u, v)[c] = base[3 * (v * res[0] + u) + c];
  for (size_t i = 1; i < n_levels; i++) {   // :358-379
    const size_t sr = std::max<size_t>(pyr[i - 1].u_res / 2, 1), tr = std::max<size_t>(pyr[i - 1].v_res / 2, 1);
    if (std::min(sr, tr) < 64) break;
    Level l;
    l.init(sr, tr);
    for (size_t t = 0; t < tr; t++)
      for (size_t x = 0; x < sr; x++) {
        double a[3], b[3], c[3], d[3];
        texel(pyr[i - 1], wrap, 2 * x, 2 * t, a); texel(pyr[i - 1], wrap, 2 * x + 1, 2 * t, b);
        texel(pyr[i - 1], wrap, 2 * x, 2 * t + 1, c); texel(pyr[i - 1], wrap, 2 * x + 1, 2 * t + 1, d);
        for (int k = 0; k < 3; k++) l.at(x, t)[k] = (((a[k] + b[k]) + c[k]) + d[k]) * 0.25;
      }
    pyr.push_back(std::move(l));
  }
  if (pyr.size() > 16) throw Unsupported("ImageTexture: more than 16 pyramid levels");
  rrt_image im{};
  im.do_trilinear = do_trilinear ? 1 : 0; im.wrap = wrap; im.max_aniso = max_aniso; im.n_levels = (int32_t)pyr.size();
  for (size_t i = 0; i < pyr.size(); i++) {
    rrt_image_level& L = im.levels[i];
    L.u_res = (uint32_t)pyr[i].u_res; L.v_res = (uint32_t)pyr[i].v_res; L.u_blocks = (uint32_t)pyr[i].u_blocks;
    L.offset = s.image_texels.size() / 3; L.n = pyr[i].data.size() / 3;
    s.image_texels.insert(s.image_texels.end(), pyr[i].data.begin(), pyr[i].data.end());
  }
  s.images.push_back(im);
  return (int)s.images.size() - 1;
}

}  // namespace rrt
