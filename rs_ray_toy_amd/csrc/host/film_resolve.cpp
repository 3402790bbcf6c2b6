// Film resolve + PNG output on the host: Film::write_image (/root/reference/src/film.rs:323-366) and
// renderprocess::write_image (:1501-1530): XYZ -> RGB, divide by filter_weight_sum, clamp at 0, scale,
// sRGB gamma (misc.rs:46-52), `clamp(255 v + 0.5, 0, 255) as u8`, alpha 255. The PNG container is written
// with zlib (the reference delegates to the `image` crate; only the decoded pixels are comparable).
#include <zlib.h>

#include <cstdio>
#include <vector>

#include "scene.hpp"

namespace rrt {
static inline double gamma_correct(double v) {  // misc.rs:46-52
  if (v <= 0.0031308) return 12.92 * v;
  return 1.055 * std::pow(v, 1.0 / 2.4) - 0.055;
}
static inline uint8_t quantise(double v) {  // clamp_t(255*g+0.5, 0, 255) as u8 (NaN -> 0)
  double x = 255.0 * gamma_correct(v) + 0.5;
  if (x < 0.0) x = 0.0; else if (x > 255.0) x = 255.0;
  if (!(x == x)) return 0;
  return (uint8_t)x;
}
template <typename T>
static void resolve(const T* film, int w, int h, double scale, uint8_t* rgba) {
  for (size_t i = 0; i < (size_t)w * h; i++) {
    double xyz[3] = {(double)film[4 * i], (double)film[4 * i + 1], (double)film[4 * i + 2]};
    double wsum = (double)film[4 * i + 3];
    double rgb[3];
    // xyz_to_rgb spectrum.rs:2075-2082
    rgb[0] = 3.240479 * xyz[0] - 1.537150 * xyz[1] - 0.498535 * xyz[2];
    rgb[1] = -0.969256 * xyz[0] + 1.875991 * xyz[1] + 0.041556 * xyz[2];
    rgb[2] = 0.055648 * xyz[0] - 0.204043 * xyz[1] + 1.057311 * xyz[2];
    if (wsum != 0.0) {
      double inv = 1.0 / wsum;
      for (int k = 0; k < 3; k++) rgb[k] = std::fmax(0.0, rgb[k] * inv);
    }
    for (int k = 0; k < 3; k++) rgba[4 * i + k] = quantise(rgb[k] * scale);  // splat term is 0 on this path
    rgba[4 * i + 3] = 255;
  }
}
}  // namespace rrt

extern "C" {

int rrt_resolve_rgba8(const void* film_xyzw, int precision, int w, int h, double scale, uint8_t* rgba) {
  if (!film_xyzw || !rgba || w <= 0 || h <= 0) { rrt::set_last_error("rrt_resolve_rgba8: bad argument"); return RRT_EINVAL; }
  if (precision == RRT_F64) rrt::resolve((const double*)film_xyzw, w, h, scale, rgba);
  else if (precision == RRT_F32) rrt::resolve((const float*)film_xyzw, w, h, scale, rgba);
  else { rrt::set_last_error("rrt_resolve_rgba8: bad precision"); return RRT_EINVAL; }
  return RRT_OK;
}

int rrt_write_png(const char* path, const uint8_t* rgba, int w, int h) {
  if (!path || !rgba || w <= 0 || h <= 0) { rrt::set_last_error("rrt_write_png: bad argument"); return RRT_EINVAL; }
  std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 4));
  for (int y = 0; y < h; y++) {
    raw[(size_t)y * (1 + (size_t)w * 4)] = 0;  // filter: none
    memcpy(&raw[(size_t)y * (1 + (size_t)w * 4) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
  }
  uLongf zlen = compressBound(raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), raw.size(), 6) != Z_OK) { rrt::set_last_error("png: deflate failed"); return RRT_EIO; }
  FILE* f = fopen(path, "wb");
  if (!f) { rrt::set_last_error(std::string("png: cannot open ") + path); return RRT_EIO; }
  auto be32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
  auto chunk = [&](const char* type, const uint8_t* data, size_t len) {
    uint8_t hdr[8];
    be32(hdr, (uint32_t)len);
    memcpy(hdr + 4, type, 4);
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(data, 1, len, f);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (len) crc = crc32(crc, data, (uInt)len);
    uint8_t c[4];
    be32(c, (uint32_t)crc);
    fwrite(c, 1, 4, f);
  };
  const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, f);
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)w); be32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", z.data(), zlen);
  chunk("IEND", nullptr, 0);
  bool ok = fclose(f) == 0;
  if (!ok) { rrt::set_last_error("png: write failed"); return RRT_EIO; }
  return RRT_OK;
}

}  // extern "C"
