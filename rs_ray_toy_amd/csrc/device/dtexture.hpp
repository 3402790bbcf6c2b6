// Texture graph evaluation and ray differentials on the device (reference: texture/*.rs, interaction.rs:223-284,
// integrator/mod.rs:183-201 / 238-292). Only kernels instantiated for scenes whose materials evaluate a texture
// include any of this in their code path (template flag TEX); every other scene keeps its constant-folded materials.
#pragma once
#include "dmath.hpp"

namespace rrtd {

constexpr int kTexDepth = 5;   // deepest texture graph evaluated (root = level 1); deeper graphs are RRT_EUNSUP

// the differential half of RayDifferential (geometry.rs:80-92)
template <typename R>
struct DiffRay {
  bool has = false;
  V3<R> rxo, rxd, ryo, ryd;
};

// what Texture::evaluate reads of a SurfaceInteraction
template <typename R>
struct TexCtx {
  V3<R> p, dpdx, dpdy;
  // the hit point in double (fp32 mode: p + the low word of the double-float spawn point): the 3D textures work on it, because
  // Material::bump differentiates them over a step of 5e-4 of a triangle edge - below fp32 resolution at scene coordinates
  V3<double> pd;
  R u = 0, v = 0, dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
  uint32_t err = 0;   // set where the reference's MIPMap lookup would index out of bounds (-> RRT_EPANIC)
};

// solve_linear_system_2x2 transform.rs:153-164
template <typename R>
RRT_DEV bool solve_2x2(R a00, R a01, R a10, R a11, R b0, R b1, R* x0, R* x1) {
  const R det = a00 * a11 - a01 * a10;
  if (rabs(det) < R(1e-10)) return false;
  *x0 = (a11 * b0 - a01 * b1) / det;
  *x1 = (a00 * b1 - a10 * b0) / det;
  if (*x0 != *x0 || *x1 != *x1) return false;
  return true;
}
template <typename R> RRT_DEV R v3_at(V3<R> v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// SurfaceInteraction::compute_differentials interaction.rs:223-284 (n = geometric normal)
template <typename R>
RRT_DEV void compute_differentials(TexCtx<R>* c, V3<R> n, V3<R> dpdu, V3<R> dpdv, const DiffRay<R>& rd) {
  c->dudx = c->dvdx = c->dudy = c->dvdy = R(0);
  c->dpdx = c->dpdy = V3<R>();
  if (!rd.has) return;
  const R d = dot(n, c->p);
  const R tx = -(dot(n, rd.rxo) - d) / dot(n, rd.rxd);
  if (isinf(tx) || tx != tx) return;
  const V3<R> px = rd.rxo + rd.rxd * tx;
  const R ty = -(dot(n, rd.ryd) - d) / dot(n, rd.ryd);   // :234 reads ry_direction where pbrt reads ry_origin
  if (isinf(ty) || ty != ty) return;
  const V3<R> py = rd.ryo + rd.ryd * ty;
  c->dpdx = px - c->p;
  c->dpdy = py - c->p;
  int d0, d1;
  if (rabs(n.x) > rabs(n.y) && rabs(n.x) > rabs(n.z)) { d0 = 1; d1 = 2; }
  else if (rabs(n.y) > rabs(n.z)) { d0 = 0; d1 = 2; }
  else { d0 = 0; d1 = 1; }
  const R a00 = v3_at(dpdu, d0), a01 = v3_at(dpdv, d0), a10 = v3_at(dpdu, d1), a11 = v3_at(dpdv, d1);
  if (!solve_2x2(a00, a01, a10, a11, v3_at(px, d0) - v3_at(c->p, d0), v3_at(px, d1) - v3_at(c->p, d1), &c->dudx, &c->dvdx)) { c->dudx = R(0); c->dvdx = R(0); }
  if (!solve_2x2(a00, a01, a10, a11, v3_at(py, d0) - v3_at(c->p, d0), v3_at(py, d1) - v3_at(c->p, d1), &c->dudy, &c->dvdy)) { c->dudy = R(0); c->dvdy = R(0); }
}

// ---- Perlin noise texture/mod.rs:13-185 (Ken Perlin's published permutation) ------------------------------------
static __device__ const uint8_t kNoisePerm[256] = {
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148,
    247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175,
    74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54,
    65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64,
    52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213,
    119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104,
    218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157,
    184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180};
RRT_DEV int noise_perm(int i) { return (int)kNoisePerm[i & 255]; }   // the reference doubles the table instead of masking
RRT_DEV double noise_grad(int x, int y, int z, double dx, double dy, double dz) {
  const int h = noise_perm(noise_perm(noise_perm(x) + y) + z) & 15;
  const double u = (h < 8 || h == 12 || h == 13) ? dx : dy;
  const double v = (h < 4 || h == 12 || h == 13) ? dy : dz;
  return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
RRT_DEV double noise_weight(double t) { const double t3 = t * t * t, t4 = t3 * t; return 6.0 * t4 * t - 15.0 * t4 + 10.0 * t3; }
template <typename R> RRT_DEV R lerp_r(R t, R a, R b) { return a * (R(1) - t) + b * t; }   // misc.rs:223-228
template <typename R>
RRT_DEV int f2i_sat(R v) {   // Rust `as i32`: saturating, NaN -> 0
  if (v != v) return 0;
  if (v >= R(2147483647.0)) return 2147483647;
  if (v <= R(-2147483648.0)) return -2147483647 - 1;
  return (int)v;
}
// the noise functions run in double in both device modes (see TexCtx::pd); the f64 mode is unchanged by that
RRT_DEV double noise_flt(V3<double> q) {
  int ix = f2i_sat(floor(q.x)), iy = f2i_sat(floor(q.y)), iz = f2i_sat(floor(q.z));
  const double dx = q.x - (double)ix, dy = q.y - (double)iy, dz = q.z - (double)iz;
  ix &= 255; iy &= 255; iz &= 255;
  const double w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0, dy, dz);
  const double w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0, dy - 1.0, dz);
  const double w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0, dy, dz - 1.0);
  const double w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0, dz - 1.0), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0, dy - 1.0, dz - 1.0);
  const double wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
  const double x00 = lerp_r(wx, w000, w100), x10 = lerp_r(wx, w010, w110), x01 = lerp_r(wx, w001, w101), x11 = lerp_r(wx, w011, w111);
  const double y0 = lerp_r(wy, x00, x10), y1 = lerp_r(wy, x01, x11);
  return lerp_r(wz, y0, y1);
}
template <typename R> RRT_DEV R smooth_step(R mn, R mx, R v) { const R t = clampr((v - mn) / (mx - mn), R(0), R(1)); return t * t * (R(-2) * t + R(3)); }
// fbm :138-153 (turb = false) / turbulence :155-185 (turb = true)
RRT_DEV double noise_sum(V3<double> p, V3<double> dpdx, V3<double> dpdy, double omega, int max_octaves, bool turb) {
  const double l2 = rmax(len2(dpdx), len2(dpdy));
  const double n = clampr(-1.0 - 0.5 * log2(l2), 0.0, (double)max_octaves);
  const int n_int = f2i_sat(floor(n));
  double sum = 0.0, lambda = 1.0, o = 1.0;
  for (int i = 0; i < n_int; i++) {
    const double nz = noise_flt(p * lambda);
    sum += o * (turb ? rabs(nz) : nz);
    lambda *= 1.99; o *= omega;
  }
  const double n_partial = n - (double)n_int;
  const double nz = noise_flt(p * lambda);
  if (turb) {
    sum += o * lerp_r(smooth_step(0.3, 0.7, n_partial), 0.2, rabs(nz));
    for (int i = n_int; i < max_octaves; i++) { sum += o * 0.2; o *= omega; }
  } else {
    sum += o * smooth_step(0.3, 0.7, n_partial) * nz;
  }
  return sum;
}
// world_to_texture in double (the matrix itself is stored in the mode's precision)
template <typename R> RRT_DEV V3<double> aff_pt_d(const R* m, V3<double> p) {
  return V3<double>((double)m[0] * p.x + (double)m[1] * p.y + (double)m[2] * p.z + (double)m[3],
                    (double)m[4] * p.x + (double)m[5] * p.y + (double)m[6] * p.z + (double)m[7],
                    (double)m[8] * p.x + (double)m[9] * p.y + (double)m[10] * p.z + (double)m[11]);
}
template <typename R> RRT_DEV V3<double> aff_vec_d(const R* m, V3<R> v) {
  return V3<double>((double)m[0] * (double)v.x + (double)m[1] * (double)v.y + (double)m[2] * (double)v.z,
                    (double)m[4] * (double)v.x + (double)m[5] * (double)v.y + (double)m[6] * (double)v.z,
                    (double)m[8] * (double)v.x + (double)m[9] * (double)v.y + (double)m[10] * (double)v.z);
}

// ---- TextureMapping2D::map texture/mod.rs:205-352 ----------------------------------------------------------------
template <typename R>
RRT_DEV void map_round(const TexDev<R>& t, V3<R> p, R* s_, R* t_) {   // SphericalMapping2D::sphere / CylindricalMapping2D::cylinder
  const V3<R> v = vnormalize(aff_pt(t.w2t, p));
  if (t.mapping == 1) {
    const R theta = acos(clampr(v.z, R(-1), R(1)));
    R phi = atan2(v.y, v.x);
    if (phi < R(0)) phi += R(2) * R(RRT_PI);
    *s_ = theta / R(RRT_PI); *t_ = phi / (R(RRT_PI) * R(2));
  } else {
    *s_ = (R(RRT_PI) + atan2(v.y, v.x)) / (R(2) * R(RRT_PI)); *t_ = v.z;
  }
}
template <typename R>
RRT_DEV void tex_map_2d(const TexDev<R>& t, const TexCtx<R>& c, R st[2], R dx[2], R dy[2]) {
  if (t.mapping == 0) {
    dx[0] = t.map[0] * c.dudx; dx[1] = t.map[1] * c.dvdx;
    dy[0] = t.map[0] * c.dudy; dy[1] = t.map[1] * c.dvdy;
    st[0] = t.map[0] * c.u + t.map[2]; st[1] = t.map[1] * c.v + t.map[3];
  } else if (t.mapping == 1 || t.mapping == 2) {
    const R delta = R(0.1);
    R sx[2], sy[2];
    map_round(t, c.p, &st[0], &st[1]);
    map_round(t, c.p + c.dpdx * delta, &sx[0], &sx[1]);
    dx[0] = (sx[0] - st[0]) / delta; dx[1] = (sx[1] - st[1]) / delta;
    map_round(t, c.p + c.dpdy * delta, &sy[0], &sy[1]);
    dy[0] = (sy[0] - st[0]) / delta; dy[1] = (sy[1] - st[1]) / delta;
    if (dx[1] > R(0.5)) dx[1] = R(1) - dx[1]; else if (dx[1] < R(-0.5)) dx[1] = -(dx[1] + R(1));
    if (dy[1] > R(0.5)) dy[1] = R(1) - dy[1]; else if (dy[1] < R(-0.5)) dy[1] = -(dy[1] + R(1));
  } else {
    const V3<R> vs(t.vs), vt(t.vt);
    dx[0] = dot(c.dpdx, vs); dx[1] = dot(c.dpdx, vt);
    dy[0] = dot(c.dpdy, vs); dy[1] = dot(c.dpdy, vt);
    st[0] = t.map[0] + dot(c.p, vs); st[1] = t.map[1] + dot(c.p, vt);
  }
}
template <typename R> RRT_DEV R bump_int(R x) { return floor(x / R(2)) + R(2) * rmax(x / R(2) - floor(x / R(2)) - R(0.5), R(0)); }   // checkerboard.rs:46-48

// ---- MIPMap lookups mipmap.rs:98-268 (level storage: rrt_image in include/rrt.h) ---------------------------------------
template <typename R> RRT_DEV uint32_t f2u_sat(R v) { return !(v > R(0)) ? 0u : (v >= R(4294967295.0) ? 0xffffffffu : (uint32_t)v); }   // `as usize` (32 bits suffice: res <= 2^16)
template <typename R>
RRT_DEV Rgb<R> mip_texel(const SceneDev<R>& s, const ImageDev<R>& im, uint32_t level, uint32_t su, uint32_t tu, uint32_t* err) {   // :107-131
  const ImageLevelDev L = im.levels[level];
  uint32_t ts = 0, tt = 0;
  if (im.wrap == 0) { ts = su % L.u_res; tt = tu % L.v_res; }
  else if (im.wrap == 1) { if (su >= L.u_res || tu >= L.v_res) return Rgb<R>(); }   // in range: texel (0, 0), as written there
  else { ts = su > L.u_res ? L.u_res : su; tt = tu > L.v_res ? L.v_res : tu; }
  const uint32_t i = 16u * (L.u_blocks * (tt & 3u) + (ts & 3u)) + 4u * (tt >> 2) + (ts >> 2);   // memory.rs:76-85
  if (i >= L.n) { *err = 1; return Rgb<R>(); }
  return Rgb<R>(s.image_texels + 3 * (size_t)(L.offset + i));
}
template <typename R>
RRT_DEV Rgb<R> mip_triangle(const SceneDev<R>& s, const ImageDev<R>& im, uint32_t level, const R st[2], uint32_t* err) {   // :193-205
  level = level > (uint32_t)im.n_levels - 1u ? (uint32_t)im.n_levels - 1u : level;
  const ImageLevelDev L = im.levels[level];
  const R sx = st[0] * (R)L.u_res - R(0.5), tx = st[1] * (R)L.v_res - R(0.5);
  const uint32_t s0 = f2u_sat(floor(sx)), t0 = f2u_sat(floor(tx));
  const R ds = sx - trunc(sx), dt = tx - trunc(tx);
  return mip_texel(s, im, level, s0, t0, err) * (R(1) - ds) * (R(1) - dt) + mip_texel(s, im, level, s0, t0 + 1u, err) * (R(1) - ds) * dt +
         mip_texel(s, im, level, s0 + 1u, t0, err) * ds * (R(1) - dt) + mip_texel(s, im, level, s0 + 1u, t0 + 1u, err) * ds * dt;
}
template <typename R>
RRT_DEV Rgb<R> mip_ewa(const SceneDev<R>& s, const ImageDev<R>& im, uint32_t level, const R st_in[2], const R d0[2], const R d1[2], uint32_t* err) {   // :206-268
  if (level >= (uint32_t)im.n_levels) { *err = 1; return Rgb<R>(); }   // level == levels: pyramid[level] out of bounds (level > levels cannot occur: i_lod <= levels - 1)
  const ImageLevelDev L = im.levels[level];
  const R st[2] = {st_in[0] * (R)L.u_res - R(0.5), st_in[1] * (R)L.v_res - R(0.5)};
  const R dst0[2] = {d0[0] * (R)L.u_res, d0[1] * (R)L.v_res}, dst1[2] = {d1[0] * (R)L.u_res, d1[1] * (R)L.v_res};
  R a = dst0[1] * dst0[1] + dst1[1] * dst1[1] + R(1);
  R b = R(-2) * (dst0[0] * dst0[1] + dst1[0] * dst1[1]);
  R c = dst0[0] * dst0[0] + dst1[0] * dst1[0] + R(1);
  const R inv_f = R(1) / (a * c - b * b * R(0.25));
  a *= inv_f; b *= inv_f; c *= inv_f;
  const R det = -b * b + R(4) * a * c, inv_det = R(1) / det;
  const R u_sqrt = sqrt(det * c), v_sqrt = sqrt(det * a);
  const uint32_t s0 = f2u_sat(ceil(st[0] - R(2) * inv_det * u_sqrt)), s1 = f2u_sat(floor(st[0] + R(2) * inv_det * u_sqrt));
  const uint32_t t0 = f2u_sat(ceil(st[1] - R(2) * inv_det * v_sqrt)), t1 = f2u_sat(floor(st[1] + R(2) * inv_det * v_sqrt));
  Rgb<R> sum;
  R sum_wts = R(0);
  for (uint32_t it = t0; it <= t1 && it >= t0; it++) {
    const R tt = (R)it - st[0];   // (st[0], as written at :250)
    for (uint32_t is = s0; is <= s1 && is >= s0; is++) {
      const R ss = (R)is - st[0];
      const R r2 = a * ss * ss + b * ss * tt + c * tt * tt;
      if (r2 < R(1)) {
        const uint32_t index = f2u_sat(rmin(r2 * R(128), R(127)));
        const R weight = (R)(exp(-2.0 * ((double)index / 127.0)) - exp(-2.0));   // WEIGHT_LUT :13-23 (f64 table)
        sum = sum + mip_texel(s, im, level, is, it, err) * weight;
        sum_wts += weight;
      }
    }
  }
  return sum / sum_wts;
}
template <typename R>
RRT_DEV Rgb<R> mip_lookup_d(const SceneDev<R>& s, const ImageDev<R>& im, const R st[2], const R dstdx[2], const R dstdy[2], uint32_t* err) {   // :150-192
  const uint32_t levels = (uint32_t)im.n_levels;
  if (im.do_trilinear) {   // lookup_w :132-149
    const R width = rmax(rmax(rabs(dstdx[0]), rabs(dstdx[1])), rmax(rabs(dstdy[0]), rabs(dstdy[1])));
    const R level = (R)levels - R(1) + (R)log2(rmax(width, R(1e-8)));
    if (level < R(0)) return mip_triangle(s, im, 0u, st, err);
    if (level >= (R)(levels - 1u)) return mip_texel(s, im, levels - 1u, 0u, 0u, err);
    const uint32_t il = f2u_sat(floor(level));
    const R delta = level - trunc(level);
    return mip_triangle(s, im, il, st, err) * (R(1) - delta) + mip_triangle(s, im, il + 1u, st, err) * delta;
  }
  R dst0[2], dst1[2];
  if (dstdx[0] * dstdx[0] + dstdx[1] * dstdx[1] < dstdy[0] * dstdy[0] + dstdy[1] * dstdy[1]) { dst0[0] = dstdy[0]; dst0[1] = dstdy[1]; dst1[0] = dstdx[0]; dst1[1] = dstdx[1]; }
  else { dst0[0] = dstdx[0]; dst0[1] = dstdx[1]; dst1[0] = dstdy[0]; dst1[1] = dstdy[1]; }
  const R major_length = sqrt(dst0[0] * dst0[0] + dst0[1] * dst0[1]);
  R minor_length = sqrt(dst1[0] * dst1[0] + dst1[1] * dst1[1]);
  if (minor_length * im.max_aniso < major_length && minor_length > R(0)) {
    const R scale = major_length / (minor_length * im.max_aniso);
    dst1[0] *= scale; dst1[1] *= scale;
    minor_length *= scale;
  }
  if (minor_length == R(0)) return mip_triangle(s, im, 0u, st, err);
  const R lod = rmax((R)(levels - 1u) + (R)log2(minor_length), R(0));
  const uint32_t i_lod = f2u_sat(floor(lod));
  const R fr = lod - trunc(lod);
  return mip_ewa(s, im, i_lod, st, dst0, dst1, err) * (R(1) - fr) + mip_ewa(s, im, i_lod + 1u, st, dst0, dst1, err) * fr;
}

// Texture::evaluate. LEVEL bounds the recursion at compile time (kTexDepth levels, each its own function: children are
// real calls, not inlined copies); level 0 is never reached for graphs the host accepted.
template <typename R, int LEVEL>
struct TexEval {
  static __device__ __noinline__ Rgb<R> eval(const SceneDev<R>& s, int id, TexCtx<R>& c) {
    const TexDev<R>* texs = s.textures;
    const TexDev<R>& t = texs[id];
    auto child = [&](int slot) -> Rgb<R> {
      return t.child[slot] >= 0 ? TexEval<R, LEVEL - 1>::eval(s, t.child[slot], c) : Rgb<R>(t.fallback[slot]);
    };
    switch (t.type) {
      case 0: return Rgb<R>(t.v[0]);
      case 1: {   // MixTexture mix.rs:32-38
        const R amt = child(2).r;
        return child(0) * (R(1) - amt) + child(1) * amt;
      }
      case 5: return child(0) * child(1);   // ScaleTexture scale.rs:30-32
      case 2: {   // BilerpTexture bilerp.rs:33-43
        R st[2], dx[2], dy[2];
        tex_map_2d(t, c, st, dx, dy);
        return Rgb<R>(t.v[0]) * (R(1) - st[0]) * (R(1) - st[1]) + Rgb<R>(t.v[1]) * (R(1) - st[0]) * st[1] + Rgb<R>(t.v[2]) * st[0] * (R(1) - st[1]) + Rgb<R>(t.v[3]) * st[0] * st[1];
      }
      case 8: {   // UVTexture uv.rs:20-28
        R st[2], dx[2], dy[2];
        tex_map_2d(t, c, st, dx, dy);
        return Rgb<R>(st[0] - floor(st[0]), st[1] - floor(st[1]), R(0));
      }
      case 3: {   // Checkerboard2DTexture checkerboard.rs:54-97
        R st[2], dx[2], dy[2];
        tex_map_2d(t, c, st, dx, dy);
        const bool first = (f2i_sat(floor(st[0])) + f2i_sat(floor(st[1]))) % 2 == 0;
        if (t.aa_none) return child(first ? 0 : 1);
        const R ds = rmax(rabs(dx[0]), rabs(dx[1])), dt = rmax(rabs(dy[0]), rabs(dy[1]));
        const R s0 = st[0] - ds, s1 = st[0] + ds, t0 = st[1] - dt, t1 = st[1] + dt;
        if (floor(s0) == floor(s1) && floor(t0) == floor(t1)) return child(first ? 0 : 1);
        const R sint = (bump_int(s1) - bump_int(s0)) / (R(2) * ds), tint = (bump_int(t1) - bump_int(t0)) / (R(2) * dt);
        R area2 = sint + tint - R(2) * sint * tint;
        if (ds > R(1) || dt > R(1)) area2 = R(0.5);
        return child(0) * (R(1) - area2) + child(1) * area2;
      }
      case 4: {   // Checkerboard3DTexture checkerboard.rs:121-131
        const V3<double> p = aff_pt_d(t.w2t, c.pd);
        return child(f2i_sat(floor(p.x) + floor(p.y) + floor(p.z)) % 2 == 0 ? 0 : 1);
      }
      case 9: {   // ImageTexture imagemap.rs:74-81
        R st[2], dx[2], dy[2];
        tex_map_2d(t, c, st, dx, dy);
        return mip_lookup_d(s, s.images[t.image], st, dx, dy, &c.err);
      }
      case 6: {   // WindyTexture windy.rs:15-23
        const V3<double> p = aff_pt_d(t.w2t, c.pd), dpdx = aff_vec_d(t.w2t, c.dpdx), dpdy = aff_vec_d(t.w2t, c.dpdy);
        const double wind_strength = noise_sum(p * 0.1, dpdx * 0.1, dpdy * 0.1, 0.5, 3, false);
        const double wave_height = noise_sum(p, dpdx, dpdy, 0.5, 6, false);
        return Rgb<R>((R)(rabs(wind_strength) * wave_height));
      }
      default: {   // 7: WrinkledTexture wrinkled.rs:21-28
        const V3<double> p = aff_pt_d(t.w2t, c.pd), dpdx = aff_vec_d(t.w2t, c.dpdx), dpdy = aff_vec_d(t.w2t, c.dpdy);
        return Rgb<R>((R)noise_sum(p, dpdx, dpdy, (double)t.omega, t.octaves, true));
      }
    }
  }
};
template <typename R>
struct TexEval<R, 0> {
  static __device__ Rgb<R> eval(const SceneDev<R>&, int, TexCtx<R>&) { return Rgb<R>(); }
};

// the material with every textured parameter replaced by its value at this hit (materials evaluate their textures at
// the top of compute_scattering_functions, e.g. matte.rs:47-49)
template <typename R>
RRT_DEV Material<R> resolve_material(const SceneDev<R>& s, const Material<R>& m0, TexCtx<R>& c) {
  Material<R> m = m0;
  if (!m.has_tex) return m;
  auto ev = [&](int slot) -> Rgb<R> { return TexEval<R, kTexDepth>::eval(s, m0.tex[slot], c); };
  auto put3 = [&](int slot, R* dst) { if (m0.tex[slot] >= 0) { const Rgb<R> v = ev(slot); dst[0] = v.r; dst[1] = v.g; dst[2] = v.b; } };
  auto put1 = [&](int slot, R* dst) { if (m0.tex[slot] >= 0) *dst = ev(slot).r; };
  put3(0, m.kd); put3(1, m.ks); put3(2, m.kr); put3(3, m.eta); put3(4, m.k);
  put1(5, &m.sigma); put1(6, &m.roughness); put1(7, &m.u_roughness); put1(8, &m.v_roughness);
  put3(9, m.kt); put3(10, m.reflect); put3(11, m.transmit); put1(12, &m.index);
  return m;
}

}  // namespace rrtd
