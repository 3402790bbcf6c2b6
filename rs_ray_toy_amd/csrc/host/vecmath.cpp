// f64 matrix / transform routines for the host scene build; operation order follows
// /root/reference/src/transform.rs (cited per function in vecmath.hpp).
#include "vecmath.hpp"

namespace rrt {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

M4 m4_mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
  return r;
}

M4 m4_transpose(const M4& a) {
  M4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
  return r;
}

M4 m4_inverse(const M4& a) {
  int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
  M4 minv = a;
  for (int i = 0; i < 4; i++) {
    int irow = 0, icol = 0;
    double big = 0.0;
    for (int j = 0; j < 4; j++) {
      if (ipiv[j] != 1) {
        for (int k = 0; k < 4; k++) {
          if (ipiv[k] == 0) {
            double ab = std::fabs(minv.m[j][k]);
            if (ab >= big) { big = ab; irow = j; icol = k; }
          }
        }
      }
    }
    ipiv[icol] += 1;
    if (irow != icol)
      for (int k = 0; k < 4; k++) { double t = minv.m[irow][k]; minv.m[irow][k] = minv.m[icol][k]; minv.m[icol][k] = t; }
    indxr[i] = irow;
    indxc[i] = icol;
    double pivinv = 1.0 / minv.m[icol][icol];
    minv.m[icol][icol] = 1.0;
    for (int j = 0; j < 4; j++) minv.m[icol][j] *= pivinv;
    for (int j = 0; j < 4; j++) {
      if (j != icol) {
        double save = minv.m[j][icol];
        minv.m[j][icol] = 0.0;
        for (int k = 0; k < 4; k++) minv.m[j][k] -= minv.m[icol][k] * save;
      }
    }
  }
  for (int i = 0; i < 4; i++) {
    int j = 3 - i;
    if (indxr[j] != indxc[j])
      for (int k = 0; k < 4; k++) { double t = minv.m[k][indxr[j]]; minv.m[k][indxr[j]] = minv.m[k][indxc[j]]; minv.m[k][indxc[j]] = t; }
  }
  return minv;
}

Xf xf_mul(const Xf& a, const Xf& b) {
  Xf r;
  r.m = m4_mul(a.m, b.m);
  r.minv = m4_mul(b.minv, a.minv);
  return r;
}
Xf xf_inverse(const Xf& a) { Xf r; r.m = a.minv; r.minv = a.m; return r; }

Xf xf_translate(V3 d) {
  Xf r;
  r.m.m[0][3] = d.x; r.m.m[1][3] = d.y; r.m.m[2][3] = d.z;
  r.minv.m[0][3] = -d.x; r.minv.m[1][3] = -d.y; r.minv.m[2][3] = -d.z;
  return r;
}
Xf xf_scale(double x, double y, double z) {
  Xf r;
  r.m.m[0][0] = x; r.m.m[1][1] = y; r.m.m[2][2] = z;
  r.minv.m[0][0] = 1.0 / x; r.minv.m[1][1] = 1.0 / y; r.minv.m[2][2] = 1.0 / z;
  return r;
}
Xf xf_rotate(double theta, V3 axis) {
  V3 a = normalize(axis);
  double s = std::sin(radians(theta)), c = std::cos(radians(theta));
  M4 m = M4::identity();
  m.m[0][0] = a.x * a.x + (1.0 - a.x * a.x) * c;
  m.m[0][1] = a.x * a.y * (1.0 - c) - a.z * s;
  m.m[0][2] = a.x * a.z * (1.0 - c) + a.y * s;
  m.m[0][3] = 0.0;
  m.m[1][0] = a.x * a.y * (1.0 - c) + a.z * s;
  m.m[1][1] = a.y * a.y + (1.0 - a.y * a.y) * c;
  m.m[1][2] = a.y * a.z * (1.0 - c) - a.x * s;
  m.m[1][3] = 0.0;
  m.m[2][0] = a.x * a.z * (1.0 - c) - a.y * s;
  m.m[2][1] = a.y * a.z * (1.0 - c) + a.x * s;
  m.m[2][2] = a.z * a.z + (1.0 - a.z * a.z) * c;
  m.m[2][3] = 0.0;
  Xf r;
  r.m = m;
  r.minv = m4_transpose(m);
  return r;
}
Xf xf_look_at(V3 pos, V3 look, V3 up) {
  M4 c2w = M4::identity();
  c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1.0;
  V3 dir = normalize(look - pos);
  if (length(cross(normalize(up), dir)) == 0.0) return Xf();  // identity + stderr note in the reference
  V3 left = normalize(cross(normalize(up), dir));
  V3 new_up = cross(dir, left);
  c2w.m[0][0] = left.x; c2w.m[1][0] = left.y; c2w.m[2][0] = left.z; c2w.m[3][0] = 0.0;
  c2w.m[0][1] = new_up.x; c2w.m[1][1] = new_up.y; c2w.m[2][1] = new_up.z; c2w.m[3][1] = 0.0;
  c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.0;
  Xf r;
  r.m = m4_inverse(c2w);
  r.minv = c2w;
  return r;
}
bool xf_is_identity(const Xf& a) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (a.m.m[i][j] != ((i == j) ? 1.0 : 0.0)) return false;
  return true;
}
V3 xf_point(const Xf& t, V3 p) {
  const auto& m = t.m.m;
  double xp = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
  double yp = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
  double zp = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
  double wp = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
  if (wp == 0.0) throw Panic("transform.rs:479 assert!(wp != 0.0)");
  if (wp == 1.0) return {xp, yp, zp};
  double inv = 1.0 / wp;
  return {inv * xp, inv * yp, inv * zp};
}
V3 xf_vector(const Xf& t, V3 v) {
  const auto& m = t.m.m;
  return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
          m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}
V3 xf_normal(const Xf& t, V3 n) {
  const auto& mi = t.minv.m;
  return {mi[0][0] * n.x + mi[1][0] * n.y + mi[2][0] * n.z, mi[0][1] * n.x + mi[1][1] * n.y + mi[2][1] * n.z,
          mi[0][2] * n.x + mi[1][2] * n.y + mi[2][2] * n.z};
}
B3 xf_bounds(const Xf& t, const B3& b) {
  V3 p = xf_point(t, {b.pmin.x, b.pmin.y, b.pmin.z});
  B3 r;
  r.pmin = p; r.pmax = p;
  r = bunion(r, xf_point(t, {b.pmax.x, b.pmin.y, b.pmin.z}));
  r = bunion(r, xf_point(t, {b.pmin.x, b.pmax.y, b.pmin.z}));
  r = bunion(r, xf_point(t, {b.pmin.x, b.pmin.y, b.pmax.z}));
  r = bunion(r, xf_point(t, {b.pmin.x, b.pmax.y, b.pmax.z}));
  r = bunion(r, xf_point(t, {b.pmax.x, b.pmax.y, b.pmin.z}));
  r = bunion(r, xf_point(t, {b.pmax.x, b.pmin.y, b.pmax.z}));
  r = bunion(r, xf_point(t, {b.pmax.x, b.pmax.y, b.pmax.z}));
  return r;
}
void xf_to_abi(const Xf& t, rrt_xform* out) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { out->m[i * 4 + j] = t.m.m[i][j]; out->m_inv[i * 4 + j] = t.minv.m[i][j]; }
}

}  // namespace rrt
