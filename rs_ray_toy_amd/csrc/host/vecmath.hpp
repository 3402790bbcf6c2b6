// Host-side f64 vector / matrix helpers for scene build (loader, BVH builder, camera init).
// Semantics follow the reference's geometry.rs / transform.rs where results depend on them
// (operation order is kept so f64 results are reproducible; build with -ffp-contract=off).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <stdexcept>

#include "rrt.h"

namespace rrt {

struct Panic : std::runtime_error {  // "the reference would panic here"
  using std::runtime_error::runtime_error;
};
struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct ParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct IoError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void set_last_error(const std::string& msg);

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() = default;
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// geometry.rs:1099-1107
inline V3 cross(V3 a, V3 b) {
  return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}
inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double length(V3 a) { return std::sqrt(length_squared(a)); }
// Vector3::normalize geometry.rs:925-931 (zero vector is returned unchanged)
inline V3 normalize(V3 a) {
  double l = length(a);
  if (l == 0.0) return a;
  return a / l;
}

// Bounds3f with the reference's inverted-MAX default (geometry.rs:1549-1567)
struct B3 {
  V3 pmin{std::numeric_limits<double>::max(), std::numeric_limits<double>::max(),
          std::numeric_limits<double>::max()};
  V3 pmax{std::numeric_limits<double>::lowest(), std::numeric_limits<double>::lowest(),
          std::numeric_limits<double>::lowest()};
};
// PointMin / PointMax geometry.rs:361-420: `if a < b {a} else {b}` (the *other* operand wins on NaN/ties)
inline double fmin_ref(double a, double b) { return a < b ? a : b; }
inline double fmax_ref(double a, double b) { return a > b ? a : b; }
inline B3 bunion(const B3& b, V3 p) {
  B3 r;
  r.pmin = {fmin_ref(b.pmin.x, p.x), fmin_ref(b.pmin.y, p.y), fmin_ref(b.pmin.z, p.z)};
  r.pmax = {fmax_ref(b.pmax.x, p.x), fmax_ref(b.pmax.y, p.y), fmax_ref(b.pmax.z, p.z)};
  return r;
}
inline B3 bunion(const B3& a, const B3& b) {
  B3 r;
  r.pmin = {fmin_ref(a.pmin.x, b.pmin.x), fmin_ref(a.pmin.y, b.pmin.y), fmin_ref(a.pmin.z, b.pmin.z)};
  r.pmax = {fmax_ref(a.pmax.x, b.pmax.x), fmax_ref(a.pmax.y, b.pmax.y), fmax_ref(a.pmax.z, b.pmax.z)};
  return r;
}
// Bounds3::new geometry.rs:1571-1587 (component-wise ordering of two points)
inline B3 bnew(V3 p1, V3 p2) {
  B3 r;
  r.pmin = {p1.x > p2.x ? p2.x : p1.x, p1.y > p2.y ? p2.y : p1.y, p1.z > p2.z ? p2.z : p1.z};
  r.pmax = {p1.x > p2.x ? p1.x : p2.x, p1.y > p2.y ? p1.y : p2.y, p1.z > p2.z ? p1.z : p2.z};
  return r;
}
// surface_area geometry.rs:1618-1626 (overflows to +inf on the default box: part of Q27)
inline double surface_area(const B3& b) {
  V3 d = b.pmax - b.pmin;
  double r = d.x * d.y + d.x * d.z + d.y * d.z;
  return r + r;
}
// maximum_extent geometry.rs:1627-1639
inline int maximum_extent(const B3& b) {
  V3 d = b.pmax - b.pmin;
  if (d.x > d.y && d.x > d.z) return 0;
  if (d.y > d.z) return 1;
  return 2;
}
// offset geometry.rs:1640-1655
inline V3 boffset(const B3& b, V3 p) {
  V3 o = p - b.pmin;
  if (b.pmax.x > b.pmin.x) o.x /= b.pmax.x - b.pmin.x;
  if (b.pmax.y > b.pmin.y) o.y /= b.pmax.y - b.pmin.y;
  if (b.pmax.z > b.pmin.z) o.z /= b.pmax.z - b.pmin.z;
  return o;
}

struct M4 {
  double m[4][4];
  static M4 identity() {
    M4 r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
    return r;
  }
};
M4 m4_mul(const M4& a, const M4& b);      // mtx_mul transform.rs:166-178
M4 m4_transpose(const M4& a);
M4 m4_inverse(const M4& a);               // Matrix4x4::inverse transform.rs:66-134 (Gauss-Jordan, full pivot)

struct Xf {
  M4 m = M4::identity(), minv = M4::identity();
};
Xf xf_mul(const Xf& a, const Xf& b);      // impl Mul for Transform transform.rs:440-448
Xf xf_inverse(const Xf& a);
Xf xf_translate(V3 d);                    // transform.rs:252-264
Xf xf_scale(double x, double y, double z);// transform.rs:265-289
Xf xf_rotate(double theta_deg, V3 axis);  // transform.rs:327-351
Xf xf_look_at(V3 pos, V3 look, V3 up);    // transform.rs:352-392
bool xf_is_identity(const Xf& a);
V3 xf_point(const Xf& t, V3 p);           // Transformable for Point3f transform.rs:455-491
V3 xf_vector(const Xf& t, V3 v);          // transform.rs:493-504
V3 xf_normal(const Xf& t, V3 n);          // transform.rs:506-523
B3 xf_bounds(const Xf& t, const B3& b);   // transform.rs:539-612
void xf_to_abi(const Xf& t, rrt_xform* out);

inline double radians(double deg) { return (M_PI / 180.0) * deg; }  // misc.rs:55-57
inline double lerp(double t, double a, double b) { return a * (1.0 - t) + b * t; }  // misc.rs:223-228
// misc.rs:231-251
inline bool quadratic(double a, double b, double c, double* t0, double* t1) {
  double discrim = b * b - 4.0 * a * c;
  if (discrim < 0.0) return false;
  double root = std::sqrt(discrim);
  double q = (b < 0.0) ? -0.5 * (b - root) : -0.5 * (b + root);
  *t0 = q / a;
  *t1 = c / q;
  if (*t0 > *t1) { double t = *t0; *t0 = *t1; *t1 = t; }
  return true;
}

}  // namespace rrt
