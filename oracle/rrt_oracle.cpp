// ORACLE — test infrastructure, not product code.
//
// Scalar f64 CPU restatement of rs_ray_toy's render hot path (reference @ /root/reference/src, cited
// per function). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library, and only as the checker / reported CPU baseline. The product (librrt.so) never links it.
//
// Input: the same flat rrt_scene_desc (include/rrt.h) the HIP executor consumes; instancing is kept as
// the reference does it (per-primitive ray transform, primitives.rs:115-139) rather than flattened.
//
// Parity status: the reference cannot be built here (no cargo/rustc, nightly-only crate, un-vendored
// deps), so this restatement is pinned by the reference's known-answer unit values
// (geometry.rs tests, test_sphere, test_primitive), the Halton index constants and copper RGB derived
// in SURVEY §8c, and samples/{scene.json,cube.obj}. RNG streams (Halton digit permutations,
// thread_rng) are "parity unpinned" by construction: they are replaced by a seeded table in the desc.
//
// Build: g++ -O2 -ffp-contract=off (no FMA contraction: Rust does not contract) -fopenmp.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "rrt.h"
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const double PI = 3.14159265358979323846;
const double INF = std::numeric_limits<double>::infinity();
const double MACHINE_EPSILON = 2.220446049250313e-16 * 0.5;  // main.rs:53
const double ONE_MINUS_EPSILON = 1.0 - MACHINE_EPSILON;       // misc.rs:19
const double SHADOW_EPSILON = 0.0001;                         // misc.rs:18
const double MAX_DIST = 1999999999.0;                         // main.rs:51
const double INV_PI = 0.31830988618379067154;
const double PI_OVER_2 = 1.57079632679489661923, PI_OVER_4 = 0.78539816339744830961;

struct OraclePanic { std::string msg; };
thread_local std::string g_err;

inline double gamma_n(int n) { return (n * MACHINE_EPSILON) / (1.0 - n * MACHINE_EPSILON); }  // misc.rs:40-42
inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }  // misc.rs:98-112
inline double rmax(double a, double b) { return std::fmax(a, b); }  // Rust f64::max
inline double rmin(double a, double b) { return std::fmin(a, b); }

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  // geometry.rs:106-113
inline double absdot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline V3 cross(V3 a, V3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
inline double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double len(V3 a) { return std::sqrt(len2(a)); }
inline V3 vnormalize(V3 a) { double l = len(a); return l == 0.0 ? a : a / l; }  // Vector3::normalize geometry.rs:925
inline V3 nnormalize(V3 a) { return a / len(a); }                                // Normal3::normalize geometry.rs:1209
inline V3 faceforward(V3 n, V3 v) { return dot(n, v) < 0.0 ? -n : n; }           // geometry.rs:1381-1387
// vec3_coordinate_system geometry.rs:1146-1161
inline void coordinate_system(V3 v1, V3* v2, V3* v3) {
  if (std::fabs(v1.x) > std::fabs(v1.y)) *v2 = V3(-v1.z, 0.0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
  else *v2 = V3(0.0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
  *v3 = cross(v1, *v2);
}

struct Rgb {
  double c[3] = {0, 0, 0};
  Rgb() {}
  Rgb(double a, double b, double d) { c[0] = a; c[1] = b; c[2] = d; }
  explicit Rgb(double v) { c[0] = c[1] = c[2] = v; }
  explicit Rgb(const double* p) { c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; }
  bool is_black() const { return c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0; }  // spectrum.rs:2162
  double y() const { return 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2]; }  // spectrum.rs:2733-2736
  double max_component() const { return rmax(rmax(c[0], c[1]), c[2]); }
  bool has_nan() const { return c[0] != c[0] || c[1] != c[1] || c[2] != c[2]; }
};
inline Rgb operator+(Rgb a, Rgb b) { return {a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]}; }
inline Rgb operator-(Rgb a, Rgb b) { return {a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]}; }
inline Rgb operator*(Rgb a, Rgb b) { return {a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]}; }
inline Rgb operator/(Rgb a, Rgb b) { return {a.c[0] / b.c[0], a.c[1] / b.c[1], a.c[2] / b.c[2]}; }
inline Rgb operator*(Rgb a, double s) { return {a.c[0] * s, a.c[1] * s, a.c[2] * s}; }
inline Rgb operator/(Rgb a, double s) { return {a.c[0] / s, a.c[1] / s, a.c[2] / s}; }
inline Rgb rsqrt(Rgb a) { return {std::sqrt(a.c[0]), std::sqrt(a.c[1]), std::sqrt(a.c[2])}; }
inline Rgb rclamp0(Rgb a) { return {clampd(a.c[0], 0.0, INF), clampd(a.c[1], 0.0, INF), clampd(a.c[2], 0.0, INF)}; }

struct Ray { V3 o, d; double t_max = INF; };
// the differential half of RayDifferential (geometry.rs:80-92)
struct RayDiff { bool has = false; V3 rxo, rxd, ryo, ryd; };
inline Ray ray_new(V3 o, V3 d, double tmax) { Ray r; r.o = o; r.d = vnormalize(d); r.t_max = tmax; return r; }  // geometry.rs:1841
inline V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }

struct Xf { double m[16], mi[16]; };
inline V3 xf_pt(const double* m, V3 p) {  // transform.rs:455-491
  double xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
  double yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
  double zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
  double wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
  if (wp == 0.0) throw OraclePanic{"transform.rs:479 assert!(wp != 0.0)"};
  if (wp == 1.0) return {xp, yp, zp};
  double inv = 1.0 / wp;
  return {inv * xp, inv * yp, inv * zp};
}
inline V3 xf_vec(const double* m, V3 v) {  // transform.rs:493-504
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
inline V3 xf_nrm(const double* minv, V3 n) {  // transform.rs:506-523 (inverse transpose)
  return {minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z,
          minv[2] * n.x + minv[6] * n.y + minv[10] * n.z};
}
inline Ray xf_ray(const double* m, const Ray& r) {  // transform.rs:525-537
  return ray_new(xf_pt(m, r.o), vnormalize(xf_vec(m, r.d)), r.t_max);
}
// linear part orthonormal within 1e-9 and affine: what the device flattens to world space (rrt_impl.hpp is_rigid); anything else
// (scale, shear) keeps the reference's per-primitive ray transform there too
inline bool is_rigid(const double* m) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += m[k * 4 + i] * m[k * 4 + j];
      if (std::fabs(s - (i == j ? 1.0 : 0.0)) > 1e-9) return false;
    }
  return m[12] == 0.0 && m[13] == 0.0 && m[14] == 0.0 && m[15] == 1.0;
}
inline bool is_identity(const double* m) {
  for (int i = 0; i < 16; i++)
    if (m[i] != ((i % 5 == 0) ? 1.0 : 0.0)) return false;
  return true;
}

// misc.rs:231-251
inline bool quadratic(double a, double b, double c, double* t0, double* t1) {
  double discrim = b * b - 4.0 * a * c;
  if (discrim < 0.0) return false;
  double root = std::sqrt(discrim);
  double q = (b < 0.0) ? -0.5 * (b - root) : -0.5 * (b + root);
  *t0 = q / a;
  *t1 = c / q;
  if (*t0 > *t1) std::swap(*t0, *t1);
  return true;
}

// ---- SurfaceInteraction (interaction.rs:95-181), trimmed to what in-scope materials read ----------
struct SI {
  V3 p, n, wo;                 // BaseInteraction
  double u = 0, v = 0;         // uv
  V3 dpdu, dpdv;
  V3 dndu, dndv;
  V3 sn, sdpdu, sdpdv;         // shading
  V3 sdndu, sdndv;
  V3 dpdx, dpdy;               // compute_differentials interaction.rs:223-284
  double dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
  int prim = -1;               // index into prims (GeometricPrimitive)
  bool valid = false;
};

// SurfaceInteraction::new interaction.rs:131-181
inline void si_new(SI* s, V3 p, double u, double v, V3 wo, V3 dpdu, V3 dpdv, V3 dndu = V3(), V3 dndv = V3()) {
  V3 n = vnormalize(cross(dpdu, dpdv));
  s->p = p; s->u = u; s->v = v; s->wo = wo; s->dpdu = dpdu; s->dpdv = dpdv;
  s->n = n; s->sn = n; s->sdpdu = dpdu; s->sdpdv = dpdv;
  s->dndu = dndu; s->dndv = dndv; s->sdndu = dndu; s->sdndv = dndv;
  s->dpdx = s->dpdy = V3(); s->dudx = s->dvdx = s->dudy = s->dvdy = 0.0;
}
// set_shading_geometry interaction.rs:183-202
inline void si_set_shading(SI* s, V3 dpdus, V3 dpdvs, V3 dndus, V3 dndvs, bool authoritative) {
  V3 n = nnormalize(cross(dpdus, dpdvs));
  if (authoritative) n = faceforward(s->n, n); else n = faceforward(n, s->n);
  s->sn = n; s->sdpdu = dpdus; s->sdpdv = dpdvs; s->sdndu = dndus; s->sdndv = dndvs;
}
// Transformable for SurfaceInteraction transform.rs:628-655
inline void si_transform(SI* s, const double* m, const double* minv) {
  s->p = xf_pt(m, s->p);
  s->wo = xf_vec(m, s->wo);
  s->n = xf_nrm(minv, s->n);                 // BaseInteraction::t_by: not re-normalised
  s->dpdu = xf_vec(m, s->dpdu);
  s->dpdv = xf_vec(m, s->dpdv);
  s->dndu = xf_nrm(minv, s->dndu);
  s->dndv = xf_nrm(minv, s->dndv);
  s->sn = nnormalize(xf_nrm(minv, s->sn));
  s->sdpdu = xf_vec(m, s->sdpdu);
  s->sdpdv = xf_vec(m, s->sdpdv);
  s->sdndu = xf_nrm(minv, s->sdndu);
  s->sdndv = xf_nrm(minv, s->sdndv);
  s->sn = faceforward(s->sn, s->n);
}

struct Scene {
  const rrt_scene_desc* d;
  // flat = true evaluates rigid instances the way the device does: vertices (and vertex normals) are moved
  // to world space with the instance matrix and the *world* ray is tested, instead of the reference's
  // per-primitive ray transform (primitives.rs:115-139). Same semantics, different rounding: used to check
  // the HIP path bit-for-bit, and to measure how often the two evaluations break a tie differently.
  bool flat = false;
  V3 P(uint32_t i) const { return {d->positions[3 * i], d->positions[3 * i + 1], d->positions[3 * i + 2]}; }
  V3 N(uint32_t i) const { return {d->normals[3 * i], d->normals[3 * i + 1], d->normals[3 * i + 2]}; }
  void UV(uint32_t i, double* u, double* v) const { *u = d->uvs[2 * i]; *v = d->uvs[2 * i + 1]; }
};

// ---- Triangle (shape/triangle.rs) ------------------------------------------------------------------
// intersect_p :167-205 (E2 = p2 - p1: Q11)
// device upload formula (rrt_impl.hpp upload_scene): row-major affine product, left-to-right sums
inline V3 flat_pt(const double* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
inline V3 flat_nrm(const double* mi, V3 n) {
  return {mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z};
}
bool tri_intersect_p(const Scene& sc, const rrt_tri& t, const Ray& r, const rrt_xform* fx = nullptr) {
  V3 p0 = sc.P(t.v[0]), p1 = sc.P(t.v[1]), p2 = sc.P(t.v[2]);
  if (fx) { p0 = flat_pt(fx->m, p0); p1 = flat_pt(fx->m, p1); p2 = flat_pt(fx->m, p2); }
  V3 E1 = p1 - p0, E2 = p2 - p1, D = r.d;
  V3 Pv = cross(D, E2);
  double a = dot(E1, Pv);
  if (a > -0.0000001 && a < 0.0000001) return false;
  double f = 1.0 / a;
  V3 T = r.o - p0;
  double u = f * dot(T, Pv);
  if (u < 0.0 || u > 1.0) return false;
  V3 Q = cross(T, E1);
  double v = f * dot(D, Q);
  if (v < 0.0 || (u + v) > 1.0) return false;
  double tt = f * dot(E2, Q);
  if (tt < 0.0000001) return false;
  return true;
}
// intersect :226-391 (never compares t with ray.t_max: Q10)
bool tri_intersect(const Scene& sc, const rrt_tri& t, const Ray& r, double* thit, SI* ist, double* bu, double* bv, const rrt_xform* fx = nullptr) {
  V3 p0 = sc.P(t.v[0]), p1 = sc.P(t.v[1]), p2 = sc.P(t.v[2]);
  if (fx) { p0 = flat_pt(fx->m, p0); p1 = flat_pt(fx->m, p1); p2 = flat_pt(fx->m, p2); }
  V3 E1 = p1 - p0, E2 = p2 - p0, D = r.d;
  V3 Pv = cross(D, E2);
  double a = dot(E1, Pv);
  if (a > -0.0000001 && a < 0.0000001) return false;
  double f = 1.0 / a;
  V3 T = r.o - p0;
  double u = f * dot(T, Pv);
  if (u < 0.0 || u > 1.0) return false;
  V3 Q = cross(T, E1);
  double v = f * dot(D, Q);
  if (v < 0.0 || (u + v) > 1.0) return false;
  double tt = f * dot(E2, Q);
  if (tt < 0.0000001) return false;
  *thit = tt;
  // get_uvs :113-128
  double uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
  // mesh_has_uv != 0 <=> `!self.mesh.uv.is_empty()`; the loader stores index 0 of the mesh three times
  // when uv_indices is empty, as Triangle::new does (triangle.rs:90-99)
  if (t.mesh_has_uv) for (int k = 0; k < 3; k++) sc.UV(t.uv[k], &uv[k][0], &uv[k][1]);
  double duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
  V3 dp02 = p0 - p2, dp12 = p1 - p2;
  double determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
  bool degenerate_uv = std::fabs(determinant) < 1e-8;
  V3 dpdu, dpdv;
  if (!degenerate_uv) {
    double i_det = 1.0 / determinant;
    dpdu = (dp02 * duv12[1] - dp12 * duv02[1]) * i_det;
    dpdv = (dp02 * -duv12[0] + dp12 * duv02[0]) * i_det;
  }
  if (degenerate_uv || len2(cross(dpdu, dpdv)) == 0.0) {
    V3 ng = cross(p2 - p0, p1 - p0);
    if (len2(ng) == 0.0) return false;
    coordinate_system(vnormalize(ng), &dpdu, &dpdv);
  }
  V3 p_hit = ray_at(r, tt);
  double uvh[2] = {uv[0][0] * (1.0 - u - v) + uv[1][0] * u + uv[2][0] * v, uv[0][1] * (1.0 - u - v) + uv[1][1] * u + uv[2][1] * v};
  si_new(ist, p_hit, uvh[0], uvh[1], -r.d, dpdu, dpdv);
  V3 ist_n = vnormalize(cross(dp02, dp12));
  ist->n = ist_n;
  ist->sn = ist_n;
  if (t.mesh_has_n == 1) {  // n and normal_indices both non-empty (mesh.s is always empty: objparser.rs:91)
    V3 n0 = sc.N(t.n[0]), n1 = sc.N(t.n[1]), n2 = sc.N(t.n[2]);
    if (fx) { n0 = flat_nrm(fx->m_inv, n0); n1 = flat_nrm(fx->m_inv, n1); n2 = flat_nrm(fx->m_inv, n2); }
    V3 ns = n0 * (1.0 - u - v) + n1 * u + n2 * v;
    if (len2(ns) > 0.0) ns = nnormalize(ns); else ns = ist_n;
    V3 ss = vnormalize(ist->dpdu);
    V3 ts = cross(ss, ns);
    if (len2(ts) > 0.0) { ts = vnormalize(ts); ss = cross(ts, ns); }
    else coordinate_system(ns, &ss, &ts);
    // dndu / dndv of the shading geometry :351-386 (read by the specular ray differentials, integrator/mod.rs:188-196)
    V3 dndu, dndv;
    {
      V3 dn1 = n0 - n2, dn2 = n1 - n2;
      if (degenerate_uv) {
        V3 dn = cross(n2 - n0, n1 - n0);
        if (len2(dn) != 0.0) coordinate_system(dn, &dndu, &dndv);
      } else {
        double i_det = 1.0 / determinant;
        dndu = (dn1 * duv12[1] - dn2 * duv02[1]) * i_det;
        dndv = (dn1 * -duv12[0] + dn2 * duv02[0]) * i_det;
      }
    }
    si_set_shading(ist, ss, ts, dndu, dndv, true);
  }
  *bu = u; *bv = v;
  return true;
}
double tri_area(const Scene& sc, const rrt_tri& t) {  // :420-425
  V3 p0 = sc.P(t.v[0]), p1 = sc.P(t.v[1]), p2 = sc.P(t.v[2]);
  return 0.5 * len(cross(p1 - p0, p2 - p0));
}

// ---- Sphere (shape/sphere.rs) ----------------------------------------------------------------------
struct SphereRef { const rrt_sphere* s; const double *m, *mi; };
SphereRef sphere_ref(const Scene& sc, uint32_t i) { const rrt_sphere* s = &sc.d->spheres[i]; return {s, sc.d->xforms[s->xform].m, sc.d->xforms[s->xform].m_inv}; }

// intersect_p :51-108 (p_hit/phi start at 0: the first clip test sees p_hit = origin)
bool sphere_intersect_p(const SphereRef& S, const Ray& r) {
  const rrt_sphere& s = *S.s;
  double phi = 0.0;
  V3 p_hit;
  Ray ray = xf_ray(S.mi, r);
  double ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  double a = dx * dx + dy * dy + dz * dz, b = 2.0 * (dx * ox + dy * oy + dz * oz), c = ox * ox + oy * oy + oz * oz - s.radius * s.radius;
  double t0 = 0, t1 = 0;
  if (!quadratic(a, b, c, &t0, &t1)) return false;
  if (t0 > MAX_DIST || t1 <= 0.0) return false;
  double t_hit = t0;
  if (t0 <= 0.0) { t_hit = t1; if (t_hit > MAX_DIST) return false; }
  if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || (phi > s.phi_max)) {
    if (t_hit == t1) return false;
    if (t1 > MAX_DIST) return false;
    t_hit = t1;
    p_hit = ray_at(ray, t_hit);
    p_hit = p_hit * (s.radius / len(p_hit - V3()));
    if (p_hit.x == 0.0 && p_hit.y == 0.0) p_hit.x = 1e-5 * s.radius;
    phi = std::atan2(p_hit.y, p_hit.x);
    if (phi < 0.0) phi += 2.0 * PI;
    if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || (phi > s.phi_max)) return false;
  }
  return true;
}
// intersect :124-259 (first p_hit uses the *world* ray: Q16)
bool sphere_intersect(const SphereRef& S, const Ray& r, double* thit, SI* ist) {
  const rrt_sphere& s = *S.s;
  Ray ray = xf_ray(S.mi, r);
  double ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  double a = dx * dx + dy * dy + dz * dz, b = 2.0 * (dx * ox + dy * oy + dz * oz), c = ox * ox + oy * oy + oz * oz - s.radius * s.radius;
  double t0 = 0, t1 = 0;
  if (!quadratic(a, b, c, &t0, &t1)) return false;
  if (t0 > MAX_DIST || t1 <= 0.0) return false;
  double t_hit = t0;
  if (t0 <= 0.0) { t_hit = t1; if (t_hit > MAX_DIST) return false; }
  V3 p_hit = ray_at(r, t_hit);
  if (p_hit.x == 0.0 && p_hit.y == 0.0) p_hit.x = 1e-5 * s.radius;
  double phi = std::atan2(p_hit.y, p_hit.x);
  if (phi < 0.0) phi += 2.0 * PI;
  if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || (phi > s.phi_max)) {
    if (t_hit == t1) return false;
    if (t1 > MAX_DIST) return false;
    t_hit = t1;
    p_hit = ray_at(ray, t_hit);
    p_hit = p_hit * (s.radius / len(p_hit - V3()));
    if (p_hit.x == 0.0 && p_hit.y == 0.0) p_hit.x = 1e-5 * s.radius;
    phi = std::atan2(p_hit.y, p_hit.x);
    if (phi < 0.0) phi += 2.0 * PI;
    if ((s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || (phi > s.phi_max)) return false;
  }
  double u = phi / s.phi_max;
  double theta = std::acos(clampd(p_hit.z / s.radius, -1.0, 1.0));
  double v = (theta - s.theta_min) / (s.theta_max - s.theta_min);
  double z_radius = std::sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
  double inv_z_radius = 1.0 / z_radius;
  double cos_phi = p_hit.x * inv_z_radius, sin_phi = p_hit.y * inv_z_radius;
  V3 dpdu(-s.phi_max * p_hit.y, s.phi_max * p_hit.x, 0.0);
  V3 dpdv = V3(p_hit.z * cos_phi, p_hit.z * sin_phi, -s.radius * std::sin(theta)) * (s.theta_max - s.theta_min);
  // sphere.rs:214-242: dndu / dndv from the fundamental forms
  V3 d2pduu = V3(p_hit.x, p_hit.y, 0.0) * -s.phi_max * s.phi_max;
  V3 d2pduv = V3(-sin_phi, cos_phi, 0.0) * (s.theta_max - s.theta_min) * p_hit.z * s.phi_max;
  V3 d2pdvv = p_hit * -(s.theta_max - s.theta_min) * (s.theta_max - s.theta_min);
  double E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
  V3 N = vnormalize(cross(dpdu, dpdv));
  double e = dot(N, d2pduu), f = dot(N, d2pduv), g = dot(N, d2pdvv);
  double inv_EFG2 = 1.0 / (E * G - F * F);
  V3 dndu = dpdu * ((f * F - e * G) * inv_EFG2) + dpdv * ((e * F - f * E) * inv_EFG2);
  V3 dndv = dpdu * ((g * F - f * G) * inv_EFG2) + dpdv * ((f * F - g * E) * inv_EFG2);
  si_new(ist, p_hit, u, v, -ray.d, dpdu, dpdv, dndu, dndv);
  si_transform(ist, S.m, S.mi);                  // *ist = obj2world.t(ist)
  *thit = t_hit;
  return true;
}

// sampling.rs
inline V3 uniform_sample_sphere(double u0, double u1) {  // :233-243
  double z = 1.0 - 2.0 * u0;
  double r = std::sqrt(rmax(0.0, 1.0 - z * z));
  double phi = 2.0 * PI * u1;
  return {r * std::cos(phi), r * std::sin(phi), z};
}
inline void concentric_sample_disk(double u0, double u1, double* dx, double* dy) {  // :282-304
  double ox = u0 * 2.0 - 1.0, oy = u1 * 2.0 - 1.0;
  if (ox == 0.0 && oy == 0.0) { *dx = 0; *dy = 0; return; }
  double theta, r;
  if (std::fabs(ox) > std::fabs(oy)) { r = ox; theta = PI_OVER_4 * (oy / ox); }
  else { r = oy; theta = PI_OVER_2 - PI_OVER_4 * (ox / oy); }
  *dx = std::cos(theta) * r; *dy = std::sin(theta) * r;
}
inline V3 cosine_sample_hemisphere(double u0, double u1) {  // :270-275
  double dx, dy;
  concentric_sample_disk(u0, u1, &dx, &dy);
  double z = std::sqrt(rmax(0.0, 1.0 - dx * dx - dy * dy));
  return {dx, dy, z};
}
inline double power_heuristic(int nf, double fp, int ng, double gp) {  // :324-328
  double f = nf * fp, g = ng * gp;
  return (f * f) / (f * f + g * g);
}

// ---- Halton sampler (samplers/halton.rs, samplers/mod.rs GlobalSampler, lowdiscrepancy.rs) -----------
struct Primes {
  uint16_t p[1024];
  uint32_t sums[1000];
  Primes() {
    int n = 0;
    for (int c = 2; n < 1024; c++) {
      bool ok = true;
      for (int d = 2; d * d <= c; d++) if (c % d == 0) { ok = false; break; }
      if (ok) p[n++] = (uint16_t)c;
    }
    uint32_t acc = 0;
    for (int i = 0; i < 1000; i++) { sums[i] = acc; acc += p[i]; }
  }
};
const Primes& primes() { static Primes t; return t; }

inline uint64_t reverse_bits_64(uint64_t n) {  // lowdiscrepancy.rs:169-186
  auto r32 = [](uint32_t v) {
    v = (v << 16) | (v >> 16);
    v = ((v & 0x00ff00ff) << 8) | ((v & 0xff00ff00) >> 8);
    v = ((v & 0x0f0f0f0f) << 4) | ((v & 0xf0f0f0f0) >> 4);
    v = ((v & 0x33333333) << 2) | ((v & 0xcccccccc) >> 2);
    v = ((v & 0x55555555) << 1) | ((v & 0xaaaaaaaa) >> 1);
    return v;
  };
  uint64_t n0 = r32((uint32_t)n), n1 = r32((uint32_t)(n >> 32));
  return (n0 << 32) | n1;
}
double radical_inverse(int base_index, uint64_t a) {  // :188-202,230-236
  if (base_index == 0) return (double)reverse_bits_64(a) * 0.00000000000000000005421010862427522;
  uint64_t base = primes().p[base_index];
  double inv_base = 1.0 / (double)base, inv_base_n = 1.0;
  uint64_t reversed = 0;
  while (a != 0) {
    uint64_t next = a / base, digit = a - next * base;
    reversed = reversed * base + digit;
    inv_base_n *= inv_base;
    a = next;
  }
  return rmin((double)reversed * inv_base_n, ONE_MINUS_EPSILON);
}
double scrambled_radical_inverse(int base_index, uint64_t a, const uint16_t* perm) {  // :204-227,272
  uint64_t base = primes().p[base_index];
  double inv_base = 1.0 / (double)base, inv_base_n = 1.0;
  uint64_t reversed = 0;
  while (a > 0) {
    uint64_t next = a / base, digit = a - next * base;
    reversed = reversed * base + perm[digit];
    inv_base_n *= inv_base;
    a = next;
  }
  return rmin(inv_base_n * ((double)reversed + inv_base * (double)perm[0] / (1.0 - inv_base)), ONE_MINUS_EPSILON);
}
uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, uint64_t n_digits) {  // :239-248
  uint64_t index = 0;
  for (uint64_t i = 0; i < n_digits; i++) {
    uint64_t digit = inverse % base;
    inverse /= base;
    index = index * base + digit;
  }
  return index;
}

struct Sampler {  // GlobalSampler<Halton>, samplers/mod.rs:266-447 with array_start_dim = array_end_dim = 0
  const rrt_sampler* h;
  int64_t px = 0, py = 0;
  int64_t pixel_for_offset[2] = {0, 0};
  uint64_t offset_for_current_pixel = 0;
  uint64_t current_pixel_sample_index = 0;
  uint64_t interval_sample_index = 0;
  uint32_t dimension = 0;

  // get_index_for_sample halton.rs:75-105 (dim-0 digit count uses base_exponents[1]: Q24)
  uint64_t get_index_for_sample(uint64_t sample_num) {
    if (px != pixel_for_offset[0] || py != pixel_for_offset[1]) {
      offset_for_current_pixel = 0;
      if (h->sample_stride > 1) {
        auto mod = [](int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; };  // misc.rs:334-349
        int64_t pm[2] = {mod(px, 128), mod(py, 128)};
        for (int i = 0; i < 2; i++) {
          uint64_t dim_offset = (i == 0) ? inverse_radical_inverse(2, (uint64_t)pm[i], (uint64_t)h->base_exponents[1])
                                         : inverse_radical_inverse(3, (uint64_t)pm[i], (uint64_t)h->base_exponents[i]);
          offset_for_current_pixel += dim_offset * (h->sample_stride / (uint64_t)h->base_scales[i]) * h->mult_inverse[i];
        }
        offset_for_current_pixel %= h->sample_stride;
      }
      pixel_for_offset[0] = px; pixel_for_offset[1] = py;
    }
    return offset_for_current_pixel + sample_num * h->sample_stride;
  }
  // sample_dimension halton.rs:107-128
  double sample_dimension(uint64_t index, uint32_t dim) const {
    if (h->sample_at_center && (dim == 0 || dim == 1)) return 0.5;
    if (dim == 0) return radical_inverse(0, index >> h->base_exponents[0]);
    if (dim == 1) return radical_inverse(1, index / (uint64_t)h->base_scales[1]);
    if (dim >= 1000) throw OraclePanic{"halton.rs:65 HaltonSampler can only sample 1000 dimensions."};
    return scrambled_radical_inverse((int)dim, index, h->perms + primes().sums[dim]);
  }
  // ---- StratifiedSampler = PixelSampler<Stratified> (samplers/stratified.rs, samplers/mod.rs:147-252) -------------
  // The reference fills per-pixel arrays of jittered strata and shuffles them with rand::thread_rng; dimensions beyond
  // `dimension` return rng.gen_range(-1.0..1.0) (mod.rs:211-226 - note the range). thread_rng cannot be reproduced, so
  // the same structure is driven by counter-based randomness shared with the device: the stratum of
  // sample s in dimension k of pixel P is a keyed bijection of s (the shuffle), its jitter a hash of (key, stratum).
  int xres = 0;
  uint32_t cur1d = 0, cur2d = 0;
  static uint32_t st_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
  uint32_t st_key(uint32_t tag) const {
    const uint32_t pixel = (uint32_t)(py * (int64_t)xres + px);
    return st_mix(st_mix(pixel ^ (uint32_t)h->perm_seed) + tag * 0x9e3779b9u + (uint32_t)(h->perm_seed >> 32));
  }
  static uint32_t st_permute(uint32_t i, uint32_t n, uint32_t p) {   // Kensler 2013
    uint32_t w = n - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
      i ^= p; i *= 0xe170893du; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3fu; i ^= p >> 23; i ^= (i & w) >> 1;
      i *= 1u | p >> 27; i *= 0x6935fa69u; i ^= (i & w) >> 11; i *= 0x74dcb303u; i ^= (i & w) >> 2; i *= 0x9e501cc3u;
      i ^= (i & w) >> 2; i *= 0xc860a3dfu; i &= w; i ^= i >> 5;
    } while (i >= n);
    return (i + p) % n;
  }
  static double st_rand(uint32_t key, uint32_t i) { return (double)st_mix(key ^ st_mix(i + 0x632be5abu)) * 2.3283064365386963e-10; }
  double st_get_1d() {
    const uint32_t k = cur1d & 0xffu, sn = (uint32_t)current_pixel_sample_index;
    cur1d = (k + 1u) & 0xffu;
    if (k >= (uint32_t)h->dimension) return 2.0 * st_rand(st_key(0x10000u + k), sn) - 1.0;          // mod.rs:211-214
    const uint32_t spp = (uint32_t)(h->xsamp * h->ysamp), key = st_key(k);
    const uint32_t j = st_permute(sn, spp, key);                                                      // shuffle, stratified.rs:46-50
    const double delta = h->jitter ? st_rand(key, j) : 0.5;                                           // stratified_sample1d :103-110
    return rmin(((double)j + delta) * (1.0 / (double)spp), ONE_MINUS_EPSILON);
  }
  void st_get_2d(double* a, double* b) {
    const uint32_t k = cur2d & 0xffu, sn = (uint32_t)current_pixel_sample_index;
    cur2d = (k + 1u) & 0xffu;
    if (k >= (uint32_t)h->dimension) {                                                                // mod.rs:222-226
      const uint32_t key = st_key(0x20000u + k);
      *a = 2.0 * st_rand(key, 2u * sn) - 1.0; *b = 2.0 * st_rand(key, 2u * sn + 1u) - 1.0;
      return;
    }
    const uint32_t nx = (uint32_t)h->xsamp, ny = (uint32_t)h->ysamp, key = st_key(0x1000u + k);
    const uint32_t j = st_permute(sn, nx * ny, key), x = j % nx, y = j / nx;                           // stratified_sample2d :112-130
    const double jx = h->jitter ? st_rand(key, 2u * j) : 0.5, jy = h->jitter ? st_rand(key, 2u * j + 1u) : 0.5;
    *a = rmin(((double)x + jx) * (1.0 / (double)nx), ONE_MINUS_EPSILON);
    *b = rmin(((double)y + jy) * (1.0 / (double)ny), ONE_MINUS_EPSILON);
  }

  void start_pixel(int64_t x, int64_t y) {  // samplers/mod.rs:58-65,322-372
    px = x; py = y;
    current_pixel_sample_index = 0;
    dimension = 0; cur1d = cur2d = 0;
    if (h->type == RRT_SAMPLER_HALTON) interval_sample_index = get_index_for_sample(0);
  }
  uint64_t samples_per_pixel() const { return h->samples_per_pixel; }   // samplers/mod.rs:245,439
  bool start_next_sample() {  // :378-386 + BaseSampler::start_next_sample :71-76 (Q1); PixelSampler :195-199
    dimension = 0; cur1d = cur2d = 0;
    if (h->type == RRT_SAMPLER_HALTON) interval_sample_index = get_index_for_sample(current_pixel_sample_index + 1);
    current_pixel_sample_index += 1;
    return current_pixel_sample_index < h->samples_per_pixel;
  }
  double get_1d() {  // :396-409
    if (h->type == RRT_SAMPLER_STRATIFIED) return st_get_1d();
    dimension += 1; return sample_dimension(interval_sample_index, dimension - 1);
  }
  void get_2d(double* a, double* b) {  // :411-433
    if (h->type == RRT_SAMPLER_STRATIFIED) { st_get_2d(a, b); return; }
    *a = sample_dimension(interval_sample_index, dimension);
    *b = sample_dimension(interval_sample_index, dimension + 1);
    dimension += 2;
  }
};

// ---- RealisticCamera per-sample path (camera.rs:156-253,492-628) ---------------------------------------
struct Camera {
  const rrt_camera* c;
  const rrt_film* f;

  static bool refract(V3 wi, V3 n, double eta, V3* wt) {  // reflection.rs:122-134
    double cos_i = dot(n, wi);
    double sin2_i = rmax(0.0, 1.0 - cos_i * cos_i);
    double sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1.0) return false;
    double cos_t = std::sqrt(1.0 - sin2_t);
    *wt = (-wi) * eta + n * (eta * cos_i - cos_t);
    return true;
  }
  static Ray flip_z(const Ray& r) {  // Transform::scale(1,1,-1).t(ray)
    double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1};
    return xf_ray(m, r);
  }
  static bool intersect_spherical(double radius, double z_center, const Ray& ray, double* t, V3* n) {  // :220-253
    V3 o = ray.o - V3(0.0, 0.0, z_center);
    double a = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
    double b = 2.0 * (ray.d.x * o.x + ray.d.y * o.y + ray.d.z * o.z);
    double cc = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    double t0 = 0, t1 = 0;
    if (!quadratic(a, b, cc, &t0, &t1)) return false;
    bool use_closer = (ray.d.z > 0.0) ^ (radius < 0.0);
    *t = use_closer ? rmin(t0, t1) : rmax(t0, t1);
    if (*t < 0.0) return false;
    V3 nn = o + ray.d * *t;
    *n = faceforward(nnormalize(nn), -ray.d);
    return true;
  }
  bool trace_from_film(const Ray& r_camera, Ray* out) const {  // :156-219
    double element_z = 0.0;
    Ray r = flip_z(r_camera);
    for (int i = c->n_elems - 1; i >= 0; i--) {
      const rrt_lens_elem& el = c->elems[i];
      element_z -= el.thickness;
      double t = 0.0;
      V3 n;
      bool is_stop = el.curvature_radius == 0.0;
      if (is_stop) {
        if (r.d.z >= 0.0) return false;
        t = (element_z - r.o.z) / r.d.z;
      } else {
        if (!intersect_spherical(el.curvature_radius, element_z + el.curvature_radius, r, &t, &n)) return false;
      }
      if (!(t >= 0.0)) throw OraclePanic{"camera.rs:186 assert!(t >= 0)"};
      V3 p_hit = ray_at(r, t);
      double r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
      if (r2 >= el.aperture_radius * el.aperture_radius) return false;
      r.o = p_hit;
      if (!is_stop) {
        V3 w;
        double eta_i = el.eta;
        double eta_t = (i > 0 && c->elems[i - 1].eta != 0.0) ? c->elems[i - 1].eta : 1.0;
        if (!refract(vnormalize(-r.d), n, eta_i / eta_t, &w)) return false;
        r.d = w;
      }
    }
    *out = flip_z(r);
    return true;
  }
  // sample_exit_pupil :492-521 (Q6: cast happens before the multiply)
  V3 sample_exit_pupil(double fx, double fy, double lx, double ly, double* area) const {
    double r_film = std::sqrt(fx * fx + fy * fy);
    double q = r_film / (f->diagonal / 2.0);
    size_t qi = (q != q || q <= 0.0) ? 0 : (q >= 1.8e19 ? ~(size_t)0 : (size_t)q);
    size_t r_index = qi * 64;
    r_index = std::min(r_index, (size_t)63);
    if (!c->exit_pupil_valid[r_index]) throw OraclePanic{"exit_pupil_bounds slab not computed by the host"};
    const double* pb = c->exit_pupil_bounds[r_index];
    double plx = pb[0] * (1.0 - lx) + pb[2] * lx, ply = pb[1] * (1.0 - ly) + pb[3] * ly;  // Bounds2f::lerp
    double sin_t = r_film != 0.0 ? fy / r_film : 0.0, cos_t = r_film != 0.0 ? fx / r_film : 1.0;
    *area = (pb[2] - pb[0]) * (pb[3] - pb[1]);
    return {cos_t * plx - sin_t * ply, sin_t * plx + cos_t * ply, c->elems[c->n_elems - 1].thickness};
  }
  // generate_ray :534-580
  double generate_ray(double pfx, double pfy, double lx, double ly, Ray* ray) const {
    double sx = pfx / (double)f->xres, sy = pfy / (double)f->yres;
    const double* pe = f->physical_extent;
    double p2x = pe[0] * (1.0 - sx) + pe[2] * sx, p2y = pe[1] * (1.0 - sy) + pe[3] * sy;
    V3 p_film(-p2x, p2y, 0.0);
    double area;
    V3 p_rear = sample_exit_pupil(p_film.x, p_film.y, lx, ly, &area);
    Ray r_film = ray_new(p_film, p_rear - p_film, INF);
    Ray r;
    if (!trace_from_film(r_film, &r)) return 0.0;
    r = xf_ray(c->camera_to_world.m, r);
    r.d = vnormalize(r.d);
    *ray = r;
    double cos_theta = vnormalize(r_film.d).z;
    double cos4 = (cos_theta * cos_theta) * (cos_theta * cos_theta);
    if (c->simple_weighting) {
      const double* b0 = c->exit_pupil_bounds[0];
      return cos4 * area / ((b0[2] - b0[0]) * (b0[3] - b0[1]));
    }
    double rz = c->elems[c->n_elems - 1].thickness;
    return (c->shutter_close - c->shutter_open) * (cos4 * area) / rz * rz;
  }
  // generate_ray_differential :582-628
  double generate_ray_differential(double pfx, double pfy, double lx, double ly, Ray* ray, RayDiff* rd = nullptr) const {
    double wt = generate_ray(pfx, pfy, lx, ly, ray);
    if (wt == 0.0) return 0.0;
    RayDiff d;
    double wtx = 0.0;
    for (double eps : {0.05, -0.05}) {
      Ray rx;
      wtx = generate_ray(pfx + eps, pfy, lx, ly, &rx);
      d.rxo = ray->o + (rx.o - ray->o) / eps;
      d.rxd = ray->d + (rx.d - ray->d) / eps;
      if (wtx != 0.0) break;
    }
    if (wtx == 0.0) return 0.0;
    double wty = 0.0;
    for (double eps : {0.05, -0.05}) {
      Ray ry;
      wty = generate_ray(pfx, pfy + eps, lx, ly, &ry);
      d.ryo = ray->o + (ry.o - ray->o) / eps;
      d.ryd = ray->d + (ry.d - ray->d) / eps;
      if (wty != 0.0) break;
    }
    if (wty == 0.0) return 0.0;
    d.has = true;
    if (rd) *rd = d;
    return wt;
  }
};
// RayDifferential::scale_differentials geometry.rs:1883-1888
inline void scale_differentials(const Ray& r, RayDiff* d, double s) {
  d->rxo = r.o + (d->rxo - r.o) * s;
  d->ryo = r.o + (d->ryo - r.o) * s;
  d->rxd = r.d + (d->rxd - r.d) * s;
  d->ryd = r.d + (d->ryd - r.d) * s;
}

// ---- BVH traversal (bvh.rs:124-236) over primitives.rs wrappers ----------------------------------------
struct Counters { uint64_t nodes = 0, prims = 0, closest = 0, any = 0; };

// Bounds3::intersect_p geometry.rs:1767-1800
// `margin` (optional) tracks the smallest relative gap of the two final comparisons over a traversal: rays
// whose result depends on a gap of a few ulps are "ties" (e.g. a hit on a face coplanar with a flat leaf box).
inline bool box_intersect_p(const double* b, const Ray& ray, V3 inv_dir, const int dir_is_neg[3], double* margin = nullptr) {
  double t_min = (b[dir_is_neg[0] * 3 + 0] - ray.o.x) * inv_dir.x;
  double t_max = (b[(1 - dir_is_neg[0]) * 3 + 0] - ray.o.x) * inv_dir.x;
  double ty_min = (b[dir_is_neg[1] * 3 + 1] - ray.o.y) * inv_dir.y;
  double ty_max = (b[(1 - dir_is_neg[1]) * 3 + 1] - ray.o.y) * inv_dir.y;
  t_max *= 1.0 + 2.0 * gamma_n(3);
  ty_max *= 1.0 + 2.0 * gamma_n(3);
  if (t_min > ty_max || ty_min > t_max) return false;
  if (ty_min > t_min) t_min = ty_min;
  if (ty_max < t_max) t_max = ty_max;
  double tz_min = (b[dir_is_neg[2] * 3 + 2] - ray.o.z) * inv_dir.z;
  double tz_max = (b[(1 - dir_is_neg[2]) * 3 + 2] - ray.o.z) * inv_dir.z;
  tz_max *= 1.0 + 2.0 * gamma_n(3);
  if (t_min > tz_max || tz_min > t_max) return false;
  if (tz_min > t_min) t_min = tz_min;
  if (tz_max < t_max) t_max = tz_max;
  if (margin) {
    if (std::isfinite(ray.t_max)) *margin = rmin(*margin, std::fabs(t_min - ray.t_max) / rmax(std::fabs(ray.t_max), 1e-300));
    *margin = rmin(*margin, std::fabs(t_max) / rmax(std::fabs(t_min), 1.0));
  }
  return (t_min < ray.t_max) && (t_max > 0.0);
}

// GeometricPrimitive / TransformedPrimitive ::intersect primitives.rs:51-68,124-139
bool prim_intersect(const Scene& sc, uint32_t pi, Ray* r, SI* si, double* bu, double* bv) {
  const rrt_prim& p = sc.d->prims[pi];
  auto geometric = [&](Ray* rr) {
    double t_hit = 0.0;
    if (p.type == RRT_PRIM_TRIANGLE) { if (!tri_intersect(sc, sc.d->tris[p.shape], *rr, &t_hit, si, bu, bv)) return false; }
    else { *bu = 0; *bv = 0; if (!sphere_intersect(sphere_ref(sc, p.shape), *rr, &t_hit, si)) return false; }
    si->prim = (int)pi;
    si->valid = true;
    rr->t_max = t_hit;
    if (!(dot(si->n, si->sn) >= 0.0)) throw OraclePanic{"primitives.rs:66 assert!(dot3(&si.ist.n, &si.shading.n) >= 0.0)"};
    return true;
  };
  if (p.instance < 0) return geometric(r);
  const rrt_xform& x = sc.d->xforms[p.instance];
  if (sc.flat && p.type == RRT_PRIM_TRIANGLE && is_rigid(x.m)) {
    double t_hit = 0.0;
    if (!tri_intersect(sc, sc.d->tris[p.shape], *r, &t_hit, si, bu, bv, &x)) return false;
    si->prim = (int)pi; si->valid = true;
    r->t_max = t_hit;
    if (!(dot(si->n, si->sn) >= 0.0)) throw OraclePanic{"primitives.rs:66 assert!(dot3(&si.ist.n, &si.shading.n) >= 0.0)"};
    return true;
  }
  Ray ray = xf_ray(x.m_inv, *r);  // world_to_prim = inverse(primitive_to_world)
  if (!geometric(&ray)) return false;
  r->t_max = ray.t_max;           // copied across spaces (Q15)
  if (!is_identity(x.m)) si_transform(si, x.m, x.m_inv);
  return true;
}
bool prim_intersect_p(const Scene& sc, uint32_t pi, const Ray& r) {  // primitives.rs:41-45,117-122
  const rrt_prim& p = sc.d->prims[pi];
  auto geometric = [&](const Ray& rr) {
    return p.type == RRT_PRIM_TRIANGLE ? tri_intersect_p(sc, sc.d->tris[p.shape], rr) : sphere_intersect_p(sphere_ref(sc, p.shape), rr);
  };
  if (p.instance < 0) return geometric(r);
  if (sc.flat && p.type == RRT_PRIM_TRIANGLE && is_rigid(sc.d->xforms[p.instance].m)) return tri_intersect_p(sc, sc.d->tris[p.shape], r, &sc.d->xforms[p.instance]);
  return geometric(xf_ray(sc.d->xforms[p.instance].m_inv, r));
}

struct HitInfo { int order_index = -1; double u = 0, v = 0; };

// BVHAccel::intersect bvh.rs:183-236. The reference's nodes_to_visit is [usize; 64]; deeper trees index
// out of bounds there (panic) — reported as such unless the tree was built with the fixed builder.
bool scene_intersect(const Scene& sc, Ray* r, SI* si, HitInfo* hi, Counters* cnt, uint32_t* nodes_c = nullptr, uint32_t* prims_c = nullptr, double* margin = nullptr) {
  if (len(r->d) == 0.0) throw OraclePanic{"scene.rs:70 assert_ne!(r.d.length(), 0.0)"};
  if (cnt) cnt->closest++;
  bool hit = false;
  const rrt_scene_desc* d = sc.d;
  if (d->n_bvh_nodes == 0) return false;
  V3 inv_dir(1.0 / r->d.x, 1.0 / r->d.y, 1.0 / r->d.z);
  int dir_is_neg[3] = {inv_dir.x < 0.0, inv_dir.y < 0.0, inv_dir.z < 0.0};
  uint32_t stack_small[128];
  std::vector<uint32_t> stack_big;
  uint32_t* stack = stack_small;
  if (d->bvh_depth + 2 > 128) { stack_big.resize(d->bvh_depth + 2); stack = stack_big.data(); }
  size_t to_visit = 0;
  uint32_t cur = 0;
  uint32_t nn = 0, np = 0;
  const bool strict64 = !(d->flags & RRT_FIXED_BVH);
  while (true) {
    const rrt_bvh_node& node = d->bvh_nodes[cur];
    nn++;
    if (box_intersect_p(node.bounds, *r, inv_dir, dir_is_neg, margin)) {
      if (node.n_primitives > 0) {
        for (uint32_t i = 0; i < node.n_primitives; i++) {
          np++;
          double bu, bv;
          if (prim_intersect(sc, d->prim_order[node.offset + i], r, si, &bu, &bv)) {
            hit = true;
            if (hi) { hi->order_index = (int)(node.offset + i); hi->u = bu; hi->v = bv; }
          }
        }
        if (to_visit == 0) break;
        cur = stack[--to_visit];
      } else {
        if (strict64 && to_visit >= 64) throw OraclePanic{"bvh.rs:217 nodes_to_visit[64] index out of bounds (tree deeper than 64)"};
        if (dir_is_neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
        else { stack[to_visit++] = node.offset; cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = stack[--to_visit];
    }
  }
  if (cnt) { cnt->nodes += nn; cnt->prims += np; }
  if (nodes_c) *nodes_c = nn;
  if (prims_c) *prims_c = np;
  return hit;
}
// BVHAccel::intersect_p bvh.rs:124-173
bool scene_intersect_p(const Scene& sc, const Ray& r, Counters* cnt, uint32_t* nodes_c = nullptr, uint32_t* prims_c = nullptr, double* margin = nullptr) {
  if (len(r.d) == 0.0) throw OraclePanic{"scene.rs:77 assert_ne!(r.d.length(), 0.0)"};
  if (cnt) cnt->any++;
  const rrt_scene_desc* d = sc.d;
  if (d->n_bvh_nodes == 0) return false;
  V3 inv_dir(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
  int dir_is_neg[3] = {inv_dir.x < 0.0, inv_dir.y < 0.0, inv_dir.z < 0.0};
  uint32_t stack_small[128];
  std::vector<uint32_t> stack_big;
  uint32_t* stack = stack_small;
  if (d->bvh_depth + 2 > 128) { stack_big.resize(d->bvh_depth + 2); stack = stack_big.data(); }
  size_t to_visit = 0;
  uint32_t cur = 0, nn = 0, np = 0;
  bool result = false;
  const bool strict64 = !(d->flags & RRT_FIXED_BVH);
  while (true) {
    const rrt_bvh_node& node = d->bvh_nodes[cur];
    nn++;
    if (box_intersect_p(node.bounds, r, inv_dir, dir_is_neg, margin)) {
      if (node.n_primitives > 0) {
        for (uint32_t i = 0; i < node.n_primitives; i++) {
          np++;
          if (prim_intersect_p(sc, d->prim_order[node.offset + i], r)) { result = true; break; }
        }
        if (result) break;
        if (to_visit == 0) break;
        cur = stack[--to_visit];
      } else {
        if (strict64 && to_visit >= 64) throw OraclePanic{"bvh.rs:154 nodes_to_visit[64] index out of bounds (tree deeper than 64)"};
        if (dir_is_neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
        else { stack[to_visit++] = node.offset; cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = stack[--to_visit];
    }
  }
  if (cnt) { cnt->nodes += nn; cnt->prims += np; }
  if (nodes_c) *nodes_c = nn;
  if (prims_c) *prims_c = np;
  return result;
}

// ---- BxDFs / Bsdf (reflection.rs, microfacet.rs) ------------------------------------------------------
enum : uint8_t { BXDF_REFLECTION = 1, BXDF_TRANSMISSION = 2, BXDF_DIFFUSE = 4, BXDF_GLOSSY = 8, BXDF_SPECULAR = 16, BXDF_ALL = 31, BXDF_NONE = 0 };
enum LobeKind { LAMBERT, OREN_NAYAR, MICROFACET, SPEC_REFL, DEBUG_DIFFUSE, DEBUG_SPECULAR, SPEC_TRANS, FRESNEL_SPEC, LAMBERT_TRANS, MICROFACET_TRANS };
enum FresnelKind { FR_NOOP, FR_DIELECTRIC, FR_CONDUCTOR };

inline double cos_theta(V3 w) { return w.z; }
inline double cos2_theta(V3 w) { return w.z * w.z; }
inline double abs_cos_theta(V3 w) { return std::fabs(w.z); }
inline double sin2_theta(V3 w) { return rmax(0.0, 1.0 - cos2_theta(w)); }
inline double sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
inline double tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline double tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
inline double cos_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 1.0 : clampd(w.x / s, -1.0, 1.0); }
inline double sin_phi(V3 w) { double s = sin_theta(w); return s == 0.0 ? 0.0 : clampd(w.y / s, -1.0, 1.0); }
inline double cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
inline double sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0; }
inline V3 reflect(V3 wo, V3 n) { return -wo + n * 2.0 * dot(wo, n); }  // reflection.rs:115-117

double fr_dielectric(double cos_i, double eta_i, double eta_t) {  // reflection.rs:145-168
  cos_i = clampd(cos_i, -1.0, 1.0);
  bool entering = cos_i > 0.0;
  if (!entering) { std::swap(eta_i, eta_t); cos_i = std::fabs(cos_i); }
  double sin_i = std::sqrt(rmax(0.0, 1.0 - cos_i * cos_i));
  double sin_t = eta_i / eta_t * sin_i;
  if (sin_t >= 1.0) return 1.0;
  double cos_t = std::sqrt(rmax(0.0, 1.0 - sin_t * sin_t));
  double r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
  double r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
  return (r_parl * r_parl + r_perp * r_perp) / 2.0;
}
Rgb fr_conductor(double cos_i_in, Rgb eta_i, Rgb eta_t, Rgb k) {  // reflection.rs:170-195
  double cos_i = clampd(cos_i_in, -1.0, 1.0);
  Rgb eta = eta_t / eta_i, eta_k = k / eta_i;
  double cos2 = cos_i * cos_i, sin2 = 1.0 - cos2;
  Rgb eta2 = eta * eta, eta_k2 = eta_k * eta_k;
  Rgb t0 = eta2 - eta_k2 - Rgb(sin2);
  Rgb a2_plus_b2 = rsqrt(t0 * t0 + eta2 * eta_k2 * Rgb(4.0));
  Rgb t1 = a2_plus_b2 + Rgb(cos2);
  Rgb a = rsqrt((a2_plus_b2 + t0) * 0.5);
  Rgb t2 = a * 2.0 * cos_i;
  Rgb rs = (t1 - t2) / (t1 + t2);
  Rgb t3 = a2_plus_b2 * cos2 + Rgb(sin2 * sin2);
  Rgb t4 = t2 * sin2;
  Rgb rp = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) * Rgb(0.5);
}

struct Lobe {
  LobeKind kind;
  uint8_t type;
  Rgb r;
  double a = 0, b = 0;               // OrenNayar A, B
  double alpha_x = 0, alpha_y = 0;   // TrowbridgeReitz (sample_visible_area = true)
  FresnelKind fr = FR_NOOP;
  Rgb eta_i, eta_t, k;               // conductor;  dielectric uses eta_i.c[0], eta_t.c[0]
  Rgb t;                             // transmission colour (FresnelSpecular carries r and t)
  double eta_a = 1.0, eta_b = 1.0;   // transmissive lobes (always 1.0 / material eta)
};
inline bool bsdf_refract(V3 wi, V3 n, double eta, V3* wt) {  // reflection.rs:122-134
  double cos_i = dot(n, wi);
  double sin2_i = rmax(0.0, 1.0 - cos_i * cos_i);
  double sin2_t = eta * eta * sin2_i;
  if (sin2_t >= 1.0) return false;
  double cos_t = std::sqrt(1.0 - sin2_t);
  *wt = (-wi) * eta + n * (eta * cos_i - cos_t);
  return true;
}

// TrowbridgeReitzDistribution microfacet.rs:253-425
double tr_d(const Lobe& l, V3 wh) {
  double tan2 = tan2_theta(wh);
  if (std::isinf(tan2)) return 0.0;
  double cos4 = cos2_theta(wh) * cos2_theta(wh);
  double e = (cos2_phi(wh) / (l.alpha_x * l.alpha_x) + sin2_phi(wh) / (l.alpha_y * l.alpha_y)) * tan2;
  return 1.0 / (PI * l.alpha_x * l.alpha_y * cos4 * (1.0 + e) * (1.0 + e));
}
double tr_lambda(const Lobe& l, V3 w) {
  double abs_tan = std::fabs(tan_theta(w));
  if (std::isinf(abs_tan)) return 0.0;
  double alpha = std::sqrt(cos2_phi(w) * (l.alpha_x * l.alpha_x) + sin2_phi(w) * (l.alpha_y * l.alpha_y));
  double a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
  return (-1.0 + std::sqrt(1.0 + a2t2)) / 2.0;
}
double tr_g1(const Lobe& l, V3 w) { return 1.0 / (1.0 + tr_lambda(l, w)); }
double tr_g(const Lobe& l, V3 wo, V3 wi) { return 1.0 / (1.0 + tr_lambda(l, wo) + tr_lambda(l, wi)); }
double tr_pdf(const Lobe& l, V3 wo, V3 wh) { return tr_d(l, wh) * tr_g1(l, wo) * absdot(wo, wh) / abs_cos_theta(wo); }  // microfacet.rs:30-36
void tr_sample_11(double cos_t, double u1, double u2, double* slope_x, double* slope_y) {  // :268-323
  if (cos_t > 0.9999) {
    double r = std::sqrt(u1 / (1.0 - u1));
    double phi = 6.28318530718 * u2;
    *slope_x = r * std::cos(phi);
    *slope_y = r * std::sin(phi);
    return;
  }
  double sin_t = std::sqrt(rmax(0.0, 1.0 - cos_t * cos_t));
  double tan_t = sin_t / cos_t;
  double a = 1.0 / tan_t;
  double g1 = 2.0 / (1.0 + std::sqrt(1.0 + 1.0 / (a * a)));
  a = 2.0 * u1 / g1 - 1.0;
  double tmp = 1.0 / (a * a - 1.0);
  if (tmp > 1e10) tmp = 1e10;
  double b = tan_t;
  double d = std::sqrt(rmax(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0));
  double sx1 = b * tmp - d, sx2 = b * tmp + d;
  *slope_x = (a < 0.0 || sx2 > 1.0 / tan_t) ? sx1 : sx2;
  double s, nu2;
  if (u2 > 0.5) { s = 1.0; nu2 = 2.0 * (u2 - 0.5); } else { s = -1.0; nu2 = 2.0 * (0.5 - u2); }
  double z = (nu2 * (nu2 * (nu2 * 0.27385 - 0.73369) + 0.46341)) / (nu2 * (nu2 * (nu2 * 0.093073 + 0.309420) - 1.0) + 0.597999);
  *slope_y = s * z * std::sqrt(1.0 + *slope_x * *slope_x);
  if (std::isinf(*slope_y) || *slope_y != *slope_y) throw OraclePanic{"microfacet.rs:321 assert!(!slope_y.is_infinite()/is_nan())"};
}
V3 tr_sample(V3 wi, double ax, double ay, double u1, double u2) {  // :325-363
  V3 ws = vnormalize(V3(ax * wi.x, ay * wi.y, wi.z));
  double sx = 0, sy = 0;
  tr_sample_11(cos_theta(ws), u1, u2, &sx, &sy);
  double tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
  sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
  sx = tmp;
  sx *= ax; sy *= ay;
  return vnormalize(V3(-sx, -sy, 1.0));
}
V3 tr_sample_wh(const Lobe& l, V3 wo, double u0, double u1) {  // :387-421 (sample_visible_area branch)
  bool flip = wo.z < 0.0;
  if (flip) return -tr_sample(-wo, l.alpha_x, l.alpha_y, u0, u1);
  return tr_sample(wo, l.alpha_x, l.alpha_y, u0, u1);
}

Rgb fresnel_eval(const Lobe& l, double cos_i) {  // reflection.rs:599-615
  if (l.fr == FR_NOOP) return Rgb(1.0);
  if (l.fr == FR_DIELECTRIC) return Rgb(fr_dielectric(cos_i, l.eta_i.c[0], l.eta_t.c[0]));
  return fr_conductor(std::fabs(cos_i), l.eta_i, l.eta_t, l.k);
}

Rgb lobe_f(const Lobe& l, V3 wo, V3 wi) {
  switch (l.kind) {
    case LAMBERT: return l.r / PI;  // reflection.rs:818-820
    case OREN_NAYAR: {              // :917-941
      double sin_i = sin_theta(wi), sin_o = sin_theta(wo), max_cos = 0.0;
      if (sin_i > 1e-4 && sin_o > 1e-4) {
        double d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
        max_cos = rmax(d_cos, 0.0);
      }
      double sin_alpha, tan_beta;
      if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_o; tan_beta = sin_i / abs_cos_theta(wi); }
      else { sin_alpha = sin_i; tan_beta = sin_o / abs_cos_theta(wo); }
      return l.r / PI * (l.a + l.b * max_cos * sin_alpha * tan_beta);
    }
    case MICROFACET: {              // :971-992
      double cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
      V3 wh = wi + wo;
      if (cos_i == 0.0 || cos_o == 0.0) return Rgb();
      if (wh.x == 0.0 && wh.y == 0.0 && wh.z == 0.0) return Rgb();
      wh = vnormalize(wh);
      Rgb f = fresnel_eval(l, dot(wi, faceforward(wh, V3(0.0, 0.0, 1.0))));
      return l.r * tr_d(l, wh) * tr_g(l, wo, wi) * f / (4.0 * cos_i * cos_o);
    }
    case SPEC_REFL: return Rgb();   // :635-637
    case SPEC_TRANS: case FRESNEL_SPEC: return Rgb();   // :682-684, :750-752
    case LAMBERT_TRANS: return l.t / PI;                 // :854-856
    case MICROFACET_TRANS: {                             // :1059-1097 (mode = Radiance)
      if (same_hemisphere(wo, wi)) return Rgb();
      double cos_o = cos_theta(wo), cos_i = cos_theta(wi);
      if (cos_i == 0.0 || cos_o == 0.0) return Rgb();
      double eta = cos_theta(wo) > 0.0 ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
      V3 wh = vnormalize(wo + wi * eta);
      if (wh.z < 0.0) wh = -wh;
      double f = fr_dielectric(dot(wo, wh), l.eta_a, l.eta_b);
      double sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
      double factor = 1.0 / eta;
      return (Rgb(1.0) - Rgb(f)) * l.t *
             std::fabs(tr_d(l, wh) * tr_g(l, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cos_i * cos_o * sqrt_denom * sqrt_denom));
    }
    case DEBUG_DIFFUSE: return Rgb(0.0, 1.0, 0.0);   // debug_material.rs:13-15
    case DEBUG_SPECULAR: return Rgb(0.0, 0.0, 1.0);  // debug_material.rs:25-27
  }
  return Rgb();
}
double lobe_pdf(const Lobe& l, V3 wo, V3 wi) {
  if (l.kind == MICROFACET) {  // :1019-1025
    if (!same_hemisphere(wo, wi)) return 0.0;
    V3 wh = vnormalize(wo + wi);
    return tr_pdf(l, wo, wh) / (4.0 * dot(wo, wh));
  }
  if (l.kind == SPEC_REFL || l.kind == SPEC_TRANS || l.kind == FRESNEL_SPEC) return 0.0;
  if (l.kind == LAMBERT_TRANS) return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) / PI : 0.0;  // :887-893
  if (l.kind == MICROFACET_TRANS) {  // :1124-1144
    if (same_hemisphere(wo, wi)) return 0.0;
    double eta = cos_theta(wo) > 0.0 ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
    V3 wh = vnormalize(wo + wi * eta);
    double sqrt_denom = dot(wo, wh) + dot(wi, wh) * eta;
    double dwh_dwi = std::fabs((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
    return tr_pdf(l, wo, wh) * dwh_dwi;
  }
  return same_hemisphere(wo, wi) ? abs_cos_theta(wi) / PI : 0.0;  // BxDF::pdf default :492-498
}
Rgb lobe_sample_f(const Lobe& l, V3 wo, V3* wi, double u0, double u1, double* pdf, uint8_t* sampled_type = nullptr) {
  if (l.kind == SPEC_TRANS || l.kind == FRESNEL_SPEC) {  // :690-716, :754-795 (mode = Radiance)
    double fr = 0.0;
    if (l.kind == FRESNEL_SPEC) {
      fr = fr_dielectric(cos_theta(wo), l.eta_a, l.eta_b);
      if (u0 < fr) {
        *wi = V3(-wo.x, -wo.y, wo.z);
        if (sampled_type) *sampled_type = BXDF_SPECULAR | BXDF_REFLECTION;
        *pdf = fr;
        return l.r * fr / abs_cos_theta(*wi);
      }
    }
    bool entering = cos_theta(wo) > 0.0;
    double eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
    if (!bsdf_refract(wo, faceforward(V3(0.0, 0.0, 1.0), wo), eta_i / eta_t, wi)) return Rgb();
    Rgb ft;
    if (l.kind == FRESNEL_SPEC) { ft = l.t * (1.0 - fr); *pdf = 1.0 - fr; if (sampled_type) *sampled_type = BXDF_SPECULAR | BXDF_TRANSMISSION; }
    else { ft = l.t * (Rgb(1.0) - Rgb(fr_dielectric(cos_theta(*wi), l.eta_a, l.eta_b))); *pdf = 1.0; }
    ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
    return ft / abs_cos_theta(*wi);
  }
  if (l.kind == LAMBERT_TRANS) {  // :857-869
    *wi = cosine_sample_hemisphere(u0, u1);
    if (wo.z > 0.0) wi->z *= -1.0;
    *pdf = lobe_pdf(l, wo, *wi);
    return lobe_f(l, wo, *wi);
  }
  if (l.kind == MICROFACET_TRANS) {  // :1098-1123
    if (wo.z == 0.0) return Rgb();
    V3 wh = tr_sample_wh(l, wo, u0, u1);
    if (dot(wo, wh) < 0.0) return Rgb();
    double eta = cos_theta(wo) > 0.0 ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
    if (!bsdf_refract(wo, wh, eta, wi)) return Rgb();
    *pdf = lobe_pdf(l, wo, *wi);
    return lobe_f(l, wo, *wi);
  }
  if (l.kind == MICROFACET) {  // :993-1018
    if (wo.z == 0.0) return Rgb();
    V3 wh = tr_sample_wh(l, wo, u0, u1);
    if (dot(wo, wh) < 0.0) return Rgb();
    *wi = reflect(wo, wh);
    if (!same_hemisphere(wo, *wi)) return Rgb();
    *pdf = tr_pdf(l, wo, wh) / (4.0 * dot(wo, wh));
    return lobe_f(l, wo, *wi);
  }
  if (l.kind == SPEC_REFL) {   // :639-650
    *wi = V3(-wo.x, -wo.y, wo.z);
    *pdf = 1.0;
    return fresnel_eval(l, cos_theta(*wi)) * l.r / abs_cos_theta(*wi);
  }
  // BxDF::sample_f default :427-443 (also used by both Debug lobes)
  *wi = cosine_sample_hemisphere(u0, u1);
  if (wo.z < 0.0) wi->z *= -1.0;
  *pdf = lobe_pdf(l, wo, *wi);
  return lobe_f(l, wo, *wi);
}

double roughness_to_alpha(double roughness) {  // microfacet.rs:12-20
  roughness = rmax(roughness, 1e-3);
  double x = std::log(roughness);
  return 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * x * x * x + 0.000640711 * x * x * x * x;
}

struct Bsdf {  // reflection.rs:205-405
  V3 ns, ng, ss, ts;
  Lobe lobes[8];
  int n = 0;
  bool present = false;
  double eta = 1.0;   // Bsdf::new(si, eta): only glass / translucent pass something else than 1

  void init(const SI& si) {  // Bsdf::new :215-226
    ns = si.sn; ss = vnormalize(si.sdpdu); ng = si.n; ts = cross(ns, ss); n = 0; present = true;
  }
  void add(const Lobe& l) { if (n >= 8) throw OraclePanic{"reflection.rs:228 assert!(self.bxdfs.len() < MAX_BXDFS)"}; lobes[n++] = l; }
  static bool match(const Lobe& l, uint8_t flags) { return (l.type & flags) == l.type; }
  int num_components(uint8_t flags) const { int c = 0; for (int i = 0; i < n; i++) if (match(lobes[i], flags)) c++; return c; }
  V3 to_local(V3 v) const { return {dot(v, ss), dot(v, ts), dot(v, ns)}; }
  V3 to_world(V3 v) const { return {ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z}; }
  Rgb f(V3 wo_w, V3 wi_w, uint8_t flags) const {  // :252-268
    V3 wi = to_local(wi_w), wo = to_local(wo_w);
    if (wo.z == 0.0) return Rgb();
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0;
    Rgb r;
    for (int i = 0; i < n; i++) {
      const Lobe& l = lobes[i];
      if (match(l, flags) && ((refl && (l.type & BXDF_REFLECTION)) || (!refl && (l.type & BXDF_TRANSMISSION)))) r = r + lobe_f(l, wo, wi);
    }
    return r;
  }
  double pdf(V3 wo_w, V3 wi_w, uint8_t flags) const {  // :382-404
    if (n == 0) return 0.0;
    V3 wo = to_local(wo_w), wi = to_local(wi_w);
    if (wo.z == 0.0) return 0.0;
    double p = 0.0;
    int matching = 0;
    for (int i = 0; i < n; i++) if (match(lobes[i], flags)) { matching++; p += lobe_pdf(lobes[i], wo, wi); }
    return matching > 0 ? p / (double)matching : 0.0;
  }
  // sample_f :302-381 (returns only the chosen lobe's f; other pdfs added only for non-reflective lobes: Q21)
  Rgb sample_f(V3 wo_w, V3* wi_w, double u0, double u1, double* pdf_out, uint8_t flags, uint8_t* sampled) const {
    int matching = num_components(flags);
    if (matching == 0) { *pdf_out = 0.0; *sampled = BXDF_NONE; return Rgb(); }
    double fl = std::floor(u0 * (double)matching);
    size_t comp = (fl != fl || fl <= 0.0) ? 0 : (size_t)fl;
    comp = std::min(comp, (size_t)matching);
    int count = (int)comp, chosen = -1;
    for (int i = 0; i < n; i++)
      if (match(lobes[i], flags)) { if (count == 0) { chosen = i; break; } count--; }
    if (chosen < 0) throw OraclePanic{"reflection.rs:337 Did not Choose Any BxDF"};
    const Lobe& bx = lobes[chosen];
    double ur0 = rmin(u0 * (double)matching - (double)comp, ONE_MINUS_EPSILON), ur1 = u1;
    V3 wi, wo = to_local(wo_w);
    if (wo.z == 0.0) return Rgb();  // pdf / sampled_type are left as the caller initialised them
    *pdf_out = 0.0;
    *sampled = bx.type;
    Rgb f = lobe_sample_f(bx, wo, &wi, ur0, ur1, pdf_out, sampled);
    if (*pdf_out == 0.0) { *sampled = BXDF_NONE; return Rgb(); }
    *wi_w = to_world(wi);
    if (!(bx.type & BXDF_REFLECTION) && matching > 1)
      for (int i = 0; i < n; i++) if (i != chosen && match(lobes[i], flags)) *pdf_out += lobe_pdf(lobes[i], wo, wi);
    if (matching > 1) *pdf_out /= (double)matching;
    return f;
  }
};

// ---- SurfaceInteraction::compute_differentials interaction.rs:223-284 --------------------------------------------
// solve_linear_system_2x2 transform.rs:153-164
inline bool solve_2x2(const double a[2][2], const double b[2], double* x0, double* x1) {
  double det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
  if (std::fabs(det) < 1e-10) return false;
  *x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
  *x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
  if (std::isnan(*x0) || std::isnan(*x1)) return false;
  return true;
}
inline double v3_at(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
void compute_differentials(SI* si, const RayDiff& rd) {
  si->dudx = si->dvdx = si->dudy = si->dvdy = 0.0;
  si->dpdx = si->dpdy = V3();
  if (!rd.has) return;
  double d = dot(si->n, si->p);
  double tx = -(dot(si->n, rd.rxo) - d) / dot(si->n, rd.rxd);
  if (std::isinf(tx) || std::isnan(tx)) return;
  V3 px = rd.rxo + rd.rxd * tx;
  double ty = -(dot(si->n, rd.ryd) - d) / dot(si->n, rd.ryd);   // :234 reads ry_direction where pbrt reads ry_origin
  if (std::isinf(ty) || std::isnan(ty)) return;
  V3 py = rd.ryo + rd.ryd * ty;
  si->dpdx = px - si->p;
  si->dpdy = py - si->p;
  int dim[2];
  if (std::fabs(si->n.x) > std::fabs(si->n.y) && std::fabs(si->n.x) > std::fabs(si->n.z)) { dim[0] = 1; dim[1] = 2; }
  else if (std::fabs(si->n.y) > std::fabs(si->n.z)) { dim[0] = 0; dim[1] = 2; }
  else { dim[0] = 0; dim[1] = 1; }
  double a[2][2] = {{v3_at(si->dpdu, dim[0]), v3_at(si->dpdv, dim[0])}, {v3_at(si->dpdu, dim[1]), v3_at(si->dpdv, dim[1])}};
  double bx[2] = {v3_at(px, dim[0]) - v3_at(si->p, dim[0]), v3_at(px, dim[1]) - v3_at(si->p, dim[1])};
  double by[2] = {v3_at(py, dim[0]) - v3_at(si->p, dim[0]), v3_at(py, dim[1]) - v3_at(si->p, dim[1])};
  if (!solve_2x2(a, bx, &si->dudx, &si->dvdx)) { si->dudx = 0.0; si->dvdx = 0.0; }
  if (!solve_2x2(a, by, &si->dudy, &si->dvdy)) { si->dudy = 0.0; si->dvdy = 0.0; }
}

// ---- textures (texture/*.rs) over the flat graph of rrt_scene_desc.textures ---------------------------------------
// Perlin noise, texture/mod.rs:13-160 (NOISE_PERM is Ken Perlin's published reference permutation, doubled)
static const uint8_t kNoisePerm[512] = {
#define RRT_PERLIN_256 \
  151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, \
  247, 120, 234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, \
  74, 165, 71, 134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, \
  65, 25, 63, 161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, \
  52, 217, 226, 250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, \
  119, 248, 152, 2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, \
  218, 246, 97, 228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, \
  184, 84, 204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180
  RRT_PERLIN_256, RRT_PERLIN_256
#undef RRT_PERLIN_256
};
inline double noise_grad(int x, int y, int z, double dx, double dy, double dz) {   // grad :111-130
  int h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z] & 15;
  double u = (h < 8 || h == 12 || h == 13) ? dx : dy;
  double v = (h < 4 || h == 12 || h == 13) ? dy : dz;
  return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
inline double noise_weight(double t) { double t3 = t * t * t, t4 = t3 * t; return 6.0 * t4 * t - 15.0 * t4 + 10.0 * t3; }
inline double lerp_r(double t, double a, double b) { return a * (1.0 - t) + b * t; }   // misc.rs:223-228
inline int f2i_sat(double v) {   // Rust `as i32`: saturating, NaN -> 0
  if (v != v) return 0;
  if (v >= 2147483647.0) return 2147483647;
  if (v <= -2147483648.0) return -2147483647 - 1;
  return (int)v;
}
double noise_flt(double x, double y, double z) {   // :73-105
  int ix = f2i_sat(std::floor(x)), iy = f2i_sat(std::floor(y)), iz = f2i_sat(std::floor(z));
  double dx = x - (double)ix, dy = y - (double)iy, dz = z - (double)iz;
  ix &= 255; iy &= 255; iz &= 255;
  double w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0, dy, dz);
  double w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0, dy - 1.0, dz);
  double w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0, dy, dz - 1.0);
  double w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0, dz - 1.0), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0, dy - 1.0, dz - 1.0);
  double wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
  double x00 = lerp_r(wx, w000, w100), x10 = lerp_r(wx, w010, w110), x01 = lerp_r(wx, w001, w101), x11 = lerp_r(wx, w011, w111);
  double y0 = lerp_r(wy, x00, x10), y1 = lerp_r(wy, x01, x11);
  return lerp_r(wz, y0, y1);
}
inline double smooth_step(double mn, double mx, double v) { double t = clampd((v - mn) / (mx - mn), 0.0, 1.0); return t * t * (-2.0 * t + 3.0); }
double tex_fbm(V3 p, V3 dpdx, V3 dpdy, double omega, int max_octaves) {   // fbm :138-153
  double l2 = rmax(len2(dpdx), len2(dpdy));
  double n = clampd(-1.0 - 0.5 * std::log2(l2), 0.0, (double)max_octaves);
  int n_int = f2i_sat(std::floor(n));
  double sum = 0.0, lambda = 1.0, o = 1.0;
  for (int i = 0; i < n_int; i++) { V3 q = p * lambda; sum += o * noise_flt(q.x, q.y, q.z); lambda *= 1.99; o *= omega; }
  double n_partial = n - (double)n_int;
  V3 q = p * lambda;
  sum += o * smooth_step(0.3, 0.7, n_partial) * noise_flt(q.x, q.y, q.z);
  return sum;
}
double tex_turbulence(V3 p, V3 dpdx, V3 dpdy, double omega, int max_octaves) {   // turbulence :155-185
  double l2 = rmax(len2(dpdx), len2(dpdy));
  double n = clampd(-1.0 - 0.5 * std::log2(l2), 0.0, (double)max_octaves);
  int n_int = f2i_sat(std::floor(n));
  double sum = 0.0, lambda = 1.0, o = 1.0;
  for (int i = 0; i < n_int; i++) { V3 q = p * lambda; sum += o * std::fabs(noise_flt(q.x, q.y, q.z)); lambda *= 1.99; o *= omega; }
  double n_partial = n - (double)n_int;
  V3 q = p * lambda;
  sum += o * lerp_r(smooth_step(0.3, 0.7, n_partial), 0.2, std::fabs(noise_flt(q.x, q.y, q.z)));
  for (int i = n_int; i < max_octaves; i++) { sum += o * 0.2; o *= omega; }
  return sum;
}

// TextureMapping2D::map texture/mod.rs:205-352
inline void map_sphere(const double* m, V3 p, double* s, double* t) {
  V3 v = vnormalize(xf_pt(m, p) - V3());
  double theta = std::acos(clampd(v.z, -1.0, 1.0));
  double phi = std::atan2(v.y, v.x);
  if (phi < 0.0) phi += 2.0 * PI;
  *s = theta / PI; *t = phi / (PI * 2.0);
}
inline void map_cylinder(const double* m, V3 p, double* s, double* t) {
  V3 v = vnormalize(xf_pt(m, p) - V3());
  *s = (PI + std::atan2(v.y, v.x)) / (2.0 * PI); *t = v.z;
}
void tex_map_2d(const rrt_texture& t, const SI& si, double st[2], double dstdx[2], double dstdy[2]) {
  switch (t.mapping) {
    case RRT_MAP_UV:
      dstdx[0] = t.map[0] * si.dudx; dstdx[1] = t.map[1] * si.dvdx;
      dstdy[0] = t.map[0] * si.dudy; dstdy[1] = t.map[1] * si.dvdy;
      st[0] = t.map[0] * si.u + t.map[2]; st[1] = t.map[1] * si.v + t.map[3];
      return;
    case RRT_MAP_SPHERICAL: case RRT_MAP_CYLINDRICAL: {
      auto f = t.mapping == RRT_MAP_SPHERICAL ? map_sphere : map_cylinder;
      const double delta = 0.1;
      double sx[2], sy[2];
      f(t.world_to_texture, si.p, &st[0], &st[1]);
      f(t.world_to_texture, si.p + si.dpdx * delta, &sx[0], &sx[1]);
      dstdx[0] = (sx[0] - st[0]) / delta; dstdx[1] = (sx[1] - st[1]) / delta;
      f(t.world_to_texture, si.p + si.dpdy * delta, &sy[0], &sy[1]);
      dstdy[0] = (sy[0] - st[0]) / delta; dstdy[1] = (sy[1] - st[1]) / delta;
      if (dstdx[1] > 0.5) dstdx[1] = 1.0 - dstdx[1]; else if (dstdx[1] < -0.5) dstdx[1] = -(dstdx[1] + 1.0);
      if (dstdy[1] > 0.5) dstdy[1] = 1.0 - dstdy[1]; else if (dstdy[1] < -0.5) dstdy[1] = -(dstdy[1] + 1.0);
      return;
    }
    default: {   // RRT_MAP_PLANAR
      V3 vs(t.vs[0], t.vs[1], t.vs[2]), vt(t.vt[0], t.vt[1], t.vt[2]);
      dstdx[0] = dot(si.dpdx, vs); dstdx[1] = dot(si.dpdx, vt);
      dstdy[0] = dot(si.dpdy, vs); dstdy[1] = dot(si.dpdy, vt);
      st[0] = t.map[0] + dot(si.p, vs); st[1] = t.map[1] + dot(si.p, vt);
      return;
    }
  }
}
inline double bump_int(double x) { return std::floor(x / 2.0) + 2.0 * rmax(x / 2.0 - std::floor(x / 2.0) - 0.5, 0.0); }   // checkerboard.rs:46-48

// ---- MIPMap lookups mipmap.rs:98-268 over the level storage the loader built (rrt_image, include/rrt.h) ----------------
struct OracleMip {
  const rrt_scene_desc* d;
  const rrt_image& im;
  size_t levels() const { return (size_t)im.n_levels; }
  static size_t f2usize(double v) { return !(v > 0.0) ? 0 : (v >= 18446744073709551615.0 ? (size_t)-1 : (size_t)v); }   // Rust `as usize`
  Rgb at(size_t level, size_t u, size_t v) const {   // BlockedArray::index memory.rs:76-85
    const rrt_image_level& L = im.levels[level];
    const size_t i = 16 * ((size_t)L.u_blocks * (v & 3) + (u & 3)) + 4 * (v >> 2) + (u >> 2);
    if (i >= L.n) throw OraclePanic{"memory.rs:84 BlockedArray index out of bounds"};
    return Rgb(d->image_texels + 3 * (L.offset + i));
  }
  Rgb texel(size_t level, size_t s, size_t t) const {   // :107-131
    if (level >= levels()) throw OraclePanic{"mipmap.rs:108 assert!(level < self.pyramid.len())"};
    const rrt_image_level& L = im.levels[level];
    size_t ts = 0, tt = 0;
    if (im.wrap == RRT_WRAP_REPEAT) { ts = s - (s / L.u_res) * L.u_res; tt = t - (t / L.v_res) * L.v_res; }
    else if (im.wrap == RRT_WRAP_BLACK) { if (s >= L.u_res || t >= L.v_res) return Rgb(); }   // in range: texel (0, 0), as written there
    else { ts = s > L.u_res ? L.u_res : s; tt = t > L.v_res ? L.v_res : t; }                 // clamp_t(s, 0, u_size): u_size itself passes
    return at(level, ts, tt);
  }
  Rgb triangle(size_t level, const double st[2]) const {   // :193-205
    level = level > levels() - 1 ? levels() - 1 : level;
    const rrt_image_level& L = im.levels[level];
    const double s = st[0] * (double)L.u_res - 0.5, t = st[1] * (double)L.v_res - 0.5;
    const size_t s0 = f2usize(std::floor(s)), t0 = f2usize(std::floor(t));
    const double ds = s - std::trunc(s), dt = t - std::trunc(t);   // f64::fract
    return texel(level, s0, t0) * (1.0 - ds) * (1.0 - dt) + texel(level, s0, t0 + 1) * (1.0 - ds) * dt + texel(level, s0 + 1, t0) * ds * (1.0 - dt) +
           texel(level, s0 + 1, t0 + 1) * ds * dt;
  }
  Rgb lookup_w(const double st[2], double width) const {   // :132-149
    const double level = (double)levels() - 1.0 + std::log2(rmax(width, 1e-8));
    if (level < 0.0) return triangle(0, st);
    if (level >= (double)(levels() - 1)) return texel(levels() - 1, 0, 0);
    const size_t il = f2usize(std::floor(level));
    const double delta = level - std::trunc(level);
    return triangle(il, st) * (1.0 - delta) + triangle(il + 1, st) * delta;
  }
  Rgb ewa(size_t level, const double st_in[2], const double dst0_in[2], const double dst1_in[2]) const {   // :206-268
    if (level > levels()) return texel(levels() - 1, 0, 0);
    if (level >= levels()) throw OraclePanic{"mipmap.rs:217 pyramid[level]: index out of bounds (ewa of the level past the last)"};
    const rrt_image_level& L = im.levels[level];
    const double st[2] = {st_in[0] * (double)L.u_res - 0.5, st_in[1] * (double)L.v_res - 0.5};
    const double dst0[2] = {dst0_in[0] * (double)L.u_res, dst0_in[1] * (double)L.v_res};
    const double dst1[2] = {dst1_in[0] * (double)L.u_res, dst1_in[1] * (double)L.v_res};
    double a = dst0[1] * dst0[1] + dst1[1] * dst1[1] + 1.0;
    double b = -2.0 * (dst0[0] * dst0[1] + dst1[0] * dst1[1]);
    double c = dst0[0] * dst0[0] + dst1[0] * dst1[0] + 1.0;
    const double inv_f = 1.0 / (a * c - b * b * 0.25);
    a *= inv_f; b *= inv_f; c *= inv_f;
    const double det = -b * b + 4.0 * a * c, inv_det = 1.0 / det;
    const double u_sqrt = std::sqrt(det * c), v_sqrt = std::sqrt(det * a);
    const size_t s0 = f2usize(std::ceil(st[0] - 2.0 * inv_det * u_sqrt)), s1 = f2usize(std::floor(st[0] + 2.0 * inv_det * u_sqrt));
    const size_t t0 = f2usize(std::ceil(st[1] - 2.0 * inv_det * v_sqrt)), t1 = f2usize(std::floor(st[1] + 2.0 * inv_det * v_sqrt));
    Rgb sum;
    double sum_wts = 0.0;
    for (size_t it = t0; it <= t1 && it >= t0; it++) {
      const double tt = (double)it - st[0];   // (st[0], as written at :250)
      for (size_t is = s0; is <= s1 && is >= s0; is++) {
        const double ss = (double)is - st[0];
        const double r2 = a * ss * ss + b * ss * tt + c * tt * tt;
        if (r2 < 1.0) {
          const size_t index = f2usize(std::fmin(r2 * 128.0, 127.0));
          const double r2i = (double)index / 127.0;
          const double weight = std::exp(-2.0 * r2i) - std::exp(-2.0);   // WEIGHT_LUT :13-23
          sum = sum + texel(level, is, it) * weight;
          sum_wts += weight;
        }
      }
    }
    return sum / sum_wts;
  }
  Rgb lookup_d(const double st[2], const double dstdx[2], const double dstdy[2]) const {   // :150-192
    if (im.do_trilinear) {
      const double width = rmax(rmax(std::fabs(dstdx[0]), std::fabs(dstdx[1])), rmax(std::fabs(dstdy[0]), std::fabs(dstdy[1])));
      return lookup_w(st, width);
    }
    double dst0[2], dst1[2];
    if (dstdx[0] * dstdx[0] + dstdx[1] * dstdx[1] < dstdy[0] * dstdy[0] + dstdy[1] * dstdy[1]) { dst0[0] = dstdy[0]; dst0[1] = dstdy[1]; dst1[0] = dstdx[0]; dst1[1] = dstdx[1]; }
    else { dst0[0] = dstdx[0]; dst0[1] = dstdx[1]; dst1[0] = dstdy[0]; dst1[1] = dstdy[1]; }
    const double major_length = std::sqrt(dst0[0] * dst0[0] + dst0[1] * dst0[1]);
    double minor_length = std::sqrt(dst1[0] * dst1[0] + dst1[1] * dst1[1]);
    if (minor_length * im.max_aniso < major_length && minor_length > 0.0) {
      const double scale = major_length / (minor_length * im.max_aniso);
      dst1[0] *= scale; dst1[1] *= scale;
      minor_length *= scale;
    }
    if (minor_length == 0.0) return triangle(0, st);
    const double lod = rmax((double)(levels() - 1) + std::log2(minor_length), 0.0);
    const size_t i_lod = f2usize(std::floor(lod));
    const double fr = lod - std::trunc(lod);
    return ewa(i_lod, st, dst0, dst1) * (1.0 - fr) + ewa(i_lod + 1, st, dst0, dst1) * fr;   // lerp evaluates both
  }
};

Rgb tex_eval(const rrt_scene_desc* d, int id, const SI& si);
inline Rgb tex_child(const rrt_scene_desc* d, const rrt_texture& t, int slot, const SI& si) {
  return t.child[slot] >= 0 ? tex_eval(d, t.child[slot], si) : Rgb(t.fallback[slot]);
}
// Texture::evaluate of every in-scope texture; a float texture carries its value in all three channels
Rgb tex_eval(const rrt_scene_desc* d, int id, const SI& si) {
  const rrt_texture& t = d->textures[id];
  switch (t.type) {
    case RRT_TEX_CONSTANT: return Rgb(t.v[0]);
    case RRT_TEX_MIX: {   // mix.rs:32-38
      double amt = tex_child(d, t, 2, si).c[0];
      return tex_child(d, t, 0, si) * (1.0 - amt) + tex_child(d, t, 1, si) * amt;
    }
    case RRT_TEX_SCALE: return tex_child(d, t, 0, si) * tex_child(d, t, 1, si);   // scale.rs:30-32
    case RRT_TEX_BILERP: {   // bilerp.rs:33-43
      double st[2], dx[2], dy[2];
      tex_map_2d(t, si, st, dx, dy);
      return Rgb(t.v[0]) * (1.0 - st[0]) * (1.0 - st[1]) + Rgb(t.v[1]) * (1.0 - st[0]) * st[1] + Rgb(t.v[2]) * st[0] * (1.0 - st[1]) + Rgb(t.v[3]) * st[0] * st[1];
    }
    case RRT_TEX_UV: {   // uv.rs:20-28
      double st[2], dx[2], dy[2];
      tex_map_2d(t, si, st, dx, dy);
      return Rgb(st[0] - std::floor(st[0]), st[1] - std::floor(st[1]), 0.0);
    }
    case RRT_TEX_CHECKER2D: {   // checkerboard.rs:54-97
      double st[2], dx[2], dy[2];
      tex_map_2d(t, si, st, dx, dy);
      bool first = (f2i_sat(std::floor(st[0])) + f2i_sat(std::floor(st[1]))) % 2 == 0;
      if (t.aa_none) return tex_child(d, t, first ? 0 : 1, si);
      double ds = rmax(std::fabs(dx[0]), std::fabs(dx[1])), dt = rmax(std::fabs(dy[0]), std::fabs(dy[1]));
      double s0 = st[0] - ds, s1 = st[0] + ds, t0 = st[1] - dt, t1 = st[1] + dt;
      if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1)) return tex_child(d, t, first ? 0 : 1, si);
      double sint = (bump_int(s1) - bump_int(s0)) / (2.0 * ds), tint = (bump_int(t1) - bump_int(t0)) / (2.0 * dt);
      double area2 = sint + tint - 2.0 * sint * tint;
      if (ds > 1.0 || dt > 1.0) area2 = 0.5;
      return tex_child(d, t, 0, si) * (1.0 - area2) + tex_child(d, t, 1, si) * area2;
    }
    case RRT_TEX_CHECKER3D: {   // checkerboard.rs:121-131
      V3 p = xf_pt(t.world_to_texture, si.p);
      return tex_child(d, t, f2i_sat(std::floor(p.x) + std::floor(p.y) + std::floor(p.z)) % 2 == 0 ? 0 : 1, si);
    }
    case RRT_TEX_WINDY: {   // windy.rs:15-23
      V3 p = xf_pt(t.world_to_texture, si.p), dpdx = xf_vec(t.world_to_texture, si.dpdx), dpdy = xf_vec(t.world_to_texture, si.dpdy);
      double wind_strength = tex_fbm(p * 0.1, dpdx * 0.1, dpdy * 0.1, 0.5, 3);
      double wave_height = tex_fbm(p, dpdx, dpdy, 0.5, 6);
      return Rgb(std::fabs(wind_strength) * wave_height);
    }
    case RRT_TEX_WRINKLED: {   // wrinkled.rs:21-28
      V3 p = xf_pt(t.world_to_texture, si.p), dpdx = xf_vec(t.world_to_texture, si.dpdx), dpdy = xf_vec(t.world_to_texture, si.dpdy);
      return Rgb(tex_turbulence(p, dpdx, dpdy, t.omega, t.octaves));
    }
    case RRT_TEX_IMAGE: {   // imagemap.rs:74-81
      double st[2], dx[2], dy[2];
      tex_map_2d(t, si, st, dx, dy);
      if (t.image < 0 || (size_t)t.image >= d->n_images) throw OraclePanic{"image texture without a decoded image"};
      return OracleMip{d, d->images[t.image]}.lookup_d(st, dx, dy);
    }
    default: throw OraclePanic{"texture type outside the oracle's scope"};
  }
}
// a material with every textured parameter replaced by its value at this hit
rrt_material resolve_material(const rrt_scene_desc* d, const rrt_material& m0, const SI& si) {
  rrt_material m = m0;
  double* dst3[RRT_P_COUNT] = {m.kd, m.ks, m.kr, m.eta, m.k, nullptr, nullptr, nullptr, nullptr, m.kt, m.reflect, m.transmit, nullptr};
  double* dst1[RRT_P_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, &m.sigma, &m.roughness, &m.u_roughness, &m.v_roughness, nullptr, nullptr, nullptr, &m.index};
  for (int k = 0; k < RRT_P_COUNT; k++) {
    if (m.tex[k] < 0) continue;
    Rgb v = tex_eval(d, m.tex[k], si);
    if (dst3[k]) { dst3[k][0] = v.c[0]; dst3[k][1] = v.c[1]; dst3[k][2] = v.c[2]; } else *dst1[k] = v.c[0];
  }
  return m;
}

// SurfaceInteraction::compute_scattering_functions interaction.rs:203-214 (compute_differentials first) +
// Material::compute_scattering_functions for the in-scope materials
// Material::bump material/mod.rs:22-62 (every in-scope material calls it first when it has a bump_map)
void bump_shading(const rrt_scene_desc* d, int tex, SI* si) {
  SI ev = *si;
  double du = std::fabs(si->dudx) * 0.5 + std::fabs(si->dudy);   // (as written at :26; pbrt halves the sum)
  if (du == 0.0) du = 0.0005;
  ev.p = si->p + si->sdpdu * du;
  ev.u = si->u + du; ev.v = si->v;
  const double u_displace = tex_eval(d, tex, ev).c[0];
  double dv = (std::fabs(si->dvdx) + std::fabs(si->dvdy)) * 0.5;
  if (dv == 0.0) dv = 0.0005;
  ev.p = si->p + si->sdpdv * dv;
  ev.u = si->u; ev.v = si->v + dv;
  const double v_displace = tex_eval(d, tex, ev).c[0];
  const double displace = tex_eval(d, tex, *si).c[0];
  V3 dpdu = si->sdpdu + si->sn * (u_displace - displace) / du + si->sdndu * displace;
  V3 dpdv = si->sdpdv + si->sn * (v_displace - displace) / dv + si->sdndv * displace;
  si_set_shading(si, dpdu, dpdv, si->sdndu, si->sdndv, false);
}

void compute_scattering(const Scene& sc, SI& si, const RayDiff& rd, Bsdf* bsdf, bool allow_multiple_lobes = true) {
  compute_differentials(&si, rd);
  if (!(dot(si.n, si.sn) >= 0.0)) throw OraclePanic{"primitives.rs:100 assert!(dot3(&si.ist.n, &si.shading.n) >= 0.0)"};
  {
    const rrt_material& mb = sc.d->materials[sc.d->prims[si.prim].material];
    if (mb.bump >= 0 && mb.type != RRT_MAT_DEBUG) bump_shading(sc.d, mb.bump, &si);   // (the Debug material has no bump_map)
  }
  const rrt_material m = resolve_material(sc.d, sc.d->materials[sc.d->prims[si.prim].material], si);
  bsdf->init(si);
  switch (m.type) {
    case RRT_MAT_MATTE: {  // matte.rs:35-60
      Rgb r = rclamp0(Rgb(m.kd));
      double sig = clampd(m.sigma, 0.0, 90.0);
      if (!r.is_black()) {
        Lobe l; l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = r;
        if (sig == 0.0) l.kind = LAMBERT;
        else {  // OrenNayar::new reflection.rs:907-913
          l.kind = OREN_NAYAR;
          double s = (PI / 180.0) * sig;
          double sigma2 = s * s;
          l.a = 1. - (sigma2 / (2. * (sigma2 + 0.33)));
          l.b = 0.45 * sigma2 / (sigma2 + 0.09);
        }
        bsdf->add(l);
      }
      break;
    }
    case RRT_MAT_PLASTIC: {  // plastic.rs:42-73 (specular lobe gated on kd: Q31)
      Rgb kd = rclamp0(Rgb(m.kd)), ks = rclamp0(Rgb(m.ks));
      if (!kd.is_black()) { Lobe l; l.kind = LAMBERT; l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = kd; bsdf->add(l); }
      if (!kd.is_black()) {
        double rough = m.roughness;
        if (m.remap_roughness) rough = roughness_to_alpha(rough);
        Lobe l; l.kind = MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = ks; l.alpha_x = rough; l.alpha_y = rough;
        l.fr = FR_DIELECTRIC; l.eta_i = Rgb(1.5); l.eta_t = Rgb(1.0);
        bsdf->add(l);
      }
      break;
    }
    case RRT_MAT_METAL: {  // metal.rs:48-89
      double ur = m.u_roughness, vr = m.v_roughness;
      if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
      Lobe l; l.kind = MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = Rgb(1.0); l.alpha_x = ur; l.alpha_y = vr;
      l.fr = FR_CONDUCTOR; l.eta_i = Rgb(1.0); l.eta_t = Rgb(m.eta); l.k = Rgb(m.k);
      bsdf->add(l);
      break;
    }
    case RRT_MAT_MIRROR: {  // mirror.rs:27-47
      Rgb r = rclamp0(Rgb(m.kr));
      if (!r.is_black()) { Lobe l; l.kind = SPEC_REFL; l.type = BXDF_REFLECTION | BXDF_SPECULAR; l.r = r; l.fr = FR_NOOP; bsdf->add(l); }
      break;
    }
    case RRT_MAT_DEBUG: {  // debug_material.rs:37-48
      Lobe a; a.kind = DEBUG_DIFFUSE; a.type = BXDF_DIFFUSE | BXDF_REFLECTION; bsdf->add(a);
      Lobe b; b.kind = DEBUG_SPECULAR; b.type = BXDF_SPECULAR | BXDF_REFLECTION; bsdf->add(b);
      break;
    }
    case RRT_MAT_GLASS: {  // glass.rs:52-112 (allow_multiple_lobes as passed by the integrator, mode = Radiance)
      double eta = m.index, ur = m.u_roughness, vr = m.v_roughness;
      Rgb r = rclamp0(Rgb(m.kr)), t = rclamp0(Rgb(m.kt));
      bsdf->eta = eta;
      if (r.is_black() && t.is_black()) throw OraclePanic{"glass.rs:70 null BSDF: path.rs:103 `bounces -= 1` underflows"};
      bool is_specular = ur == 0.0 && vr == 0.0;
      if (is_specular && allow_multiple_lobes) {
        Lobe l; l.kind = FRESNEL_SPEC; l.type = BXDF_SPECULAR | BXDF_ALL; l.r = r; l.t = t; l.eta_a = 1.0; l.eta_b = eta; bsdf->add(l);
      } else {
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        if (!r.is_black()) {
          Lobe l; l.r = r; l.fr = FR_DIELECTRIC; l.eta_i = Rgb(1.0); l.eta_t = Rgb(eta);
          if (is_specular) { l.kind = SPEC_REFL; l.type = BXDF_REFLECTION | BXDF_SPECULAR; }
          else { l.kind = MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.alpha_x = ur; l.alpha_y = vr; }
          bsdf->add(l);
        }
        if (!t.is_black()) {
          Lobe l; l.t = t; l.eta_a = 1.0; l.eta_b = eta;
          if (is_specular) { l.kind = SPEC_TRANS; l.type = BXDF_SPECULAR | BXDF_TRANSMISSION; }
          else { l.kind = MICROFACET_TRANS; l.type = BXDF_GLOSSY | BXDF_TRANSMISSION; l.alpha_x = ur; l.alpha_y = vr; }
          bsdf->add(l);
        }
      }
      break;
    }
    case RRT_MAT_TRANSLUCENT: {  // translucent.rs:50-107
      const double eta = 1.5;
      bsdf->eta = eta;
      Rgb r = rclamp0(Rgb(m.reflect)), t = rclamp0(Rgb(m.transmit));
      if (r.is_black() && t.is_black()) throw OraclePanic{"translucent.rs:66 null BSDF: path.rs:103 `bounces -= 1` underflows"};
      Rgb kd = rclamp0(Rgb(m.kd));
      if (!kd.is_black()) {
        if (!r.is_black()) { Lobe l; l.kind = LAMBERT; l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = r * kd; bsdf->add(l); }
        if (!t.is_black()) { Lobe l; l.kind = LAMBERT_TRANS; l.type = BXDF_DIFFUSE | BXDF_TRANSMISSION; l.t = t * kd; bsdf->add(l); }
      }
      Rgb ks = rclamp0(Rgb(m.ks));
      if (!ks.is_black() && (!r.is_black() || !t.is_black())) {
        double rough = m.roughness;
        if (m.remap_roughness) rough = roughness_to_alpha(rough);
        if (!r.is_black()) {
          Lobe l; l.kind = MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = r * ks; l.alpha_x = rough; l.alpha_y = rough;
          l.fr = FR_DIELECTRIC; l.eta_i = Rgb(1.0); l.eta_t = Rgb(eta);
          bsdf->add(l);
        }
        if (!t.is_black()) {
          Lobe l; l.kind = MICROFACET_TRANS; l.type = BXDF_GLOSSY | BXDF_TRANSMISSION; l.t = t * ks; l.alpha_x = rough; l.alpha_y = rough; l.eta_a = 1.0; l.eta_b = eta;
          bsdf->add(l);
        }
      }
      break;
    }
    default: throw OraclePanic{"unknown material type in scene desc"};
  }
}

// ---- lights (lights/point.rs, lights/diffuse.rs, shape/mod.rs sample_ref/pdf_ref) ----------------------
struct LightSample { Rgb li; V3 wi; double pdf = 0; V3 p1, n1; bool have_vis = false; };

// Shape::sample for the two light shapes
void shape_sample(const Scene& sc, const rrt_light& L, double u0, double u1, V3* p, V3* n, double* pdf) {
  if (L.shape_type == RRT_PRIM_SPHERE) {  // Sphere::sample sphere.rs:265-285
    SphereRef S = sphere_ref(sc, L.shape);
    V3 p_obj = V3() + uniform_sample_sphere(u0, u1) * S.s->radius;
    *n = nnormalize(xf_nrm(S.mi, V3(p_obj.x, p_obj.y, p_obj.z)));
    p_obj = p_obj * (S.s->radius / len(p_obj - V3()));
    *p = xf_pt(S.m, p_obj);
    *pdf = 1.0 / L.area;
  } else {                                // Triangle::sample triangle.rs:393-418 (Q19)
    const rrt_tri& t = sc.d->tris[L.shape];
    V3 b = uniform_sample_sphere(u0, u1);
    V3 p0 = sc.P(t.v[0]), p1 = sc.P(t.v[1]), p2 = sc.P(t.v[2]);
    *p = p0 * b.x + p1 * b.y + p2 * b.z;
    *n = vnormalize(cross(p1 - p0, p2 - p0));
    if (t.mesh_has_n) {  // `!self.mesh.n.is_empty()` (2 = normals present but no indices: n = [0,0,0])
      V3 ns = sc.N(t.n[0]) * b.x + sc.N(t.n[1]) * b.y + sc.N(t.n[2]) * b.z;
      *n = faceforward(*n, ns);
    }
    *pdf = 1.0 / tri_area(sc, t);
  }
}
// Shape::pdf_ref shape/mod.rs:49-66
double shape_pdf_ref(const Scene& sc, const rrt_light& L, V3 ref_p, V3 wi) {
  Ray r = ray_new(ref_p, wi, INF);  // ref.spawn_ray(wi)
  double thit = 0;
  SI isl;
  double bu, bv;
  bool hit = (L.shape_type == RRT_PRIM_SPHERE) ? sphere_intersect(sphere_ref(sc, L.shape), r, &thit, &isl)
                                               : tri_intersect(sc, sc.d->tris[L.shape], r, &thit, &isl, &bu, &bv);
  if (!hit) return 0.0;
  double area = (L.shape_type == RRT_PRIM_SPHERE) ? L.area : tri_area(sc, sc.d->tris[L.shape]);
  double pdf = len2(ref_p - isl.p) / (absdot(-wi, isl.n) * area);
  if (std::isinf(pdf)) pdf = 0.0;
  return pdf;
}

LightSample light_sample_li(const Scene& sc, const rrt_light& L, V3 ref_p, double u0, double u1) {
  LightSample s;
  if (L.type == RRT_LIGHT_POINT) {  // point.rs:55-77
    V3 pl(L.p_light[0], L.p_light[1], L.p_light[2]);
    s.wi = vnormalize(pl - ref_p);
    s.pdf = 1.0;
    s.p1 = pl; s.n1 = V3(); s.have_vis = true;
    s.li = Rgb(L.spectrum) / len2(pl - ref_p);
    return s;
  }
  if (L.type == RRT_LIGHT_DISTANT) {  // distant.rs:67-92: wi = w_light, pdf = 1, p1 = p + w_light * 2 * world_radius
    V3 w(L.w_light[0], L.w_light[1], L.w_light[2]);
    s.wi = w;
    s.pdf = 1.0;
    s.p1 = ref_p + w * (2.0 * L.world_radius); s.n1 = V3(); s.have_vis = true;
    s.li = Rgb(L.spectrum);
    return s;
  }
  // DiffuseAreaLight::sample_li diffuse.rs:63-79 over Shape::sample_ref shape/mod.rs:33-48
  V3 p, n;
  double pdf;
  shape_sample(sc, L, u0, u1, &p, &n, &pdf);
  V3 wi = p - ref_p;
  double wl2 = len2(wi);
  if (wl2 == 0.0) pdf = 0.0;
  else {
    wi = vnormalize(wi);
    pdf = wl2 / absdot(-wi, n);
    if (std::isinf(pdf)) pdf = 0.0;
  }
  s.pdf = pdf;
  if (pdf == 0.0 || len2(p - ref_p) == 0.0) { s.pdf = 0.0; s.li = Rgb(); return s; }
  s.wi = vnormalize(p - ref_p);
  s.p1 = p; s.n1 = n; s.have_vis = true;
  s.li = (dot(n, -s.wi) > 0.0) ? Rgb(L.spectrum) : Rgb();  // AreaLight::l diffuse.rs:133-141
  return s;
}
inline bool is_delta_light(const rrt_light& L) { return L.type == RRT_LIGHT_POINT || L.type == RRT_LIGHT_DISTANT; }  // LIGHT_DELTAPOSITION / _DELTADIRECTION

// pnt3_offset_ray_origin geometry.rs:721-749 with p_error == 0 everywhere (Q8): offset = n*0, sign flips
// only produce -0.0 components and `offset[i] > 0 / < 0` never fires -> po = p + (+-0).
inline V3 offset_ray_origin(V3 p, V3 n, V3 w) {
  double d = (std::fabs(n.x) * 0.0) + (std::fabs(n.y) * 0.0) + (std::fabs(n.z) * 0.0);
  V3 offset = n * d;
  if (dot(w, n) < 0.0) offset = -offset;
  return p + offset;
}
// VisibilityTester::unoccluded lights/mod.rs:60-66 over spawn_ray_to_si interaction.rs:66-77 (Q9)
bool unoccluded(const Scene& sc, V3 p0, V3 n0, V3 p1, V3 n1, Counters* cnt) {
  V3 origin = offset_ray_origin(p0, n0, p1 - p0);
  V3 target = offset_ray_origin(p1, n1, origin - p1);
  V3 d = target - origin;
  Ray r = ray_new(origin, d, 1.0 - SHADOW_EPSILON);
  return !scene_intersect_p(sc, r, cnt);
}

// estimate_direct integrator/mod.rs:403-558 (handle_media = false, specular = false)
Rgb estimate_direct(const Scene& sc, const SI& si, const Bsdf& bsdf, double us0, double us1, const rrt_light& L,
                    double ul0, double ul1, Counters* cnt) {
  const uint8_t flags = BXDF_ALL & ~BXDF_SPECULAR;
  Rgb ld;
  double light_pdf = 0.0, scattering_pdf = 0.0;
  LightSample ls = light_sample_li(sc, L, si.p, ul0, ul1);
  light_pdf = ls.pdf;
  Rgb li = ls.li;
  V3 wi = ls.wi;
  if (light_pdf > 0.0 && !li.is_black()) {
    Rgb f;
    if (bsdf.present) {
      f = bsdf.f(si.wo, wi, flags) * absdot(wi, si.sn);
      scattering_pdf = bsdf.pdf(si.wo, wi, flags);
    }
    if (!f.is_black()) {
      if (!unoccluded(sc, si.p, si.n, ls.p1, ls.n1, cnt)) li = Rgb();
      if (!li.is_black()) {
        if (is_delta_light(L)) ld = ld + f * li / light_pdf;
        else { double w = power_heuristic(1, light_pdf, 1, scattering_pdf); ld = ld + li * f * w / light_pdf; }
      }
    }
  }
  if (!is_delta_light(L)) {
    Rgb f;
    bool sampled_specular = false;
    if (bsdf.present) {
      uint8_t st = BXDF_NONE;
      f = bsdf.sample_f(si.wo, &wi, us0, us1, &scattering_pdf, flags, &st);
      f = f * absdot(wi, si.sn);
      sampled_specular = (st & BXDF_SPECULAR) != 0;
    }
    if (!f.is_black() && scattering_pdf > 0.0) {
      if (!sampled_specular) {
        light_pdf = shape_pdf_ref(sc, L, si.p, wi);  // DiffuseAreaLight::pdf_li diffuse.rs:85-87
        if (light_pdf == 0.0) return ld;
      }
      // The BSDF-sampled ray: no primitive carries an area light (Q18) and DiffuseAreaLight::le is the
      // trait default 0, so `li` below is always black; the trace only costs time (and counters).
      if (!(sc.d->flags & RRT_SKIP_MIS_BSDF_RAY)) {
        Ray ray = ray_new(si.p, wi, INF);
        SI light_isect;
        scene_intersect(sc, &ray, &light_isect, nullptr, cnt);
      }
    }
  }
  return ld;
}

// Distribution1D::new + sample_discrete sampling.rs:17-46,93-123 for func = [1; n]
struct UniformLightDistrib {
  std::vector<double> cdf;
  double func_int = 0;
  size_t n = 0;
  void init(size_t nl) {
    n = nl;
    cdf.assign(n + 1, 0.0);
    for (size_t i = 1; i <= n; i++) cdf[i] = cdf[i - 1] + 1.0 / (double)n;
    func_int = cdf[n];
    if (func_int == 0.0) for (size_t i = 1; i <= n; i++) cdf[i] = (double)i / (double)n;
    else for (size_t i = 1; i <= n; i++) cdf[i] /= func_int;
  }
  size_t sample_discrete(double u, double* pdf) const {
    size_t first = 0, len = cdf.size();
    while (len > 0) {
      size_t half = len >> 1, middle = first + half;
      if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
    }
    // `clamp_t(first - 1, 0, len-2)` on usize: first >= 1 because cdf[0] = 0 <= u for u >= 0
    size_t off = first - 1;
    if (off > cdf.size() - 2) off = cdf.size() - 2;
    *pdf = func_int > 0.0 ? 1.0 / (func_int * (double)n) : 0.0;
    return off;
  }
};

// uniform_sample_one_light integrator/mod.rs:359-401
Rgb uniform_sample_one_light(const Scene& sc, const SI& si, const Bsdf& bsdf, Sampler& smp, const UniformLightDistrib* distrib, Counters* cnt) {
  size_t n_lights = sc.d->n_lights;
  if (n_lights == 0) return Rgb();
  size_t light_num;
  double light_pdf = 0.0;
  if (distrib) {
    light_num = distrib->sample_discrete(smp.get_1d(), &light_pdf);
    if (light_pdf == 0.0) return Rgb();
  } else {
    double v = smp.get_1d() * (double)n_lights;
    size_t vi = (v != v || v <= 0.0) ? 0 : (size_t)v;
    light_num = std::min(vi, n_lights - 1);
    light_pdf = 1.0 / (double)n_lights;
  }
  double ul0, ul1, us0, us1;
  smp.get_2d(&ul0, &ul1);
  smp.get_2d(&us0, &us1);
  return estimate_direct(sc, si, bsdf, us0, us1, sc.d->lights[light_num], ul0, ul1, cnt) / light_pdf;
}
// uniform_sample_all_lights :304-355 (tile samplers hold no sample arrays: single-sample branch, Q30)
Rgb uniform_sample_all_lights(const Scene& sc, const SI& si, const Bsdf& bsdf, Sampler& smp, Counters* cnt) {
  Rgb l;
  for (size_t j = 0; j < sc.d->n_lights; j++) {
    double ul0, ul1, us0, us1;
    smp.get_2d(&ul0, &ul1);
    smp.get_2d(&us0, &us1);
    l = l + estimate_direct(sc, si, bsdf, us0, us1, sc.d->lights[j], ul0, ul1, cnt);
  }
  return l;
}

struct Integ {
  const Scene& sc;
  const rrt_integrator& in;
  UniformLightDistrib distrib;
  Counters* cnt;

  // PathIntegrator::li path.rs:51-226
  Rgb li_path(Ray ray, RayDiff rdiff, Sampler& smp) {
    Rgb l, beta(1.0);
    bool specular_bounce = false;
    long bounces = 0;
    double eta_scale = 1.0;
    while (true) {
      SI isect;
      bool found = scene_intersect(sc, &ray, &isect, nullptr, cnt);
      // isect.le() is 0 (no area-light primitives, Q18); infinite_lights is empty on supported scenes
      (void)specular_bounce;
      if (!found || bounces >= (long)in.max_depth) break;
      Bsdf bsdf;
      compute_scattering(sc, isect, rdiff, &bsdf);
      if (bsdf.num_components(BXDF_ALL & ~BXDF_SPECULAR) > 0) {
        Rgb ld = beta * uniform_sample_one_light(sc, isect, bsdf, smp, &distrib, cnt);
        l = l + ld;
      }
      V3 wo = -ray.d, wi;
      double pdf = 0.0, u0, u1;
      uint8_t flags = 0;
      smp.get_2d(&u0, &u1);
      Rgb f = bsdf.sample_f(wo, &wi, u0, u1, &pdf, BXDF_ALL, &flags);
      if (f.is_black() || pdf == 0.0) break;
      beta = beta * (f * absdot(wi, isect.sn) / pdf);
      if (!(beta.y() > 0.0)) throw OraclePanic{"path.rs:146 assert!(beta.y() > 0.0)"};
      if (!std::isfinite(beta.y())) throw OraclePanic{"path.rs:147 assert!(beta.y().is_finite())"};
      specular_bounce = (flags & BXDF_SPECULAR) != 0;
      if ((flags & BXDF_SPECULAR) && (flags & BXDF_TRANSMISSION)) {   // path.rs:150-162
        double eta = bsdf.eta;
        eta_scale *= (dot(wo, isect.n) > 0.0) ? eta * eta : 1.0 / (eta * eta);
      }
      ray = ray_new(isect.p, wi, INF);  // spawn_ray: no origin offset (Q8)
      rdiff = RayDiff();                // `.into()`: has_differentials = false (path.rs:163)
      Rgb rr_beta = beta * eta_scale;
      if (rr_beta.max_component() < in.rr_threshold && bounces > 3) {
        double q = rmax(1.0 - rr_beta.max_component(), 0.05);
        if (smp.get_1d() < q) break;
        beta = beta / (1.0 - q);
        if (!std::isfinite(beta.y())) throw OraclePanic{"path.rs:220 assert!(beta.y().is_finite())"};
      }
      bounces += 1;
    }
    return l;
  }

  // specular_reflect integrator/mod.rs:150-198, then specular_transmit :199-301: each draws a 2D sample and recurses;
  // the reflect subtree consumes its sampler dimensions before the transmit draw (depth-first).
  template <typename F>
  Rgb specular_terms(const Ray& ray, const RayDiff& rdiff, const SI& isect, const Bsdf& bsdf, Sampler& smp, int depth, F&& li) {
    Rgb out;
    {
      V3 wo = isect.wo, wi;
      double pdf = 0.0, u0, u1;
      uint8_t st = 0;
      smp.get_2d(&u0, &u1);
      Rgb f = bsdf.sample_f(wo, &wi, u0, u1, &pdf, BXDF_SPECULAR | BXDF_REFLECTION, &st);
      V3 ns = isect.sn;
      if (pdf > 0.0 && !f.is_black() && absdot(wi, ns) != 0.0) {
        Ray rd = ray_new(isect.p, wi, INF);
        RayDiff cd;
        if (rdiff.has) {   // integrator/mod.rs:183-201
          cd.has = true;
          cd.rxo = isect.p + isect.dpdx;
          cd.ryo = isect.p + isect.dpdy;
          V3 dndx = isect.sdndu * isect.dudx + isect.sdndv * isect.dvdx;
          V3 dndy = isect.sdndu * isect.dudy + isect.sdndv * isect.dvdy;
          V3 dwodx = -rdiff.rxd - wo, dwody = -rdiff.ryd - wo;
          double ddndx = dot(dwodx, ns) + dot(wo, dndx);
          double ddndy = dot(dwody, ns) + dot(wo, dndy);
          cd.rxd = wi - dwodx + (dndx * dot(wo, ns) + ns * ddndx) * 0.2;
          cd.ryd = wi - dwody + (dndy * dot(wo, ns) + ns * ddndy) * 0.2;
        }
        out = out + f * li(rd, cd, depth + 1) * absdot(wi, ns) / pdf;
      }
    }
    {
      V3 wo = isect.wo, wi;
      double pdf = 0.0, u0, u1;
      uint8_t st = 0;
      smp.get_2d(&u0, &u1);
      Rgb f = bsdf.sample_f(wo, &wi, u0, u1, &pdf, BXDF_SPECULAR | BXDF_TRANSMISSION, &st);
      V3 ns = isect.sn;
      if (pdf > 0.0 && !f.is_black() && absdot(wi, ns) != 0.0) {   // specular_transmit integrator/mod.rs:199-301
        Ray rd = ray_new(isect.p, wi, INF);
        RayDiff cd;
        if (rdiff.has) {   // :238-292
          cd.has = true;
          cd.rxo = isect.p + isect.dpdx;
          cd.ryo = isect.p + isect.dpdy;
          V3 dndx = isect.sdndu * isect.dudx + isect.sdndv * isect.dvdx;
          V3 dndy = isect.sdndu * isect.dudy + isect.sdndv * isect.dvdy;
          double eta = 1.0 / bsdf.eta;
          if (dot(wo, ns) < 0.0) { eta = 1.0 / eta; ns = -ns; dndx = -dndx; dndy = -dndy; }
          V3 dwodx = -rdiff.rxd - wo, dwody = -rdiff.ryd - wo;
          double ddndx = dot(dwodx, ns) + dot(wo, dndx);
          double ddndy = dot(dwody, ns) + dot(wo, dndy);
          double mu = eta * dot(wo, ns) - absdot(wi, ns);
          double dmudx = ddndx * (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns));
          double dmudy = ddndy * (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns));
          cd.rxd = wi - dwodx * eta + (dndx * mu + ns * dmudx);
          cd.ryd = wi - dwody * eta + (dndy * mu + ns * dmudy);
        }
        out = out + f * li(rd, cd, depth + 1) * absdot(wi, ns) / pdf;   // `ns` is the flipped one here when it was flipped (:293)
      }
    }
    (void)ray;
    return out;
  }

  // DirectLightingIntegrator::li directlighting.rs:72-132
  Rgb li_direct(Ray ray, RayDiff rdiff, Sampler& smp, int depth) {
    SI isect;
    if (!scene_intersect(sc, &ray, &isect, nullptr, cnt)) {
      if (sc.d->n_lights > 0) return Rgb();  // `for light in lights { l += le; return l }` (le = 0)
      throw OraclePanic{"directlighting.rs:91 unbounded recursion: miss with an empty light list (Q20)"};
    }
    Bsdf bsdf;
    compute_scattering(sc, isect, rdiff, &bsdf, false);   // allow_multiple_lobes = false (directlighting.rs:91)
    Rgb l;  // isect.le() = 0 (Q18)
    if (sc.d->n_lights > 0) {
      if (in.light_strategy == RRT_STRATEGY_ALL) l = l + uniform_sample_all_lights(sc, isect, bsdf, smp, cnt);
      else l = l + uniform_sample_one_light(sc, isect, bsdf, smp, nullptr, cnt);
    }
    if ((depth + 1) < in.max_depth) l = l + specular_terms(ray, rdiff, isect, bsdf, smp, depth, [&](Ray r, RayDiff cd, int d) { return li_direct(r, cd, smp, d); });
    return l;
  }
  // IntersectDebugIntegrator::li intersect_debug.rs:56-89
  Rgb li_debug(Ray ray, RayDiff rdiff, Sampler& smp, int depth) {
    SI isect;
    if (!scene_intersect(sc, &ray, &isect, nullptr, cnt)) return Rgb();
    Rgb l(0.1, 0.1, 0.1);
    Bsdf bsdf;
    compute_scattering(sc, isect, rdiff, &bsdf, false);   // intersect_debug.rs:71
    Rgb s_l;
    if (sc.d->n_lights > 0) s_l = s_l + uniform_sample_all_lights(sc, isect, bsdf, smp, cnt);
    if ((depth + 1) < in.max_depth) s_l = s_l + specular_terms(ray, rdiff, isect, bsdf, smp, depth, [&](Ray r, RayDiff cd, int d) { return li_debug(r, cd, smp, d); });
    return l + s_l;
  }
  // AOIntegrator::li ao.rs:52-99: compute_scattering_functions is never called, so isect.bsdf is None and
  // every hit returns zero before any sample is drawn.
  Rgb li_ao(Ray ray) {
    SI isect;
    scene_intersect(sc, &ray, &isect, nullptr, cnt);
    return Rgb();
  }
  Rgb li(const Ray& ray, const RayDiff& rdiff, Sampler& smp) {
    switch (in.type) {
      case RRT_INT_PATH: return li_path(ray, rdiff, smp);
      case RRT_INT_DIRECT: return li_direct(ray, rdiff, smp, 1);
      case RRT_INT_DEBUG: return li_debug(ray, rdiff, smp, 1);
      default: return li_ao(ray);
    }
  }
};

// ---- film (film.rs) ------------------------------------------------------------------------------------
struct FilmTile {
  int x0, y0, x1, y1;  // pixel_bounds
  std::vector<double> contrib;  // rgb per pixel
  std::vector<double> wsum;
};
inline int64_t f2i(double v) { return (int64_t)v; }  // Point2i::from(Point2f): `as i64` truncation (finite inputs)

// get_film_tile film.rs:216-234
FilmTile get_film_tile(const rrt_film& f, int sx0, int sy0, int sx1, int sy1) {
  FilmTile t;
  int64_t p0x = f2i(std::ceil((double)sx0 - 0.5 - f.filter_radius[0])), p0y = f2i(std::ceil((double)sy0 - 0.5 - f.filter_radius[1]));
  int64_t p1x = f2i(std::floor((double)sx1 - 0.5 + f.filter_radius[0])) + 1, p1y = f2i(std::floor((double)sy1 - 0.5 + f.filter_radius[1])) + 1;
  // Bounds2i::new orders the corners; intersect with cropped_pixel_bounds
  int64_t ax = std::min(p0x, p1x), bx = std::max(p0x, p1x), ay = std::min(p0y, p1y), by = std::max(p0y, p1y);
  t.x0 = (int)std::max<int64_t>(ax, f.crop[0]); t.y0 = (int)std::max<int64_t>(ay, f.crop[1]);
  t.x1 = (int)std::min<int64_t>(bx, f.crop[2]); t.y1 = (int)std::min<int64_t>(by, f.crop[3]);
  size_t area = (size_t)std::max(0, (t.x1 - t.x0)) * (size_t)std::max(0, (t.y1 - t.y0));
  t.contrib.assign(area * 3, 0.0);
  t.wsum.assign(area, 0.0);
  return t;
}
// FilmTile::add_sample film.rs:77-130
void add_sample(const rrt_film& f, FilmTile& t, double pfx, double pfy, Rgb l, double sample_weight) {
  if (l.y() > f.max_sample_luminance) l = l * (f.max_sample_luminance / l.y());
  double dx = pfx - 0.5, dy = pfy - 0.5;
  int64_t p0x = f2i(std::ceil(dx - f.filter_radius[0])), p0y = f2i(std::ceil(dy - f.filter_radius[1]));
  int64_t p1x = f2i(dx + f.filter_radius[0]) + 1, p1y = f2i(dy + f.filter_radius[1]) + 1;
  p0x = std::max<int64_t>(p0x, t.x0); p0y = std::max<int64_t>(p0y, t.y0);
  p1x = std::min<int64_t>(p1x, t.x1); p1y = std::min<int64_t>(p1y, t.y1);
  const int ts = 16;
  double inv_rx = 1.0 / f.filter_radius[0], inv_ry = 1.0 / f.filter_radius[1];
  for (int64_t y = p0y; y < p1y; y++) {
    double fy = std::fabs(((double)y - dy) * inv_ry * (double)ts);
    int64_t ify = std::min<int64_t>((int64_t)std::floor(fy), ts - 1);
    for (int64_t x = p0x; x < p1x; x++) {
      double fx = std::fabs(((double)x - dx) * inv_rx * (double)ts);
      int64_t ifx = std::min<int64_t>((int64_t)std::floor(fx), ts - 1);
      double fw = f.filter_table[ify * ts + ifx];
      size_t off = (size_t)(x - t.x0) + (size_t)(y - t.y0) * (size_t)(t.x1 - t.x0);
      Rgb c = (l * sample_weight) * fw;
      t.contrib[3 * off] += c.c[0]; t.contrib[3 * off + 1] += c.c[1]; t.contrib[3 * off + 2] += c.c[2];
      t.wsum[off] += fw;
    }
  }
}
// merge_film_tile film.rs:248-263 (filter_weight_sum added inside the i-loop: Q3). film = xyz[3], wsum.
void merge_film_tile(const rrt_film& f, const FilmTile& t, double* film) {
  int W = f.crop[2] - f.crop[0];
  for (int y = t.y0; y < t.y1; y++)
    for (int x = t.x0; x < t.x1; x++) {
      size_t off = (size_t)(x - t.x0) + (size_t)(y - t.y0) * (size_t)(t.x1 - t.x0);
      const double* c = &t.contrib[3 * off];
      double xyz[3] = {0.412453 * c[0] + 0.357580 * c[1] + 0.180423 * c[2], 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2],
                       0.019334 * c[0] + 0.119193 * c[1] + 0.950227 * c[2]};  // rgb_to_xyz spectrum.rs:2084-2090
      double* px = &film[4 * ((size_t)(x - f.crop[0]) + (size_t)(y - f.crop[1]) * (size_t)W)];
      for (int i = 0; i < 3; i++) {
#pragma omp atomic
        px[i] += xyz[i];
#pragma omp atomic
        px[3] += t.wsum[off];
      }
    }
}

void check_supported(const rrt_scene_desc* d) {
  if (d->abi_version != RRT_ABI_VERSION) throw OraclePanic{"scene desc ABI mismatch"};
  if (d->sampler.type != RRT_SAMPLER_HALTON && d->sampler.type != RRT_SAMPLER_STRATIFIED) throw OraclePanic{"oracle: unknown sampler type"};
}

template <typename F>
int guarded(F&& fn) {
  try { fn(); return 0; }
  catch (const OraclePanic& p) { g_err = "panic: " + p.msg; return RRT_EPANIC; }
  catch (const std::exception& e) { g_err = e.what(); return RRT_EINVAL; }
}

}  // namespace

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

// Known-answer surfaces -----------------------------------------------------------------------------------
uint64_t oracle_halton_index(const rrt_scene_desc* d, int64_t px, int64_t py, uint64_t sample_num) {
  Sampler s; s.h = &d->sampler;
  s.start_pixel(px, py);
  return s.get_index_for_sample(sample_num);
}
double oracle_halton_dim(const rrt_scene_desc* d, uint64_t index, uint32_t dim) {
  Sampler s; s.h = &d->sampler;
  return s.sample_dimension(index, dim);
}
double oracle_radical_inverse(int base_index, uint64_t a) { return radical_inverse(base_index, a); }
// geometry.rs tests (test_vec3 / test_bound3) are exercised through these
void oracle_vec3_ops(const double* a, const double* b, double* out_dot, double* out_cross, double* out_len2_a) {
  V3 A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
  *out_dot = dot(A, B);
  V3 c = cross(A, B);
  out_cross[0] = c.x; out_cross[1] = c.y; out_cross[2] = c.z;
  *out_len2_a = len2(A);
}
int oracle_sphere_intersect_p(const rrt_scene_desc* d, uint32_t sphere, const double* o, const double* dir) {
  Scene sc{d, false};
  Ray r = ray_new(V3(o[0], o[1], o[2]), V3(dir[0], dir[1], dir[2]), INF);
  return sphere_intersect_p(sphere_ref(sc, sphere), r) ? 1 : 0;
}

// Bsdf of material `material` on a surface with n = ns = +z, dpdu = +x (local frame = world frame), for unit tests of
// the BxDFs against closed forms: out = {f.rgb, pdf(wo, wi), sampled wi.xyz, sampled f.rgb, sampled pdf, sampled flags,
// bsdf.eta, number of lobes, num_components(ALL & !SPECULAR)}.
int oracle_bsdf_eval(const rrt_scene_desc* d, uint32_t material, int allow_multiple_lobes, const double* wo_in, const double* wi_in,
                     double u0, double u1, double* out16) {
  return guarded([&]() {
    if (material >= d->n_materials) throw OraclePanic{"oracle_bsdf_eval: material index out of range"};
    // a prim that carries this material is needed by compute_scattering(); build a private one-prim view
    rrt_scene_desc view = *d;
    rrt_prim pr{};
    pr.type = RRT_PRIM_TRIANGLE; pr.shape = 0; pr.instance = -1; pr.material = material;
    view.prims = &pr; view.n_prims = 1;
    Scene sv{&view, false};
    SI si;
    si_new(&si, V3(), 0.0, 0.0, V3(wo_in[0], wo_in[1], wo_in[2]), V3(1.0, 0.0, 0.0), V3(0.0, 1.0, 0.0));
    si.prim = 0; si.valid = true;
    Bsdf b;
    compute_scattering(sv, si, RayDiff(), &b, allow_multiple_lobes != 0);
    V3 wo(wo_in[0], wo_in[1], wo_in[2]), wi(wi_in[0], wi_in[1], wi_in[2]);
    Rgb f = b.f(wo, wi, BXDF_ALL);
    out16[0] = f.c[0]; out16[1] = f.c[1]; out16[2] = f.c[2];
    out16[3] = b.pdf(wo, wi, BXDF_ALL);
    V3 ws;
    double ps = 0.0;
    uint8_t fl = 0;
    Rgb fs = b.sample_f(wo, &ws, u0, u1, &ps, BXDF_ALL, &fl);
    out16[4] = ws.x; out16[5] = ws.y; out16[6] = ws.z;
    out16[7] = fs.c[0]; out16[8] = fs.c[1]; out16[9] = fs.c[2];
    out16[10] = ps; out16[11] = (double)fl; out16[12] = b.eta; out16[13] = (double)b.n;
    out16[14] = (double)b.num_components(BXDF_ALL & ~BXDF_SPECULAR);
    out16[15] = 0.0;
  });
}

// Texture::evaluate of textures[tex] at a hand-made interaction: si15 = {p.xyz, uv, dpdx.xyz, dpdy.xyz, dudx, dvdx, dudy, dvdy}
int oracle_texture_eval(const rrt_scene_desc* d, int32_t tex, const double* si15, double* out3) {
  return guarded([&]() {
    if (tex < 0 || (size_t)tex >= d->n_textures) throw OraclePanic{"oracle_texture_eval: texture index out of range"};
    SI si;
    si.p = V3(si15[0], si15[1], si15[2]); si.u = si15[3]; si.v = si15[4];
    si.dpdx = V3(si15[5], si15[6], si15[7]); si.dpdy = V3(si15[8], si15[9], si15[10]);
    si.dudx = si15[11]; si.dvdx = si15[12]; si.dudy = si15[13]; si.dvdy = si15[14];
    Rgb v = tex_eval(d, tex, si);
    out3[0] = v.c[0]; out3[1] = v.c[1]; out3[2] = v.c[2];
  });
}
// SurfaceInteraction::compute_differentials: in24 = {n, p, dpdu, dpdv, rx_origin, rx_direction, ry_origin, ry_direction},
// out10 = {dpdx.xyz, dpdy.xyz, dudx, dvdx, dudy, dvdy}
int oracle_surface_differentials(const double* in24, double* out10) {
  return guarded([&]() {
    auto v = [&](int k) { return V3(in24[3 * k], in24[3 * k + 1], in24[3 * k + 2]); };
    SI si;
    si.n = v(0); si.p = v(1); si.dpdu = v(2); si.dpdv = v(3);
    RayDiff rd;
    rd.has = true; rd.rxo = v(4); rd.rxd = v(5); rd.ryo = v(6); rd.ryd = v(7);
    compute_differentials(&si, rd);
    out10[0] = si.dpdx.x; out10[1] = si.dpdx.y; out10[2] = si.dpdx.z; out10[3] = si.dpdy.x; out10[4] = si.dpdy.y; out10[5] = si.dpdy.z;
    out10[6] = si.dudx; out10[7] = si.dvdx; out10[8] = si.dudy; out10[9] = si.dvdy;
  });
}

// BVHAccel::intersect / intersect_p on ray batches (rays are used as given: d is NOT re-normalised) --------
int oracle_trace_closest(const rrt_scene_desc* d, const double* o, const double* dir, const double* tmax, size_t n,
                         double* t_out, int32_t* prim_out, double* u_out, double* v_out, uint32_t* nodes, uint32_t* prims,
                         double* p_out /* optional 3n: hit point */, double* n_out /* optional 3n: ist.n */,
                         int mode /* bit0: flat */, double* margin_out /* optional n */) {
  return guarded([&]() {
    Scene sc{d, (mode & 1) != 0};
    std::string panic;   // an exception must not leave the OpenMP region: first message is re-thrown after the loop
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; i++) {
      Ray r; r.o = V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.d = V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]); r.t_max = tmax[i];
      SI si; HitInfo hi;
      uint32_t nn = 0, np = 0;
      double margin = INF;
      bool hit = false;
      try { hit = scene_intersect(sc, &r, &si, &hi, nullptr, &nn, &np, margin_out ? &margin : nullptr); }
      catch (const OraclePanic& e) {
#pragma omp critical
        if (panic.empty()) panic = e.msg;
      }
      if (margin_out) margin_out[i] = margin;
      t_out[i] = r.t_max;
      prim_out[i] = hit ? hi.order_index : -1;
      if (u_out) u_out[i] = hi.u;
      if (v_out) v_out[i] = hi.v;
      if (nodes) nodes[i] = nn;
      if (prims) prims[i] = np;
      if (p_out) { p_out[3 * i] = si.p.x; p_out[3 * i + 1] = si.p.y; p_out[3 * i + 2] = si.p.z; }
      if (n_out) { n_out[3 * i] = si.n.x; n_out[3 * i + 1] = si.n.y; n_out[3 * i + 2] = si.n.z; }
    }
    if (!panic.empty()) throw OraclePanic{panic};
  });
}
int oracle_trace_any(const rrt_scene_desc* d, const double* o, const double* dir, const double* tmax, size_t n, uint8_t* occluded,
                     uint32_t* nodes, uint32_t* prims, int mode, double* margin_out) {
  return guarded([&]() {
    Scene sc{d, (mode & 1) != 0};
    std::string panic;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; i++) {
      Ray r; r.o = V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.d = V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]); r.t_max = tmax[i];
      uint32_t nn = 0, np = 0;
      double margin = INF;
      try { occluded[i] = scene_intersect_p(sc, r, nullptr, &nn, &np, margin_out ? &margin : nullptr) ? 1 : 0; }
      catch (const OraclePanic& e) {
#pragma omp critical
        if (panic.empty()) panic = e.msg;
      }
      if (margin_out) margin_out[i] = margin;
      if (nodes) nodes[i] = nn;
      if (prims) prims[i] = np;
    }
    if (!panic.empty()) throw OraclePanic{panic};
  });
}

// get_camerasample + generate_ray_differential for [pixel in rect][sample_num in s0..s1) ----------------------
int oracle_camera_samples(const rrt_scene_desc* d, const int32_t rect[4], uint64_t s0, uint64_t s1, double* dims5, double* ray_od6, double* weight) {
  return guarded([&]() {
    check_supported(d);
    Camera cam{&d->camera, &d->film};
    size_t k = 0;
    for (int y = rect[1]; y < rect[3]; y++)
      for (int x = rect[0]; x < rect[2]; x++) {
        Sampler s; s.h = &d->sampler; s.xres = d->film.xres;
        s.start_pixel(x, y);
        for (uint64_t sn = s0; sn < s1; sn++, k++) {
          s.dimension = 0; s.cur1d = s.cur2d = 0;
          s.current_pixel_sample_index = sn;
          if (d->sampler.type == RRT_SAMPLER_HALTON) s.interval_sample_index = s.get_index_for_sample(sn);
          double f0, f1, l0, l1, tm;
          s.get_2d(&f0, &f1); s.get_2d(&l0, &l1); tm = s.get_1d();
          double* dd = &dims5[5 * k];
          dd[0] = f0; dd[1] = f1; dd[2] = l0; dd[3] = l1; dd[4] = tm;
          Ray r;
          double w = cam.generate_ray_differential((double)x + f0, (double)y + f1, l0 + 0.5, l1 + 0.5, &r);  // Q5
          weight[k] = w;
          double* ro = &ray_od6[6 * k];
          if (w > 0.0) { ro[0] = r.o.x; ro[1] = r.o.y; ro[2] = r.o.z; ro[3] = r.d.x; ro[4] = r.d.y; ro[5] = r.d.z; }
          else for (int q = 0; q < 6; q++) ro[q] = 0.0;
        }
      }
  });
}

// SamplerIntegrator::si_render integrator/mod.rs:48-139 restricted to the pixels of `rect`.
// film: W*H*4 doubles (xyz sums + filter_weight_sum as Film::pixels holds them), accumulated (+=).
// faithful_sampler_rebuild != 0 re-derives the digit permutation table per tile like the reference's
// per-tile sampler.build() (time only; the table content is identical because it is seeded).
int oracle_render_rect(const rrt_scene_desc* d, const int32_t rect[4], double* film, rrt_render_stats* stats, int n_threads,
                       int faithful_sampler_rebuild, int mode /* bit0: flat */) {
  return guarded([&]() {
    check_supported(d);
    Scene sc{d, (mode & 1) != 0};
    const rrt_film& f = d->film;
    Camera cam{&d->camera, &f};
    const int tile = 16;
    int sb0 = f.sample_bounds[0], sb1 = f.sample_bounds[1], sb2 = f.sample_bounds[2], sb3 = f.sample_bounds[3];
    int ntx = ((sb2 - sb0) + tile - 1) / tile, nty = ((sb3 - sb1) + tile - 1) / tile;
    uint64_t cam_samples = 0, cam_rays = 0;
    Counters total;
    std::string panic;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)n_threads;
#pragma omp parallel
    {
      Counters cnt;
      uint64_t my_samples = 0, my_rays = 0;
      Integ integ{sc, d->integrator, UniformLightDistrib(), &cnt};
      integ.distrib.init(d->n_lights);  // PathIntegrator::preprocess path.rs:47-49
#pragma omp for schedule(dynamic, 1) collapse(2)
      for (int tx = 0; tx < ntx; tx++)
        for (int ty = 0; ty < nty; ty++) {
          int x0 = sb0 + tx * tile, x1 = std::min(x0 + tile, sb2), y0 = sb1 + ty * tile, y1 = std::min(y0 + tile, sb3);
          if (x1 <= rect[0] || x0 >= rect[2] || y1 <= rect[1] || y0 >= rect[3]) continue;
          try {
            if (faithful_sampler_rebuild) {
              // cost model of Halton::new per tile (halton.rs:23-25): rebuild + reshuffle of the 3.67M-entry table
              std::vector<uint16_t> tmp(d->sampler.n_perms);
              uint64_t st = d->sampler.perm_seed | 1;
              size_t p = 0;
              for (int i = 0; i < 1000; i++) {
                uint32_t c = primes().p[i];
                for (uint32_t j = 0; j < c; j++) tmp[p + j] = (uint16_t)j;
                for (uint32_t j = 0; j < c; j++) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; uint32_t o = j + (uint32_t)((st >> 33) % (c - j)); std::swap(tmp[p + j], tmp[p + o]); }
                p += c;
              }
              volatile uint16_t sink = tmp[p - 1]; (void)sink;
            }
            Sampler smp; smp.h = &d->sampler; smp.xres = d->film.xres;
            FilmTile ft = get_film_tile(f, x0, y0, x1, y1);
            for (int y = y0; y < y1; y++)
              for (int x = x0; x < x1; x++) {
                smp.start_pixel(x, y);
                if (x < rect[0] || x >= rect[2] || y < rect[1] || y >= rect[3]) continue;
                if (!(x >= 0 && x < f.xres && y >= 0 && y < f.yres)) continue;  // pixel_bounds, integrator/mod.rs:82
                while (smp.start_next_sample()) {
                  double f0, f1, l0, l1;
                  smp.get_2d(&f0, &f1); smp.get_2d(&l0, &l1); (void)smp.get_1d();
                  double pfx = (double)x + f0, pfy = (double)y + f1;
                  Ray ray;
                  RayDiff rdiff;
                  double w = cam.generate_ray_differential(pfx, pfy, l0 + 0.5, l1 + 0.5, &ray, &rdiff);
                  // integrator/mod.rs:94-96 (scales the default-initialised differentials of a dead sample too; unobservable)
                  if (rdiff.has) scale_differentials(ray, &rdiff, 1.0 / std::sqrt((double)smp.samples_per_pixel()));
                  my_samples++;
                  Rgb L;
                  if (w > 0.0) { my_rays++; L = integ.li(ray, rdiff, smp); }
                  if (L.has_nan()) L = Rgb();
                  else if (L.y() < -1e-5) L = Rgb();
                  else if (std::isinf(L.y())) L = Rgb();
                  add_sample(f, ft, pfx, pfy, L, w);
                }
              }
            merge_film_tile(f, ft, film);
          } catch (const OraclePanic& p) {
#pragma omp critical
            panic = p.msg;
          }
        }
#pragma omp critical
      {
        total.nodes += cnt.nodes; total.prims += cnt.prims; total.closest += cnt.closest; total.any += cnt.any;
        cam_samples += my_samples; cam_rays += my_rays;
      }
    }
    if (!panic.empty()) throw OraclePanic{panic};
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->camera_samples = cam_samples; stats->camera_rays = cam_rays;
      stats->closest_queries = total.closest; stats->any_queries = total.any;
      stats->nodes_visited = total.nodes; stats->prims_tested = total.prims;
    }
  });
}

}  // extern "C"
