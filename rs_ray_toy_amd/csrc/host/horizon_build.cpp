// Horizon tables of the fp32 path integrator (device/dkernels.hpp k_shade_path "Horizon cull"): the host builder.
#include "horizon_build.hpp"
#include <algorithm>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>

namespace rrtd {
// On an open scene most bounce rays leave: on BASELINE config 4 three of four rays spawned at a hit find nothing, each after walking ~50 nodes (the ancestors of its own
// leaf all contain the origin). A ray p + t d from a point p of triangle T can only meet geometry points q with (q - p) parallel to d - the same azimuth about an axis e,
// the same elevation. So, per triangle T, hemisphere (+e / -e) and azimuth sector s (16 wedges, hz_sector()):
//     H[T][+-][s] >= sup { +-(q - p).e / |q - p| :  p in T, q in any OTHER triangle, q != p, azimuth(q - p) in s }
// and a ray from T whose |d.e| exceeds H in its hemisphere and sector provably misses everything: BVHAccel::intersect would return false. The bound must hold for
// geometry that TOUCHES T as well (its neighbours: a valley wall starts on T's own edge), which is what shapes it:
//  * every difference q - p is a convex combination of the nine vertex differences g = u_j - t_i (zero for shared vertices);
//  * no shared vertex: (q - p).e <= max g.e =: vk and the horizontal part of q - p is at least the distance hmin of the origin from the convex hull of the projected g
//    (the smallest distance to a segment between two of them, valid when all of them lie on the nearest point's side; else 0: an overhang), so the sine of the
//    elevation is at most vk / sqrt(vk^2 + hmin^2); azimuths: between the most clockwise and the most counter-clockwise projected g (everything if hmin = 0);
//  * near pairs and shared vertices: the directions form the CONE spanned by the non-zero g; for any n with n.g <= 0 for every generator, every direction v of the cone has
//    n.v <= 0, which bounds the rise per unit of horizontal distance along each azimuth, sector by sector; candidates n = the planes through two generators - the planes
//    of T and of the neighbour are among them, which makes the bound the neighbour's slope in a valley - all 72 of them for a cone that is not pointed, the (at most
//    nine) faces of the generators' convex hull seen along their mean direction for one that is;
//  * far geometry is bounded node by node with the plain formula on boxes (B - bbox(T)), walking the tree BEST FIRST (a heap keyed by the bound) and pruning what cannot
//    raise H in any sector it touches;
//  * the ray's origin is only NEAR T (HzTables::tau: the kernel's edge guard), so geometry that comes within 4 rho of T's interior - a crossing or resting triangle,
//    one a hair above or below, a neighbour folded back over T at a shallow angle - takes T's tables away (the contact rules in pair_bound()).
// e = the axis along which the root box is thinnest (a terrain's "up"). Stored as ceil(254 (H + kHzMargin)) + 1 per byte (255 = never); the margin (0.2 degrees at the
// horizon) covers the fp32 rounding of the ray's direction and of the generators' plane tests. Exactness is tested: tests/test_horizon.py (no GPU: the builder's own
// brute force and the oracle's traversal, origins on and rho off the triangles; terrains, boxes in every kind of contact) and, on the device, frames with and without
// the tables identical bit for bit (tests/test_gpu_parity.py::test_horizon_cull_changes_nothing, ..._at_baseline_size).
constexpr double kHzMargin = 0.004;
// a triangle whose nearest point is at more than 1 / (kFar - 1) = 5 times the spread of the vertex differences gets the plain rise-over-distance bound only
constexpr double kFar = 1.2;
HzTables build_horizons(const HzNode* nodes, size_t n_nodes, const HzTri* tris, size_t nt, long check_rays) {
  HzTables res;
  res.bytes.assign(nt * 32u, 255u);
  std::vector<uint8_t>& out = res.bytes;
  if (n_nodes == 0 || nt == 0) return res;
  uint32_t k = 0;
  for (uint32_t c = 1; c < 3; c++) if (nodes[0].bmax[c] - nodes[0].bmin[c] < nodes[0].bmax[k] - nodes[0].bmin[k]) k = c;
  res.axis = k;
  res.tau.assign(nt, INFINITY);
  double coord_max = 0.0;
  for (int c = 0; c < 3; c++) coord_max = std::max({coord_max, std::fabs((double)nodes[0].bmin[c]), std::fabs((double)nodes[0].bmax[c])});
  const double rho = std::ldexp(coord_max, -24);
  const uint32_t ia = (k + 1u) % 3u, ib = (k + 2u) % 3u;   // (a, b): the other two axes in cyclic order - what the kernel hands to hz_sector
  const double kPi = 3.14159265358979323846;
  // angular range of every sector id in degrees of atan2(b, a), from the definition of hz_sector: first-quadrant wedges, then mirrored by the sign bits
  double w_lo[16], w_hi[16];
  for (uint32_t id = 0; id < 16u; id++) {
    const bool sa = id & 8u, sb = id & 4u, sw = id & 2u, sub = id & 1u;
    double lo = !sw ? (sub ? 22.5 : 0.0) : (sub ? 45.0 : 67.5), hi = lo + 22.5;
    if (sa) { const double l2 = 180.0 - hi, h2 = 180.0 - lo; lo = l2; hi = h2; }
    if (sb) { const double l2 = -hi, h2 = -lo; lo = l2; hi = h2; }
    w_lo[id] = lo; w_hi[id] = hi;
  }
  // Sector sets by comparisons alone (no atan2): position of every wedge in angular order (0 = [-180, -157.5] ... 15 = [157.5, 180]); a point on a wedge's border
  // must mark both wedges (the kernel's float comparisons may send such a ray either way), so the ends of an interval are turned outward by 0.05 degrees first
  uint32_t order_of[16], id_at[16];
  for (uint32_t id = 0; id < 16u; id++) { order_of[id] = (uint32_t)std::floor((0.5 * (w_lo[id] + w_hi[id]) + 180.0) / 22.5); id_at[order_of[id]] = id; }
  const double kc = std::cos(0.05 * kPi / 180.0), ks = std::sin(0.05 * kPi / 180.0);
  // wedges met by the azimuths from direction lo counter-clockwise to direction hi (less than a half turn apart), each end turned outward by the margin
  auto between = [&](const double* lo, const double* hi) -> uint32_t {
    const uint32_t o_lo = order_of[hz_sector((float)(lo[0] * kc + lo[1] * ks), (float)(lo[1] * kc - lo[0] * ks))];   // turned clockwise
    const uint32_t o_hi = order_of[hz_sector((float)(hi[0] * kc - hi[1] * ks), (float)(hi[0] * ks + hi[1] * kc))];   // turned counter-clockwise
    // the float conversion can move an end across a border the double one is on: take the wedges of the unturned ends too
    const uint32_t p_lo = order_of[hz_sector((float)lo[0], (float)lo[1])], p_hi = order_of[hz_sector((float)hi[0], (float)hi[1])];
    uint32_t m = (1u << id_at[p_lo]) | (1u << id_at[p_hi]);
    for (uint32_t o = o_lo, n = 0; n < 16u; o = (o + 1u) & 15u, n++) { m |= 1u << id_at[o]; if (o == o_hi) break; }
    return m;
  };
  // the same for a rectangle [a0, a1] x [b0, b1] that does not hold the origin: which two corners are the extreme ones follows from where the rectangle lies
  // (turned into the half plane a > 0: the lowest azimuth is at (a1, b0) if b0 >= 0, else at (a0, b0); the highest at (a1, b1) if b1 <= 0, else at (a0, b1))
  auto rect_mask = [&](double a0, double a1, double b0, double b1) -> uint32_t {
    int rot;   // quarter turns that bring the rectangle to the right of the origin
    double A0, A1, B0, B1;
    if (a0 > 0.0) { rot = 0; A0 = a0; A1 = a1; B0 = b0; B1 = b1; }
    else if (a1 < 0.0) { rot = 2; A0 = -a1; A1 = -a0; B0 = -b1; B1 = -b0; }
    else if (b0 > 0.0) { rot = 3; A0 = b0; A1 = b1; B0 = -a1; B1 = -a0; }     // (x, y) -> (y, -x)
    else if (b1 < 0.0) { rot = 1; A0 = -b1; A1 = -b0; B0 = a0; B1 = a1; }     // (x, y) -> (-y, x)
    else return 0xffffu;
    double lo[2] = {B0 >= 0.0 ? A1 : A0, B0}, hi[2] = {B1 <= 0.0 ? A1 : A0, B1};
    for (int r = 0; r < rot; r++) { const double lx = lo[0], hx = hi[0]; lo[0] = lo[1]; lo[1] = -lx; hi[0] = hi[1]; hi[1] = -hx; }   // back: (x, y) -> (y, -x), `rot` times
    return between(lo, hi);
  };
  // unit vectors of every wedge's two borders, turned outward by the same margin: the largest cosine between a direction and a wedge needs no trigonometry
  double e_lo[16][2], e_hi[16][2];
  for (uint32_t id = 0; id < 16u; id++) {
    e_lo[id][0] = std::cos((w_lo[id] - 0.05) * kPi / 180.0); e_lo[id][1] = std::sin((w_lo[id] - 0.05) * kPi / 180.0);
    e_hi[id][0] = std::cos((w_hi[id] + 0.05) * kPi / 180.0); e_hi[id][1] = std::sin((w_hi[id] + 0.05) * kPi / 180.0);
  }
  std::atomic<uint64_t> open_sum{0};
  struct Bound { double up, dn; uint32_t mask; };   // sines of the largest elevation above / below the horizontal, sectors concerned
  // per-sector bounds of one pair of triangles (sectors outside `mask` are not concerned)
  struct PairBound { double up[16], dn[16]; uint32_t mask; bool contact; };
  auto pair_bound = [&](const HzTri& T, const HzTri& U, PairBound* pb, const double* Hup, const double* Hdn) {
    const float* tv[3] = {T.p[0], T.p[1], T.p[2]};
    const float* uv[3] = {U.p[0], U.p[1], U.p[2]};
    double g[9][3]; int ng = 0, n_zero = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      const double v[3] = {(double)uv[j][0] - (double)tv[i][0], (double)uv[j][1] - (double)tv[i][1], (double)uv[j][2] - (double)tv[i][2]};
      if (v[0] == 0.0 && v[1] == 0.0 && v[2] == 0.0) { n_zero++; continue; }
      g[ng][0] = v[0]; g[ng][1] = v[1]; g[ng][2] = v[2]; ng++;
    }
    pb->mask = 0u; pb->contact = false;
    if (ng == 0) return;   // (the same three points)
    // A neighbour that shares a vertex or an edge and FOLDS BACK over T at a shallow angle (a lid, a fin): near the shared feature it is closer to T's interior than an
    // origin's rounding, and the edge guard (tau) only keeps origins 4 rho from the edge - enough against a lid of slope >= 1/4 over T's plane (an origin within rho of
    // the plane is still under it), not against a shallower one. Such a T gets no tables. Seen in T's plane: the two triangles' projections overlap (no separating
    // edge normal), and the neighbour's free vertices rise by less than a quarter of their distance from the shared ones (or stand on both sides of the plane).
    if (n_zero > 0) {
      double e0[3], e1[3], nT[3];
      for (int c = 0; c < 3; c++) { e0[c] = (double)tv[1][c] - tv[0][c]; e1[c] = (double)tv[2][c] - tv[0][c]; }
      nT[0] = e0[1] * e1[2] - e0[2] * e1[1]; nT[1] = e0[2] * e1[0] - e0[0] * e1[2]; nT[2] = e0[0] * e1[1] - e0[1] * e1[0];
      const double ln = std::sqrt(nT[0] * nT[0] + nT[1] * nT[1] + nT[2] * nT[2]), l0 = std::sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]);
      if (ln > 0.0 && l0 > 0.0) {
        double bx[3], by[3];
        for (int c = 0; c < 3; c++) { nT[c] /= ln; bx[c] = e0[c] / l0; }
        by[0] = nT[1] * bx[2] - nT[2] * bx[1]; by[1] = nT[2] * bx[0] - nT[0] * bx[2]; by[2] = nT[0] * bx[1] - nT[1] * bx[0];
        double A[3][2], B[3][2], hB[3];
        bool sharedB[3] = {false, false, false};
        for (int i = 0; i < 3; i++) {
          double da[3], db[3];
          for (int c = 0; c < 3; c++) { da[c] = (double)tv[i][c] - tv[0][c]; db[c] = (double)uv[i][c] - tv[0][c]; }
          A[i][0] = da[0] * bx[0] + da[1] * bx[1] + da[2] * bx[2]; A[i][1] = da[0] * by[0] + da[1] * by[1] + da[2] * by[2];
          B[i][0] = db[0] * bx[0] + db[1] * bx[1] + db[2] * bx[2]; B[i][1] = db[0] * by[0] + db[1] * by[1] + db[2] * by[2];
          hB[i] = db[0] * nT[0] + db[1] * nT[1] + db[2] * nT[2];
          for (int j = 0; j < 3; j++) if (uv[i][0] == tv[j][0] && uv[i][1] == tv[j][1] && uv[i][2] == tv[j][2]) sharedB[i] = true;
        }
        const double size = std::max(l0, std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2])), eps = 1e-9 * size;
        bool separated = false;
        for (int which = 0; which < 2 && !separated; which++) for (int i = 0; i < 3 && !separated; i++) {
          const double (*P)[2] = which == 0 ? A : B;
          const double ax = -(P[(i + 1) % 3][1] - P[i][1]), ay = P[(i + 1) % 3][0] - P[i][0];
          double mnA = 1e300, mxA = -1e300, mnB = 1e300, mxB = -1e300;
          for (int v = 0; v < 3; v++) { const double a = A[v][0] * ax + A[v][1] * ay, b = B[v][0] * ax + B[v][1] * ay; mnA = std::min(mnA, a); mxA = std::max(mxA, a); mnB = std::min(mnB, b); mxB = std::max(mxB, b); }
          const double tol = eps * std::sqrt(ax * ax + ay * ay);
          if (mxA <= mnB + tol || mxB <= mnA + tol) separated = true;
        }
        if (!separated) {
          double slope = 1e300; bool pos = false, neg = false;
          for (int i = 0; i < 3; i++) {
            if (sharedB[i]) continue;
            double dmin = 1e300;
            for (int j = 0; j < 3; j++) if (sharedB[j]) dmin = std::min(dmin, std::sqrt((B[i][0] - B[j][0]) * (B[i][0] - B[j][0]) + (B[i][1] - B[j][1]) * (B[i][1] - B[j][1])));
            if (hB[i] > 0.0) pos = true; else if (hB[i] < 0.0) neg = true; else { pos = true; neg = true; }
            if (dmin > 0.0 && dmin < 1e300) slope = std::min(slope, std::fabs(hB[i]) / dmin);
          }
          if ((pos && neg) || slope < 0.25) { pb->contact = true; pb->mask = 0xffffu; for (int sct = 0; sct < 16; sct++) pb->up[sct] = pb->dn[sct] = 1.0; return; }
        }
      }
    }
    double ph[9][2];
    for (int i = 0; i < ng; i++) { ph[i][0] = g[i][ia]; ph[i][1] = g[i][ib]; }
    double vk = -1e300, vd = -1e300;
    for (int i = 0; i < ng; i++) { vk = std::max(vk, g[i][k]); vd = std::max(vd, -g[i][k]); }
    // distance of the origin from the convex hull of the projected differences: when the origin is outside the hull, the nearest point of the hull lies on one of its
    // edges, every segment between two of the points lies inside the hull, and the edges are among them - so it is the smallest distance to any of the segments;
    // and the origin IS outside exactly when every point lies on the nearest point's side (p.c > 0), else the distance is 0 (an overhang: anything goes)
    double hmin = 0.0, far2 = 0.0;
    bool fits = false;
    {
      double best2 = 1e300, cx = 0.0, cy = 0.0;
      for (int i = 0; i < ng; i++) {
        const double l2 = ph[i][0] * ph[i][0] + ph[i][1] * ph[i][1];
        far2 = std::max(far2, l2);
        if (l2 < best2) { best2 = l2; cx = ph[i][0]; cy = ph[i][1]; }
        for (int j = i + 1; j < ng; j++) {
          const double ex = ph[j][0] - ph[i][0], ey = ph[j][1] - ph[i][1], ee = ex * ex + ey * ey;
          if (!(ee > 0.0)) continue;
          const double tt = -(ph[i][0] * ex + ph[i][1] * ey);
          if (tt <= 0.0 || tt >= ee) continue;   // (nearest at an end point: the points' own turn)
          const double u = tt / ee, qx = ph[i][0] + u * ex, qy = ph[i][1] + u * ey, q2 = qx * qx + qy * qy;
          if (q2 < best2) { best2 = q2; cx = qx; cy = qy; }
        }
      }
      fits = best2 > 0.0;
      for (int i = 0; i < ng && fits; i++) if (!(ph[i][0] * cx + ph[i][1] * cy > 0.0)) fits = false;
      if (fits) hmin = std::sqrt(best2) * (1.0 - 1e-12);
    }
    auto simple = [&](double v) { return v > 0.0 ? (hmin > 0.0 ? v / std::sqrt(v * v + hmin * hmin) : 1.0) : 0.0; };
    pb->mask = 0xffffu;
    if (fits) {   // all in an open half plane: the most clockwise and the most counter-clockwise of the differences bound their azimuths
      int lo = 0, hi = 0;
      for (int i = 1; i < ng; i++) { if (ph[lo][0] * ph[i][1] - ph[lo][1] * ph[i][0] < 0.0) lo = i; if (ph[hi][0] * ph[i][1] - ph[hi][1] * ph[i][0] > 0.0) hi = i; }
      pb->mask = between(ph[lo], ph[hi]);
    }
    // Contact without a shared vertex: the footprints overlap (or come within 4 rho) and the vertical differences reach within 4 rho of zero - a triangle that crosses T,
    // rests on it, or lies a hair above or below it. The guard (HzTables::tau) keeps a ray's origin away from T's EDGES; nothing keeps it from geometry that close to T's
    // interior (the origin is only within rho of the plane, it may be on the far side of such a triangle): T gets no tables at all. (Exactly coplanar triangles the walk
    // never opens - their boxes bound the rise by 0 - are the device's business: it excludes them by plane id.)
    if (n_zero == 0 && hmin <= 4.0 * rho && vk >= -4.0 * rho && vd >= -4.0 * rho) { pb->contact = true; pb->mask = 0xffffu; for (int sct = 0; sct < 16; sct++) pb->up[sct] = pb->dn[sct] = 1.0; return; }
    // (1) no shared vertex: the largest rise over the smallest horizontal distance - one number for every sector the hull meets; and nothing finer for a triangle
    // well away from T (at more than half the differences' own spread: the half spaces below tell sectors apart, which such a pair hardly spans)
    const double all_up = n_zero == 0 ? simple(vk) : 1.0, all_dn = n_zero == 0 ? simple(vd) : 1.0;
    for (int sct = 0; sct < 16; sct++) { pb->up[sct] = all_up; pb->dn[sct] = all_dn; }
    if (n_zero == 0) {
      bool can = false;   // a pair whose plain bound cannot raise a horizon in any sector it concerns needs no tighter one
      for (uint32_t id = 0; id < 16u && !can; id++) if ((pb->mask & (1u << id)) && (all_up > Hup[id] || all_dn > Hdn[id])) can = true;
      if (!can) { pb->mask = 0u; return; }
      if (hmin > 0.0 && kFar * kFar * hmin * hmin > far2) return;   // far < kFar hmin
    }
    double glen[9];
    for (int i = 0; i < ng; i++) glen[i] = std::sqrt(g[i][0] * g[i][0] + g[i][1] * g[i][1] + g[i][2] * g[i][2]);
    // (2) the cone of the generators (with or without shared vertices: every q - p is a convex combination of them), bounded by each half space n.v <= 0 that holds
    // them all: along the horizontal direction of azimuth psi such a v = (h, z) has n_h.h + n_k z <= 0, i.e. z <= -(n_h.h) / n_k for n_k > 0 (a bound on the rise
    // per unit of horizontal distance, sector by sector: a wall rising towards the east does not raise the northern horizon) and -z <= -(n_h.h) / |n_k| for n_k < 0
    auto half_space = [&](double* n) {   // n: unit normal; applied when every generator lies in n.v <= 0
      if (n[k] == 0.0) return;
      for (int i = 0; i < ng; i++) if (n[0] * g[i][0] + n[1] * g[i][1] + n[2] * g[i][2] > 1e-9 * glen[i]) return;
      const double rr = std::sqrt(n[ia] * n[ia] + n[ib] * n[ib]), ank = std::fabs(n[k]);
      const double mx = rr > 0.0 ? -n[ia] / rr : 1.0, my = rr > 0.0 ? -n[ib] / rr : 0.0;   // the horizontal direction in which this half space is most open
      double* dst = n[k] > 0.0 ? pb->up : pb->dn;
      for (uint32_t id = 0; id < 16u; id++) {
        if (!(pb->mask & (1u << id))) continue;
        // max over the (widened) wedge of rr cos(psi - psi0): 1 inside it, else at the nearer border
        const bool inside = e_lo[id][0] * my - e_lo[id][1] * mx >= 0.0 && mx * e_hi[id][1] - my * e_hi[id][0] >= 0.0;
        const double cmax = inside ? 1.0 : std::max(e_lo[id][0] * mx + e_lo[id][1] * my, e_hi[id][0] * mx + e_hi[id][1] * my);
        const double z = rr * cmax / ank;
        const double bnd = z > 0.0 ? z / std::sqrt(1.0 + z * z) : 0.0;
        dst[id] = std::min(dst[id], bnd);
      }
    };
    auto try_pair = [&](int a, int b, int sign) {   // sign: +1 / -1 = that orientation of g_a x g_b only, 0 = both
      double n[3] = {g[a][1] * g[b][2] - g[a][2] * g[b][1], g[a][2] * g[b][0] - g[a][0] * g[b][2], g[a][0] * g[b][1] - g[a][1] * g[b][0]};
      const double ln = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(ln > 0.0)) return;
      for (int c = 0; c < 3; c++) n[c] /= ln;
      if (sign >= 0) half_space(n);
      for (int c = 0; c < 3; c++) n[c] = -n[c];
      if (sign <= 0) half_space(n);
    };
    // The candidates are the planes through two generators. A pointed cone (every generator within ~78 degrees of their mean direction m: any pair that shares no vertex
    // and is not on top of T) shows which pairs can hold: seen from the apex along m the generators are points of a plane, and only neighbours on their convex hull span
    // a face of the cone - at most nine planes instead of 72. (half_space() still checks each against every generator, so a wrong hull costs tightness, never safety.)
    bool done = false;
    if (n_zero == 0) {
      double m[3] = {0.0, 0.0, 0.0};
      for (int i = 0; i < ng; i++) for (int c = 0; c < 3; c++) m[c] += g[i][c] / glen[i];
      const double lm = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
      bool pointed = lm > 0.0;
      double w[9];
      for (int i = 0; i < ng && pointed; i++) { w[i] = (g[i][0] * m[0] + g[i][1] * m[1] + g[i][2] * m[2]) / lm; if (!(w[i] > 0.2 * glen[i])) pointed = false; }
      if (pointed) {
        // a basis of the plane across m
        const int sm = std::fabs(m[0]) < std::fabs(m[1]) ? (std::fabs(m[0]) < std::fabs(m[2]) ? 0 : 2) : (std::fabs(m[1]) < std::fabs(m[2]) ? 1 : 2);
        double ax[3] = {0.0, 0.0, 0.0}; ax[sm] = 1.0;
        double u[3] = {m[1] * ax[2] - m[2] * ax[1], m[2] * ax[0] - m[0] * ax[2], m[0] * ax[1] - m[1] * ax[0]};
        double v[3] = {m[1] * u[2] - m[2] * u[1], m[2] * u[0] - m[0] * u[2], m[0] * u[1] - m[1] * u[0]};
        double P[9][2];
        int idx[9];
        for (int i = 0; i < ng; i++) { P[i][0] = (g[i][0] * u[0] + g[i][1] * u[1] + g[i][2] * u[2]) / w[i]; P[i][1] = (g[i][0] * v[0] + g[i][1] * v[1] + g[i][2] * v[2]) / w[i]; idx[i] = i; }
        std::sort(idx, idx + ng, [&](int x, int y) { return P[x][0] < P[y][0] || (P[x][0] == P[y][0] && P[x][1] < P[y][1]); });
        auto turn = [&](int o, int a2, int b2) { return (P[a2][0] - P[o][0]) * (P[b2][1] - P[o][1]) - (P[a2][1] - P[o][1]) * (P[b2][0] - P[o][0]); };
        int hull[20], nh = 0;   // monotone chain; collinear points are KEPT OUT (a face needs its two extreme generators)
        for (int i = 0; i < ng; i++) { while (nh >= 2 && turn(hull[nh - 2], hull[nh - 1], idx[i]) <= 0.0) nh--; hull[nh++] = idx[i]; }
        const int lower = nh + 1;
        for (int i = ng - 2; i >= 0; i--) { while (nh >= lower && turn(hull[nh - 2], hull[nh - 1], idx[i]) <= 0.0) nh--; hull[nh++] = idx[i]; }
        nh--;   // (the first point again)
        if (nh >= 3) {
          for (int j = 0; j < nh; j++) try_pair(hull[j], hull[(j + 1) % nh], 0);
          done = true;
        }
      }
    }
    if (!done) for (int a = 0; a < ng; a++) for (int b = a + 1; b < ng; b++) try_pair(a, b, 0);
  };
  auto work = [&](size_t t0, size_t t1) {
    struct Open { double key; uint32_t node; Bound b; };
    std::vector<Open> heap;
    uint64_t open_local = 0;
    for (size_t ti = t0; ti < t1; ti++) {
      const HzTri& T = tris[ti];
      if (T.skip) continue;   // (mixed scenes build no tables at all: see the caller)
      double amin[3], amax[3];
      for (int c = 0; c < 3; c++) { amin[c] = std::min({(double)T.p[0][c], (double)T.p[1][c], (double)T.p[2][c]}); amax[c] = std::max({(double)T.p[0][c], (double)T.p[1][c], (double)T.p[2][c]}); }
      double Hup[16], Hdn[16], hup_min = 0.0, hdn_min = 0.0;
      for (int sct = 0; sct < 16; sct++) Hup[sct] = Hdn[sct] = 0.0;
      auto can_raise = [&](const Bound& b) { for (uint32_t id = 0; id < 16u; id++) if ((b.mask & (1u << id)) && (b.up > Hup[id] || b.dn > Hdn[id])) return true; return false; };
      auto box_bound = [&](const HzNode& nd) -> Bound {
        const double vk = (double)nd.bmax[k] - amin[k], vd = amax[k] - (double)nd.bmin[k];
        const double da = std::max(0.0, std::max((double)nd.bmin[ia] - amax[ia], amin[ia] - (double)nd.bmax[ia]));
        const double db = std::max(0.0, std::max((double)nd.bmin[ib] - amax[ib], amin[ib] - (double)nd.bmax[ib]));
        const double hmin = std::sqrt(da * da + db * db);
        Bound b{vk > 0.0 ? (hmin > 0.0 ? vk / std::sqrt(vk * vk + hmin * hmin) : 1.0) : 0.0, vd > 0.0 ? (hmin > 0.0 ? vd / std::sqrt(vd * vd + hmin * hmin) : 1.0) : 0.0, 0xffffu};
        if (b.up <= hup_min && b.dn <= hdn_min) { b.mask = 0u; return b; }   // cannot raise any sector: no need to know which ones it concerns
        if (hmin > 0.0) {   // the difference rectangle does not hold the origin: its four corners span its azimuths
          b.mask = rect_mask((double)nd.bmin[ia] - amax[ia], (double)nd.bmax[ia] - amin[ia], (double)nd.bmin[ib] - amax[ib], (double)nd.bmax[ib] - amin[ib]);
        }
        return b;
      };
      // best first: the node that could raise a horizon the most is opened next, so the sectors reach their final values early and everything lower is pruned
      // unopened (near first took ~10x the nodes: every far ridge raised the horizon a little more)
      heap.clear();
      auto push = [&](uint32_t ni) {
        hup_min = *std::min_element(Hup, Hup + 16); hdn_min = *std::min_element(Hdn, Hdn + 16);
        const Bound bb = box_bound(nodes[ni]);
        if (!can_raise(bb)) return;
        heap.push_back(Open{std::max(bb.up, bb.dn), ni, bb});
        std::push_heap(heap.begin(), heap.end(), [](const Open& x, const Open& y) { return x.key < y.key; });
      };
      push(0u);
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), [](const Open& x, const Open& y) { return x.key < y.key; });
        const Open op = heap.back(); heap.pop_back();
        if (!can_raise(op.b)) continue;   // (the horizons have risen since it was pushed)
        const uint32_t ni = op.node;
        const HzNode& nd = nodes[ni];
        const uint32_t np = nd.n_prims;
        if (np != 0u) {
          for (uint32_t j = 0; j < np; j++) {
            const size_t ui = (size_t)nd.offset + j;
            if (ui == ti || ui >= nt) continue;
            PairBound pb;
            pair_bound(T, tris[ui], &pb, Hup, Hdn);
            for (uint32_t id = 0; id < 16u; id++) if (pb.mask & (1u << id)) { Hup[id] = std::max(Hup[id], pb.up[id]); Hdn[id] = std::max(Hdn[id], pb.dn[id]); }
          }
        } else {
          push(ni + 1u); push(nd.offset);
        }
      }
      {   // smallest altitude = 2 x area / longest edge
        const double e0[3] = {(double)T.p[1][0] - T.p[0][0], (double)T.p[1][1] - T.p[0][1], (double)T.p[1][2] - T.p[0][2]}, e1[3] = {(double)T.p[2][0] - T.p[0][0], (double)T.p[2][1] - T.p[0][1], (double)T.p[2][2] - T.p[0][2]};
        const double e2[3] = {e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2]};
        const double cx = e0[1] * e1[2] - e0[2] * e1[1], cy = e0[2] * e1[0] - e0[0] * e1[2], cz = e0[0] * e1[1] - e0[1] * e1[0];
        const double area2 = std::sqrt(cx * cx + cy * cy + cz * cz);
        const double longest = std::sqrt(std::max({e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2], e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]}));
        if (area2 > 0.0 && longest > 0.0) { const double t4 = 4.0 * rho / (area2 / longest); res.tau[ti] = t4 < 3.0e38 ? (float)(t4 * (1.0 + 1e-6)) : INFINITY; }
      }
      for (uint32_t id = 0; id < 16u; id++) {
        const double hu = std::min(1.0, Hup[id] + kHzMargin), hd = std::min(1.0, Hdn[id] + kHzMargin);
        out[ti * 32u + id] = (uint8_t)std::min(255.0, std::ceil(254.0 * hu) + 1.0);
        out[ti * 32u + 16u + id] = (uint8_t)std::min(255.0, std::ceil(254.0 * hd) + 1.0);
        open_local += 255u - out[ti * 32u + id];
      }
    }
    open_sum += open_local;
  };
  {
    // up to 16 threads (a GPU process's share of the host), triangles handed out in chunks of 256 as the threads come free (a triangle under an overhang costs
    // several times one on open ground); a worker's exception (allocation) is rethrown on the caller's thread
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<size_t> next{0};
    std::exception_ptr failed;
    std::mutex failed_mu;
    auto guarded = [&]() {
      try {
        for (;;) { const size_t a = next.fetch_add(256); if (a >= nt) break; work(a, std::min(nt, a + 256)); }
      } catch (...) { std::lock_guard<std::mutex> lk(failed_mu); if (!failed) failed = std::current_exception(); }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < hw && (size_t)t * 256u < nt; t++) pool.emplace_back(guarded);
    guarded();
    for (auto& th : pool) th.join();
    if (failed) std::rethrow_exception(failed);
  }
  res.mean_open = (double)open_sum.load() / (255.0 * 16.0 * (double)nt);
  // Self-check (check_rays > 0; the handle passes RRT_HZ_CHECK=<rays>, tests/test_horizon.py calls it without a GPU): random rays from random points of random triangles
  // that the tables declare free are tested against EVERY other triangle (double precision Moeller-Trumbore, any t > 0, inclusive edges with a tolerance): none may hit.
  if (check_rays > 0) {
    const long n_rays = check_rays;
    uint64_t st = 0x243F6A8885A308D3ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    long tested = 0, hits = 0;
    for (long it = 0; it < 50 * n_rays && tested < n_rays; it++) {
      const size_t ti = (size_t)(rnd() * (double)nt) % nt;
      const HzTri& T = tris[ti];
      double b0 = rnd(), b1 = rnd();
      if (b0 + b1 > 1.0) { b0 = 1.0 - b0; b1 = 1.0 - b1; }
      if (rnd() < 0.3) { b0 *= 1e-3; }              // near an edge / a vertex: where touching neighbours matter
      if (rnd() < 0.3) { b1 *= 1e-3; }
      double p[3], d[3];
      for (int c = 0; c < 3; c++) p[c] = (double)T.p[0][c] * (1.0 - b0 - b1) + (double)T.p[1][c] * b0 + (double)T.p[2][c] * b1;
      const double z = 2.0 * rnd() - 1.0, ph = 2.0 * kPi * rnd(), rr = std::sqrt(std::max(0.0, 1.0 - z * z));
      d[k] = z; d[ia] = rr * std::cos(ph); d[ib] = rr * std::sin(ph);
      const uint32_t q = out[ti * 32u + (d[k] < 0.0 ? 16u : 0u) + hz_sector((float)d[ia], (float)d[ib])];
      if (!((float)std::fabs(d[k]) * 254.0f > (float)q)) continue;   // not declared free
      tested++;
      for (size_t ui = 0; ui < nt; ui++) {
        if (ui == ti) continue;
        const HzTri& U = tris[ui];
        const double e1[3] = {(double)U.p[1][0] - U.p[0][0], (double)U.p[1][1] - U.p[0][1], (double)U.p[1][2] - U.p[0][2]}, e2[3] = {(double)U.p[2][0] - U.p[0][0], (double)U.p[2][1] - U.p[0][1], (double)U.p[2][2] - U.p[0][2]};
        const double pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
        const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
        if (std::fabs(det) < 1e-300) continue;
        const double tv[3] = {p[0] - U.p[0][0], p[1] - U.p[0][1], p[2] - U.p[0][2]};
        const double uu = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
        const double qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
        const double vv = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det, tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
        if (uu >= -1e-9 && vv >= -1e-9 && uu + vv <= 1.0 + 1e-9 && tt > 1e-7) { hits++; break; }
      }
    }
    res.checked = tested; res.check_hits = hits;
  }
  return res;
}
}  // namespace rrtd

// Test and timing hook (NOT part of include/rrt.h; tests/test_horizon.py and tools/hz_time.py bind it with ctypes): the tables of a scene desc's world-space
// triangles, built from the same fp32 values the device holds for them (vertices cast, boxes rounded outward), on the CPU alone. Instanced triangles and spheres
// are skipped here (the handle bakes rigid instances first; a desc holding any of them gets its tables from the handle only).
#include <chrono>
#include "rrt.h"
extern "C" __attribute__((visibility("default"))) int rrt_internal_horizons(const rrt_scene_desc* d, uint8_t* out, uint32_t* axis, double* mean_open, long check_rays,
                                                                            long* checked, long* check_hits, double* seconds, float* tau_out) {
  if (!d || !out) return 1;
  std::vector<rrtd::HzNode> hn(d->n_bvh_nodes);
  std::vector<rrtd::HzTri> ht(d->n_prim_order);
  for (size_t i = 0; i < hn.size(); i++) {
    const rrt_bvh_node& n = d->bvh_nodes[i];
    for (int c = 0; c < 3; c++) {
      float lo = (float)n.bounds[c], hi = (float)n.bounds[3 + c];
      if ((double)lo > n.bounds[c]) lo = nextafterf(lo, -INFINITY);
      if ((double)hi < n.bounds[3 + c]) hi = nextafterf(hi, INFINITY);
      hn[i].bmin[c] = lo; hn[i].bmax[c] = hi;
    }
    hn[i].offset = n.offset; hn[i].n_prims = n.n_primitives;
  }
  for (size_t i = 0; i < ht.size(); i++) {
    const rrt_prim& pr = d->prims[d->prim_order[i]];
    ht[i].skip = (pr.type != RRT_PRIM_TRIANGLE || pr.instance >= 0) ? 1u : 0u;
    for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) ht[i].p[k][c] = ht[i].skip ? 0.0f : (float)d->positions[3 * (size_t)d->tris[pr.shape].v[k] + c];
  }
  const auto t0 = std::chrono::steady_clock::now();
  const rrtd::HzTables tab = rrtd::build_horizons(hn.data(), hn.size(), ht.data(), ht.size(), check_rays);
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();   // (with the self-check when one was asked for)
  std::copy(tab.bytes.begin(), tab.bytes.end(), out);
  if (tau_out) std::copy(tab.tau.begin(), tab.tau.end(), tau_out);
  if (axis) *axis = tab.axis;
  if (mean_open) *mean_open = tab.mean_open;
  if (checked) *checked = tab.checked;
  if (check_hits) *check_hits = tab.check_hits;
  return 0;
}
