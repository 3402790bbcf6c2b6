// Wavefront kernels (gfx950, wave64): camera ray generation, BVH traversal (closest / any), shading
// (next-event estimation + BSDF sampling + Russian roulette), film accumulation. One thread = one path
// slot; live paths are carried between kernels as index queues compacted with wave ballots and one atomic per block (block_push).
#pragma once
#include "dmath.hpp"
#include "dtexture.hpp"

namespace rrtd {

constexpr int kBlock = 256;
// device counters, one 128-byte line each (atomics on different queues must not share an L2 line)
enum { C_ACTIVE = 0, C_NEXT = 32, C_SHADOW = 64, C_CAMERA_RAYS = 96, C_ERROR = 128, C_WORK_CLOSEST = 160, C_WORK_SHADOW = 192, C_WORK_AUX = 224, C_SHADOW2 = 256,
       // per-XCD work cursors of the persistent traversal kernels: 8 lines each (one cursor per eighth of the queue)
       C_WORK8_CLOSEST = 288, C_WORK8_SHADOW = 544,
       // queue entries [0, C_TT_DONE) came from the camera workgroups in whole chunks (k_raygen_main_f32's chunk records), the rest from stage B
       C_TT_DONE = 800,
       // camera rays answered by the camera kernels (SceneDev::root_cull): closest-hit queries that never entered a queue
       C_CULLED = 832,
       // bounce rays the path shading kernel proves to leave the scene (SceneDev::horizon): closest-hit queries that never entered a queue
       C_SKY = 864, C_COUNT = 896 };
// shading kernels push to their queues once per block (measured: 256 <= 512 <= 1024 threads by 5 %: smaller blocks retire
// and refill a CU sooner, and one atomic per 256 paths no longer serialises)
template <typename R> struct ShadeBlock { static constexpr int n = 256; };
#ifndef RRT_SHADE_BLOCK
#define RRT_SHADE_BLOCK 256
#endif
template <> struct ShadeBlock<float> { static constexpr int n = RRT_SHADE_BLOCK; };
enum { ERR_SHADING_NORMAL = 1, ERR_BETA = 4, ERR_NULL_BSDF = 8, ERR_MIPMAP = 16, ERR_HALTON_DIMS = 32, ERR_ST_DIMS = 64, ERR_NO_LIGHTS = 128,
       ERR_KIND_SET = 256 /* a material produced a lobe outside the shading kernel's kind set (dmath.hpp): host-side selection error */ };
static_assert(ERR_ST_DIMS == kErrStDims, "error bit shared with dmath.hpp");
static_assert(ERR_HALTON_DIMS == kErrHaltonDims, "error bit shared with dmath.hpp");

// ------------------------------------------------------------------------------------------------------------
// BVH traversal: BVHAccel::intersect / intersect_p (bvh.rs:124-236), same node order, same leaf order, every
// accepted triangle overwrites the hit and t_max (Q10), triangle tests ignore t_max, box test uses it.
// ------------------------------------------------------------------------------------------------------------
template <typename R>
struct RayCtx {
  V3<R> o, d, inv;
  V3<R> lo;   // o = o_hi + lo (fp32 spawned rays; zero otherwise)
  int neg[3];
  R tmax;
};

// ---- pool records (dtypes.hpp Pools): 4-word loads / stores and raw-bit integers ------------------------------------
RRT_DEV float bits_to_real(uint32_t u, float) { return __uint_as_float(u); }
RRT_DEV double bits_to_real(uint32_t u, double) { return __longlong_as_double((long long)u); }
RRT_DEV uint32_t real_to_bits(float v) { return __float_as_uint(v); }
RRT_DEV uint32_t real_to_bits(double v) { return (uint32_t)__double_as_longlong(v); }
template <typename R> RRT_DEV typename Vec4T<R>::type mk4(R x, R y, R z, R w) { typename Vec4T<R>::type v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
template <typename R> RRT_DEV typename Vec4T<R>::type mk4u(R x, R y, R z, uint32_t w) { return mk4<R>(x, y, z, bits_to_real(w, R(0))); }
// Low word of a double-float origin (spawn_point()), 3 x 10 bits: lo_k is at most half an ulp of hi_k, so it is stored
// as round(lo_k / ulp(hi_k) * 1024) + 512 (resolution ulp / 1024 ~ 4e-9 at coordinates of ~35). Top bits 11 tag the word.
RRT_DEV uint32_t pack_lo(V3<float> hi, V3<float> lo) {
  auto q = [](float h, float l) -> uint32_t {
    const uint32_t e = (__float_as_uint(h) >> 23) & 0xffu;
    if (e < 64u || e > 250u) return 512u;
    const float scaled = l * __uint_as_float((287u - e) << 23);   // l * 2^(33 - (e - 127)) = l / ulp(h) * 1024
    const float r = rintf(scaled) + 512.0f;
    return (uint32_t)fminf(fmaxf(r, 0.0f), 1023.0f);
  };
  return 0xC0000000u | (q(hi.x, lo.x) << 20) | (q(hi.y, lo.y) << 10) | q(hi.z, lo.z);
}
RRT_DEV V3<float> unpack_lo(V3<float> hi, uint32_t w) {
  auto u = [](float h, uint32_t q) -> float {
    const uint32_t e = (__float_as_uint(h) >> 23) & 0xffu;
    if (e < 64u || e > 250u) return 0.0f;
    return ((float)(int)q - 512.0f) * __uint_as_float((e - 33u) << 23);
  };
  return V3<float>(u(hi.x, (w >> 20) & 1023u), u(hi.y, (w >> 10) & 1023u), u(hi.z, w & 1023u));
}
template <typename R> RRT_DEV void store_ray(typename Vec4T<R>::type* ro, typename Vec4T<R>::type* rd, uint32_t i, V3<R> o, V3<R> lo, V3<R> d, R tmax, int skip);
template <> RRT_DEV void store_ray<float>(float4* ro, float4* rd, uint32_t i, V3<float> o, V3<float> lo, V3<float> d, float tmax, int skip) {
  const bool has_lo = lo.x != 0.0f || lo.y != 0.0f || lo.z != 0.0f;
  ro[i] = make_float4(o.x, o.y, o.z, has_lo ? __uint_as_float(pack_lo(o, lo)) : tmax);
  rd[i] = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)skip));
}
template <> RRT_DEV void store_ray<double>(double4* ro, double4* rd, uint32_t i, V3<double> o, V3<double>, V3<double> d, double tmax, int skip) {
  ro[i] = mk4<double>(o.x, o.y, o.z, tmax);
  rd[i] = mk4u<double>(d.x, d.y, d.z, (uint32_t)skip);
}
// t_max and origin low word of a stored ray; `queue_tmax` = the t_max every spawned ray of this queue has
RRT_DEV void ray_tail(const float4& ro, float queue_tmax, float* tmax, V3<float>* lo) {
  const uint32_t w = __float_as_uint(ro.w);
  if ((w >> 30) == 3u) { *tmax = queue_tmax; *lo = unpack_lo(V3<float>(ro.x, ro.y, ro.z), w); }
  else { *tmax = ro.w; *lo = V3<float>(); }
}
RRT_DEV void ray_tail(const double4& ro, double, double* tmax, V3<double>* lo) { *tmax = ro.w; *lo = V3<double>(); }

// Bounds3::intersect_p geometry.rs:1767-1800 with gamma(3) of the arithmetic type
template <typename R>
RRT_DEV bool box_hit(const Node<R>& nd, const RayCtx<R>& r) {
  const R g = R(1) + R(2) * ((R(3) * Const<R>::machine_eps) / (R(1) - R(3) * Const<R>::machine_eps));
  R t_min = ((r.neg[0] ? nd.bmax[0] : nd.bmin[0]) - r.o.x) * r.inv.x;
  R t_max = ((r.neg[0] ? nd.bmin[0] : nd.bmax[0]) - r.o.x) * r.inv.x;
  R ty_min = ((r.neg[1] ? nd.bmax[1] : nd.bmin[1]) - r.o.y) * r.inv.y;
  R ty_max = ((r.neg[1] ? nd.bmin[1] : nd.bmax[1]) - r.o.y) * r.inv.y;
  t_max *= g;
  ty_max *= g;
  if (t_min > ty_max || ty_min > t_max) return false;
  if (ty_min > t_min) t_min = ty_min;
  if (ty_max < t_max) t_max = ty_max;
  R tz_min = ((r.neg[2] ? nd.bmax[2] : nd.bmin[2]) - r.o.z) * r.inv.z;
  R tz_max = ((r.neg[2] ? nd.bmin[2] : nd.bmax[2]) - r.o.z) * r.inv.z;
  tz_max *= g;
  if (t_min > tz_max || tz_min > t_max) return false;
  if (tz_min > t_min) t_min = tz_min;
  if (tz_max < t_max) t_max = tz_max;
  return (t_min < r.tmax) && (t_max > R(0));
}

// Triangle::intersect shape/triangle.rs:226-266 (Moller-Trumbore, E2 = p2 - p0)
template <typename R>
RRT_DEV bool tri_closest(const Tri<R>& t, const RayCtx<R>& r, R* th, R* uh, R* vh) {
  V3<R> p0(t.p0), p1(t.p1), p2(t.p2);
  V3<R> E1 = p1 - p0, E2 = p2 - p0;
  V3<R> P = cross(r.d, E2);
  R a = dot(E1, P);
  if (a > R(-0.0000001) && a < R(0.0000001)) return false;
  R f = rcp_r(a);
  V3<R> T = r.o - p0;
  if (sizeof(R) == 4) T = T + r.lo;   // double-float origin: o_hi - p0 is exact for nearby vertices
  R u = f * dot(T, P);
  if (u < R(0) || u > R(1)) return false;
  V3<R> Q = cross(T, E1);
  R v = f * dot(r.d, Q);
  if (v < R(0) || (u + v) > R(1)) return false;
  R tt = f * dot(E2, Q);
  if (tt < R(0.0000001)) return false;
  *th = tt; *uh = u; *vh = v;
  return true;
}
// Triangle::intersect_p shape/triangle.rs:167-205 (E2 = p2 - p1: Q11)
template <typename R>
RRT_DEV bool tri_any(const Tri<R>& t, const RayCtx<R>& r) {
  V3<R> p0(t.p0), p1(t.p1), p2(t.p2);
  V3<R> E1 = p1 - p0, E2 = p2 - p1;
  V3<R> P = cross(r.d, E2);
  R a = dot(E1, P);
  if (a > R(-0.0000001) && a < R(0.0000001)) return false;
  R f = rcp_r(a);
  V3<R> T = r.o - p0;
  if (sizeof(R) == 4) T = T + r.lo;   // double-float origin: o_hi - p0 is exact for nearby vertices
  R u = f * dot(T, P);
  if (u < R(0) || u > R(1)) return false;
  V3<R> Q = cross(T, E1);
  R v = f * dot(r.d, Q);
  if (v < R(0) || (u + v) > R(1)) return false;
  R tt = f * dot(E2, Q);
  if (tt < R(0.0000001)) return false;
  return true;
}

template <typename R>
RRT_DEV RayCtx<R> make_ctx(V3<R> o, V3<R> d, R tmax, V3<R> lo = V3<R>()) {
  RayCtx<R> c;
  c.o = o; c.d = d; c.tmax = tmax; c.lo = lo;
  c.inv = V3<R>(R(1) / d.x, R(1) / d.y, R(1) / d.z);
  c.neg[0] = c.inv.x < R(0); c.neg[1] = c.inv.y < R(0); c.neg[2] = c.inv.z < R(0);
  return c;
}

// Stack policies: a private array for trees no deeper than the reference's 64 entries, a strided global
// array for deeper (compat-built) trees the reference itself could not traverse.
struct PrivStack {
  uint32_t a[64];
  RRT_DEV void put(uint32_t i, uint32_t v) { a[i] = v; }
  RRT_DEV uint32_t get(uint32_t i) const { return a[i]; }
};
struct GlobStack {
  uint32_t* base;
  uint32_t stride;
  RRT_DEV void put(uint32_t i, uint32_t v) { base[(size_t)i * stride] = v; }
  RRT_DEV uint32_t get(uint32_t i) const { return base[(size_t)i * stride]; }
};

// Spawned rays start exactly on their triangle (spawn_ray applies no offset, Q8) and the reference relies on
// f64 to see that triangle again at t ~ 1e-15 < 1e-7. In fp32 the same test returns |t| ~ 1e-6, so the fp32
// path excludes the originating triangle instead — equivalent in exact arithmetic, because a ray meets its own
// triangle's plane only at t = 0. The same holds for every triangle *coplanar* with it (the other half of a
// quad, and — for shadow rays — the sheared triangle Triangle::intersect_p really tests, Q11), so the exclusion
// is by plane id. The f64 parity mode keeps the reference behaviour (skip = -1).
template <typename R> RRT_DEV int self_prim(int prim) { return sizeof(R) == 4 ? prim : -1; }
template <typename R> RRT_DEV uint32_t skip_plane_of(const SceneDev<R>& s, int skip) { return skip >= 0 ? s.tris[skip].plane : 0xffffffffu; }

// Sphere::intersect (sphere.rs:124-259) / Sphere::intersect_p (:51-108) behind GeometricPrimitive /
// TransformedPrimitive (primitives.rs:51-68,115-139), replayed transform by transform.
//   * the quadratic is solved with the object-space ray, the first p_hit uses the ray *handed to the sphere* (Q16);
//   * intersect_p tests the clipping planes against p_hit = 0, phi = 0 first (:75-83);
//   * neither compares t with ray.t_max (Q10).
// There is no epsilon in sphere.rs: a ray spawned on a sphere re-tests it with c = |o|^2 - r^2 ~ 1 ulp of either
// sign and re-hits its own sphere at t ~ 0 whenever the stored hit point landed inside, i.e. for about half of the
// spawned rays. The reference's pixels on spheres are that rounding noise. The f64 mode replays it bit for bit; the
// fp32 mode deliberately keeps the same test (no self exclusion, unlike triangles), so it shows the same noise in
// distribution (mean radiance within a few % of the oracle) though not pixel by pixel. (Tried in round 2 and dropped: solving the
// quadric in double from a double-float hit point inside the fp32 mode. The means did not come closer - a point's coin is not one flip
// but a chain over all later bounces at that point, which a 4e-9 origin does not replay - see DESIGN.md section 4.)
template <typename R>
struct SphereSI { V3<R> p, n, wo, sn, sdpdu; };
// the parts of SurfaceInteraction only textured scenes read: uv, geometric dpdu / dpdv (compute_differentials),
// shading dndu / dndv (specular ray differentials, integrator/mod.rs:188-196)
template <typename R>
struct SurfExt { R u, v; V3<R> dpdu, dpdv, sdpdv, sdndu, sdndv; };

// The rays the sphere code works with: the ray handed to the sphere (instance space, Q16) and the object-space ray.
template <typename R>
RRT_DEV void sphere_rays(const SphereDev<R>& S, V3<R> wo_, V3<R> wd_, V3<R>* ro, V3<R>* rd, V3<R>* oo, V3<R>* od) {
  *ro = wo_; *rd = wd_;
  if (S.has_inst) { *ro = aff_pt(S.imi, wo_); *rd = vnormalize(vnormalize(aff_vec(S.imi, wd_))); }   // xf_ray + Ray::new
  *oo = aff_pt(S.mi, *ro); *od = vnormalize(vnormalize(aff_vec(S.mi, *rd)));
}
// Hit decision. *branch = 1 when the accepted hit is the second root after the clipping test (sphere.rs:170-191):
// the hit record keeps it so that the shading kernel rebuilds the surface without deciding anything again.
template <typename R, bool ANY>
RRT_DEV bool sphere_prim_hit(const SphereDev<R>& S, V3<R> wo_, V3<R> wd_, R* t_out, R* branch) {
  V3<R> ro, rd, oo, od;
  sphere_rays(S, wo_, wd_, &ro, &rd, &oo, &od);
  const R a = od.x * od.x + od.y * od.y + od.z * od.z;
  const R b = R(2) * (od.x * oo.x + od.y * oo.y + od.z * oo.z);
  const R c = oo.x * oo.x + oo.y * oo.y + oo.z * oo.z - S.radius * S.radius;
  R t0, t1;
  if (!quadratic(a, b, c, &t0, &t1)) return false;
  const R kMax = R(1999999999.0);   // MAX_DIST misc.rs
  if (t0 > kMax || t1 <= R(0)) return false;
  R th = t0;
  if (t0 <= R(0)) { th = t1; if (th > kMax) return false; }
  V3<R> ph = ANY ? V3<R>() : ro + rd * th;
  R phi = R(0);
  if (!ANY) {
    if (ph.x == R(0) && ph.y == R(0)) ph.x = R(1e-5) * S.radius;
    phi = atan2(ph.y, ph.x);
    if (phi < R(0)) phi += R(2) * R(RRT_PI);
  }
  *branch = R(0);
  if ((S.z_min > -S.radius && ph.z < S.z_min) || (S.z_max < S.radius && ph.z > S.z_max) || (phi > S.phi_max)) {
    if (th == t1) return false;
    if (t1 > kMax) return false;
    th = t1;
    ph = oo + od * th;
    ph = ph * (S.radius / len(ph));
    if (ph.x == R(0) && ph.y == R(0)) ph.x = R(1e-5) * S.radius;
    phi = atan2(ph.y, ph.x);
    if (phi < R(0)) phi += R(2) * R(RRT_PI);
    if ((S.z_min > -S.radius && ph.z < S.z_min) || (S.z_max < S.radius && ph.z > S.z_max) || (phi > S.phi_max)) return false;
    *branch = R(1);
  }
  *t_out = th;
  return true;
}
// SurfaceInteraction of an accepted sphere hit (sphere.rs:192-259) from (t, branch)
template <typename R>
RRT_DEV void sphere_surface(const SphereDev<R>& S, V3<R> wo_, V3<R> wd_, R th, R branch, SphereSI<R>* si, SurfExt<R>* ext = nullptr) {
  V3<R> ro, rd, oo, od;
  sphere_rays(S, wo_, wd_, &ro, &rd, &oo, &od);
  V3<R> ph;
  if (branch == R(0)) ph = ro + rd * th;
  else { ph = oo + od * th; ph = ph * (S.radius / len(ph)); }
  if (ph.x == R(0) && ph.y == R(0)) ph.x = R(1e-5) * S.radius;
  const R theta = acos(clampr(ph.z / S.radius, R(-1), R(1)));
  const R z_radius = sqrt(ph.x * ph.x + ph.y * ph.y);
  const R inv_zr = R(1) / z_radius;
  const R cphi = ph.x * inv_zr, sphi = ph.y * inv_zr;
  const V3<R> dpdu(-S.phi_max * ph.y, S.phi_max * ph.x, R(0));
  const V3<R> dpdv = V3<R>(ph.z * cphi, ph.z * sphi, -S.radius * R(sin(theta))) * (S.theta_max - S.theta_min);
  // SurfaceInteraction::new (interaction.rs:131-181) then obj_to_world.t(&ist) (transform.rs:628-655)
  V3<R> n = vnormalize(cross(dpdu, dpdv));
  si->p = aff_pt(S.m, ph);
  si->wo = aff_vec(S.m, -od);
  si->n = aff_nrm(S.mi, n);
  si->sn = faceforward(nnormalize(aff_nrm(S.mi, n)), si->n);
  si->sdpdu = aff_vec(S.m, dpdu);
  if (ext) {   // sphere.rs:198-242
    R phi = atan2(ph.y, ph.x);
    if (phi < R(0)) phi += R(2) * R(RRT_PI);
    ext->u = phi / S.phi_max;
    ext->v = (theta - S.theta_min) / (S.theta_max - S.theta_min);
    const V3<R> d2pduu = V3<R>(ph.x, ph.y, R(0)) * -S.phi_max * S.phi_max;
    const V3<R> d2pduv = V3<R>(-sphi, cphi, R(0)) * (S.theta_max - S.theta_min) * ph.z * S.phi_max;
    const V3<R> d2pdvv = ph * -(S.theta_max - S.theta_min) * (S.theta_max - S.theta_min);
    const R E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
    const R e = dot(n, d2pduu), f = dot(n, d2pduv), g = dot(n, d2pdvv);
    const R inv_EFG2 = R(1) / (E * G - F * F);
    const V3<R> dndu = dpdu * ((f * F - e * G) * inv_EFG2) + dpdv * ((e * F - f * E) * inv_EFG2);
    const V3<R> dndv = dpdu * ((g * F - f * G) * inv_EFG2) + dpdv * ((f * F - g * E) * inv_EFG2);
    ext->dpdu = aff_vec(S.m, dpdu); ext->dpdv = aff_vec(S.m, dpdv);
    ext->sdndu = aff_nrm(S.mi, dndu); ext->sdndv = aff_nrm(S.mi, dndv);
    ext->sdpdv = ext->dpdv;
  }
  if (S.has_inst && !S.inst_identity) {   // TransformedPrimitive::intersect primitives.rs:131-136
    si->p = aff_pt(S.im, si->p);
    si->wo = aff_vec(S.im, si->wo);
    const V3<R> n2 = aff_nrm(S.imi, si->n);
    si->sn = faceforward(nnormalize(aff_nrm(S.imi, si->sn)), n2);
    si->n = n2;
    si->sdpdu = aff_vec(S.im, si->sdpdu);
    if (ext) {
      ext->dpdu = aff_vec(S.im, ext->dpdu); ext->dpdv = aff_vec(S.im, ext->dpdv);
      ext->sdndu = aff_nrm(S.imi, ext->sdndu); ext->sdndv = aff_nrm(S.imi, ext->sdndv);
      ext->sdpdv = ext->dpdv;
    }
  }
}

// triangle of a non-rigid instance (dtypes.hpp kInstBase): the ray as TransformedPrimitive hands it to the triangle
template <typename R> RRT_DEV bool is_inst_tri(const Tri<R>& t) { return (t.material & kInstFlag) != 0u; }
template <typename R> RRT_DEV uint32_t inst_of(const Tri<R>& t) { return (t.material >> 16) & 0x7fffu; }
template <typename R>
RRT_DEV RayCtx<R> inst_ray(const InstDev<R>& I, const RayCtx<R>& r) {   // world_to_prim.t(r): xf_ray + Ray::new (direction normalised twice)
  RayCtx<R> o;
  o.o = aff_pt(I.mi, r.o); o.d = vnormalize(vnormalize(aff_vec(I.mi, r.d))); o.lo = V3<R>(); o.tmax = r.tmax;
  return o;
}

template <typename R, typename Stack>
RRT_DEV int traverse_closest(const SceneDev<R>& s, RayCtx<R>& r, Stack& st, int skip, R* hu, R* hv, uint32_t* nn, uint32_t* np) {
  int hit = -1;
  const uint32_t skip_plane = skip_plane_of(s, skip);
  uint32_t to_visit = 0, cur = 0, cn = 0, cp = 0;
  if (s.n_nodes == 0) return -1;
  while (true) {
    const Node<R> nd = s.nodes[cur];
    cn++;
    if (box_hit(nd, r)) {
      const uint32_t nprims = nd.meta >> 2;
      if (nprims > 0) {
        for (uint32_t i = 0; i < nprims; i++) {
          cp++;
          R t, u, v;
          const Tri<R> tr = s.tris[nd.offset + i];
          if (tr.plane == kSphereMark) {
            if (sphere_prim_hit<R, false>(s.spheres[tr.shade], r.o, r.d, &t, &u)) { r.tmax = t; hit = (int)(nd.offset + i); *hu = u; *hv = R(0); }   // u carries the root branch
            continue;
          }
          if (tr.plane == skip_plane) continue;
          if (is_inst_tri(tr)) {   // object-space test, object-space t copied to the world ray (Q15)
            const RayCtx<R> ro = inst_ray(s.insts[inst_of(tr)], r);
            if (tri_closest(tr, ro, &t, &u, &v)) { r.tmax = t; hit = (int)(nd.offset + i); *hu = u; *hv = v; }
            continue;
          }
          if (tri_closest(tr, r, &t, &u, &v)) { r.tmax = t; hit = (int)(nd.offset + i); *hu = u; *hv = v; }
        }
        if (to_visit == 0) break;
        cur = st.get(--to_visit);
      } else {
        if (r.neg[nd.meta & 3]) { st.put(to_visit++, cur + 1); cur = nd.offset; }
        else { st.put(to_visit++, nd.offset); cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = st.get(--to_visit);
    }
  }
  *nn = cn; *np = cp;
  return hit;
}
template <typename R, typename Stack>
RRT_DEV bool traverse_any(const SceneDev<R>& s, const RayCtx<R>& r, Stack& st, int skip, uint32_t* nn, uint32_t* np) {
  uint32_t to_visit = 0, cur = 0, cn = 0, cp = 0;
  bool found = false;
  const uint32_t skip_plane = skip_plane_of(s, skip);
  if (s.n_nodes == 0) return false;
  while (true) {
    const Node<R> nd = s.nodes[cur];
    cn++;
    if (box_hit(nd, r)) {
      const uint32_t nprims = nd.meta >> 2;
      if (nprims > 0) {
        for (uint32_t i = 0; i < nprims; i++) {
          cp++;
          const Tri<R> tr = s.tris[nd.offset + i];
          if (tr.plane == kSphereMark) {
            R t, br;
            if (sphere_prim_hit<R, true>(s.spheres[tr.shade], r.o, r.d, &t, &br)) { found = true; break; }
            continue;
          }
          if (tr.plane == skip_plane) continue;
          if (is_inst_tri(tr)) {
            if (tri_any(tr, inst_ray(s.insts[inst_of(tr)], r))) { found = true; break; }
            continue;
          }
          if (tri_any(tr, r)) { found = true; break; }
        }
        if (found) break;
        if (to_visit == 0) break;
        cur = st.get(--to_visit);
      } else {
        if (r.neg[nd.meta & 3]) { st.put(to_visit++, cur + 1); cur = nd.offset; }
        else { st.put(to_visit++, nd.offset); cur = cur + 1; }
      }
    } else {
      if (to_visit == 0) break;
      cur = st.get(--to_visit);
    }
  }
  *nn = cn; *np = cp;
  return found;
}

// Closest-hit kernel over the rays of the active queue (records in queue order: `queue` itself is not read);
// `count` is read on device so the host never synchronises between bounces. Optional per-ray counters feed the roofline's byte model.
template <typename R, bool DEEP, bool COUNT>
__global__ void __launch_bounds__(kBlock) k_closest(SceneDev<R> s, Pools<R> p, const uint32_t* queue, const uint32_t* count, uint32_t n_fixed,
                                                     uint32_t* deep_stack, uint32_t deep_stride, uint32_t* nodes_out, uint32_t* prims_out,
                                                     unsigned long long* totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = count ? *count : n_fixed;
  if (i >= n) return;
  const typename Vec4T<R>::type ro = p.ray_o[i], rd = p.ray_d[i];
  R r_tmax; V3<R> r_lo;
  ray_tail(ro, Const<R>::inf, &r_tmax, &r_lo);
  RayCtx<R> r = make_ctx(V3<R>(ro.x, ro.y, ro.z), V3<R>(rd.x, rd.y, rd.z), r_tmax, r_lo);
  R hu = 0, hv = 0;
  uint32_t nn = 0, np = 0;
  int hit;
  const int skip = (int)real_to_bits(rd.w);
  if (DEEP) { GlobStack st{deep_stack + i, deep_stride}; hit = traverse_closest(s, r, st, skip, &hu, &hv, &nn, &np); }
  else { PrivStack st; hit = traverse_closest(s, r, st, skip, &hu, &hv, &nn, &np); }
  p.hit[i] = mk4<R>(r.tmax, bits_to_real((uint32_t)hit, R(0)), hu, hv);
  if (COUNT) {
    if (nodes_out) { nodes_out[i] = nn; prims_out[i] = np; }
    if (totals) { atomicAdd(&totals[0], (unsigned long long)nn); atomicAdd(&totals[1], (unsigned long long)np); }
  }
}

// unoccluded shadow ray: L += Ld (a path owns at most one shadow ray per launch, so the update needs no atomic)
template <typename R> RRT_DEV void add_pending(const Pools<R>& p, uint32_t i) {
  const typename Vec4T<R>::type ld = p.sld[i];
  const uint32_t slot = real_to_bits(ld.w);
  typename Vec4T<R>::type L = p.L[slot];
  L.x += ld.x; L.y += ld.y; L.z += ld.z;
  p.L[slot] = L;
}

// Any-hit kernel over the shadow queue: VisibilityTester::unoccluded (lights/mod.rs:60-66); unoccluded paths
// add their pending contribution to L (estimate_direct integrator/mod.rs:459-481).
template <typename R, bool DEEP, bool COUNT>
__global__ void __launch_bounds__(kBlock) k_shadow(SceneDev<R> s, Pools<R> p, const uint32_t* queue, const uint32_t* count,
                                                    uint32_t* deep_stack, uint32_t deep_stride, unsigned long long* totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *count) return;
  const typename Vec4T<R>::type ro = p.sray_o[i], rd = p.sray_d[i];
  R r_tmax; V3<R> r_lo;
  ray_tail(ro, R(1) - R(0.0001), &r_tmax, &r_lo);
  RayCtx<R> r = make_ctx(V3<R>(ro.x, ro.y, ro.z), V3<R>(rd.x, rd.y, rd.z), r_tmax, r_lo);
  uint32_t nn = 0, np = 0;
  bool occ;
  const int skip = (int)real_to_bits(rd.w);
  if (DEEP) { GlobStack st{deep_stack + i, deep_stride}; occ = traverse_any(s, r, st, skip, &nn, &np); }
  else { PrivStack st; occ = traverse_any(s, r, st, skip, &nn, &np); }
  if (!occ) add_pending(p, i);
  if (COUNT && totals) { atomicAdd(&totals[0], (unsigned long long)nn); atomicAdd(&totals[1], (unsigned long long)np); }
}

// Public batches (rrt_rays / rrt_hits are SoA arrays): pack into / unpack from the pool's records.
template <typename R>
__global__ void __launch_bounds__(kBlock) k_pack_rays(Pools<R> p, const R* ox, const R* oy, const R* oz, const R* dx, const R* dy, const R* dz, const R* tmax,
                                                       const int32_t* skip, uint32_t n, uint32_t n_tris) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // caller data: an index outside the triangle array means "none" (the kernels index tris[] with it)
  const int32_t sk = (skip && (uint32_t)skip[i] < n_tris) ? skip[i] : -1;
  store_ray<R>(p.ray_o, p.ray_d, i, V3<R>(ox[i], oy[i], oz[i]), V3<R>(), V3<R>(dx[i], dy[i], dz[i]), tmax[i], sk);
}
template <typename R>
__global__ void __launch_bounds__(kBlock) k_unpack_hits(Pools<R> p, const Tri<R>* tris, R* t, int32_t* prim, R* u, R* v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename Vec4T<R>::type h = p.hit[i];
  const int32_t pr = (int32_t)real_to_bits(h.y);
  t[i] = h.x; prim[i] = pr;
  if (u) u[i] = (pr >= 0 && tris[pr].plane == kSphereMark) ? R(0) : h.z;   // sphere hits keep their root branch there internally
  if (v) v[i] = h.w;
}

// Public any-hit on caller rays (rrt_trace_any): rays live in the closest-ray arrays of the pool.
template <typename R, bool DEEP>
__global__ void __launch_bounds__(kBlock) k_any_public(SceneDev<R> s, Pools<R> p, uint32_t n, uint8_t* occluded, uint32_t* deep_stack, uint32_t deep_stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const typename Vec4T<R>::type ro = p.ray_o[i], rd = p.ray_d[i];
  R r_tmax; V3<R> r_lo;
  ray_tail(ro, Const<R>::inf, &r_tmax, &r_lo);
  RayCtx<R> r = make_ctx(V3<R>(ro.x, ro.y, ro.z), V3<R>(rd.x, rd.y, rd.z), r_tmax, r_lo);
  uint32_t nn, np;
  bool occ;
  const int skip = (int)real_to_bits(rd.w);
  if (DEEP) { GlobStack st{deep_stack + i, deep_stride}; occ = traverse_any(s, r, st, skip, &nn, &np); }
  else { PrivStack st; occ = traverse_any(s, r, st, skip, &nn, &np); }
  occluded[i] = occ ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------
// Camera ray generation: ISampler::get_camerasample (samplers/mod.rs:28-34) + generate_ray_differential.
// Slot layout of a pass: slot = sample_local * npix + pixel_local, so the film kernel can sum a pixel's samples
// in sample order without atomics.
// ------------------------------------------------------------------------------------------------------------
struct PassDesc {
  int32_t rx0, ry0, rw;        // rect origin / width the pixel group is enumerated in
  uint32_t pix_begin, npix;    // pixel group [pix_begin, pix_begin + npix) in rect-linear order
  uint32_t s_begin, ns;        // sample_num range [s_begin, s_begin + ns)
  // interleaved row bands (multi-GPU film partition): local row r -> y = ry0 + ((r / band_h) * n_ranks + rank) * band_h + r % band_h
  uint32_t band_h, n_ranks, rank;
  // tiled = 1: the (local row, x) grid is enumerated tile by tile (kTileW x kTileH = 64 pixels; a workgroup of the dense camera kernel takes
  // one tile x 8 consecutive samples), row-major inside a tile and over the tiles, instead of row by row - the host sets it when the width is
  // a multiple of kTileW and the number of local rows a multiple of kTileH. Neighbouring queue entries are then neighbours in BOTH image
  // directions: the rays a wave holds cross a compact patch of the scene instead of a strip of an image row. Measured (frame, config 4):
  // rows 36.8 ms; 32 x 16 tiles, one sample per workgroup 35.7; 8 x 8 tiles x 8 samples 35.0 (16 x 8 x 4, 16 x 16 x 2: in between).
  uint32_t tiled;
};
#ifndef RRT_TILE_W
#define RRT_TILE_W 8u
#define RRT_TILE_H 8u
#endif
constexpr uint32_t kTileW = RRT_TILE_W, kTileH = RRT_TILE_H;
RRT_DEV void pass_pixel(const PassDesc& pd, uint32_t lin, uint32_t* px, uint32_t* py) {
  uint32_t row = lin / (uint32_t)pd.rw, col = lin % (uint32_t)pd.rw;
  if (pd.tiled) {
    const uint32_t tile = lin / (kTileW * kTileH), within = lin % (kTileW * kTileH), tiles_per_row = (uint32_t)pd.rw / kTileW;
    col = (tile % tiles_per_row) * kTileW + within % kTileW;
    row = (tile / tiles_per_row) * kTileH + within / kTileW;
  }
  *px = (uint32_t)pd.rx0 + col;
  *py = (uint32_t)pd.ry0 + ((row / pd.band_h) * pd.n_ranks + pd.rank) * pd.band_h + row % pd.band_h;
}

// Stage 1 (every slot): Halton index + dims 0..3, film point, lens sample, and the *main* lens trace
// (generate_ray, camera.rs:534-580). About 70 % of the samples die here (lens samples are drawn in [0.5,1.5)^2,
// Q5); the survivors are compacted into q_next so that stage 2 runs with full lanes. The main ray waits in the
// next-ray arrays *by slot* until stage 2 knows whether the sample lives.
template <typename R>
__global__ void __launch_bounds__(kBlock) k_raygen(SceneDev<R> s, Pools<R> p, PassDesc pd, double* dbg_dims) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = pd.npix * pd.ns;
  bool alive = false;
  if (slot < total) {
    const uint32_t pl = slot % pd.npix, sl = slot / pd.npix;
    uint32_t px, py;
    pass_pixel(pd, pd.pix_begin + pl, &px, &py);
    const uint32_t sample_num = pd.s_begin + sl;
    uint32_t index;
    double d0, d1, d2, d3, d4 = 0.0;
    if (s.sampler_type == 1u) {   // StratifiedSampler: get_camerasample = 2D film, 2D lens, 1D time (samplers/mod.rs:28-34)
      index = ((py * (uint32_t)s.xres + px) << 10) | sample_num;
      uint32_t cd = 0;
      st_get_2d(s, index, &cd, &d0, &d1);
      st_get_2d(s, index, &cd, &d2, &d3);
      d4 = st_get_1d(s, index, &cd);
    } else {
      index = halton_pixel_offset(s, px, py) + sample_num * s.stride;
      d0 = halton_dim(s, index, 0); d1 = halton_dim(s, index, 1); d2 = halton_cam_dim(s, index, 0); d3 = halton_cam_dim(s, index, 1);
    }
    // dimension 4 (time) is drawn and unused by a static scene
    const R pfx = (R)px + to_real<R>(d0), pfy = (R)py + to_real<R>(d1);
    const R lx = to_real<R>(d2) + R(0.5), ly = to_real<R>(d3) + R(0.5);  // Q5
    RayT<R> ray;
    const R w = generate_ray(s, pfx, pfy, lx, ly, &ray);
    alive = w != R(0);
    p.hindex[slot] = index;
    p.weight[slot] = alive ? w : R(0);
    p.samp[slot] = mk4<R>(pfx, pfy, lx, ly);
    if (alive) {
      store_ray<R>(p.nray_o, p.nray_d, slot, ray.o, V3<R>(), ray.d, Const<R>::inf, -1);
    }
    if (dbg_dims) {
      double* dd = dbg_dims + 5 * (size_t)(pl * pd.ns + sl);   // [pixel][sample]
      dd[0] = d0; dd[1] = d1; dd[2] = d2; dd[3] = d3; dd[4] = s.sampler_type == 1u ? d4 : halton_dim(s, index, 4);
    }
  }
  __shared__ uint32_t push_lds[kBlock / 64 + 1];
  const uint32_t q = block_push(&p.counters[C_NEXT], alive, push_lds);
  if (alive) p.q_next[q] = QEnt{slot, 0u, 0u, 0u};
}

// Stage 2 (survivors): the auxiliary rays of generate_ray_differential (camera.rs:582-628) at p_film +- 0.05 px in
// x then y. A sample whose x or y pair both fail gets weight 0; on scenes with textured materials the auxiliary rays
// themselves are kept as the camera ray's differentials (p.rdx_* / p.rdy_*). Living samples
// enter q_active with their ray at the same position. `enqueue` = 0 for AOIntegrator, whose li returns 0 before
// drawing anything (ao.rs:62-64).
template <typename R>
__global__ void __launch_bounds__(kBlock) k_raygen_aux(SceneDev<R> s, Pools<R> p, int enqueue) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = p.counters[C_NEXT];
  bool alive = false;
  uint32_t slot = 0;
  if (i < n) {
    slot = p.q_next[i].slot;
    const typename Vec4T<R>::type cs = p.samp[slot];
    const R pfx = cs.x, pfy = cs.y, lx = cs.z, ly = cs.w;
    RayT<R> aux, auy;
    R epsx = R(0.05), epsy = R(0.05);
    R wtx = generate_ray(s, pfx + R(0.05), pfy, lx, ly, &aux);
    if (wtx == R(0)) { epsx = R(-0.05); wtx = generate_ray(s, pfx + R(-0.05), pfy, lx, ly, &aux); }
    R wty = R(0);
    if (wtx != R(0)) {
      wty = generate_ray(s, pfx, pfy + R(0.05), lx, ly, &auy);
      if (wty == R(0)) { epsy = R(-0.05); wty = generate_ray(s, pfx, pfy + R(-0.05), lx, ly, &auy); }
    }
    alive = wtx != R(0) && wty != R(0) && p.weight[slot] > R(0);   // `if ray_weight > 0.0` integrator/mod.rs:100
    if (!(wtx != R(0) && wty != R(0))) p.weight[slot] = R(0);
    if (alive && p.rdx_o) {
      // rx / ry of generate_ray_differential (camera.rs:597-598, 613-614), then scale_differentials (geometry.rs:1883-1888)
      const typename Vec4T<R>::type mo = p.nray_o[slot], md = p.nray_d[slot];
      const V3<R> o(mo.x, mo.y, mo.z), d(md.x, md.y, md.z);
      V3<R> rxo = o + (aux.o - o) / epsx, rxd = d + (aux.d - d) / epsx;
      V3<R> ryo = o + (auy.o - o) / epsy, ryd = d + (auy.d - d) / epsy;
      rxo = o + (rxo - o) * s.diff_scale; ryo = o + (ryo - o) * s.diff_scale;
      rxd = d + (rxd - d) * s.diff_scale; ryd = d + (ryd - d) * s.diff_scale;
      p.rdx_o[slot] = mk4<R>(rxo.x, rxo.y, rxo.z, R(0)); p.rdx_d[slot] = mk4<R>(rxd.x, rxd.y, rxd.z, R(0));
      p.rdy_o[slot] = mk4<R>(ryo.x, ryo.y, ryo.z, R(0)); p.rdy_d[slot] = mk4<R>(ryd.x, ryd.y, ryd.z, R(0));
    }
  }
  __shared__ uint32_t push_lds[kBlock / 64 + 1];
  const bool enq = alive && enqueue;
  const uint32_t q = block_push(&p.counters[C_ACTIVE], enq, push_lds);
  if (enq) {
    p.q_active[q] = QEnt{slot, s.cam_db, p.hindex[slot], 0u};   // camera dimensions consumed, bounce 0
    p.path[q] = mk4<R>(R(1), R(1), R(1), R(1));
    p.ray_o[q] = p.nray_o[slot]; p.ray_d[q] = p.nray_d[slot];
  }
  if (alive) p.L[slot] = mk4<R>(R(0), R(0), R(0), R(0));
  (void)block_push(&p.counters[C_CAMERA_RAYS], alive, push_lds);
}

// debug / unit-test surface of rrt_camera_samples: rays + weights of a pass, [pixel][sample] order (dbg_ray zeroed
// by the host: dead samples report a zero ray)
template <typename R>
__global__ void __launch_bounds__(kBlock) k_camera_dump(Pools<R> p, PassDesc pd, double* dbg_ray, double* dbg_w) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < pd.npix * pd.ns) {
    const uint32_t pl = t % pd.npix, sl = t / pd.npix;
    dbg_w[pl * pd.ns + sl] = (double)p.weight[t];
  }
  if (t < p.counters[C_ACTIVE]) {
    const uint32_t slot = p.q_active[t].slot;
    const uint32_t pl = slot % pd.npix, sl = slot / pd.npix;
    const typename Vec4T<R>::type ro = p.ray_o[t], rd = p.ray_d[t];
    double* rr = dbg_ray + 6 * (size_t)(pl * pd.ns + sl);
    rr[0] = (double)ro.x; rr[1] = (double)ro.y; rr[2] = (double)ro.z;
    rr[3] = (double)rd.x; rr[4] = (double)rd.y; rr[5] = (double)rd.z;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Shading
// ------------------------------------------------------------------------------------------------------------
template <typename R>
struct Surf {   // the parts of SurfaceInteraction (interaction.rs:95-181) the in-scope materials read
  V3<R> p, p_lo, n, wo, sn, sdpdu;   // p_lo: see spawn_point()
  uint32_t material;
  bool ok;
};

// Hit point. The reference uses o + d * t in f64 and then relies on its `t < 1e-7` rejection to swallow the ~1e-15
// by which that point misses the surface. In fp32 the same expression misses by ~3e-6 at scene coordinates of ~50,
// more than the threshold: a spawned ray that starts just below the plane of the *adjacent* triangle (origin within
// ~1e-4 of a shared edge) then "hits" it at t ~ 1e-5 - about 1e-4 of all shadow rays on the 100k-triangle config,
// 3-4 % of its pixels off by a whole sample. So the fp32 mode builds the point from the barycentrics on the stored
// (fp32) triangle as an unevaluated sum p_hi + p_lo (TwoSum), and the triangle tests form (o_hi - p0) + o_lo:
// o_hi - p0 is exact for nearby vertices, so the origin lies on its triangle to ~1e-9 and on the right side of every
// neighbour. f64 mode: the reference expression, p_lo = 0.
template <typename R>
RRT_DEV void spawn_point(V3<R> o, V3<R> d, R t, R u, R v, V3<R> p0, V3<R> p1, V3<R> p2, V3<R>* p, V3<R>* p_lo) {
#ifdef RRT_SPAWN_OLD
  { *p = o + d * t; *p_lo = V3<R>(); return; }
#endif
  if (sizeof(R) == 8) { *p = o + d * t; *p_lo = V3<R>(); return; }
  const V3<R> w = (p1 - p0) * u + (p2 - p0) * v;
  const V3<R> s = p0 + w;
  const V3<R> bb = s - p0;
  *p = s;
  *p_lo = (p0 - (s - bb)) + (w - bb);
}

// Triangle::intersect's SurfaceInteraction (shape/triangle.rs:267-390) rebuilt from (triangle, t, u, v)
template <typename R>
RRT_DEV Surf<R> build_surface(const SceneDev<R>& s, int prim, V3<R> o, V3<R> d, R t, R u, R v, SurfExt<R>* ext = nullptr) {
  Surf<R> si;
  const Tri<R> tr = s.tris[prim];
  if (tr.plane == kSphereMark) {   // u = root branch recorded by the traversal (public API reports 0, 0 for spheres)
    SphereSI<R> ss;
    sphere_surface(s.spheres[tr.shade], o, d, t, u, &ss, ext);
    si.ok = true;
    si.p = ss.p; si.p_lo = V3<R>(); si.n = ss.n; si.wo = ss.wo; si.sn = ss.sn; si.sdpdu = ss.sdpdu;
    si.material = tr.material;
    if (!(dot(si.n, si.sn) >= R(0))) si.ok = false;   // primitives.rs:66
    return si;
  }
  const bool inst = is_inst_tri(tr);
  if (inst) {   // the triangle works with the ray TransformedPrimitive::intersect handed it (primitives.rs:124-127)
    const InstDev<R>& I = s.insts[inst_of(tr)];
    o = aff_pt(I.mi, o); d = vnormalize(vnormalize(aff_vec(I.mi, d)));
  }
  V3<R> p0(tr.p0), p1(tr.p1), p2(tr.p2);
  R uv[3][2] = {{R(0), R(0)}, {R(1), R(0)}, {R(1), R(1)}};  // get_uvs :113-128
  uint32_t has_n = 0;
  V3<R> vn0, vn1, vn2;
  if (tr.shade != 0xffffffffu) {
    const TriShade<R>& sh = s.shades[tr.shade];
    if (sh.has_uv) for (int k = 0; k < 3; k++) { uv[k][0] = sh.uv[k][0]; uv[k][1] = sh.uv[k][1]; }
    has_n = sh.has_n;
    if (has_n == 1) { vn0 = V3<R>(sh.n[0]); vn1 = V3<R>(sh.n[1]); vn2 = V3<R>(sh.n[2]); }
  }
  R duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]}, duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
  V3<R> dp02 = p0 - p2, dp12 = p1 - p2;
  R determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
  bool degenerate_uv = rabs(determinant) < R(1e-8);
  V3<R> dpdu, dpdv;
  if (!degenerate_uv) {
    R i_det = R(1) / determinant;
    dpdu = (dp02 * duv12[1] - dp12 * duv02[1]) * i_det;
    dpdv = (dp02 * -duv12[0] + dp12 * duv02[0]) * i_det;
  }
  if (degenerate_uv || len2(cross(dpdu, dpdv)) == R(0)) {
    V3<R> ng = cross(p2 - p0, p1 - p0);
    coordinate_system(vnormalize(ng), &dpdu, &dpdv);
  }
  spawn_point(o, d, t, u, v, p0, p1, p2, &si.p, &si.p_lo);
  si.wo = -d;
  si.n = vnormalize(cross(dp02, dp12));
  si.sn = si.n;
  si.sdpdu = dpdu;
  si.material = inst ? (tr.material & 0xffffu) : tr.material;
  si.ok = true;
  if (ext) {
    ext->u = uv[0][0] * (R(1) - u - v) + uv[1][0] * u + uv[2][0] * v;
    ext->v = uv[0][1] * (R(1) - u - v) + uv[1][1] * u + uv[2][1] * v;
    ext->dpdu = dpdu; ext->dpdv = dpdv; ext->sdpdv = dpdv;
    ext->sdndu = ext->sdndv = V3<R>();
    if (has_n == 1) {   // triangle.rs:351-386
      const V3<R> dn1 = vn0 - vn2, dn2 = vn1 - vn2;
      if (degenerate_uv) {
        const V3<R> dn = cross(vn2 - vn0, vn1 - vn0);
        if (len2(dn) != R(0)) coordinate_system(dn, &ext->sdndu, &ext->sdndv);
      } else {
        const R i_det = R(1) / determinant;
        ext->sdndu = (dn1 * duv12[1] - dn2 * duv02[1]) * i_det;
        ext->sdndv = (dn1 * -duv12[0] + dn2 * duv02[0]) * i_det;
      }
    }
  }
  if (has_n == 1) {
    V3<R> ns = vn0 * (R(1) - u - v) + vn1 * u + vn2 * v;
    if (len2(ns) > R(0)) ns = nnormalize(ns); else ns = si.n;
    V3<R> ss = vnormalize(dpdu);
    V3<R> ts = cross(ss, ns);
    if (len2(ts) > R(0)) { ts = vnormalize(ts); ss = cross(ts, ns); }
    else coordinate_system(ns, &ss, &ts);
    // set_shading_geometry(.., orientation_is_authoritative = true) interaction.rs:183-202: the *geometric*
    // normal is flipped towards the interpolated frame and stored as the shading normal (Q14) ...
    V3<R> nn = nnormalize(cross(ss, ts));
    si.sn = faceforward(si.n, nn);
    si.sdpdu = ss;
    if (ext) ext->sdpdv = ts;
    // ... and primitives.rs:66 asserts dot(ist.n, shading.n) >= 0, i.e. panics when vn opposes the winding.
    if (!(dot(si.n, si.sn) >= R(0))) si.ok = false;
  }
  if (inst) {   // `*si = primitive_to_world.t(si)` primitives.rs:131-136 + SurfaceInteraction::t_by transform.rs:628-655
    const InstDev<R>& I = s.insts[inst_of(tr)];
    if (!I.identity) {
      si.p = aff_pt(I.m, si.p + si.p_lo); si.p_lo = V3<R>();
      si.wo = aff_vec(I.m, si.wo);                       // (not re-normalised: its length is the instance's scale along the ray)
      const V3<R> n2 = aff_nrm(I.mi, si.n);              // BaseInteraction::t_by: not re-normalised either
      si.sn = faceforward(nnormalize(aff_nrm(I.mi, si.sn)), n2);
      si.n = n2;
      si.sdpdu = aff_vec(I.m, si.sdpdu);
      if (ext) {
        ext->dpdu = aff_vec(I.m, ext->dpdu); ext->dpdv = aff_vec(I.m, ext->dpdv); ext->sdpdv = aff_vec(I.m, ext->sdpdv);
        ext->sdndu = aff_nrm(I.mi, ext->sdndu); ext->sdndv = aff_nrm(I.mi, ext->sdndv);
      }
    }
  }
  return si;
}

template <typename R, int NL, uint32_t KM>
RRT_DEV bool build_bsdf(const SceneDev<R>& s, const Surf<R>& si, Bsdf<R, NL, KM>* b, bool allow_multiple_lobes = true) {  // Bsdf::new reflection.rs:215-226
  b->ns = si.sn;
  b->ss = vnormalize(si.sdpdu);
  b->ng = si.n;
  b->ts = cross(b->ns, b->ss);
  return build_lobes(s.materials[si.material], b, allow_multiple_lobes);
}
// textured scenes: compute_differentials (interaction.rs:209), then the material's textures at this hit, then the lobes.
// Returns false for a Glass / Translucent whose evaluated colours are all black: the reference leaves `bsdf` None there
// and path.rs:103 underflows `bounces` (constant materials are checked on the host).
template <typename R, int NL, uint32_t KM>
RRT_DEV bool build_bsdf_tex(const SceneDev<R>& s, Surf<R>& si, const SurfExt<R>& ext, const DiffRay<R>& rd, TexCtx<R>* c,
                            Bsdf<R, NL, KM>* b, bool allow_multiple_lobes = true) {
  c->p = si.p; c->u = ext.u; c->v = ext.v;
  c->pd = V3<double>((double)si.p.x + (double)si.p_lo.x, (double)si.p.y + (double)si.p_lo.y, (double)si.p.z + (double)si.p_lo.z);
  compute_differentials(c, si.n, ext.dpdu, ext.dpdv, rd);
  const int bump = s.materials[si.material].bump;
  if (bump >= 0) {   // Material::bump material/mod.rs:22-62: the displacement texture at (u + du, v), (u, v + dv) and (u, v)
    TexCtx<R> ev = *c;
    R du = rabs(c->dudx) * R(0.5) + rabs(c->dudy);   // (as written at :26)
    if (du == R(0)) du = R(0.0005);
    auto step_d = [&](V3<R> dir, R h) { return V3<double>(c->pd.x + (double)dir.x * (double)h, c->pd.y + (double)dir.y * (double)h, c->pd.z + (double)dir.z * (double)h); };
    ev.p = si.p + si.sdpdu * du; ev.pd = step_d(si.sdpdu, du); ev.u = c->u + du; ev.v = c->v;
    const R u_displace = TexEval<R, kTexDepth>::eval(s, bump, ev).r;
    R dv = (rabs(c->dvdx) + rabs(c->dvdy)) * R(0.5);
    if (dv == R(0)) dv = R(0.0005);
    ev.p = si.p + ext.sdpdv * dv; ev.pd = step_d(ext.sdpdv, dv); ev.u = c->u; ev.v = c->v + dv;
    const R v_displace = TexEval<R, kTexDepth>::eval(s, bump, ev).r;
    const R displace = TexEval<R, kTexDepth>::eval(s, bump, *c).r;
    c->err |= ev.err;
    const V3<R> dpdu = si.sdpdu + si.sn * (u_displace - displace) / du + ext.sdndu * displace;
    const V3<R> dpdv = ext.sdpdv + si.sn * (v_displace - displace) / dv + ext.sdndv * displace;
    // set_shading_geometry(.., orientation_is_authoritative = false) interaction.rs:183-202
    si.sn = faceforward(nnormalize(cross(dpdu, dpdv)), si.n);
    si.sdpdu = dpdu;
  }
  b->ns = si.sn;
  b->ss = vnormalize(si.sdpdu);
  b->ng = si.n;
  b->ts = cross(b->ns, b->ss);
  const Material<R> m = resolve_material(s, s.materials[si.material], *c);
  build_lobes(m, b, allow_multiple_lobes);
  if (c->err) return false;
  if (m.type == 5 && rgb_clamp0(Rgb<R>(m.kr)).is_black() && rgb_clamp0(Rgb<R>(m.kt)).is_black()) return false;
  if (m.type == 6 && rgb_clamp0(Rgb<R>(m.reflect)).is_black() && rgb_clamp0(Rgb<R>(m.transmit)).is_black()) return false;
  return true;
}

// Light::sample_li for PointLight (point.rs:55-77) and DiffuseAreaLight (diffuse.rs:63-79) over
// Shape::sample_ref (shape/mod.rs:33-48), Sphere::sample (sphere.rs:265-285), Triangle::sample (triangle.rs:393-418)
template <typename R, bool AREA = true>
RRT_DEV Rgb<R> light_sample_li(const Light<R>& L, V3<R> ref_p, R u0, R u1, V3<R>* wi, R* pdf, V3<R>* p1, V3<R>* n1) {
  if (!AREA && L.type != 0 && L.type != 2) __builtin_unreachable();   // (instantiation for scenes whose lights are all point / distant: the host checks, rrt_impl.hpp)
  if (L.type == 0) {
    V3<R> pl(L.p_light);
    *wi = vnormalize(pl - ref_p);
    *pdf = R(1);
    *p1 = pl; *n1 = V3<R>();
    return Rgb<R>(L.spectrum) / len2(pl - ref_p);
  }
  if (L.type == 2) {   // DistantLight::sample_li distant.rs:67-92
    const V3<R> w(L.w_light);
    *wi = w;
    *pdf = R(1);
    *p1 = ref_p + w * (R(2) * L.world_radius); *n1 = V3<R>();
    return Rgb<R>(L.spectrum);
  }
  V3<R> p, n;
  if (L.shape_type == 1) {
    V3<R> p_obj = uniform_sample_sphere(u0, u1) * L.radius;
    n = nnormalize(aff_nrm(L.mi, p_obj));
    p_obj = p_obj * (L.radius / len(p_obj));
    p = aff_pt(L.m, p_obj);
  } else {
    V3<R> b = uniform_sample_sphere(u0, u1);  // used as barycentrics (Q19)
    V3<R> q0(L.tp[0]), q1(L.tp[1]), q2(L.tp[2]);
    p = q0 * b.x + q1 * b.y + q2 * b.z;
    n = vnormalize(cross(q1 - q0, q2 - q0));
    if (L.tri_has_n) {
      V3<R> ns = V3<R>(L.tn[0]) * b.x + V3<R>(L.tn[1]) * b.y + V3<R>(L.tn[2]) * b.z;
      n = faceforward(n, ns);
    }
  }
  V3<R> w = p - ref_p;
  R wl2 = len2(w), pd;
  if (wl2 == R(0)) pd = R(0);
  else {
    w = vnormalize(w);
    pd = wl2 / absdot(-w, n);
    if (isinf(pd)) pd = R(0);
  }
  *pdf = pd;
  if (pd == R(0) || len2(p - ref_p) == R(0)) { *pdf = R(0); return Rgb<R>(); }
  *wi = vnormalize(p - ref_p);
  *p1 = p; *n1 = n;
  return (dot(n, -*wi) > R(0)) ? Rgb<R>(L.spectrum) : Rgb<R>();  // DiffuseAreaLight::l diffuse.rs:133-141
}

// Sphere::intersect (sphere.rs:124-259) reduced to what Shape::pdf_ref (shape/mod.rs:49-66) reads: hit point
// and ist.n. Only needed for the MIS weight of area lights.
template <typename R>
RRT_DEV bool sphere_hit_for_pdf(const Light<R>& L, V3<R> ro, V3<R> rd, V3<R>* p_world, V3<R>* n_world) {
  V3<R> oo = aff_pt(L.mi, ro), od = vnormalize(vnormalize(aff_vec(L.mi, rd)));
  R a = od.x * od.x + od.y * od.y + od.z * od.z;
  R b = R(2) * (od.x * oo.x + od.y * oo.y + od.z * oo.z);
  R c = oo.x * oo.x + oo.y * oo.y + oo.z * oo.z - L.radius * L.radius;
  R t0, t1;
  if (!quadratic(a, b, c, &t0, &t1)) return false;
  if (t0 > R(1999999999.0) || t1 <= R(0)) return false;
  R th = t0;
  if (t0 <= R(0)) { th = t1; if (th > R(1999999999.0)) return false; }
  V3<R> ph = ro + rd * th;  // world ray (Q16)
  if (ph.x == R(0) && ph.y == R(0)) ph.x = R(1e-5) * L.radius;
  R phi = atan2(ph.y, ph.x);
  if (phi < R(0)) phi += R(2) * R(RRT_PI);
  if ((L.z_min > -L.radius && ph.z < L.z_min) || (L.z_max < L.radius && ph.z > L.z_max) || (phi > L.phi_max)) {
    if (th == t1) return false;
    if (t1 > R(1999999999.0)) return false;
    th = t1;
    ph = oo + od * th;
    ph = ph * (L.radius / len(ph));
    if (ph.x == R(0) && ph.y == R(0)) ph.x = R(1e-5) * L.radius;
    phi = atan2(ph.y, ph.x);
    if (phi < R(0)) phi += R(2) * R(RRT_PI);
    if ((L.z_min > -L.radius && ph.z < L.z_min) || (L.z_max < L.radius && ph.z > L.z_max) || (phi > L.phi_max)) return false;
  }
  R theta = acos(clampr(ph.z / L.radius, R(-1), R(1)));
  R z_radius = sqrt(ph.x * ph.x + ph.y * ph.y);
  R inv_zr = R(1) / z_radius;
  R cphi = ph.x * inv_zr, sphi = ph.y * inv_zr;
  V3<R> dpdu(-L.phi_max * ph.y, L.phi_max * ph.x, R(0));
  V3<R> dpdv = V3<R>(ph.z * cphi, ph.z * sphi, -L.radius * R(sin(theta))) * (L.theta_max - L.theta_min);
  V3<R> n = vnormalize(cross(dpdu, dpdv));
  *p_world = aff_pt(L.m, ph);
  *n_world = aff_nrm(L.mi, n);
  return true;
}

// estimate_direct's light-sampling half (integrator/mod.rs:403-481), handle_media = false, specular = false.
// Returns true and fills the shadow ray + contribution when a visibility test is needed.
// The BSDF-sampling half (:483-556) can only add `li * f * weight / pdf` with li = 0 (no primitive carries an
// area light, Q18; DiffuseAreaLight::le is the trait default 0), so it is not executed: see DESIGN.md.
template <bool AREA = true, typename R, int NL, uint32_t KM>
RRT_DEV bool estimate_direct_light(const Surf<R>& si, const Bsdf<R, NL, KM>& bsdf, const Light<R>& L, R ul0, R ul1, V3<R>* so, V3<R>* sd, Rgb<R>* ld) {
  const uint32_t flags = BXDF_ALL & ~BXDF_SPECULAR;
  V3<R> wi, p1, n1;
  R light_pdf = R(0);
  Rgb<R> li = light_sample_li<R, AREA>(L, si.p, ul0, ul1, &wi, &light_pdf, &p1, &n1);
  if (!(light_pdf > R(0)) || li.is_black()) return false;
  Rgb<R> f = bsdf.f(si.wo, wi, flags) * absdot(wi, si.sn);
  R scattering_pdf = bsdf.pdf(si.wo, wi, flags);
  if (f.is_black()) return false;
  // spawn_ray_to_si interaction.rs:66-77 with p_error = 0 (Q8): origin = p, target = p1, Ray::new normalises d
  // and keeps t_max = 1 - SHADOW_EPSILON (Q9)
  *so = si.p;
  *sd = vnormalize(p1 - si.p);
  if (!AREA || L.type != 1) *ld = f * li / light_pdf;   // delta lights (point, distant): no MIS
  else { R w = power_heuristic1(light_pdf, scattering_pdf); *ld = li * f * w / light_pdf; }
  return true;
}

// Distribution1D::sample_discrete sampling.rs:93-123 on the uniform light distribution
template <typename R>
RRT_DEV uint32_t sample_light_discrete(const SceneDev<R>& s, R u) {
  uint32_t first = 0, len = s.n_lights + 1;
  while (len > 0) {
    uint32_t half = len >> 1, middle = first + half;
    if (s.light_cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
  }
  uint32_t off = first - 1;
  if (off > s.n_lights - 1) off = s.n_lights - 1;
  return off;
}

// PathIntegrator::li loop body (path.rs:74-223) for one bounce of every active path.
// occupancy target of the path shading kernel: asking for 4 waves per SIMD lets the register allocator use the full
// 128-VGPR budget of that occupancy (measured: 8.9 ms against 9.9 ms without the hint; 5 or 6 waves spill: 10.6 / 13.2 ms)
#ifndef RRT_SHADE_CHUNK
#define RRT_SHADE_CHUNK 4u   // queue entries per workgroup and trip of the hit compaction, in workgroup sizes (measured: 2 / 4 / 8 - see DESIGN.md)
#endif
#ifndef RRT_SHADE_WAVES
#define RRT_SHADE_WAVES 4
#endif
// The Lambert-only instantiation needs 96 VGPRs unforced; measured (shading alone / frame with two in flight, config 4): 4 waves per SIMD
// 5.87 ms, 5: 5.58 / 37.8, 6: 5.43 / 37.7, 7: 5.35 / 37.6, 8: 5.64; with 512-thread blocks 5: 5.61 / 38.0, 6: 4.92 / 37.4, 7: 5.01 / 37.3; 1024: 6.39.
#ifndef RRT_SHADE_WAVES_LAMBERT
#define RRT_SHADE_WAVES_LAMBERT 6
#endif
#ifndef RRT_SHADE_BLOCK_LAMBERT
#define RRT_SHADE_BLOCK_LAMBERT 512
#endif
#ifndef RRT_SHADE_WAVES_GLOSSY
#define RRT_SHADE_WAVES_GLOSSY RRT_SHADE_WAVES
#endif
#ifndef RRT_SHADE_BLOCK_GLOSSY
#define RRT_SHADE_BLOCK_GLOSSY RRT_SHADE_BLOCK
#endif
template <typename R, uint32_t KM> constexpr int shade_path_block() {
  return sizeof(R) != 4 ? ShadeBlock<R>::n : (KM == kKindsLambert ? RRT_SHADE_BLOCK_LAMBERT : (KM == kKindsGlossy ? RRT_SHADE_BLOCK_GLOSSY : ShadeBlock<R>::n));
}
template <uint32_t KM> constexpr int shade_path_waves() { return KM == kKindsLambert ? RRT_SHADE_WAVES_LAMBERT : (KM == kKindsGlossy ? RRT_SHADE_WAVES_GLOSSY : RRT_SHADE_WAVES); }
// TEX: the scene has materials that evaluate a texture per hit (SurfExt + ray differentials of the camera ray at bounce 0;
// `ray = isect.spawn_ray(wi).into()` drops them afterwards, path.rs:163).
// KM: the lobe kinds the scene's materials can produce (dmath.hpp "Lobe-kind sets"); kAllKinds = the general kernel.
// AREA = false: every light of the scene is a point or distant light (no area-light sampling code, no light sample dimensions drawn).
template <typename R, int NL, bool TEX = false, uint32_t KM = kAllKinds, bool AREA = true>
__global__ void __launch_bounds__((shade_path_block<R, KM>())) __attribute__((amdgpu_waves_per_eu(TEX ? 1 : shade_path_waves<KM>(), 8))) k_shade_path(SceneDev<R> s, Pools<R> p) {
  __shared__ uint32_t push_lds[shade_path_block<R, KM>() / 64 + 1];
  // [r4] Hit compaction (option "shade_compact"). More than half of a queue's rays MISS on an open scene (config 4: 104 M queued closest-hit rays, ~47 M hits per
  // frame), and the path integrator shades a miss with nothing (Q18) - with one thread per queue entry half of every wave sat idle through the whole kernel. A
  // workgroup now takes kShadeChunk x its size consecutive entries, reads their hit words, packs the indices of the hits into LDS in queue order (block_rank) and
  // shades those with full waves. The next / shadow queues receive the same entries in the same order inside a workgroup's chunk; every sample adds to its own slot:
  // frames are identical bit for bit (tests/test_gpu_parity.py::test_shade_compaction_changes_nothing).
  constexpr uint32_t kShadeChunk = RRT_SHADE_CHUNK;
  __shared__ uint32_t hit_idx[kShadeChunk * shade_path_block<R, KM>()];
  const uint32_t n = p.counters[C_ACTIVE];
  using V4 = typename Vec4T<R>::type;
  const uint32_t chunk = s.shade_compact ? kShadeChunk : 1u;
  // a bounded grid walks the queue (block-uniform trip counts, as block_push needs): the host does not know the queue
  // size, and a grid sized for the whole pass costs 0.2 ms of empty blocks per launch once the queues are short
  for (uint32_t base0 = blockIdx.x * blockDim.x * chunk; base0 < n; base0 += gridDim.x * blockDim.x * chunk) {
  uint32_t n_hits = min(blockDim.x, n - base0);   // (no compaction: the entries themselves)
  if (chunk > 1u) {
    n_hits = 0u;
    for (uint32_t k = 0; k < chunk; k++) {
      const uint32_t e = base0 + k * blockDim.x + threadIdx.x;
      const bool is_hit = e < n && (int)real_to_bits(p.hit[e].y) >= 0;
      uint32_t cnt = 0;
      const uint32_t at = block_rank(is_hit, push_lds, &cnt);
      if (is_hit) hit_idx[n_hits + at] = e;
      n_hits += cnt;
    }
    __syncthreads();
  }
  for (uint32_t base = 0; base < n_hits; base += blockDim.x) {
  const uint32_t i = base + threadIdx.x < n_hits ? (chunk > 1u ? hit_idx[base + threadIdx.x] : base0 + base + threadIdx.x) : n;   // n: no entry for this thread
  bool want_shadow = false, want_next = false, sky = false;
  uint32_t slot = 0;
  int prim = -1;
  V3<R> sh_o, sh_d, nx_o, nx_d, o_lo;   // shadow ray / next ray + path state, stored at their queue positions at the end
  Rgb<R> sh_ld, nx_beta;
  R nx_eta_scale = R(1);
  uint32_t index = 0, nx_db = 0, sh_tab = 0;
  if (i < n) {
    const QEnt qe = p.q_active[i];
    slot = qe.slot;
    const V4 h = p.hit[i];
    prim = (int)real_to_bits(h.y);
    const uint32_t db = qe.db;
    uint32_t dim = db_dim(s, db), bounces = db_bounce(s, db);
    // `if !found_intersection || bounces >= max_depth { break }` (:91); emitted light is 0 (Q18)
    if (prim >= 0 && (int)bounces < s.max_depth) {
      const V4 ro = p.ray_o[i], rd = p.ray_d[i];
      V3<R> o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
      SurfExt<R> ext;
      Surf<R> si = build_surface(s, prim, o, d, h.x, h.z, h.w, TEX ? &ext : nullptr);
      if (!si.ok) { atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_SHADING_NORMAL); }
      else {
        Bsdf<R, NL, KM> bsdf;
        if (TEX) {
          DiffRay<R> dr;
          if (bounces == 0) {
            const V4 a = p.rdx_o[slot], b = p.rdx_d[slot], c = p.rdy_o[slot], e = p.rdy_d[slot];
            dr.has = true;
            dr.rxo = V3<R>(a.x, a.y, a.z); dr.rxd = V3<R>(b.x, b.y, b.z); dr.ryo = V3<R>(c.x, c.y, c.z); dr.ryd = V3<R>(e.x, e.y, e.z);
          }
          TexCtx<R> tc;
          if (!build_bsdf_tex(s, si, ext, dr, &tc, &bsdf)) atomicOr(&p.counters[C_ERROR], (uint32_t)(tc.err ? ERR_MIPMAP : ERR_NULL_BSDF));
        } else if (!build_bsdf(s, si, &bsdf)) atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_KIND_SET);
        // a camera ray's path record is {1, 1, 1, 1} whoever produced it: not read (and, by the dense fp32 camera kernels under this integrator, not written)
        const V4 st_b = bounces == 0u ? mk4<R>(R(1), R(1), R(1), R(1)) : p.path[i];
        index = qe.index;
        Rgb<R> beta(st_b.x, st_b.y, st_b.z);
        R eta_scale = st_b.w;
        // uniform_sample_one_light integrator/mod.rs:359-401 with the uniform Distribution1D (path.rs:47-49)
        if (bsdf.num_components(BXDF_ALL & ~BXDF_SPECULAR) > 0 && s.n_lights > 0) {
          // Sampler values nobody reads are not computed, only counted (a radical inverse in a high dimension is ~6 digit
          // steps of f64 / u64 arithmetic, and these three were half of this kernel's draws): with one light the discrete
          // distribution returns light 0 for every u; point and distant lights ignore their 2D sample (point.rs:55-77,
          // distant.rs:67-92).
          double du0 = 0.0, du1 = 0.0;
          uint32_t ln = 0;
          if (s.n_lights == 1u) skip_1d(s, &dim);
          else ln = sample_light_discrete(s, to_real<R>(draw_1d(s, index, &dim)));
          if (!AREA || s.lights[ln].type != 1) skip_2d(s, &dim);
          else draw_2d(s, index, &dim, &du0, &du1);
          R ul0 = to_real<R>(du0), ul1 = to_real<R>(du1);
          skip_2d(s, &dim);  // u_scattering: drawn, only used by the BSDF-sampling half
          V3<R> so, sd;
          Rgb<R> ld;
          if (s.light_pick_pdf != R(0) && estimate_direct_light<AREA>(si, bsdf, s.lights[ln], ul0, ul1, &so, &sd, &ld)) {
            sh_ld = beta * (ld / s.light_pick_pdf);
            sh_o = so; sh_d = sd;
            sh_tab = s.use_shadow_tabs ? s.lights[ln].shadow_tab : 0u;
            want_shadow = true;
          }
        }
        o_lo = si.p_lo;
        // Sample BSDF to get new path direction (:125-148)
        double db0, db1;
        draw_2d(s, index, &dim, &db0, &db1);
        R u0 = to_real<R>(db0), u1 = to_real<R>(db1);
        V3<R> wi;
        R pdf = R(0);
        uint32_t flags = 0;
        // `let wo = -ray.d` (path.rs:126): the WORLD ray's direction, not the interaction's wo - which estimate_direct uses (integrator/mod.rs:441)
        // and which differs from it on a non-rigid instance (transformed back, not re-normalised: primitives.rs:131-136) and by rounding on a sphere
        const V3<R> wo_path = -d;
        Rgb<R> f = bsdf.sample_f(wo_path, &wi, u0, u1, &pdf, BXDF_ALL, &flags);
        if (!(f.is_black() || pdf == R(0))) {
          beta = beta * (f * absdot(wi, si.sn) / pdf);
          if (!(beta.y() > R(0)) || isinf(beta.y()) || beta.y() != beta.y()) atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_BETA);  // path.rs:146-147 asserts
          if ((flags & BXDF_SPECULAR) && (flags & BXDF_TRANSMISSION))   // path.rs:150-162
            eta_scale *= dot(wo_path, si.n) > R(0) ? bsdf.eta * bsdf.eta : R(1) / (bsdf.eta * bsdf.eta);
          V3<R> nd = vnormalize(wi);  // spawn_ray -> Ray::new_od (Q8: no origin offset)
          bool cont = true;
          // Russian roulette (:214-222) on beta * eta_scale
          const R rr_max = (beta * eta_scale).max_component();
          if (rr_max < s.rr_threshold && bounces > 3) {
            R q = rmax(R(1) - rr_max, R(0.05));
            R ur = to_real<R>(draw_1d(s, index, &dim));
            if (ur < q) cont = false;
            else beta = beta / (R(1) - q);
          }
          bounces += 1;
          // the next loop iteration would trace and then break on `bounces >= max_depth` without using the
          // hit (emission is 0): that dead closest-hit query is not issued.
          // Horizon cull (SceneDev::horizon, host/horizon_build.cpp): the next ray starts on triangle `prim`; if its elevation above / below the scene's
          // flattest axis exceeds everything the host found visible from ANY point of that triangle in the ray's azimuth sector, BVHAccel::intersect would
          // return false for it and the path would end at `if !found_intersection { break }` (path.rs:91) - it is counted as the closest-hit query it is, and not traced.
          if (cont && (int)bounces < s.max_depth && s.horizon != nullptr) {
            const float u = s.hz_axis == 0u ? (float)nd.x : (s.hz_axis == 1u ? (float)nd.y : (float)nd.z);
            const float a = s.hz_axis == 0u ? (float)nd.y : (s.hz_axis == 1u ? (float)nd.z : (float)nd.x), b = s.hz_axis == 0u ? (float)nd.z : (s.hz_axis == 1u ? (float)nd.x : (float)nd.y);
            const uint32_t q = s.horizon[(size_t)prim * 32u + (u < 0.0f ? 16u : 0u) + hz_sector(a, b)];
            // (the tables speak for rays that start ON the triangle; the spawn point is within 2^-24 x the scene's size of its plane, and the ray's line then meets the
            // plane inside the triangle - where the tables hold - if the point is this far from the edges for this inclination: HzTables::tau. Both loads are issued
            // before either comparison: one latency, not two in a row)
#ifdef RRT_HZ_NO_GUARD   // measurement variant only
            const float tau = -1.0f;
#else
            const float tau = s.hz_tau[prim];
#endif
            const bool above = fabsf(u) * 254.0f > (float)q;
            const bool inside = fminf(fminf((float)h.z, (float)h.w), 1.0f - (float)h.z - (float)h.w) * fabsf((float)dot(si.n, nd)) > tau;
            if (above && inside) { cont = false; sky = true; }
          }
          if (cont && (int)bounces < s.max_depth) {
            nx_o = si.p; nx_d = nd;
            nx_beta = beta; nx_eta_scale = eta_scale;
            nx_db = db_pack(s, dim, bounces);
            want_next = true;
          }
        }
      }
    }
  }
  const uint32_t qs = block_push(p.shadow_count, want_shadow, push_lds);
  if (want_shadow) {
    store_ray<R>(p.sray_o, p.sray_d, qs, sh_o, o_lo, sh_d, R(1) - R(0.0001), self_prim<R>(prim) | (int)(sh_tab << 24));   // (sh_tab != 0 only in fp32, where prim >= 0 and < 2^19)
    p.sld[qs] = mk4u<R>(sh_ld.r, sh_ld.g, sh_ld.b, slot);
  }
  const uint32_t qn = block_push(&p.counters[C_NEXT], want_next, push_lds);
  if (want_next) {
    p.q_next[qn] = QEnt{slot, nx_db, index, 0u};
    p.npath[qn] = mk4<R>(nx_beta.r, nx_beta.g, nx_beta.b, nx_eta_scale);
    store_ray<R>(p.nray_o, p.nray_d, qn, nx_o, o_lo, nx_d, Const<R>::inf, self_prim<R>(prim));
  }
  if (s.horizon != nullptr) (void)block_push(&p.counters[C_SKY], sky, push_lds);   // (block-uniform condition)
  }  // the chunk's hits
  __syncthreads();   // hit_idx is rewritten by the next chunk
  }  // queue walk
}

// DirectLighting / Debug integrators (directlighting.rs:72-132, intersect_debug.rs:56-89) as a wavefront chain:
// one NEE launch per light (strategy all) or one (strategy one), then the specular continuation.
// mode: light_j >= 0 -> uniform_sample_all_lights' j-th term; light_j == -1 -> uniform_sample_one_light(None).
template <typename R>
__global__ void __launch_bounds__(ShadeBlock<R>::n) k_shade_nee(SceneDev<R> s, Pools<R> p, int light_j, int first) {
  __shared__ uint32_t push_lds[ShadeBlock<R>::n / 64 + 1];
  using V4 = typename Vec4T<R>::type;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = p.counters[C_ACTIVE];
  if (blockIdx.x * blockDim.x >= n) return;   // whole block past the queue end (the grid is sized for the worst case)
  bool want_shadow = false;
  uint32_t slot = 0, sh_tab = 0;
  int prim = -1;
  V3<R> sh_o, sh_d, o_lo;
  Rgb<R> sh_ld;
  if (i < n) {
    const QEnt qe = p.q_active[i];
    slot = qe.slot;
    const V4 h = p.hit[i];
    prim = (int)real_to_bits(h.y);
    if (prim >= 0) {
      const V4 ro = p.ray_o[i], rd = p.ray_d[i];
      V3<R> o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
      Surf<R> si = build_surface(s, prim, o, d, h.x, h.z, h.w);
      if (!si.ok) atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_SHADING_NORMAL);
      else {
        Bsdf<R> bsdf;
        build_bsdf(s, si, &bsdf, false);   // allow_multiple_lobes = false (directlighting.rs:91)
        const uint32_t db = qe.db;
        uint32_t dim = db_dim(s, db);
        const V4 st_b = p.path[i];
        const uint32_t index = qe.index;
        Rgb<R> beta(st_b.x, st_b.y, st_b.z);
        if (first && s.integrator == 2) {  // Debug: l = Spectrum(0.1) on a hit (intersect_debug.rs:66-70)
          V4 l = p.L[slot];
          l.x += beta.r * R(0.1); l.y += beta.g * R(0.1); l.z += beta.b * R(0.1);
          p.L[slot] = l;
        }
        uint32_t ln;
        R pick_pdf = R(1);
        if (light_j >= 0) ln = (uint32_t)light_j;
        else {
          R u_pick = to_real<R>(draw_1d(s, index, &dim));
          R v = u_pick * (R)s.n_lights;
          uint32_t vi = (v != v || v <= R(0)) ? 0u : (uint32_t)v;
          ln = vi < s.n_lights - 1 ? vi : s.n_lights - 1;
          pick_pdf = R(1) / (R)s.n_lights;
        }
        double du0, du1;
        draw_2d(s, index, &dim, &du0, &du1);
        R ul0 = to_real<R>(du0), ul1 = to_real<R>(du1);
        skip_2d(s, &dim);  // u_scattering
        V3<R> so, sd;
        Rgb<R> ld;
        if (estimate_direct_light(si, bsdf, s.lights[ln], ul0, ul1, &so, &sd, &ld)) {
          sh_ld = beta * (ld / pick_pdf);
          sh_o = so; sh_d = sd; o_lo = si.p_lo;
          sh_tab = s.use_shadow_tabs ? s.lights[ln].shadow_tab : 0u;
          want_shadow = true;
        }
        p.q_active[i].db = db_pack(s, dim, db_bounce(s, db));
      }
    }
  }
  const uint32_t qs = block_push(p.shadow_count, want_shadow, push_lds);
  if (want_shadow) {
    store_ray<R>(p.sray_o, p.sray_d, qs, sh_o, o_lo, sh_d, R(1) - R(0.0001), self_prim<R>(prim) | (int)(sh_tab << 24));
    p.sld[qs] = mk4u<R>(sh_ld.r, sh_ld.g, sh_ld.b, slot);
  }
}

// specular_reflect (integrator/mod.rs:150-198) as the chain's continuation; this chain only runs on scenes without
// transmissive or textured materials (those take k_direct_tree), where specular_transmit (:199-301) finds no lobe and
// its 2D draw lands after the recursion returns.
// `depth` of the reference starts at 1; the high half of dim_bounce stores depth - 1.
template <typename R>
__global__ void __launch_bounds__(ShadeBlock<R>::n) k_shade_specular(SceneDev<R> s, Pools<R> p, int grey_only) {
  __shared__ uint32_t push_lds[ShadeBlock<R>::n / 64 + 1];
  using V4 = typename Vec4T<R>::type;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = p.counters[C_ACTIVE];
  if (blockIdx.x * blockDim.x >= n) return;   // whole block past the queue end (the grid is sized for the worst case)
  bool want_next = false;
  uint32_t slot = 0;
  int prim = -1;
  V3<R> nx_o, nx_d, o_lo;
  Rgb<R> nx_beta;
  uint32_t index = 0, nx_db = 0;
  if (i < n) {
    const QEnt qe = p.q_active[i];
    slot = qe.slot;
    const V4 h = p.hit[i];
    prim = (int)real_to_bits(h.y);
    // DirectLightingIntegrator::li on a miss: `for light in &scene.lights { .. return l }` falls through to a call of itself when the
    // list is empty (directlighting.rs:83-99, Q20): unbounded recursion in the reference, a panic here
    if (prim < 0 && s.integrator == 1 && s.n_lights == 0u) atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_NO_LIGHTS);
    if (prim >= 0) {
      const V4 st_b = p.path[i];
      const uint32_t db = qe.db;
      uint32_t dim = db_dim(s, db);
      const uint32_t depth = db_bounce(s, db) + 1;
      Rgb<R> beta(st_b.x, st_b.y, st_b.z);
      index = qe.index;
      if (grey_only) {  // Debug with no lights still adds the 0.1 grey
        V4 l = p.L[slot];
        l.x += beta.r * R(0.1); l.y += beta.g * R(0.1); l.z += beta.b * R(0.1);
        p.L[slot] = l;
      }
      if ((int)(depth + 1) < s.max_depth) {
        const V4 ro = p.ray_o[i], rd = p.ray_d[i];
        V3<R> o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
        Surf<R> si = build_surface(s, prim, o, d, h.x, h.z, h.w);
        if (si.ok) {
          Bsdf<R> bsdf;
          build_bsdf(s, si, &bsdf, false);
          double db0, db1;
          draw_2d(s, index, &dim, &db0, &db1);
          R u0 = to_real<R>(db0), u1 = to_real<R>(db1);
          V3<R> wi;
          R pdf = R(0);
          uint32_t st = 0;
          Rgb<R> f = bsdf.sample_f(si.wo, &wi, u0, u1, &pdf, BXDF_SPECULAR | BXDF_REFLECTION, &st);
          if (pdf > R(0) && !f.is_black() && absdot(wi, si.sn) != R(0)) {
            nx_beta = beta * (f * absdot(wi, si.sn) / pdf);
            nx_o = si.p; nx_d = vnormalize(wi); o_lo = si.p_lo;
            nx_db = db_pack(s, dim, depth);
            want_next = true;
          }
        }
      }
    }
  }
  const uint32_t qn = block_push(&p.counters[C_NEXT], want_next, push_lds);
  if (want_next) {
    p.q_next[qn] = QEnt{slot, nx_db, index, 0u};
    p.npath[qn] = mk4<R>(nx_beta.r, nx_beta.g, nx_beta.b, R(1));
    store_ray<R>(p.nray_o, p.nray_d, qn, nx_o, o_lo, nx_d, Const<R>::inf, self_prim<R>(prim));
  }
}

// DirectLighting / Debug with transmissive materials. `l += specular_reflect(..) + specular_transmit(..)`
// (directlighting.rs:126-129, integrator/mod.rs:150-301) makes li() a binary recursion, and the sampler dimensions are
// consumed depth-first: the transmit draw of a vertex comes after everything its reflect subtree drew, which depends on
// what that subtree hit. A breadth-first wavefront cannot know that count, so here one thread walks one camera
// sample's whole tree with an explicit stack (traversal, shading and shadow tests inline; a vertex is re-intersected
// when its reflect subtree returns instead of keeping its interaction on the stack). A correctness path, not a fast one.
// TEX: textured materials; every frame then carries its ray's differentials, which specular children inherit
// (integrator/mod.rs:183-201, 238-292) - the reason textured scenes take this kernel even without transmissive materials.
// The explicit stack holds one frame per level of the recursion (at most max_depth): the first kTreeMax in registers / scratch, deeper
// ones in a strided global array `deep` ([level - kTreeMax][thread of the launch], TreeFrame<R> each; null when max_depth <= kTreeMax).
constexpr int kTreeMax = 16;
template <typename R> struct TreeFrame { V3<R> o, d, lo; Rgb<R> beta; int skip, depth, phase; DiffRay<R> dr; };
template <typename R, bool TEX>
__global__ void __launch_bounds__(kBlock) k_direct_tree(SceneDev<R> s, Pools<R> p, unsigned long long* totals, TreeFrame<R>* deep, uint32_t deep_stride) {
  using V4 = typename Vec4T<R>::type;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.counters[C_ACTIVE]) return;
  uint32_t n_closest = 0, n_any = 0;   // query counts for rrt_stats (totals[2], totals[3])
  const QEnt qe = p.q_active[i];
  const uint32_t index = qe.index;
  uint32_t dim = db_dim(s, qe.db);
  using Frame = TreeFrame<R>;
  Frame st[kTreeMax];
  auto frame = [&](int k) -> Frame& { return k < kTreeMax ? st[k] : deep[(size_t)(k - kTreeMax) * deep_stride + i]; };
  const int sp_max = deep ? s.max_depth : kTreeMax;
  int sp = 0;
  {
    const V4 ro = p.ray_o[i], rd = p.ray_d[i];
    st[0].o = V3<R>(ro.x, ro.y, ro.z); st[0].d = V3<R>(rd.x, rd.y, rd.z); st[0].lo = V3<R>();
    st[0].beta = Rgb<R>(R(1)); st[0].skip = -1; st[0].depth = 1; st[0].phase = 0;   // li(ray, .., depth = 1)
    if (TEX) {
      const V4 a = p.rdx_o[qe.slot], b = p.rdx_d[qe.slot], c = p.rdy_o[qe.slot], e = p.rdy_d[qe.slot];
      st[0].dr.has = true;
      st[0].dr.rxo = V3<R>(a.x, a.y, a.z); st[0].dr.rxd = V3<R>(b.x, b.y, b.z); st[0].dr.ryo = V3<R>(c.x, c.y, c.z); st[0].dr.ryd = V3<R>(e.x, e.y, e.z);
    }
    sp = 1;
  }
  Rgb<R> L;
  const bool all_lights = s.integrator == 2 || s.light_strategy == 1 /* RRT_STRATEGY_ALL */;
  while (sp > 0) {
    Frame& f = frame(sp - 1);
    RayCtx<R> r = make_ctx(f.o, f.d, Const<R>::inf, f.lo);
    R hu = 0, hv = 0;
    uint32_t nn, np;
    int hit;
    { PrivStack stack; hit = traverse_closest(s, r, stack, f.skip, &hu, &hv, &nn, &np); }
    if (f.phase == 0) n_closest++;   // the phase-1 re-intersection is bookkeeping, not a reference query
    if (hit < 0) {   // `for light in lights { l += le; return l }`: le = 0; with no light at all DirectLighting recurses for ever (Q20)
      if (s.integrator == 1 && s.n_lights == 0u) atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_NO_LIGHTS);
      sp--; continue;
    }
    SurfExt<R> ext;
    Surf<R> si = build_surface(s, hit, f.o, f.d, r.tmax, hu, hv, TEX ? &ext : nullptr);
    if (!si.ok) { atomicOr(&p.counters[C_ERROR], (uint32_t)ERR_SHADING_NORMAL); sp--; continue; }
    Bsdf<R, 4> bsdf;
    TexCtx<R> tc;
    const DiffRay<R> fdr = f.dr;   // (a child frame may reuse this frame's stack entry below)
    // allow_multiple_lobes = false (directlighting.rs:91, intersect_debug.rs:71)
    if (TEX) { if (!build_bsdf_tex(s, si, ext, fdr, &tc, &bsdf, false)) atomicOr(&p.counters[C_ERROR], (uint32_t)(tc.err ? ERR_MIPMAP : ERR_NULL_BSDF)); }
    else build_bsdf(s, si, &bsdf, false);
    const Rgb<R> beta = f.beta;
    const int depth = f.depth;
    uint32_t want = 0;   // lobe class of the continuation to try now
    if (f.phase == 0) {
      if (s.integrator == 2) L = L + beta * R(0.1);   // intersect_debug.rs:66-70
      if (s.n_lights > 0) {
        const uint32_t n_iter = all_lights ? s.n_lights : 1u;
        for (uint32_t j = 0; j < n_iter; j++) {
          uint32_t ln = j;
          R pick_pdf = R(1);
          if (!all_lights) {   // uniform_sample_one_light(.., None) integrator/mod.rs:377-386
            const R v = to_real<R>(draw_1d(s, index, &dim)) * (R)s.n_lights;
            const uint32_t vi = (v != v || v <= R(0)) ? 0u : (uint32_t)v;
            ln = vi < s.n_lights - 1 ? vi : s.n_lights - 1;
            pick_pdf = R(1) / (R)s.n_lights;
          }
          double du0, du1;
          draw_2d(s, index, &dim, &du0, &du1);
          skip_2d(s, &dim);   // u_scattering
          V3<R> so, sd;
          Rgb<R> ld;
          if (estimate_direct_light(si, bsdf, s.lights[ln], to_real<R>(du0), to_real<R>(du1), &so, &sd, &ld)) {
            RayCtx<R> sr = make_ctx(so, sd, R(1) - R(0.0001), si.p_lo);
            PrivStack stack;
            n_any++;
            if (!traverse_any(s, sr, stack, self_prim<R>(hit), &nn, &np)) L = L + beta * (ld / pick_pdf);
          }
        }
      }
      if (depth + 1 < s.max_depth) { f.phase = 1; want = BXDF_SPECULAR | BXDF_REFLECTION; }
      else sp--;
    } else {
      want = BXDF_SPECULAR | BXDF_TRANSMISSION;
      sp--;   // this vertex is finished once its transmit child (if any) is pushed
    }
    if (want) {   // specular_reflect / specular_transmit: one 2D draw each, whether or not such a lobe exists
      double db0, db1;
      draw_2d(s, index, &dim, &db0, &db1);
      V3<R> wi;
      R pdf = R(0);
      uint32_t sampled = 0;
      const Rgb<R> fs = bsdf.sample_f(si.wo, &wi, to_real<R>(db0), to_real<R>(db1), &pdf, want, &sampled);
      if (pdf > R(0) && !fs.is_black() && absdot(wi, si.sn) != R(0) && sp < sp_max) {
        Frame& c = frame(sp++);
        c.o = si.p; c.lo = si.p_lo; c.d = vnormalize(wi);
        c.beta = beta * (fs * absdot(wi, si.sn) / pdf);
        c.skip = self_prim<R>(hit); c.depth = depth + 1; c.phase = 0;
        c.dr = DiffRay<R>();
        if (TEX && fdr.has) {
          DiffRay<R> cd;
          cd.has = true;
          cd.rxo = si.p + tc.dpdx; cd.ryo = si.p + tc.dpdy;
          V3<R> ns = si.sn;
          V3<R> dndx = ext.sdndu * tc.dudx + ext.sdndv * tc.dvdx;
          V3<R> dndy = ext.sdndu * tc.dudy + ext.sdndv * tc.dvdy;
          const V3<R> wo = si.wo;
          if (want & BXDF_REFLECTION) {   // integrator/mod.rs:183-201 (the factor is 0.2 there, not pbrt's 2)
            const V3<R> dwodx = -fdr.rxd - wo, dwody = -fdr.ryd - wo;
            const R ddndx = dot(dwodx, ns) + dot(wo, dndx), ddndy = dot(dwody, ns) + dot(wo, dndy);
            cd.rxd = wi - dwodx + (dndx * dot(wo, ns) + ns * ddndx) * R(0.2);
            cd.ryd = wi - dwody + (dndy * dot(wo, ns) + ns * ddndy) * R(0.2);
          } else {                        // :238-292
            R eta = R(1) / bsdf.eta;
            if (dot(wo, ns) < R(0)) { eta = R(1) / eta; ns = -ns; dndx = -dndx; dndy = -dndy; }
            const V3<R> dwodx = -fdr.rxd - wo, dwody = -fdr.ryd - wo;
            const R ddndx = dot(dwodx, ns) + dot(wo, dndx), ddndy = dot(dwody, ns) + dot(wo, dndy);
            const R mu = eta * dot(wo, ns) - absdot(wi, ns);
            const R dmudx = ddndx * (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns));
            const R dmudy = ddndy * (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns));
            cd.rxd = wi - dwodx * eta + (dndx * mu + ns * dmudx);
            cd.ryd = wi - dwody * eta + (dndy * mu + ns * dmudy);
          }
          c.dr = cd;
        }
      }
    }
  }
  V4 l = p.L[qe.slot];
  l.x += L.r; l.y += L.g; l.z += L.b;
  p.L[qe.slot] = l;
  if (n_closest) atomicAdd(&totals[2], (unsigned long long)n_closest);
  if (n_any) atomicAdd(&totals[3], (unsigned long long)n_any);
}

// queue rotation between bounces: active <- next, next <- 0, shadow <- 0 (single thread)
static __global__ void k_rotate(uint32_t* c, int what) {
  if (what == 0) { c[C_ACTIVE] = c[C_NEXT]; c[C_NEXT] = 0; c[C_SHADOW] = 0; }
  else if (what == 1) { c[C_SHADOW] = 0; }
  else if (what == 3) { /* only the work counters */ }
  else if (what == 4) { c[C_NEXT] = 0; }
  // 5 / 6: the two halves of `0` when the shadow launch runs on its own stream beside the next closest-hit launch
  else if (what == 5) { c[C_ACTIVE] = c[C_NEXT]; c[C_NEXT] = 0; c[C_WORK_CLOSEST] = 0; c[C_WORK_AUX] = 0; for (int k = 0; k < 8; k++) c[C_WORK8_CLOSEST + 32 * k] = 0; return; }
  else if (what == 6) { c[C_SHADOW] = 0; c[C_WORK_SHADOW] = 0; for (int k = 0; k < 8; k++) c[C_WORK8_SHADOW + 32 * k] = 0; return; }
  else if (what == 7) { c[C_SHADOW2] = 0; c[C_WORK_SHADOW] = 0; for (int k = 0; k < 8; k++) c[C_WORK8_SHADOW + 32 * k] = 0; return; }
  else { c[C_ACTIVE] = 0; c[C_NEXT] = 0; c[C_SHADOW] = 0; }
  c[C_WORK_CLOSEST] = 0; c[C_WORK_SHADOW] = 0; c[C_WORK_AUX] = 0;
  for (int k = 0; k < 8; k++) { c[C_WORK8_CLOSEST + 32 * k] = 0; c[C_WORK8_SHADOW + 32 * k] = 0; }
}
static __global__ void k_accumulate_counts(uint32_t* c, unsigned long long* totals) {
  // totals[2] closest queries, totals[3] shadow queries, totals[4] camera rays
  totals[2] += c[C_ACTIVE];
}
static __global__ void k_accumulate_shadow(const uint32_t* shadow_count, unsigned long long* totals) { totals[3] += *shadow_count; }
// bounce rays answered by the horizon tables in the shading launch just finished: queries all the same (totals[8] says how many)
static __global__ void k_accumulate_sky(uint32_t* c, unsigned long long* totals) { totals[2] += c[C_SKY]; totals[8] += c[C_SKY]; c[C_SKY] = 0; }
static __global__ void k_accumulate_camera(uint32_t* c, unsigned long long* totals) {
  totals[4] += c[C_CAMERA_RAYS]; c[C_CAMERA_RAYS] = 0;
  totals[2] += c[C_CULLED]; totals[7] += c[C_CULLED]; c[C_CULLED] = 0;   // queries all the same: answered by the root-box test in the camera kernel
}

// ------------------------------------------------------------------------------------------------------------
// Film: FilmTile::add_sample (film.rs:77-130) + merge_film_tile (:248-263, Q3) for the box filter of radius
// <= 0.5: every sample lands in its own pixel, so one thread owns a pixel and sums the pass's samples in sample
// order (deterministic, no atomics). film = per pixel {X, Y, Z sums, filter_weight_sum}.
// ------------------------------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(kBlock) k_film_box(SceneDev<R> s, Pools<R> p, PassDesc pd, R* film) {
  const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= pd.npix) return;
  uint32_t px_, py_;
  pass_pixel(pd, pd.pix_begin + pl, &px_, &py_);
  const uint32_t pix = py_ * (uint32_t)s.xres + px_;
  // The handle's internal film holds the pixel's running contribution sum as RGB + weight (FilmTilePixel film.rs:22-27): the samples of every
  // pool pass are added to it one by one, in sample order, so the sum does not depend on how the samples were cut into passes (a frame and
  // its band partition agree bit for bit also when they need different numbers of passes); k_film_add converts to XYZ once, as
  // merge_film_tile does (film.rs:248-263).
  R* px = film + 4 * (size_t)pix;
  R cr = px[0], cg = px[1], cb = px[2], wsum = px[3];
  constexpr uint32_t kBatch = 8;   // independent loads in flight per thread: a band of a frame has too few pixels to hide the latency otherwise
  auto add = [&](R w, const typename Vec4T<R>::type& l) {
    Rgb<R> L;
    if (w > R(0)) L = Rgb<R>(l.x, l.y, l.z);   // dead samples: L = 0, w = 0 (Q2); their record is never initialised
    // integrator/mod.rs:105-122
    if (L.has_nan()) L = Rgb<R>();
    else if (L.y() < R(-1e-5)) L = Rgb<R>();
    else if (isinf(L.y())) L = Rgb<R>();
    if (L.y() > s.max_sample_luminance) L = L * (s.max_sample_luminance / L.y());
    cr += (L.r * w) * R(1); cg += (L.g * w) * R(1); cb += (L.b * w) * R(1);  // box filter table weight 1
    wsum += R(1);
  };
  uint32_t sl = 0;
  // few pixels (a band of a frame): latency-bound, batch the loads; many pixels: bandwidth-bound, and seven samples in ten are
  // dead on the 100k-triangle config, so their L records are better left unread
  if (pd.npix < (1u << 18))
  for (; sl + kBatch <= pd.ns; sl += kBatch) {   // samples are added in sample order, as the tile loop does
    R wb[kBatch];
    typename Vec4T<R>::type lb[kBatch];
#pragma unroll
    for (uint32_t k = 0; k < kBatch; k++) { wb[k] = p.weight[(sl + k) * pd.npix + pl]; lb[k] = p.L[(sl + k) * pd.npix + pl]; }
#pragma unroll
    for (uint32_t k = 0; k < kBatch; k++) add(wb[k], lb[k]);
  }
  for (; sl < pd.ns; sl++) {
    const R w = p.weight[sl * pd.npix + pl];
    typename Vec4T<R>::type l = mk4<R>(R(0), R(0), R(0), R(0));
    if (w > R(0)) l = p.L[sl * pd.npix + pl];
    add(w, l);
  }
  px[0] = cr; px[1] = cg; px[2] = cb; px[3] = wsum;
}

// Filters wider than one pixel (TriangleFilter / GaussianFilter / wide BoxFilter, film.rs:77-130 with the 16x16
// table): a sample at p_film touches pixels [ceil(p - 0.5 - r), trunc(p - 0.5 + r)] in x and y. Gather form: one
// thread owns a film pixel and visits the samples of every pixel of this pass within reach, applying exactly the
// reference's footprint test and table lookup - deterministic, no atomics, and contributions that cross a rect /
// band / rank border simply land in that pixel of this handle's film (the multi-GPU reduce is a sum).
RRT_DEV bool pass_pixel_inverse(const PassDesc& pd, int x, int y, uint32_t* pl) {
  if (x < pd.rx0 || x >= pd.rx0 + pd.rw || y < pd.ry0) return false;
  const uint32_t yy = (uint32_t)(y - pd.ry0), band = yy / pd.band_h;
  if (band % pd.n_ranks != pd.rank) return false;
  const uint32_t row = (band / pd.n_ranks) * pd.band_h + yy % pd.band_h, col = (uint32_t)(x - pd.rx0);
  uint64_t lin = (uint64_t)row * (uint32_t)pd.rw + col;
  if (pd.tiled) lin = ((uint64_t)(row / kTileH) * ((uint32_t)pd.rw / kTileW) + col / kTileW) * (kTileW * kTileH) + (row % kTileH) * kTileW + col % kTileW;
  if (lin < pd.pix_begin || lin >= (uint64_t)pd.pix_begin + pd.npix) return false;
  *pl = (uint32_t)(lin - pd.pix_begin);
  return true;
}
template <typename R>
__global__ void __launch_bounds__(kBlock) k_film_wide(SceneDev<R> s, Pools<R> p, PassDesc pd, R* film, int ex0, int ey0, int ew, int eh, int reach_x, int reach_y, int ymax) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint32_t)ew * (uint32_t)eh) return;
  const int x = ex0 + (int)(t % (uint32_t)ew), y = ey0 + (int)(t / (uint32_t)ew);
  R* px = film + 4 * ((size_t)y * (size_t)s.xres + (size_t)x);
  R cr = px[0], cg = px[1], cb = px[2], wsum = px[3];   // running RGB + weight sums (see k_film_box)
  const R inv_rx = s.filter_inv_rx, inv_ry = s.filter_inv_ry;
  for (int sy = y - reach_y; sy <= y + reach_y; sy++) {
    if (sy < 0 || sy >= ymax) continue;
    for (int sx = x - reach_x; sx <= x + reach_x; sx++) {
      uint32_t pl;
      if (sx < 0 || sx >= s.xres || !pass_pixel_inverse(pd, sx, sy, &pl)) continue;
      for (uint32_t sl = 0; sl < pd.ns; sl++) {
        const uint32_t slot = sl * pd.npix + pl;
        const typename Vec4T<R>::type cs = p.samp[slot];
        const R dx = cs.x - R(0.5), dy = cs.y - R(0.5);
        // p0 = ceil(d - r), p1 = trunc(d + r) + 1 (Point2i::from truncates), clipped to the film by the tile bounds
        const R p0x = ceil(dx - s.filter_rx), p0y = ceil(dy - s.filter_ry);
        const R p1x = trunc(dx + s.filter_rx) + R(1), p1y = trunc(dy + s.filter_ry) + R(1);
        if ((R)x < p0x || (R)x >= p1x || (R)y < p0y || (R)y >= p1y) continue;
        const R fy = rabs(((R)y - dy) * inv_ry * R(16)), fx = rabs(((R)x - dx) * inv_rx * R(16));
        const int ify = (int)rmin(floor(fy), R(15)), ifx = (int)rmin(floor(fx), R(15));
        const R fw = s.filter_table[ify * 16 + ifx];
        const R w = p.weight[slot];
        Rgb<R> L;
        if (w > R(0)) { const typename Vec4T<R>::type l = p.L[slot]; L = Rgb<R>(l.x, l.y, l.z); }
        if (L.has_nan()) L = Rgb<R>();
        else if (L.y() < R(-1e-5)) L = Rgb<R>();
        else if (isinf(L.y())) L = Rgb<R>();
        if (L.y() > s.max_sample_luminance) L = L * (s.max_sample_luminance / L.y());
        cr += (L.r * w) * fw; cg += (L.g * w) * fw; cb += (L.b * w) * fw;
        wsum += fw;
      }
    }
  }
  px[0] = cr; px[1] = cg; px[2] = cb; px[3] = wsum;
}

// Film::merge_film_tile (film.rs:248-263): the pixel's RGB contribution sum -> XYZ (rgb_to_xyz spectrum.rs:2084-2090), added to the
// caller's film; filter_weight_sum is added three times per merged pixel (Q3). `n` = pixels.
template <typename R>
__global__ void k_film_add(const R* src, R* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const R cr = src[4 * i], cg = src[4 * i + 1], cb = src[4 * i + 2], wsum = src[4 * i + 3];
  R* px = dst + 4 * i;
  px[0] += R(0.412453) * cr + R(0.357580) * cg + R(0.180423) * cb;
  px[1] += R(0.212671) * cr + R(0.715160) * cg + R(0.072169) * cb;
  px[2] += R(0.019334) * cr + R(0.119193) * cg + R(0.950227) * cb;
  px[3] += wsum; px[3] += wsum; px[3] += wsum;
}

}  // namespace rrtd
