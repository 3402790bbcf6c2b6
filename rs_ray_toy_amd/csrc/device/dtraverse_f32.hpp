// fp32 production traversal kernels for gfx950.
//
// Layout: interior nodes are re-packed on the host into 64-byte "pair" nodes that hold BOTH children's boxes
// (first child = linear index + 1, second child = LinearBVHNode::offset, bvh.rs:728-751), so one step of the
// walk costs one dependent 64 B fetch instead of two dependent 32 B fetches. The reference tests the second
// child's box later, with the t_max left by the first subtree; here it is tested together with the first one
// and pushed with its entry distance, and the one t_max-dependent comparison (`t_min < ray.t_max`,
// geometry.rs:1799) is repeated at pop time - with the t_max of THAT moment: a closest-hit t_max can grow as well as shrink (each
// accepted triangle overwrites it, Q10), so a far child is pushed whenever its slabs are hit and judged only when popped; every other
// part of the box test does not depend on t_max. Each ray therefore makes exactly the decisions BVHAccel::intersect / intersect_p make,
// in the same order (near child by split-axis sign, leaf triangles in ordered_prims order, every accepted hit overwrites the
// previous one: Q10) — the same winners, t, u, v as the generic kernels in dkernels.hpp (tests compare them; the slab test below treats a
// NaN plane distance differently, see box_slabs_f32).
//
// What these kernels cost on MI355X, measured (tools/micro/tcp_gather2.hip, valu_rate.hip):
//  * vector L1 (TCP): a 16-byte load whose 64 lanes read their own nodes costs the CU's one TCP 39 cycles (1.6 lanes per clock, 26 B/clk per
//    CU - L1 hits; no cheaper when the lanes of a quad share a line, twice as dear per instruction for 4-byte loads), i.e. 156 cycles per
//    64 pair-node fetches, against ~225 SIMD cycles of arithmetic spread over four SIMDs. The closest-hit kernel keeps that unit busiest.
//  * VALU: v_mul / v_fma / v_sub / v_mov / v_and are full rate (2.7 cycles per wave-instruction), v_min / v_max / v_min3 / v_cmp / v_cndmask /
//    v_bfe / shifts half rate (4.3), packed fp32 (v_pk_*) half rate too - two results for the price of two, so it only pays where the
//    operands already sit in register pairs: PairNode's field order.
//  * SALU: 4.3 cycles per instruction and SIMD (one scalar unit per CU) - condition logic on lane masks runs beside the VALU, not for free.
// Hence (a) the top pair nodes (BFS order: the levels every ray walks) are staged in LDS once per workgroup - no TCP traffic for them;
// (b) the traversal stack lives in LDS ([entry][thread], conflict-free), deeper entries spill to a strided global array; (c) workgroups
// are persistent, so the treelet is loaded once per workgroup.
#pragma once
#include "dkernels.hpp"

// refill thresholds of the persistent-thread kernels: idle lanes of a wave before it fetches new work
#ifndef RRT_VOTE_A
#define RRT_VOTE_A 1u
#define RRT_VOTE_B 2u
#endif
#ifndef RRT_NODE_STEPS
#define RRT_NODE_STEPS 2
#endif
#ifndef RRT_TR_REFILL
#define RRT_TR_REFILL 16u
#endif
namespace rrtd {

constexpr int kTravBlock = 512;     // 8 waves share one LDS copy of the treelet
constexpr int kStackLds = 8;        // LDS stack entries per thread (8 B each); deeper entries spill to global (>= kAnyList: a list starts on the LDS part)
constexpr int kTreeletNodes = 512;  // top of the tree (BFS order) staged in LDS: 512 pair nodes = 32 KB

// Field order chosen for the packed fp32 VALU forms (v_pk_add_f32 / v_pk_mul_f32 work on aligned register pairs, and a 128-bit load
// lands in four consecutive registers): every 64-bit half of the three box words pairs two plane coordinates with the SAME ray
// constants - (x, y) against (o.x, o.y) / (inv.x, inv.y), (z, z) against o.z / inv.z - so the 24 subtract / multiply operations of the two
// slab tests are 12 packed instructions. The child words are what the traversal stack holds, ready made.
struct alignas(64) PairNode {
  float xy0[4];               // first child (linear index + 1):  bmin.x, bmin.y, bmax.x, bmax.y
  float xy1[4];               // second child:                    bmin.x, bmin.y, bmax.x, bmax.y
  float zz[4];                // first child bmin.z, bmax.z, second child bmin.z, bmax.z
  uint32_t id0, id1;          // child words: interior = byte offset of its PairNode (bit 31 clear); leaf = kLeafBit | kSpecialLeaf? | n_prims << 19 | first triangle
  uint32_t axis;              // split axis (bvh.rs:183-236: dir_is_neg[axis] visits the second child first)
  uint32_t pad;
};
constexpr uint32_t kLeafBit = 0x80000000u;
// Leaf word = kLeafBit | kSpecialLeaf? | n_prims (11 bits) << 19 | first primitive (19 bits). kSpecialLeaf: the leaf holds a primitive that is
// not a world-space triangle - a sphere (Tri::plane == kSphereMark) or a triangle of a kept instance (Tri::material & kInstFlag, tested in
// object space through the instance's own ray transform, primitives.rs:115-139) - and takes the rare path special_leaf_f32(); only the
// MIXED instantiations of the kernels look at the bit (scenes without such primitives never set it).
constexpr uint32_t kSpecialLeaf = 0x40000000u;
constexpr uint32_t kLeafCountMask = 0x7ffu;
// t_max of the pool's shadow rays (spawn_ray_to: 1 - SHADOW_EPSILON with a unit direction, Q9): the any-hit kernels give every pool shadow
// ray this length, and the host's any-hit start lists (build_pairs()) derive their reach from the same constant
constexpr float kShadowTmax = 1.0f - 0.0001f;
typedef float v2f __attribute__((ext_vector_type(2)));

struct F4 { float x, y, z, w; };
RRT_DEV F4 ld4(const float* p) { const float4 v = *reinterpret_cast<const float4*>(p); return {v.x, v.y, v.z, v.w}; }

struct LaneRay {
  v2f oxy, ixy;       // origin and inverse direction, x and y as a register pair (operands of the packed slab arithmetic)
  v2f ozz, izz;       // z twice
  float dx, dy, dz, tmax;
  float lx, ly, lz;   // low word of the double-float origin (dkernels.hpp spawn_point())
  uint32_t neg;       // bit k: inv_dir[k] < 0
  uint32_t skip_plane;
#ifdef RRT_SLAB_FMA
  float nox, noy, noz;   // -(o * inv), rounded once: a plane distance is fma(b, inv, no) (see lane_ray_set_inv)
#endif
};

// Slab arithmetic, two forms.
//  default       t = (b - o) * inv - the reference's operations (geometry.rs:1767-1800), relative error 2 ulp of t, which the test's widening factor
//                1 + 2 gamma(3) on the far planes absorbs: 24 subtract / multiply operations per pair node (12 packed instructions, half rate).
//  RRT_SLAB_FMA  t = fma(b, inv, -(o * inv)): 12 full-rate instructions per pair node, no subtraction. The rounding of o * inv is an ABSOLUTE error in t
//                of u |o| |inv| (u = 2^-24), i.e. a plane shifted by up to u (2 |o| + |b|) in world units; the host pads every fp32 box outward by
//                kSlabPadUlps u M (M = the largest coordinate of the root box and of the camera, rrt_impl.hpp upload_scene), so a box the exact
//                arithmetic hits is never missed - conservative like the reference's test, for rays that start within M of the world origin.
//                |inv| is clamped to 1e30 so that an axis-parallel ray gives +-huge plane distances of the right sign instead of inf - inf.
RRT_DEV void lane_ray_set_inv(LaneRay& r) {
#ifdef RRT_SLAB_FMA
  const float ix = fminf(fmaxf(1.0f / r.dx, -1e30f), 1e30f), iy = fminf(fmaxf(1.0f / r.dy, -1e30f), 1e30f), iz = fminf(fmaxf(1.0f / r.dz, -1e30f), 1e30f);
  r.ixy = v2f{ix, iy}; r.izz = v2f{iz, iz};
  r.nox = -(r.oxy.x * ix); r.noy = -(r.oxy.y * iy); r.noz = -(r.ozz.x * iz);
#else
  r.ixy = v2f{1.0f / r.dx, 1.0f / r.dy}; r.izz.x = 1.0f / r.dz; r.izz.y = r.izz.x;
#endif
  r.neg = (r.ixy.x < 0.0f ? 1u : 0u) | (r.ixy.y < 0.0f ? 2u : 0u) | (r.izz.x < 0.0f ? 4u : 0u);
}
constexpr float kSlabPadUlps = 4.0f;

// Bounds3::intersect_p geometry.rs:1767-1800, split: everything except the `t_min < ray.t_max` comparison.
// Returns false when the slabs miss or t_max <= 0; *tmin_out is the entry distance compared against ray.t_max.
// Straight-line form of the reference's sequence of ifs: the same comparisons on the same values (a comparison
// with NaN is false in both), evaluated unconditionally instead of returning early.
RRT_DEV bool box_slabs_f32(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, const LaneRay& r, float* tmin_out) {
  const float g = 1.0f + 2.0f * ((3.0f * 5.9604645e-8f) / (1.0f - 3.0f * 5.9604645e-8f));
  // The same decisions with a third fewer instructions (the kernel is VALU-issue bound). Per axis the reference picks the near / far plane
  // by the sign of inv_dir, i.e. near = min, far = max of the two plane distances; its chain of pairwise rejections
  //   t_min > ty_max || ty_min > t_max, then the same against z, with the far values widened by g, and finally t_max > 0
  // accepts exactly when max(near_x, near_y, near_z) <= g * min(far_x, far_y, far_z) and that minimum is positive: three intervals
  // that all reach beyond 0 intersect pairwise iff they share a point, and multiplying by g > 0 commutes with min (rounding is monotone).
  // Only a NaN plane distance is treated differently (0 * inf: a ray exactly parallel to a slab AND starting exactly on its plane - the
  // reference lets such an x slab reject and ignores such a y / z slab, v_min / v_max ignore it on every axis).
#ifdef RRT_SLAB_FMA
  const float x0 = __builtin_fmaf(bminx, r.ixy.x, r.nox), x1 = __builtin_fmaf(bmaxx, r.ixy.x, r.nox);
  const float y0 = __builtin_fmaf(bminy, r.ixy.y, r.noy), y1 = __builtin_fmaf(bmaxy, r.ixy.y, r.noy);
  const float z0 = __builtin_fmaf(bminz, r.izz.x, r.noz), z1 = __builtin_fmaf(bmaxz, r.izz.x, r.noz);
#else
  const float x0 = (bminx - r.oxy.x) * r.ixy.x, x1 = (bmaxx - r.oxy.x) * r.ixy.x;
  const float y0 = (bminy - r.oxy.y) * r.ixy.y, y1 = (bmaxy - r.oxy.y) * r.ixy.y;
  const float z0 = (bminz - r.ozz.x) * r.izz.x, z1 = (bmaxz - r.ozz.x) * r.izz.x;
#endif
  const float t_min = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
  const float t_max = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * g;
  *tmin_out = t_min;
  return (t_min <= t_max) & (t_max > 0.0f);
}

// The two slab tests of one pair node: box_slabs_f32 for both children, bit for bit, with the subtractions and multiplications packed
// two to an instruction (PairNode's field order). t[k] = entry distance of child k, the return bits k = "slabs hit and t_max > 0".
// Results as lane masks (scalar registers): what follows is logic on conditions, which belongs on the scalar unit.
struct PairHit { float t0, t1; uint64_t s0, s1; };
RRT_DEV PairHit pair_slabs_f32(const float4 a, const float4 b, const float4 c, const LaneRay& r) {
  const float g = 1.0f + 2.0f * ((3.0f * 5.9604645e-8f) / (1.0f - 3.0f * 5.9604645e-8f));
#ifdef RRT_SLAB_FMA
  const v2f lo0 = v2f{__builtin_fmaf(a.x, r.ixy.x, r.nox), __builtin_fmaf(a.y, r.ixy.y, r.noy)}, hi0 = v2f{__builtin_fmaf(a.z, r.ixy.x, r.nox), __builtin_fmaf(a.w, r.ixy.y, r.noy)};
  const v2f lo1 = v2f{__builtin_fmaf(b.x, r.ixy.x, r.nox), __builtin_fmaf(b.y, r.ixy.y, r.noy)}, hi1 = v2f{__builtin_fmaf(b.z, r.ixy.x, r.nox), __builtin_fmaf(b.w, r.ixy.y, r.noy)};
  const v2f z0 = v2f{__builtin_fmaf(c.x, r.izz.x, r.noz), __builtin_fmaf(c.y, r.izz.x, r.noz)}, z1 = v2f{__builtin_fmaf(c.z, r.izz.x, r.noz), __builtin_fmaf(c.w, r.izz.x, r.noz)};
#else
  const v2f lo0 = (v2f{a.x, a.y} - r.oxy) * r.ixy, hi0 = (v2f{a.z, a.w} - r.oxy) * r.ixy;   // child 0: (x, y) plane distances of bmin, bmax
  const v2f lo1 = (v2f{b.x, b.y} - r.oxy) * r.ixy, hi1 = (v2f{b.z, b.w} - r.oxy) * r.ixy;
  const v2f z0 = (v2f{c.x, c.y} - r.ozz) * r.izz, z1 = (v2f{c.z, c.w} - r.ozz) * r.izz;     // (bmin.z, bmax.z) of child 0, of child 1
#endif
  PairHit h;
  h.t0 = fmaxf(fmaxf(fminf(lo0.x, hi0.x), fminf(lo0.y, hi0.y)), fminf(z0.x, z0.y));
  h.t1 = fmaxf(fmaxf(fminf(lo1.x, hi1.x), fminf(lo1.y, hi1.y)), fminf(z1.x, z1.y));
  const float f0 = fminf(fminf(fmaxf(lo0.x, hi0.x), fmaxf(lo0.y, hi0.y)), fmaxf(z0.x, z0.y)) * g;
  const float f1 = fminf(fminf(fmaxf(lo1.x, hi1.x), fmaxf(lo1.y, hi1.y)), fmaxf(z1.x, z1.y)) * g;
  h.s0 = __builtin_amdgcn_ballot_w64(h.t0 <= f0) & __builtin_amdgcn_ballot_w64(f0 > 0.0f);
  h.s1 = __builtin_amdgcn_ballot_w64(h.t1 <= f1) & __builtin_amdgcn_ballot_w64(f1 > 0.0f);
  return h;
}

// Moller-Trumbore of Triangle::intersect (ANY = false, E2 = p2 - p0) / intersect_p (ANY = true, E2 = p2 - p1: Q11)
// (Measured and dropped: taking the accept / reject decision from origin-relative edge functions U = D.(B x C) ..., which two triangles
// sharing an edge evaluate identically up to the sign - no ray can slip between them. The products are of the size |O - p|^2 while their
// differences are of the size |O - p| * edge, so the barycentrics they imply are good to 2 % only on the 100k-triangle mesh against
// Moller-Trumbore's 3e-5: full-size fp32 parity fell from 98.4 % to 68.6 % of the pixels within 1e-4. DESIGN.md section 4.)
template <bool ANY>
RRT_DEV bool tri_test_vals_f32(const F4 q0, const F4 q1, const F4 q2, const LaneRay& r, float* th, float* uh, float* vh) {
  const V3<float> p0(q0.x, q0.y, q0.z), p1(q0.w, q1.x, q1.y), p2(q1.z, q1.w, q2.x);
  const V3<float> D(r.dx, r.dy, r.dz), O(r.oxy.x, r.oxy.y, r.ozz.x);
  const V3<float> E1 = p1 - p0, E2 = ANY ? (p2 - p1) : (p2 - p0);
  const V3<float> P = cross(D, E2);
  const float a = dot(E1, P);
  const float f = rcp_r(a);
  const V3<float> T = (O - p0) + V3<float>(r.lx, r.ly, r.lz);
  const float u = f * dot(T, P);
  const V3<float> Q = cross(T, E1);
  const float v = f * dot(D, Q);
  const float tt = f * dot(E2, Q);
  // the reference's rejections (triangle.rs:182-201 / 245-264), same comparisons, evaluated together
  const bool reject = (__float_as_uint(q2.w) == r.skip_plane) | ((a > -0.0000001f) & (a < 0.0000001f)) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) |
                      ((u + v) > 1.0f) | (tt < 0.0000001f);
  *th = tt; *uh = u; *vh = v;
  return !reject;
}
template <bool ANY>
RRT_DEV bool tri_test_f32(const float* tp, const LaneRay& r, float* th, float* uh, float* vh) {
  return tri_test_vals_f32<ANY>(ld4(tp), ld4(tp + 4), ld4(tp + 8), r, th, uh, vh);
}

struct TravScene {
  const PairNode* pairs;     // interior nodes, pre-order
  const float* tris;         // Tri<float> array viewed as 12 floats per triangle
  float root_box[6];
  uint32_t root_id;          // the root's child word (a leaf word when the whole tree is one leaf)
  uint32_t n_nodes;
  uint32_t n_treelet;        // pair nodes [0, n_treelet) are the BFS top of the tree (host renumbering)
  uint32_t* overflow;        // stack entries beyond the LDS ones: [entry][thread of the launch], 2 words each
  uint32_t overflow_stride;
  // Shadow rays (any-hit, t_max = 1 - 1e-4, unit direction: Q9) of the pool start ON a triangle and reach less than one unit far. Every
  // ancestor of that triangle's leaf contains the origin, so its box test passes whatever the direction, and a sibling subtree whose box is
  // more than a unit away from the leaf's box fails it whatever the direction: a pair node on the way down whose off-path child is that far
  // away - half of the 17 levels above an average leaf of the 100k-triangle mesh - decides nothing. any_list holds, per triangle, the nodes
  // that do decide something (host: build_pairs()): word 0 = where the ordinary walk resumes (a child word), words 1..6 = the pair nodes
  // above it whose off-path child is within reach, flagged with the child to leave out (kSkip0 / kSkip1: the on-path child - the walk does
  // not go down through it), word 7 = how many of those there are. They start on the lane's stack. The boxes tested, the leaves visited and
  // the triangles tested are the reference's, less the tests that cannot pass; an occlusion query is order independent. Null = off.
  const uint4* any_list;
  // mixed scenes (kSpecialLeaf): the records the generic per-primitive tests of dkernels.hpp read
  const SphereDev<float>* spheres;
  const InstDev<float>* insts;
  // two levels per fetch (k_trace_pt_f32<false, false, true>): the QuadNode array (root = node 0), its first n_qtreelet nodes in BFS order
  const void* quads;
  uint32_t n_qtreelet;
};

// A lane's position in the walk is one child word: an interior node to visit (byte offset of its PairNode, < kIdle), a leaf to test
// (kLeafBit set), or kIdle. The traversal stack holds the same words with the child's entry distance.
constexpr uint32_t kIdle = 0x7fffffffu;
constexpr uint32_t kSkip0 = 1u, kSkip1 = 2u;   // any-hit list entries (TravScene::any_list): low bits of an interior child word
constexpr int kAnyList = 6;                    // flagged entries per list
RRT_DEV bool is_node(uint32_t w) { return w < kIdle; }
RRT_DEV bool is_leaf(uint32_t w) { return (int32_t)w < 0; }

// Ray `idx` of the queue this launch serves (POOL_SHADOW: the pool's shadow rays, t_max = 1 - 1e-4; otherwise the closest-ray arrays
// with their own t_max) -> lane registers; returns the word the walk starts with (kIdle: the ray misses the root box).
template <bool POOL_SHADOW>
RRT_DEV uint32_t lane_ray_begin(const TravScene& ts, const Pools<float>& p, bool pool_shadow, uint32_t idx, LaneRay& r, int* start_tri) {
  const float4 ro = (POOL_SHADOW && pool_shadow) ? p.sray_o[idx] : p.ray_o[idx], rd = (POOL_SHADOW && pool_shadow) ? p.sray_d[idx] : p.ray_d[idx];
  V3<float> lo;
  ray_tail(ro, (POOL_SHADOW && pool_shadow) ? kShadowTmax : Const<float>::inf, &r.tmax, &lo);
  const int sk = (int)__float_as_uint(rd.w);
  r.oxy = v2f{ro.x, ro.y}; r.ozz = v2f{ro.z, ro.z}; r.dx = rd.x; r.dy = rd.y; r.dz = rd.z;
  r.lx = lo.x; r.ly = lo.y; r.lz = lo.z;
  r.skip_plane = sk >= 0 ? __float_as_uint(ts.tris[(size_t)sk * 12 + 11]) : 0xffffffffu;
  lane_ray_set_inv(r);
  *start_tri = (POOL_SHADOW && pool_shadow && ts.any_list) ? sk : -1;   // >= 0: the walk starts from any_list[start_tri] (the caller owns the stack)
  if (*start_tri >= 0) return kIdle;
  float tmin;
  if (ts.n_nodes != 0 && box_slabs_f32(ts.root_box[0], ts.root_box[1], ts.root_box[2], ts.root_box[3], ts.root_box[4], ts.root_box[5], r, &tmin) && tmin < r.tmax)
    return ts.root_id;
  return kIdle;
}

// One pair node: both slab tests, then the reference's order (bvh.rs:183-236: the second child first when dir_is_neg[axis]). The near
// child is judged now, like the reference does. The far child's `t_min < t_max` belongs to the moment it is popped: a closest-hit t_max
// can also GROW in between (each accepted hit overwrites it, Q10), so it is pushed whenever its slabs are hit and judged at pop time.
// Shadow rays never change t_max: they prune at once. The selections are written on lane masks (scalar unit), not on 0 / 1 values.
struct PairStep { uint32_t id_near, id_far; float t_far; bool go_near, push_far; };
template <bool ANY>
RRT_DEV PairStep pair_step_f32(const float4 a, const float4 b, const float4 c, const uint4 d, const LaneRay& r, uint32_t word) {
  PairHit h = pair_slabs_f32(a, b, c, r);
  if (ANY) {   // a list entry (TravScene::any_list) leaves out the child the walk does not go down through
    h.s0 &= ~__builtin_amdgcn_ballot_w64((word & kSkip0) != 0u);
    h.s1 &= ~__builtin_amdgcn_ballot_w64((word & kSkip1) != 0u);
  }
  // lane masks and scalar logic: the compiler turns a select between two conditions into 0 / 1 values and five vector instructions
  const uint64_t m_sf = __builtin_amdgcn_ballot_w64(((r.neg >> d.z) & 1u) != 0u);
  const uint64_t m_s0 = h.s0, m_s1 = h.s1;
  const uint64_t m_h0 = m_s0 & __builtin_amdgcn_ballot_w64(h.t0 < r.tmax), m_h1 = m_s1 & __builtin_amdgcn_ballot_w64(h.t1 < r.tmax);
  const bool sf = __builtin_amdgcn_inverse_ballot_w64(m_sf);
  PairStep st;
  st.go_near = __builtin_amdgcn_inverse_ballot_w64((m_sf & m_h1) | (~m_sf & m_h0));
  st.push_far = __builtin_amdgcn_inverse_ballot_w64(ANY ? ((m_sf & m_h0) | (~m_sf & m_h1)) : ((m_sf & m_s0) | (~m_sf & m_s1)));
  st.id_near = sf ? d.y : d.x;
  st.id_far = sf ? d.x : d.y;
  st.t_far = sf ? h.t0 : h.t1;
  return st;
}

// A leaf that holds a sphere or a triangle of a kept instance (kSpecialLeaf), primitive by primitive in ordered_prims order with the generic
// tests of dkernels.hpp (what traverse_closest / traverse_any do in a leaf): Sphere::intersect / intersect_p behind Geometric /
// TransformedPrimitive (sphere.rs:51-259), Triangle::intersect through the instance's ray transform (primitives.rs:115-139, Q15), plain
// triangles as usual. Rare path: NOT inlined, arguments and results by value, so that its registers (atan2, two affine transforms ...) and
// its address-taken temporaries stay out of the traversal loop, whose occupancy is what the kernel lives on.
struct SpecialOut { float tmax, hu, hv; int hit; bool found; };
template <bool ANY>
__attribute__((noinline)) __device__ SpecialOut special_leaf_f32(const Tri<float>* tris, const SphereDev<float>* spheres, const InstDev<float>* insts, uint32_t word,
                                                                  float ox, float oy, float oz, float dx, float dy, float dz, float lx, float ly, float lz,
                                                                  float tmax, uint32_t skip_plane, int hit, float hu, float hv) {
  SpecialOut out{tmax, hu, hv, hit, false};
  RayCtx<float> r = make_ctx(V3<float>(ox, oy, oz), V3<float>(dx, dy, dz), tmax, V3<float>(lx, ly, lz));
  uint32_t lf = word & 0x7ffffu, ln = (word >> 19) & kLeafCountMask;
  do {
    const Tri<float> tr = tris[lf];
    float t, u, v;
    if (tr.plane == kSphereMark) {
      if (sphere_prim_hit<float, ANY>(spheres[tr.shade], r.o, r.d, &t, &u)) {
        if (ANY) { out.found = true; return out; }
        r.tmax = t; out.hit = (int)lf; out.hu = u; out.hv = 0.0f;   // u carries the root branch
      }
    } else if (tr.plane != skip_plane) {
      if (is_inst_tri(tr)) {   // object-space test, object-space t copied to the world ray (Q15)
        const RayCtx<float> ro = inst_ray(insts[inst_of(tr)], r);
        if (ANY) { if (tri_any(tr, ro)) { out.found = true; return out; } }
        else if (tri_closest(tr, ro, &t, &u, &v)) { r.tmax = t; out.hit = (int)lf; out.hu = u; out.hv = v; }
      } else {
        if (ANY) { if (tri_any(tr, r)) { out.found = true; return out; } }
        else if (tri_closest(tr, r, &t, &u, &v)) { r.tmax = t; out.hit = (int)lf; out.hu = u; out.hv = v; }
      }
    }
    lf++; ln--;
  } while (ln != 0);
  out.tmax = r.tmax;
  return out;
}

// Every triangle of a leaf in ordered_prims order; each accepted hit overwrites the previous one and t_max (Q10). ANY: true at the first hit.
template <bool ANY, bool MIXED>
RRT_DEV bool leaf_step_f32(const TravScene& ts, uint32_t word, LaneRay& r, int* hit, float* hu, float* hv) {
  if (MIXED && (word & kSpecialLeaf) != 0u) {
    const SpecialOut so = special_leaf_f32<ANY>(reinterpret_cast<const Tri<float>*>(ts.tris), ts.spheres, ts.insts, word, r.oxy.x, r.oxy.y, r.ozz.x, r.dx, r.dy, r.dz,
                                                r.lx, r.ly, r.lz, r.tmax, r.skip_plane, *hit, *hu, *hv);
    r.tmax = so.tmax; *hit = so.hit; *hu = so.hu; *hv = so.hv;
    return so.found;
  }
  uint32_t lf = word & 0x7ffffu, ln = (word >> 19) & kLeafCountMask;
  // (Measured and dropped: the next triangle's loads in flight while this one is tested - 12 more registers, 7 waves per SIMD: +0.5 ms per frame.)
  do {
    float t, u, v;
    if (tri_test_f32<ANY>(ts.tris + (size_t)lf * 12, r, &t, &u, &v)) {
      if (ANY) return true;
      r.tmax = t; *hit = (int)lf; *hu = u; *hv = v;
    }
    lf++; ln--;
  } while (ln != 0);
  return false;
}

// ANY = false: closest hit for rays in the pool's ray arrays -> pool hit arrays.
// ANY = true : shadow rays (pool shadow arrays) -> L += Ld when unoccluded; with `occluded` != nullptr the rays
//              are the pool's closest-ray arrays and the verdict is written to occluded[i] (public rrt_trace_any).
// Grid-stride form for small queues: the wave alternates between "every lane walks interior nodes until it holds a leaf" and "every lane
// tests its leaf" (while-while); the BFS top of the tree is read from LDS.
template <bool ANY, bool MIXED = false>
__global__ void __launch_bounds__(kTravBlock) k_trace_pairs_f32(TravScene ts, Pools<float> p, const uint32_t* queue, const uint32_t* count,
                                                                 uint32_t n_fixed, uint8_t* occluded, uint32_t n_lo, uint32_t n_hi) {
  {  // queue-size regime of this kernel (the other traversal kernel is launched next to it for the other regime)
    const uint32_t nn = count ? *count : n_fixed;
    if (nn < n_lo || nn >= n_hi) return;
  }
  __shared__ float4 treelet[kTreeletNodes * 4];
  __shared__ uint2 stk[kStackLds * kTravBlock];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < ts.n_treelet * 4u; i += kTravBlock) treelet[i] = reinterpret_cast<const float4*>(ts.pairs)[i];
  __syncthreads();
  const uint32_t n = count ? *count : n_fixed;
  const uint32_t col = blockIdx.x * kTravBlock + tid;   // overflow-stack column of this resident thread
  const uint32_t treelet_bytes = ts.n_treelet * 64u;
  for (uint32_t gid = col; gid < n; gid += gridDim.x * kTravBlock) {
    LaneRay r;
    int start_tri;
    uint32_t cur = lane_ray_begin<ANY>(ts, p, !occluded, gid, r, &start_tri);
    int hit = -1;
    float hu = 0.0f, hv = 0.0f;
    bool found = false;
    uint32_t sp = 0;
    if (ANY && start_tri >= 0) {   // TravScene::any_list
      const uint4 la = ts.any_list[2 * (size_t)start_tri], lb = ts.any_list[2 * (size_t)start_tri + 1];
      cur = la.x; sp = lb.w;
      stk[0 * kTravBlock + tid] = make_uint2(la.y, 0u); stk[1 * kTravBlock + tid] = make_uint2(la.z, 0u); stk[2 * kTravBlock + tid] = make_uint2(la.w, 0u);
      stk[3 * kTravBlock + tid] = make_uint2(lb.x, 0u); stk[4 * kTravBlock + tid] = make_uint2(lb.y, 0u); stk[5 * kTravBlock + tid] = make_uint2(lb.z, 0u);
    }
    auto pop = [&]() {   // re-checks the one comparison that depends on the current t_max
      cur = kIdle;
      while (sp > 0) {
        sp--;
        uint2 e;
        if (sp < (uint32_t)kStackLds) e = stk[sp * kTravBlock + tid];
        else e = *reinterpret_cast<const uint2*>(ts.overflow + ((size_t)(sp - kStackLds) * ts.overflow_stride + col) * 2);
        if (__uint_as_float(e.y) < r.tmax) { cur = e.x; return; }
      }
    };
    while (__ballot(cur != kIdle) != 0ull) {
      while (is_node(cur)) {
        float4 a, b, c; uint4 d;
        const uint32_t off = ANY ? (cur & ~63u) : cur;
        if (off < treelet_bytes) {
          const float4* tp = treelet + (off >> 4);
          a = tp[0]; b = tp[1]; c = tp[2]; const float4 dd = tp[3];
          d = make_uint4(__float_as_uint(dd.x), __float_as_uint(dd.y), __float_as_uint(dd.z), 0u);
        } else {
          const char* np = reinterpret_cast<const char*>(ts.pairs) + off;
          a = *reinterpret_cast<const float4*>(np); b = *reinterpret_cast<const float4*>(np + 16); c = *reinterpret_cast<const float4*>(np + 32);
          d = *reinterpret_cast<const uint4*>(np + 48);
        }
        const PairStep st = pair_step_f32<ANY>(a, b, c, d, r, cur);
        if (st.push_far) {
          const uint2 e = make_uint2(st.id_far, __float_as_uint(st.t_far));
          if (sp < (uint32_t)kStackLds) stk[sp * kTravBlock + tid] = e;
          else *reinterpret_cast<uint2*>(ts.overflow + ((size_t)(sp - kStackLds) * ts.overflow_stride + col) * 2) = e;
          sp++;
        }
        if (st.go_near) cur = st.id_near;
        else pop();
      }
      if (is_leaf(cur)) {
        found = leaf_step_f32<ANY, MIXED>(ts, cur, r, &hit, &hu, &hv);
        if (ANY && found) cur = kIdle;
        else pop();
      }
    }
    if (ANY) {
      if (occluded) occluded[gid] = found ? 1 : 0;
      else if (!found) add_pending(p, gid);
    } else {
      p.hit[gid] = make_float4(r.tmax, __uint_as_float((uint32_t)hit), hu, hv);
    }
  }  // grid-stride loop over rays
}


// ------------------------------------------------------------------------------------------------------------
// Two levels per fetch ("quad nodes"), closest-hit rays of the persistent kernel only (option "quad_nodes").
// A QuadNode belongs to an interior node N and holds the boxes of N's four GRANDCHILDREN (slots 0 / 1 = the children of N's first child,
// 2 / 3 = of its second child; a child that is a leaf fills one slot with its own box and leaves the other empty: NaN box, kIdle word). The
// children's own boxes are NOT there and are not tested: the reference (bvh.rs:183-236) visits a child, and then tests its children's boxes,
// only if the child's own box passes - but a child's box contains its children's (the host narrows the f64 boxes outward, monotonically) and
// every operation of the slab test is monotone under rounding, so a grandchild's box that passes implies its parent's passes, with an entry
// distance that is no larger; and a child neither of whose children's boxes passes is a visit that tests two boxes, pushes nothing and pops:
// leaving it out changes no triangle test. (As everywhere in this file the one exception is a NaN plane distance.)
// Order: the four slots are visited as the two pair steps would visit them - the first child's two before the second child's two, or the
// other way round when dir_is_neg[axis of N]; inside a child by dir_is_neg[axis of that child]. The first slot in that order whose box is
// hit NOW (slabs and t_min < t_max) is where the lane goes; the slots before it are judged now as well - nothing can change t_max before
// the reference reaches them - and dropped; the slots behind it are pushed whenever their slabs are hit and judged when popped, with the
// t_max of that moment (Q10: it can grow), exactly as pair_step_f32 treats a far child. The reference's `t_min_child < t_max` for a child
// whose first slot is popped later is implied by the slot's own comparison at the same moment (t_min_child <= t_min_slot), and for its
// second slot either nothing happened since (same t_max) or the first slot's subtree was visited, i.e. the child had been accepted.
// Same leaves in the same order, same triangle tests, same t_max sequence: tests/test_gpu_parity.py::test_quad_nodes_change_nothing.
// The three split axes ride in bits 28-29 of the first three child words (leaf words keep 9 bits of primitive count, interior words are
// byte offsets below 2^28).
// ------------------------------------------------------------------------------------------------------------
struct alignas(128) QuadNode {
  float mnx[4], mny[4], mnz[4];   // bmin of the four slots, one axis per 16-byte word
  float mxx[4], mxy[4], mxz[4];   // bmax
  uint32_t id[4];                 // child words (interior: byte offset of its QuadNode); bits 28-29: split axis of N / of its first child / of its second child / -
  uint32_t pad[4];
};
constexpr uint32_t kQuadAxisShift = 28u, kQuadAxisMask = 3u << 28;
constexpr uint32_t kQuadLeafMax = 511u;   // primitives per leaf the stolen bits leave room for
#ifndef RRT_QUAD_TREELET
#define RRT_QUAD_TREELET 32
#endif
constexpr int kQuadTreelet = RRT_QUAD_TREELET;   // top quad nodes (BFS order) in the persistent kernel's LDS: the 4 KB the pair-node treelet takes

struct QuadStep { uint32_t id[4]; float t[4]; };   // the four slots in visit order; t = entry distance, +inf where the slabs miss
RRT_DEV QuadStep quad_step_f32(const float4 mnx, const float4 mny, const float4 mnz, const float4 mxx, const float4 mxy, const float4 mxz, const uint4 ids, const LaneRay& r) {
  const float g = 1.0f + 2.0f * ((3.0f * 5.9604645e-8f) / (1.0f - 3.0f * 5.9604645e-8f));
  const float ox = r.oxy.x, oy = r.oxy.y, oz = r.ozz.x, ix = r.ixy.x, iy = r.ixy.y, iz = r.izz.x;
  float t[4];
#define RRT_QSLOT(k, c)                                                                                                        \
  {                                                                                                                            \
    const float x0 = (mnx.c - ox) * ix, x1 = (mxx.c - ox) * ix, y0 = (mny.c - oy) * iy, y1 = (mxy.c - oy) * iy;                \
    const float z0 = (mnz.c - oz) * iz, z1 = (mxz.c - oz) * iz;                                                                \
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));                                                \
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * g;                                            \
    t[k] = ((tn <= tf) & (tf > 0.0f)) ? tn : Const<float>::inf;   /* box_slabs_f32's verdict, folded into the entry distance */ \
  }
  RRT_QSLOT(0, x) RRT_QSLOT(1, y) RRT_QSLOT(2, z) RRT_QSLOT(3, w)
#undef RRT_QSLOT
  const bool sp = ((r.neg >> ((ids.x >> kQuadAxisShift) & 3u)) & 1u) != 0u;   // dir_is_neg[axis of N]: the second child's slots first
  const bool sa = ((r.neg >> ((ids.y >> kQuadAxisShift) & 3u)) & 1u) != 0u;   // inside the first child
  const bool sb = ((r.neg >> ((ids.z >> kQuadAxisShift) & 3u)) & 1u) != 0u;   // inside the second child
  const uint32_t i0 = ids.x & ~kQuadAxisMask, i1 = ids.y & ~kQuadAxisMask, i2 = ids.z & ~kQuadAxisMask, i3 = ids.w;
  const uint32_t an = sa ? i1 : i0, af = sa ? i0 : i1, bn = sb ? i3 : i2, bf = sb ? i2 : i3;
  const float tan_ = sa ? t[1] : t[0], taf = sa ? t[0] : t[1], tbn = sb ? t[3] : t[2], tbf = sb ? t[2] : t[3];
  QuadStep q;
  q.id[0] = sp ? bn : an; q.id[1] = sp ? bf : af; q.id[2] = sp ? an : bn; q.id[3] = sp ? af : bf;
  q.t[0] = sp ? tbn : tan_; q.t[1] = sp ? tbf : taf; q.t[2] = sp ? tan_ : tbn; q.t[3] = sp ? taf : tbf;
  return q;
}

// ------------------------------------------------------------------------------------------------------------
// Persistent-thread variant: the kernel is VALU-issue bound with ~26 % of the lanes doing useful work when a wave
// owns 64 fixed rays (a wave lasts as long as its longest ray). Here a wave keeps pulling rays: it reserves
// kGrain rays at a time from a global cursor (one atomic per kGrain rays) and refills idle lanes from that private
// range, so lanes stay busy until the queue is empty. Each lane advances one step per loop iteration: node steps
// (RRT_NODE_STEPS pair nodes) or the triangles of its current leaf, whichever the wave votes for; the others wait.
// Per ray the sequence of box tests, triangle tests, acceptances and t_max updates is unchanged.
// ------------------------------------------------------------------------------------------------------------
#ifndef RRT_PT_BLOCK
#define RRT_PT_BLOCK 256
#endif
#ifndef RRT_PT_STACK
#define RRT_PT_STACK 8
#endif
#ifndef RRT_PT_STACK_ANY
#define RRT_PT_STACK_ANY 10
#endif
#ifndef RRT_PT_TREELET
#define RRT_PT_TREELET 64
#endif
// LDS of a 256-thread workgroup, 20 KB so that eight of them (8 waves per SIMD) fit a CU's 160 KB: closest-hit = 8 stack entries per lane +
// the top 64 pair nodes of the tree (BFS order; 28 % of its node fetches, which then do not queue at the vector L1 - the unit this kernel
// keeps busiest); any-hit = 10 stack entries and no treelet (its rays start from their lists, far below the top). Measured frame times
// with (entries, nodes) = (10, 0) 44.4 ms, (9, 32) 43.5, (8, 64) 43.4, (7, 96) 44.6, (6, 128) 45.2, (5, 160) 46.4 when both kernels shared one setting.
constexpr int kPtTreelet = RRT_PT_TREELET;   // closest-hit only
constexpr uint32_t kXcdParts = 8u;   // parts of a queue with their own work cursor (8 XCDs)
constexpr int kPtBlock = RRT_PT_BLOCK;
constexpr int kPtStack = RRT_PT_STACK, kPtStackAny = RRT_PT_STACK_ANY;   // LDS stack entries per lane (closest-hit, any-hit); deeper entries spill to global
constexpr uint32_t kGrain = 256;

#ifdef RRT_PT_STATS
// tuning instrumentation (variant builds only): [0] iterations, [1] node iterations, [2] active lanes over node steps, [3] node steps (wave level),
// [4] leaf iterations, [5] active lanes over leaf iterations, [6] refills, [7] lanes refilled, [8] pop-loop rounds (wave level), [9] lanes over pop rounds,
// [10] idle lanes over iterations, [11] leaf-waiting lanes over node iterations, [12] node-waiting lanes over leaf iterations
__device__ unsigned long long g_pt_stats[2][16];
#define PT_STAT(i, v) st_[i] += (v)
#else
#define PT_STAT(i, v)
#endif
#ifdef RRT_PT_WAVES   // tuning variants: force the register budget of that many waves per SIMD
#define RRT_PT_ATTR __attribute__((amdgpu_waves_per_eu(RRT_PT_WAVES)))
#else
#define RRT_PT_ATTR
#endif
template <bool ANY, bool MIXED = false, bool QUAD = false>
__global__ void __launch_bounds__(kPtBlock) RRT_PT_ATTR k_trace_pt_f32(TravScene ts, Pools<float> p, const uint32_t* queue, const uint32_t* count,
                                                            uint32_t n_fixed, uint32_t* work, uint8_t* occluded, uint32_t n_lo, uint32_t n_hi) {
  static_assert(!QUAD || (!ANY && !MIXED), "quad nodes: closest-hit rays of plain triangle scenes only");
  {
    const uint32_t nn = count ? *count : n_fixed;
    if (nn < n_lo || nn >= n_hi) return;
  }
  constexpr int kStack = ANY ? kPtStackAny : kPtStack, kTl = ANY ? 0 : (QUAD ? 2 * kQuadTreelet : kPtTreelet);   // kTl: LDS treelet in 64-byte units
  __shared__ uint2 stk[kStack * kPtBlock];   // [entry][thread]: {child word, entry distance}, one ds_write_b64 / ds_read_b64 each
  // Top of the tree in LDS. A lane reads the four 16-byte words of ITS node, so with nodes laid out as in memory all lanes of an instruction
  // would hit the 4 of 16 bank groups their word index selects: word w of node n lives at slot w ^ ((n >> 2) & 3) instead.
  __shared__ float4 pt_treelet[kTl > 0 ? kTl * 4 : 1];
  const uint32_t tl_bytes = QUAD ? (ts.n_qtreelet < (uint32_t)kQuadTreelet ? ts.n_qtreelet : (uint32_t)kQuadTreelet) * 128u
                                  : (kTl > 0 ? (ts.n_treelet < (uint32_t)kTl ? ts.n_treelet : (uint32_t)kTl) * 64u : 0u);
  if (QUAD) {   // word w (of 8) of quad node n at slot w ^ (n & 7): the 64 lanes of one read spread over every bank group
    for (uint32_t i = threadIdx.x; i < tl_bytes / 16u; i += kPtBlock) pt_treelet[(i & ~7u) | ((i ^ (i >> 3)) & 7u)] = reinterpret_cast<const float4*>(ts.quads)[i];
    __syncthreads();
  } else if (kTl > 0) {
    for (uint32_t i = threadIdx.x; i < tl_bytes / 16u; i += kPtBlock) pt_treelet[(i & ~3u) | ((i ^ (i >> 4)) & 3u)] = reinterpret_cast<const float4*>(ts.pairs)[i];
    __syncthreads();
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t gtid = blockIdx.x * blockDim.x + tid;   // overflow-stack column of this lane
  const uint32_t n = count ? *count : n_fixed;
  LaneRay r;
  r.oxy = r.ixy = r.ozz = r.izz = v2f{0.0f, 0.0f};
  r.dx = r.dy = r.dz = r.tmax = r.lx = r.ly = r.lz = 0.0f; r.neg = 0; r.skip_plane = 0xffffffffu;
#ifdef RRT_SLAB_FMA
  r.nox = r.noy = r.noz = 0.0f;
#endif
  uint32_t cur = kIdle, qidx = 0, sp = 0;
  int hit = -1;
  float hu = 0.0f, hv = 0.0f;
  uint32_t lo = 0, hi = 0;      // wave-private range of reserved rays
  bool exhausted = false;       // wave-uniform
  const uint32_t home = blockIdx.x & (kXcdParts - 1u);
  uint32_t parts_done = 0;      // parts of the queue this wave has found empty
  // reservation grain: large enough to amortise the atomic, small enough that a short queue still spreads over
  // every resident wave (a wave that reserves 256 rays of a 100k-ray queue would serialise four ray chains)
  const uint32_t n_waves = gridDim.x * (kPtBlock / 64);
  uint32_t grain = (n / (2u * n_waves) + 63u) & ~63u;
  grain = grain < 64u ? 64u : (grain > kGrain ? kGrain : grain);

#ifdef RRT_PT_STATS
  unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  auto finish = [&](bool found) {
    if (ANY) {
      if (occluded) occluded[qidx] = found ? 1 : 0;
      else if (!found) add_pending(p, qidx);
    } else {
      p.hit[qidx] = make_float4(r.tmax, __uint_as_float((uint32_t)hit), hu, hv);
    }
    cur = kIdle;
  };
  auto pop = [&]() {   // re-checks the one comparison that depends on the current t_max
    // (This loop runs 1.7 rounds per node step with 14 lanes in each - a closest-hit lane that has found its hit rejects most of what is left on
    // its stack, tools/pt_stats.py. Measured and dropped: reading the two top entries per round (closest-hit alone 18.0 -> 19.4 ms); one pop
    // attempt per node step for all the lanes that need one, a rejected lane trying again at the next step (1.0 round of 23 lanes, frame -0.3 ms,
    // the kernel alone +0.3 ms: a wash).)
    while (sp > 0) {
      sp--;
      PT_STAT(8, lane == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)) ? 1 : 0); PT_STAT(9, 1);
      // (the LDS read is unconditional and the overflow entry a rare fix-up: an if / else between the two address spaces compiles to flat loads)
      uint2 e = stk[(sp < (uint32_t)kStack ? sp : 0u) * kPtBlock + tid];
      asm volatile("" : "+v"(e.x), "+v"(e.y));
      if (__builtin_expect(sp >= (uint32_t)kStack, 0)) e = *reinterpret_cast<const uint2*>(ts.overflow + ((size_t)(sp - kStack) * ts.overflow_stride + gtid) * 2);
      if (__uint_as_float(e.y) < r.tmax) { cur = e.x; return; }
    }
    finish(false);
  };

  while (true) {
    // ---- refill idle lanes ---------------------------------------------------------------------------------------
    const uint64_t idle = __ballot(cur == kIdle);
    const uint32_t n_idle = (uint32_t)__popcll(idle);
    if (!exhausted && (n_idle >= RRT_TR_REFILL)) {
      if (lo == hi) {
        // XCD-aware work distribution: the queue is cut into 8 contiguous parts, one per group of workgroups that share an XCD
        // (blockIdx % 8, MI355X_MICROARCH.md: blocks are dealt round-robin over the XCDs). The queues are roughly in image order (the camera
        // kernel emits pixel block by pixel block, shading preserves the order), so a part's rays walk one region of the BVH, which then
        // fits that XCD's 4 MiB L2 instead of all 8 L2s each holding a third of the 11 MB tree. A group that runs dry helps with the next part.
        while (!exhausted) {
          const uint32_t part = (home + parts_done) & (kXcdParts - 1u);
          const uint32_t p_lo = (uint32_t)(((uint64_t)n * part) / kXcdParts), p_hi = (uint32_t)(((uint64_t)n * (part + 1u)) / kXcdParts);
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(work + 32u * part, grain);
          base = __shfl(base, 0);
          if (base < p_hi - p_lo) { lo = p_lo + base; hi = (p_hi - p_lo) - base < grain ? p_hi : lo + grain; break; }
          parts_done++;
          if (parts_done >= kXcdParts) { exhausted = true; lo = hi = 0; }
        }
      }
      if (!exhausted) {
        const uint32_t take = (hi - lo) < n_idle ? (hi - lo) : n_idle;
        const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
        PT_STAT(6, lane == 0 ? 1 : 0);
        if (cur == kIdle && rank < take) {
          PT_STAT(7, 1);
          qidx = lo + rank;
          sp = 0; hit = -1; hu = 0.0f; hv = 0.0f;
          int start_tri;
          cur = lane_ray_begin<ANY>(ts, p, !occluded, qidx, r, &start_tri);
          if (ANY && start_tri >= 0) {   // TravScene::any_list
            const uint4 la = ts.any_list[2 * (size_t)start_tri], lb = ts.any_list[2 * (size_t)start_tri + 1];
            cur = la.x; sp = lb.w;
            stk[0 * kPtBlock + tid] = make_uint2(la.y, 0u); stk[1 * kPtBlock + tid] = make_uint2(la.z, 0u); stk[2 * kPtBlock + tid] = make_uint2(la.w, 0u);
            stk[3 * kPtBlock + tid] = make_uint2(lb.x, 0u); stk[4 * kPtBlock + tid] = make_uint2(lb.y, 0u); stk[5 * kPtBlock + tid] = make_uint2(lb.z, 0u);
          }
          if (cur == kIdle) finish(false);
        }
        lo += take;
      }
    }
    const uint64_t m_node = __ballot(is_node(cur)), m_leaf = __ballot(is_leaf(cur));
    if ((m_node | m_leaf) == 0ull) { if (exhausted) break; else continue; }

    // ---- one step, for the lanes of ONE kind only: the other path is not executed at all this iteration (its lanes
    // wait). Node steps run while they outnumber the waiting triangle tests 2 : 1 (measured best of 1:1 ... 1:8; a leaf
    // holds 1-3 triangles against ~23 node steps per ray, so triangle lanes must not wait for a majority).
    const bool do_node = (uint32_t)__popcll(m_node) * RRT_VOTE_A >= (uint32_t)__popcll(m_leaf) * RRT_VOTE_B;
    PT_STAT(0, lane == 0 ? 1 : 0); PT_STAT(10, cur == kIdle ? 1 : 0);
    if (do_node) {
      PT_STAT(1, lane == 0 ? 1 : 0); PT_STAT(11, is_leaf(cur) ? 1 : 0);
      for (int rep_k = 0; rep_k < RRT_NODE_STEPS; rep_k++) {
        { const bool any_node = __builtin_amdgcn_ballot_w64(is_node(cur)) != 0ull; PT_STAT(3, (lane == 0 && any_node) ? 1 : 0); (void)any_node; }
        if (QUAD) {
          if (is_node(cur)) {
            PT_STAT(2, 1);
            const uint32_t off = cur;
            float4 mnx, mny, mnz, mxx, mxy, mxz; uint4 ids;
            if (off < tl_bytes) {
              const char* lp = reinterpret_cast<const char*>(pt_treelet) + off;
              const uint32_t sw = (off >> 3) & 0x70u;   // (n & 7) << 4
              mnx = *reinterpret_cast<const float4*>(lp + sw); mny = *reinterpret_cast<const float4*>(lp + (sw ^ 16u)); mnz = *reinterpret_cast<const float4*>(lp + (sw ^ 32u));
              mxx = *reinterpret_cast<const float4*>(lp + (sw ^ 48u)); mxy = *reinterpret_cast<const float4*>(lp + (sw ^ 64u)); mxz = *reinterpret_cast<const float4*>(lp + (sw ^ 80u));
              const float4 dd = *reinterpret_cast<const float4*>(lp + (sw ^ 96u));
              ids = make_uint4(__float_as_uint(dd.x), __float_as_uint(dd.y), __float_as_uint(dd.z), __float_as_uint(dd.w));
            } else {
              const char* np = reinterpret_cast<const char*>(ts.quads) + off;
              mnx = *reinterpret_cast<const float4*>(np); mny = *reinterpret_cast<const float4*>(np + 16); mnz = *reinterpret_cast<const float4*>(np + 32);
              mxx = *reinterpret_cast<const float4*>(np + 48); mxy = *reinterpret_cast<const float4*>(np + 64); mxz = *reinterpret_cast<const float4*>(np + 80);
              ids = *reinterpret_cast<const uint4*>(np + 96);
            }
            const QuadStep qs = quad_step_f32(mnx, mny, mnz, mxx, mxy, mxz, ids, r);
            // lane masks: h = hit now (slabs and t_min < t_max), s = slabs hit; a slot behind the one the lane goes to is pushed on its slabs alone
            const uint64_t h0 = __builtin_amdgcn_ballot_w64(qs.t[0] < r.tmax), h1 = __builtin_amdgcn_ballot_w64(qs.t[1] < r.tmax);
            const uint64_t h2 = __builtin_amdgcn_ballot_w64(qs.t[2] < r.tmax), h3 = __builtin_amdgcn_ballot_w64(qs.t[3] < r.tmax);
            const uint64_t s1 = __builtin_amdgcn_ballot_w64(qs.t[1] < Const<float>::inf), s2 = __builtin_amdgcn_ballot_w64(qs.t[2] < Const<float>::inf);
            const uint64_t s3 = __builtin_amdgcn_ballot_w64(qs.t[3] < Const<float>::inf);
            const uint64_t e2 = h0 | h1, e3 = e2 | h2;
            auto push = [&](uint32_t w, float t) {
              const uint2 e = make_uint2(w, __float_as_uint(t));
              if (sp < (uint32_t)kStack) stk[sp * kPtBlock + tid] = e;
              else *reinterpret_cast<uint2*>(ts.overflow + ((size_t)(sp - kStack) * ts.overflow_stride + gtid) * 2) = e;
              sp++;
            };
            if (__builtin_amdgcn_inverse_ballot_w64(s3 & e3)) push(qs.id[3], qs.t[3]);
            if (__builtin_amdgcn_inverse_ballot_w64(s2 & e2)) push(qs.id[2], qs.t[2]);
            if (__builtin_amdgcn_inverse_ballot_w64(s1 & h0)) push(qs.id[1], qs.t[1]);
            uint32_t nxt = __builtin_amdgcn_inverse_ballot_w64(h2) ? qs.id[2] : qs.id[3];
            nxt = __builtin_amdgcn_inverse_ballot_w64(h1) ? qs.id[1] : nxt;
            nxt = __builtin_amdgcn_inverse_ballot_w64(h0) ? qs.id[0] : nxt;
            if (__builtin_amdgcn_inverse_ballot_w64(e3 | h3)) cur = nxt;
            else pop();
          }
        } else
        if (is_node(cur)) {
          PT_STAT(2, 1);
          const uint32_t off = ANY ? (cur & ~63u) : cur;
          float4 a, b, c; uint4 d;
          if (kTl > 0 && off < tl_bytes) {
            const uint32_t x = off ^ ((off >> 4) & 0x30u);   // byte address of word 0's slot
            const char* lp = reinterpret_cast<const char*>(pt_treelet);
            a = *reinterpret_cast<const float4*>(lp + x); b = *reinterpret_cast<const float4*>(lp + (x ^ 16u)); c = *reinterpret_cast<const float4*>(lp + (x ^ 32u));
            const float4 dd = *reinterpret_cast<const float4*>(lp + (x ^ 48u));
            d = make_uint4(__float_as_uint(dd.x), __float_as_uint(dd.y), __float_as_uint(dd.z), 0u);
          } else {
            const char* np = reinterpret_cast<const char*>(ts.pairs) + off;   // 32-bit byte offset from a uniform base: no address arithmetic
            a = *reinterpret_cast<const float4*>(np); b = *reinterpret_cast<const float4*>(np + 16); c = *reinterpret_cast<const float4*>(np + 32);
            d = *reinterpret_cast<const uint4*>(np + 48);
          }
          const PairStep st = pair_step_f32<ANY>(a, b, c, d, r, cur);
          if (st.push_far) {
            const uint2 e = make_uint2(st.id_far, __float_as_uint(st.t_far));
            if (sp < (uint32_t)kStack) stk[sp * kPtBlock + tid] = e;
            else *reinterpret_cast<uint2*>(ts.overflow + ((size_t)(sp - kStack) * ts.overflow_stride + gtid) * 2) = e;
            sp++;
          }
          if (st.go_near) cur = st.id_near;
          else pop();
        }
      }
    } else {
      PT_STAT(4, lane == 0 ? 1 : 0); PT_STAT(12, is_node(cur) ? 1 : 0);
      if (is_leaf(cur)) {
        PT_STAT(5, 1);
        // the whole leaf (1-3 triangles) in one step: fewer scheduling rounds than one triangle per step
        if (leaf_step_f32<ANY, MIXED>(ts, cur, r, &hit, &hu, &hv)) finish(true);
        else pop();
      }
    }
  }
#ifdef RRT_PT_STATS
  for (int i = 0; i < 16; i++) if (st_[i]) atomicAdd(&g_pt_stats[ANY ? 1 : 0][i], st_[i]);
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Camera rays through per-tile sub-trees in LDS ("tile trees").
// The camera rays of a 32 x 32-pixel patch of the image - some 80 000 of them at 256 spp - visit 260-290 DISTINCT interior nodes of config 4's
// 49 000 (tools/tile_subtree_stats.py: an 8 x 8 tile 64-150, 16 x 16 170-205), 33-55 each: their node fetches are the same few hundred lines
// over and over, through the unit the persistent kernel keeps busiest (the vector L1: 39 cycles per 64-lane gather instruction against 8 for
// a ds_read_b128). Per patch the host lists the kTtNodes pair nodes its rays visit most (rrt_impl.hpp build_tile_trees(): a census with a few
// camera rays per pixel; counts only decide WHICH nodes are copied) and stores a local copy of them whose child words say "slot k of this
// copy" (byte offsets below kTtNodes * 64) or "node n of the whole tree" (offsets from there on: TravScene::pairs of this launch is a copy of
// the whole tree behind kTtNodes unused slots, its interior child words shifted alike). Boxes, leaf words and split axes are the tree's own,
// so a ray makes the decisions it makes in k_trace_pt_f32, in the same order, whatever the census chose:
// tests/test_gpu_parity.py::test_tile_trees_change_nothing holds frames bit-identical with and without.
// The queue is NOT reordered. The camera kernel's workgroup is one 8 x 8 tile x 8 samples and pushes its survivors as one contiguous run
// (block_push_range): the runs are recorded ("chunks", indexed by workgroup) and an ITEM of work here is the chunks of two neighbouring
// tiles - 16 x 8 pixels, all samples, ~10 000 rays - walked by one 1 024-thread workgroup with that patch's copy in LDS, lanes refilled from
// the item's chunks as in k_trace_pt_f32. Survivors of stage B (k_raygen_aux2_f32: the few whose auxiliary rays needed tracing) sit behind
// the chunked entries in no tile order; they are walked in ranges of kTtRange with the copy of the top of the tree.
// ------------------------------------------------------------------------------------------------------------
#ifndef RRT_TT_BLOCK
#define RRT_TT_BLOCK 1024
#endif
#ifndef RRT_TT_STACK
#define RRT_TT_STACK 8
#endif
#ifndef RRT_TT_NODES
#define RRT_TT_NODES 232
#endif
#ifndef RRT_TT_WG_PER_CU
#define RRT_TT_WG_PER_CU 2
#endif
#ifndef RRT_TT_REFILL
#define RRT_TT_REFILL 16u
#endif
#ifndef RRT_TT_NODE_STEPS
#define RRT_TT_NODE_STEPS 3
#endif
#ifndef RRT_TT_VOTE_A
#define RRT_TT_VOTE_A 1u
#define RRT_TT_VOTE_B 2u
#endif
// Triangle packets in LDS (the north_star's "BVH sub-trees and triangle packets are staged through LDS"): a build with RRT_TT_TRIS > 0 keeps, behind a patch's
// copy of the tree, the triangles of the leaves its camera rays test most (only leaves that are children of copied nodes: their leaf words live in the copy
// and are rewritten to kLeafBit | kSpecialLeaf | n << 19 | index into the local triangle array; the record's material word - which no triangle test reads -
// carries the triangle's index in the whole array, what the hit record needs). The LDS they take comes out of the stack (RRT_TT_STACK entries per lane).
// Measured: see there; the default build has RRT_TT_TRIS = 0 and none of this code.
#ifndef RRT_TT_TRIS
#define RRT_TT_TRIS 0
#endif
constexpr uint32_t kTtTris = RRT_TT_TRIS;
constexpr int kTtBlock = RRT_TT_BLOCK, kTtStack = RRT_TT_STACK;
constexpr uint32_t kTtNodes = RRT_TT_NODES;      // pair nodes per local copy (15 KB + 64 KB of stacks: two workgroups per CU, 8 waves per SIMD)
// LDS byte address of slot k of a copy = the child word that names it (see the kernel): 16 bytes of padding after every four slots, so that slots
// never overlap and the lanes of one read spread over every bank group. kTtLocalBytes: the first byte offset that means "node of the whole tree".
constexpr uint32_t tt_local_addr(uint32_t k) { return 64u * k + 16u * (k >> 2); }
constexpr uint32_t kTtLocalBytes = (tt_local_addr(kTtNodes - 1u) + 64u + 63u) & ~63u;
constexpr bool tt_local_layout_ok() { for (uint32_t k = 0; k + 1u < kTtNodes; k++) if (tt_local_addr(k + 1u) < tt_local_addr(k) + 64u) return false; return true; }
static_assert(tt_local_layout_ok() && kTtLocalBytes + kTtTris * 48u + (uint32_t)kTtStack * kTtBlock * 8u + 8u <= (160u * 1024u) / RRT_TT_WG_PER_CU - 512u, "tile trees: LDS layout / budget");
constexpr uint32_t kTtMacro = 32u;               // edge of the image patch that shares a copy, in pixels
#ifndef RRT_TT_ITEM_TILES
#define RRT_TT_ITEM_TILES 2
#endif
constexpr uint32_t kTtItemTiles = RRT_TT_ITEM_TILES;            // 8 x 8 tiles per item (one tile row of a patch)
constexpr uint32_t kTtRange = 16384u, kTtRangeChunk = 256u;
struct TileTrees {
  const float4* trees;     // [n_trees + 1][kTtNodes][4]: the local copies, patch by patch (row-major over the image); the last one = top of the tree
  const float4* tris;      // [n_trees + 1][kTtTris][3]: the patches' triangle packets (RRT_TT_TRIS > 0 builds), else null
  const uint2* chunks;     // per camera workgroup of the pass (pixel block x sample group): {first queue entry, entries}
  uint32_t tiles_x, tiles_y;   // 8 x 8 tiles of the pass's pixel grid
  uint32_t groups;         // camera workgroups (sample groups) per tile
  uint32_t mt_x, n_trees;  // patches per image row, number of patches
  PassDesc pd;
};

// between the camera kernel's two stages: the entries so far are the chunked ones
static __global__ void k_tt_snapshot(uint32_t* c) { c[C_TT_DONE] = c[C_ACTIVE]; }

template <bool MIXED = false>
__global__ void __launch_bounds__(kTtBlock) __attribute__((amdgpu_waves_per_eu(RRT_TT_WG_PER_CU * RRT_TT_BLOCK / 256))) k_trace_tiles_f32(TravScene ts, Pools<float> p, const uint32_t* count, uint32_t* work, TileTrees tt,
                                                                                  uint32_t n_lo, uint32_t n_hi) {
  const uint32_t n = *count;
  if (n < n_lo || n >= n_hi) return;
  // The copy comes first in LDS: its byte addresses are the local child words themselves and fit the 16-bit offset field of ds_read_b128.
  // Slot k of the copy lives at tt_local_addr(k) = 64 k + 16 (k >> 2): the four 16-byte words of a node are consecutive (one address register,
  // immediate offsets 0 / 16 / 32 / 48 - no address arithmetic in the node step), and the 16 bytes of padding after every four slots spread the
  // 64 lanes of one read over all 16 bank groups like the XOR swizzle of k_trace_pt_f32's treelet does.
  __shared__ __attribute__((aligned(1024))) float4 tree[kTtLocalBytes / 16];   // (the most aligned LDS object is placed first: offset 0)
  __shared__ float4 ltris[kTtTris > 0 ? kTtTris * 3 : 1];                      // the patch's triangle packet (RRT_TT_TRIS > 0)
  __shared__ uint2 stk[kTtStack * kTtBlock];
  __shared__ uint32_t s_item, s_next;
  constexpr uint32_t tl_bytes = kTtLocalBytes;
  constexpr uint32_t kNone = 0xffffffffu;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t gtid = blockIdx.x * blockDim.x + tid;   // overflow-stack column of this lane
  const uint32_t n_done = min(p.counters[C_TT_DONE], n);
  const uint32_t groups_x = (tt.tiles_x + kTtItemTiles - 1u) / kTtItemTiles;
  const uint32_t n_tile_items = groups_x * tt.tiles_y;
  const uint32_t n_items = n_tile_items + (n - n_done + kTtRange - 1u) / kTtRange;
  const uint32_t home = blockIdx.x & (kXcdParts - 1u);
  uint32_t parts_done = 0;      // thread 0 only
  LaneRay r;
  r.oxy = r.ixy = r.ozz = r.izz = v2f{0.0f, 0.0f};
  r.dx = r.dy = r.dz = r.tmax = r.lx = r.ly = r.lz = 0.0f; r.neg = 0; r.skip_plane = 0xffffffffu;
#ifdef RRT_SLAB_FMA
  r.nox = r.noy = r.noz = 0.0f;
#endif
  uint32_t cur = kIdle, qidx = 0, sp = 0;
  int hit = -1;
  float hu = 0.0f, hv = 0.0f;

  auto finish = [&]() {
    p.hit[qidx] = make_float4(r.tmax, __uint_as_float((uint32_t)hit), hu, hv);
    cur = kIdle;
  };
  auto pop = [&]() {   // as in k_trace_pt_f32
    while (sp > 0) {
      sp--;
      uint2 e = stk[(sp < (uint32_t)kTtStack ? sp : 0u) * kTtBlock + tid];
      asm volatile("" : "+v"(e.x), "+v"(e.y));
      if (__builtin_expect(sp >= (uint32_t)kTtStack, 0)) e = *reinterpret_cast<const uint2*>(ts.overflow + ((size_t)(sp - kTtStack) * ts.overflow_stride + gtid) * 2);
      if (__uint_as_float(e.y) < r.tmax) { cur = e.x; return; }
    }
    finish();
  };

  for (;;) {
    // ---- next item: the items are cut into 8 contiguous parts with their own cursors, the workgroups of an XCD take "their" part first
    // (as the rays of k_trace_pt_f32: neighbouring items walk neighbouring parts of the tree, which then stay in that XCD's L2)
    if (tid == 0) {
      uint32_t item = kNone;
      while (parts_done < kXcdParts) {
        const uint32_t part = (home + parts_done) & (kXcdParts - 1u);
        const uint32_t p_lo = (uint32_t)(((uint64_t)n_items * part) / kXcdParts), p_hi = (uint32_t)(((uint64_t)n_items * (part + 1u)) / kXcdParts);
        const uint32_t k = atomicAdd(work + 32u * part, 1u);
        if (k < p_hi - p_lo) { item = p_lo + k; break; }
        parts_done++;
      }
      s_item = item; s_next = 0u;
    }
    __syncthreads();
    const uint32_t item = s_item;
    if (item == kNone) break;
    uint32_t tree_id = tt.n_trees, n_chunks, chunk0 = 0, r_lo = 0, r_hi = 0;
    const bool tiles = item < n_tile_items;
    if (tiles) {
      const uint32_t ty = item / groups_x, tx0 = (item % groups_x) * kTtItemTiles;
      const uint32_t nt = min(kTtItemTiles, tt.tiles_x - tx0), tile0 = ty * tt.tiles_x + tx0;
      uint32_t px, py;
      pass_pixel(tt.pd, tile0 * (kTileW * kTileH), &px, &py);
      tree_id = (py / kTtMacro) * tt.mt_x + px / kTtMacro;
      n_chunks = nt * tt.groups; chunk0 = tile0 * tt.groups;   // tiles of one tile row are consecutive pixel blocks of the camera kernel's grid
    } else {
      r_lo = n_done + (item - n_tile_items) * kTtRange; r_hi = min(n, r_lo + kTtRange);
      n_chunks = (r_hi - r_lo + kTtRangeChunk - 1u) / kTtRangeChunk;
    }
    {
      const float4* src = tt.trees + (size_t)tree_id * (kTtNodes * 4u);
      for (uint32_t i = tid; i < kTtNodes * 4u; i += kTtBlock) tree[i + (i >> 4)] = src[i];   // word i & 3 of slot i >> 2 at tt_local_addr(i >> 2) / 16 + (i & 3)
      if (kTtTris > 0) {
        const float4* tsrc = tt.tris + (size_t)tree_id * (kTtTris * 3u);
        for (uint32_t i = tid; i < kTtTris * 3u; i += kTtBlock) ltris[i] = tsrc[i];
      }
    }
    __syncthreads();

    uint32_t lo = 0, hi = 0;      // wave-private run of reserved rays
    bool exhausted = false;       // wave-uniform: the item has no chunk left
    while (true) {
      // ---- refill idle lanes from the item's chunks
      const uint64_t idle = __ballot(cur == kIdle);
      const uint32_t n_idle = (uint32_t)__popcll(idle);
      if (!exhausted && (n_idle >= RRT_TT_REFILL)) {
        while (lo == hi && !exhausted) {
          uint32_t k = 0;
          if (lane == 0) k = atomicAdd(&s_next, 1u);
          k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
          if (k >= n_chunks) { exhausted = true; break; }
          if (tiles) { const uint2 c = tt.chunks[chunk0 + k]; lo = c.x; hi = c.x + c.y; }
          else { lo = r_lo + k * kTtRangeChunk; hi = min(r_hi, lo + kTtRangeChunk); }
        }
        if (!exhausted) {
          const uint32_t take = (hi - lo) < n_idle ? (hi - lo) : n_idle;
          const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
          if (cur == kIdle && rank < take) {
            qidx = lo + rank;
            sp = 0; hit = -1; hu = 0.0f; hv = 0.0f;
            int start_tri;
            cur = lane_ray_begin<false>(ts, p, false, qidx, r, &start_tri);
            if (cur == kIdle) finish();
          }
          lo += take;
        }
      }
      const uint64_t m_node = __ballot(is_node(cur)), m_leaf = __ballot(is_leaf(cur));
      if ((m_node | m_leaf) == 0ull) { if (exhausted) break; else continue; }
      const bool do_node = (uint32_t)__popcll(m_node) * RRT_TT_VOTE_A >= (uint32_t)__popcll(m_leaf) * RRT_TT_VOTE_B;
      if (do_node) {
        for (int rep_k = 0; rep_k < RRT_TT_NODE_STEPS; rep_k++) {
          if (is_node(cur)) {
            const uint32_t off = cur;
            float4 a, b, c; uint4 d;
            if (off < tl_bytes) {
              const char* lp = reinterpret_cast<const char*>(tree) + off;   // a local child word IS the node's LDS address
              a = *reinterpret_cast<const float4*>(lp); b = *reinterpret_cast<const float4*>(lp + 16); c = *reinterpret_cast<const float4*>(lp + 32);
              const float4 dd = *reinterpret_cast<const float4*>(lp + 48);
              d = make_uint4(__float_as_uint(dd.x), __float_as_uint(dd.y), __float_as_uint(dd.z), 0u);
            } else {
              const char* np = reinterpret_cast<const char*>(ts.pairs) + off;
              a = *reinterpret_cast<const float4*>(np); b = *reinterpret_cast<const float4*>(np + 16); c = *reinterpret_cast<const float4*>(np + 32);
              d = *reinterpret_cast<const uint4*>(np + 48);
            }
            const PairStep st = pair_step_f32<false>(a, b, c, d, r, cur);
            if (st.push_far) {
              const uint2 e = make_uint2(st.id_far, __float_as_uint(st.t_far));
              if (sp < (uint32_t)kTtStack) stk[sp * kTtBlock + tid] = e;
              else *reinterpret_cast<uint2*>(ts.overflow + ((size_t)(sp - kTtStack) * ts.overflow_stride + gtid) * 2) = e;
              sp++;
            }
            if (st.go_near) cur = st.id_near;
            else pop();
          }
        }
      } else {
        if (is_leaf(cur)) {
          if (kTtTris > 0 && (cur & kSpecialLeaf) != 0u) {   // a leaf of the patch's triangle packet: the same tests on the LDS copy of the same 48 bytes
            uint32_t lf = cur & 0x7ffffu, ln = (cur >> 19) & kLeafCountMask;
            do {
              const float4 a = ltris[3u * lf], b = ltris[3u * lf + 1u], c = ltris[3u * lf + 2u];
              float t, u, v;
              if (tri_test_vals_f32<false>(F4{a.x, a.y, a.z, a.w}, F4{b.x, b.y, b.z, b.w}, F4{c.x, c.y, c.z, c.w}, r, &t, &u, &v)) {
                r.tmax = t; hit = (int)__float_as_uint(c.y); hu = u; hv = v;   // (the material word of the copy = the triangle's index in the whole array)
              }
              lf++; ln--;
            } while (ln != 0);
            pop();
          }
          else if (leaf_step_f32<false, MIXED>(ts, cur, r, &hit, &hu, &hv)) finish();
          else pop();
        }
      }
    }
    __syncthreads();   // every wave is done with this item's copy before the next one is loaded
  }
}

// ------------------------------------------------------------------------------------------------------------
// Shadow rays towards delta lights by candidate lists (no tree walk).
// An occlusion query is order independent, and a leaf's box test implies its ancestors' (their boxes contain it and every step of the slab
// arithmetic is monotone under rounding), so BVHAccel::intersect_p's verdict is: "some triangle of some leaf whose OWN box the ray passes is
// hit" - the tree only finds those leaves. A pool shadow ray starts on a known triangle T, is kShadowTmax long (Q9) and points at a light
// whose position (point lights: all at the world origin, Q17) or direction (distant lights) is fixed: the leaves such a ray can reach at all
// are few and can be listed per (light, T) at scene load - the host sweeps T towards the light over the ray's length, fattens that prism by
// the spread of directions over T and by the fp32 slack, and collects every leaf whose box meets it (rrt_impl.hpp build_shadow_lists()).
// Here a ray runs down its list: the reference's own slab test on each leaf's box (same function, same box), the reference's triangle tests on
// the leaves that pass. Same boxes tested as the walk would test at the leaves, less the ones that cannot pass; same verdict, bit for bit
// (tests/test_gpu_parity.py::test_shadow_candidate_lists_change_nothing). 17.8 pair-node steps per shadow ray become ~8 leaf-box tests.
// A triangle without a list (light closer than a few triangle sizes, more than kShadowListMax candidates) walks the tree right here.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kShadowListMax = 48u;        // candidates per (light table, triangle); 0xff in the header = no list: walk the tree
constexpr uint32_t kShadowTabShift = 24u;       // a pool shadow ray's start word: triangle (24 bits) | light table + 1 (bits 24-27)
struct LeafRec { float bmin[3]; uint32_t word; float bmax[3]; uint32_t pad; };   // a BVH leaf: its (fp32, outward) box and its leaf word
struct ShadowLists {
  const uint32_t* headers;   // [table][triangle]: (first entry / 4) << 8 | count (0xff: none)
  const uint32_t* entries;   // leaf ids
  const LeafRec* leaves;
  uint32_t n_tris, n_tables;
};

// the pair-node walk of k_trace_pairs_f32<true> with a private stack (rare path of k_shadow_lists_f32)
RRT_DEV bool walk_any_private_f32(const TravScene& ts, LaneRay& r) {
  float tmin;
  if (!(ts.n_nodes != 0 && box_slabs_f32(ts.root_box[0], ts.root_box[1], ts.root_box[2], ts.root_box[3], ts.root_box[4], ts.root_box[5], r, &tmin) && tmin < r.tmax)) return false;
  uint32_t stack[64];
  uint32_t sp = 0, cur = ts.root_id;
  int hit; float hu, hv;
  for (;;) {
    if (is_node(cur)) {
      const char* np = reinterpret_cast<const char*>(ts.pairs) + cur;
      const float4 a = *reinterpret_cast<const float4*>(np), b = *reinterpret_cast<const float4*>(np + 16), c = *reinterpret_cast<const float4*>(np + 32);
      const uint4 d = *reinterpret_cast<const uint4*>(np + 48);
      float t0, t1;
      const bool h0 = box_slabs_f32(a.x, a.y, c.x, a.z, a.w, c.y, r, &t0) && t0 < r.tmax;
      const bool h1 = box_slabs_f32(b.x, b.y, c.z, b.z, b.w, c.w, r, &t1) && t1 < r.tmax;
      if (h0 && h1) { stack[sp++] = d.y; cur = d.x; }
      else if (h0) cur = d.x;
      else if (h1) cur = d.y;
      else { if (sp == 0) return false; cur = stack[--sp]; }
    } else {
      if (leaf_step_f32<true, false>(ts, cur, r, &hit, &hu, &hv)) return true;
      if (sp == 0) return false;
      cur = stack[--sp];
    }
  }
}

#ifndef RRT_SL_UNROLL
#define RRT_SL_UNROLL 4
#endif
#ifndef RRT_SL_BLOCK
#define RRT_SL_BLOCK 256
#endif
#ifndef RRT_SL_MERGE
#define RRT_SL_MERGE 1
#endif
#ifndef RRT_SL_SORT
#define RRT_SL_SORT 0   // measured and off: see the kernel
#endif
constexpr int kSlBlock = RRT_SL_BLOCK;
static __global__ void __launch_bounds__(kSlBlock) k_shadow_lists_f32(TravScene ts, ShadowLists sl, Pools<float> p, const uint32_t* count) {
  const uint32_t n = *count;
#if RRT_SL_SORT
  // MEASURED AND OFF (round 4). A wave lasts as long as its longest list (0 .. 48 candidates, 11.5 on average on config 4), so this variant deals the workgroup's
  // kSlBlock consecutive rays to its waves BY LIST LENGTH - a counting sort of the ray indices over the list length through LDS (rays without a list sort behind every
  // list). Same verdicts (the list tests pass with it on); any-hit alone 3.12 -> 3.47 ms at 256 threads, 3.9 ms at 512 / 1 024: neighbouring lanes no longer serve
  // neighbouring rays, whose candidate lists and leaf records are mostly the same lines - the coherence the queue order gives is worth more than even list lengths.
  __shared__ uint32_t s_bin[64], s_order[kSlBlock];
  const uint32_t tid = threadIdx.x;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {   // (block-uniform trip count: barriers inside)
    const uint32_t i0 = base + tid;
    uint32_t key = 63u;   // no ray
    if (i0 < n) {
      const uint32_t sw0 = __float_as_uint(p.sray_d[i0].w);
      const uint32_t sk0 = sw0 & ((1u << kShadowTabShift) - 1u), tab0 = (int32_t)sw0 >= 0 ? (sw0 >> kShadowTabShift) & 15u : 0u;
      uint32_t h0 = 0xffu;
      if (tab0 != 0u && tab0 <= sl.n_tables && sk0 < sl.n_tris) h0 = sl.headers[(size_t)(tab0 - 1u) * sl.n_tris + sk0];
      key = (h0 & 0xffu) == 0xffu ? 62u : min(h0 & 0xffu, 61u);
    }
    if (tid < 64u) s_bin[tid] = 0u;
    __syncthreads();
    const uint32_t in_bin = atomicAdd(&s_bin[key], 1u);
    __syncthreads();
    if (tid < 64u) {   // exclusive prefix over the 64 bins, one wave
      const uint32_t v = s_bin[tid];
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if ((int)tid >= d) incl += up; }
      s_bin[tid] = incl - v;
    }
    __syncthreads();
    s_order[s_bin[key] + in_bin] = i0;
    __syncthreads();
    const uint32_t i = s_order[tid];
    __syncthreads();   // (s_bin / s_order are rewritten by the next trip)
    if (i >= n) continue;
    const float4 ro = p.sray_o[i], rd = p.sray_d[i];
#else
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 ro = p.sray_o[i], rd = p.sray_d[i];
#endif
    LaneRay r;
    V3<float> lo;
    ray_tail(ro, kShadowTmax, &r.tmax, &lo);
    const uint32_t sw = __float_as_uint(rd.w);
    const bool has_start = (int32_t)sw >= 0;
    const uint32_t sk = sw & ((1u << kShadowTabShift) - 1u), tab = has_start ? (sw >> kShadowTabShift) & 15u : 0u;
    r.oxy = v2f{ro.x, ro.y}; r.ozz = v2f{ro.z, ro.z}; r.dx = rd.x; r.dy = rd.y; r.dz = rd.z;
    r.lx = lo.x; r.ly = lo.y; r.lz = lo.z;
    r.skip_plane = has_start ? __float_as_uint(ts.tris[(size_t)sk * 12 + 11]) : 0xffffffffu;
    lane_ray_set_inv(r);
    uint32_t hdr = 0xffu;
    if (tab != 0u && tab <= sl.n_tables && sk < sl.n_tris) hdr = sl.headers[(size_t)(tab - 1u) * sl.n_tris + sk];
    bool found = false;
    if ((hdr & 0xffu) == 0xffu) found = walk_any_private_f32(ts, r);   // no list: the tree walk, here
    else {
      // four candidates per round: their ids are one aligned 16-byte load (the host pads a list to a multiple of four with id 0xffffffff), their
      // eight box words are in flight together, and only then the slab tests - a lane's rounds are a chain of dependent loads otherwise
      const uint32_t cnt = hdr & 0xffu;
      const uint4* e4 = reinterpret_cast<const uint4*>(sl.entries) + (hdr >> 8);
      constexpr int U = RRT_SL_UNROLL;   // candidates per round (a multiple of 4)
      for (uint32_t k = 0; k < cnt && !found; k += (uint32_t)U) {
        uint32_t id[U];
#pragma unroll
        for (int q = 0; q < U / 4; q++) {
          const uint4 ids = (k + 4u * (uint32_t)q < cnt) ? e4[(k >> 2) + (uint32_t)q] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
          id[4 * q] = ids.x; id[4 * q + 1] = ids.y; id[4 * q + 2] = ids.z; id[4 * q + 3] = ids.w;
        }
        float4 a[U], b[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
          const float4* lp = reinterpret_cast<const float4*>(sl.leaves + (id[j] != 0xffffffffu ? id[j] : 0u));
          a[j] = lp[0]; b[j] = lp[1];   // {bmin.xyz, word}, {bmax.xyz, -}
        }
#if RRT_SL_MERGE
        // All U box tests first, then the leaves that passed, ONE PER LANE AND STEP: with `if (box j passes) test leaf j` per candidate slot every slot's triangle
        // tests ran as their own divergent block - U of them per round, each with the few lanes whose j-th candidate passed (a sixth of the candidates pass on
        // config 4: some lane of 64 nearly always does) - where max over the lanes of the number of passing boxes, mostly 1 or 2, steps do. Same boxes, same
        // leaves, same triangle tests per ray (an occlusion query does not depend on their order; a lane still stops at its first hit).
        uint32_t pass = 0u;
#pragma unroll
        for (int j = 0; j < U; j++) {
          float tmin;
          const bool ok = id[j] != 0xffffffffu && box_slabs_f32(a[j].x, a[j].y, a[j].z, b[j].x, b[j].y, b[j].z, r, &tmin) && tmin < r.tmax;
          pass |= ok ? (1u << j) : 0u;
        }
        while (__ballot(pass != 0u) != 0ull) {
          if (pass != 0u) {
            const uint32_t j = (uint32_t)__ffs((int)pass) - 1u;
            uint32_t word = __float_as_uint(a[0].w);
#pragma unroll
            for (int q = 1; q < U; q++) word = j == (uint32_t)q ? __float_as_uint(a[q].w) : word;
            int hit; float hu, hv;
            found = leaf_step_f32<true, false>(ts, word, r, &hit, &hu, &hv);
            pass = found ? 0u : (pass & (pass - 1u));
          }
        }
#else
#pragma unroll
        for (int j = 0; j < U; j++) {
          float tmin;
          if (!found && id[j] != 0xffffffffu && box_slabs_f32(a[j].x, a[j].y, a[j].z, b[j].x, b[j].y, b[j].z, r, &tmin) && tmin < r.tmax) {
            int hit; float hu, hv;
            found = leaf_step_f32<true, false>(ts, __float_as_uint(a[j].w), r, &hit, &hu, &hv);
          }
        }
#endif
      }
    }
    if (!found) add_pending(p, i);
  }
}

}  // namespace rrtd

// ------------------------------------------------------------------------------------------------------------
// Camera ray generation of the fp32 product path: shared pieces.
// ------------------------------------------------------------------------------------------------------------
namespace rrtd {

struct RgLane {
  V3<float> o, d;      // ray in lens space
  float element_z;
  int i;               // next interface to process (counts down), -1 = through
  int phase;           // 0 main, 1 x+0.05, 2 x-0.05, 3 y+0.05, 4 y-0.05
};

// per pixel of the pass: Halton pixel offset (halton.rs:75-105) and the packed pixel coordinates - no division by runtime values per sample
static __global__ void __launch_bounds__(kBlock) k_pixel_offsets(SceneDev<float> s, Pools<float> p, PassDesc pd) {
  const uint32_t pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= pd.npix) return;
  uint32_t px, py;
  pass_pixel(pd, pd.pix_begin + pl, &px, &py);
  p.pix_off[2 * pl] = halton_pixel_offset(s, px, py);
  p.pix_off[2 * pl + 1] = (py << 16) | px;
}

}  // namespace rrtd

// ------------------------------------------------------------------------------------------------------------
// Lean lens arithmetic of the fp32 product path (the f64 parity mode keeps the reference's operation order).
// On a spherical interface |p_hit - centre| = |radius|, so the normal is (p_hit - centre) / |radius| instead of a
// normalisation; the ray direction is re-normalised with one v_rsq of the d.d the quadratic needs anyway instead of
// Ray::new's repeated normalisations; cos_i of refract() is the dot product faceforward() already formed; the
// reference's early returns become one accumulated `ok` flag (straight-line code: a dead lane computes garbage that
// nothing reads). Per interface: 2 v_sqrt + 3 v_rcp / v_rsq and ~75 VALU slots instead of 3 sqrt + 8 divisions and
// ~176 with a branch per rejection. Differences to the reference-order fp32 evaluation are of the size of its own
// rounding (1e-7 relative per interface); tests/test_gpu_parity.py::test_camera_samples holds both to the f64 oracle.
// ------------------------------------------------------------------------------------------------------------
namespace rrtd {

struct RgLensLds { float4 a[32]; float2 b[32]; };   // a = {curvature_radius, thickness, eta_i / eta_t, aperture_radius^2}, b = {1 / |curvature_radius|, curvature_radius^2}

RRT_DEV void rg_lens_to_lds(const SceneDev<float>& s, RgLensLds* L, uint32_t tid) {
  if (tid < (uint32_t)s.n_lens) {
    const LensElem<float> e = s.lens[tid];
    const float eta_prev = tid > 0 ? s.lens[tid - 1].eta : 0.0f;
    const float eta_t = (tid > 0 && eta_prev != 0.0f) ? eta_prev : 1.0f;   // camera.rs:196-201
    L->a[tid] = make_float4(e.curvature_radius, e.thickness, e.eta / eta_t, e.aperture_radius * e.aperture_radius);
    L->b[tid] = make_float2(e.curvature_radius != 0.0f ? 1.0f / fabsf(e.curvature_radius) : 0.0f, e.curvature_radius * e.curvature_radius);
  }
}

// generate_ray up to trace_lenses_from_film (camera.rs:534-556), lean form; returns the sample weight (cos^4 term)
RRT_DEV float rg_begin_lean(const SceneDev<float>& s, float pfx, float pfy, float lx, float ly, RgLane* L) {
  const float sx = pfx / (float)s.xres, sy = pfy / (float)s.yres;
  const float p2x = s.extent[0] * (1.0f - sx) + s.extent[2] * sx, p2y = s.extent[1] * (1.0f - sy) + s.extent[3] * sy;
  const float fx = -p2x, fy = p2y;
  const float r2 = fx * fx + fy * fy;
  const float r_film = __builtin_amdgcn_sqrtf(r2);
  const float* pb = (r_film / (s.diagonal / 2.0f) >= 1.0f) ? s.pupil63 : s.pupil0;   // Q6
  // (Measured and reverted in round 4: choosing the box with four selects on the kernel arguments instead of this per-lane pointer - which compiles to vector loads of the
  // argument block - saves ~10 instructions, but the compiler then contracts `pb[0] * (1 - lx) + pb[2] * lx` differently, the camera rays move in their last bit, every
  // sphere coin of DESIGN.md section 4 is tossed again and test_fp32_spheres_at_a_converged_sample_count[matte] lands at 1.037 against its 1.03: the bar sits on the bias itself.)
  const float plx = pb[0] * (1.0f - lx) + pb[2] * lx, ply = pb[1] * (1.0f - ly) + pb[3] * ly;
  const float inv_r = __builtin_amdgcn_rcpf(r_film);
  const float sin_t = r_film != 0.0f ? fy * inv_r : 0.0f, cos_t = r_film != 0.0f ? fx * inv_r : 1.0f;
  const float area = (pb[2] - pb[0]) * (pb[3] - pb[1]);
  const float rear_z = s.lens[s.n_lens - 1].thickness;
  const V3<float> dir(cos_t * plx - sin_t * ply - fx, sin_t * plx + cos_t * ply - fy, rear_z);
  const float il = __builtin_amdgcn_rsqf(len2(dir));
  const V3<float> d = dir * il;
  const float cos4 = (d.z * d.z) * (d.z * d.z);
  // flip_z: (x, y, -z); the film point has z = 0
  L->o = V3<float>(fx, fy, 0.0f); L->d = V3<float>(d.x, d.y, -d.z); L->element_z = 0.0f; L->i = s.n_lens - 1;
  if (s.simple_weighting) return cos4 * area / ((s.pupil0[2] - s.pupil0[0]) * (s.pupil0[3] - s.pupil0[1]));
  return (s.shutter_close - s.shutter_open) * (cos4 * area) / rear_z * rear_z;
}

// One interface of trace_lenses_from_film (camera.rs:163-211) for an interface index that is uniform across the wave (the dense
// kernels step all lanes together): the element is a scalar operand and stop-vs-sphere is a scalar branch. Returns false = blocked.
// `safe` (optional): cleared unless this interface is passed with the margin within which an auxiliary ray (0.05 px beside this one)
// cannot be blocked either: radius below aperture - 16 c_i m (safe_lim = {aperture, 16 c_i}, m = the sample's displacement scale),
// parameter t above 4 x that margin, away from grazing incidence and from total internal reflection (rrt_impl.hpp
// calibrate_aux_margins()).
RRT_DEV bool rg_step_lean(const float4 el, const float2 el2, RgLane* L, const float2 safe_lim = make_float2(0.0f, 0.0f), float m = 0.0f, bool* safe = nullptr) {
  L->element_z -= el.y;
  const V3<float> o = L->o, d = L->d;
  if (el.x == 0.0f) {   // aperture stop
    const float t = (L->element_z - o.z) * __builtin_amdgcn_rcpf(d.z);
    const V3<float> p_hit = o + d * t;
    L->o = p_hit;
    const float r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
    if (safe) {
      const float margin = safe_lim.y * m, rl = fmaxf(safe_lim.x - margin, 0.0f);
      *safe &= (safe_lim.y > 0.0f) & (d.z < -0.05f) & (t > 4.0f * margin) & (r2 < rl * rl);
    }
    return (d.z < 0.0f) & (t >= 0.0f) & (r2 < el.w);
  }
  // intersect_spherical_element camera.rs:220-253 + quadratic misc.rs:231-251
  const V3<float> oc(o.x, o.y, o.z - (L->element_z + el.x));
  const float a = len2(d);
  const float b = 2.0f * dot(d, oc);
  const float c = len2(oc) - el2.y;
  const float disc = b * b - 4.0f * a * c;
  const float root = __builtin_amdgcn_sqrtf(disc);
  const float q = (b < 0.0f) ? -0.5f * (b - root) : -0.5f * (b + root);
  const float t0 = q * __builtin_amdgcn_rcpf(a), t1 = c * __builtin_amdgcn_rcpf(q);
  const bool use_closer = (d.z > 0.0f) ^ (el.x < 0.0f);
  const float t = use_closer ? fminf(t0, t1) : fmaxf(t0, t1);
  bool ok = (disc >= 0.0f) & (t >= 0.0f);   // (a NaN t fails `t >= 0` like the reference's `t < 0` / assert pair)
  const V3<float> p_hit = o + d * t;
  const float r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
  ok &= r2 < el.w;
  L->o = p_hit;
  V3<float> n = (oc + d * t) * el2.x;
  const V3<float> wi = d * -__builtin_amdgcn_rsqf(a);
  float cos_i = dot(n, wi);
  const float sgn = cos_i < 0.0f ? -1.0f : 1.0f;   // faceforward(n, -ray.d)
  n = n * sgn; cos_i = cos_i * sgn;
  const float eta = el.z;
  const float sin2_t = eta * eta * fmaxf(0.0f, 1.0f - cos_i * cos_i);   // refract reflection.rs:122-134
  ok &= sin2_t < 1.0f;
  if (safe) {
    const float margin = safe_lim.y * m, rl = fmaxf(safe_lim.x - margin, 0.0f);
    *safe &= (safe_lim.y > 0.0f) & (r2 < rl * rl) & (t > 4.0f * margin) & (cos_i > 0.15f) & (sin2_t < 0.98f);
  }
  const float cos_t = __builtin_amdgcn_sqrtf(1.0f - sin2_t);
  L->d = wi * -eta + n * (eta * cos_i - cos_t);
  return ok;
}

}  // namespace rrtd

// ------------------------------------------------------------------------------------------------------------
// Dense two-stage camera ray generation with the lean lens arithmetic (the default fp32 path).
// With ~70 instructions per interface the persistent-thread machinery (ballots, refill, state machine: a fixed
// cost per loop iteration, at ~6 interfaces per sample) costs more than the idle lanes it saves, so:
//   stage A  one thread per sample: get_camerasample + the whole main trace. Lanes die on the way (69 %), the
//            survivors are pushed - one atomic per 1024-thread block - to a staging queue of 48-byte records in
//            queue order {world ray, p_film, p_lens, slot, weight}. Nothing is written for a dead sample.
//   stage B  one thread per survivor: the auxiliary traces (x + 0.05 and y + 0.05 together for every lane, the
//            rare opposite shifts in a second round), then the living samples enter q_active.
// ------------------------------------------------------------------------------------------------------------
namespace rrtd {

#ifndef RRT_RG_DENSE
#define RRT_RG_DENSE 512
#endif
constexpr int kRgDense = RRT_RG_DENSE;
static_assert((kRgDense & (kRgDense - 1)) == 0, "the camera workgroup is split into pixels x samples by shifts");
#ifndef RRT_RG_REPACK
#define RRT_RG_REPACK 3
#endif
constexpr int kRgRepack = RRT_RG_REPACK;   // lens interfaces traced before the block packs its survivors (0 = never)

// lane_ray_begin()'s root test on a camera ray as the queue would hold it (plain fp32 origin, t_max = inf): the same function on the same values, so the
// camera kernel's verdict is the traversal kernels' verdict. A ray that fails it is a miss (kIdle at once, hit record {inf, -1}).
RRT_DEV bool camera_ray_meets_root(const SceneDev<float>& s, const V3<float>& wo, const V3<float>& wd) {
  LaneRay r;
  r.oxy = v2f{wo.x, wo.y}; r.ozz = v2f{wo.z, wo.z}; r.dx = wd.x; r.dy = wd.y; r.dz = wd.z;
  lane_ray_set_inv(r);
  float tmin;
  return box_slabs_f32(s.root_box[0], s.root_box[1], s.root_box[2], s.root_box[3], s.root_box[4], s.root_box[5], r, &tmin) && tmin < Const<float>::inf;
}

// staging records live in the next-queue arrays, which are free until the first shading launch:
//   nray_o[i] = {o.xyz (world), slot}, nray_d[i] = {d.xyz (world), weight}, npath[i] = {p_film.xy, p_lens.xy}, hindex[i] = Halton index
static __global__ void __launch_bounds__(kRgDense) k_raygen_main_f32(SceneDev<float> s, Pools<float> p, PassDesc pd, int write_samp, double* dims_out,
                                                                     const float2* safe_lim, float aux_delta, float aux_pupil, int enqueue, uint32_t spb, uint2* chunks) {
  __shared__ RgLensLds lens;
  __shared__ float2 safe_s[32];
  __shared__ uint32_t push_lds[kRgDense / 64 + 2];
  const uint32_t tid = threadIdx.x;
  rg_lens_to_lds(s, &lens, tid);
  if (tid < (uint32_t)s.n_lens) safe_s[tid] = safe_lim ? safe_lim[tid] : make_float2(0.0f, 0.0f);   // 16 c_i = 0: never safe
  __syncthreads();
  // grid: x = sample of the pass, (y, z) = pixel block - blocks are dispatched x first, so the survivors reach the queue pixel block by pixel
  // block (all samples of 512 neighbouring pixels together): the queue is in image order, which the XCD-aware traversal relies on
  // A block is kRgPix pixels x (kRgDense / kRgPix) consecutive samples (spb: samples per block, a kernel argument: 1 = one sample of kRgDense pixels)
  // (spb is 1, 2, 4 or 8 - rrt_impl.hpp checks -, so the workgroup's split into pixels x samples is shifts and masks, not the divisions by a kernel argument that
  // used to open every thread's prologue with ~40 instructions)
  const uint32_t spb_log2 = (uint32_t)__builtin_ctz(spb), ppb_log2 = (uint32_t)__builtin_ctz((uint32_t)kRgDense) - spb_log2, ppb = 1u << ppb_log2;   // pixels per block
  const uint32_t pl = ((blockIdx.z * gridDim.y + blockIdx.y) << ppb_log2) + (tid & (ppb - 1u)), sl = (blockIdx.x << spb_log2) + (tid >> ppb_log2);
  bool alive = false;
  uint32_t slot = 0, index = 0;
  float pfx = 0, pfy = 0, lx = 0, ly = 0, w = 0;
  RgLane L; L.i = -1; L.phase = 0; L.element_z = 0;
  if (pl < pd.npix && sl < pd.ns) {
    slot = sl * pd.npix + pl;
    // dead samples: weight 0 (Q2), nothing else is written for them. Every sample's first owner writes it here (coalesced, one store instead of a 1 GB
    // memset per frame in front of the kernel); a survivor's weight is stored at the end of this kernel - behind the block's barriers, by whichever thread of
    // the block holds the sample after the re-pack - or by stage B.
    p.weight[slot] = 0.0f;
    const uint2 po = reinterpret_cast<const uint2*>(p.pix_off)[pl];
    const uint32_t px = po.y & 0xffffu, py = po.y >> 16;
    index = po.x + (pd.s_begin + sl) * s.stride;
    double d0, d1, d2, d3;
    halton_cam4(s, index, &d0, &d1, &d2, &d3);
    pfx = (float)px + to_real<float>(d0); pfy = (float)py + to_real<float>(d1);
    lx = to_real<float>(d2) + 0.5f; ly = to_real<float>(d3) + 0.5f;   // Q5
    if (write_samp) p.samp[slot] = make_float4(pfx, pfy, lx, ly);
    if (dims_out) { double* dd = dims_out + 5 * (size_t)(pl * pd.ns + sl); dd[0] = d0; dd[1] = d1; dd[2] = d2; dd[3] = d3; dd[4] = halton_dim(s, index, 4); }
    w = rg_begin_lean(s, pfx, pfy, lx, ly, &L);
    alive = w != 0.0f;
  }
  // the sample's displacement scale m = delta (1 + P / r_film) (calibrate_aux_margins()); the film point is the lens-space origin
  const float r_film = __builtin_amdgcn_sqrtf(L.o.x * L.o.x + L.o.y * L.o.y);
  float m_scale = aux_delta * (1.0f + aux_pupil * __builtin_amdgcn_rcpf(r_film));
  bool safe = safe_lim != nullptr && r_film > 0.0f;
  // The first interfaces stop most of the doomed samples (29 % at the second, 16 % at the third on the scene.json lens): after
  // kRgRepack of them the block's survivors are packed into its first threads through LDS, so that the remaining interfaces run in
  // full waves and the emptied waves skip them.
  const int k_pack = (kRgRepack > 0 && s.n_lens - 1 - kRgRepack >= 0) ? s.n_lens - 1 - kRgRepack : -1;   // first interface after the re-pack; block-uniform
  for (int k = s.n_lens - 1; k > k_pack; k--) {   // (no lane-divergent branch inside: dead lanes ride along)
    if (__ballot(alive) == 0ull) break;
    alive &= rg_step_lean(lens.a[k], lens.b[k], &L, safe_s[k], m_scale, &safe);
  }
  if (k_pack >= 0) {
    int k = k_pack;
    __shared__ float xf[13][kRgDense];
    __shared__ uint32_t xu[2][kRgDense];
    uint32_t n_alive = 0;
    const uint32_t at = block_rank(alive, push_lds, &n_alive);
    if (alive) {
      xf[0][at] = L.o.x; xf[1][at] = L.o.y; xf[2][at] = L.o.z; xf[3][at] = L.d.x; xf[4][at] = L.d.y; xf[5][at] = L.d.z;
      xf[6][at] = pfx; xf[7][at] = pfy; xf[8][at] = lx; xf[9][at] = ly; xf[10][at] = w; xf[11][at] = m_scale; xf[12][at] = safe ? 1.0f : 0.0f;
      xu[0][at] = slot; xu[1][at] = index;
    }
    __syncthreads();
    alive = tid < n_alive;
    if (alive) {
      L.o = V3<float>(xf[0][tid], xf[1][tid], xf[2][tid]); L.d = V3<float>(xf[3][tid], xf[4][tid], xf[5][tid]);
      pfx = xf[6][tid]; pfy = xf[7][tid]; lx = xf[8][tid]; ly = xf[9][tid]; w = xf[10][tid]; m_scale = xf[11][tid]; safe = xf[12][tid] != 0.0f;
      slot = xu[0][tid]; index = xu[1][tid];
    }
    // element_z is the same for every lane at a given interface: -(sum of the thicknesses passed so far)
    float ez = 0.0f;
    for (int j = s.n_lens - 1; j > k; j--) ez -= lens.a[j].y;
    L.element_z = ez;
    for (; k >= 0; k--) {
      if (__ballot(alive) == 0ull) break;
      alive &= rg_step_lean(lens.a[k], lens.b[k], &L, safe_s[k], m_scale, &safe);
    }
  }
  // survivors whose auxiliary rays cannot be blocked are done: straight to q_active; the others wait in the staging queue for stage B
  const bool done = alive & safe, staged = alive & !safe;
  // ray out of the lens = flip_z, camera_to_world, normalise (camera.rs:558-565)
  V3<float> wo, wd;
  bool meets = true;
  if (alive) {
    wo = aff_pt(s.cam_m, V3<float>(L.o.x, L.o.y, -L.o.z));
    const V3<float> wdu = aff_vec(s.cam_m, V3<float>(L.d.x, L.d.y, -L.d.z));
    wd = wdu * __builtin_amdgcn_rsqf(len2(wdu));
    if (s.root_cull && done) meets = camera_ray_meets_root(s, wo, wd);
  }
  uint32_t qa_first, qa_total;
  const uint32_t qa = block_push_range(&p.counters[C_ACTIVE], done && enqueue && meets, push_lds, &qa_first, &qa_total);
  if (s.root_cull) (void)block_push(&p.counters[C_CULLED], done && enqueue && !meets, push_lds);   // block-uniform condition
  // chunk record of this workgroup (k_trace_tiles_f32): its entries of the first queue are one contiguous run, all from one tile of the image
  if (chunks && tid == 0) chunks[(size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = make_uint2(qa_first, qa_total);
  const uint32_t qs = block_push(&p.counters[C_NEXT], staged, push_lds);
  (void)block_push(&p.counters[C_CAMERA_RAYS], done, push_lds);
  if (alive) {
    if (staged) {
      p.nray_o[qs] = make_float4(wo.x, wo.y, wo.z, __uint_as_float(slot));
      p.nray_d[qs] = make_float4(wd.x, wd.y, wd.z, w);
      p.npath[qs] = make_float4(pfx, pfy, lx, ly);
      p.hindex[qs] = index;
    } else {
      if (enqueue && meets) {
        p.q_active[qa] = QEnt{slot, 5u, index, 0u};          // five camera dimensions consumed, bounce 0
        if (s.integrator != 0 /* RRT_INT_PATH */) p.path[qa] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);    // beta, eta_scale (k_shade_path knows a camera ray's without reading it)
        p.ray_o[qa] = make_float4(wo.x, wo.y, wo.z, Const<float>::inf);
        p.ray_d[qa] = make_float4(wd.x, wd.y, wd.z, __uint_as_float(0xffffffffu));
      }
      p.L[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      p.weight[slot] = w;
    }
  }
}

// one whole lens trace, lean; `out`: the ray itself, world space (textured scenes keep the auxiliary rays)
RRT_DEV bool rg_trace_lean(const SceneDev<float>& s, const RgLensLds& lens, float pfx, float pfy, float lx, float ly, bool want_ray, RayT<float>& out) {
  RgLane L; L.phase = 0;
  const float w = rg_begin_lean(s, pfx, pfy, lx, ly, &L);
  bool ok = w != 0.0f;
  for (int k = s.n_lens - 1; k >= 0; k--) ok &= rg_step_lean(lens.a[k], lens.b[k], &L);
  if (ok && want_ray) {
    out.o = aff_pt(s.cam_m, V3<float>(L.o.x, L.o.y, -L.o.z));
    const V3<float> wdu = aff_vec(s.cam_m, V3<float>(L.d.x, L.d.y, -L.d.z));
    out.d = wdu * __builtin_amdgcn_rsqf(len2(wdu));
  }
  return ok;
}

static __global__ void __launch_bounds__(kRgDense) k_raygen_aux2_f32(SceneDev<float> s, Pools<float> p, int enqueue) {
  __shared__ RgLensLds lens;
  __shared__ uint32_t push_lds[kRgDense / 64 + 1];
  const uint32_t tid = threadIdx.x;
  const uint32_t n = p.counters[C_NEXT];
  if (blockIdx.x * blockDim.x >= n) return;   // the grid is sized for the worst case
  rg_lens_to_lds(s, &lens, tid);
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + tid;
  const bool diffs = p.rdx_o != nullptr;   // block-uniform
  bool alive = false;
  uint32_t slot = 0;
  float4 ro = make_float4(0, 0, 0, 0), rd = ro, cs = ro;
  if (i < n) {
    ro = p.nray_o[i]; rd = p.nray_d[i];
    slot = __float_as_uint(ro.w);
    cs = p.npath[i];
  }
  // Four rounds with ONE inlined trace: x + 0.05 and y + 0.05 for every lane (nearly every survivor passes both), then the rare
  // opposite shifts for the lanes that need them (a wave skips a round none of its lanes needs).
  RayT<float> aux, auy;
  float epsx = 0.05f, epsy = 0.05f;
  bool okx = false, oky = false;
  for (int round = 0; round < 4; round++) {
    const bool is_x = (round & 1) == 0;
    const bool act = i < n && (round < 2 || (round == 2 ? !okx : (okx && !oky)));
    if (__ballot(act) == 0ull) continue;
    const float e = round < 2 ? 0.05f : -0.05f;
    RayT<float> out;
    bool ok = false;
    if (act) ok = rg_trace_lean(s, lens, cs.x + (is_x ? e : 0.0f), cs.y + (is_x ? 0.0f : e), cs.z, cs.w, diffs, out);
    if (act && is_x) { okx = ok; epsx = e; if (diffs) aux = out; }
    if (act && !is_x) { oky = ok; epsy = e; if (diffs) auy = out; }
  }
  alive = okx && oky;
  if (i < n) {
    if (alive && diffs) {   // rx / ry (camera.rs:597-598, 613-614), then scale_differentials (geometry.rs:1883-1888)
      const V3<float> o(ro.x, ro.y, ro.z), d(rd.x, rd.y, rd.z);
      V3<float> rxo = o + (aux.o - o) / epsx, rxd = d + (aux.d - d) / epsx;
      V3<float> ryo = o + (auy.o - o) / epsy, ryd = d + (auy.d - d) / epsy;
      rxo = o + (rxo - o) * s.diff_scale; ryo = o + (ryo - o) * s.diff_scale;
      rxd = d + (rxd - d) * s.diff_scale; ryd = d + (ryd - d) * s.diff_scale;
      p.rdx_o[slot] = make_float4(rxo.x, rxo.y, rxo.z, 0.0f); p.rdx_d[slot] = make_float4(rxd.x, rxd.y, rxd.z, 0.0f);
      p.rdy_o[slot] = make_float4(ryo.x, ryo.y, ryo.z, 0.0f); p.rdy_d[slot] = make_float4(ryd.x, ryd.y, ryd.z, 0.0f);
    }
  }
  const bool meets = !(s.root_cull && alive) || camera_ray_meets_root(s, V3<float>(ro.x, ro.y, ro.z), V3<float>(rd.x, rd.y, rd.z));
  const bool enq = alive && enqueue && meets;
  const uint32_t q = block_push(&p.counters[C_ACTIVE], enq, push_lds);
  if (s.root_cull) (void)block_push(&p.counters[C_CULLED], alive && enqueue && !meets, push_lds);
  (void)block_push(&p.counters[C_CAMERA_RAYS], alive, push_lds);
  if (enq) {
    p.q_active[q] = QEnt{slot, 5u, p.hindex[i], 0u};   // five camera dimensions consumed, bounce 0
    if (s.integrator != 0 /* RRT_INT_PATH */) p.path[q] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);   // beta, eta_scale
    p.ray_o[q] = make_float4(ro.x, ro.y, ro.z, Const<float>::inf);
    p.ray_d[q] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(0xffffffffu));
  }
  if (alive) { p.L[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); p.weight[slot] = rd.w; }
}

}  // namespace rrtd
