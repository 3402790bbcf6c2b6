// Device-side data layout (HBM, SoA) for the wavefront path tracer. gfx950 only.
//
// Everything the kernels read is flattened to world space and stored in *traversal order* (ordered_prims
// of bvh.rs:336-357), so a leaf's primitives are contiguous: nodes[i].offset indexes tri arrays directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "horizon_build.hpp"   // hz_sector(): shared with the host builder
namespace rrtd {
// LinearBVHNode bvh.rs:103-109, narrowed: f32 = 32 B (bounds rounded outward from the f64 build), f64 = 64 B.
// Block tables of a sampler dimension's digit loop (fp32 mode): the sample index is split as hi * block + lo, block = base^low_digits;
// lo[lo_off + lo] = the permuted reversal of exactly low_digits digits, hi[hi_off + hi] = {permuted reversal of hi's digits, base^(digits of hi),
// the two words of the f64 inv_base^(all digits) the loop's running product arrives at}. See scrambled_radical_inverse_tab() in dmath.hpp.
struct HaltonBlk {
  uint32_t block;         // 0 = no table for this dimension
  uint32_t shift;         // l - 1 of the division by `block` (div_base())
  uint32_t magic;         // m'
  uint32_t lo_off, hi_off;
  uint32_t pad[3];
};

template <typename R>
struct alignas(sizeof(R) * 8) Node {
  R bmin[3];
  R bmax[3];
  uint32_t offset;   // leaf: first triangle (traversal order); interior: second child
  uint32_t meta;     // n_primitives << 2 | axis
  // f64: 48 + 8 = 56 -> padded to 64 by alignas
};

// One triangle = 3 world-space vertices + 3 words (48 B in f32). Read by the traversal kernels.
template <typename R>
struct alignas(16) Tri {
  R p0[3], p1[3], p2[3];
  uint32_t material;   // index into materials
  uint32_t shade;      // index into TriShade (0xffffffff: no normals / uvs -> defaults)
  uint32_t plane;      // id shared by exactly coplanar triangles (host, see plane_ids()); used by self_prim()
};

// A sphere primitive of the aggregate occupies one Tri slot (so node.offset still indexes one array in traversal
// order): plane == kSphereMark, shade = index into SceneDev::spheres, material as usual.
constexpr uint32_t kSphereMark = 0xfffffffeu;

// A triangle of a NON-RIGID instance (scale / shear) is not flattened either: the reference transforms the ray into the instance's space,
// re-normalises its direction there and copies the object-space t back to the world ray (TransformedPrimitive::intersect
// primitives.rs:115-139, transform.rs:525-537: Q15), which no world-space triangle reproduces. Such a Tri keeps the mesh's raw vertices
// and carries its instance in the material word: kInstFlag | instance index (15 bits) << 16 | material (16 bits). Its plane id is
// computed from the world-space vertices like everybody's (coplanarity is the same in both spaces).
constexpr uint32_t kInstFlag = 0x80000000u;
template <typename R>
struct InstDev {
  R m[12], mi[12];      // primitive_to_world / its inverse, rows 0..2
  uint32_t identity;    // Transform::is_identity (value compare): the interaction is not transformed then
  uint32_t pad[3];
};

// Sphere (shape/sphere.rs:14-49) + the TransformedPrimitive around it (primitives.rs:100-139). Spheres are not
// flattened: the reference's own sequence of ray transforms is replayed, including its quirks (Q15, Q16).
template <typename R>
struct SphereDev {
  R m[12], mi[12];       // the sphere's obj_to_world / world_to_obj (rows 0..2)
  R im[12], imi[12];     // instance primitive_to_world / its inverse
  R radius, z_min, z_max, theta_min, theta_max, phi_max;
  uint32_t has_inst;     // 0: GeometricPrimitive used directly; 1: wrapped in a TransformedPrimitive
  uint32_t inst_identity;  // TransformedPrimitive::intersect skips the interaction transform for the identity
};

// Optional per-triangle shading attributes (meshes with vn / vt), world space.
template <typename R>
struct TriShade {
  R n[3][3];
  R uv[3][2];
  uint32_t has_n, has_uv;  // mesh_has_* of rrt_tri (0,1,2)
};

template <typename R>
struct Material {
  int32_t type, remap_roughness;
  R kd[3], ks[3], kr[3], eta[3], k[3];
  R sigma, roughness, u_roughness, v_roughness;
  R kt[3], reflect[3], transmit[3], index;   // glass / translucent
  int32_t tex[13];       // RRT_P_* slot -> SceneDev::textures index evaluated at every hit, -1 = the constant above
  int32_t has_tex;       // any slot >= 0
  int32_t bump, pad;     // bump_map texture (Material::bump), -1 = none
};

// one node of the texture graph (rrt_texture, include/rrt.h): float textures carry their value in all three channels
template <typename R>
struct TexDev {
  int32_t type, mapping, aa_none, octaves;
  int32_t child[3], image;   // image: index into SceneDev::images (ImageTexture)
  R fallback[3][3];
  R v[4][3];
  R omega;
  R map[4];
  R vs[3], vt[3];
  R w2t[12];             // world_to_texture rows 0..2 (affine; the loader only composes T * R * S)
};

// MIPMap (rrt_image, include/rrt.h): per level the BlockedArray's data vector as the reference's index expression fills it
struct ImageLevelDev { uint32_t u_res, v_res, u_blocks, n; uint32_t offset, pad[3]; };   // offset / n in texels of SceneDev::image_texels
template <typename R>
struct ImageDev {
  int32_t do_trilinear, wrap, n_levels, pad;
  R max_aniso, pad2;
  ImageLevelDev levels[16];
};

template <typename R>
struct Light {
  int32_t type, shape_type;
  R spectrum[3];
  R p_light[3];
  R area;
  // sphere light shape (object space + transform) / triangle light shape (raw mesh vertices, Q13)
  R m[12], mi[12];                 // obj_to_world rows 0..2 (affine), and inverse
  R radius, z_min, z_max, theta_min, theta_max, phi_max;
  R tp[3][3];                      // triangle vertices
  R tn[3][3];                      // triangle vertex normals (if tri_has_n)
  uint32_t tri_has_n;
  R w_light[3], world_radius;      // DistantLight (lights/distant.rs)
  uint32_t shadow_tab;             // fp32: shadow candidate table of this light + 1 (dtraverse_f32.hpp), 0 = none
};

template <typename R>
struct LensElem { R curvature_radius, thickness, eta, aperture_radius; };

struct HaltonDim {   // one entry per sampler dimension >= 2
  uint32_t base;
  uint32_t perm_offset;   // PRIME_SUMS[dim]
  uint64_t magic;         // m' | (l - 1) << 32 of div_base(): exact a / base for every 32-bit a
  double inv;             // 1 / base: a / base for any 32-bit a = (uint32_t)(a * inv) with a +-1 fix-up (div_base())
  double tail;            // inv * perm[0] / (1 - inv): the infinitely many trailing zero digits of the scrambled radical inverse
};

template <typename R>
struct SceneDev {
  const Node<R>* nodes;
  const Tri<R>* tris;
  const TriShade<R>* shades;
  const SphereDev<R>* spheres;
  const InstDev<R>* insts;     // non-rigid triangle instances (kInstBase)
  const Material<R>* materials;
  const TexDev<R>* textures;   // texture graph nodes (children before parents)
  const ImageDev<R>* images;
  const R* image_texels;       // RGB triples
  R diff_scale;                // scale_differentials factor 1 / sqrt(samples_per_pixel), integrator/mod.rs:94-96
  const Light<R>* lights;
  const R* light_cdf;          // Distribution1D([1; n]).cdf, n_lights + 1 entries
  uint32_t n_nodes, n_tris, n_lights;
  R light_pick_pdf;            // 1 / (func_int * n)
  uint32_t stack_depth;        // >= bvh depth + 1
  uint32_t flags;
  uint32_t use_shadow_tabs;    // the shading kernel stores Light::shadow_tab with a shadow ray's start triangle (the launch that follows is k_shadow_lists_f32)
  // fp32 path integrator: the camera kernels answer a camera ray that misses the root box themselves (the test lane_ray_begin() makes, on the ray as stored:
  // such a ray is a miss, which the path integrator shades with nothing) instead of sending it through the queue - see camera_ray_meets_root()
  uint32_t root_cull;
  // horizon tables (fp32 path integrator, host/horizon_build.cpp build_horizons()): per triangle 2 x 16 bytes - hemisphere +hz_axis / -hz_axis, 16 azimuth sectors (hz_sector) -
  // holding ceil(254 x sin(max elevation at which anything is visible from any point of the triangle in that sector)) + margin; null = off
  const uint8_t* horizon;
  const float* hz_tau;         // per triangle: the cull applies where min(barycentric) |n.d| > hz_tau (HzTables::tau: the ray's origin is only NEAR its triangle's plane)
  uint32_t hz_axis;
  float root_box[6];
  // camera
  const LensElem<R>* lens;
  int32_t n_lens, simple_weighting;
  R cam_m[12];                 // camera_to_world rows 0..2
  R pupil0[4], pupil63[4];
  R shutter_open, shutter_close;
  // film
  int32_t xres, yres;
  R diagonal, extent[4];
  R max_sample_luminance;
  const R* filter_table;       // 16 x 16 (film.rs:163-173, incl. Q4), only read by the wide-filter film kernel
  R filter_rx, filter_ry;
  R filter_inv_rx, filter_inv_ry;   // 1 / radius (film.rs:43-46), divided on the host: the fp32 build's fast reciprocal is 1 ulp off, and
                               // `distance * inv_radius * 16` sits exactly on table-index boundaries for unjittered strata
  // sampler
  const HaltonDim* hdims;
  uint32_t* err;               // the handle's error word (counters[C_ERROR]): device-side counterparts of the reference's panics
  const uint16_t* perms;
  uint32_t nsamp, sample_at_center;
  uint32_t base_exp0, base_exp1, base_scale0, base_scale1, stride, mult_inv0, mult_inv1;
  uint32_t fast_div;           // all sample indices < 2^26
  // StratifiedSampler (samplers/stratified.rs) with counter-based randomness, see draw_1d() in dmath.hpp
  uint32_t sampler_type;       // RRT_SAMPLER_*
  uint32_t st_nx, st_ny, st_jitter, st_dims, st_seed_lo, st_seed_hi;
  uint32_t cam_db;             // dimension-counter word after the camera sample (Halton: 5; stratified: one 1D, two 2D)
  uint32_t shade_compact;      // k_shade_path packs the hits of a chunk of queue entries before shading them (option "shade_compact")
  uint32_t db_shift;           // a path's queue word = dimension counter(s) | bounce << db_shift: 16 under the HaltonSampler (one counter below 1 000, 65 535 bounces), 24 under the StratifiedSampler (two 12-bit counters, 255 bounces); dmath.hpp db_pack
  // camera lens dimensions 2 and 3 (bases 5 and 7): digit permutation packed 3 bits per digit, and the
  // reference's running product inv_base^k (lowdiscrepancy.rs:204-227) tabulated by the same f64 multiplications
  uint32_t cam_perm[2];
  double cam_invpow[2][16];
  double cam_tail[2];          // inv_base * perm[0] / (1 - inv_base)
  double inv_base_scale1;      // 1 / base_scale1
  double inv3pow[24];          // (1/3)^k by the reference's running product (radical_inverse of dimension 1, lowdiscrepancy.rs:188-202)
  // The digit loops of camera dimensions 1-3 as two table look-ups each (fp32 camera kernel; null = loops): the index is split as
  // hi * B + lo, B = 3^6 / 5^6 / 7^5; cam_lo[w][lo] = the permuted reversal of exactly that many low digits, cam_hi[w][hi] = {permuted
  // reversal of hi's digits, base^(digits of hi), inv_base^(all digits) as the two words of the f64 the loop's table holds} - see halton_cam4()
  const uint32_t* cam_lo[3];
  const uint4* cam_hi[3];
  // the same for the sampler dimensions the integrators draw (dimension < n_hblk; null / 0 = loops)
  const HaltonBlk* hblk;
  const uint32_t* hlo;
  const uint4* hhi;
  uint32_t n_hblk;
  // integrator
  int32_t integrator, max_depth, light_strategy;
  R rr_threshold;
};

// Wavefront pools. One slot = one camera sample of the current pass.
//
// Rays, hits and shadow rays are 16-byte-per-word records stored in QUEUE ORDER (index = position in the queue the
// kernel iterates), not by slot, and so is the path state that changes per bounce (beta, sampler dimension, bounce
// count): every bounce's traversal and shading kernels stream them with coalesced 128-bit loads / stores; the only
// per-slot gather left is the radiance update of an unoccluded shadow ray. (The first version kept rays / hits as 4-byte
// SoA arrays indexed by slot through the queue: 8 + 4 scattered dword accesses per ray, each its own 32/64-B
// request - the PMC write traffic of the closest-hit kernel was 7x its algorithmic bytes.)
// Integers ride in the 4th component as raw bits (never touched by arithmetic).
template <typename R> struct Vec4T;
template <> struct Vec4T<float> { using type = float4; };
template <> struct Vec4T<double> { using type = double4; };

struct alignas(16) QEnt { uint32_t slot, db, index, pad; };   // slot, dimension | bounces << 16, Halton global sample index

template <typename R>
struct Pools {
  using V4 = typename Vec4T<R>::type;
  // rays of q_active (also the public rrt_rays batch): {o.x, o.y, o.z, t_max | packed o_lo}, {d.x, d.y, d.z, skip}
  // skip = triangle (traversal order) a spawned ray starts on, -1 = none (see self_prim())
  // fp32 spawned rays replace t_max (a constant of their queue: inf / 1 - 1e-4) by the packed low word of a double-float
  // origin (pack_lo() / spawn_point() in dkernels.hpp); top bits 11 mark that form (a real t_max is never negative)
  V4 *ray_o, *ray_d;
  V4 *nray_o, *nray_d;   // rays being spawned for q_next (the host swaps them together with the queues)
  V4* hit;               // {t, prim, u, v} for the ray at the same queue position
  V4 *sray_o, *sray_d;   // shadow rays, in shadow-queue order
  V4* sld;               // {Ld.r, Ld.g, Ld.b, slot}: pending contribution of the shadow ray at the same position
  // path state that changes every bounce travels with the queue too (cur / next, swapped with the rays)
  V4 *path, *npath;      // {beta.r, beta.g, beta.b, eta_scale (path.rs:70, 150-162)}
  QEnt *q_active, *q_next;   // {slot, dimension counter (low 16) | bounces (high 16), Halton index, -}
  // per-slot state: only what outlives a path's queue entries
  V4* L;                 // {L.r, L.g, L.b, -}: radiance; touched by unoccluded shadow rays and the film kernel
  R* weight;             // camera ray weight (0 = dead sample: its L is never read)
  V4* samp;              // camera sample {p_film.x, p_film.y, p_lens.x, p_lens.y}: one 128-bit gather per survivor
  uint32_t* hindex;      // Halton index (raygen -> first queue entry)
  uint32_t* pix_off;     // per pixel of the pass, two words: {Halton index of its sample 0, y << 16 | x} (halton_pixel_offset: two 64-bit products and a 64-bit
                         // modulo, the same for all samples of a pixel), filled by k_pixel_offsets before the fp32 sampling kernel
  // camera ray differentials after scale_differentials (scenes with textured materials only, else null):
  // {rx_origin, -}, {rx_direction, -}, {ry_origin, -}, {ry_direction, -} per slot
  V4 *rdx_o, *rdx_d, *rdy_o, *rdy_d;
  uint32_t* counters;    // [0] active, [1] next, [2] shadow, [3] camera rays, [4..] stats
  uint32_t* shadow_count;  // the shadow queue's counter (the path integrator alternates two shadow queues, see render_impl)
};

}  // namespace rrtd
