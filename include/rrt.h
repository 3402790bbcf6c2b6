/*
 * rrt.h — C ABI of the MI355X-native render hot path that replaces
 *         rs_ray_toy's  renderprocess.rs -> integrator/ -> bvh.rs + shape/ ->
 *         reflection.rs/material/ -> film.rs   (reference @ /root/reference/src).
 *
 * The reference has no FFI of its own (single Rust crate, no extern "C"); the
 * boundary a Rust maintainer would bind is therefore defined here, one entry
 * point per reference call it replaces (file:line cited on each prototype).
 * INTEGRATION.md shows the matching `extern "C"` block and the ctypes stub.
 *
 * Conventions
 *   - plain pointers + sizes, no C++ / torch types;
 *   - every function returns 0 on success or a negative RRT_E* code and never
 *     throws / aborts; rrt_last_error() gives the message for this thread;
 *   - the reference's panics (assert!/unwrap) map to RRT_EPANIC with the same
 *     condition in the message;
 *   - scene data is *host* memory owned by an rrt_scene; device state is owned
 *     by an rrt_handle (one HIP stream per handle, thread-compatible);
 *   - ray / hit / film batches may be host or device memory (`mem` field).
 *
 * All reference arithmetic is f64 (geometry.rs:12-20); the scene description
 * below therefore carries f64, and the device handle picks its compute type
 * (RRT_F32 = product path, RRT_F64 = bit-tight parity mode) at create time.
 */
#ifndef RRT_H
#define RRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_ABI_VERSION 11

/* ---- error codes ------------------------------------------------------- */
enum {
  RRT_OK = 0,
  RRT_EINVAL = -1,   /* bad argument                                         */
  RRT_EIO = -2,      /* file could not be read / written                     */
  RRT_EPARSE = -3,   /* scene.json / OBJ syntax                              */
  RRT_EPANIC = -4,   /* the reference would panic here (message says where)  */
  RRT_EUNSUP = -5,   /* feature outside SURVEY §8 scope (named in message)   */
  RRT_EDEVICE = -6,  /* HIP error / no GPU / extension not usable            */
  RRT_ENOMEM = -7
};

/* ---- compat flags (SURVEY §8.0 quirk ledger). Default = all reference. --- */
enum {
  RRT_FIX_BVH_LBVH_SLICE = 1u << 0, /* Q26 off: emit_lbvh second child uses slice[split..] */
  RRT_FIX_BVH_SAH        = 1u << 1, /* Q27 off: real 12-bucket SAH in build_upper_sah      */
  RRT_SKIP_MIS_BSDF_RAY  = 1u << 2, /* do not trace estimate_direct's BSDF-sampled ray;
                                       result-invariant because of Q18     */
  /* How the DEVICE evaluates triangle instances (TransformedPrimitive, primitives.rs:115-139; transform.rs:525-537). "Kept": every
   * instance goes through the reference's own per-primitive ray transform (world_to_prim.t(ray), object-space test, t copied
   * back, interaction transformed) - the reference's evaluation order, bit for bit in RRT_F64. "Flattened": rigid instances
   * are moved to world space once at upload (one ray per traversal; exact box / face ties may break differently, DESIGN.md
   * section 4). Defaults: RRT_F64 keeps (parity mode = the reference's order), RRT_F32 flattens (product speed; non-rigid
   * instances are always kept). The two flags override the default of either mode. */
  RRT_INSTANCES_FLATTEN  = 1u << 3,
  RRT_INSTANCES_KEEP     = 1u << 4
};
#define RRT_FIXED_BVH (RRT_FIX_BVH_LBVH_SLICE | RRT_FIX_BVH_SAH)

/* ---- enums mirrored from the reference's scene.json vocabulary ---------- */
enum { RRT_PRIM_TRIANGLE = 0, RRT_PRIM_SPHERE = 1 };
enum { /* material_type, renderprocess.rs:664-871 */
  RRT_MAT_MATTE = 0, RRT_MAT_PLASTIC = 1, RRT_MAT_METAL = 2, RRT_MAT_MIRROR = 3, RRT_MAT_DEBUG = 4,
  RRT_MAT_GLASS = 5, RRT_MAT_TRANSLUCENT = 6   /* material/glass.rs, material/translucent.rs */
};
enum { RRT_LIGHT_POINT = 0, RRT_LIGHT_DIFFUSE = 1, RRT_LIGHT_DISTANT = 2 };  /* renderprocess.rs:991-1031 */
enum { RRT_SAMPLER_HALTON = 0, RRT_SAMPLER_STRATIFIED = 1 };    /* renderprocess.rs:1306-1325 */
enum { RRT_FILTER_BOX = 0, RRT_FILTER_TRIANGLE = 1, RRT_FILTER_GAUSSIAN = 2 };
enum { /* integrator_type, renderprocess.rs:1399-1499 */
  RRT_INT_PATH = 0, RRT_INT_DIRECT = 1, RRT_INT_DEBUG = 2, RRT_INT_AO = 3
};
enum { RRT_STRATEGY_ONE = 0, RRT_STRATEGY_ALL = 1 };
enum { RRT_F32 = 0, RRT_F64 = 1 };
enum { RRT_MEM_HOST = 0, RRT_MEM_DEVICE = 1 };

/* ---- flattened scene description (host memory, read-only to callers) ---- */

/* Transform{m, m_inv}, transform.rs:181-184, row-major 4x4 */
typedef struct rrt_xform { double m[16]; double m_inv[16]; } rrt_xform;

/* Triangle{v,n,uv,mesh}, shape/triangle.rs:63-70 (indices into the pooled arrays) */
typedef struct rrt_tri {
  uint32_t v[3];
  uint32_t n[3];    /* valid iff mesh_has_n  */
  uint32_t uv[3];   /* valid iff mesh_has_uv */
  /* Triangle::new triangle.rs:73-111. 0 = mesh.n (mesh.uv) empty; 1 = array and its index list present;
   * 2 = array present, index list empty (the reference then uses element 0 three times). */
  uint8_t mesh_has_n;
  uint8_t mesh_has_uv;
  uint8_t pad[2];
} rrt_tri;

/* Sphere, shape/sphere.rs:17-26 */
typedef struct rrt_sphere {
  int32_t xform;    /* obj_to_world (index into xforms) */
  double radius, z_min, z_max, theta_min, theta_max, phi_max;
} rrt_sphere;

/* GeometricPrimitive (+ optional TransformedPrimitive), primitives.rs:20-30 */
typedef struct rrt_prim {
  uint8_t type;         /* RRT_PRIM_*                                 */
  uint8_t pad[3];
  uint32_t shape;       /* index into tris / spheres                  */
  int32_t instance;     /* primitive_to_world xform index, -1 = none  */
  uint32_t material;    /* index into materials                       */
} rrt_prim;

/* Texture graph (texture/ module, built by make_textures renderprocess.rs:298-515). Float and rgb textures share one
 * array; a float texture carries its value in all three channels. Children are indices of textures created EARLIER
 * (the reference captures the Arc at creation time, so a later texture of the same name does not rebind them), or -1
 * with the fallback constant in `fallback[]`. ImageTexture: 8-bit non-interlaced PNG files (other formats / depths of
 * the `image` crate are RRT_EUNSUP where a material uses the texture), as the MIPMap of mipmap.rs in `images[]`. */
enum { RRT_TEX_CONSTANT = 0, RRT_TEX_MIX = 1, RRT_TEX_BILERP = 2, RRT_TEX_CHECKER2D = 3, RRT_TEX_CHECKER3D = 4,
       RRT_TEX_SCALE = 5, RRT_TEX_WINDY = 6, RRT_TEX_WRINKLED = 7, RRT_TEX_UV = 8, RRT_TEX_IMAGE = 9 };
enum { RRT_WRAP_REPEAT = 0, RRT_WRAP_BLACK = 1, RRT_WRAP_CLAMP = 2 };   /* ImageWrap mipmap.rs:50-55 */
enum { RRT_MAP_UV = 0, RRT_MAP_SPHERICAL = 1, RRT_MAP_CYLINDRICAL = 2, RRT_MAP_PLANAR = 3, RRT_MAP_IDENTITY3D = 4 };
typedef struct rrt_texture {
  int32_t type;            /* RRT_TEX_*                                                              */
  int32_t mapping;         /* RRT_MAP_* (texture/mod.rs:205-374)                                     */
  int32_t child[3];        /* t1, t2, amount (Mix) -> textures[] index or -1                         */
  int32_t aa_none;         /* Checkerboard2D: aamode == "none" (checkerboard.rs:13-16)               */
  int32_t octaves;         /* Wrinkled                                                               */
  int32_t image;           /* IMAGE: index into rrt_scene_desc.images                                */
  double fallback[3][3];   /* get_text_fallback's ConstantTexture per child (renderprocess.rs:282-296) */
  double v[4][3];          /* CONSTANT: v[0]; BILERP: v00, v01, v10, v11                              */
  double omega;            /* Wrinkled                                                               */
  double map[4];           /* UV: su, sv, du, dv;  PLANAR: ds, dt                                    */
  double vs[3], vt[3];     /* PLANAR                                                                 */
  double world_to_texture[16];   /* SPHERICAL, CYLINDRICAL: inverse(to_world); IDENTITY3D: to_world itself
                                    (renderprocess.rs:368,384,388 pass `to_world` as world_to_texture) */
} rrt_texture;

/* MIPMap (mipmap.rs:70-96) as MIPMap::create (:270-382) leaves it. Each pyramid level is the BlockedArray's own
 * `data` vector (memory.rs:24-98): texel (u, v) lives at
 *     16 * (u_blocks * (v & 3) + (u & 3)) + 4 * (v >> 2) + (u >> 2)
 * - the index expression of memory.rs:76-85, which swaps pbrt's block / offset roles, so texels alias each other and the
 * level holds whatever was written last at each cell. Lookups go through the same expression. */
typedef struct rrt_image_level { uint32_t u_res, v_res, u_blocks, pad; uint64_t offset, n; /* RGB triples in image_texels */ } rrt_image_level;
typedef struct rrt_image {
  int32_t do_trilinear, wrap;   /* RRT_WRAP_* */
  double max_aniso;
  int32_t n_levels, pad;
  rrt_image_level levels[16];
} rrt_image;

/* material parameter slots of rrt_material.tex[] */
enum { RRT_P_KD = 0, RRT_P_KS, RRT_P_KR, RRT_P_ETA, RRT_P_K, RRT_P_SIGMA, RRT_P_ROUGHNESS, RRT_P_UROUGHNESS,
       RRT_P_VROUGHNESS, RRT_P_KT, RRT_P_REFLECT, RRT_P_TRANSMIT, RRT_P_INDEX, RRT_P_COUNT };

/* material parameters, material/{matte,plastic,metal,mirror,debug_material,glass,translucent}.rs: the constant
 * value of each parameter, or (tex[slot] >= 0) the texture that replaces it at every hit */
typedef struct rrt_material {
  int32_t type;         /* RRT_MAT_*                                  */
  int32_t remap_roughness;
  double kd[3], ks[3], kr[3];
  double eta[3], k[3];
  double sigma, roughness, u_roughness, v_roughness;
  double kt[3];         /* glass */
  double reflect[3], transmit[3];   /* translucent */
  double index;         /* glass "eta" (a float texture there; `eta` above is MetalMaterial's spectrum) */
  int32_t tex[RRT_P_COUNT];   /* index into rrt_scene_desc.textures per RRT_P_* slot, -1 = the constant above */
  int32_t bump;               /* "bump_map": float texture displacing the shading geometry (Material::bump, material/mod.rs:22-62), -1 = none */
} rrt_material;

/* PointLight lights/point.rs:13-19, DiffuseAreaLight lights/diffuse.rs:13-22 */
typedef struct rrt_light {
  int32_t type;         /* RRT_LIGHT_*                                */
  int32_t n_samples;
  double spectrum[3];   /* I (point) or Lemit (diffuse)               */
  double p_light[3];    /* always 0,0,0 in the reference (Q17)        */
  int32_t shape_type;   /* RRT_PRIM_* of light_shape (diffuse only)   */
  uint32_t shape;       /* index into spheres / tris                  */
  double area;
  double w_light[3];    /* distant: normalize(light_to_world(from - to)), lights/distant.rs:30 */
  double world_radius;  /* distant: bounding sphere of aggregate.world_bound(), distant.rs:31-33, geometry.rs:1656-1668 */
} rrt_light;

/* LinearBVHNode, bvh.rs:103-109 (f64 bounds as built; device narrows conservatively) */
typedef struct rrt_bvh_node {
  double bounds[6];     /* pmin xyz, pmax xyz                         */
  uint32_t offset;      /* leaf: first prim in prim_order; interior: second child */
  uint32_t n_primitives;
  uint32_t axis;
  uint32_t pad;
} rrt_bvh_node;

/* LensElementInterface after RealisticCamera::new, camera.rs:32-38,80-99 (metres) */
typedef struct rrt_lens_elem { double curvature_radius, thickness, eta, aperture_radius; } rrt_lens_elem;

typedef struct rrt_camera {
  rrt_xform camera_to_world;
  double shutter_open, shutter_close;
  int32_t simple_weighting;
  int32_t n_elems;
  const rrt_lens_elem* elems;
  double exit_pupil_bounds[64][4];   /* Bounds2f pmin.x,pmin.y,pmax.x,pmax.y; camera.rs:121-133 */
  uint8_t exit_pupil_valid[64];      /* only slabs the path can index are computed (Q6): 0 and 63 */
} rrt_camera;

typedef struct rrt_film {
  int32_t xres, yres;
  int32_t crop[4];                   /* cropped_pixel_bounds x0,y0,x1,y1; film.rs:151-160 */
  int32_t sample_bounds[4];          /* get_sample_bounds, film.rs:188-199 */
  double diagonal;                   /* metres */
  double physical_extent[4];         /* get_physical_extent, film.rs:200-208 */
  int32_t filter_type;
  double filter_radius[2];
  double filter_alpha;
  double filter_table[256];          /* film.rs:163-173 (incl. Q4) */
  double scale, max_sample_luminance;
} rrt_film;

typedef struct rrt_sampler {
  int32_t type;
  int32_t sample_at_center;
  uint64_t samples_per_pixel;        /* nsamp; effective = nsamp-1 (Q1) */
  /* Halton, samplers/halton.rs:23-61 */
  int64_t base_scales[2], base_exponents[2];
  uint64_t sample_stride, mult_inverse[2];
  const uint16_t* perms;             /* radical_inverse_permutations, lowdiscrepancy.rs:250-270 (seeded) */
  size_t n_perms;
  uint64_t perm_seed;
  /* Stratified (oracle/host only) */
  int32_t xsamp, ysamp, dimension, jitter;
} rrt_sampler;

typedef struct rrt_integrator {
  int32_t type;
  int32_t max_depth;
  double rr_threshold;
  int32_t light_strategy;
  int32_t cos_sample, n_samples;
} rrt_integrator;

typedef struct rrt_scene_desc {
  uint32_t abi_version, flags;
  /* pooled mesh data (TriangleMesh, triangle.rs:16-28) */
  const double* positions; size_t n_positions;   /* xyz triples */
  const double* normals;   size_t n_normals;
  const double* uvs;       size_t n_uvs;         /* uv pairs */
  const rrt_tri* tris;       size_t n_tris;
  const rrt_sphere* spheres; size_t n_spheres;
  const rrt_xform* xforms;   size_t n_xforms;
  const rrt_prim* prims;     size_t n_prims;     /* aggregate input order, renderprocess.rs:1178-1304 */
  const rrt_material* materials; size_t n_materials;
  const rrt_texture* textures;   size_t n_textures;    /* every declared texture, declaration order */
  const rrt_image* images;       size_t n_images;
  const double* image_texels;    size_t n_image_texels;   /* RGB triples of all pyramid levels */
  const rrt_light* lights;   size_t n_lights;    /* scene.lights; infinite_lights unsupported */
  /* BVHAccel, bvh.rs:116-121 */
  const rrt_bvh_node* bvh_nodes; size_t n_bvh_nodes;
  const uint32_t* prim_order;    size_t n_prim_order;  /* ordered_prims -> index into prims */
  uint32_t max_prims_in_node, bvh_depth;
  double world_bound[6];
  rrt_camera camera;
  rrt_film film;
  rrt_sampler sampler;
  rrt_integrator integrator;
} rrt_scene_desc;

typedef struct rrt_scene rrt_scene;    /* owns every array above */
typedef struct rrt_handle rrt_handle;  /* device state            */

/* ---- batches ------------------------------------------------------------ */
typedef struct rrt_rays {   /* SoA; element type follows the handle's precision */
  int32_t mem;              /* RRT_MEM_*                                          */
  int32_t precision;        /* RRT_F32 / RRT_F64 of the arrays below              */
  const void *ox, *oy, *oz, *dx, *dy, *dz, *tmax;
  /* optional (may be NULL): for a ray spawned on a surface, the triangle it starts on (index as returned in
   * rrt_hits.prim), excluded from the tests; -1 = none. The reference needs no such field because f64 places
   * the self-hit at t ~ 1e-15 < 1e-7 (triangle.rs:200,263); fp32 callers should pass it. */
  const int32_t* skip_prim;
} rrt_rays;

typedef struct rrt_hits {
  int32_t mem, precision;
  void* t;                  /* ray.t_max after traversal (unchanged if miss)      */
  int32_t* prim;            /* index into prim_order (traversal order), -1 = miss */
  void *u, *v;              /* barycentrics of the winning hit                    */
  uint32_t* nodes_visited;  /* optional (may be NULL): per-ray counters, §8(d)    */
  uint32_t* prims_tested;
} rrt_hits;

typedef struct rrt_render_stats {
  uint64_t camera_samples;      /* W*H*(nsamp-1) in range                         */
  uint64_t camera_rays;         /* samples with weight>0: integrator/mod.rs:101   */
  uint64_t closest_queries, any_queries;
  uint64_t nodes_visited, prims_tested;   /* closest + any, when counting is enabled (count_traversal option) */
  double ms_total, ms_raygen, ms_closest, ms_any, ms_shade, ms_film;
  uint64_t closest_launches, any_launches;
  uint64_t closest_nodes, closest_prims;  /* split of nodes_visited / prims_tested per kernel: the roofline's */
  uint64_t any_nodes, any_prims;          /* algorithmic-byte model needs the closest-hit kernel's own counts */
  uint64_t tile_launches;       /* closest-hit launches over camera rays issued with the per-patch sub-trees ("tile_trees" option) */
  uint64_t root_culled;         /* closest_queries answered by the camera kernels: camera rays that miss the root box ("root_cull" option) */
  uint64_t sky_culled;          /* closest_queries answered by the horizon tables: bounce rays the path shading kernel proves to leave the scene ("horizon_cull" option) */
  uint64_t list_launches;       /* any-hit launches served by the shadow candidate lists ("shadow_lists" option) instead of the tree walk */
  double ms_gather;             /* the collective rrt_film_gather / _gather_all enqueued behind a frame in flight (HIP events on the handle's stream around
                                 * the grouped send / recv or the reduce): lets a multi-GPU run separate the ranks' render imbalance (ms_total) from the
                                 * collective (film.rs:248-263 merge_film_tile is what it replaces); 0 when no collective followed the frame */
  double s_horizon_build;       /* host seconds rrt_create spent building this handle's horizon tables (a setup cost, like the BVH build; 0 when another handle
                                 * of the process had built them for the same geometry, or the scene gets none) */
} rrt_render_stats;

/* ---- host side: scene build (stays on the host in the north_star) -------- */

/* deploy_render's loader half: renderprocess.rs:92-105 make_scene + make_integrator
 * (incl. objparser.rs parse_obj, BVHAccel::new bvh.rs:307, RealisticCamera::new camera.rs:66). */
int rrt_scene_load(const char* scene_json_path, uint32_t flags, uint64_t perm_seed, rrt_scene** out);
/* same, JSON text + root dir for relative assets (renderprocess.rs:94-99,114-120) */
int rrt_scene_load_str(const char* json_text, const char* root_dir, uint32_t flags, uint64_t perm_seed,
                       rrt_scene** out);
const rrt_scene_desc* rrt_scene_desc_of(const rrt_scene*);
void rrt_scene_free(rrt_scene*);
/* the reference's non-fatal eprintln! diagnostics raised while loading (unsupported texture types, ...) */
size_t rrt_scene_warning_count(const rrt_scene*);
const char* rrt_scene_warning(const rrt_scene*, size_t i);

/* what a host needs of the film to size its buffers and resolve the image (Film::full_resolution film.rs:137, Film::scale :142) without
 * reading rrt_scene_desc's layout; any pointer may be NULL */
int rrt_scene_film(const rrt_scene*, int32_t* xres, int32_t* yres, double* scale);

/* Film::write_image film.rs:323-366 (Q3 already folded into w) + write_image renderprocess.rs:1501-1530.
 * film_xyzw: W*H*4 host floats/doubles (X,Y,Z sums and filter_weight_sum per pixel). */
int rrt_resolve_rgba8(const void* film_xyzw, int precision, int w, int h, double scale, uint8_t* rgba);
int rrt_write_png(const char* path, const uint8_t* rgba, int w, int h);

/* ---- device side --------------------------------------------------------- */
int rrt_device_count(void);
/* uploads the scene (rigid triangle instances flattened to world space; spheres keep their transforms); the wavefront
 * pools are allocated on first use. RRT_EDEVICE without a HIP device: there is no CPU fallback. */
int rrt_create(int device, const rrt_scene_desc* desc, int precision, rrt_handle** out);
void rrt_destroy(rrt_handle*);
/* non-fatal diagnostics of rrt_create, in the manner of the reference's eprintln! lines (rrt_scene_warning): what this handle's precision
 * mode does not claim for this scene - RRT_F32 with transmissive sphere primitives (sphere.rs:124-259 has no epsilon: the reference's own
 * sphere pixels hang on last-bit coins that fp32 cannot replay; use RRT_F64) - and shortcuts that were switched off */
size_t rrt_warning_count(const rrt_handle*);
const char* rrt_warning(const rrt_handle*, size_t i);
/* raw hipStream_t of the handle (for event timing by the caller) */
void* rrt_stream(rrt_handle*);

/* BVHAccel::intersect bvh.rs:183-236 + Triangle/Sphere::intersect (closest = "last accepted", Q10) */
int rrt_trace_closest(rrt_handle*, const rrt_rays* rays, size_t n, rrt_hits* out);
/* BVHAccel::intersect_p bvh.rs:124-173 + Triangle::intersect_p (Q11) */
int rrt_trace_any(rrt_handle*, const rrt_rays* rays, size_t n, uint8_t* occluded /* mem as rays->mem */);

/* ISampler::get_camerasample samplers/mod.rs:28-34 + RealisticCamera::generate_ray_differential
 * camera.rs:582-628 for sample_num in [s0,s1) of every pixel in [x0,x1)x[y0,y1) (unit test surface).
 * Outputs (host, doubles): per sample 5 sampler dims, ray o(3) d(3), weight. Layout [pixel][sample]. */
int rrt_camera_samples(rrt_handle*, const int32_t rect[4], uint64_t s0, uint64_t s1,
                       double* dims5, double* ray_od6, double* weight);

/* SamplerIntegrator::si_render integrator/mod.rs:48-139 restricted to pixel rect [x0,y0,x1,y1):
 * all samples of those pixels, Li by the scene's integrator, FilmTile::add_sample + merge_film_tile.
 * film_xyzw: full-frame W*H*4 buffer (handle precision, host or device per film_mem); only pixels the
 * rect's samples touch are written (+=). */
int rrt_render_rect(rrt_handle*, const int32_t rect[4], void* film_xyzw, int film_mem,
                    rrt_render_stats* stats /* may be NULL */);

/* Multi-GPU film partition: the same, for the rows of the interleaved 16-row tile bands b with
 * b % world == rank (tile height of integrator/mod.rs:55), rendered as one pixel set. Box filter: the ranks' films are
 * disjoint; wider filters: a sample's splat may land in a neighbour's rows of this rank's film. Either way summing
 * the ranks' films (one RCCL reduce) reassembles Film::pixels. */
int rrt_render_bands(rrt_handle*, int rank, int world, void* film_xyzw, int film_mem, rrt_render_stats* stats);

/* Frames in flight (no counterpart in the reference, whose si_render returns with the frame done): _begin enqueues
 * rrt_render_bands on the handle's stream for a DEVICE film and returns; _end waits for that frame and reports what
 * rrt_render_bands would have returned (RRT_EPANIC ...). One frame per handle; a second handle on the same GPU can
 * render the next frame meanwhile, which hides the latency-bound last bounces of one frame behind the camera rays of
 * the next. The film must stay untouched by the caller between the two calls. For the frames of two handles to overlap,
 * both need rrt_set_option("nonblocking_streams", 1): the handle's streams then no longer synchronise with the legacy
 * default stream, so work the caller issued there on the film (a memset ...) must be finished before _begin. */
int rrt_render_bands_begin(rrt_handle*, int rank, int world, void* film_xyzw_device);
int rrt_render_end(rrt_handle*);
/* the same, and the frame's statistics; kernel timings (HIP events on the handle's streams around every launch) are recorded for frames in
 * flight when the option "frame_stats" is 1 */
int rrt_render_end_stats(rrt_handle*, rrt_render_stats* stats);

/* ---- multi-GPU film reassembly: RCCL over xGMI, one collective per frame ----
 * The reference has one address space: its rayon tiles merge under a lock (Film::merge_film_tile film.rs:248-263, driven from
 * integrator/mod.rs:64-74,133). Across GPUs every rank renders its bands (rrt_render_bands / _begin) into its own device
 * film and the films meet on `root`. Box filter of radius <= 0.5: the bands are disjoint and only they travel (grouped
 * ncclSend / ncclRecv - a gather: every rank ships 1/world of the film over its direct xGMI link to root); wider filters
 * splat across band borders and the collective is ncclReduce(sum) of the whole film. The call is enqueued on the handle's
 * stream (after the frame it follows) and returns; rrt_render_end() or a stream synchronisation completes it. */

/* rows [y0, y1) of the bands `rank` owns out of `world` (bands b of 16 rows, integrator/mod.rs:55, with b % world == rank).
 * Returns the number of bands and fills up to max_bands {y0, y1} pairs (y0y1 may be NULL); host arithmetic only. */
int rrt_band_rows(int yres, int rank, int world, int32_t* y0y1, int max_bands);

typedef struct rrt_comm rrt_comm;
#define RRT_COMM_ID_BYTES 128
/* one process per GPU: rank 0 draws an id (ncclGetUniqueId) and hands the bytes to the other ranks by any channel it has;
 * every rank then joins (ncclCommInitRank; collective: returns once all `world` ranks have called it) */
int rrt_comm_id(uint8_t id[RRT_COMM_ID_BYTES]);
int rrt_comm_create(const uint8_t id[RRT_COMM_ID_BYTES], int rank, int world, int device, rrt_comm** out);
void rrt_comm_destroy(rrt_comm*);
/* the collective; film_xyzw_device = the W*H*4 device film this rank rendered with (comm rank, comm world) */
int rrt_film_gather(rrt_handle*, rrt_comm*, void* film_xyzw_device, int root);
/* one process that owns all GPUs of the node (the shape of the reference's single binary): handles[i] rendered rank i of
 * n into films_device[i] on its own device; communicators (ncclCommInitAll) are created on first use and kept */
int rrt_film_gather_all(rrt_handle* const* handles, void* const* films_device, int n, int root);

/* Handle options (rrt_set_option). "fp32" = only the RRT_F32 product mode looks at it. Every option marked "invariant" switches a
 * shortcut whose results are identical bit for bit with it on or off (frames, filter weights, query counts); the named test holds that.
 *
 * name                  default   values / meaning                                                              invariant it keeps (test in tests/test_gpu_parity.py)
 * --------------------  --------  ----------------------------------------------------------------------------  ------------------------------------------------------
 * max_paths             2^28      wavefront pool slots (>= 64); clamped to half of the free HBM at creation      frames (fp32: to rounding of the per-pass sums)
 * frame_stats           0         1: frames in flight record kernel timings (rrt_render_end_stats)              -
 * nonblocking_streams   0         1: the handle's streams stop synchronising with the legacy default stream     - (see rrt_render_bands_begin)
 * count_traversal       0         1: exact node / triangle-test counters in rrt_render_stats (generic kernels)  -
 * overlap_shadow        1         0: shadow launches on the main stream instead of a second one                 frames
 * persistent_traversal  3  fp32   0 generic kernels, 1 grid-stride pair-node kernel, 2 persistent-thread kernel, frames: every mode makes the reference's decisions in its
 *                                 3 both by queue size                                                           order (test_trace_modes_agree, sphere / instance tests)
 * pt_split_closest/_any 100000    queue size at which mode 3 switches kernels                                    frames
 * quad_nodes            0  fp32   1: closest-hit rays of the persistent kernel fetch two tree levels at a time   invariant (test_quad_nodes_change_nothing)
 *                                 (128-byte nodes with the four grandchild boxes, visited in the tree's order)
 * tile_order            1         pixels of a pass enumerated in 8 x 8 tiles where the rect is whole tiles      invariant (test_tile_order_of_the_pixels_changes_nothing)
 * tile_trees            1  fp32   camera rays walk per-patch copies of the most visited nodes in LDS;            invariant (test_tile_trees_change_nothing);
 *                                 rrt_render_stats::tile_launches counts those launches                          films above 4096^2 pixels keep the ordinary kernel
 * tt_census             2  fp32   camera samples per pixel of the census that chooses the copied nodes           invariant (same test, two densities)
 * root_cull             1  fp32   path integrator: camera rays that miss the BVH's root box are answered by     invariant (test_root_cull_changes_nothing);
 *                                 the camera kernel (rrt_render_stats::root_culled), never queued                counted in closest_queries
 * shadow_lists          1  fp32   shadow rays towards point / distant / sphere-area lights run down their       invariant (test_shadow_candidate_lists_change_nothing);
 *                                 start triangle's list of candidate leaves (rrt_render_stats::list_launches)    same box and triangle tests
 * sl_grid               32768     workgroup cap of the list kernel                                               frames
 * any_entry             1  fp32   shadow rays that walk the tree start from their triangle's deciding nodes     invariant (test_any_hit_entry_nodes_change_nothing)
 * cam_tables            1  fp32   block tables instead of the digit loops of the camera's Halton dimensions      invariant (test_camera_halton_block_tables_change_nothing)
 * halton_tables         1  fp32   the same for the integrators' first 64 dimensions                             invariant (test_halton_block_tables_change_nothing)
 * raygen_lean           1  fp32   1: dense camera kernels with the lean lens arithmetic; 0: the generic          fp32 camera samples within 1e-3 of the f64 oracle either
 *                                 two-stage kernels in the reference's operation order                           way (test_camera_samples)
 * rg_spb                8  fp32   samples of one 8 x 8 tile per camera workgroup: 1, 2, 4 or 8                   frames
 * aux_margin            1  fp32   skip the auxiliary camera rays (camera.rs:593-620) where the main ray clears   a CALIBRATED HEURISTIC, not a proven bound: 16 x the largest
 *                                 every lens interface by the calibrated margin                                  auxiliary-ray displacement measured at scene load (16 384 host
 *                                                                                                                samples of the scene's own lens); validated bit for bit against
 *                                                                                                                the full traces (test_aux_margins_change_nothing)
 * horizon_cull          1  fp32   path integrator: a bounce ray whose elevation exceeds everything visible from  invariant (test_horizon_cull_changes_nothing);
 *                                 its start triangle in its azimuth sector (host-built tables) is answered as    counted in closest_queries and sky_culled
 *                                 the miss it is, never queued. The tables are built by rrt_create (host, all
 *                                 cores; ~5 s for 100k triangles, once per geometry and process:
 *                                 rrt_render_stats::s_horizon_build); RRT_HORIZON_TABLES=0 in the environment
 *                                 builds none
 * shade_compact         1         path shading kernel packs the HITS of a chunk of queue entries through LDS       invariant (test_shade_compaction_changes_nothing)
 *                                 before shading them (a miss is shaded with nothing)
 * shade_spec            1  fp32   path shading kernel instantiated for the lobe kinds the scene's materials      same arithmetic per lobe: frames equal to fp32 rounding,
 *                                 can produce; 0: always the general kernel                                      weights and counts identical (test_shading_kernel_specialisation)
 */
int rrt_set_option(rrt_handle*, const char* key, double value);

const char* rrt_last_error(void);
const char* rrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
