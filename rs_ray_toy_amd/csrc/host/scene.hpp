// rrt_scene: owner of every array an rrt_scene_desc points at (host memory).
#pragma once
#include <string>
#include <vector>

#include "rrt.h"
#include "vecmath.hpp"

namespace rrt {

struct SceneData {
  std::vector<double> positions, normals, uvs;
  std::vector<rrt_tri> tris;
  std::vector<rrt_sphere> spheres;
  std::vector<rrt_xform> xforms;
  std::vector<rrt_prim> prims;
  std::vector<rrt_material> materials;
  std::vector<rrt_texture> textures;
  std::vector<rrt_image> images;
  std::vector<double> image_texels;
  std::vector<rrt_light> lights;
  std::vector<rrt_bvh_node> bvh_nodes;
  std::vector<uint32_t> prim_order;
  std::vector<rrt_lens_elem> lens;
  std::vector<uint16_t> perms;
  rrt_scene_desc desc{};
  std::vector<std::string> warnings;  // the reference's non-fatal eprintln! diagnostics
  void finalize();                    // point desc at the vectors
};

// ImageTexture support (imagemap.cpp): PNG -> rgb8 as `image::open(..).decode().into_rgb8()` gives it (0 ok, 1 not
// decodable = the reference skips the texture, 2 decodable there but not restated here), and MIPMap::create.
int decode_png_rgb8(const std::string& path, uint32_t* w, uint32_t* h, std::vector<uint8_t>* rgb, std::string* why);
int build_mipmap(SceneData& s, uint32_t w, uint32_t h, const std::vector<uint8_t>& rgb8, bool do_trilinear, double max_aniso, int wrap);

// BVHAccel::new bvh.rs:307-363 (HLBVH); fills bvh_nodes, prim_order, bvh_depth, world_bound.
void build_bvh(SceneData& s, uint32_t max_prims_in_node, uint32_t flags);
// world_bound of prim i as the reference computes it (Shape/Transformed world_bound).
B3 prim_world_bound(const SceneData& s, size_t prim_index);

// samplers/halton.rs:23-61 + lowdiscrepancy.rs:250-270 (seeded)
void init_halton(SceneData& s, uint64_t nsamp, bool sample_at_center, const int32_t sample_bounds[4], uint64_t seed);
const uint16_t* prime_table();          // first 1024 primes (lowdiscrepancy.rs:101, PRIME_NUMS)
const uint32_t* prime_sums_table();     // PRIME_SUMS lowdiscrepancy.rs:8
constexpr int kPrimeTableSize = 1000;   // PRIME_TABLE_SIZE lowdiscrepancy.rs:3
double radical_inverse_host(int base_index, uint64_t a);  // lowdiscrepancy.rs:230-236

// RealisticCamera::new camera.rs:66-135 (thick-lens focus + exit pupil bounds)
void init_camera(SceneData& s, const Xf& camera_to_world, double shutter_open, double shutter_close,
                 double aperture_diameter, double focus_distance, const std::vector<double>& lens_data,
                 bool simple_weighting);

// objparser.rs:83-196
struct ObjMesh {
  std::vector<double> p, n, uv;
  std::vector<uint32_t> vi, ni, uvi;
};
ObjMesh parse_obj(const std::string& path, std::vector<std::string>& warnings);

}  // namespace rrt

struct rrt_scene {
  rrt::SceneData data;
};
