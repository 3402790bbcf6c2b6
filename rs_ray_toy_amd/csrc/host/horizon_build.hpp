// Horizon tables of the fp32 path integrator: host builder (plain C++, no HIP) - see horizon_build.cpp for the derivation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include <cmath>

#if defined(__HIPCC__)
#define RRT_HZ_HD __host__ __device__
#else
#define RRT_HZ_HD
#endif

namespace rrtd {
// azimuth sector of a horizontal direction (a, b) - the two components of a vector other than its hz_axis one, in cyclic order: 16 wedges of 22.5 degrees, by
// comparisons only. Shared by the host builder and the shading kernel (build_horizons() marks a wedge's neighbours too where an interval ends on its border).
RRT_HZ_HD inline uint32_t hz_sector(float a, float b) {
  const float aa = fabsf(a), ab = fabsf(b), hi = fmaxf(aa, ab), lo = fminf(aa, ab);
  return (a < 0.0f ? 8u : 0u) | (b < 0.0f ? 4u : 0u) | (ab > aa ? 2u : 0u) | (lo > hi * 0.41421356f ? 1u : 0u);
}

// the builder's view of the device scene: fp32 boxes and vertices exactly as the kernels hold them, triangles in traversal order
struct HzNode { float bmin[3], bmax[3]; uint32_t offset; uint32_t n_prims; };   // leaf: first triangle; interior: second child (the first is the next node)
struct HzTri { float p[3][3]; uint32_t skip; };                                  // skip: not a world-space triangle (no table is built for it)
struct HzTables {
  std::vector<uint8_t> bytes;   // 32 per triangle: hemisphere +axis then -axis, 16 sectors each
  // per triangle: the tables speak for rays that start ON the triangle; a spawned ray starts within rho of its plane (fp32 evaluation of the barycentric sum and the
  // packed low word of the origin: rho = 2^-24 x the scene's largest coordinate bounds both), so its line meets the plane within rho (1 + tan) <= 2 rho / |n.d| of the
  // stored point - inside the triangle, where the tables hold, when min(barycentric) x (smallest altitude) exceeds that. tau = 4 rho / smallest altitude (twice the
  // bound); the kernel culls only where min(barycentric) |n.d| > tau. Infinite for a degenerate triangle.
  std::vector<float> tau;
  uint32_t axis = 0;
  double mean_open = 0.0;       // mean share of the upper hemisphere the tables declare free
  long checked = 0, check_hits = 0;   // self-check (check_rays > 0): rays declared free / those of them that hit a triangle (must be 0)
};
HzTables build_horizons(const HzNode* nodes, size_t n_nodes, const HzTri* tris, size_t n_tris, long check_rays);
}  // namespace rrtd
