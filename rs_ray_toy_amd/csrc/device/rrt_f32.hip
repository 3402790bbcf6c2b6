// Product path: fp32 arithmetic (FMA contraction allowed), 32 B BVH nodes, 48 B triangles.
#include "rrt_impl.hpp"
namespace rrtd {
HandleBase* make_handle_f32(int device, const rrt_scene_desc* d) { return new Handle<float>(device, d); }
}
#ifdef RRT_PT_STATS
extern "C" int rrt_debug_pt_stats(unsigned long long* out, int reset) {   // tuning instrumentation, variant builds only
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rrtd::g_pt_stats), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rrtd::g_pt_stats), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
