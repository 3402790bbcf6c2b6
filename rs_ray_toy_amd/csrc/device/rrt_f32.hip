// Product path: fp32 arithmetic (FMA contraction allowed), 32 B BVH nodes, 48 B triangles.
#include "rrt_impl.hpp"
namespace rrtd {
HandleBase* make_handle_f32(int device, const rrt_scene_desc* d) { return new Handle<float>(device, d); }
}
