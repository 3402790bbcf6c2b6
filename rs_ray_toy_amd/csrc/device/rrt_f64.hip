// Parity mode: fp64 arithmetic, compiled with -ffp-contract=off so results track the f64 oracle to rounding.
#include "rrt_impl.hpp"
namespace rrtd {
HandleBase* make_handle_f64(int device, const rrt_scene_desc* d) { return new Handle<double>(device, d); }
}
