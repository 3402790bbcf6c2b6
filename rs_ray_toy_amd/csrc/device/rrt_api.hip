// extern "C" device entry points of include/rrt.h: argument checks, precision dispatch, error mapping.
#include "rrt_impl.hpp"

namespace rrt { void set_last_error(const std::string& msg); void comm_cache_release(int device); }

struct rrt_handle { rrtd::HandleBase* impl; };

namespace {
template <typename F>
int guarded(F&& fn) {
  try { fn(); return RRT_OK; }
  catch (const rrtd::DeviceError& e) { rrt::set_last_error(e.what()); return RRT_EDEVICE; }
  catch (const rrtd::UnsupportedError& e) { rrt::set_last_error(std::string("unsupported: ") + e.what()); return RRT_EUNSUP; }
  catch (const rrtd::PanicError& e) { rrt::set_last_error(std::string("panic: ") + e.what()); return RRT_EPANIC; }
  catch (const std::invalid_argument& e) { rrt::set_last_error(e.what()); return RRT_EINVAL; }
  catch (const std::bad_alloc&) { rrt::set_last_error("out of memory"); return RRT_ENOMEM; }
  catch (const std::exception& e) { rrt::set_last_error(e.what()); return RRT_EINVAL; }
}
}  // namespace

extern "C" {

int rrt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int rrt_create(int device, const rrt_scene_desc* desc, int precision, rrt_handle** out) {
  if (!desc || !out) { rrt::set_last_error("rrt_create: null argument"); return RRT_EINVAL; }
  *out = nullptr;
  if (precision != RRT_F32 && precision != RRT_F64) { rrt::set_last_error("rrt_create: bad precision"); return RRT_EINVAL; }
  // the desc is checked before anything touches a device (and also where there is none): a caller-filled desc with a bad index or a
  // cyclic BVH must never reach a kernel
  { const int rc = guarded([&]() { rrtd::validate_desc(desc); }); if (rc != RRT_OK) return rc; }
  int n = rrt_device_count();
  if (n <= 0) { rrt::set_last_error("rrt_create: no HIP device visible (this library has no CPU fallback)"); return RRT_EDEVICE; }
  if (device < 0 || device >= n) { rrt::set_last_error("rrt_create: device index out of range"); return RRT_EINVAL; }
  return guarded([&]() {
    rrtd::HandleBase* h = precision == RRT_F32 ? rrtd::make_handle_f32(device, desc) : rrtd::make_handle_f64(device, desc);
    *out = new rrt_handle{h};
  });
}

void rrt_destroy(rrt_handle* h) {
  if (!h) return;
  rrt::comm_cache_release(h->impl->device());   // communicators rrt_film_gather_all cached for this device (rrt_comm.hip)
  delete h->impl;
  delete h;
}

size_t rrt_warning_count(const rrt_handle* h) { return h ? h->impl->warnings.size() : 0; }
const char* rrt_warning(const rrt_handle* h, size_t i) { return (h && i < h->impl->warnings.size()) ? h->impl->warnings[i].c_str() : nullptr; }

void* rrt_stream(rrt_handle* h) { return h ? (void*)h->impl->stream() : nullptr; }

int rrt_trace_closest(rrt_handle* h, const rrt_rays* rays, size_t n, rrt_hits* out) {
  if (!h || !rays || !out) { rrt::set_last_error("rrt_trace_closest: null argument"); return RRT_EINVAL; }
  if (n == 0) return RRT_OK;
  if (n > 0x7fffffffu) { rrt::set_last_error("rrt_trace_closest: batch too large"); return RRT_EINVAL; }
  if (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !rays->tmax || !out->t || !out->prim) {
    rrt::set_last_error("rrt_trace_closest: null ray/hit array"); return RRT_EINVAL;
  }
  if (rays->mem != out->mem) { rrt::set_last_error("rrt_trace_closest: rays and hits must live in the same memory kind"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->trace_closest(rays, n, out); });
}

int rrt_trace_any(rrt_handle* h, const rrt_rays* rays, size_t n, uint8_t* occluded) {
  if (!h || !rays || !occluded) { rrt::set_last_error("rrt_trace_any: null argument"); return RRT_EINVAL; }
  if (n == 0) return RRT_OK;
  if (n > 0x7fffffffu) { rrt::set_last_error("rrt_trace_any: batch too large"); return RRT_EINVAL; }
  if (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !rays->tmax) {
    rrt::set_last_error("rrt_trace_any: null ray array"); return RRT_EINVAL;
  }
  return guarded([&]() { h->impl->trace_any(rays, n, occluded); });
}

int rrt_camera_samples(rrt_handle* h, const int32_t rect[4], uint64_t s0, uint64_t s1, double* dims5, double* ray_od6, double* weight) {
  if (!h || !rect || !dims5 || !ray_od6 || !weight || s1 < s0) { rrt::set_last_error("rrt_camera_samples: bad argument"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->camera_samples(rect, s0, s1, dims5, ray_od6, weight); });
}

int rrt_render_rect(rrt_handle* h, const int32_t rect[4], void* film_xyzw, int film_mem, rrt_render_stats* stats) {
  if (!h || !rect || !film_xyzw) { rrt::set_last_error("rrt_render_rect: null argument"); return RRT_EINVAL; }
  if (film_mem != RRT_MEM_HOST && film_mem != RRT_MEM_DEVICE) { rrt::set_last_error("rrt_render_rect: bad film_mem"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->render_rect(rect, film_xyzw, film_mem, stats); });
}

int rrt_render_bands(rrt_handle* h, int rank, int world, void* film_xyzw, int film_mem, rrt_render_stats* stats) {
  if (!h || !film_xyzw) { rrt::set_last_error("rrt_render_bands: null argument"); return RRT_EINVAL; }
  if (film_mem != RRT_MEM_HOST && film_mem != RRT_MEM_DEVICE) { rrt::set_last_error("rrt_render_bands: bad film_mem"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->render_bands(rank, world, film_xyzw, film_mem, stats); });
}

int rrt_render_bands_begin(rrt_handle* h, int rank, int world, void* film_xyzw_device) {
  if (!h || !film_xyzw_device) { rrt::set_last_error("rrt_render_bands_begin: null argument"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->render_bands_begin(rank, world, film_xyzw_device); });
}

int rrt_render_end(rrt_handle* h) {
  if (!h) { rrt::set_last_error("rrt_render_end: null argument"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->render_end(nullptr); });
}

int rrt_render_end_stats(rrt_handle* h, rrt_render_stats* stats) {
  if (!h || !stats) { rrt::set_last_error("rrt_render_end_stats: null argument"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->render_end(stats); });
}

int rrt_set_option(rrt_handle* h, const char* key, double value) {
  if (!h || !key) { rrt::set_last_error("rrt_set_option: null argument"); return RRT_EINVAL; }
  return guarded([&]() { h->impl->set_option(key, value); });
}

}  // extern "C"
