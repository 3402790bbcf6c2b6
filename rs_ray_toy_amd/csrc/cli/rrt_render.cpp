// rrt_render <scene.json> <out.png> — the reference's command line (main.rs:55-61 -> deploy_render renderprocess.rs:92-105)
// over the C ABI of include/rrt.h, plain C++ (no HIP, no Python): the caller a Rust host would be, see INTEGRATION.md.
//
//   deploy_render:  make_scene + make_integrator        -> rrt_scene_load (host: loader, OBJ parser, BVH build, camera init)
//                   inte.render(&scene)                 -> rrt_create + rrt_render_bands_begin on every GPU this process owns,
//                                                          rrt_film_gather_all (RCCL; nothing to do on one GPU) to device 0
//                   film.write_image -> write_image     -> rrt_resolve_rgba8 + rrt_write_png
// Diagnostics go to stderr like the reference's eprintln!; its "N rays generated" line (integrator/mod.rs:137) goes to stdout.
// Environment: RRT_GPUS = number of GPUs to partition the film over (default 1), RRT_PRECISION = f32 (default) | f64,
// RRT_FIXED_BVH = 1 builds pbrt's intended tree instead of the reference's (quirks Q26 / Q27).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rrt.h"

namespace {
int fail(const char* what, int rc) {
  std::fprintf(stderr, "rrt_render: %s failed (%d): %s\n", what, rc, rrt_last_error());
  return rc == RRT_EPANIC ? 101 : 1;   // a Rust panic exits with 101
}
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) {   // main.rs:56-59 indexes args[1], args[2]: fewer is an index panic there
    std::fprintf(stderr, "usage: %s <scene.json> <out.png>\n", argc > 0 ? argv[0] : "rrt_render");
    return 2;
  }
  const char* env_gpus = std::getenv("RRT_GPUS");
  const char* env_prec = std::getenv("RRT_PRECISION");
  const char* env_fix = std::getenv("RRT_FIXED_BVH");
  const int precision = (env_prec && std::strcmp(env_prec, "f64") == 0) ? RRT_F64 : RRT_F32;
  const uint32_t flags = (env_fix && std::atoi(env_fix) != 0) ? RRT_FIXED_BVH : 0u;
  int n_gpus = env_gpus ? std::atoi(env_gpus) : 1;
  const int visible = rrt_device_count();
  if (visible <= 0) { std::fprintf(stderr, "rrt_render: no HIP device visible (there is no CPU fallback)\n"); return 1; }
  if (n_gpus < 1) n_gpus = 1;
  if (n_gpus > visible) n_gpus = visible;

  rrt_scene* scene = nullptr;
  int rc = rrt_scene_load(argv[1], flags, 0x853C49E6748FEA9Bull, &scene);
  if (rc != RRT_OK) return fail("rrt_scene_load", rc);
  for (size_t i = 0; i < rrt_scene_warning_count(scene); i++) std::fprintf(stderr, "%s\n", rrt_scene_warning(scene, i));
  const rrt_scene_desc* desc = rrt_scene_desc_of(scene);   // opaque here: handed to rrt_create as it is
  int32_t W = 0, H = 0;
  double film_scale = 1.0;
  (void)rrt_scene_film(scene, &W, &H, &film_scale);
  const size_t word = precision == RRT_F32 ? 4 : 8, film_bytes = (size_t)W * (size_t)H * 4 * word;
  // the tiles banner of integrator/mod.rs:59-62
  std::fprintf(stderr, "Rendering %d x %d, %d tile rows of 16 over %d GPU(s)\n", W, H, (H + 15) / 16, n_gpus);

  std::vector<rrt_handle*> handles(n_gpus, nullptr);
  std::vector<void*> films(n_gpus, nullptr);
  auto cleanup = [&]() {
    for (int i = 0; i < n_gpus; i++) {
      if (handles[i]) rrt_destroy(handles[i]);
      if (films[i]) { (void)hipSetDevice(i); (void)hipFree(films[i]); }
    }
    rrt_scene_free(scene);
  };
  for (int i = 0; i < n_gpus; i++) {
    rc = rrt_create(i, desc, precision, &handles[i]);
    if (rc != RRT_OK) { const int e = fail("rrt_create", rc); cleanup(); return e; }
    if (i == 0) for (size_t k = 0; k < rrt_warning_count(handles[i]); k++) std::fprintf(stderr, "%s\n", rrt_warning(handles[i], k));
    if (hipSetDevice(i) != hipSuccess || hipMalloc(&films[i], film_bytes) != hipSuccess || hipMemset(films[i], 0, film_bytes) != hipSuccess) {
      std::fprintf(stderr, "rrt_render: cannot allocate the %zu-byte film on device %d\n", film_bytes, i);
      cleanup();
      return 1;
    }
  }
  // every GPU renders its interleaved bands at the same time (the calls only enqueue), then one collective, then wait
  for (int i = 0; i < n_gpus; i++) {
    rc = rrt_render_bands_begin(handles[i], i, n_gpus, films[i]);
    if (rc != RRT_OK) { const int e = fail("rrt_render_bands_begin", rc); cleanup(); return e; }
  }
  rc = rrt_film_gather_all(handles.data(), films.data(), n_gpus, 0);
  if (rc != RRT_OK) { const int e = fail("rrt_film_gather_all", rc); cleanup(); return e; }
  unsigned long long rays_generated = 0;
  for (int i = n_gpus - 1; i >= 0; i--) {   // rank 0 last: its stream carries the receiving half of the collective
    rrt_render_stats st;
    rc = rrt_render_end_stats(handles[i], &st);
    if (rc != RRT_OK) { const int e = fail("rrt_render_end", rc); cleanup(); return e; }
    rays_generated += st.camera_rays;
  }
  std::printf("%llu rays generated\n", rays_generated);   // integrator/mod.rs:137 (camera samples with weight > 0, over all tiles)
  std::vector<unsigned char> host(film_bytes), rgba((size_t)W * (size_t)H * 4);
  if (hipSetDevice(0) != hipSuccess || hipMemcpy(host.data(), films[0], film_bytes, hipMemcpyDeviceToHost) != hipSuccess) {
    std::fprintf(stderr, "rrt_render: film copy-out failed\n");
    cleanup();
    return 1;
  }
  rc = rrt_resolve_rgba8(host.data(), precision, W, H, film_scale, rgba.data());
  if (rc == RRT_OK) rc = rrt_write_png(argv[2], rgba.data(), W, H);
  if (rc != RRT_OK) { const int e = fail("write_image", rc); cleanup(); return e; }
  cleanup();
  return 0;
}
