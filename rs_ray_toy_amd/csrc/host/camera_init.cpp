// Host-side RealisticCamera initialisation, following /root/reference/src/camera.rs:66-135:
// lens table ingest, thick-lens focusing (focus_thick_lens :332-358) and exit-pupil bounding
// (bound_exit_pupil :421-480). The per-sample work (generate_ray_differential) is a device kernel.
//
// Deliberate economies, all result-invariant for the render:
//  * focus_binary_search (:359-380) is only printed by the reference (camera.rs:111-112) -> not run;
//  * sample_exit_pupil indexes exit_pupil_bounds with `(r/(diag/2)) as usize * 64` (Q6), which can only
//    be 0 or (clamped) 63, so only those two slabs are bounded (the reference bounds all 64).
#include <thread>

#include "scene.hpp"

namespace rrt {
namespace {

struct RayH { V3 o, d; };
RayH ray_new(V3 o, V3 d) { return {o, normalize(d)}; }  // Ray::new / new_od geometry.rs:1841-1858

struct Lens {
  std::vector<rrt_lens_elem> e;
  double film_diagonal;

  double lens_rear_z() const { return e.back().thickness; }
  double lens_front_z() const { double z = 0; for (auto& x : e) z += x.thickness; return z; }
  double rear_element_radius() const { return e.back().aperture_radius; }

  static RayH flip_z(const RayH& r) {  // Transform::scale(1,1,-1).t(ray): transform.rs:525-537
    // the 4x4 products reduce exactly to (x, y, -z); Ray::new then normalises d a second time
    V3 o{r.o.x, r.o.y, -r.o.z};
    V3 d{r.d.x, r.d.y, -r.d.z};
    return ray_new(o, normalize(d));
  }

  // refract reflection.rs:122-134
  static bool refract(V3 wi, V3 n, double eta, V3* wt) {
    double cos_i = dot(n, wi);
    double sin2_i = std::fmax(0.0, 1.0 - cos_i * cos_i);
    double sin2_t = eta * eta * sin2_i;
    if (sin2_t >= 1.0) return false;
    double cos_t = std::sqrt(1.0 - sin2_t);
    *wt = (-wi) * eta + n * (eta * cos_i - cos_t);
    return true;
  }

  // intersect_spherical_element camera.rs:220-253
  static bool intersect_spherical(double radius, double z_center, const RayH& ray, double* t, V3* n) {
    V3 o = ray.o - V3{0.0, 0.0, z_center};
    double a = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
    double b = 2.0 * (ray.d.x * o.x + ray.d.y * o.y + ray.d.z * o.z);
    double c = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    double t0 = 0, t1 = 0;
    if (!quadratic(a, b, c, &t0, &t1)) return false;
    bool use_closer = (ray.d.z > 0.0) ^ (radius < 0.0);
    *t = use_closer ? std::fmin(t0, t1) : std::fmax(t0, t1);
    if (*t < 0.0) return false;
    V3 nn = o + ray.d * *t;
    nn = nn / length(nn);  // Normal3::normalize geometry.rs:1209-1211 (no zero guard)
    V3 md = -ray.d;
    if (dot(nn, md) < 0.0) nn = -nn;  // faceforward geometry.rs:1381-1387
    *n = nn;
    return true;
  }

  // trace_lenses_from_film camera.rs:156-219
  bool trace_from_film(const RayH& r_camera, RayH* out) const {
    double element_z = 0.0;
    RayH r = flip_z(r_camera);
    for (int i = (int)e.size() - 1; i >= 0; i--) {
      const rrt_lens_elem& el = e[i];
      element_z -= el.thickness;
      double t = 0.0;
      V3 n;
      bool is_stop = el.curvature_radius == 0.0;
      if (is_stop) {
        if (r.d.z >= 0.0) return false;
        t = (element_z - r.o.z) / r.d.z;
      } else {
        double radius = el.curvature_radius, z_center = element_z + el.curvature_radius;
        if (!intersect_spherical(radius, z_center, r, &t, &n)) return false;
      }
      if (!(t >= 0.0)) throw Panic("camera.rs:186 assert!(t >= 0)");
      V3 p_hit = r.o + r.d * t;
      double r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
      if (r2 >= el.aperture_radius * el.aperture_radius) return false;
      r.o = p_hit;
      if (!is_stop) {
        V3 w;
        double eta_i = el.eta;
        double eta_t = (i > 0 && e[i - 1].eta != 0.0) ? e[i - 1].eta : 1.0;
        if (!refract(normalize(-r.d), n, eta_i / eta_t, &w)) return false;
        r.d = w;
      }
    }
    if (out) *out = flip_z(r);
    return true;
  }

  // trace_lenses_from_scene camera.rs:254-306
  bool trace_from_scene(const RayH& r_camera, RayH* out) const {
    double element_z = -lens_front_z();
    RayH r = flip_z(r_camera);
    for (size_t i = 0; i < e.size(); i++) {
      const rrt_lens_elem& el = e[i];
      double t = 0.0;
      V3 n;
      bool is_stop = el.curvature_radius == 0.0;
      if (is_stop) {
        t = (element_z - r.o.z) / r.d.z;
      } else {
        double radius = el.curvature_radius, z_center = element_z + el.curvature_radius;
        if (!intersect_spherical(radius, z_center, r, &t, &n)) return false;
      }
      if (!(t >= 0.0)) throw Panic("camera.rs:273 assert!(t >= 0)");
      V3 p_hit = r.o + r.d * t;
      double r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
      if (r2 >= el.aperture_radius * el.aperture_radius) return false;
      r.o = p_hit;
      if (!is_stop) {
        V3 wt;
        double eta_i = (i == 0 || e[i - 1].eta == 0.0) ? 1.0 : e[i - 1].eta;
        double eta_t = (el.eta != 0.0) ? el.eta : 1.0;
        if (!refract(-normalize(r.d), n, eta_i / eta_t, &wt)) return false;
        r.d = wt;
      }
      element_z += el.thickness;
    }
    if (out) *out = flip_z(r);
    return true;
  }

  // compute_cardinal_points camera.rs:317-324
  static void cardinal(const RayH& r_in, const RayH& r_out, double* pz, double* fz) {
    double tf = -r_out.o.x / r_out.d.x;
    *fz = -(r_out.o + r_out.d * tf).z;
    double tp = (r_in.o.x - r_out.o.x) / r_out.d.x;
    *pz = -(r_out.o + r_out.d * tp).z;
  }

  // focus_thick_lens camera.rs:332-358 over compute_thick_lens_approximation :325-331
  double focus_thick_lens(double focus_distance) const {
    double x = 0.001 * film_diagonal;
    RayH r_scene = ray_new({x, 0.0, lens_front_z() + 1.0}, {0.0, 0.0, -1.0});
    RayH r_film;
    if (!trace_from_scene(r_scene, &r_film))
      throw Panic("camera.rs:334 Unable to trace ray from scene to film for thick lens approximation");
    double pz[2], fz[2];
    cardinal(r_scene, r_film, &pz[0], &fz[0]);
    RayH r_film2 = ray_new({x, 0.0, lens_rear_z() - 1.0}, {0.0, 0.0, 1.0});
    RayH r_scene2;
    if (!trace_from_film(r_film2, &r_scene2))
      throw Panic("camera.rs:342 Unable to trace ray from film to scene for thick lens approximation");
    cardinal(r_film2, r_scene2, &pz[1], &fz[1]);
    double f = fz[0] - pz[0];
    double z = -focus_distance;
    double c = (pz[1] - z - pz[0]) * (pz[1] - z - 4.0 * f - pz[0]);
    if (!(c > 0.0)) throw Panic("camera.rs:364 Coefficient must be positive (focusDistance too short for the lens)");
    double delta = 0.5 * (pz[1] - z + pz[0] - std::sqrt(c));
    return e.back().thickness + delta;
  }

  // bound_exit_pupil camera.rs:421-480 (incl. Q7: zero-box default, expand() that shifts)
  void bound_exit_pupil(double x0, double x1, double out[4]) const {
    const uint64_t n_samples = 1024 * 1024;
    const double rear_radius = rear_element_radius();
    const double pr_min = -1.5 * rear_radius, pr_max = 1.5 * rear_radius;
    const double rear_z = lens_rear_z();
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    struct Acc { double mnx = 0, mny = 0, mxx = 0, mxy = 0; uint64_t exiting = 0; };  // Bounds2f::default = zero box
    std::vector<Acc> acc(n_threads);
    std::vector<std::thread> th;
    std::string panic_msg;
    for (unsigned t = 0; t < n_threads; t++) {
      th.emplace_back([&, t]() {
        Acc a;
        try {
          for (uint64_t i = t; i < n_samples; i += n_threads) {
            V3 p_film{lerp(((double)i + 0.5) / (double)n_samples, x0, x1), 0.0, 0.0};
            double u0 = radical_inverse_host(0, i), u1 = radical_inverse_host(1, i);
            V3 p_rear{lerp(u0, pr_min, pr_max), lerp(u1, pr_min, pr_max), rear_z};
            // `inside(p, bounds) || trace(..)`: the short-circuit never changes the union, so every
            // sample is traced here and the bounds reduced afterwards (order independent min/max).
            if (trace_from_film(ray_new(p_film, p_rear - p_film), nullptr)) {
              a.mnx = fmin_ref(a.mnx, p_rear.x); a.mny = fmin_ref(a.mny, p_rear.y);
              a.mxx = fmax_ref(a.mxx, p_rear.x); a.mxy = fmax_ref(a.mxy, p_rear.y);
              a.exiting++;
            }
          }
        } catch (const std::exception& ex) {
          panic_msg = ex.what();
        }
        acc[t] = a;
      });
    }
    for (auto& x : th) x.join();
    if (!panic_msg.empty()) throw Panic(panic_msg);
    Acc r;
    for (auto& a : acc) {
      r.mnx = fmin_ref(r.mnx, a.mnx); r.mny = fmin_ref(r.mny, a.mny);
      r.mxx = fmax_ref(r.mxx, a.mxx); r.mxy = fmax_ref(r.mxy, a.mxy);
      r.exiting += a.exiting;
    }
    if (r.exiting == 0) { out[0] = pr_min; out[1] = pr_min; out[2] = pr_max; out[3] = pr_max; return; }
    // pupil_bounds.expand(2 * |diag(proj_rear_bounds)| / sqrt(n)); Bounds2::expand geometry.rs:1448-1454
    // is `new(p_min - delta, p_max - delta)` (Q7)
    double dx = pr_max - pr_min, dy = pr_max - pr_min;
    double delta = 2.0 * std::sqrt(dx * dx + dy * dy) / std::sqrt((double)n_samples);
    double ax = r.mnx - delta, ay = r.mny - delta, bx = r.mxx - delta, by = r.mxy - delta;
    out[0] = ax > bx ? bx : ax; out[1] = ay > by ? by : ay;
    out[2] = ax > bx ? ax : bx; out[3] = ay > by ? ay : by;
  }
};

}  // namespace

void init_camera(SceneData& s, const Xf& camera_to_world, double shutter_open, double shutter_close,
                 double aperture_diameter, double focus_distance, const std::vector<double>& lens_data,
                 bool simple_weighting) {
  if (lens_data.size() % 4 != 0) throw Panic("camera.rs:77 assert!(lens_data.len() % 4 == 0)");
  if (lens_data.empty()) throw Panic("camera.rs:139 Error Getting Last Lens Element");
  Lens L;
  L.film_diagonal = s.desc.film.diagonal;
  for (size_t idx = 0; idx < lens_data.size() / 4; idx++) {
    size_t i = idx * 4;
    double aperture_radius = lens_data[i + 3];
    if (lens_data[i] == 0.0) {
      if (!(aperture_diameter > lens_data[i + 3])) aperture_radius = aperture_diameter;
    }
    L.e.push_back({lens_data[i] * 0.001, lens_data[i + 1] * 0.001, lens_data[i + 2], aperture_radius * 0.001 / 2.0});
  }
  double thick = L.focus_thick_lens(focus_distance);
  L.e.back().thickness = thick;
  s.lens = L.e;
  rrt_camera& c = s.desc.camera;
  xf_to_abi(camera_to_world, &c.camera_to_world);
  c.shutter_open = shutter_open;
  c.shutter_close = shutter_close;
  c.simple_weighting = simple_weighting ? 1 : 0;
  c.n_elems = (int32_t)s.lens.size();
  memset(c.exit_pupil_bounds, 0, sizeof(c.exit_pupil_bounds));
  memset(c.exit_pupil_valid, 0, sizeof(c.exit_pupil_valid));
  const int n_slabs = 64;
  for (int i : {0, n_slabs - 1}) {
    double r0 = (double)i / (double)n_slabs * L.film_diagonal / 2.0;
    double r1 = (double)(i + 1) / (double)n_slabs * L.film_diagonal / 2.0;
    L.bound_exit_pupil(r0, r1, c.exit_pupil_bounds[i]);
    c.exit_pupil_valid[i] = 1;
  }
}

}  // namespace rrt
