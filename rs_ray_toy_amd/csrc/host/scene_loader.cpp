// scene.json loader: the host half of deploy_render (/root/reference/src/renderprocess.rs:92-105), i.e.
// make_scene (:254-284) + make_integrator (:1399-1499), with every key / default as the reference reads
// it (read_i64/read_f64/read_bool/read_string :136-169). Output is the flat rrt_scene_desc both the HIP
// executor and the oracle consume. Out-of-scope features fail loudly with RRT_EUNSUP when a
// scene actually *uses* them; declared-but-unused ones are only noted, as the reference would load them.
#include <fstream>
#include <functional>
#include <memory>
#include <map>
#include <sstream>

#include "json.hpp"
#include "scene.hpp"

namespace rrt {
const char* last_error_cstr();

void SceneData::finalize() {
  desc.abi_version = RRT_ABI_VERSION;
  desc.positions = positions.data(); desc.n_positions = positions.size() / 3;
  desc.normals = normals.data(); desc.n_normals = normals.size() / 3;
  desc.uvs = uvs.data(); desc.n_uvs = uvs.size() / 2;
  desc.tris = tris.data(); desc.n_tris = tris.size();
  desc.spheres = spheres.data(); desc.n_spheres = spheres.size();
  desc.xforms = xforms.data(); desc.n_xforms = xforms.size();
  desc.prims = prims.data(); desc.n_prims = prims.size();
  desc.materials = materials.data(); desc.n_materials = materials.size();
  desc.textures = textures.data(); desc.n_textures = textures.size();
  desc.images = images.data(); desc.n_images = images.size();
  desc.image_texels = image_texels.data(); desc.n_image_texels = image_texels.size() / 3;
  desc.lights = lights.data(); desc.n_lights = lights.size();
  desc.bvh_nodes = bvh_nodes.data(); desc.n_bvh_nodes = bvh_nodes.size();
  desc.prim_order = prim_order.data(); desc.n_prim_order = prim_order.size();
  desc.camera.elems = lens.data();
  desc.camera.n_elems = (int32_t)lens.size();
  desc.sampler.perms = perms.data();
  desc.sampler.n_perms = perms.size();
}

namespace {

// ---- read_* helpers, renderprocess.rs:136-169 ----
int64_t read_i64(const Json& root, const char* key, int64_t dflt) {
  const Json* v = root.get(key);
  if (v && v->kind == Json::Num && v->is_int) return v->inum;
  return dflt;
}
double read_f64(const Json& root, const char* key, double dflt) {
  const Json* v = root.get(key);
  if (v && v->kind == Json::Num) return v->num;
  return dflt;
}
bool read_bool(const Json& root, const char* key, bool dflt) {
  const Json* v = root.get(key);
  if (v && v->kind == Json::Bool) return v->b;
  return dflt;
}
std::string read_string(const Json& root, const char* key, const std::string& dflt) {
  const Json* v = root.get(key);
  if (v && v->kind == Json::Str) return v->str;
  return dflt;
}
// read_num_array renderprocess.rs:171-198
bool read_num_array(const Json& v, size_t desired, std::vector<double>* out) {
  if (v.kind != Json::Arr) return false;
  size_t len = desired == 0 ? v.arr.size() : desired;
  if (v.arr.size() != len) return false;
  out->clear();
  for (auto& n : v.arr) {
    if (n.kind != Json::Num) return false;
    out->push_back(n.num);
  }
  return true;
}
V3 fetch_v3(const Json& cfg, const char* key, V3 dflt) {  // fetch_point3f / fetch_vector3f :214-232
  const Json* v = cfg.get(key);
  std::vector<double> a;
  if (v && read_num_array(*v, 3, &a)) return {a[0], a[1], a[2]};
  return dflt;
}
// make_to_world renderprocess.rs:242-252: T * R * S
Xf make_to_world(const Json& root) {
  V3 world_pos = fetch_v3(root, "world_pos", {0, 0, 0});
  V3 axis = normalize(fetch_v3(root, "rotation_axis", {0, 0, 0}));
  double angle = read_f64(root, "rotation_angle", 0.0);
  V3 sc = fetch_v3(root, "scale", {1, 1, 1});
  return xf_mul(xf_mul(xf_translate(world_pos), xf_rotate(angle, axis)), xf_scale(sc.x, sc.y, sc.z));
}

struct Rgb { double c[3]; };
Rgb rgb1(double v) { return {{v, v, v}}; }

// make_spectrum renderprocess.rs:1055-1077
Rgb make_spectrum(const Json& cfg, const char* key, double dflt) {
  const Json* sc = cfg.get(key);
  if (sc) {
    const Json* vals = sc->get("values");
    if (vals) {
      std::vector<double> a;
      if (!read_num_array(*vals, 3, &a)) throw Panic(std::string("renderprocess.rs:1066 Failed to parse Spectrum for key ") + key);
      return {{a[0], a[1], a[2]}};
    }
  }
  return rgb1(dflt);
}

// Texture tables (make_textures renderprocess.rs:298-515). Every declared texture becomes a node of the flat graph
// in rrt_scene_desc.textures; `node` < 0 marks one that cannot be represented (ImageTexture, or a graph that
// contains one): using it from a material is RRT_EUNSUP, declaring it is not. `is_const` textures (the value is the
// same at every hit) are folded into the material's constant, as before.
struct TexF { bool is_const = false; double v = 0; std::string type; int node = -1; };
struct TexC { bool is_const = false; Rgb v{{0, 0, 0}}; std::string type; int node = -1; };

struct Loader {
  SceneData& s;
  std::string root_dir;
  uint32_t flags;
  uint64_t seed;
  std::map<std::string, TexF> float_tex;
  std::map<std::string, TexC> rgb_tex;
  std::vector<bool> tex_ok;   // per s.textures entry: representable (every image in its graph decodable here)
  std::map<int, std::string> tex_why;
  std::map<std::string, int> image_cache;   // `images: HashMap<TexInfo, Arc<MIPMap>>` renderprocess.rs:305
  struct MatEntry { int index = -1; std::string unsupported_type; };
  std::map<std::string, MatEntry> materials;
  struct MeshEntry { uint32_t first_tri = 0, n_tris = 0; };
  std::map<std::string, MeshEntry> meshes;

  void warn(const std::string& m) { s.warnings.push_back(m); }

  std::string asset_path(const std::string& passed) {  // preprocess_filepath :122-128
    std::string t = passed;
    while (t.rfind("./", 0) == 0) t = t.substr(2);
    for (auto& ch : t) if (ch == '\\') ch = '/';
    return root_dir + "/" + t;
  }

  // make_texture_mapping_2d renderprocess.rs:561-610 (defaults du = dv = 1.0 as read there)
  void make_mapping_2d(const Json& tc, const Xf& to_world, rrt_texture* t) {
    const Json* mc = tc.get("mapping");
    t->mapping = RRT_MAP_UV;
    t->map[0] = 1.0; t->map[1] = 1.0; t->map[2] = 0.0; t->map[3] = 0.0;   // UVMapping2D::new(1, 1, 0, 0) when "mapping" is absent :608
    if (!mc) return;
    std::string type = read_string(*mc, "mapping", "uv");
    if (type == "uv") {
      t->map[0] = read_f64(*mc, "su", 1.0); t->map[1] = read_f64(*mc, "sv", 1.0);
      t->map[2] = read_f64(*mc, "du", 1.0); t->map[3] = read_f64(*mc, "dv", 1.0);
    } else if (type == "spherical" || type == "cylindrical") {
      t->mapping = type == "spherical" ? RRT_MAP_SPHERICAL : RRT_MAP_CYLINDRICAL;
      Xf inv = xf_inverse(to_world);
      for (int k = 0; k < 16; k++) t->world_to_texture[k] = inv.m.m[k / 4][k % 4];
    } else if (type == "planar") {
      t->mapping = RRT_MAP_PLANAR;
      V3 v1 = fetch_v3(*mc, "v1", {1, 0, 0}), v2 = fetch_v3(*mc, "v2", {0, 1, 0});
      t->vs[0] = v1.x; t->vs[1] = v1.y; t->vs[2] = v1.z; t->vt[0] = v2.x; t->vt[1] = v2.y; t->vt[2] = v2.z;
      t->map[0] = read_f64(*mc, "udelta", 0.0); t->map[1] = read_f64(*mc, "vdelta", 0.0);
    } else throw Panic("renderprocess.rs:601 Unsupported Mapping Type " + type);
  }
  void identity_mapping_3d(const Xf& to_world, rrt_texture* t) {   // IdentityMapping3D::new(to_world): not inverted
    t->mapping = RRT_MAP_IDENTITY3D;
    for (int k = 0; k < 16; k++) t->world_to_texture[k] = to_world.m.m[k / 4][k % 4];
  }
  int add_texture(const rrt_texture& t, bool representable) {
    s.textures.push_back(t);
    tex_ok.push_back(representable);
    return (int)s.textures.size() - 1;
  }

  // make_textures renderprocess.rs:298-515
  void make_textures(const Json& cfg) {
    const Json* ft = cfg.get("float_texture");
    if (ft && ft->kind == Json::Arr) {
      for (auto& tc : ft->arr) {
        Xf to_world = make_to_world(tc);
        std::string type = read_string(tc, "texture_type", ""), name = read_string(tc, "texture_name", "DefaultTextureName");
        TexF t; t.type = type;
        rrt_texture n{};
        n.child[0] = n.child[1] = n.child[2] = -1;
        bool ok = true;
        auto fallback = [&](int slot, const std::string& nm, double d) {
          auto it = float_tex.find(nm);
          TexF r = it != float_tex.end() ? it->second : TexF{true, d, "Constant", -1};
          for (int k = 0; k < 3; k++) n.fallback[slot][k] = r.v;
          if (it != float_tex.end()) { n.child[slot] = r.node; if (r.node < 0 || !tex_ok[r.node]) ok = false; }
          return r;
        };
        if (type == "MixTexture") {
          n.type = RRT_TEX_MIX;
          TexF t1 = fallback(0, read_string(tc, "t1", "ErrorTextureName"), 0.0), t2 = fallback(1, read_string(tc, "t2", "ErrorTextureName"), 1.0);
          TexF am = fallback(2, read_string(tc, "t2", "ErrorTextureName"), 0.5);  // reads "t2" for amount, :318
          if (t1.is_const && t2.is_const && am.is_const) { t.is_const = true; t.v = t1.v * (1.0 - am.v) + t2.v * am.v; }
        } else if (type == "BilerpTexture") {
          n.type = RRT_TEX_BILERP;
          make_mapping_2d(tc, to_world, &n);
          double v00 = read_f64(tc, "v00", 0.0), v01 = read_f64(tc, "v01", 1.0), v10 = read_f64(tc, "v01", 0.0), v11 = read_f64(tc, "v01", 1.0);
          for (int k = 0; k < 3; k++) { n.v[0][k] = v00; n.v[1][k] = v01; n.v[2][k] = v10; n.v[3][k] = v11; }
          if (v00 == v01 && v01 == v10 && v10 == v11) { t.is_const = true; t.v = v00; }
        } else if (type == "CheckerBoardTexture") {
          int64_t dim = read_i64(tc, "dimension", 2);
          if (dim != 2 && dim != 3) { warn(std::to_string(dim) + " dimensional checkerboard texture not supported"); continue; }
          fallback(0, read_string(tc, "t1", "ErrorTextureName"), 1.0); fallback(1, read_string(tc, "t2", "ErrorTextureName"), 0.0);
          if (dim == 2) {
            n.type = RRT_TEX_CHECKER2D;
            make_mapping_2d(tc, to_world, &n);
            n.aa_none = read_string(tc, "aamode", "closedform") == "none";
          } else { n.type = RRT_TEX_CHECKER3D; identity_mapping_3d(to_world, &n); }
        } else if (type == "ScaleTexture") {
          n.type = RRT_TEX_SCALE;
          TexF t1 = fallback(0, read_string(tc, "t1", "ErrorTextureName"), 1.0), t2 = fallback(1, read_string(tc, "t2", "ErrorTextureName"), 1.0);
          if (t1.is_const && t2.is_const) { t.is_const = true; t.v = t1.v * t2.v; }
        } else if (type == "WindyTexture") {
          n.type = RRT_TEX_WINDY; identity_mapping_3d(to_world, &n);
        } else if (type == "WrinkledTexture") {
          n.type = RRT_TEX_WRINKLED; identity_mapping_3d(to_world, &n);
          n.octaves = (int32_t)read_i64(tc, "octaves", 8); n.omega = read_f64(tc, "omega", 0.5);
        } else { warn("Unsupported Texture Type " + type); continue; }
        t.node = add_texture(n, ok);
        float_tex[name] = t;
      }
    }
    const Json* rt = cfg.get("rgb_texture");
    if (rt && rt->kind == Json::Arr) {
      for (auto& tc : rt->arr) {
        Xf to_world = make_to_world(tc);
        std::string type = read_string(tc, "texture_type", ""), name = read_string(tc, "texture_name", "DefaultTextureName");
        TexC t; t.type = type;
        rrt_texture n{};
        n.child[0] = n.child[1] = n.child[2] = -1;
        bool ok = true;
        auto fallback = [&](int slot, const std::string& nm, double d) {
          auto it = rgb_tex.find(nm);
          TexC r = it != rgb_tex.end() ? it->second : TexC{true, rgb1(d), "Constant", -1};
          for (int k = 0; k < 3; k++) n.fallback[slot][k] = r.v.c[k];
          if (it != rgb_tex.end()) { n.child[slot] = r.node; if (r.node < 0 || !tex_ok[r.node]) ok = false; }
          return r;
        };
        if (type == "MixTexture") {
          n.type = RRT_TEX_MIX;
          TexC t1 = fallback(0, read_string(tc, "t1", "ErrorTextureName"), 0.0), t2 = fallback(1, read_string(tc, "t2", "ErrorTextureName"), 1.0);
          auto it = float_tex.find(read_string(tc, "t2", "ErrorTextureName"));   // amount: a *float* texture named by "t2", :412-414
          TexF am = it != float_tex.end() ? it->second : TexF{true, 0.5, "Constant", -1};
          for (int k = 0; k < 3; k++) n.fallback[2][k] = am.v;
          if (it != float_tex.end()) { n.child[2] = am.node; if (am.node < 0 || !tex_ok[am.node]) ok = false; }
          if (t1.is_const && t2.is_const && am.is_const) { t.is_const = true; for (int k = 0; k < 3; k++) t.v.c[k] = t1.v.c[k] * (1.0 - am.v) + t2.v.c[k] * am.v; }
        } else if (type == "UVTexture") {
          n.type = RRT_TEX_UV;
          make_mapping_2d(tc, to_world, &n);
        } else if (type == "BilerpTexture") {
          n.type = RRT_TEX_BILERP;
          make_mapping_2d(tc, to_world, &n);
          Rgb v00 = make_spectrum(tc, "v00", 0.0), v01 = make_spectrum(tc, "v01", 1.0), v10 = make_spectrum(tc, "v01", 0.0), v11 = make_spectrum(tc, "v01", 1.0);
          bool same = true;
          for (int k = 0; k < 3; k++) {
            n.v[0][k] = v00.c[k]; n.v[1][k] = v01.c[k]; n.v[2][k] = v10.c[k]; n.v[3][k] = v11.c[k];
            same = same && v00.c[k] == v01.c[k] && v01.c[k] == v10.c[k] && v10.c[k] == v11.c[k];
          }
          if (same) { t.is_const = true; t.v = v00; }
        } else if (type == "ScaleTexture") {
          n.type = RRT_TEX_SCALE;
          TexC t1 = fallback(0, read_string(tc, "t1", "ErrorTextureName"), 1.0), t2 = fallback(1, read_string(tc, "t2", "ErrorTextureName"), 1.0);
          if (t1.is_const && t2.is_const) { t.is_const = true; for (int k = 0; k < 3; k++) t.v.c[k] = t1.v.c[k] * t2.v.c[k]; }
        } else if (type == "CheckerBoardTexture") {
          int64_t dim = read_i64(tc, "dimension", 2);
          if (dim != 2 && dim != 3) { warn(std::to_string(dim) + " dimensional checkerboard texture not supported"); continue; }
          fallback(0, read_string(tc, "t1", "ErrorTextureName"), 1.0); fallback(1, read_string(tc, "t2", "ErrorTextureName"), 0.0);
          if (dim == 2) {
            n.type = RRT_TEX_CHECKER2D;
            make_mapping_2d(tc, to_world, &n);
            n.aa_none = read_string(tc, "aamode", "closedform") == "none";
          } else { n.type = RRT_TEX_CHECKER3D; identity_mapping_3d(to_world, &n); }
        } else if (type == "ImageTexture") {   // :421-437, make_tex_info :517-530, load_image :532-561
          n.type = RRT_TEX_IMAGE;
          make_mapping_2d(tc, to_world, &n);
          const std::string filename = asset_path(read_string(tc, "filename", "DefaultTexture"));
          const bool do_trilinear = read_bool(tc, "do_trilinear", false);
          const double max_aniso = read_f64(tc, "max_aniso", 8.0);
          const std::string wrap_s = read_string(tc, "wrap", "repeat");
          const int wrap = wrap_s == "black" ? RRT_WRAP_BLACK : (wrap_s == "clamp" ? RRT_WRAP_CLAMP : RRT_WRAP_REPEAT);
          // ("scale" and "gamma" are read into TexInfo and never applied)
          const std::string key = filename + "|" + (do_trilinear ? "1" : "0") + "|" + std::to_string(max_aniso) + "|" + std::to_string(wrap);
          auto it = image_cache.find(key);
          if (it != image_cache.end()) n.image = it->second;
          else {
            uint32_t w = 0, h = 0;
            std::vector<uint8_t> rgb;
            std::string why;
            const int rc = decode_png_rgb8(filename, &w, &h, &rgb, &why);
            if (rc == 1) { warn("ImageTexture " + name + ": image not loadable (" + why + "), texture not registered"); continue; }  // load_image Err -> not inserted
            if (rc == 2) { ok = false; tex_why[(int)s.textures.size()] = why; n.image = -1; }
            else { n.image = build_mipmap(s, w, h, rgb, do_trilinear, max_aniso, wrap); image_cache[key] = n.image; }
          }
        } else if (type == "WindyTexture") {
          n.type = RRT_TEX_WINDY; identity_mapping_3d(to_world, &n);
        } else if (type == "WrinkledTexture") {
          n.type = RRT_TEX_WRINKLED; identity_mapping_3d(to_world, &n);
          n.octaves = (int32_t)read_i64(tc, "octaves", 8); n.omega = read_f64(tc, "omega", 0.5);
        } else { warn("Unsupported Texture Type " + type); continue; }
        t.node = add_texture(n, ok);
        rgb_tex[name] = t;
      }
    }
  }

  // fetch_rgb_texture renderprocess.rs:644-661. `slot` = RRT_P_*: a non-constant texture is recorded in cur->tex[slot]
  // (the returned constant is then unused).
  rrt_material* cur = nullptr;
  void bind_texture(int node, const std::string& type, const std::string& tname, int slot, const char* key, const std::string& mat) {
    if (node < 0 || !tex_ok[node]) {
      std::string why = "an image format the restated decoder does not cover";
      for (auto& kv : tex_why) if (kv.first <= node) why = kv.second;
      throw Unsupported("material '" + mat + "' key '" + key + "' uses " + type + " '" + tname + "' whose graph holds an ImageTexture that cannot be decoded here: " + why);
    }
    cur->tex[slot] = node;
  }
  Rgb fetch_rgb(const Json& mc, const char* key, Rgb dflt, const std::string& mat, int slot) {
    const Json* v = mc.get(key);
    if (v && v->kind == Json::Str) {
      auto it = rgb_tex.find(v->str);
      if (it != rgb_tex.end()) {
        if (!it->second.is_const) bind_texture(it->second.node, it->second.type, v->str, slot, key, mat);
        return it->second.v;
      }
    }
    return dflt;
  }
  // fetch_float_texture :612-625 (HashMap index panics when the name is missing)
  double fetch_float(const Json& mc, const char* key, double dflt, const std::string& mat, int slot) {
    const Json* v = mc.get(key);
    if (v && v->kind == Json::Str) {
      auto it = float_tex.find(v->str);
      if (it == float_tex.end()) throw Panic("renderprocess.rs:619 float_texture[\"" + v->str + "\"] missing (material " + mat + ")");
      if (!it->second.is_const) {
        bind_texture(it->second.node, it->second.type, v->str, slot, key, mat);
      }
      return it->second.v;
    }
    return dflt;
  }
  bool fetch_float_opt(const Json& mc, const char* key, double* out, const std::string& mat, int slot) {  // :627-642
    const Json* v = mc.get(key);
    if (v && v->kind == Json::Str) { *out = fetch_float(mc, key, 0.0, mat, slot); return true; }
    return false;
  }

  // make_materials renderprocess.rs:664-871
  void make_materials(const Json& cfg) {
    const Json* arr = cfg.get("materials");
    if (!arr || arr->kind != Json::Arr) return;
    // MetalMaterial defaults: COPPER_N / COPPER_K = Spectrum::from_sampled(...) (metal.rs:167-178) evaluated
    // once over the reference's CIE tables (SURVEY §8c known answers).
    const Rgb copper_n{{0.19998972096819712, 0.922085788777433, 1.0998762520488314}};
    const Rgb copper_k{{3.9046381767086675, 2.4476332238684626, 2.1376510366555137}};
    for (auto& mc : arr->arr) {
      std::string type = read_string(mc, "material_type", ""), name = read_string(mc, "material_name", "DefaultMaterialName");
      rrt_material m{};
      for (int k = 0; k < RRT_P_COUNT; k++) m.tex[k] = -1;
      cur = &m;
      m.bump = -1;
      auto no_bump = [&]() {   // fetch_float_texture_opt(.., "bump_map", None) :627-642: any float texture, constant ones included
        const Json* v = mc.get("bump_map");
        if (!(v && v->kind == Json::Str)) return;
        auto it = float_tex.find(v->str);
        if (it == float_tex.end()) throw Panic("renderprocess.rs:634 float_texture[\"" + v->str + "\"] missing (material " + name + ")");
        if (it->second.node < 0 || !tex_ok[it->second.node]) throw Unsupported("material '" + name + "': bump_map '" + v->str + "' holds an ImageTexture that cannot be decoded here");
        m.bump = it->second.node;
      };
      auto put = [&](const Rgb& r, double* dst) { for (int k = 0; k < 3; k++) dst[k] = r.c[k]; };
      if (type == "MatteMaterial") {
        m.type = RRT_MAT_MATTE;
        put(fetch_rgb(mc, "kd", rgb1(0.5), name, RRT_P_KD), m.kd);
        m.sigma = fetch_float(mc, "sigma", 0.0, name, RRT_P_SIGMA);
        no_bump();
      } else if (type == "PlasticMaterial") {
        m.type = RRT_MAT_PLASTIC;
        put(fetch_rgb(mc, "kd", rgb1(0.25), name, RRT_P_KD), m.kd);
        put(fetch_rgb(mc, "ks", rgb1(0.25), name, RRT_P_KS), m.ks);
        m.roughness = fetch_float(mc, "roughness", 0.1, name, RRT_P_ROUGHNESS);
        no_bump();
        m.remap_roughness = read_bool(mc, "remap_roughness", false);
      } else if (type == "MetalMaterial") {
        m.type = RRT_MAT_METAL;
        put(fetch_rgb(mc, "eta", copper_n, name, RRT_P_ETA), m.eta);
        put(fetch_rgb(mc, "k", copper_k, name, RRT_P_K), m.k);
        m.roughness = fetch_float(mc, "roughness", 0.01, name, RRT_P_ROUGHNESS);
        // metal.rs:60-71: u / v roughness evaluate their own texture when the key is present, else `roughness`
        m.u_roughness = m.roughness; m.v_roughness = m.roughness;
        m.tex[RRT_P_UROUGHNESS] = m.tex[RRT_P_VROUGHNESS] = m.tex[RRT_P_ROUGHNESS];
        double r = 0.0;
        if (mc.get("u_roughness") && mc.get("u_roughness")->kind == Json::Str) { m.tex[RRT_P_UROUGHNESS] = -1; fetch_float_opt(mc, "u_roughness", &r, name, RRT_P_UROUGHNESS); m.u_roughness = r; }
        if (mc.get("v_roughness") && mc.get("v_roughness")->kind == Json::Str) { m.tex[RRT_P_VROUGHNESS] = -1; fetch_float_opt(mc, "v_roughness", &r, name, RRT_P_VROUGHNESS); m.v_roughness = r; }
        no_bump();
        m.remap_roughness = read_bool(mc, "remap_roughness", false);
      } else if (type == "MirrorMaterial") {
        m.type = RRT_MAT_MIRROR;
        put(fetch_rgb(mc, "kr", rgb1(0.9), name, RRT_P_KR), m.kr);
        no_bump();
      } else if (type == "Debug") {
        m.type = RRT_MAT_DEBUG;
      } else if (type == "GlassMaterial") {   // renderprocess.rs:772-798
        m.type = RRT_MAT_GLASS;
        put(fetch_rgb(mc, "kr", rgb1(1.0), name, RRT_P_KR), m.kr);
        put(fetch_rgb(mc, "kt", rgb1(1.0), name, RRT_P_KT), m.kt);
        m.index = fetch_float(mc, "eta", 1.5, name, RRT_P_INDEX);
        m.u_roughness = fetch_float(mc, "u_roughness", 0.0, name, RRT_P_UROUGHNESS);
        m.v_roughness = fetch_float(mc, "v_roughness", 0.0, name, RRT_P_VROUGHNESS);
        no_bump();
        m.remap_roughness = read_bool(mc, "remap_roughness", false);
      } else if (type == "TranslucentMaterial") {   // renderprocess.rs:695-720
        m.type = RRT_MAT_TRANSLUCENT;
        put(fetch_rgb(mc, "kd", rgb1(0.25), name, RRT_P_KD), m.kd);
        put(fetch_rgb(mc, "ks", rgb1(0.25), name, RRT_P_KS), m.ks);
        m.roughness = fetch_float(mc, "roughness", 0.1, name, RRT_P_ROUGHNESS);
        put(fetch_rgb(mc, "reflect", rgb1(0.25), name, RRT_P_REFLECT), m.reflect);
        put(fetch_rgb(mc, "transmit", rgb1(0.25), name, RRT_P_TRANSMIT), m.transmit);
        no_bump();
        m.remap_roughness = read_bool(mc, "remap_roughness", false);
      } else if (type == "DisneyMaterial") {
        materials[name] = MatEntry{-1, type};  // loads in the reference; RRT_EUNSUP only if a primitive uses it
        continue;
      } else if (type == "MixMaterial") {
        // renderprocess.rs:680-697: inserted only if both names are already present, then indexes an
        // empty map (scene_global.materials) -> panic. Unused Mix entries are silently absent.
        std::string m1 = read_string(mc, "mat1", ""), m2 = read_string(mc, "mat2", "");
        if (materials.count(m1) && materials.count(m2)) throw Panic("renderprocess.rs:687 scene_global.materials[mat1] indexes an empty map");
        continue;
      } else {
        warn("Unsupported Material Type " + type);
        continue;
      }
      s.materials.push_back(m);
      materials[name] = MatEntry{(int)s.materials.size() - 1, ""};
    }
  }

  // make_triangle_mesh renderprocess.rs:873-919 (obj-level to_world is computed but never applied: Q13)
  void make_meshes(const Json& cfg) {
    const Json* objs = cfg.get("objs");
    if (!objs || objs->kind != Json::Arr) return;
    for (auto& oc : objs->arr) {
      std::string filename = read_string(oc, "filename", "DefaultObj"), obj_name = read_string(oc, "obj_name", "DefaultObjName");
      std::string path = asset_path(filename);
      ObjMesh m;
      try {
        m = parse_obj(path, s.warnings);
      } catch (const IoError& e) { warn(std::string("parse_result ") + path + " :: " + e.what()); continue; }
      catch (const ParseError& e) { warn(std::string("parse_result ") + path + " :: " + e.what()); continue; }
      uint32_t vbase = (uint32_t)(s.positions.size() / 3), nbase = (uint32_t)(s.normals.size() / 3), uvbase = (uint32_t)(s.uvs.size() / 2);
      s.positions.insert(s.positions.end(), m.p.begin(), m.p.end());
      s.normals.insert(s.normals.end(), m.n.begin(), m.n.end());
      s.uvs.insert(s.uvs.end(), m.uv.begin(), m.uv.end());
      // Triangle::new triangle.rs:73-111: n / uv index triples are [0,0,0] unless both the attribute array
      // and its index array are non-empty. mesh_has_* : 0 = attribute array empty, 1 = array and indices
      // present, 2 = array present but no indices (index 0 used three times).
      bool has_n = !m.n.empty() && !m.ni.empty(), has_uv = !m.uv.empty() && !m.uvi.empty();
      MeshEntry me{(uint32_t)s.tris.size(), (uint32_t)(m.vi.size() / 3)};
      for (size_t t = 0; t < m.vi.size() / 3; t++) {
        rrt_tri tri{};
        for (int k = 0; k < 3; k++) {
          tri.v[k] = vbase + m.vi[3 * t + k];
          tri.n[k] = has_n ? nbase + m.ni[3 * t + k] : nbase;
          tri.uv[k] = has_uv ? uvbase + m.uvi[3 * t + k] : uvbase;
        }
        tri.mesh_has_n = m.n.empty() ? 0 : (has_n ? 1 : 2);
        tri.mesh_has_uv = m.uv.empty() ? 0 : (has_uv ? 1 : 2);
        s.tris.push_back(tri);
      }
      meshes[obj_name] = me;
    }
  }

  uint32_t add_xform(const Xf& x) { rrt_xform a; xf_to_abi(x, &a); s.xforms.push_back(a); return (uint32_t)s.xforms.size() - 1; }

  // make_sphere renderprocess.rs:1097-1107 + Sphere::new sphere.rs:29-48
  uint32_t make_sphere(const Json& sc) {
    Xf to_world = make_to_world(sc);
    double radius = read_f64(sc, "radius", 1.0);
    double z_min = read_f64(sc, "z_min", -radius), z_max = read_f64(sc, "z_max", radius), phi_max = read_f64(sc, "phi_max", 360.0);
    auto clamp = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    rrt_sphere sp{};
    sp.xform = (int32_t)add_xform(to_world);
    sp.radius = radius; sp.z_min = z_min; sp.z_max = z_max;
    sp.theta_min = std::acos(clamp(std::fmin(z_min, z_max) / radius, -1.0, 1.0));
    sp.theta_max = std::acos(clamp(std::fmax(z_min, z_max) / radius, -1.0, 1.0));
    sp.phi_max = clamp(phi_max, 0.0, 360.0) * (M_PI / 180.0);  // f64::to_radians
    s.spheres.push_back(sp);
    return (uint32_t)s.spheres.size() - 1;
  }

  void no_media(const Json& cfg, const std::string& what) {
    const Json* mi = cfg.get("medium_interface");
    if (mi && (mi->get("inside") || mi->get("outside"))) throw Unsupported(what + ": participating media are out of scope (SURVEY §2 row 36)");
  }
  uint32_t material_index(const std::string& name, const std::string& who) {
    const MatEntry& me = materials[name];
    if (me.index < 0) throw Unsupported(who + " uses material '" + name + "' of type " + me.unsupported_type + " (outside the hot-path scope, SURVEY §2 row 22)");
    return (uint32_t)me.index;
  }

  // make_aggregate renderprocess.rs:1178-1304
  void make_aggregate(const Json& cfg) {
    const Json* ac = cfg.get("Aggregate");
    if (!ac) throw Panic("renderprocess.rs:1181 No Aggregate Config Defined");
    uint32_t max_prims = (uint32_t)read_i64(*ac, "max_prims_in_node", 4);
    const Json* pv = ac->get("primitives");
    if (pv && pv->kind == Json::Arr) {
      for (auto& pc : pv->arr) {
        std::string ptype = read_string(pc, "primitive_type", "");
        if (ptype == "sphere") {
          std::string mat = read_string(pc, "material_name", "DefaultMaterialName");
          uint32_t sphere = make_sphere(pc);
          if (!materials.count(mat)) continue;  // silently skipped, :1192
          no_media(pc, "sphere primitive");
          uint32_t mi = material_index(mat, "sphere primitive");
          const Json* inst = pc.get("instances");
          if (inst && inst->kind == Json::Arr) {
            for (auto& ic : inst->arr) {
              rrt_prim p{}; p.type = RRT_PRIM_SPHERE; p.shape = sphere; p.instance = (int32_t)add_xform(make_to_world(ic)); p.material = mi;
              s.prims.push_back(p);
            }
          } else {
            rrt_prim p{}; p.type = RRT_PRIM_SPHERE; p.shape = sphere; p.instance = -1; p.material = mi;
            s.prims.push_back(p);
          }
        } else if (ptype == "triangle") {
          std::string obj = read_string(pc, "obj_name", "DefaultObjName"), mat = read_string(pc, "material_name", "DefaultMaterialName");
          if (!meshes.count(obj) || !materials.count(mat)) { warn("Error creating triangle instances " + obj + ":" + mat); continue; }
          no_media(pc, "triangle primitive");
          uint32_t mi = material_index(mat, "triangle primitive '" + obj + "'");
          const MeshEntry& me = meshes[obj];
          const Json* inst = pc.get("instances");
          if (inst && inst->kind == Json::Arr) {
            for (auto& ic : inst->arr) {
              int32_t x = (int32_t)add_xform(make_to_world(ic));
              for (uint32_t t = 0; t < me.n_tris; t++) {
                rrt_prim p{}; p.type = RRT_PRIM_TRIANGLE; p.shape = me.first_tri + t; p.instance = x; p.material = mi;
                s.prims.push_back(p);
              }
            }
          } else {
            for (uint32_t t = 0; t < me.n_tris; t++) {
              rrt_prim p{}; p.type = RRT_PRIM_TRIANGLE; p.shape = me.first_tri + t; p.instance = -1; p.material = mi;
              s.prims.push_back(p);
            }
          }
        } else {
          warn("Unsupported primitive_type! " + ptype);
        }
      }
    } else {
      warn("Found 0 primitives! You're rendering nothing!");
    }
    build_bvh(s, max_prims, flags);
  }

  // make_light renderprocess.rs:968-1053
  bool make_light(const Json& lc, rrt_light* out, bool infinite_list) {
    if (lc.kind != Json::Obj) throw Panic("renderprocess.rs:974 assert!(light_config.is_object())");
    const Json* lt = lc.get("light_type");
    if (!lt || lt->kind != Json::Str) throw Panic("renderprocess.rs:1052 Failed to parse light");
    no_media(lc, "light");
    rrt_light l{};
    if (lt->str == "point") {
      l.type = RRT_LIGHT_POINT;
      Rgb i = make_spectrum(lc, "spectrum", 1.0);
      for (int k = 0; k < 3; k++) l.spectrum[k] = i.c[k];
      l.n_samples = 1;
      // PointLight::new(light_to_world, mi, Point3f::default(), i): p_light is the origin (Q17)
    } else if (lt->str == "diffuse") {
      l.type = RRT_LIGHT_DIFFUSE;
      Rgb e = make_spectrum(lc, "spectrum", 1.0);
      for (int k = 0; k < 3; k++) l.spectrum[k] = e.c[k];
      l.n_samples = (int32_t)read_i64(lc, "n_samples", 1);
      const Json* sc = lc.get("light_shape");
      if (!sc) throw Panic("renderprocess.rs:1013 Shape Required for a DiffuseLight!");
      const Json* st = sc->get("shape_type");
      if (st && st->kind == Json::Str && st->str == "sphere") {
        l.shape_type = RRT_PRIM_SPHERE;
        l.shape = make_sphere(*sc);
        const rrt_sphere& sp = s.spheres[l.shape];
        l.area = sp.phi_max * sp.radius * (sp.z_max - sp.z_min);  // Sphere::area sphere.rs:261-263
      } else if (st && st->kind == Json::Str && st->str == "triangle") {
        std::string obj = read_string(*sc, "obj_name", "");
        if (!meshes.count(obj)) throw Panic("renderprocess.rs:1087 triangle_mesh[\"" + obj + "\"] missing");
        uint64_t tri_num = (uint64_t)read_i64(*sc, "tri_num", 0);
        const MeshEntry& me = meshes[obj];
        if (tri_num >= me.n_tris) throw Panic("renderprocess.rs:1090 mesh[tri_num] out of bounds");
        l.shape_type = RRT_PRIM_TRIANGLE;
        l.shape = me.first_tri + (uint32_t)tri_num;
        const rrt_tri& t = s.tris[l.shape];
        auto P = [&](uint32_t i) { return V3{s.positions[3 * i], s.positions[3 * i + 1], s.positions[3 * i + 2]}; };
        V3 p0 = P(t.v[0]), p1 = P(t.v[1]), p2 = P(t.v[2]);
        l.area = 0.5 * length(cross(p1 - p0, p2 - p0));  // Triangle::area triangle.rs:420-425
      } else {
        throw Panic("renderprocess.rs:1095 Failed to parse a Shape");
      }
    } else if (lt->str == "distant") {
      if (infinite_list) { warn("infinite_lights entry of type distant contributes Le = 0; ignored"); return false; }
      // DistantLight::new lights/distant.rs:23-43 (renderprocess.rs:1018-1031)
      l.type = RRT_LIGHT_DISTANT;
      l.n_samples = 1;
      Rgb li = make_spectrum(lc, "l", 1.0), sc = make_spectrum(lc, "scale", 1.0);
      for (int k = 0; k < 3; k++) l.spectrum[k] = li.c[k] * sc.c[k];
      V3 from = fetch_v3(lc, "from", {0, 0, 0}), to = fetch_v3(lc, "to", {0, 0, 1});
      V3 w = normalize(xf_vector(make_to_world(lc), from - to));
      l.w_light[0] = w.x; l.w_light[1] = w.y; l.w_light[2] = w.z;
      // Bounds3f::bounding_sphere(aggregate.world_bound()) geometry.rs:1656-1668
      const double* wb = s.desc.world_bound;
      V3 c{(wb[0] + wb[3]) / 2.0, (wb[1] + wb[4]) / 2.0, (wb[2] + wb[5]) / 2.0};
      bool inside = c.x >= wb[0] && c.x <= wb[3] && c.y >= wb[1] && c.y <= wb[4] && c.z >= wb[2] && c.z <= wb[5];
      l.world_radius = inside ? length(V3{wb[3], wb[4], wb[5]} - c) : 0.0;
    } else if (lt->str == "infinite") {
      throw Unsupported("light_type infinite needs MIPMap/Distribution2D (out of scope, SURVEY §2 row 24)");
    } else {
      throw Panic("renderprocess.rs:1047 Failed to parse light \"" + lt->str + "\"");
    }
    *out = l;
    return true;
  }

  void make_all_lights(const Json& cfg) {  // :921-966
    const Json* ls = cfg.get("lights");
    size_t n = 0;
    if (ls && ls->kind == Json::Arr)
      for (auto& lc : ls->arr) { rrt_light l; if (make_light(lc, &l, false)) { s.lights.push_back(l); n++; } }
    const Json* il = cfg.get("infinite_lights");
    if (il && il->kind == Json::Arr)
      for (auto& lc : il->arr) {
        // Path adds light.le(ray) for these on a miss (path.rs:84-86); point/diffuse/distant le() is 0
        rrt_light l;
        if (make_light(lc, &l, true)) { warn("infinite_lights entry with Le = 0 ignored"); n++; }
      }
    if (n == 0) warn("No lights found!");
  }

  // make_film renderprocess.rs:1327-1366 + Film::new film.rs:141-186
  void make_film(const Json& fc) {
    rrt_film& f = s.desc.film;
    f.xres = (int32_t)read_i64(fc, "xres", 1280);
    f.yres = (int32_t)read_i64(fc, "yres", 720);
    f.scale = read_f64(fc, "scale", 1.0);
    double diagonal = read_f64(fc, "diagonal", 35.0);
    f.max_sample_luminance = read_f64(fc, "max_sample_luminance", INFINITY);
    const Json* flc = fc.get("Filter");
    if (!flc || fc.kind != Json::Obj) throw Panic("renderprocess.rs:1365 Failed to create Film (filter_config not found)");
    std::string ft = read_string(*flc, "filter_type", "BoxFilter");
    auto fetch_v2 = [&](double dx, double dy, double out[2]) {  // fetch_vector2f :234-240
      const Json* v = flc->get("radius");
      std::vector<double> a;
      if (v && read_num_array(*v, 2, &a)) { out[0] = a[0]; out[1] = a[1]; } else { out[0] = dx; out[1] = dy; }
    };
    f.filter_alpha = 0;
    if (ft == "TriangleFilter") { f.filter_type = RRT_FILTER_TRIANGLE; fetch_v2(2.0, 2.0, f.filter_radius); }
    else if (ft == "GaussianFilter") { f.filter_type = RRT_FILTER_GAUSSIAN; fetch_v2(2.0, 2.0, f.filter_radius); f.filter_alpha = read_f64(*flc, "alpha", 2.0); }
    else { f.filter_type = RRT_FILTER_BOX; fetch_v2(0.5, 0.5, f.filter_radius); }
    // cropped_pixel_bounds with crop window (0,0)-(1,1), film.rs:151-160
    f.crop[0] = (int32_t)std::ceil((double)f.xres * 0.0); f.crop[1] = (int32_t)std::ceil((double)f.yres * 0.0);
    f.crop[2] = (int32_t)std::ceil((double)f.xres * 1.0); f.crop[3] = (int32_t)std::ceil((double)f.yres * 1.0);
    // filter table, film.rs:163-173 (Q4: p.x assigned twice, p.y stays 0)
    double ex = std::exp(-f.filter_alpha * f.filter_radius[0] * f.filter_radius[0]);
    double ey = std::exp(-f.filter_alpha * f.filter_radius[1] * f.filter_radius[1]);
    int off = 0;
    for (int y = 0; y < 16; y++)
      for (int x = 0; x < 16; x++) {
        double px = ((double)x + 0.5) * f.filter_radius[0] / 16.0;
        px = ((double)y + 0.5) * f.filter_radius[1] / 16.0;
        double py = 0.0, v;
        if (f.filter_type == RRT_FILTER_BOX) v = 1.0;
        else if (f.filter_type == RRT_FILTER_TRIANGLE) v = std::fmax(0.0, f.filter_radius[0] - std::fabs(px)) * std::fmax(0.0, f.filter_radius[1] - std::fabs(py));
        else v = std::fmax(0.0, std::exp(-f.filter_alpha * px * px) - ex) * std::fmax(0.0, std::exp(-f.filter_alpha * py * py) - ey);
        f.filter_table[off++] = v;
      }
    f.diagonal = diagonal * 0.001;
    // get_sample_bounds film.rs:188-199
    f.sample_bounds[0] = (int32_t)std::floor((double)f.crop[0] + 0.5 - f.filter_radius[0]);
    f.sample_bounds[1] = (int32_t)std::floor((double)f.crop[1] + 0.5 - f.filter_radius[1]);
    f.sample_bounds[2] = (int32_t)std::ceil((double)f.crop[2] - 0.5 + f.filter_radius[0]);
    f.sample_bounds[3] = (int32_t)std::ceil((double)f.crop[3] - 0.5 + f.filter_radius[1]);
    // get_physical_extent film.rs:200-208
    double aspect = (double)f.yres / (double)f.xres;
    double x = std::sqrt(f.diagonal * f.diagonal / (1.0 + aspect * aspect));
    double y = aspect * x;
    f.physical_extent[0] = -x / 2.0; f.physical_extent[1] = -y / 2.0; f.physical_extent[2] = x / 2.0; f.physical_extent[3] = y / 2.0;
  }

  // make_camera renderprocess.rs:1368-1397
  void make_camera(const Json& cc) {
    V3 world_pos = fetch_v3(cc, "world_pos", {0, 0, 0}), look = fetch_v3(cc, "look", {1, 1, 1}), up = fetch_v3(cc, "up", {0, 0, 1});
    Xf to_camera = xf_look_at(world_pos, look, up);
    double so = read_f64(cc, "shutter_open", 0.0), sc = read_f64(cc, "shutter_close", 1.0);
    double ap = read_f64(cc, "aperture_diameter", 1.0), fd = read_f64(cc, "focus_distance", 10.0);
    bool sw = read_bool(cc, "simple_weighting", true);
    const Json* ld = cc.get("lens_data");
    std::vector<double> lens;
    if (!ld || !read_num_array(*ld, 0, &lens)) throw Panic("renderprocess.rs:1379 lens_data missing or not a number array");
    if (cc.get("medium")) throw Unsupported("Camera.medium: participating media are out of scope (SURVEY §2 row 36)");
    init_camera(s, xf_inverse(to_camera), so, sc, ap, fd, lens, sw);
  }

  // make_sampler renderprocess.rs:1306-1325 (sample_bounds = film.cropped_pixel_bounds, :1410)
  void make_sampler(const Json& sc) {
    std::string t = read_string(sc, "sampler_type", "");
    rrt_sampler& sp = s.desc.sampler;
    if (t == "StratifiedSampler") {
      sp.type = RRT_SAMPLER_STRATIFIED;
      sp.jitter = read_bool(sc, "jitter", true);
      sp.xsamp = (int32_t)read_i64(sc, "xsamp", 4); sp.ysamp = (int32_t)read_i64(sc, "ysamp", 4);
      sp.dimension = (int32_t)read_i64(sc, "dimension", 4);
      sp.samples_per_pixel = (uint64_t)sp.xsamp * (uint64_t)sp.ysamp;
      sp.perm_seed = seed;
    } else if (t == "HaltonSampler") {
      uint64_t nsamp = (uint64_t)read_i64(sc, "nsamp", 16);
      bool center = read_bool(sc, "sample_at_center", false);
      init_halton(s, nsamp, center, s.desc.film.crop, seed);
    } else {
      throw Panic("renderprocess.rs:1322 Unsupported Sampler type");
    }
  }

  void make_integrator(const Json& cfg) {  // :1399-1499
    const Json *ic = cfg.get("Integrator"), *sc = cfg.get("Sampler"), *fc = cfg.get("Film"), *cc = cfg.get("Camera");
    if (!ic || !sc || !fc || !cc) throw Panic("renderprocess.rs:1498 Failed to create Integrator");
    make_film(*fc);
    make_camera(*cc);
    make_sampler(*sc);
    rrt_integrator& in = s.desc.integrator;
    std::string t = read_string(*ic, "integrator_type", "AO");
    in.max_depth = 5; in.rr_threshold = 1.0; in.light_strategy = RRT_STRATEGY_ONE; in.cos_sample = 1; in.n_samples = 64;
    if (t == "DirectLighting") {
      in.type = RRT_INT_DIRECT;
      in.light_strategy = read_string(*ic, "light_strategy", "one") == "all" ? RRT_STRATEGY_ALL : RRT_STRATEGY_ONE;
      in.max_depth = (int32_t)read_i64(*ic, "max_depth", 5);
    } else if (t == "Path") {
      in.type = RRT_INT_PATH;
      in.max_depth = (int32_t)read_i64(*ic, "max_depth", 5);
      in.rr_threshold = read_f64(*ic, "rr_threshold", 1.0);
    } else if (t == "Volpath" || t == "SPPM") {
      throw Unsupported("integrator_type " + t + " is out of scope (SURVEY §2 rows 9-10)");
    } else if (t == "Debug") {
      in.type = RRT_INT_DEBUG;
      in.max_depth = (int32_t)read_i64(*ic, "max_depth", 5);
    } else {
      in.type = RRT_INT_AO;
      in.cos_sample = read_bool(*ic, "cos_sample", true);
      in.n_samples = (int32_t)read_i64(*ic, "n_samples", 64);
    }
  }

  void load(const Json& cfg) {
    s.desc.flags = flags;
    make_textures(cfg);
    make_materials(cfg);
    make_meshes(cfg);
    make_aggregate(cfg);
    make_all_lights(cfg);
    make_integrator(cfg);
    s.finalize();
  }
};

int guarded(const std::function<void()>& fn) {
  try {
    fn();
    return RRT_OK;
  } catch (const Panic& e) { set_last_error(std::string("panic: ") + e.what()); return RRT_EPANIC; }
  catch (const Unsupported& e) { set_last_error(std::string("unsupported: ") + e.what()); return RRT_EUNSUP; }
  catch (const ParseError& e) { set_last_error(e.what()); return RRT_EPARSE; }
  catch (const IoError& e) { set_last_error(e.what()); return RRT_EIO; }
  catch (const std::bad_alloc&) { set_last_error("out of memory"); return RRT_ENOMEM; }
  catch (const std::exception& e) { set_last_error(e.what()); return RRT_EINVAL; }
}

}  // namespace
}  // namespace rrt

using namespace rrt;

extern "C" {

int rrt_scene_load_str(const char* json_text, const char* root_dir, uint32_t flags, uint64_t perm_seed, rrt_scene** out) {
  if (!json_text || !out) { set_last_error("rrt_scene_load_str: null argument"); return RRT_EINVAL; }
  *out = nullptr;
  std::unique_ptr<rrt_scene> sc(new rrt_scene());
  std::string text(json_text), root(root_dir ? root_dir : ".");
  int rc = guarded([&]() {
    Json cfg = JsonParser(text).parse();
    Loader L{sc->data, root, flags, perm_seed};
    L.load(cfg);
  });
  if (rc == RRT_OK) *out = sc.release();
  return rc;
}

int rrt_scene_load(const char* path, uint32_t flags, uint64_t perm_seed, rrt_scene** out) {
  if (!path || !out) { set_last_error("rrt_scene_load: null argument"); return RRT_EINVAL; }
  *out = nullptr;
  std::ifstream f(path);
  if (!f) { set_last_error(std::string("cannot open ") + path); return RRT_EIO; }
  std::stringstream ss;
  ss << f.rdbuf();
  // scene_config_root = parent of the canonicalised path, renderprocess.rs:94-99
  std::string p(path), root = ".";
  char* real = realpath(path, nullptr);
  if (real) { p = real; free(real); }
  size_t slash = p.find_last_of('/');
  if (slash != std::string::npos) root = p.substr(0, slash);
  return rrt_scene_load_str(ss.str().c_str(), root.c_str(), flags, perm_seed, out);
}

const rrt_scene_desc* rrt_scene_desc_of(const rrt_scene* s) { return s ? &s->data.desc : nullptr; }
void rrt_scene_free(rrt_scene* s) { delete s; }

// diagnostics the reference prints with eprintln! while loading (non-fatal)
size_t rrt_scene_warning_count(const rrt_scene* s) { return s ? s->data.warnings.size() : 0; }
const char* rrt_scene_warning(const rrt_scene* s, size_t i) { return (s && i < s->data.warnings.size()) ? s->data.warnings[i].c_str() : nullptr; }

const char* rrt_last_error(void) { return last_error_cstr(); }
int rrt_scene_film(const rrt_scene* s, int32_t* xres, int32_t* yres, double* scale) {
  if (!s) { set_last_error("rrt_scene_film: null scene"); return RRT_EINVAL; }
  if (xres) *xres = s->data.desc.film.xres;
  if (yres) *yres = s->data.desc.film.yres;
  if (scale) *scale = s->data.desc.film.scale;
  return RRT_OK;
}

#define RRT_STR2(x) #x
#define RRT_STR(x) RRT_STR2(x)
const char* rrt_version(void) { return "rs_ray_toy_amd 0.3 (abi " RRT_STR(RRT_ABI_VERSION) ", gfx950)"; }

}  // extern "C"
