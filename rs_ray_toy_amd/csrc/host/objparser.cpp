// Wavefront OBJ subset reader with the reference's exact acceptance rules (/root/reference/src/objparser.rs:83-247):
// `v x y z`, `vt u [v]`, `vn x y z` (normalised on read), `f a[/b[/c]] x3` (first three vertices only,
// 1-based, no negative indices); a line whose first token is exactly "#" is a comment; anything else is
// reported and skipped.
#include <fstream>
#include <sstream>

#include "scene.hpp"

namespace rrt {
namespace {

bool parse_f64(const std::string& tok, double* out) {
  // Rust f64::from_str: no leading/trailing junk, accepts inf/nan/exponents
  if (tok.empty()) return false;
  char* end = nullptr;
  errno = 0;
  double v = strtod(tok.c_str(), &end);
  if (end != tok.c_str() + tok.size()) return false;
  if (tok[0] == ' ' || tok.find('x') != std::string::npos || tok.find('X') != std::string::npos) return false;
  *out = v;
  return true;
}

bool parse_usize(const std::string& tok, uint64_t* out) {
  // usize::from_str: optional '+', digits only
  if (tok.empty()) return false;
  size_t i = 0;
  if (tok[0] == '+') i = 1;
  if (i >= tok.size()) return false;
  uint64_t v = 0;
  for (; i < tok.size(); i++) {
    if (tok[i] < '0' || tok[i] > '9') return false;
    uint64_t nv = v * 10 + (uint64_t)(tok[i] - '0');
    if (nv < v) return false;
    v = nv;
  }
  *out = v;
  return true;
}

struct FaceElem { bool has[3] = {false, false, false}; uint64_t idx[3] = {0, 0, 0}; };

// parse_face_element objparser.rs:215-226
FaceElem parse_face_element(const std::string& s, const std::string& file, unsigned line) {
  FaceElem fe;
  size_t start = 0;
  int k = 0;
  while (true) {
    size_t slash = s.find('/', start);
    std::string part = s.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
    uint64_t v;
    if (k < 3 && parse_usize(part, &v)) {
      if (v == 0) throw Panic("objparser.rs:219 `i - 1` underflows for index 0 (" + file + ":" + std::to_string(line) + ")");
      fe.has[k] = true;
      fe.idx[k] = v - 1;
    }
    k++;
    if (slash == std::string::npos) break;
    start = slash + 1;
  }
  return fe;
}

}  // namespace

ObjMesh parse_obj(const std::string& path, std::vector<std::string>& warnings) {
  std::ifstream f(path);
  if (!f) throw IoError("parse_obj: cannot open " + path);
  ObjMesh m;
  std::string line;
  unsigned lineno = 0;
  size_t n_normals = 0, n_uvs = 0;
  auto err = [&](const std::string& d) {
    return ParseError("Error Parsing File: " + path + " at line " + std::to_string(lineno) + ", desc: " + d);
  };
  while (std::getline(f, line)) {
    lineno++;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ss(line);
    std::vector<std::string> tok;
    std::string t;
    while (ss >> t) tok.push_back(t);
    if (tok.empty()) { warnings.push_back("ParseObjError: unsupported Element None"); continue; }
    const std::string& k = tok[0];
    if (k == "v" || k == "vn") {
      double v[3];
      for (int i = 0; i < 3; i++) {
        if ((size_t)(1 + i) >= tok.size()) throw err("ParseObjError: Failed to get v" + std::to_string(i + 1));
        if (!parse_f64(tok[1 + i], &v[i])) throw err("invalid float literal");
      }
      if (k == "v") {
        m.p.insert(m.p.end(), v, v + 3);
      } else {
        // Normal3f::normalize, objparser.rs:132 (no zero guard: 0/0 = NaN as in the reference)
        double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        m.n.push_back(v[0] / l); m.n.push_back(v[1] / l); m.n.push_back(v[2] / l);
        n_normals++;
      }
    } else if (k == "vt") {
      double u, v = 0.0;
      if (tok.size() < 2) throw err("ParseObjError: Failed to get v1");
      if (!parse_f64(tok[1], &u)) throw err("invalid float literal");
      // make_uv objparser.rs:207-212: a missing second value parses "" -> error
      if (tok.size() < 3 || !parse_f64(tok[2], &v)) throw err("cannot parse float from empty string");
      m.uv.push_back(u); m.uv.push_back(v);
      n_uvs++;
    } else if (k == "f") {
      if (tok.size() < 4) throw err("ParseObjError: Failed to get face element");
      FaceElem fe[3];
      for (int i = 0; i < 3; i++) fe[i] = parse_face_element(tok[1 + i], path, lineno);
      if (fe[0].has[0] && fe[1].has[0] && fe[2].has[0]) {
        for (int i = 0; i < 3; i++) m.vi.push_back((uint32_t)fe[i].idx[0]);
        if (fe[0].has[1] && fe[1].has[1] && fe[2].has[1]) {
          if (n_uvs > 0 && fe[0].idx[1] < n_uvs && fe[1].idx[1] < n_uvs && fe[2].idx[1] < n_uvs)
            for (int i = 0; i < 3; i++) m.uvi.push_back((uint32_t)fe[i].idx[1]);
        }
        if (fe[0].has[2] && fe[1].has[2] && fe[2].has[2]) {
          if (n_normals > 0 && fe[0].idx[2] < n_normals && fe[1].idx[2] < n_normals && fe[2].idx[2] < n_normals)
            for (int i = 0; i < 3; i++) m.ni.push_back((uint32_t)fe[i].idx[2]);
        }
      }
    } else if (k == "#") {
      // comment
    } else {
      warnings.push_back("ParseObjError: unsupported Element Some(\"" + k + "\")");
    }
  }
  // ParseResult::new objparser.rs:62-67
  if (!m.uvi.empty() && m.vi.size() != m.uvi.size()) throw Panic("objparser.rs:63 assert!(vertex_indices.len() == uv_indices.len())");
  if (!m.ni.empty() && m.vi.size() != m.ni.size()) throw Panic("objparser.rs:66 assert!(vertex_indices.len() == normal_indices.len())");
  size_t nv = m.p.size() / 3;
  for (uint32_t i : m.vi)
    if (i >= nv) throw Panic("triangle.rs index out of bounds: vertex index " + std::to_string(i + 1) + " > " + std::to_string(nv) + " in " + path);
  return m;
}

}  // namespace rrt
