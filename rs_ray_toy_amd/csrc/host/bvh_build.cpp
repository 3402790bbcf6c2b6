// Host BVH construction: HLBVH exactly as /root/reference/src/bvh.rs:307-751 builds it, because the
// reference's hit selection is traversal-order dependent and the linearised tree is an
// input of the traversal kernels. Quirks Q26 (emit_lbvh slice) and Q27 (degenerate SAH) are reproduced
// unless RRT_FIX_BVH_* flags are set. f64 throughout (build with -ffp-contract=off).
#include <algorithm>
#include <functional>

#include "scene.hpp"

namespace rrt {

static V3 pos_of(const SceneData& s, uint32_t i) { return {s.positions[3 * i], s.positions[3 * i + 1], s.positions[3 * i + 2]}; }

static Xf xf_from_abi(const rrt_xform& x) {
  Xf t;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { t.m.m[i][j] = x.m[i * 4 + j]; t.minv.m[i][j] = x.m_inv[i * 4 + j]; }
  return t;
}

B3 prim_world_bound(const SceneData& s, size_t pi) {
  const rrt_prim& p = s.prims[pi];
  B3 b;
  if (p.type == RRT_PRIM_TRIANGLE) {
    // Triangle::world_bound triangle.rs:220-225 — raw mesh.p, obj_to_world is never applied (Q13)
    const rrt_tri& t = s.tris[p.shape];
    b = bunion(bnew(pos_of(s, t.v[0]), pos_of(s, t.v[1])), pos_of(s, t.v[2]));
  } else {
    // Shape::world_bound shape/mod.rs:13-15 over Sphere::object_bound sphere.rs:117-122
    const rrt_sphere& sp = s.spheres[p.shape];
    B3 ob = bnew({-sp.radius, -sp.radius, sp.z_min}, {sp.radius, sp.radius, sp.z_max});
    b = xf_bounds(xf_from_abi(s.xforms[sp.xform]), ob);
  }
  // TransformedPrimitive::world_bound primitives.rs:112-114
  if (p.instance >= 0) b = xf_bounds(xf_from_abi(s.xforms[p.instance]), b);
  return b;
}

namespace {

struct BuildNode {  // BVHBuildNode bvh.rs:47-54
  B3 bounds;
  int child[2] = {-1, -1};
  uint32_t split_axis = 0, first_prim_offset = 0, n_primitives = 0;
};
struct MortonPrim { uint32_t primitive_index = 0, morton_code = 0; };

// Rust `f64 as u32`: truncating, saturating, NaN -> 0
uint32_t f64_as_u32(double v) {
  if (!(v == v)) return 0;
  if (v <= 0.0) return 0;
  if (v >= 4294967295.0) return 4294967295u;
  return (uint32_t)v;
}
size_t f64_as_usize(double v) {
  if (!(v == v)) return 0;
  if (v <= 0.0) return 0;
  if (v >= 18446744073709551615.0) return ~(size_t)0;
  return (size_t)v;
}

uint32_t left_shift3(uint32_t x) {  // bvh.rs:17-32
  if (x > (1u << 10)) throw Panic("bvh.rs:19 assert!(x <= (1 << 10))");
  if (x == (1u << 10)) x -= 1;
  x = (x | (x << 16)) & 0b00000011000000000000000011111111u;
  x = (x | (x << 8)) & 0b00000011000000001111000000001111u;
  x = (x | (x << 4)) & 0b00000011000011000011000011000011u;
  x = (x | (x << 2)) & 0b00001001001001001001001001001001u;
  return x;
}
uint32_t encode_morton3(V3 v) {  // bvh.rs:34-39
  if (!(v.x >= 0.0) || !(v.y >= 0.0) || !(v.z >= 0.0)) throw Panic("bvh.rs:35-37 assert!(v >= 0.0)");
  return (left_shift3(f64_as_u32(v.z)) << 2) | (left_shift3(f64_as_u32(v.y)) << 1) | left_shift3(f64_as_u32(v.x));
}

void radix_sort(std::vector<MortonPrim>& v) {  // bvh.rs:247-304
  std::vector<MortonPrim> tmp(v.size());
  const int bits_per_pass = 6, n_bits = 30, n_passes = n_bits / bits_per_pass;
  for (int pass = 0; pass < n_passes; pass++) {
    int low_bit = pass * bits_per_pass;
    std::vector<MortonPrim>& in = (pass & 1) ? tmp : v;
    std::vector<MortonPrim>& out = (pass & 1) ? v : tmp;
    const int n_buckets = 1 << bits_per_pass;
    const uint32_t mask = (1u << bits_per_pass) - 1;
    size_t count[64] = {0}, out_index[64];
    for (auto& mp : in) count[(mp.morton_code >> low_bit) & mask]++;
    out_index[0] = 0;
    for (int i = 1; i < n_buckets; i++) out_index[i] = out_index[i - 1] + count[i - 1];
    for (auto& mp : in) out[out_index[(mp.morton_code >> low_bit) & mask]++] = mp;
  }
  if (n_passes & 1) std::swap(v, tmp);
}

struct Builder {
  SceneData& s;
  uint32_t max_prims_in_node, flags;
  std::vector<B3> prim_bounds;
  std::vector<BuildNode> nodes;  // arena
  uint32_t total_nodes = 0, ordered_offset = 0;
  std::vector<uint32_t> ordered;

  int new_node(const BuildNode& n) { nodes.push_back(n); return (int)nodes.size() - 1; }

  // emit_lbvh bvh.rs:516-612
  int emit_lbvh(const MortonPrim* mp, uint32_t n, int bit_index) {
    if (n == 0) throw Panic("bvh.rs:527 assert!(n_primitives > 0)");
    if (bit_index == -1 || n < max_prims_in_node) {
      total_nodes++;
      BuildNode node;
      B3 b;
      uint32_t first = ordered_offset;
      ordered_offset += n;
      for (uint32_t i = 0; i < n; i++) {
        uint32_t pi = mp[i].primitive_index;
        ordered[first + i] = pi;
        b = bunion(b, prim_bounds[pi]);
      }
      node.first_prim_offset = first;
      node.n_primitives = n;
      node.bounds = b;
      return new_node(node);
    }
    uint32_t mask = 1u << bit_index;
    if ((mp[0].morton_code & mask) == (mp[n - 1].morton_code & mask)) return emit_lbvh(mp, n, bit_index - 1);
    uint32_t ss = 0, se = n - 1;
    while (ss + 1 != se) {
      uint32_t mid = (ss + se) / 2;
      if ((mp[ss].morton_code & mask) == (mp[mid].morton_code & mask)) ss = mid; else se = mid;
    }
    uint32_t split = se;
    total_nodes++;
    int c0 = emit_lbvh(mp, split, bit_index - 1);
    // Q26: the reference passes the *same* slice start for the second child (bvh.rs:598-607)
    const MortonPrim* mp1 = (flags & RRT_FIX_BVH_LBVH_SLICE) ? mp + split : mp;
    int c1 = emit_lbvh(mp1, n - split, bit_index - 1);
    BuildNode node;
    node.bounds = bunion(nodes[c0].bounds, nodes[c1].bounds);
    node.child[0] = c0; node.child[1] = c1;
    node.split_axis = (uint32_t)(bit_index % 3);
    node.n_primitives = 0;
    return new_node(node);
  }

  // build_upper_sah bvh.rs:614-726; roots[] holds arena indices and is permuted in place
  int build_upper_sah(std::vector<int>& roots, uint32_t start, uint32_t end) {
    if (!(start < end)) throw Panic("bvh.rs:622 assert!(start < end)");
    uint32_t n_nodes = end - start;
    if (n_nodes == 1) return roots[start];
    total_nodes++;
    B3 bounds;
    for (uint32_t i = start; i < end; i++) bounds = bunion(bounds, nodes[roots[i]].bounds);
    B3 cb;
    for (uint32_t i = start; i < end; i++) {
      const B3& b = nodes[roots[i]].bounds;
      cb = bunion(cb, (b.pmin + b.pmax) * 0.5);
    }
    int dim = maximum_extent(cb);
    if (!(cb.pmax[dim] != cb.pmin[dim])) throw Panic("bvh.rs:647 assert!(centroid_bounds.p_max[dim] != centroid_bounds.p_min[dim])");
    const size_t n_buckets = 12;
    struct Bucket { uint32_t count = 0; B3 bounds; } buckets[12];
    auto bucket_of = [&](const B3& b) {
      double centroid = (b.pmin[dim] + b.pmax[dim]) * 0.5;
      size_t bi = f64_as_usize((double)n_buckets * ((centroid - cb.pmin[dim]) / (cb.pmax[dim] - cb.pmin[dim])));
      if (bi == n_buckets) bi = n_buckets - 1;
      if (!(bi < n_buckets)) throw Panic("bvh.rs:664 assert!(b < n_buckets)");
      return bi;
    };
    for (uint32_t i = start; i < end; i++) {
      size_t b = bucket_of(nodes[roots[i]].bounds);
      buckets[b].count++;
      buckets[b].bounds = bunion(buckets[b].bounds, nodes[roots[i]].bounds);
    }
    double costs[11];
    for (size_t i = 0; i < n_buckets - 1; i++) {
      B3 b0, b1;
      uint32_t c0 = 0, c1 = 0;
      // Q27: reference loops are 0..i and i+1..n (bucket i in neither side) -> costs[0] is NaN
      size_t left_end = (flags & RRT_FIX_BVH_SAH) ? i + 1 : i;
      for (size_t j = 0; j < left_end; j++) { b0 = bunion(b0, buckets[j].bounds); c0 += buckets[j].count; }
      for (size_t j = i + 1; j < n_buckets; j++) { b1 = bunion(b1, buckets[j].bounds); c1 += buckets[j].count; }
      costs[i] = 0.125 + ((double)c0 * surface_area(b0) + (double)c1 * surface_area(b1)) / surface_area(bounds);
    }
    double min_cost = costs[0];
    size_t min_bucket = 0;
    for (size_t i = 1; i < n_buckets - 1; i++)
      if (costs[i] < min_cost) { min_cost = costs[i]; min_bucket = i; }
    // Iterator::partition_in_place (Rust nightly std) == bidirectional std::partition swap sequence
    auto pred = [&](int r) { return bucket_of(nodes[r].bounds) <= min_bucket; };
    uint32_t first = start, last = end;
    while (true) {
      while (first != last && pred(roots[first])) first++;
      if (first == last) break;
      last--;
      while (first != last && !pred(roots[last])) last--;
      if (first == last) break;
      std::swap(roots[first], roots[last]);
      first++;
    }
    uint32_t mid = first;
    if (!(mid > start)) throw Panic("bvh.rs:716 assert!(mid > start)");
    if (!(mid < end)) throw Panic("bvh.rs:717 assert!(mid < end)");
    int c0 = build_upper_sah(roots, start, mid);
    int c1 = build_upper_sah(roots, mid, end);
    BuildNode node;
    node.bounds = bunion(nodes[c0].bounds, nodes[c1].bounds);
    node.child[0] = c0; node.child[1] = c1;
    node.split_axis = (uint32_t)dim;
    node.n_primitives = 0;
    return new_node(node);
  }

  // flattern_bvh bvh.rs:728-751 (pre-order; child 0 adjacent)
  uint32_t flatten(int ni, uint32_t& offset, uint32_t depth, uint32_t& max_depth) {
    const BuildNode bn = nodes[ni];
    uint32_t my = offset++;
    if (depth > max_depth) max_depth = depth;
    rrt_bvh_node& ln = s.bvh_nodes[my];
    ln.bounds[0] = bn.bounds.pmin.x; ln.bounds[1] = bn.bounds.pmin.y; ln.bounds[2] = bn.bounds.pmin.z;
    ln.bounds[3] = bn.bounds.pmax.x; ln.bounds[4] = bn.bounds.pmax.y; ln.bounds[5] = bn.bounds.pmax.z;
    if (bn.n_primitives > 0) {
      if (!(bn.n_primitives < (2u << 15))) throw Panic("bvh.rs:735 assert!(node.n_primitives < (2 << 15))");
      ln.offset = bn.first_prim_offset;
      ln.n_primitives = bn.n_primitives;
      ln.axis = 0;
    } else {
      ln.axis = bn.split_axis;
      ln.n_primitives = 0;
      flatten(bn.child[0], offset, depth + 1, max_depth);
      uint32_t second = flatten(bn.child[1], offset, depth + 1, max_depth);
      s.bvh_nodes[my].offset = second;
    }
    return my;
  }
};

}  // namespace

void build_bvh(SceneData& s, uint32_t max_prims_in_node, uint32_t flags) {
  size_t n = s.prims.size();
  if (n == 0) throw Panic("bvh.rs:319 assert!(bvhaccel.primitives.len() > 0)");
  Builder b{s, max_prims_in_node, flags};
  b.prim_bounds.resize(n);
  std::vector<V3> centroid(n);
  for (size_t i = 0; i < n; i++) {
    b.prim_bounds[i] = prim_world_bound(s, i);
    centroid[i] = (b.prim_bounds[i].pmin + b.prim_bounds[i].pmax) * 0.5;  // bvh.rs:329
  }
  // hlbvh_build bvh.rs:365-514
  B3 bounds;
  for (size_t i = 0; i < n; i++) bounds = bunion(bounds, centroid[i]);
  std::vector<MortonPrim> mps(n);
  const double morton_scale = (double)(1 << 10);
  for (size_t i = 0; i < n; i++) {
    mps[i].primitive_index = (uint32_t)i;
    mps[i].morton_code = encode_morton3(boffset(bounds, centroid[i]) * morton_scale);
  }
  radix_sort(mps);
  struct Treelet { size_t start, n; int root; };
  std::vector<Treelet> treelets;
  for (size_t start = 0, end = 1; end <= n; end++) {
    const uint32_t mask = 0b00111111111111000000000000000000u;
    if (end == n || ((mps[start].morton_code & mask) != (mps[end].morton_code & mask))) {
      treelets.push_back({start, end - start, -1});
      start = end;
    }
  }
  b.ordered.assign(n, 0);
  b.nodes.reserve(2 * n + 16);
  const int first_bit_index = 29 - 12;
  for (auto& tr : treelets) tr.root = b.emit_lbvh(&mps[tr.start], (uint32_t)tr.n, first_bit_index);
  std::vector<int> roots;
  roots.reserve(treelets.size());
  for (auto& tr : treelets) roots.push_back(tr.root);
  int root = b.build_upper_sah(roots, 0, (uint32_t)roots.size());
  s.prim_order = b.ordered;
  s.bvh_nodes.assign(b.total_nodes, rrt_bvh_node{});
  uint32_t offset = 0, max_depth = 0;
  b.flatten(root, offset, 1, max_depth);
  if (offset != b.total_nodes) throw Panic("bvh.rs:361 assert_eq!(total_nodes, offset)");
  s.desc.max_prims_in_node = max_prims_in_node;
  s.desc.bvh_depth = max_depth;
  const rrt_bvh_node& r = s.bvh_nodes[0];
  for (int k = 0; k < 6; k++) s.desc.world_bound[k] = r.bounds[k];
}

}  // namespace rrt
