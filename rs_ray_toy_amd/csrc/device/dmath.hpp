// Device functions of the wavefront path tracer (gfx950): vector math, samplers, RealisticCamera ray
// generation, BxDFs, lights. Templated on the arithmetic type R (float = product path, double = parity
// mode whose results track the f64 oracle to rounding). Each function cites the reference lines whose
// semantics it implements (/root/reference/src/...); formulas keep the reference's operation order.
#pragma once
#include "dtypes.hpp"

namespace rrtd {

#define RRT_DEV __device__ __forceinline__

template <typename R> struct Const;
template <> struct Const<float> {
  static constexpr float one_minus_eps = 0.99999994f;            // 1 - 2^-24
  static constexpr float machine_eps = 5.9604645e-8f;            // 2^-24
  static constexpr float inf = __builtin_huge_valf();
};
template <> struct Const<double> {
  static constexpr double one_minus_eps = 0.99999999999999989;   // misc.rs:19
  static constexpr double machine_eps = 1.1102230246251565e-16;  // main.rs:53 (f64::EPSILON * 0.5)
  static constexpr double inf = __builtin_huge_val();
};
#define RRT_PI 3.14159265358979323846
#define RRT_PI_OVER_2 1.57079632679489661923
#define RRT_PI_OVER_4 0.78539816339744830961

template <typename R> RRT_DEV R rsqrt_(R x) { return sqrt(x); }
// 1/a of the Moller-Trumbore tests: exact division in the f64 parity mode, v_rcp_f32 (1 ulp) in the fp32 product
RRT_DEV double rcp_r(double a) { return 1.0 / a; }
RRT_DEV float rcp_r(float a) { return __builtin_amdgcn_rcpf(a); }
template <typename R> RRT_DEV R rabs(R x) { return fabs(x); }
template <typename R> RRT_DEV R rmax(R a, R b) { return fmax(a, b); }   // Rust f64::max
template <typename R> RRT_DEV R rmin(R a, R b) { return fmin(a, b); }
template <typename R> RRT_DEV R clampr(R v, R lo, R hi) { return v < lo ? lo : (v > hi ? hi : v); }  // misc.rs:98

template <typename R>
struct V3 {
  R x, y, z;
  RRT_DEV V3() : x(0), y(0), z(0) {}
  RRT_DEV V3(R a, R b, R c) : x(a), y(b), z(c) {}
  RRT_DEV explicit V3(const R* p) : x(p[0]), y(p[1]), z(p[2]) {}
};
template <typename R> RRT_DEV V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> RRT_DEV V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> RRT_DEV V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R> RRT_DEV V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename R> RRT_DEV V3<R> operator/(V3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename R> RRT_DEV R dot(V3<R> a, V3<R> b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
template <typename R> RRT_DEV R absdot(V3<R> a, V3<R> b) { return rabs(dot(a, b)); }
template <typename R> RRT_DEV V3<R> cross(V3<R> a, V3<R> b) {  // geometry.rs:1099-1107
  return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}
template <typename R> RRT_DEV R len2(V3<R> a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
template <typename R> RRT_DEV R len(V3<R> a) { return sqrt(len2(a)); }
template <typename R> RRT_DEV V3<R> vnormalize(V3<R> a) { R l = len(a); return l == R(0) ? a : a / l; }  // geometry.rs:925
template <typename R> RRT_DEV V3<R> nnormalize(V3<R> a) { return a / len(a); }                            // geometry.rs:1209
template <typename R> RRT_DEV V3<R> faceforward(V3<R> n, V3<R> v) { return dot(n, v) < R(0) ? -n : n; }  // geometry.rs:1381
template <typename R> RRT_DEV void coordinate_system(V3<R> v1, V3<R>* v2, V3<R>* v3) {  // geometry.rs:1146-1161
  if (rabs(v1.x) > rabs(v1.y)) *v2 = V3<R>(-v1.z, R(0), v1.x) / R(sqrt(v1.x * v1.x + v1.z * v1.z));
  else *v2 = V3<R>(R(0), v1.z, -v1.y) / R(sqrt(v1.y * v1.y + v1.z * v1.z));
  *v3 = cross(v1, *v2);
}

template <typename R>
struct Rgb {
  R r, g, b;
  RRT_DEV Rgb() : r(0), g(0), b(0) {}
  RRT_DEV Rgb(R a, R c, R d) : r(a), g(c), b(d) {}
  RRT_DEV explicit Rgb(R v) : r(v), g(v), b(v) {}
  RRT_DEV explicit Rgb(const R* p) : r(p[0]), g(p[1]), b(p[2]) {}
  RRT_DEV bool is_black() const { return r == R(0) && g == R(0) && b == R(0); }   // spectrum.rs:2162
  RRT_DEV R y() const { return R(0.212671) * r + R(0.715160) * g + R(0.072169) * b; }  // spectrum.rs:2733
  RRT_DEV R max_component() const { return rmax(rmax(r, g), b); }
  RRT_DEV bool has_nan() const { return r != r || g != g || b != b; }
};
template <typename R> RRT_DEV Rgb<R> operator+(Rgb<R> a, Rgb<R> b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
template <typename R> RRT_DEV Rgb<R> operator-(Rgb<R> a, Rgb<R> b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
template <typename R> RRT_DEV Rgb<R> operator*(Rgb<R> a, Rgb<R> b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
template <typename R> RRT_DEV Rgb<R> operator/(Rgb<R> a, Rgb<R> b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
template <typename R> RRT_DEV Rgb<R> operator*(Rgb<R> a, R s) { return {a.r * s, a.g * s, a.b * s}; }
template <typename R> RRT_DEV Rgb<R> operator/(Rgb<R> a, R s) { return {a.r / s, a.g / s, a.b / s}; }
template <typename R> RRT_DEV Rgb<R> rgb_sqrt(Rgb<R> a) { return {R(sqrt(a.r)), R(sqrt(a.g)), R(sqrt(a.b))}; }
template <typename R> RRT_DEV Rgb<R> rgb_clamp0(Rgb<R> a) {
  return {clampr(a.r, R(0), Const<R>::inf), clampr(a.g, R(0), Const<R>::inf), clampr(a.b, R(0), Const<R>::inf)};
}

template <typename R> RRT_DEV bool quadratic(R a, R b, R c, R* t0, R* t1) {  // misc.rs:231-251
  R discrim = b * b - R(4) * a * c;
  if (discrim < R(0)) return false;
  R root = sqrt(discrim);
  R q = (b < R(0)) ? R(-0.5) * (b - root) : R(-0.5) * (b + root);
  *t0 = q / a;
  *t1 = c / q;
  if (*t0 > *t1) { R t = *t0; *t0 = *t1; *t1 = t; }
  return true;
}

// affine 3x4 helpers (rows of a 4x4 with last row 0 0 0 1)
template <typename R> RRT_DEV V3<R> aff_pt(const R* m, V3<R> p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
          m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
template <typename R> RRT_DEV V3<R> aff_vec(const R* m, V3<R> v) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
template <typename R> RRT_DEV V3<R> aff_nrm(const R* minv, V3<R> n) {  // transform.rs:506-523
  return {minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z,
          minv[2] * n.x + minv[6] * n.y + minv[10] * n.z};
}

// ---- wave64 queue push: one atomic per wave (ballot + popcount prefix) --------------------------------
// Must be reached by every lane of the wave (callers keep control flow convergent up to here).
RRT_DEV uint32_t wave_push(uint32_t* counter, bool pred) {
  const uint64_t mask = __ballot(pred);
  const uint32_t lane = __lane_id();
  const uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
  uint32_t base = 0;
  if (mask != 0ull) {
    const uint32_t leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
  }
  return base + prefix;
}

// Block-wide queue push: one atomic per BLOCK. All atomics on a queue counter serialise on one L2 line (~7-12 ns
// each on MI355X): with a per-wave atomic the first-bounce shading kernel (1.3 M waves, two queues) spent its
// whole 18 ms there. `lds` = (waves per block + 1) words. Must be reached by every thread of the block.
RRT_DEV uint32_t block_push(uint32_t* counter, bool pred, uint32_t* lds) {
  const uint64_t mask = __ballot(pred);
  const uint32_t lane = __lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
  if (lane == 0) lds[w] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < nw; k++) { const uint32_t c = lds[k]; lds[k] = tot; tot += c; }
    lds[nw] = tot ? atomicAdd(counter, tot) : 0u;
  }
  __syncthreads();
  const uint32_t base = lds[nw] + lds[w];
  __syncthreads();   // lds is reused by the next push
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// block_push() that also reports where the block's entries went: [*first, *first + *total) (`lds` = waves per block + 2 words)
RRT_DEV uint32_t block_push_range(uint32_t* counter, bool pred, uint32_t* lds, uint32_t* first, uint32_t* total) {
  const uint64_t mask = __ballot(pred);
  const uint32_t lane = __lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
  if (lane == 0) lds[w] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < nw; k++) { const uint32_t c = lds[k]; lds[k] = tot; tot += c; }
    lds[nw] = tot ? atomicAdd(counter, tot) : 0u;
    lds[nw + 1] = tot;
  }
  __syncthreads();
  *first = lds[nw]; *total = lds[nw + 1];
  const uint32_t base = lds[nw] + lds[w];
  __syncthreads();   // lds is reused by the next push
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Rank of this thread among the block's threads with `pred` (block order) and their number: block_push() without the queue.
// `lds` = (waves per block + 1) words. Must be reached by every thread of the block.
RRT_DEV uint32_t block_rank(bool pred, uint32_t* lds, uint32_t* total) {
  const uint64_t mask = __ballot(pred);
  const uint32_t lane = __lane_id(), w = threadIdx.x >> 6, nw = (blockDim.x + 63u) >> 6;
  if (lane == 0) lds[w] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < nw; k++) { const uint32_t c = lds[k]; lds[k] = tot; tot += c; }
    lds[nw] = tot;
  }
  __syncthreads();
  const uint32_t base = lds[w];
  *total = lds[nw];
  __syncthreads();   // lds is reused by the next push
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// ---- Halton (samplers/halton.rs, lowdiscrepancy.rs); values are produced in f64 in both modes ---------
// a / base, exact for every 32-bit a: division by an invariant integer with a 33-bit magic number (Granlund & Montgomery 1994, fig. 4.1):
// l = ceil(log2 base), m' = floor(2^32 (2^l - base) / base) + 1, t = mulhi(m', a), q = (t + ((a - t) >> 1)) >> (l - 1). One multiply-high
// and four cheap operations per digit; the first version multiplied by a 41-bit magic in 64 bits (five quarter-rate multiplies) and
// needed an f64 reciprocal for a >= 2^26. hd.magic = m' | (l - 1) << 32, built on the host.
RRT_DEV uint32_t div_base(uint32_t a, const HaltonDim& hd, uint32_t /*fast*/) {
  const uint32_t t = __umulhi((uint32_t)hd.magic, a);
  return (t + ((a - t) >> 1)) >> (uint32_t)(hd.magic >> 32);
}
// radical_inverse_specialized lowdiscrepancy.rs:188-202
RRT_DEV double radical_inverse_dev(uint32_t a, const HaltonDim& hd, uint32_t fast) {
  const double inv_base = hd.inv;   // = 1.0 / (double)base, the same correctly rounded quotient (an f64 division costs ~60 issue slots here)
  uint64_t reversed = 0;
  double inv_base_n = 1.0;
  while (a != 0) {
    uint32_t next = div_base(a, hd, fast);
    uint32_t digit = a - next * hd.base;
    reversed = reversed * hd.base + digit;
    inv_base_n *= inv_base;
    a = next;
  }
  return fmin((double)reversed * inv_base_n, 0.99999999999999989);
}
// scrambled_radical_inverse_specialized lowdiscrepancy.rs:204-227
RRT_DEV double scrambled_radical_inverse_dev(uint32_t a, const HaltonDim& hd, const uint16_t* perm, uint32_t fast) {
  const double inv_base = hd.inv;
  uint64_t reversed = 0;
  double inv_base_n = 1.0;
  while (a > 0) {
    uint32_t next = div_base(a, hd, fast);
    uint32_t digit = a - next * hd.base;
    reversed = reversed * hd.base + perm[digit];
    inv_base_n *= inv_base;
    a = next;
  }
  // hd.tail = inv_base * perm[0] / (1 - inv_base), evaluated once per dimension on the host with the same three f64 operations
  return fmin(inv_base_n * ((double)reversed + hd.tail), 0.99999999999999989);
}
// Halton::sample_dimension halton.rs:107-128 (index < 2^32 checked on the host)
constexpr uint32_t kErrHaltonDims = 32u;   // ERR_HALTON_DIMS of dkernels.hpp
template <typename R>
RRT_DEV double halton_dim(const SceneDev<R>& s, uint32_t index, uint32_t dim) {
  // permutation_for_dimension halton.rs:63-69: dimension 1000 and beyond panics (PRIME_TABLE_SIZE). A DirectLighting / Debug tree
  // over smooth glass reaches it at depth ~7 (16 dimensions per vertex, 2^depth vertices)
  if (dim >= 1000u) { atomicOr(s.err, kErrHaltonDims); return 0.0; }
  if (s.sample_at_center && dim < 2) return 0.5;
  if (dim == 0) {
    uint32_t a = index >> s.base_exp0;                     // radical_inverse(0, a) = reverse_bits_64(a) * 2^-64
    return (double)__brev(a) * 2.3283064365386963e-10;     // = rev32(a) * 2^-32 exactly, a < 2^32
  }
  if (dim == 1) return radical_inverse_dev(index / s.base_scale1, s.hdims[1], s.fast_div);
  const HaltonDim hd = s.hdims[dim];
  if (dim < s.n_hblk) {
    // Block tables (SceneDev::hblk): reversed = rev(low block) * base^(digits of hi) + rev(hi) is the integer the loop builds digit by digit,
    // and the f64 power is the loop's own running product - the same value, bit for bit; an index below one block takes the loop.
    const HaltonBlk hb = s.hblk[dim];
    if (hb.block != 0u && index >= hb.block) {
      const uint32_t t = __umulhi(hb.magic, index);
      const uint32_t hi = (t + ((index - t) >> 1)) >> hb.shift, lo = index - hi * hb.block;
      const uint4 e = s.hhi[hb.hi_off + hi];
      const uint64_t reversed = (uint64_t)s.hlo[hb.lo_off + lo] * e.y + e.x;
      return fmin(__hiloint2double((int)e.w, (int)e.z) * ((double)reversed + hd.tail), 0.99999999999999989);
    }
  }
  return scrambled_radical_inverse_dev(index, hd, s.perms + hd.perm_offset, s.fast_div);
}
// dims 2 / 3 (lens sample) on the fast path: same digits, same permutation, same f64 value as
// scrambled_radical_inverse_dev — the product inv_base^k comes from a table built with the identical multiplications
template <typename R>
RRT_DEV double halton_cam_dim(const SceneDev<R>& s, uint32_t index, int which) {
  const HaltonDim hd = s.hdims[2 + which];
  const uint32_t packed = s.cam_perm[which];
  uint64_t reversed = 0;
  uint32_t a = index, k = 0;
  while (a > 0) {
    const uint32_t next = div_base(a, hd, s.fast_div);
    const uint32_t digit = a - next * hd.base;
    reversed = reversed * hd.base + ((packed >> (3u * digit)) & 7u);
    a = next;
    k++;
  }
  return fmin(s.cam_invpow[which][k] * ((double)reversed + s.cam_tail[which]), 0.99999999999999989);
}
// The four camera dimensions of a Halton sample for the fp32 camera kernel: the bases are the first primes (2, 3, 5, 7) whatever the scene,
// so a / base is a multiply-high by a constant plus shifts, exact for every 32-bit a (Granlund-Montgomery), and q * base a shift-add -
// a third of the instructions of the general digit loop (64-bit magic products / an f64 reciprocal per digit). Digits, permutations and
// the f64 values are those of halton_dim() / halton_cam_dim(): the products inv_base^k come from tables built with the same
// multiplications (SceneDev::cam_invpow, inv3pow).
RRT_DEV uint32_t div3(uint32_t a) { return __umulhi(a, 0xAAAAAAABu) >> 1; }
RRT_DEV uint32_t div5(uint32_t a) { return __umulhi(a, 0xCCCCCCCDu) >> 2; }
RRT_DEV uint32_t div7(uint32_t a) { const uint32_t q = __umulhi(a, 0x24924925u); return (((a - q) >> 1) + q) >> 2; }
constexpr uint32_t kCamB3 = 729u, kCamB5 = 15625u, kCamB7 = 16807u;   // 3^6, 5^6, 7^5: low-digit blocks of SceneDev::cam_lo / cam_hi
template <typename R>
RRT_DEV void halton_cam4(const SceneDev<R>& s, uint32_t index, double* d0, double* d1, double* d2, double* d3) {
  // With the tables the digits of an index come in two blocks: reversed = rev(low block) * base^(digits of hi) + rev(hi) - the same
  // integer the loop builds digit by digit - times the same tabulated f64 power. An index below one block takes the loop.
  const bool tab = s.cam_lo[0] != nullptr;
  if (s.sample_at_center) { *d0 = 0.5; *d1 = 0.5; }
  else {
    *d0 = (double)__brev(index >> s.base_exp0) * 2.3283064365386963e-10;
    uint32_t a = (uint32_t)((double)index * s.inv_base_scale1), rev = 0, k = 0;   // index / 3^base_exponents[1], exact after the fix-up (cf. div_base())
    { const uint32_t r = index - a * s.base_scale1; if ((int32_t)r < 0) a -= 1u; else if (r >= s.base_scale1) a += 1u; }
    if (tab && a >= kCamB3) {
      const uint32_t hi = a / kCamB3, lo = a - hi * kCamB3;
      const uint4 e = s.cam_hi[0][hi];
      rev = s.cam_lo[0][lo] * e.y + e.x;
      *d1 = fmin((double)rev * __hiloint2double((int)e.w, (int)e.z), 0.99999999999999989);
    } else {
      while (a != 0) { const uint32_t q = div3(a); rev = rev * 3u + (a - q * 3u); a = q; k++; }
      *d1 = fmin((double)rev * s.inv3pow[k], 0.99999999999999989);
    }
  }
  if (tab && index >= kCamB5) {
    const uint32_t hi = index / kCamB5, lo = index - hi * kCamB5;
    const uint4 e = s.cam_hi[1][hi];
    const uint64_t reversed = (uint64_t)s.cam_lo[1][lo] * e.y + e.x;
    *d2 = fmin(__hiloint2double((int)e.w, (int)e.z) * ((double)reversed + s.cam_tail[0]), 0.99999999999999989);
  } else {
    const uint32_t packed = s.cam_perm[0];
    uint64_t reversed = 0;
    uint32_t a = index, k = 0;
    while (a != 0) { const uint32_t q = div5(a), digit = a - q * 5u; reversed = reversed * 5u + ((packed >> (3u * digit)) & 7u); a = q; k++; }
    *d2 = fmin(s.cam_invpow[0][k] * ((double)reversed + s.cam_tail[0]), 0.99999999999999989);
  }
  if (tab && index >= kCamB7) {
    const uint32_t hi = index / kCamB7, lo = index - hi * kCamB7;
    const uint4 e = s.cam_hi[2][hi];
    const uint64_t reversed = (uint64_t)s.cam_lo[2][lo] * e.y + e.x;
    *d3 = fmin(__hiloint2double((int)e.w, (int)e.z) * ((double)reversed + s.cam_tail[1]), 0.99999999999999989);
  } else {
    const uint32_t packed = s.cam_perm[1];
    uint64_t reversed = 0;
    uint32_t a = index, k = 0;
    while (a != 0) { const uint32_t q = div7(a), digit = a - q * 7u; reversed = reversed * 7u + ((packed >> (3u * digit)) & 7u); a = q; k++; }
    *d3 = fmin(s.cam_invpow[1][k] * ((double)reversed + s.cam_tail[1]), 0.99999999999999989);
  }
}
// ---- StratifiedSampler (samplers/stratified.rs, samplers/mod.rs:191-227) ---------------------------------------------
// The reference fills, per pixel and per sampled dimension, an array of jittered strata and Fisher-Yates-shuffles it,
// all with rand::thread_rng (not reproducible); dimensions beyond `dimension` draw rng.gen_range(-1.0..1.0) - note the
// range. Here the same structure is driven by counter-based randomness, so a sample needs no per-pixel table:
//   stratum of sample s in dimension k of pixel P = permute(s, spp, key(P, k))   (Kensler's keyed bijection = the shuffle)
//   jitter = rand(key, stratum); out-of-range dimensions = 2 * rand(P, s, k) - 1.
// The oracle restates exactly this (bit-exact parity); against the reference it is equal in distribution only.
RRT_DEV uint32_t st_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
RRT_DEV uint32_t st_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t tag) { return st_mix(st_mix(pixel ^ seed_lo) + tag * 0x9e3779b9u + seed_hi); }
RRT_DEV uint32_t st_permute(uint32_t i, uint32_t n, uint32_t p) {   // Kensler, "Correlated Multi-Jittered Sampling" (2013)
  uint32_t w = n - 1;
  w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
  do {
    i ^= p; i *= 0xe170893du; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3fu; i ^= p >> 23; i ^= (i & w) >> 1;
    i *= 1u | p >> 27; i *= 0x6935fa69u; i ^= (i & w) >> 11; i *= 0x74dcb303u; i ^= (i & w) >> 2; i *= 0x9e501cc3u;
    i ^= (i & w) >> 2; i *= 0xc860a3dfu; i &= w; i ^= i >> 5;
  } while (i >= n);
  return (i + p) % n;
}
RRT_DEV double st_rand(uint32_t key, uint32_t i) { return (double)st_mix(key ^ st_mix(i + 0x632be5abu)) * 2.3283064365386963e-10; }   // [0, 1)
// index word of a stratified sample: pixel (22 bits) << 10 | sample number (10 bits); d = 1D counter | 2D counter << kStBits
// A path's sampler position rides in its queue entry as one word: the dimension counter(s) in the low SceneDev::db_shift bits, the bounce (DirectLighting: depth - 1)
// above them (db_pack). HaltonSampler: one counter below 1 000 in 16 bits, 65 535 bounces (a mirror box reaches the reference's 1 000-dimension panic at bounce ~498).
// StratifiedSampler: two counters of kStBits each, 255 bounces. [r4] The stratified counters were 8 bits each (a sample that drew more than 255 1D or 2D dimensions - deep
// DirectLighting / Debug trees do - was refused); now 4 095 each.
constexpr uint32_t kStBits = 12u, kStMask = (1u << kStBits) - 1u;
constexpr uint32_t kDbShiftHalton = 16u, kDbShiftStratified = 2u * kStBits, kDbMaxBounceStratified = (1u << (32u - kDbShiftStratified)) - 1u;
template <typename S> RRT_DEV uint32_t db_dim(const S& s, uint32_t db) { return db & ((1u << s.db_shift) - 1u); }
template <typename S> RRT_DEV uint32_t db_bounce(const S& s, uint32_t db) { return db >> s.db_shift; }
template <typename S> RRT_DEV uint32_t db_pack(const S& s, uint32_t dim, uint32_t bounce) { return (dim & ((1u << s.db_shift) - 1u)) | (bounce << s.db_shift); }
constexpr uint32_t kErrStDims = 64u;   // ERR_ST_DIMS of dkernels.hpp: a sample ran out of the 12-bit dimension counters (device limit, not a reference panic)
template <typename R> RRT_DEV double st_get_1d(const SceneDev<R>& s, uint32_t index, uint32_t* d) {
  const uint32_t pixel = index >> 10, sn = index & 1023u, k = *d & kStMask;
  if (k == kStMask) atomicOr(s.err, kErrStDims);
  *d = (*d & ~kStMask) | ((k + 1u) & kStMask);
  if (k >= s.st_dims) return 2.0 * st_rand(st_key(s.st_seed_lo, s.st_seed_hi, pixel, 0x10000u + k), sn) - 1.0;
  const uint32_t spp = s.st_nx * s.st_ny, key = st_key(s.st_seed_lo, s.st_seed_hi, pixel, k);
  const uint32_t j = st_permute(sn, spp, key);
  const double delta = s.st_jitter ? st_rand(key, j) : 0.5;
  return fmin(((double)j + delta) * (1.0 / (double)spp), 0.99999999999999989);
}
template <typename R> RRT_DEV void st_get_2d(const SceneDev<R>& s, uint32_t index, uint32_t* d, double* a, double* b) {
  const uint32_t pixel = index >> 10, sn = index & 1023u, k = (*d >> kStBits) & kStMask;
  if (k == kStMask) atomicOr(s.err, kErrStDims);
  *d = (*d & ~(kStMask << kStBits)) | (((k + 1u) & kStMask) << kStBits);
  if (k >= s.st_dims) {
    const uint32_t key = st_key(s.st_seed_lo, s.st_seed_hi, pixel, 0x20000u + k);
    *a = 2.0 * st_rand(key, 2u * sn) - 1.0; *b = 2.0 * st_rand(key, 2u * sn + 1u) - 1.0;
    return;
  }
  const uint32_t spp = s.st_nx * s.st_ny, key = st_key(s.st_seed_lo, s.st_seed_hi, pixel, 0x1000u + k);
  const uint32_t j = st_permute(sn, spp, key), x = j % s.st_nx, y = j / s.st_nx;
  const double jx = s.st_jitter ? st_rand(key, 2u * j) : 0.5, jy = s.st_jitter ? st_rand(key, 2u * j + 1u) : 0.5;
  *a = fmin(((double)x + jx) * (1.0 / (double)s.st_nx), 0.99999999999999989);
  *b = fmin(((double)y + jy) * (1.0 / (double)s.st_ny), 0.99999999999999989);
}
// ISampler::get_1d / get_2d of the scene's sampler. `d` = db_dim() of the queue entry's counter word.
template <typename R> RRT_DEV double draw_1d(const SceneDev<R>& s, uint32_t index, uint32_t* d) {
  if (s.sampler_type == 1u) return st_get_1d(s, index, d);
  const double v = halton_dim(s, index, *d);
  *d += 1u;
  return v;
}
template <typename R> RRT_DEV void draw_2d(const SceneDev<R>& s, uint32_t index, uint32_t* d, double* a, double* b) {
  if (s.sampler_type == 1u) { st_get_2d(s, index, d, a, b); return; }
  *a = halton_dim(s, index, *d); *b = halton_dim(s, index, *d + 1u);
  *d += 2u;
}
// a 2D draw whose value is never read (u_scattering of the removed BSDF-sampling half): only the counters move
template <typename R> RRT_DEV void skip_2d(const SceneDev<R>& s, uint32_t* d) {
  if (s.sampler_type == 1u) { const uint32_t k = (*d >> kStBits) & kStMask; if (k == kStMask) atomicOr(s.err, kErrStDims); *d = (*d & ~(kStMask << kStBits)) | (((k + 1u) & kStMask) << kStBits); }
  else { if (*d + 1u >= 1000u) atomicOr(s.err, kErrHaltonDims); *d += 2u; }
}
template <typename R> RRT_DEV void skip_1d(const SceneDev<R>& s, uint32_t* d) {
  if (s.sampler_type == 1u) { const uint32_t k = *d & kStMask; if (k == kStMask) atomicOr(s.err, kErrStDims); *d = (*d & ~kStMask) | ((k + 1u) & kStMask); }
  else { if (*d >= 1000u) atomicOr(s.err, kErrHaltonDims); *d += 1u; }   // (the reference computes the value, so it panics here too)
}
template <typename R> RRT_DEV R to_real(double u) { return (R)u; }
template <> RRT_DEV float to_real<float>(double u) { return fminf((float)u, Const<float>::one_minus_eps); }  // keep u < 1 after narrowing

// inverse_radical_inverse lowdiscrepancy.rs:239-248
RRT_DEV uint32_t inverse_radical_inverse_dev(uint32_t base, uint32_t inverse, uint32_t n_digits) {
  uint32_t index = 0;
  for (uint32_t i = 0; i < n_digits; i++) {
    uint32_t digit = inverse % base;
    inverse /= base;
    index = index * base + digit;
  }
  return index;
}
// Halton::get_index_for_sample halton.rs:75-105 (dim 0 uses base_exponents[1] digits: Q24), pixel >= 0
template <typename R>
RRT_DEV uint32_t halton_pixel_offset(const SceneDev<R>& s, uint32_t px, uint32_t py) {
  if (s.stride <= 1) return 0;
  uint32_t pmx = px % 128u, pmy = py % 128u;  // K_MAX_RESOLUTION
  uint64_t off = 0;
  off += (uint64_t)inverse_radical_inverse_dev(2, pmx, s.base_exp1) * (uint64_t)(s.stride / s.base_scale0) * (uint64_t)s.mult_inv0;
  off += (uint64_t)inverse_radical_inverse_dev(3, pmy, s.base_exp1) * (uint64_t)(s.stride / s.base_scale1) * (uint64_t)s.mult_inv1;
  return (uint32_t)(off % (uint64_t)s.stride);
}

// ---- sampling.rs ------------------------------------------------------------------------------------------
template <typename R> RRT_DEV V3<R> uniform_sample_sphere(R u0, R u1) {  // :233-243
  R z = R(1) - R(2) * u0;
  R r = sqrt(rmax(R(0), R(1) - z * z));
  R phi = R(2) * R(RRT_PI) * u1;
  return {r * R(cos(phi)), r * R(sin(phi)), z};
}
template <typename R> RRT_DEV V3<R> cosine_sample_hemisphere(R u0, R u1) {  // :270-304
  R ox = u0 * R(2) - R(1), oy = u1 * R(2) - R(1);
  R dx, dy;
  if (ox == R(0) && oy == R(0)) { dx = R(0); dy = R(0); }
  else {
    R theta, r;
    if (rabs(ox) > rabs(oy)) { r = ox; theta = R(RRT_PI_OVER_4) * (oy / ox); }
    else { r = oy; theta = R(RRT_PI_OVER_2) - R(RRT_PI_OVER_4) * (ox / oy); }
    dx = R(cos(theta)) * r; dy = R(sin(theta)) * r;
  }
  R z = sqrt(rmax(R(0), R(1) - dx * dx - dy * dy));
  return {dx, dy, z};
}
template <typename R> RRT_DEV R power_heuristic1(R fp, R gp) { return (fp * fp) / (fp * fp + gp * gp); }  // :324-328, nf = ng = 1

// ---- RealisticCamera (camera.rs) -----------------------------------------------------------------------------
template <typename R> struct RayT { V3<R> o, d; };

template <typename R> RRT_DEV bool refract(V3<R> wi, V3<R> n, R eta, V3<R>* wt) {  // reflection.rs:122-134
  R cos_i = dot(n, wi);
  R sin2_i = rmax(R(0), R(1) - cos_i * cos_i);
  R sin2_t = eta * eta * sin2_i;
  if (sin2_t >= R(1)) return false;
  R cos_t = sqrt(R(1) - sin2_t);
  *wt = (-wi) * eta + n * (eta * cos_i - cos_t);
  return true;
}
// Transform::scale(1,1,-1).t(ray) then Ray::new: (x, y, -z), direction normalised twice (transform.rs:525-537)
template <typename R> RRT_DEV RayT<R> flip_z(const RayT<R>& r) {
  RayT<R> o;
  o.o = V3<R>(r.o.x, r.o.y, -r.o.z);
  o.d = vnormalize(vnormalize(V3<R>(r.d.x, r.d.y, -r.d.z)));
  return o;
}
// intersect_spherical_element camera.rs:220-253
template <typename R> RRT_DEV bool intersect_spherical(R radius, R z_center, const RayT<R>& ray, R* t, V3<R>* n) {
  V3<R> o = ray.o - V3<R>(R(0), R(0), z_center);
  R a = ray.d.x * ray.d.x + ray.d.y * ray.d.y + ray.d.z * ray.d.z;
  R b = R(2) * (ray.d.x * o.x + ray.d.y * o.y + ray.d.z * o.z);
  R c = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
  R t0 = 0, t1 = 0;
  if (!quadratic(a, b, c, &t0, &t1)) return false;
  bool use_closer = (ray.d.z > R(0)) ^ (radius < R(0));
  *t = use_closer ? rmin(t0, t1) : rmax(t0, t1);
  if (*t < R(0)) return false;
  V3<R> nn = o + ray.d * *t;
  *n = faceforward(nnormalize(nn), -ray.d);
  return true;
}
// trace_lenses_from_film camera.rs:156-219. `t >= 0` is an assert in the reference (never observed to fail
// for rays that pass the aperture tests); a failing lane is treated as a lens miss.
template <typename R> RRT_DEV bool trace_from_film(const SceneDev<R>& s, const RayT<R>& r_camera, RayT<R>* out) {
  R element_z = R(0);
  RayT<R> r = flip_z(r_camera);
  for (int i = s.n_lens - 1; i >= 0; i--) {
    const LensElem<R> el = s.lens[i];
    element_z -= el.thickness;
    R t = R(0);
    V3<R> n;
    const bool is_stop = el.curvature_radius == R(0);
    if (is_stop) {
      if (r.d.z >= R(0)) return false;
      t = (element_z - r.o.z) / r.d.z;
    } else {
      if (!intersect_spherical(el.curvature_radius, element_z + el.curvature_radius, r, &t, &n)) return false;
    }
    if (!(t >= R(0))) return false;
    V3<R> p_hit = r.o + r.d * t;
    R r2 = p_hit.x * p_hit.x + p_hit.y * p_hit.y;
    if (r2 >= el.aperture_radius * el.aperture_radius) return false;
    r.o = p_hit;
    if (!is_stop) {
      V3<R> w;
      R eta_i = el.eta;
      R eta_prev = (i > 0) ? s.lens[i - 1].eta : R(0);
      R eta_t = (i > 0 && eta_prev != R(0)) ? eta_prev : R(1);
      if (!refract(vnormalize(-r.d), n, eta_i / eta_t, &w)) return false;
      r.d = w;
    }
  }
  *out = flip_z(r);
  return true;
}
// generate_ray camera.rs:534-580 (weight; ray in world space)
template <typename R> RRT_DEV R generate_ray(const SceneDev<R>& s, R pfx, R pfy, R lx, R ly, RayT<R>* ray) {
  R sx = pfx / (R)s.xres, sy = pfy / (R)s.yres;
  R p2x = s.extent[0] * (R(1) - sx) + s.extent[2] * sx, p2y = s.extent[1] * (R(1) - sy) + s.extent[3] * sy;
  V3<R> p_film(-p2x, p2y, R(0));
  // sample_exit_pupil :492-521 — `(r / (diag/2)) as usize * 64` (Q6) is 0 unless r >= diag/2, then clamps to 63
  R r_film = sqrt(p_film.x * p_film.x + p_film.y * p_film.y);
  const R* pb = (r_film / (s.diagonal / R(2)) >= R(1)) ? s.pupil63 : s.pupil0;
  R plx = pb[0] * (R(1) - lx) + pb[2] * lx, ply = pb[1] * (R(1) - ly) + pb[3] * ly;
  R sin_t = r_film != R(0) ? p_film.y / r_film : R(0), cos_t = r_film != R(0) ? p_film.x / r_film : R(1);
  R area = (pb[2] - pb[0]) * (pb[3] - pb[1]);
  V3<R> p_rear(cos_t * plx - sin_t * ply, sin_t * plx + cos_t * ply, s.lens[s.n_lens - 1].thickness);
  RayT<R> r_film_ray;
  r_film_ray.o = p_film;
  r_film_ray.d = vnormalize(p_rear - p_film);
  RayT<R> r;
  if (!trace_from_film(s, r_film_ray, &r)) return R(0);
  // camera_to_world.t(ray) (double normalise), then ray.d.normalize()
  ray->o = aff_pt(s.cam_m, r.o);
  ray->d = vnormalize(vnormalize(vnormalize(aff_vec(s.cam_m, r.d))));
  R cos_theta = vnormalize(r_film_ray.d).z;
  R cos4 = (cos_theta * cos_theta) * (cos_theta * cos_theta);
  if (s.simple_weighting) return cos4 * area / ((s.pupil0[2] - s.pupil0[0]) * (s.pupil0[3] - s.pupil0[1]));
  R rz = s.lens[s.n_lens - 1].thickness;
  return (s.shutter_close - s.shutter_open) * (cos4 * area) / rz * rz;
}
// generate_ray_differential camera.rs:582-628: the auxiliary rays only decide whether the weight survives
template <typename R> RRT_DEV R generate_ray_differential(const SceneDev<R>& s, R pfx, R pfy, R lx, R ly, RayT<R>* ray) {
  R wt = generate_ray(s, pfx, pfy, lx, ly, ray);
  if (wt == R(0)) return R(0);
  RayT<R> aux;
  R wtx = generate_ray(s, pfx + R(0.05), pfy, lx, ly, &aux);
  if (wtx == R(0)) wtx = generate_ray(s, pfx + R(-0.05), pfy, lx, ly, &aux);
  if (wtx == R(0)) return R(0);
  R wty = generate_ray(s, pfx, pfy + R(0.05), lx, ly, &aux);
  if (wty == R(0)) wty = generate_ray(s, pfx, pfy + R(-0.05), lx, ly, &aux);
  if (wty == R(0)) return R(0);
  return wt;
}

// ---- BxDFs (reflection.rs, microfacet.rs) ---------------------------------------------------------------------
enum : uint32_t { BXDF_REFLECTION = 1, BXDF_TRANSMISSION = 2, BXDF_DIFFUSE = 4, BXDF_GLOSSY = 8, BXDF_SPECULAR = 16, BXDF_ALL = 31, BXDF_NONE = 0 };
enum : uint32_t { LOBE_LAMBERT = 0, LOBE_OREN_NAYAR, LOBE_MICROFACET, LOBE_SPEC_REFL, LOBE_DEBUG_DIFFUSE, LOBE_DEBUG_SPECULAR,
                  LOBE_SPEC_TRANS, LOBE_FRESNEL_SPEC, LOBE_LAMBERT_TRANS, LOBE_MICROFACET_TRANS, LOBE_NONE = 0xffffffffu };
enum : uint32_t { FR_NOOP = 0, FR_DIELECTRIC, FR_CONDUCTOR };

template <typename R> RRT_DEV R cos_theta(V3<R> w) { return w.z; }
template <typename R> RRT_DEV R cos2_theta(V3<R> w) { return w.z * w.z; }
template <typename R> RRT_DEV R abs_cos_theta(V3<R> w) { return rabs(w.z); }
template <typename R> RRT_DEV R sin2_theta(V3<R> w) { return rmax(R(0), R(1) - cos2_theta(w)); }
template <typename R> RRT_DEV R sin_theta(V3<R> w) { return sqrt(sin2_theta(w)); }
template <typename R> RRT_DEV R tan_theta(V3<R> w) { return sin_theta(w) / cos_theta(w); }
template <typename R> RRT_DEV R tan2_theta(V3<R> w) { return sin2_theta(w) / cos2_theta(w); }
template <typename R> RRT_DEV R cos_phi(V3<R> w) { R st = sin_theta(w); return st == R(0) ? R(1) : clampr(w.x / st, R(-1), R(1)); }
template <typename R> RRT_DEV R sin_phi(V3<R> w) { R st = sin_theta(w); return st == R(0) ? R(0) : clampr(w.y / st, R(-1), R(1)); }
template <typename R> RRT_DEV bool same_hemisphere(V3<R> w, V3<R> wp) { return w.z * wp.z > R(0); }
template <typename R> RRT_DEV V3<R> reflect(V3<R> wo, V3<R> n) { return -wo + n * R(2) * dot(wo, n); }

template <typename R> RRT_DEV R fr_dielectric(R cos_i, R eta_i, R eta_t) {  // reflection.rs:145-168
  cos_i = clampr(cos_i, R(-1), R(1));
  if (!(cos_i > R(0))) { R t = eta_i; eta_i = eta_t; eta_t = t; cos_i = rabs(cos_i); }
  R sin_i = sqrt(rmax(R(0), R(1) - cos_i * cos_i));
  R sin_t = eta_i / eta_t * sin_i;
  if (sin_t >= R(1)) return R(1);
  R cos_t = sqrt(rmax(R(0), R(1) - sin_t * sin_t));
  R r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
  R r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
  return (r_parl * r_parl + r_perp * r_perp) / R(2);
}
template <typename R> RRT_DEV Rgb<R> fr_conductor(R cos_i_in, Rgb<R> eta_i, Rgb<R> eta_t, Rgb<R> k) {  // reflection.rs:170-195
  R cos_i = clampr(cos_i_in, R(-1), R(1));
  Rgb<R> eta = eta_t / eta_i, eta_k = k / eta_i;
  R cos2 = cos_i * cos_i, sin2 = R(1) - cos2;
  Rgb<R> eta2 = eta * eta, eta_k2 = eta_k * eta_k;
  Rgb<R> t0 = eta2 - eta_k2 - Rgb<R>(sin2);
  Rgb<R> a2_plus_b2 = rgb_sqrt(t0 * t0 + eta2 * eta_k2 * Rgb<R>(R(4)));
  Rgb<R> t1 = a2_plus_b2 + Rgb<R>(cos2);
  Rgb<R> a = rgb_sqrt((a2_plus_b2 + t0) * R(0.5));
  Rgb<R> t2 = a * R(2) * cos_i;
  Rgb<R> rs = (t1 - t2) / (t1 + t2);
  Rgb<R> t3 = a2_plus_b2 * cos2 + Rgb<R>(sin2 * sin2);
  Rgb<R> t4 = t2 * sin2;
  Rgb<R> rp = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) * Rgb<R>(R(0.5));
}

// tuning variants: the microfacet helpers as real calls (code size of the glossy / general shading kernels against the 64 KB instruction cache two CUs share)
#ifdef RRT_MICRO_NOINLINE
#define RRT_MICRO __attribute__((noinline)) __device__
#else
#define RRT_MICRO RRT_DEV
#endif

template <typename R>
struct Lobe {
  uint32_t kind, type, fr;
  Rgb<R> r;
  R a, b;              // OrenNayar A, B; transmissive lobes: eta_a, eta_b
  R alpha_x, alpha_y;  // TrowbridgeReitz, sample_visible_area = true
  Rgb<R> eta_i, eta_t, k;   // transmissive lobes keep T in `r` (FresnelSpecular: R in `r`, T in `k`)
};

// TrowbridgeReitzDistribution microfacet.rs:253-425
// (the arithmetic takes its parameters by value: as real calls - RRT_MICRO_NOINLINE - nothing of the Lobe has to live in memory)
template <typename R> RRT_MICRO R tr_d_v(R alpha_x, R alpha_y, V3<R> wh) {
  R tan2 = tan2_theta(wh);
  if (isinf(tan2)) return R(0);
  R cos4 = cos2_theta(wh) * cos2_theta(wh);
  R cp = cos_phi(wh), sp = sin_phi(wh);
  R e = ((cp * cp) / (alpha_x * alpha_x) + (sp * sp) / (alpha_y * alpha_y)) * tan2;
  return R(1) / (R(RRT_PI) * alpha_x * alpha_y * cos4 * (R(1) + e) * (R(1) + e));
}
template <typename R> RRT_DEV R tr_d(const Lobe<R>& l, V3<R> wh) { return tr_d_v(l.alpha_x, l.alpha_y, wh); }
template <typename R> RRT_MICRO R tr_lambda_v(R alpha_x, R alpha_y, V3<R> w) {
  R abs_tan = rabs(tan_theta(w));
  if (isinf(abs_tan)) return R(0);
  R cp = cos_phi(w), sp = sin_phi(w);
  R alpha = sqrt((cp * cp) * (alpha_x * alpha_x) + (sp * sp) * (alpha_y * alpha_y));
  R a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
  return (R(-1) + R(sqrt(R(1) + a2t2))) / R(2);
}
template <typename R> RRT_DEV R tr_lambda(const Lobe<R>& l, V3<R> w) { return tr_lambda_v(l.alpha_x, l.alpha_y, w); }
template <typename R> RRT_DEV R tr_g1(const Lobe<R>& l, V3<R> w) { return R(1) / (R(1) + tr_lambda(l, w)); }
template <typename R> RRT_DEV R tr_g(const Lobe<R>& l, V3<R> wo, V3<R> wi) { return R(1) / (R(1) + tr_lambda(l, wo) + tr_lambda(l, wi)); }
template <typename R> RRT_DEV R tr_pdf(const Lobe<R>& l, V3<R> wo, V3<R> wh) { return tr_d(l, wh) * tr_g1(l, wo) * absdot(wo, wh) / abs_cos_theta(wo); }
template <typename R> RRT_DEV void tr_sample_11(R cos_t, R u1, R u2, R* slope_x, R* slope_y) {  // :268-323
  if (cos_t > R(0.9999)) {
    R r = sqrt(u1 / (R(1) - u1));
    R phi = R(6.28318530718) * u2;
    *slope_x = r * R(cos(phi));
    *slope_y = r * R(sin(phi));
    return;
  }
  R sin_t = sqrt(rmax(R(0), R(1) - cos_t * cos_t));
  R tan_t = sin_t / cos_t;
  R a = R(1) / tan_t;
  R g1 = R(2) / (R(1) + R(sqrt(R(1) + R(1) / (a * a))));
  a = R(2) * u1 / g1 - R(1);
  R tmp = R(1) / (a * a - R(1));
  if (tmp > R(1e10)) tmp = R(1e10);
  R b = tan_t;
  R d = sqrt(rmax(b * b * tmp * tmp - (a * a - b * b) * tmp, R(0)));
  R sx1 = b * tmp - d, sx2 = b * tmp + d;
  *slope_x = (a < R(0) || sx2 > R(1) / tan_t) ? sx1 : sx2;
  R sg, nu2;
  if (u2 > R(0.5)) { sg = R(1); nu2 = R(2) * (u2 - R(0.5)); } else { sg = R(-1); nu2 = R(2) * (R(0.5) - u2); }
  R z = (nu2 * (nu2 * (nu2 * R(0.27385) - R(0.73369)) + R(0.46341))) /
        (nu2 * (nu2 * (nu2 * R(0.093073) + R(0.309420)) - R(1)) + R(0.597999));
  *slope_y = sg * z * R(sqrt(R(1) + *slope_x * *slope_x));
}
template <typename R> RRT_MICRO V3<R> tr_sample(V3<R> wi, R ax, R ay, R u1, R u2) {  // :325-363
  V3<R> ws = vnormalize(V3<R>(ax * wi.x, ay * wi.y, wi.z));
  R sx = 0, sy = 0;
  tr_sample_11(cos_theta(ws), u1, u2, &sx, &sy);
  R tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
  sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
  sx = tmp;
  sx *= ax; sy *= ay;
  return vnormalize(V3<R>(-sx, -sy, R(1)));
}
template <typename R> RRT_DEV V3<R> tr_sample_wh(const Lobe<R>& l, V3<R> wo, R u0, R u1) {  // :387-421
  if (wo.z < R(0)) return -tr_sample(-wo, l.alpha_x, l.alpha_y, u0, u1);
  return tr_sample(wo, l.alpha_x, l.alpha_y, u0, u1);
}
template <typename R> RRT_MICRO Rgb<R> fresnel_eval_v(uint32_t fr, Rgb<R> eta_i, Rgb<R> eta_t, Rgb<R> k, R cos_i) {  // reflection.rs:599-615
  if (fr == FR_NOOP) return Rgb<R>(R(1));
  if (fr == FR_DIELECTRIC) return Rgb<R>(fr_dielectric(cos_i, eta_i.r, eta_t.r));
  return fr_conductor(rabs(cos_i), eta_i, eta_t, k);
}
template <typename R> RRT_DEV Rgb<R> fresnel_eval(const Lobe<R>& l, R cos_i) { return fresnel_eval_v(l.fr, l.eta_i, l.eta_t, l.k, cos_i); }
// Lobe-kind sets. A scene's materials fix which BxDFs can ever exist at its hits; the shading kernels are instantiated for a few such sets
// (KM = bit mask over LOBE_*, ~0u = every kind) so that the code - and above all the registers - of the kinds a scene cannot produce are not
// part of its kernel (Lambert-only scenes: 128 -> 99 VGPRs, 4 -> 5 waves per SIMD in the latency-bound path shading kernel). Same arithmetic
// per kind in every instantiation. The host picks the set from the materials the aggregate uses (rrt_impl.hpp shade_spec()); build_lobes
// drops a lobe whose kind is outside the kernel's set and raises an error flag instead of running into the `unreachable` below.
constexpr uint32_t kAllKinds = 0xffffffffu;
constexpr uint32_t kind_bit(uint32_t k) { return 1u << k; }
constexpr uint32_t kKindsLambert = kind_bit(LOBE_LAMBERT);
constexpr uint32_t kKindsGlossy = kind_bit(LOBE_LAMBERT) | kind_bit(LOBE_OREN_NAYAR) | kind_bit(LOBE_MICROFACET);
template <uint32_t KM> RRT_DEV bool kind_in(uint32_t kind) { return KM == kAllKinds || kind == LOBE_NONE || (kind < 32u && ((KM >> kind) & 1u) != 0u); }
#define RRT_KIND_SET(KM, l) do { if (!kind_in<KM>((l).kind)) __builtin_unreachable(); } while (0)

template <typename R, uint32_t KM = kAllKinds> RRT_DEV Rgb<R> lobe_f(const Lobe<R>& l, V3<R> wo, V3<R> wi) {
  RRT_KIND_SET(KM, l);
  switch (l.kind) {
    case LOBE_LAMBERT: return l.r / R(RRT_PI);
    case LOBE_OREN_NAYAR: {  // reflection.rs:917-941
      R sin_i = sin_theta(wi), sin_o = sin_theta(wo), max_cos = R(0);
      if (sin_i > R(1e-4) && sin_o > R(1e-4)) {
        R d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
        max_cos = rmax(d_cos, R(0));
      }
      R sin_alpha, tan_beta;
      if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_o; tan_beta = sin_i / abs_cos_theta(wi); }
      else { sin_alpha = sin_i; tan_beta = sin_o / abs_cos_theta(wo); }
      return l.r / R(RRT_PI) * (l.a + l.b * max_cos * sin_alpha * tan_beta);
    }
    case LOBE_MICROFACET: {  // reflection.rs:971-992
      R cos_o = abs_cos_theta(wo), cos_i = abs_cos_theta(wi);
      V3<R> wh = wi + wo;
      if (cos_i == R(0) || cos_o == R(0)) return Rgb<R>();
      if (wh.x == R(0) && wh.y == R(0) && wh.z == R(0)) return Rgb<R>();
      wh = vnormalize(wh);
      Rgb<R> f = fresnel_eval(l, dot(wi, faceforward(wh, V3<R>(R(0), R(0), R(1)))));
      return l.r * tr_d(l, wh) * tr_g(l, wo, wi) * f / (R(4) * cos_i * cos_o);
    }
    case LOBE_DEBUG_DIFFUSE: return Rgb<R>(R(0), R(1), R(0));
    case LOBE_DEBUG_SPECULAR: return Rgb<R>(R(0), R(0), R(1));
    case LOBE_LAMBERT_TRANS: return l.r / R(RRT_PI);   // reflection.rs:854-856
    case LOBE_MICROFACET_TRANS: {   // reflection.rs:1059-1097, mode = Radiance
      if (same_hemisphere(wo, wi)) return Rgb<R>();
      const R cos_o = cos_theta(wo), cos_i = cos_theta(wi);
      if (cos_i == R(0) || cos_o == R(0)) return Rgb<R>();
      const R eta = cos_theta(wo) > R(0) ? l.b / l.a : l.a / l.b;
      V3<R> wh = vnormalize(wo + wi * eta);
      if (wh.z < R(0)) wh = -wh;
      const R f = fr_dielectric(dot(wo, wh), l.a, l.b);
      const R sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
      const R factor = R(1) / eta;
      return (Rgb<R>(R(1)) - Rgb<R>(f)) * l.r *
             rabs(tr_d(l, wh) * tr_g(l, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cos_i * cos_o * sqrt_denom * sqrt_denom));
    }
    default: return Rgb<R>();  // SpecularReflection / SpecularTransmission / FresnelSpecular ::f
  }
}
template <typename R, uint32_t KM = kAllKinds> RRT_DEV R lobe_pdf(const Lobe<R>& l, V3<R> wo, V3<R> wi) {
  RRT_KIND_SET(KM, l);
  if (l.kind == LOBE_MICROFACET) {  // reflection.rs:1019-1025
    if (!same_hemisphere(wo, wi)) return R(0);
    V3<R> wh = vnormalize(wo + wi);
    return tr_pdf(l, wo, wh) / (R(4) * dot(wo, wh));
  }
  if (l.kind == LOBE_SPEC_REFL || l.kind == LOBE_SPEC_TRANS || l.kind == LOBE_FRESNEL_SPEC) return R(0);
  if (l.kind == LOBE_LAMBERT_TRANS) return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) / R(RRT_PI) : R(0);  // :887-893
  if (l.kind == LOBE_MICROFACET_TRANS) {  // :1124-1144
    if (same_hemisphere(wo, wi)) return R(0);
    const R eta = cos_theta(wo) > R(0) ? l.b / l.a : l.a / l.b;
    const V3<R> wh = vnormalize(wo + wi * eta);
    const R sqrt_denom = dot(wo, wh) + dot(wi, wh) * eta;
    const R dwh_dwi = rabs((eta * eta * dot(wi, wh)) / (sqrt_denom * sqrt_denom));
    return tr_pdf(l, wo, wh) * dwh_dwi;
  }
  return same_hemisphere(wo, wi) ? abs_cos_theta(wi) / R(RRT_PI) : R(0);  // BxDF::pdf default :492-498
}
template <typename R, uint32_t KM = kAllKinds> RRT_DEV Rgb<R> lobe_sample_f(const Lobe<R>& l, V3<R> wo, V3<R>* wi, R u0, R u1, R* pdf, uint32_t* sampled) {
  RRT_KIND_SET(KM, l);
  if (l.kind == LOBE_SPEC_TRANS || l.kind == LOBE_FRESNEL_SPEC) {  // reflection.rs:690-716, :754-795, mode = Radiance
    R fr = R(0);
    if (l.kind == LOBE_FRESNEL_SPEC) {
      fr = fr_dielectric(cos_theta(wo), l.a, l.b);
      if (u0 < fr) {
        *wi = V3<R>(-wo.x, -wo.y, wo.z);
        *sampled = BXDF_SPECULAR | BXDF_REFLECTION;
        *pdf = fr;
        return l.r * fr / abs_cos_theta(*wi);
      }
    }
    const bool entering = cos_theta(wo) > R(0);
    const R eta_i = entering ? l.a : l.b, eta_t = entering ? l.b : l.a;
    if (!refract(wo, faceforward(V3<R>(R(0), R(0), R(1)), wo), eta_i / eta_t, wi)) return Rgb<R>();
    Rgb<R> ft;
    if (l.kind == LOBE_FRESNEL_SPEC) { ft = l.k * (R(1) - fr); *pdf = R(1) - fr; *sampled = BXDF_SPECULAR | BXDF_TRANSMISSION; }
    else { ft = l.r * (Rgb<R>(R(1)) - Rgb<R>(fr_dielectric(cos_theta(*wi), l.a, l.b))); *pdf = R(1); }
    ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
    return ft / abs_cos_theta(*wi);
  }
  if (l.kind == LOBE_LAMBERT_TRANS) {  // :857-869
    *wi = cosine_sample_hemisphere(u0, u1);
    if (wo.z > R(0)) wi->z *= R(-1);
    *pdf = lobe_pdf<R, KM>(l, wo, *wi);
    return lobe_f<R, KM>(l, wo, *wi);
  }
  if (l.kind == LOBE_MICROFACET_TRANS) {  // :1098-1123
    if (wo.z == R(0)) return Rgb<R>();
    const V3<R> wh = tr_sample_wh(l, wo, u0, u1);
    if (dot(wo, wh) < R(0)) return Rgb<R>();
    const R eta = cos_theta(wo) > R(0) ? l.a / l.b : l.b / l.a;
    if (!refract(wo, wh, eta, wi)) return Rgb<R>();
    *pdf = lobe_pdf<R, KM>(l, wo, *wi);
    return lobe_f<R, KM>(l, wo, *wi);
  }
  if (l.kind == LOBE_MICROFACET) {  // reflection.rs:993-1018
    if (wo.z == R(0)) return Rgb<R>();
    V3<R> wh = tr_sample_wh(l, wo, u0, u1);
    if (dot(wo, wh) < R(0)) return Rgb<R>();
    *wi = reflect(wo, wh);
    if (!same_hemisphere(wo, *wi)) return Rgb<R>();
    *pdf = tr_pdf(l, wo, wh) / (R(4) * dot(wo, wh));
    return lobe_f<R, KM>(l, wo, *wi);
  }
  if (l.kind == LOBE_SPEC_REFL) {  // reflection.rs:639-650
    *wi = V3<R>(-wo.x, -wo.y, wo.z);
    *pdf = R(1);
    return fresnel_eval(l, cos_theta(*wi)) * l.r / abs_cos_theta(*wi);
  }
  *wi = cosine_sample_hemisphere(u0, u1);  // BxDF::sample_f default :427-443
  if (wo.z < R(0)) wi->z *= R(-1);
  *pdf = lobe_pdf<R, KM>(l, wo, *wi);
  return lobe_f<R, KM>(l, wo, *wi);
}
template <typename R> RRT_DEV R roughness_to_alpha(R roughness) {  // microfacet.rs:12-20
  roughness = rmax(roughness, R(1e-3));
  R x = log(roughness);
  return R(1.62142) + R(0.819955) * x + R(0.1734) * x * x + R(0.0171201) * x * x * x + R(0.000640711) * x * x * x * x;
}

// Bsdf reflection.rs:205-405. NL = lobe capacity: 2 covers every material except TranslucentMaterial (4).
// Lobes sit in FIXED slots chosen by the material (kind == LOBE_NONE marks an empty one) and every loop over them is
// fully unrolled: with `lobes[n++]` / `lobes[chosen]` the array lived in scratch memory (212 B per lane) and each field
// access was a memory round trip in the most latency-bound kernel of the frame. Relative order is what the reference's
// "count-th matching component" and its sums depend on, and gaps do not change it.
template <typename R, int NL = 2, uint32_t KM = kAllKinds>
struct Bsdf {
  static constexpr uint32_t kinds = KM;
  V3<R> ns, ng, ss, ts;
  Lobe<R> lobes[NL];
  int n;   // number of lobes present
  R eta;   // Bsdf::new(si, eta): 1 except glass / translucent

  RRT_DEV static bool match(const Lobe<R>& l, uint32_t flags) { return l.kind != LOBE_NONE && (l.type & flags) == l.type; }
  RRT_DEV int num_components(uint32_t flags) const {
    int c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) if (match(lobes[i], flags)) c++;
    return c;
  }
  RRT_DEV V3<R> to_local(V3<R> v) const { return {dot(v, ss), dot(v, ts), dot(v, ns)}; }
  RRT_DEV V3<R> to_world(V3<R> v) const {
    return {ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z};
  }
  RRT_DEV Rgb<R> f(V3<R> wo_w, V3<R> wi_w, uint32_t flags) const {  // :252-268
    V3<R> wi = to_local(wi_w), wo = to_local(wo_w);
    if (wo.z == R(0)) return Rgb<R>();
    bool refl = dot(wi_w, ng) * dot(wo_w, ng) > R(0);
    Rgb<R> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const Lobe<R>& l = lobes[i];
      if (match(l, flags) && ((refl && (l.type & BXDF_REFLECTION)) || (!refl && (l.type & BXDF_TRANSMISSION)))) r = r + lobe_f<R, KM>(l, wo, wi);
    }
    return r;
  }
  RRT_DEV R pdf(V3<R> wo_w, V3<R> wi_w, uint32_t flags) const {  // :382-404
    if (n == 0) return R(0);
    V3<R> wo = to_local(wo_w), wi = to_local(wi_w);
    if (wo.z == R(0)) return R(0);
    R p = R(0);
    int matching = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) if (match(lobes[i], flags)) { matching++; p += lobe_pdf<R, KM>(lobes[i], wo, wi); }
    return matching > 0 ? p / (R)matching : R(0);
  }
  // sample_f :302-381 (Q21). *pdf_out / *sampled keep the caller's values on the wo.z == 0 early-out.
  RRT_DEV Rgb<R> sample_f(V3<R> wo_w, V3<R>* wi_w, R u0, R u1, R* pdf_out, uint32_t flags, uint32_t* sampled) const {
    int matching = num_components(flags);
    if (matching == 0) { *pdf_out = R(0); *sampled = BXDF_NONE; return Rgb<R>(); }
    R fl = floor(u0 * (R)matching);
    int comp = (fl != fl || fl <= R(0)) ? 0 : (int)fl;
    if (comp > matching) comp = matching;
    int count = comp, chosen = -1;
    Lobe<R> bx = lobes[0];   // the chosen lobe, copied out with selects (no dynamic index)
#pragma unroll
    for (int i = 0; i < NL; i++)
      if (chosen < 0 && match(lobes[i], flags)) { if (count == 0) { chosen = i; bx = lobes[i]; } else count--; }
    if (chosen < 0) { *pdf_out = R(0); *sampled = BXDF_NONE; return Rgb<R>(); }  // reference: expect() panic (u0 >= 1 only)
    R ur0 = rmin(u0 * (R)matching - (R)comp, Const<R>::one_minus_eps);
    V3<R> wi, wo = to_local(wo_w);
    if (wo.z == R(0)) return Rgb<R>();
    *pdf_out = R(0);
    *sampled = bx.type;
    Rgb<R> f = lobe_sample_f<R, KM>(bx, wo, &wi, ur0, u1, pdf_out, sampled);
    if (*pdf_out == R(0)) { *sampled = BXDF_NONE; return Rgb<R>(); }
    *wi_w = to_world(wi);
    if (!(bx.type & BXDF_REFLECTION) && matching > 1) {
#pragma unroll
      for (int i = 0; i < NL; i++) if (i != chosen && match(lobes[i], flags)) *pdf_out += lobe_pdf<R, KM>(lobes[i], wo, wi);
    }
    if (matching > 1) *pdf_out /= (R)matching;
    return f;
  }
};

// Material::compute_scattering_functions (matte.rs:35-60, plastic.rs:42-73, metal.rs:48-89, mirror.rs:27-47,
// debug_material.rs:37-48, glass.rs:52-112, translucent.rs:52-107) from parameter values (textured parameters are
// evaluated first, resolve_material() in dtexture.hpp)
// Returns false when the material produced a lobe outside the kernel's kind set KM (a host-side mistake in shade_spec()): that lobe is dropped.
template <typename R, int NL, uint32_t KM> RRT_DEV bool build_lobes(const Material<R>& m, Bsdf<R, NL, KM>* b, bool allow_multiple_lobes = true) {
  b->n = 0;
  b->eta = R(1);
#pragma unroll
  for (int i = 0; i < NL; i++) { b->lobes[i].kind = LOBE_NONE; b->lobes[i].type = 0; }
  // material types that can produce nothing but kinds outside KM are not part of this instantiation
  constexpr bool kDiffuse = (KM & (kind_bit(LOBE_LAMBERT) | kind_bit(LOBE_OREN_NAYAR))) != 0u, kMicro = (KM & kind_bit(LOBE_MICROFACET)) != 0u;
  constexpr bool kMirror = (KM & kind_bit(LOBE_SPEC_REFL)) != 0u, kTrans = (KM & (kind_bit(LOBE_SPEC_TRANS) | kind_bit(LOBE_FRESNEL_SPEC) | kind_bit(LOBE_LAMBERT_TRANS) | kind_bit(LOBE_MICROFACET_TRANS))) != 0u;
  constexpr bool kDebug = (KM & (kind_bit(LOBE_DEBUG_DIFFUSE) | kind_bit(LOBE_DEBUG_SPECULAR))) != 0u;
  const uint32_t mt = (uint32_t)m.type;
  if ((mt == 0u && !kDiffuse) || (mt == 1u && !(kDiffuse && kMicro)) || (mt == 2u && !kMicro) || (mt == 3u && !kMirror) || ((mt == 5u || mt == 6u) && !kTrans) ||
      (mt == 4u && !kDebug) || mt > 6u) {
    if (KM != kAllKinds) return false;   // (ALL: the reference's `default` arm below takes every other value as the Debug material)
  }
  switch (m.type) {
    case 0: {  // MatteMaterial
      Rgb<R> r = rgb_clamp0(Rgb<R>(m.kd));
      R sig = clampr(m.sigma, R(0), R(90));
      if (!r.is_black()) {
        Lobe<R>& l = b->lobes[0]; b->n++;
        l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = r; l.fr = FR_NOOP;
        if (sig == R(0)) l.kind = LOBE_LAMBERT;
        else {
          l.kind = LOBE_OREN_NAYAR;
          R sr = (R(RRT_PI) / R(180)) * sig;
          R sigma2 = sr * sr;
          l.a = R(1) - (sigma2 / (R(2) * (sigma2 + R(0.33))));
          l.b = R(0.45) * sigma2 / (sigma2 + R(0.09));
        }
      }
      break;
    }
    case 1: {  // PlasticMaterial (specular lobe gated on kd: Q31)
      Rgb<R> kd = rgb_clamp0(Rgb<R>(m.kd)), ks = rgb_clamp0(Rgb<R>(m.ks));
      if (!kd.is_black()) {
        Lobe<R>& l = b->lobes[0]; b->n++;
        l.kind = LOBE_LAMBERT; l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = kd; l.fr = FR_NOOP;
        R rough = m.roughness;
        if (m.remap_roughness) rough = roughness_to_alpha(rough);
        Lobe<R>& s = b->lobes[1]; b->n++;
        s.kind = LOBE_MICROFACET; s.type = BXDF_GLOSSY | BXDF_REFLECTION; s.r = ks; s.alpha_x = rough; s.alpha_y = rough;
        s.fr = FR_DIELECTRIC; s.eta_i = Rgb<R>(R(1.5)); s.eta_t = Rgb<R>(R(1));
      }
      break;
    }
    case 2: {  // MetalMaterial
      R ur = m.u_roughness, vr = m.v_roughness;
      if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
      Lobe<R>& l = b->lobes[0]; b->n++;
      l.kind = LOBE_MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = Rgb<R>(R(1)); l.alpha_x = ur; l.alpha_y = vr;
      l.fr = FR_CONDUCTOR; l.eta_i = Rgb<R>(R(1)); l.eta_t = Rgb<R>(m.eta); l.k = Rgb<R>(m.k);
      break;
    }
    case 3: {  // MirrorMaterial
      Rgb<R> r = rgb_clamp0(Rgb<R>(m.kr));
      if (!r.is_black()) {
        Lobe<R>& l = b->lobes[0]; b->n++;
        l.kind = LOBE_SPEC_REFL; l.type = BXDF_REFLECTION | BXDF_SPECULAR; l.r = r; l.fr = FR_NOOP;
      }
      break;
    }
    case 5: {  // GlassMaterial glass.rs:52-112 (mode = Radiance; allow_multiple_lobes: Path true, DirectLighting / Debug false)
      const R eta = m.index;
      R ur = m.u_roughness, vr = m.v_roughness;
      const Rgb<R> r = rgb_clamp0(Rgb<R>(m.kr)), t = rgb_clamp0(Rgb<R>(m.kt));
      b->eta = eta;
      const bool is_specular = ur == R(0) && vr == R(0);
      if (is_specular && allow_multiple_lobes) {
        Lobe<R>& l = b->lobes[0]; b->n++;
        l.kind = LOBE_FRESNEL_SPEC; l.type = BXDF_SPECULAR | BXDF_ALL; l.r = r; l.k = t; l.a = R(1); l.b = eta; l.fr = FR_NOOP;
      } else if (is_specular) {
        if (!r.is_black()) {
          Lobe<R>& l = b->lobes[0]; b->n++;
          l.kind = LOBE_SPEC_REFL; l.type = BXDF_REFLECTION | BXDF_SPECULAR; l.r = r; l.fr = FR_DIELECTRIC; l.eta_i = Rgb<R>(R(1)); l.eta_t = Rgb<R>(eta);
        }
        if (!t.is_black()) {
          Lobe<R>& l = b->lobes[1]; b->n++;
          l.kind = LOBE_SPEC_TRANS; l.type = BXDF_SPECULAR | BXDF_TRANSMISSION; l.r = t; l.a = R(1); l.b = eta; l.fr = FR_NOOP;
        }
      } else {
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        if (!r.is_black()) {
          Lobe<R>& l = b->lobes[0]; b->n++;
          l.kind = LOBE_MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = r; l.alpha_x = ur; l.alpha_y = vr;
          l.fr = FR_DIELECTRIC; l.eta_i = Rgb<R>(R(1)); l.eta_t = Rgb<R>(eta);
        }
        if (!t.is_black()) {
          Lobe<R>& l = b->lobes[1]; b->n++;
          l.kind = LOBE_MICROFACET_TRANS; l.type = BXDF_GLOSSY | BXDF_TRANSMISSION; l.r = t; l.alpha_x = ur; l.alpha_y = vr; l.a = R(1); l.b = eta; l.fr = FR_NOOP;
        }
      }
      break;
    }
    case 6: if constexpr (NL >= 4) {  // TranslucentMaterial translucent.rs:50-107 (needs NL = 4)
      const R eta = R(1.5);
      b->eta = eta;
      const Rgb<R> r = rgb_clamp0(Rgb<R>(m.reflect)), t = rgb_clamp0(Rgb<R>(m.transmit));
      const Rgb<R> kd = rgb_clamp0(Rgb<R>(m.kd));
      if (!kd.is_black()) {
        if (!r.is_black()) { Lobe<R>& l = b->lobes[0]; b->n++; l.kind = LOBE_LAMBERT; l.type = BXDF_DIFFUSE | BXDF_REFLECTION; l.r = r * kd; l.fr = FR_NOOP; }
        if (!t.is_black()) { Lobe<R>& l = b->lobes[1]; b->n++; l.kind = LOBE_LAMBERT_TRANS; l.type = BXDF_DIFFUSE | BXDF_TRANSMISSION; l.r = t * kd; l.fr = FR_NOOP; }
      }
      const Rgb<R> ks = rgb_clamp0(Rgb<R>(m.ks));
      if (!ks.is_black() && (!r.is_black() || !t.is_black())) {
        R rough = m.roughness;
        if (m.remap_roughness) rough = roughness_to_alpha(rough);
        if (!r.is_black()) {
          Lobe<R>& l = b->lobes[2]; b->n++;
          l.kind = LOBE_MICROFACET; l.type = BXDF_GLOSSY | BXDF_REFLECTION; l.r = r * ks; l.alpha_x = rough; l.alpha_y = rough;
          l.fr = FR_DIELECTRIC; l.eta_i = Rgb<R>(R(1)); l.eta_t = Rgb<R>(eta);
        }
        if (!t.is_black()) {
          Lobe<R>& l = b->lobes[3]; b->n++;
          l.kind = LOBE_MICROFACET_TRANS; l.type = BXDF_GLOSSY | BXDF_TRANSMISSION; l.r = t * ks; l.alpha_x = rough; l.alpha_y = rough; l.a = R(1); l.b = eta; l.fr = FR_NOOP;
        }
      }
      break;
    } else break;
    default: {  // DebugMaterial
      Lobe<R>& a = b->lobes[0]; b->n++;
      a.kind = LOBE_DEBUG_DIFFUSE; a.type = BXDF_DIFFUSE | BXDF_REFLECTION; a.fr = FR_NOOP;
      Lobe<R>& c = b->lobes[1]; b->n++;
      c.kind = LOBE_DEBUG_SPECULAR; c.type = BXDF_SPECULAR | BXDF_REFLECTION; c.fr = FR_NOOP;
      break;
    }
  }
  bool ok = true;
  if (KM != kAllKinds) {
#pragma unroll
    for (int i = 0; i < NL; i++) if (!kind_in<KM>(b->lobes[i].kind)) { b->lobes[i].kind = LOBE_NONE; b->lobes[i].type = 0; b->n--; ok = false; }
  }
  return ok;
}

}  // namespace rrtd
