// Host side of the Halton sampler: tables built once and uploaded (the reference rebuilds them per tile,
// integrator/mod.rs:73). Follows samplers/halton.rs:23-61,131-150 and lowdiscrepancy.rs:3-270.
// The reference shuffles digit permutations with rand::thread_rng (not reproducible);
// here the shuffle is driven by PCG32(seed) and the table is shared by oracle and GPU.
#include "scene.hpp"

namespace rrt {

namespace {
struct PrimeTables {
  uint16_t primes[1024];
  uint32_t sums[1000];
  PrimeTables() {
    int n = 0;
    for (int c = 2; n < 1024; c++) {
      bool is_p = true;
      for (int d = 2; d * d <= c; d++)
        if (c % d == 0) { is_p = false; break; }
      if (is_p) primes[n++] = (uint16_t)c;
    }
    uint32_t acc = 0;
    for (int i = 0; i < 1000; i++) { sums[i] = acc; acc += primes[i]; }
  }
};
const PrimeTables& tables() { static PrimeTables t; return t; }

struct Pcg32 {  // O'Neill's PCG-XSH-RR 64/32
  uint64_t state = 0, inc = 0;
  explicit Pcg32(uint64_t seed, uint64_t seq = 1) {
    inc = (seq << 1u) | 1u;
    next();
    state += seed;
    next();
  }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  uint32_t bounded(uint32_t bound) {
    uint32_t threshold = (uint32_t)(-bound) % bound;
    for (;;) {
      uint32_t r = next();
      if (r >= threshold) return r % bound;
    }
  }
};

// extended_gcd / multiplicative_inverse, halton.rs:131-150 (base case y = 1 is the reference's, Q24)
void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) {
  if (b == 0) { *x = 1; *y = 1; return; }
  uint64_t d = a / b;
  int64_t xp = 0, yp = 0;
  extended_gcd(b, a % b, &xp, &yp);
  *x = yp;
  *y = (int64_t)((uint64_t)xp - (uint64_t)((int64_t)d * yp));
}
uint64_t mod_u64(uint64_t a, uint64_t b) {  // misc.rs:334-349 on u64 (never negative)
  return a - (a / b) * b;
}
uint64_t multiplicative_inverse(uint64_t a, uint64_t n) {
  int64_t x = 0, y = 0;
  extended_gcd(a, n, &x, &y);
  return mod_u64((uint64_t)x, n);  // `x as u64` wraps for negative x, as in the reference
}
}  // namespace

const uint16_t* prime_table() { return tables().primes; }
const uint32_t* prime_sums_table() { return tables().sums; }

double radical_inverse_host(int base_index, uint64_t a) {
  if (base_index == 0) {
    // reverse_bits_64(a) as f64 * 2^-64, lowdiscrepancy.rs:169-186,230-233
    uint64_t v = a, r = 0;
    for (int i = 0; i < 64; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return (double)r * 0.00000000000000000005421010862427522;
  }
  const uint64_t base = prime_table()[base_index];
  const double inv_base = 1.0 / (double)base;
  uint64_t reversed = 0;
  double inv_base_n = 1.0;
  while (a != 0) {
    uint64_t next = a / base, digit = a - next * base;
    reversed = reversed * base + digit;
    inv_base_n *= inv_base;
    a = next;
  }
  double v = (double)reversed * inv_base_n;
  const double one_minus_eps = 1.0 - std::numeric_limits<double>::epsilon() * 0.5;
  return v < one_minus_eps ? v : one_minus_eps;
}

void init_halton(SceneData& s, uint64_t nsamp, bool sample_at_center, const int32_t sb[4], uint64_t seed) {
  rrt_sampler& h = s.desc.sampler;
  h.type = RRT_SAMPLER_HALTON;
  h.samples_per_pixel = nsamp;
  h.sample_at_center = sample_at_center ? 1 : 0;
  h.perm_seed = seed;
  // compute_radical_inverse_permutations lowdiscrepancy.rs:250-270 + sampling::shuffle sampling.rs:181-193
  size_t total = 0;
  for (int i = 0; i < kPrimeTableSize; i++) total += prime_table()[i];
  s.perms.assign(total, 0);
  Pcg32 rng(seed);
  size_t p = 0;
  for (int i = 0; i < kPrimeTableSize; i++) {
    uint32_t count = prime_table()[i];
    for (uint32_t j = 0; j < count; j++) s.perms[p + j] = (uint16_t)j;
    for (uint32_t k = 0; k < count; k++) {
      uint32_t other = k + rng.bounded(count - k);
      std::swap(s.perms[p + k], s.perms[p + other]);
    }
    p += count;
  }
  // base scales / exponents, halton.rs:27-41 (K_MAX_RESOLUTION = 128)
  const int64_t kMaxRes = 128;
  int64_t res[2] = {(int64_t)sb[2] - sb[0], (int64_t)sb[3] - sb[1]};
  for (int i = 0; i < 2; i++) {
    int64_t base = (i == 0) ? 2 : 3, scale = 1, exp = 0;
    while (scale < std::min(res[i], kMaxRes)) { scale *= base; exp += 1; }
    h.base_scales[i] = scale;
    h.base_exponents[i] = exp;
  }
  h.sample_stride = (uint64_t)(h.base_scales[0] * h.base_scales[1]);
  h.mult_inverse[0] = multiplicative_inverse((uint64_t)h.base_scales[1], (uint64_t)h.base_scales[0]);
  h.mult_inverse[1] = multiplicative_inverse((uint64_t)h.base_scales[0], (uint64_t)h.base_scales[1]);
}

}  // namespace rrt
