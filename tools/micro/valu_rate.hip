// Micro-benchmark: issue cost of the VALU / SALU instructions the traversal and camera kernels are made of, gfx950, 1 / 2 / 8 waves per SIMD.
// Each test is 8 independent chains of one instruction, unrolled 8x; result = cycles per wave-instruction per SIMD at the nominal 2.4 GHz.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// I(n) = instruction text for chain n (operands: %n = chain register, %8 = constant register, %9 = second constant)
#define REP8(I) I(0) "\n" I(1) "\n" I(2) "\n" I(3) "\n" I(4) "\n" I(5) "\n" I(6) "\n" I(7)
#define SCALAR_TEST(NAME, I)                                                                                                         \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                                              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;              \
    const float c = 1.0001f, c2 = 0.5f;                                                                                             \
    asm volatile("s_mov_b64 s[20:21], 0x5555\ns_mov_b64 s[22:23], 0x3333\ns_mov_b64 vcc, 0x0f0f" ::: "s20", "s21", "s22", "s23", "vcc"); \
    for (int i = 0; i < iters; i++) {                                                                                               \
      _Pragma("unroll") for (int u = 0; u < 8; u++)                                                                                 \
        asm volatile(REP8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2) : "s20", "s21", "s22", "s23", "s24", "s25", "vcc", "scc"); \
    }                                                                                                                               \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                   \
  }
#define PACKED_TEST(NAME, I)                                                                                                         \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                                              \
    const float t = threadIdx.x;                                                                                                    \
    v2f a0 = {t, t + 1}, a1 = {t + 2, t + 3}, a2 = {t + 4, t + 5}, a3 = {t + 6, t + 7}, a4 = {t + 1, t}, a5 = {t + 3, t + 2}, a6 = {t + 5, t + 4}, a7 = {t + 7, t + 6}; \
    const v2f c = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};                                                                           \
    for (int i = 0; i < iters; i++) {                                                                                               \
      _Pragma("unroll") for (int u = 0; u < 8; u++)                                                                                 \
        asm volatile(REP8(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2));  \
    }                                                                                                                               \
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;                                   \
  }
#define I_MUL(n) "v_mul_f32 %" #n ", %8, %" #n
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %8"
#define I_FMA(n) "v_fma_f32 %" #n ", %8, %" #n ", %9"
#define I_MIN(n) "v_min_f32 %" #n ", %8, %" #n
#define I_MAX3(n) "v_max3_f32 %" #n ", %8, %" #n ", %9"
#define I_MOV(n) "v_mov_b32 %" #n ", %8"
#define I_ADDU(n) "v_add_u32 %" #n ", %8, %" #n
#define I_AND(n) "v_and_b32 %" #n ", %8, %" #n
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 5"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %8"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %8"
#define I_CND_VCC(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc"
#define I_CND_SGPR(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", s[20:21]"
#define I_CMP_VCC(n) "v_cmp_lt_f32 vcc, %8, %" #n
#define I_CMP_SGPR(n) "v_cmp_lt_f32 s[24:25], %8, %" #n
#define I_CMPU_SGPR(n) "v_cmp_lt_u32 s[24:25], %8, %" #n
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define I_SAND(n) "s_and_b64 s[20:21], s[20:21], s[22:23]"
#define I_SANDSAVE(n) "s_and_saveexec_b64 s[24:25], exec"
#define I_SNOP(n) "s_nop 0"
#define I_MIX_VS(n) "v_mul_f32 %" #n ", %8, %" #n "\ns_and_b64 s[20:21], s[20:21], s[22:23]"
#define I_MIX_CMPCND(n) "v_cmp_lt_f32 s[24:25], %8, %" #n "\nv_cndmask_b32_e64 %" #n ", %8, %" #n ", s[20:21]"
#define I_CMPCND_VCC(n) "v_cmp_lt_f32 vcc, %8, %" #n "\nv_cndmask_b32 %" #n ", %8, %" #n ", vcc"
#define I_MINI(n) "v_min_i32 %" #n ", %8, %" #n
#define I_MAXU(n) "v_max_u32 %" #n ", %8, %" #n
#define I_MED3(n) "v_med3_f32 %" #n ", %8, %" #n ", %9"
#define I_MINIMUM3(n) "v_minimum3_f32 %" #n ", %8, %" #n ", %9"
#define I_MAXE64(n) "v_max_f32_e64 %" #n ", %8, %" #n
#define I_OR(n) "v_or_b32 %" #n ", %8, %" #n
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n
#define I_MULE64(n) "v_mul_f32_e64 %" #n ", %8, %" #n
#define I_ADD3(n) "v_add3_u32 %" #n ", %8, %" #n ", %9"
#define I_MULLIT(n) "v_mul_f32 %" #n ", 0x3f800003, %" #n
#define I_MULNEG(n) "v_mul_f32_e64 %" #n ", -%8, %" #n
#define I_CNDVCC_E64(n) "v_cndmask_b32_e64 %" #n ", %8, %" #n ", vcc"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %8, %" #n
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %8, %" #n
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %8, %" #n ", %9"
SCALAR_TEST(k_mul, I_MUL) SCALAR_TEST(k_sub, I_SUB) SCALAR_TEST(k_fma, I_FMA) SCALAR_TEST(k_min, I_MIN) SCALAR_TEST(k_max3, I_MAX3) SCALAR_TEST(k_mov, I_MOV)
SCALAR_TEST(k_addu, I_ADDU) SCALAR_TEST(k_and, I_AND) SCALAR_TEST(k_bfe, I_BFE) SCALAR_TEST(k_lshladd, I_LSHLADD) SCALAR_TEST(k_lshlor, I_LSHLOR)
SCALAR_TEST(k_cnd_vcc, I_CND_VCC) SCALAR_TEST(k_cnd_sgpr, I_CND_SGPR) SCALAR_TEST(k_cmp_vcc, I_CMP_VCC) SCALAR_TEST(k_cmp_sgpr, I_CMP_SGPR) SCALAR_TEST(k_cmpu_sgpr, I_CMPU_SGPR)
SCALAR_TEST(k_rcp, I_RCP) SCALAR_TEST(k_movdpp, I_MOVDPP) SCALAR_TEST(k_sand, I_SAND) SCALAR_TEST(k_snop, I_SNOP) SCALAR_TEST(k_mix_vs, I_MIX_VS) SCALAR_TEST(k_mix_cmpcnd, I_MIX_CMPCND)
SCALAR_TEST(k_cmpcnd_vcc, I_CMPCND_VCC) SCALAR_TEST(k_mini, I_MINI) SCALAR_TEST(k_maxu, I_MAXU) SCALAR_TEST(k_med3, I_MED3) SCALAR_TEST(k_minimum3, I_MINIMUM3) SCALAR_TEST(k_maxe64, I_MAXE64)
SCALAR_TEST(k_or, I_OR) SCALAR_TEST(k_lshl, I_LSHL) SCALAR_TEST(k_mule64, I_MULE64) SCALAR_TEST(k_add3, I_ADD3) SCALAR_TEST(k_mullit, I_MULLIT) SCALAR_TEST(k_mulneg, I_MULNEG) SCALAR_TEST(k_cndvcc_e64, I_CNDVCC_E64)
PACKED_TEST(k_pkmul, I_PKMUL) PACKED_TEST(k_pkadd, I_PKADD) PACKED_TEST(k_pkfma, I_PKFMA)

typedef void (*kern_t)(float*, int);
double run(kern_t f, float* out, int blocks, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms;
}
int main() {
  int cus; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  const int iters = 2000;
  struct T { const char* name; kern_t f; int per; } tests[] = {
    {"v_mul_f32", k_mul, 1}, {"v_sub_f32", k_sub, 1}, {"v_fma_f32", k_fma, 1}, {"v_min_f32", k_min, 1}, {"v_max3_f32", k_max3, 1}, {"v_mov_b32", k_mov, 1}, {"v_add_u32", k_addu, 1},
    {"v_and_b32", k_and, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_lshl_or_b32", k_lshlor, 1}, {"v_cndmask vcc", k_cnd_vcc, 1}, {"v_cndmask_e64 sgpr", k_cnd_sgpr, 1},
    {"v_cmp_f32 -> vcc", k_cmp_vcc, 1}, {"v_cmp_f32 -> sgpr", k_cmp_sgpr, 1}, {"v_cmp_u32 -> sgpr", k_cmpu_sgpr, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_mov_dpp", k_movdpp, 1}, {"s_and_b64", k_sand, 1},
    {"s_nop 0", k_snop, 1}, {"v_mul + s_and (pair)", k_mix_vs, 1}, {"v_cmp + v_cndmask_e64 (pair)", k_mix_cmpcnd, 1}, {"v_cmp vcc + v_cndmask vcc (pair)", k_cmpcnd_vcc, 1}, {"v_min_i32", k_mini, 1}, {"v_max_u32", k_maxu, 1}, {"v_med3_f32", k_med3, 1}, {"v_minimum3_f32", k_minimum3, 1},
    {"v_max_f32_e64", k_maxe64, 1}, {"v_or_b32", k_or, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_mul_f32_e64", k_mule64, 1}, {"v_add3_u32", k_add3, 1}, {"v_mul_f32 literal", k_mullit, 1}, {"v_mul_f32_e64 neg", k_mulneg, 1},
    {"v_cndmask_e64 vcc", k_cndvcc_e64, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_pk_fma_f32", k_pkfma, 1}};
  printf("%-30s %8s %8s %8s   (cycles per wave-instruction per SIMD at 1 / 2 / 8 waves per SIMD, 2.4 GHz nominal)\n", "instruction", "1", "2", "8");
  for (const T& t : tests) {
    printf("%-30s", t.name);
    for (int wps : {1, 2, 8}) printf(" %8.2f", run(t.f, out, cus * wps, iters) * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    printf("\n");
  }
  return 0;
}
