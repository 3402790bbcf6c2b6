// Micro-benchmark: how fast a CU's vector L1 (TCP) serves per-lane fetches of 64-byte BVH pair nodes, by access pattern.
//   mode 0: every lane reads its own node with four 16-byte loads (what the traversal kernels did in round 1)
//   mode 1: the four lanes of a quad read one node per instruction, 16 bytes each (64 contiguous bytes per quad and instruction)
//   mode 2: the two lanes of a pair read half a node per instruction pair
// Each lane follows a dependent pseudo-random chain through a table of n nodes (like a tree walk); little arithmetic.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/tcp_gather.hip -o build/tcp_gather ; run: build/tcp_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ nodes, uint32_t mask, int steps, uint32_t* out) {
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
  uint32_t acc = 0;
  const uint32_t lane4 = threadIdx.x & 3u, lane2 = threadIdx.x & 1u;
  for (int s = 0; s < steps; s++) {
    const char* base = reinterpret_cast<const char*>(nodes);
    uint4 a, b, c, d;
    if (MODE == 0) {
      const char* np = base + (size_t)idx * 64u;
      a = *reinterpret_cast<const uint4*>(np); b = *reinterpret_cast<const uint4*>(np + 16);
      c = *reinterpret_cast<const uint4*>(np + 32); d = *reinterpret_cast<const uint4*>(np + 48);
    } else if (MODE == 1) {
      const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true), i1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true);
      const uint32_t i2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true), i3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);
      a = *reinterpret_cast<const uint4*>(base + (size_t)i0 * 64u + lane4 * 16u);
      b = *reinterpret_cast<const uint4*>(base + (size_t)i1 * 64u + lane4 * 16u);
      c = *reinterpret_cast<const uint4*>(base + (size_t)i2 * 64u + lane4 * 16u);
      d = *reinterpret_cast<const uint4*>(base + (size_t)i3 * 64u + lane4 * 16u);
    } else if (MODE == 3) {   // a wave reads 1 KiB of contiguous bytes per instruction (wave-uniform pseudo-random block, fully coalesced)
      const uint32_t blk = __builtin_amdgcn_readfirstlane(idx) & ~63u;
      const char* np = base + ((size_t)((blk + (threadIdx.x & 63u)) & mask)) * 64u;
      const uint32_t l = threadIdx.x & 63u;
      const char* q = base + (size_t)(blk & mask) * 64u;
      a = *reinterpret_cast<const uint4*>(q + l * 16u); b = *reinterpret_cast<const uint4*>(q + 1024u + l * 16u);
      c = *reinterpret_cast<const uint4*>(q + 2048u + l * 16u); d = *reinterpret_cast<const uint4*>(q + 3072u + l * 16u);
      (void)np;
    } else if (MODE == 4) {   // own node, sixteen 4-byte loads
      const uint32_t* np = reinterpret_cast<const uint32_t*>(base + (size_t)idx * 64u);
      a = make_uint4(np[0], np[1], np[2], np[3]); b = make_uint4(np[4], np[5], np[6], np[7]);
      c = make_uint4(np[8], np[9], np[10], np[11]); d = make_uint4(np[12], np[13], np[14], np[15]);
      asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    } else if (MODE == 5) {   // own node, one 16-byte load only (a quarter of the node)
      const char* np = base + (size_t)idx * 64u;
      a = *reinterpret_cast<const uint4*>(np); b = a; c = a; d = a;
    } else if (MODE == 6) {   // own node, eight 8-byte loads
      const uint2* np = reinterpret_cast<const uint2*>(base + (size_t)idx * 64u);
      uint2 v0 = np[0], v1 = np[1], v2 = np[2], v3 = np[3], v4 = np[4], v5 = np[5], v6 = np[6], v7 = np[7];
      asm volatile("" : "+v"(v0.x), "+v"(v1.x), "+v"(v2.x), "+v"(v3.x), "+v"(v4.x), "+v"(v5.x), "+v"(v6.x), "+v"(v7.x));
      a = make_uint4(v0.x, v0.y, v1.x, v1.y); b = make_uint4(v2.x, v2.y, v3.x, v3.y); c = make_uint4(v4.x, v4.y, v5.x, v5.y); d = make_uint4(v6.x, v6.y, v7.x, v7.y);
    } else {
      const uint32_t i0 = __builtin_amdgcn_mov_dpp(idx, 0xa0, 0xf, 0xf, true), i1 = __builtin_amdgcn_mov_dpp(idx, 0xf5, 0xf, 0xf, true);   // quad_perm [0,0,2,2], [1,1,3,3]
      a = *reinterpret_cast<const uint4*>(base + (size_t)i0 * 64u + lane2 * 32u);
      b = *reinterpret_cast<const uint4*>(base + (size_t)i0 * 64u + lane2 * 32u + 16u);
      c = *reinterpret_cast<const uint4*>(base + (size_t)i1 * 64u + lane2 * 32u);
      d = *reinterpret_cast<const uint4*>(base + (size_t)i1 * 64u + lane2 * 32u + 16u);
    }
    const uint32_t h = a.x ^ b.y ^ c.z ^ d.w;
    acc += h;
    idx = (idx * 1664525u + 1013904223u + (h & 1u) * 977u) & mask;   // dependent on the loaded data
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE> float run(const uint4* nodes, uint32_t n, int steps, int blocks, uint32_t* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, nodes, n - 1, steps, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, nodes, n - 1, steps, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  int cus; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t* out; CK(hipMalloc(&out, 64));
  const int steps = 2000, blocks = cus * 8;   // 8 waves per SIMD
  for (uint32_t n : {256u, 4096u, 65536u, 131072u, 1u << 22}) {
    std::vector<uint4> h((size_t)n * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_uint4((uint32_t)(i * 2654435761u), (uint32_t)(i * 40503u), (uint32_t)(i * 69069u), (uint32_t)i);
    uint4* d; CK(hipMalloc(&d, h.size() * 16)); CK(hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    const float t0 = run<0>(d, n, steps, blocks, out), t1 = run<1>(d, n, steps, blocks, out), t2 = run<2>(d, n, steps, blocks, out);
    const float t3 = run<3>(d, n, steps, blocks, out), t4 = run<4>(d, n, steps, blocks, out), t5 = run<5>(d, n, steps, blocks, out), t6 = run<6>(d, n, steps, blocks, out);
    printf("   coalesced 4 x 1 KiB %7.3f ms | 16 x dword %7.3f ms | 1 x dwordx4 %7.3f ms | 8 x dwordx2 %7.3f ms\n", t3, t4, t5, t6);
    const double lane_steps = (double)blocks * 256 * steps;
    printf("table %8.1f KB: own-node %7.3f ms (%.2f G node fetches/s, %.2f cycles per CU per wave-step @2.4GHz) | quad %7.3f ms (%.2fx) | pair %7.3f ms (%.2fx)\n", n * 64 / 1024.0, t0,
           lane_steps / t0 / 1e6, t0 * 1e-3 * 2.4e9 / ((double)blocks * 4 * steps / cus), t1, t0 / t1, t2, t0 / t2);
    CK(hipFree(d));
  }
  return 0;
}
