// Micro-benchmark 2: cost of divergent vector loads that hit the CU's L1 (16 KB table), by width and by how the loads of one step relate.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/tcp_gather2.hip -o build/tcp_gather2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE: 0 = 4 x b128 same node; 1 = 4 x b128 of four different nodes (chunk 0); 2 = 4 x b128 of four different nodes, chunks 0..3;
//       3 = 2 x b128 same node; 4 = 3 x b128 same node; 5 = 8 x b64 same node; 6 = 16 x b32 same node; 7 = 4 x b96 (48 B) same node;
//       8 = 1 x b128; 9 = 2 x b128 of two different nodes; 11 / 12 / 13 / 14 / 15 = mode 0 with every other lane / every fourth lane / the first 32 lanes / three lanes of every quad / two adjacent lanes of every quad active
template <int MODE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ base, uint32_t mask, int steps, uint32_t* out) {
  uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    uint32_t h = 0;
    const uint32_t j1 = (idx * 7u + 3u) & mask, j2 = (idx * 13u + 5u) & mask, j3 = (idx * 29u + 11u) & mask;
    auto L4 = [&](uint32_t node, uint32_t off) { const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)node * 64u + off); return v.x ^ v.y ^ v.z ^ v.w; };
    if (MODE == 0) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48);
    else if (MODE == 1) h = L4(idx, 0) ^ L4(j1, 0) ^ L4(j2, 0) ^ L4(j3, 0);
    else if (MODE == 2) h = L4(idx, 0) ^ L4(j1, 16) ^ L4(j2, 32) ^ L4(j3, 48);
    else if (MODE == 3) h = L4(idx, 0) ^ L4(idx, 16);
    else if (MODE == 4) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32);
    else if (MODE == 5) { const uint2* p = reinterpret_cast<const uint2*>(base + (size_t)idx * 64u);
      uint2 v[8]; for (int i = 0; i < 8; i++) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[i]) : "v"(p + i)); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int i = 0; i < 8; i++) h ^= v[i].x ^ v[i].y; }
    else if (MODE == 6) { const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (size_t)idx * 64u);
      uint32_t v[16]; for (int i = 0; i < 16; i++) { asm volatile("global_load_dword %0, %1, off" : "=v"(v[i]) : "v"(p + i)); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); for (int i = 0; i < 16; i++) h ^= v[i]; }
    else if (MODE == 8) h = L4(idx, 0);
    else if (MODE == 11) { if (threadIdx.x & 1u) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48); }        // 32 of 64 lanes active (every other lane)
    else if (MODE == 12) { if ((threadIdx.x & 3u) == 0u) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48); } // 16 of 64 lanes active
    else if (MODE == 14) { if ((threadIdx.x & 3u) != 3u) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48); }  // 48 of 64: three lanes of every quad
    else if (MODE == 15) { if ((threadIdx.x & 2u) == 0u) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48); }  // 32 of 64: two ADJACENT lanes of every quad
    else if (MODE == 13) { if ((threadIdx.x & 63u) < 32u) h = L4(idx, 0) ^ L4(idx, 16) ^ L4(idx, 32) ^ L4(idx, 48); } // the first 32 lanes
    else if (MODE == 9) h = L4(idx, 0) ^ L4(j1, 16);
    acc += h;
    idx = (idx * 1664525u + 1013904223u + (h & 1u) * 977u) & mask;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE> float run(const char* nodes, uint32_t n, int steps, int blocks, uint32_t* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, nodes, n - 1, steps, out);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, nodes, n - 1, steps, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
int main() {
  int cus; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t* out; CK(hipMalloc(&out, 64));
  const int steps = 2000;
  for (int wps : {8, 4, 2}) for (uint32_t n : {256u, 65536u}) {
    const int blocks = cus * wps;
    std::vector<uint32_t> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    char* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double sc = 1e-3 * 2.4e9 / ((double)wps * 4 * steps);   // ms -> cycles per CU per wave-step
    printf("table %6.0f KB, %d waves/SIMD | cycles per CU per wave-step: 4xb128 same node %5.0f | 4 nodes chunk0 %5.0f | 4 nodes chunks 0-3 %5.0f | 2xb128 %5.0f | 3xb128 %5.0f | 8xb64 %5.0f | 16xb32 %5.0f | 1xb128 %5.0f | 2 nodes %5.0f\n",
           n * 64 / 1024.0, wps, run<0>(d, n, steps, blocks, out) * sc, run<1>(d, n, steps, blocks, out) * sc, run<2>(d, n, steps, blocks, out) * sc, run<3>(d, n, steps, blocks, out) * sc,
           run<4>(d, n, steps, blocks, out) * sc, run<5>(d, n, steps, blocks, out) * sc, run<6>(d, n, steps, blocks, out) * sc, run<8>(d, n, steps, blocks, out) * sc, run<9>(d, n, steps, blocks, out) * sc);
    printf("   4xb128 same node with 32 of 64 lanes active (alternate) %5.0f | 16 of 64 %5.0f | the first 32 lanes %5.0f | 3 lanes of every quad %5.0f | 2 adjacent lanes of every quad %5.0f\n", run<11>(d, n, steps, blocks, out) * sc,
           run<12>(d, n, steps, blocks, out) * sc, run<13>(d, n, steps, blocks, out) * sc, run<14>(d, n, steps, blocks, out) * sc, run<15>(d, n, steps, blocks, out) * sc);
    CK(hipFree(d));
  }
  return 0;
}
