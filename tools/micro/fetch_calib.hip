// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the traversal kernels (MI355X_MICROARCH.md, HBM section: "Other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute").
// Each kernel below touches a KNOWN number of bytes exactly once (tables far larger than L2 + Infinity Cache, a bijective index), or a
// known cache-resident table many times; tools/fetch_calib.sh runs the program under `rocprofv3 --pmc FETCH_SIZE` (and the TCC request
// counters in a second pass) and tools/fetch_calib.py prints reported / known per kernel -> profiles/r3_fetch_calibration.txt.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o build/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t kLogN = 25;             // 2^25 nodes of 64 B = 2 GiB
constexpr uint32_t kN = 1u << kLogN;

// wide coalesced streaming read: 16 B per lane, every byte of the 2 GiB buffer once (the guide's calibrated case: FETCH_SIZE = 1/2)
__global__ void __launch_bounds__(256) calib_stream_16B_per_lane(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
// the pair-node fetch: every lane reads the four 16-byte words of ITS OWN 64-byte node; node = bijection of the lane's global index, so all
// 2^25 nodes (2 GiB) are read exactly once, in an order that shares no line between lanes and no locality between waves
__global__ void __launch_bounds__(256) calib_gather_64B_node_once(const char* __restrict__ base, uint32_t* out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < kN; i += gridDim.x * 256u) {
    const uint32_t node = (i * 2654435761u + 12345u) & (kN - 1u);   // odd multiplier: a bijection on 25 bits
    const uint4* q = reinterpret_cast<const uint4*>(base + (size_t)node * 64u);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// one 16-byte word out of every 64-byte line, each line once: 0.5 GiB requested, 2 GiB of lines touched
__global__ void __launch_bounds__(256) calib_gather_16B_of_64B_line_once(const char* __restrict__ base, uint32_t* out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < kN; i += gridDim.x * 256u) {
    const uint32_t node = (i * 2654435761u + 12345u) & (kN - 1u);
    const uint4 a = *reinterpret_cast<const uint4*>(base + (size_t)node * 64u + 16u * (node & 3u));
    acc ^= a.x ^ a.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// the triangle fetch: three 16-byte words of a 48-byte record (records are 48 B apart: a record straddles a 64-byte line 2 times in 4, so the
// 2^25 records = 1.5 GiB touch 1.5 lines each; neighbouring records share lines but are read far apart in time)
__global__ void __launch_bounds__(256) calib_gather_48B_record_once(const char* __restrict__ base, uint32_t* out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < kN; i += gridDim.x * 256u) {
    const uint32_t rec = (i * 2654435761u + 12345u) & (kN - 1u);
    const uint4* q = reinterpret_cast<const uint4*>(base + (size_t)rec * 48u);
    const uint4 a = q[0], b = q[1], c = q[2];
    acc ^= a.x ^ b.y ^ c.z;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// the same 64-byte node fetch out of a table that stays in cache: `log_nodes` = 14 -> 1 MiB (fits every XCD's 4 MiB L2), 17 -> 8 MiB (the size
// of config 4's pair nodes: L2 misses served by the Infinity Cache), 2^25 node reads (2 GiB requested) in both
template <int TAG>
__global__ void __launch_bounds__(256) calib_gather_64B_node_resident(const char* __restrict__ base, uint32_t mask, uint32_t* out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < kN; i += gridDim.x * 256u) {
    uint32_t h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const uint4* q = reinterpret_cast<const uint4*>(base + (size_t)(h & mask) * 64u);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)kN * 64u;
  char* d; uint32_t* out;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(d, 1, bytes));
  int cus; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int blocks = cus * 8;
  auto timed = [&](const char* name, double known, auto launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s known_bytes %14.0f  %8.3f ms  %8.1f GB/s\n", name, known, ms, known / ms / 1e6);
  };
  timed("calib_stream_16B_per_lane", (double)bytes, [&] { hipLaunchKernelGGL(calib_stream_16B_per_lane, dim3(blocks), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out); });
  timed("calib_gather_64B_node_once", (double)bytes, [&] { hipLaunchKernelGGL(calib_gather_64B_node_once, dim3(blocks), dim3(256), 0, 0, d, out); });
  timed("calib_gather_16B_of_64B_line_once", (double)kN * 16.0, [&] { hipLaunchKernelGGL(calib_gather_16B_of_64B_line_once, dim3(blocks), dim3(256), 0, 0, d, out); });
  timed("calib_gather_48B_record_once", (double)kN * 48.0, [&] { hipLaunchKernelGGL(calib_gather_48B_record_once, dim3(blocks), dim3(256), 0, 0, d, out); });
  // resident tables: warm them, then measure
  hipLaunchKernelGGL(calib_gather_64B_node_resident<0>, dim3(blocks), dim3(256), 0, 0, d, (1u << 14) - 1u, out);
  timed("calib_gather_64B_node_resident<1>", (double)bytes, [&] { hipLaunchKernelGGL(calib_gather_64B_node_resident<1>, dim3(blocks), dim3(256), 0, 0, d, (1u << 14) - 1u, out); });
  hipLaunchKernelGGL(calib_gather_64B_node_resident<0>, dim3(blocks), dim3(256), 0, 0, d, (1u << 17) - 1u, out);
  timed("calib_gather_64B_node_resident<2>", (double)bytes, [&] { hipLaunchKernelGGL(calib_gather_64B_node_resident<2>, dim3(blocks), dim3(256), 0, 0, d, (1u << 17) - 1u, out); });
  CK(hipDeviceSynchronize());
  return 0;
}
