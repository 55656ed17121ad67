// LDS operation rates on gfx950, random dword addresses inside a table of 2^LOG dwords: plain read, plain write,
// atomic add without return, atomic or with return, and the same with every lane on ONE address.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe tools/probe/lds_atomic_probe.hip ; run: ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE, int LOG>
__global__ __launch_bounds__(256) void k_probe(uint32_t* sink, int iters) {
  __shared__ uint32_t s[1 << LOG];
  for (int i = threadIdx.x; i < (1 << LOG); i += 256) s[i] = 0u;
  __syncthreads();
  uint32_t x = threadIdx.x * 0x9e3779b1u + blockIdx.x * 0x85ebca6bu + 1u, acc = 0;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32_t a = MODE >= 4 ? 0u : (x >> (32 - LOG));
    if (MODE == 0 || MODE == 4) acc += s[a];
    else if (MODE == 1 || MODE == 5) s[a] = x;
    else if (MODE == 2 || MODE == 6) atomicAdd(&s[a], 1u);
    else acc += atomicOr(&s[a], x & 0xffu);
  }
  __syncthreads();
  if (acc == 0x12345u || s[threadIdx.x] == 0x7777777u) sink[0] = acc;
}

template <int MODE, int LOG>
static int run(const char* name, uint32_t* sink) {
  const int iters = 2048, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_probe<MODE, LOG>), dim3(blocks), dim3(256), 0, 0, sink, iters);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((k_probe<MODE, LOG>), dim3(blocks), dim3(256), 0, 0, sink, iters);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)blocks * 256 * iters;
  printf("%-34s table 2^%2d dwords: %8.3f ms  %7.2f lane-ops / ns chip-wide = %6.2f per CU and clock (2.4 GHz)\n", name, LOG, ms,
         ops / (ms * 1e6), ops / (ms * 1e-3) / 256 / 2.4e9);
  return 0;
}

int main() {
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 64));
  run<0, 11>("read, random", sink);
  run<1, 11>("write, random", sink);
  run<2, 11>("atomic add (no return), random", sink);
  run<3, 11>("atomic or (return), random", sink);
  run<2, 7>("atomic add (no return), random", sink);
  run<3, 7>("atomic or (return), random", sink);
  run<4, 11>("read, one address", sink);
  run<6, 11>("atomic add (no return), one addr", sink);
  run<7, 11>("atomic or (return), one addr", sink);
  return 0;
}
