// span_probe3: does the LAYOUT of the five columns matter to the span-streaming skeleton?  (a) five separate arrays, as the
// batch keeps them; (b) one array, the columns of every 256-record round next to each other (4 352 bytes per round): the same
// load instructions, one contiguous stream per wave instead of five.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SPAN = 16384;
constexpr int ROUND_BYTES = 256 * 17;
template <int DEPTH, bool INTERLEAVED>
__global__ __launch_bounds__(64) void k(const unsigned* __restrict__ c0, const unsigned* __restrict__ c1, const unsigned* __restrict__ c2,
                                        const unsigned* __restrict__ c3, const unsigned char* __restrict__ fl, const unsigned char* __restrict__ il,
                                        unsigned* sink) {
  __shared__ unsigned lds[2000];   // 20 waves per CU, as k_classify
  const int lane = threadIdx.x;
  const int64_t span = blockIdx.x;
  const int64_t base = span * SPAN;
  for (int i = lane; i < 2000; i += 64) lds[i] = (unsigned)i;
  __syncthreads();
  v4u acc = {0, 0, 0, 0};
  unsigned facc = 0;
  v4u a[DEPTH][4]; unsigned f[DEPTH];
  auto load = [&](int d, int64_t round) {
    if (INTERLEAVED) {
      const unsigned char* t = il + ((base >> 8) + round) * ROUND_BYTES;
      a[d][0] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t) + lane);
      a[d][1] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 1024) + lane);
      a[d][2] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 2048) + lane);
      a[d][3] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 3072) + lane);
      f[d] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(t + 4096) + lane);
    } else {
      const int64_t i = base + round * 256 + lane * 4;
      a[d][0] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c0 + i));
      a[d][1] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c1 + i));
      a[d][2] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c2 + i));
      a[d][3] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c3 + i));
      f[d] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + i));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, d);
  for (int r = 0; r < SPAN / 256; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const v4u x0 = a[d][0], x1 = a[d][1], x2 = a[d][2], x3 = a[d][3]; const unsigned xf = f[d];
      if (r + d + DEPTH < SPAN / 256) load(d, r + d + DEPTH);
      acc ^= x0 ^ x1 ^ x2 ^ x3; facc ^= xf;
      acc.y += lds[(acc.x & 1023u)];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ facc) == 0x12345u) *sink = 1;
}
template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const int64_t n = 1000000000ll / SPAN * SPAN, nspans = n / SPAN;
  unsigned *c[4]; unsigned char *fl, *il; unsigned* sink;
  for (auto& p : c) { (void)hipMalloc(&p, n * 4 + 8192); (void)hipMemset(p, 1, n * 4 + 8192); }
  (void)hipMalloc(&fl, n + 8192); (void)hipMemset(fl, 3, n + 8192); (void)hipMalloc(&sink, 4);
  (void)hipMalloc(&il, n * 17 + 8192); (void)hipMemset(il, 1, n * 17 + 8192);
#define RUN(DEPTH, IL) { double ms = timeit([&] { hipLaunchKernelGGL((k<DEPTH, IL>), dim3((unsigned)nspans), dim3(64), 0, 0, c[0], c[1], c[2], c[3], fl, il, sink); }, 5); \
    printf("%-22s rounds in flight %d: %6.3f ms  %7.1f GB/s\n", IL ? "one interleaved array" : "five arrays", DEPTH, ms, (double)n * 17 / ms / 1e6); }
  for (int rep = 0; rep < 3; ++rep) { RUN(1, false) RUN(1, true) RUN(2, false) RUN(2, true) }
  return 0;
}
