// span_probe2: the span-streaming skeleton at k_classify's occupancy (LDS-limited to 20 or 16 waves per CU), with one or
// two rounds in flight, and with a dependent LDS chain per round standing in for the join (latency, not issue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SPAN = 16384;
template <int DEPTH, int LDSDW, int CHAIN>
__global__ __launch_bounds__(64) void k(const unsigned* __restrict__ c0, const unsigned* __restrict__ c1, const unsigned* __restrict__ c2,
                                        const unsigned* __restrict__ c3, const unsigned char* __restrict__ fl, int64_t nspans, unsigned* sink) {
  __shared__ unsigned lds[LDSDW];
  const int lane = threadIdx.x;
  const int64_t span = blockIdx.x;
  const int64_t base = span * SPAN;
  for (int i = lane; i < LDSDW; i += 64) lds[i] = (unsigned)(i * 7 + 1) % LDSDW;
  __syncthreads();
  v4u acc = {0, 0, 0, 0};
  unsigned facc = 0, p = lane;
  auto ld = [&](const unsigned* c, int64_t i) { return __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c + i)); };
  v4u a[DEPTH][4]; unsigned f[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int64_t i = base + d * 256 + lane * 4;
    a[d][0] = ld(c0, i); a[d][1] = ld(c1, i); a[d][2] = ld(c2, i); a[d][3] = ld(c3, i);
    f[d] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + i));
  }
  for (int r = 0; r < SPAN / 256; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const v4u x0 = a[d][0], x1 = a[d][1], x2 = a[d][2], x3 = a[d][3]; const unsigned xf = f[d];
      if (r + d + DEPTH < SPAN / 256) {
        const int64_t i = base + (int64_t)(r + d + DEPTH) * 256 + lane * 4;
        a[d][0] = ld(c0, i); a[d][1] = ld(c1, i); a[d][2] = ld(c2, i); a[d][3] = ld(c3, i);
        f[d] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + i));
      }
      acc ^= x0 ^ x1 ^ x2 ^ x3; facc ^= xf;
      p ^= acc.x & 63u;
#pragma unroll 1
      for (int c = 0; c < CHAIN; ++c) p = lds[p % LDSDW];   // dependent LDS round trips
      acc.y += p;
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
  unsigned *c[4]; unsigned char* fl; unsigned* sink;
  for (auto& p : c) { (void)hipMalloc(&p, n * 4 + 8192); (void)hipMemset(p, 1, n * 4 + 8192); }
  (void)hipMalloc(&fl, n + 8192); (void)hipMemset(fl, 3, n + 8192); (void)hipMalloc(&sink, 4);
#define RUN(DEPTH, LDSDW, CHAIN) { double ms = timeit([&] { hipLaunchKernelGGL((k<DEPTH, LDSDW, CHAIN>), dim3((unsigned)nspans), dim3(64), 0, 0, c[0], c[1], c[2], c[3], fl, nspans, sink); }, 5); \
    printf("rounds in flight %d, LDS %5d B per wave (%2d waves/CU), LDS chain %2d per round: %6.3f ms  %7.1f GB/s\n", DEPTH, LDSDW * 4, (160 * 1024) / (LDSDW * 4) > 32 ? 32 : (160 * 1024) / (LDSDW * 4), CHAIN, ms, (double)n * 17 / ms / 1e6); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN(1, 1024, 0) RUN(1, 2000, 0) RUN(1, 2560, 0) RUN(1, 3400, 0)
    RUN(2, 1024, 0) RUN(2, 2000, 0) RUN(2, 2560, 0) RUN(2, 3400, 0)
    RUN(1, 2000, 10) RUN(1, 2000, 20) RUN(1, 2000, 40) RUN(2, 2000, 20) RUN(2, 2560, 20) RUN(2, 2560, 40) RUN(1, 1024, 20) RUN(1, 1024, 40)
  }
  return 0;
}
