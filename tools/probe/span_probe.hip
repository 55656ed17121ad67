// What limits k_classify's streaming skeleton?  Waves that each stream a SPAN of 16 K records from five columns
// (pos, ref, alt, qual: 4 B; flags: 1 B) one round (256 records) ahead, nothing else -- in a few shapes.
// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/probe/span_probe tools/probe/span_probe.hip && tools/probe/span_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SPAN = 16384;
// VAR 0: as k_classify: per round 4 x dwordx4 + 1 dword (flags), next round in flight
// VAR 1: flags fetched once per tile (dwordx4 per lane = 1 KiB covers 4 rounds)
// VAR 2: no flags at all (16 B per record)
// VAR 3: two rounds in flight
template <int VAR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const unsigned* __restrict__ c0, const unsigned* __restrict__ c1, const unsigned* __restrict__ c2,
                                               const unsigned* __restrict__ c3, const unsigned char* __restrict__ fl, int64_t nspans, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int64_t span = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (span >= nspans) return;
  const int64_t base = span * SPAN;
  v4u acc = {0, 0, 0, 0};
  unsigned facc = 0;
  auto ld = [&](const unsigned* c, int64_t i) { return __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c + i)); };
  v4u a0 = ld(c0, base + lane * 4), a1 = ld(c1, base + lane * 4), a2 = ld(c2, base + lane * 4), a3 = ld(c3, base + lane * 4);
  unsigned f = 0;
  v4u f4 = {0, 0, 0, 0};
  if (VAR == 0 || VAR == 3) f = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + base + lane * 4));
  v4u b0 = a0, b1 = a1, b2 = a2, b3 = a3; unsigned g = 0;
  if (VAR == 3) { b0 = ld(c0, base + 256 + lane * 4); b1 = ld(c1, base + 256 + lane * 4); b2 = ld(c2, base + 256 + lane * 4); b3 = ld(c3, base + 256 + lane * 4);
                  g = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + base + 256 + lane * 4)); }
  for (int r = 0; r < SPAN / 256; ++r) {
    const v4u x0 = a0, x1 = a1, x2 = a2, x3 = a3; const unsigned xf = f;
    const int64_t nx = base + (int64_t)(r + (VAR == 3 ? 2 : 1)) * 256 + lane * 4;
    if (VAR == 3) { a0 = b0; a1 = b1; a2 = b2; a3 = b3; f = g; }
    if (r + (VAR == 3 ? 2 : 1) < SPAN / 256) {
      if (VAR == 3) { b0 = ld(c0, nx); b1 = ld(c1, nx); b2 = ld(c2, nx); b3 = ld(c3, nx); g = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + nx)); }
      else { a0 = ld(c0, nx); a1 = ld(c1, nx); a2 = ld(c2, nx); a3 = ld(c3, nx);
             if (VAR == 0) f = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + nx)); }
    }
    if (VAR == 1 && (r & 3) == 0) f4 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(fl + base + r * 256 + lane * 16));
    acc ^= x0 ^ x1 ^ x2 ^ x3; facc ^= xf ^ f4.x ^ f4.w;
    // a little dependent work per round, like the real kernel has (keeps the compiler from hoisting all loads)
    acc.x = acc.x * 2654435761u + (acc.y >> 3);
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
  const int64_t n = 1000000000ll / SPAN * SPAN, nspans = n / SPAN;   // ~1e9 records like the bench batch
  unsigned *c[4]; unsigned char* fl; unsigned* sink;
  for (auto& p : c) { (void)hipMalloc(&p, n * 4 + 4096); (void)hipMemset(p, 1, n * 4 + 4096); }
  (void)hipMalloc(&fl, n + 4096); (void)hipMemset(fl, 3, n + 4096); (void)hipMalloc(&sink, 4);
#define RUN(VAR, W, BYTES, NAME) { double ms = timeit([&] { hipLaunchKernelGGL((k<VAR, W>), dim3((unsigned)((nspans + W - 1) / W)), dim3(64 * W), 0, 0, c[0], c[1], c[2], c[3], fl, nspans, sink); }, 5); \
    printf("%-58s waves/block=%d : %6.3f ms  %7.1f GB/s\n", NAME, W, ms, (double)n * BYTES / ms / 1e6); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, 1, 17, "as k_classify (4 x 16 B + 4 B flags per lane and round)")
    RUN(1, 1, 17, "flags once per tile (16 B per lane)")
    RUN(2, 1, 16, "no flags column")
    RUN(3, 1, 17, "two rounds in flight")
    RUN(0, 4, 17, "as k_classify, 4 independent waves per block")
    RUN(1, 4, 17, "flags once per tile, 4 waves per block")
    RUN(2, 4, 16, "no flags column, 4 waves per block")
  }
  return 0;
}
