// span_probe4: does the span-streaming skeleton depend on WHERE its arrays landed, and does the layout decide how much?
// ALLOCS times: fresh allocations (a dummy of changing size in between), then the skeleton of span_probe3 over (a) five separate
// column arrays and (b) one round-interleaved array (4 352 bytes per 256-record round), each without and with the mask words of
// k_classify (two 1 KiB stores per 8 tiles into two more arrays).
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/span_probe4 tools/probe/span_probe4.hip && /tmp/span_probe4 [allocations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SPAN = 16384;
constexpr int ROUND_BYTES = 256 * 17;
template <bool INTERLEAVED, int MASKS>
__global__ __launch_bounds__(64) void k(const unsigned* __restrict__ c0, const unsigned* __restrict__ c1, const unsigned* __restrict__ c2,
                                        const unsigned* __restrict__ c3, const unsigned char* __restrict__ fl, const unsigned char* __restrict__ il,
                                        unsigned* __restrict__ m0, unsigned* __restrict__ m1, unsigned* sink) {
  __shared__ unsigned lds[2000];   // 20 waves per CU, as k_classify
  const int lane = threadIdx.x;
  const int64_t span = blockIdx.x;
  const int64_t base = span * SPAN;
  for (int i = lane; i < 2000; i += 64) lds[i] = (unsigned)i;
  __syncthreads();
  v4u acc = {0, 0, 0, 0};
  unsigned facc = 0;
  v4u a[4]; unsigned f;
  auto load = [&](int64_t round) {
    if (INTERLEAVED) {
      const unsigned char* t = il + ((base >> 8) + round) * ROUND_BYTES;
      a[0] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t) + lane);
      a[1] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 1024) + lane);
      a[2] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 2048) + lane);
      a[3] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(t + 3072) + lane);
      f = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(t + 4096) + lane);
    } else {
      const int64_t i = base + round * 256 + lane * 4;
      a[0] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c0 + i));
      a[1] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c1 + i));
      a[2] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c2 + i));
      a[3] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(c3 + i));
      f = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(fl + i));
    }
  };
  load(0);
  for (int r = 0; r < SPAN / 256; ++r) {
    const v4u x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3]; const unsigned xf = f;
    if (r + 1 < SPAN / 256) load(r + 1);
    acc ^= x0 ^ x1 ^ x2 ^ x3; facc ^= xf;
    acc.y += lds[(acc.x & 1023u)];
    // MASKS 1: as k_classify -- every 8 tiles (32 rounds) 1 KiB into each of two arrays; 2: the two pieces next to each other in one array;
    // 3: once per span, 2 KiB into each of two arrays; 4: once per span, one 4 KiB piece; 5: as 1 with cached stores
    if ((MASKS == 1 || MASKS == 2 || MASKS == 5) && (r & 31) == 31) {
      const int64_t w = (base + (int64_t)(r - 31) * 256) >> 5;   // first mask word of the batch
      v4u* d0 = reinterpret_cast<v4u*>(MASKS == 2 ? m0 + 2 * w : m0 + w) + lane;
      v4u* d1 = reinterpret_cast<v4u*>(MASKS == 2 ? m0 + 2 * w + 256 : m1 + w) + lane;
      if (MASKS == 5) { *d0 = acc; *d1 = acc ^ x1; }
      else { __builtin_nontemporal_store(acc, d0); __builtin_nontemporal_store(acc ^ x1, d1); }
    }
    if ((MASKS == 3 || MASKS == 4) && r == SPAN / 256 - 1) {
      const int64_t w = base >> 5;
      v4u* d0 = reinterpret_cast<v4u*>(MASKS == 4 ? m0 + 2 * w : m0 + w) + lane;
      v4u* d1 = reinterpret_cast<v4u*>(MASKS == 4 ? m0 + 2 * w + 512 : m1 + w) + lane;
      __builtin_nontemporal_store(acc, d0); __builtin_nontemporal_store(acc ^ x0, d0 + 64);
      __builtin_nontemporal_store(acc ^ x1, d1); __builtin_nontemporal_store(acc ^ x2, d1 + 64);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ facc) == 0x12345u) *sink = 1;
}
template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms / reps;
}
int main(int argc, char** argv) {
  const int allocs = argc > 1 ? atoi(argv[1]) : 6;
  const int64_t n = 1000000000ll / SPAN * SPAN, nspans = n / SPAN;
  for (int al = 0; al < allocs; ++al) {
    void* dummy = nullptr;
    (void)hipMalloc(&dummy, (size_t)(1 + (al * 7) % 5) << 28);   // 0.25 ... 1.25 GB in the way of the next allocations
    unsigned *c[4]; unsigned char *fl, *il; unsigned *sink, *m0, *m1;
    for (auto& p : c) { if (hipMalloc(&p, n * 4 + 8192) != hipSuccess) { printf("alloc failed\n"); return 1; } (void)hipMemset(p, 1, n * 4 + 8192); }
    (void)hipMalloc(&fl, n + 8192); (void)hipMemset(fl, 3, n + 8192); (void)hipMalloc(&sink, 4);
    if (hipMalloc(&il, n * 17 + 8192) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(il, 1, n * 17 + 8192);
    (void)hipMalloc(&m0, n / 4 + 16384); (void)hipMalloc(&m1, n / 8 + 8192);   // (m0 holds both masks in the adjacent variants)
    double t[7];
#define RUN(IL, MK, slot) t[slot] = timeit([&] { hipLaunchKernelGGL((k<IL, MK>), dim3((unsigned)nspans), dim3(64), 0, 0, c[0], c[1], c[2], c[3], fl, il, m0, m1, sink); }, 5);
    RUN(false, 0, 0) RUN(true, 0, 1) RUN(false, 1, 2) RUN(false, 2, 3) RUN(false, 3, 4) RUN(false, 4, 5) RUN(false, 5, 6)
    printf("allocation %d: no masks: five arrays %.3f interleaved %.3f | five arrays + masks: as k_classify %.3f  adjacent %.3f  per span 2x2K %.3f  per span 4K %.3f  cached %.3f ms\n",
           al, t[0], t[1], t[2], t[3], t[4], t[5], t[6]);
    fflush(stdout);
    if (argc > 2) {   // second argument: on THIS allocation of the columns, that many fresh allocations of the two mask arrays alone
      for (int ma = 0; ma < atoi(argv[2]); ++ma) {
        void* d2 = nullptr;
        (void)hipMalloc(&d2, (size_t)(1 + (ma * 3) % 7) << 25);   // 32 ... 224 MB in the way
        unsigned *n0, *n1;
        (void)hipMalloc(&n0, n / 4 + 16384); (void)hipMalloc(&n1, n / 8 + 8192);
        const double tm = timeit([&] { hipLaunchKernelGGL((k<false, 1>), dim3((unsigned)nspans), dim3(64), 0, 0, c[0], c[1], c[2], c[3], fl, il, n0, n1, sink); }, 5);
        printf("    mask arrays %d: %.3f ms\n", ma, tm);
        (void)hipFree(n0); (void)hipFree(n1); (void)hipFree(d2);
      }
      fflush(stdout);
    }
    for (auto& p : c) (void)hipFree(p);
    (void)hipFree(fl); (void)hipFree(il); (void)hipFree(sink); (void)hipFree(m0); (void)hipFree(m1); (void)hipFree(dummy);
  }
  return 0;
}
