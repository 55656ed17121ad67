// Does the rate of plain streaming stores depend on where hipMalloc put the buffer?  One process; a 4 GiB buffer allocated,
// written (16-byte-per-lane stores, 4 KB per wave = the shape that reaches the part's best write rate), freed, allocated again ...
// between allocations other buffers of varying size are allocated and freed to move the allocator.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/alloc_store tools/probe/alloc_store.hip && /tmp/alloc_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void kw(v4u* dst, int64_t per_wave16) {
  v4u* base = dst + (int64_t)blockIdx.x * per_wave16;
  const v4u x = {(unsigned)blockIdx.x, 1, 2, threadIdx.x};
  for (int64_t j = threadIdx.x; j < per_wave16; j += 64) __builtin_nontemporal_store(x, base + j);
}
__global__ void kr(const v4u* src, int64_t per_wave16, unsigned* sink) {
  const v4u* base = src + (int64_t)blockIdx.x * per_wave16;
  v4u a = {0, 0, 0, 0};
  for (int64_t j = threadIdx.x; j < per_wave16; j += 64) a ^= __builtin_nontemporal_load(base + j);
  if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) *sink = 1;
}
int main(int argc, char**) {
  const int64_t bytes = 4ll << 30;
  hipEvent_t a, e; (void)hipEventCreate(&a); (void)hipEventCreate(&e);
  unsigned* sink; (void)hipMalloc(&sink, 4);
  std::vector<void*> junk;
  for (int rep = 0; rep < 10; ++rep) {
    v4u* d = nullptr;
    const bool contig = argc > 1 && rep % 2 == 1;   // any argument: every other allocation physically contiguous (hipDeviceMallocContiguous)
    if ((contig ? hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocContiguous) : hipMalloc(&d, bytes)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("%s ", contig ? "contiguous" : "plain     ");
    (void)hipMemset(d, 1, bytes);
    float best_w = 1e9f, best_r = 1e9f, worst_w = 0.f;
    for (int pw : {4096}) {
      const int64_t p16 = pw / 16;
      const int nblk = (int)(bytes / pw);
      for (int k = 0; k < 5; ++k) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(kw, dim3(nblk), dim3(64), 0, 0, d, p16); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, a, e);
        if (k) { if (ms < best_w) best_w = ms; if (ms > worst_w) worst_w = ms; }
      }
      for (int k = 0; k < 4; ++k) {
        (void)hipEventRecord(a); hipLaunchKernelGGL(kr, dim3(nblk), dim3(64), 0, 0, d, p16, sink); (void)hipEventRecord(e); (void)hipEventSynchronize(e);
        float ms; (void)hipEventElapsedTime(&ms, a, e);
        if (k && ms < best_r) best_r = ms;
      }
    }
    printf("allocation %d at %p: write %.3f..%.3f ms (%.0f GB/s)  read %.3f ms (%.0f GB/s)\n", rep, (void*)d, best_w, worst_w, bytes / best_w / 1e6, best_r, bytes / best_r / 1e6);
    fflush(stdout);
    (void)hipFree(d);
    // move the allocator: something else of another size stays allocated from now on
    void* j = nullptr;
    if (hipMalloc(&j, (size_t)(rep + 1) * (300ull << 20)) == hipSuccess) junk.push_back(j);
  }
  return 0;
}
