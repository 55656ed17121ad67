// What bounds k_compact's stores?  One wave per block (as k_compact: one wave per span), every wave writes its own contiguous
// region(s) with 16-byte-per-lane stores, 1 KiB per wave-instruction.  Knobs: bytes per wave, a second (thin) stream per wave,
// how many 1 KiB stores leave back to back, idle cycles between bursts, store flavour, block -> region mapping.
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shape tools/probe/store_shape.hip && /tmp/store_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Cfg {
  int64_t per_wave;   // bytes of the main stream per wave (multiple of 1 KiB)
  int64_t thin;       // bytes of the second stream per wave (0 = none), written 1 KiB per `thin_every` main KiB
  int burst;          // 1 KiB stores back to back
  int idle;           // s_sleep units (64 clocks each) between bursts
  int flavour;        // 0 nt, 1 plain, 2 sc1 (write-through)
  int map;            // 0 block b owns region b; 1 XCD-contiguous: the 8 XCDs own eighths of the regions
  int bt;             // threads per block (64 or 256): with 256 the four waves write neighbouring KiB
};
template <int FL> __device__ __forceinline__ void st(v4u* p, v4u x) {
  if (FL == 0) __builtin_nontemporal_store(x, p);
  else if (FL == 1) *p = x;
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(x) : "memory");
}
template <int FL>
__global__ void k(v4u* dst, v4u* dst2, Cfg c, int nblocks) {
  int b = blockIdx.x;
  if (c.map == 1) { const int per = (nblocks + 7) / 8; b = (b & 7) * per + (b >> 3); if (b >= nblocks) return; }
  const int waves = c.bt / 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t kib = c.per_wave * waves / 1024;            // KiB of the block
  v4u* base = dst + (int64_t)b * (c.per_wave * waves / 16);
  v4u* base2 = dst2 + (int64_t)b * (c.thin * waves / 16);
  const int64_t thin_kib = c.thin * waves / 1024;
  int64_t t2 = 0;
  v4u x = {(unsigned)b, 1, 2, (unsigned)lane};
  for (int64_t i = wave; i < kib; i += (int64_t)waves * c.burst) {
    for (int u = 0; u < c.burst; ++u) {
      const int64_t j = i + (int64_t)u * waves;
      if (j < kib) st<FL>(base + j * 64 + lane, x);
    }
    if (thin_kib && wave == 0 && t2 < thin_kib && (i * thin_kib) / kib >= t2) { st<FL>(base2 + t2 * 64 + lane, x); ++t2; }
    for (int s = 0; s < c.idle; ++s) __builtin_amdgcn_s_sleep(1);
    x.y += 1;
  }
}
// k_compact's traffic in the small: every wave first READS `rd` bytes per lane-group from a separate buffer (its mask bytes: 64 B per
// wave-instruction, at its own place), then writes its piece, 1 KiB per instruction -- do a few reads among the writes change the rate?
template <int FL>
__global__ void krw(v4u* dst, const unsigned char* src, int64_t piece, int nrd, int idle) {
  const int lane = threadIdx.x & 63;
  const int64_t b = blockIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < nrd; ++r) acc += src[(b * nrd + r) * 64 + lane];          // nrd x 64 B, contiguous per wave
  v4u x = {(unsigned)b, acc, 2, (unsigned)lane};
  v4u* base = dst + b * (piece / 16);
  for (int64_t j = 0; j < piece / 1024; ++j) {
    st<FL>(base + j * 64 + lane, x);
    for (int s2 = 0; s2 < idle; ++s2) __builtin_amdgcn_s_sleep(1);
    x.y += 1;
  }
}
// persistent, strided: the grid fills the chip once; wave g writes pieces g, g + G, g + 2G, ... of `piece` bytes each (G = all
// waves of the grid), so that what the chip writes at any one time is ONE window of G pieces that moves through the buffer
template <int FL>
__global__ void kp(v4u* dst, int64_t piece, int64_t npieces, int idle, int xmap) {
  const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t G = (int64_t)gridDim.x * waves;
  int64_t g = (int64_t)blockIdx.x * waves + wave;
  if (xmap) { const int64_t per = G / 8; g = (blockIdx.x & 7) * per + (int64_t)(blockIdx.x >> 3) * waves + wave; }   // every XCD owns an eighth of the window
  v4u x = {(unsigned)g, 1, 2, (unsigned)lane};
  for (int64_t i = g; i < npieces; i += G) {
    v4u* base = dst + i * (piece / 16);
    for (int64_t j = 0; j < piece / 1024; ++j) st<FL>(base + j * 64 + lane, x);
    for (int s = 0; s < idle; ++s) __builtin_amdgcn_s_sleep(1);
    x.y += 1;
  }
}
int main() {
  const int64_t bytes = 4ll << 30;
  v4u *d, *d2;
  if (getenv("STORE_CONTIG")) {   // physically contiguous memory (hipDeviceMallocContiguous): linear physical addresses
    if (hipExtMallocWithFlags((void**)&d, bytes, hipDeviceMallocContiguous) != hipSuccess || hipExtMallocWithFlags((void**)&d2, bytes / 4, hipDeviceMallocContiguous) != hipSuccess) { printf("no contiguous memory\n"); return 1; }
    printf("physically contiguous buffers\n");
  } else { hipMalloc(&d, bytes); hipMalloc(&d2, bytes / 4); }
  hipMemset(d, 1, bytes); hipMemset(d2, 1, bytes / 4);
  hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
  auto run = [&](const char* name, Cfg c) {
    const int nblocks = (int)(bytes / (c.per_wave * (c.bt / 64)));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      if (c.flavour == 0) hipLaunchKernelGGL(k<0>, dim3(nblocks), dim3(c.bt), 0, 0, d, d2, c, nblocks);
      else if (c.flavour == 1) hipLaunchKernelGGL(k<1>, dim3(nblocks), dim3(c.bt), 0, 0, d, d2, c, nblocks);
      else hipLaunchKernelGGL(k<2>, dim3(nblocks), dim3(c.bt), 0, 0, d, d2, c, nblocks);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, a, e);
      if (rep && ms < best) best = ms;
    }
    const double tot = (double)nblocks * (c.per_wave + c.thin) * (c.bt / 64);
    printf("%-44s per_wave=%7lld thin=%6lld burst=%d idle=%3d fl=%d map=%d bt=%3d : %.3f ms %7.1f GB/s\n", name, (long long)c.per_wave,
           (long long)c.thin, c.burst, c.idle, c.flavour, c.map, c.bt, best, tot / best / 1e6);
    fflush(stdout);
  };
  // the span shape: 60 KiB per wave
  for (int fl = 0; fl < 3; ++fl) run("span 60K", {61440, 0, 1, 0, fl, 0, 64});
  run("span 60K + thin 5K", {61440, 5120, 1, 0, 0, 0, 64});
  run("span 60K burst 2", {61440, 0, 2, 0, 0, 0, 64});
  run("span 60K burst 4", {61440, 0, 4, 0, 0, 0, 64});
  run("span 60K idle 4 (256 clk)", {61440, 0, 1, 4, 0, 0, 64});
  run("span 60K idle 16 (1k clk)", {61440, 0, 1, 16, 0, 0, 64});
  run("span 60K burst 2 idle 32", {61440, 0, 2, 32, 0, 0, 64});
  run("span 60K burst 4 idle 64", {61440, 0, 4, 64, 0, 0, 64});
  run("span 60K xcd-contiguous", {61440, 0, 1, 0, 0, 1, 64});
  run("span 60K xcd-contiguous plain", {61440, 0, 1, 0, 1, 1, 64});
  run("span 60K xcd-contiguous idle 16", {61440, 0, 1, 16, 0, 1, 64});
  // shorter waves
  for (int64_t pw : {4096ll, 8192ll, 16384ll, 32768ll, 131072ll, 262144ll}) run("per-wave sweep nt", {pw, 0, 1, 0, 0, 0, 64});
  for (int64_t pw : {4096ll, 16384ll, 262144ll}) run("per-wave sweep plain", {pw, 0, 1, 0, 1, 0, 64});
  for (int64_t pw : {4096ll, 16384ll}) run("per-wave sweep nt idle 16", {pw, 0, 1, 16, 0, 0, 64});
  // four waves per block writing neighbouring KiB
  run("block 256: 240K", {61440, 0, 1, 0, 0, 0, 256});
  run("block 256: 240K plain", {61440, 0, 1, 0, 1, 0, 256});
  run("block 256: 16K", {4096, 0, 1, 0, 0, 0, 256});
  run("block 256: 16K plain", {4096, 0, 1, 0, 1, 0, 256});
  run("block 256: 240K idle 16", {61440, 0, 1, 16, 0, 0, 256});
  auto runp = [&](const char* name, int blocks, int bt, int64_t piece, int idle, int fl, int xmap) {
    const int64_t npieces = bytes / piece;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      if (fl == 0) hipLaunchKernelGGL(kp<0>, dim3(blocks), dim3(bt), 0, 0, d, piece, npieces, idle, xmap);
      else hipLaunchKernelGGL(kp<1>, dim3(blocks), dim3(bt), 0, 0, d, piece, npieces, idle, xmap);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, a, e);
      if (rep && ms < best) best = ms;
    }
    printf("%-28s blocks=%5d bt=%4d piece=%6lld idle=%3d fl=%d xmap=%d : %.3f ms %7.1f GB/s\n", name, blocks, bt, (long long)piece, idle, fl, xmap,
           best, (double)bytes / best / 1e6);
    fflush(stdout);
  };
  {
    unsigned char* src; hipMalloc(&src, 1ll << 30); hipMemset(src, 1, 1ll << 30);
    auto runrw = [&](const char* name, int64_t piece, int nrd, int idle, int fl) {
      const int nblocks = (int)(bytes / piece);
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        if (fl == 0) hipLaunchKernelGGL(krw<0>, dim3(nblocks), dim3(64), 0, 0, d, src, piece, nrd, idle);
        else hipLaunchKernelGGL(krw<1>, dim3(nblocks), dim3(64), 0, 0, d, src, piece, nrd, idle);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, a, e);
        if (rep && ms < best) best = ms;
      }
      printf("%-28s piece=%6lld reads=%2d x 64 B idle=%3d fl=%d : %.3f ms %7.1f GB/s written (+ %.0f MB read)\n", name, (long long)piece, nrd, idle, fl, best,
             (double)bytes / best / 1e6, (double)nblocks * nrd * 64 / 1e6);
      fflush(stdout);
    };
    for (int nrd : {0, 1, 4, 16}) runrw("read then write", 16384, nrd, 0, 0);
    for (int nrd : {0, 4, 16}) runrw("read then write, idle 8", 16384, nrd, 8, 0);
    for (int nrd : {0, 2, 8}) runrw("read then write 4K", 4096, nrd, 0, 0);
  }
  for (int fl = 0; fl < 2; ++fl)
    for (int xmap = 0; xmap < 2; ++xmap) {
      runp("persistent 8 WG/CU x 4 waves", 2048, 256, 4096, 0, fl, xmap);
      runp("persistent, idle 16", 2048, 256, 4096, 16, fl, xmap);
      runp("persistent, idle 48", 2048, 256, 4096, 48, fl, xmap);
    }
  runp("persistent 4 WG/CU", 1024, 256, 4096, 16, 1, 1);
  runp("persistent 2 WG/CU", 512, 256, 4096, 16, 1, 1);
  runp("persistent 1-wave WGs", 8192, 64, 4096, 16, 1, 0);
  runp("persistent piece 1K", 2048, 256, 1024, 4, 1, 1);
  runp("persistent piece 16K", 2048, 256, 16384, 64, 1, 1);
  return 0;
}
