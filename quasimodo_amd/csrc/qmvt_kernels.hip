// qmvt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the variant-truth engine.
//
// Pure integer / index work, HBM-bound: no MFMA.  wave = 64 lanes everywhere.
//
//   k_classify   one wave per SPAN (<= SPAN_TILES consecutive 512-record tiles of
//                one VCF), no barriers.  Streams pos/ref/alt/qual (dwordx4 per lane)
//                + flags (dword per lane) with the next tile in flight, stages each
//                tile's slice of the sorted truth keys in double-buffered LDS and
//                merge-joins by LDS binary search.  Emits
//                wave-ballot class masks (kept / TP, 1 bit per record each),
//                per-tile TP/FP line counts, per-span QUAL-bin histograms
//                (TP, FP, distinct truth keys) and scalar counters.
//   k_finalize   one workgroup per VCF: span histograms -> ROC suffix sums,
//                scalars, exclusive scan of tile counts, per-truth-set sums.
//   k_compact    one wave per tile: expands the ballot masks into the
//                compacted TP / FP line-index lists (mbcnt prefix ranks).
//   k_sort_*     per-VCF LSD radix sort (wave multisplit) for unsorted VCFs.
//   k_synth_*    on-device generator of the BASELINE.json config-3/4 workload.
//
// Reference stages replaced (file:line in /root/reference):
//   fgrep -wf / -wvf            program/extract_TP_FP_SNPs.py:50-57
//   R intersect/setdiff/length  scripts/caller_performance_compare.R:94-96
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmvt_dev.h"

namespace qm {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }

__device__ __forceinline__ int qual_bin(float q, int n_bins) {
  // floor(q) clamped to [-1, n_bins-1]; NaN and negatives -> -1 (passes no threshold)
  if (!(q >= 0.0f)) return -1;
  if (q >= (float)n_bins) return n_bins - 1;
  return (int)q;  // q >= 0: truncation == floor
}

__device__ __forceinline__ bool is_snp(int r, int a) { return ((uint32_t)r < 4u) & ((uint32_t)a < 4u); }
__device__ __forceinline__ uint32_t pack_key(int p, int r, int a) {
  return ((uint32_t)p << 4) | ((uint32_t)r << 2) | (uint32_t)a;
}

// number of slice keys < key (lower bound), uniform trip count over the block.
// `top` = largest power of two <= m (0 when m == 0).
__device__ __forceinline__ int lds_lower_bound(const uint32_t* __restrict__ k, int m, int top, uint32_t key) {
  int pos = 0;
  for (int step = top; step > 0; step >>= 1) {
    int idx = pos + step;
    int rd = idx <= m ? idx - 1 : m - 1;
    uint32_t v = k[rd];
    if (idx <= m && v < key) pos = idx;
  }
  return pos;
}

// ---------------------------------------------------------------------------
// k_classify -- wave-autonomous streaming merge-join.
//
// One workgroup = ONE wave (64 lanes) = one span of up to SPAN_TILES consecutive
// tiles of one VCF; no workgroup barriers.  Units:
//   round = 256 records, register resident (4 consecutive records per lane:
//           dwordx4 per column + one dword of flags).  The next round's loads are
//           issued before the current round is touched (register double buffer).
//   tile  = K1_ROUNDS rounds = the unit that owns a slice of the sorted truth keys
//           in LDS (double buffered: tile t+1's slice is staged while t finishes),
//           the per-truth-entry state for U(t)/TP_R, and one TP/FP line count.
// ---------------------------------------------------------------------------
struct Rec4 {
  int p[4], r[4], a[4];
  float q[4];
  uint32_t f;  // 4 flag bytes
};

__device__ __forceinline__ void load_rec4(const ClassifyParams& P, int64_t idx, Rec4& R) {
  const int4 pv = *reinterpret_cast<const int4*>(P.pos + idx);
  const int4 rv = *reinterpret_cast<const int4*>(P.ref + idx);
  const int4 av = *reinterpret_cast<const int4*>(P.alt + idx);
  const float4 qv = *reinterpret_cast<const float4*>(P.qual + idx);
  R.f = *reinterpret_cast<const uint32_t*>(P.flags + idx);
  R.p[0] = pv.x; R.p[1] = pv.y; R.p[2] = pv.z; R.p[3] = pv.w;
  R.r[0] = rv.x; R.r[1] = rv.y; R.r[2] = rv.z; R.r[3] = rv.w;
  R.a[0] = av.x; R.a[1] = av.y; R.a[2] = av.z; R.a[3] = av.w;
  R.q[0] = qv.x; R.q[1] = qv.y; R.q[2] = qv.z; R.q[3] = qv.w;
}

struct TileBounds {
  int a, b;          // first / last position of the tile
  int prevp, nextp;  // position just before / after it (INT32_MIN at the VCF edge)
};

__device__ __forceinline__ TileBounds tile_bounds(const ClassifyParams& P, int64_t tb, int64_t te, int64_t vbegin, int64_t vend) {
  TileBounds t;
  t.a = P.pos[tb];
  t.b = P.pos[te - 1];
  t.prevp = (tb > vbegin) ? P.pos[tb - 1] : INT32_MIN;
  t.nextp = (te < vend) ? P.pos[te] : INT32_MIN;
  return t;
}

// truth slice [lo, hi) covering positions a..b, from the coarse position index
__device__ __forceinline__ void slice_range(const TruthDev& tr, int a, int b, int& lo, int& hi) {
  uint32_t ba = (uint32_t)a >> tr.shift;
  uint32_t bb = ((uint32_t)b >> tr.shift) + 1u;
  const uint32_t lim = (uint32_t)tr.nb + 1u;
  ba = ba < lim ? ba : lim;
  bb = bb < lim ? bb : lim;
  lo = tr.tidx[ba];
  hi = tr.tidx[bb];
  if (hi < lo) hi = lo;  // only on unsorted input (results discarded)
}

struct Slice {
  uint32_t* keys;
  uint32_t* smax;  // per key: max(bin + 1) over '.'-ID single-base matches (U histogram)
  uint32_t* srf;   // per key: matched by a kept record (TP_R), one bit each
  int m;           // keys staged
  int top;         // largest power of two <= m
};

__device__ __forceinline__ void stage_slice(const TruthDev& tr, int c0, Slice& S, int lane) {
  for (int j = lane; j < S.m; j += 64) { S.keys[j] = tr.keys[c0 + j]; S.smax[j] = 0; }
  if (lane < K1_SLICE / 32) S.srf[lane] = 0;
  S.top = S.m > 0 ? 1 << (31 - __clz(S.m)) : 0;
}

__device__ __forceinline__ void slice_update(const Slice& S, int j, uint32_t fl, float q, int nb) {
  const int bin = qual_bin(q, nb);
  if ((fl & QMF_IDDOT) && bin >= 0) atomicMax(&S.smax[j], (uint32_t)(bin + 1));
  if (fl & QMF_PASS) atomicOr(&S.srf[j >> 5], 1u << (j & 31));
}

// search the 4 records of a round in the slice; returns the hit nibble.  `own_a` is
// INT32_MIN when the tile owns the run at its first position, else that position
// (the run started in an earlier tile, which owns its truth entries).
__device__ __forceinline__ uint32_t search_round(const Slice& S, const Rec4& R, int64_t i0, int64_t te, int own_a, int nb) {
  uint32_t hits = 0;
  if (S.m <= 0) return 0;
  uint32_t key[4];
  int fnd[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) key[k] = pack_key(R.p[k], R.r[k], R.a[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) fnd[k] = lds_lower_bound(S.keys, S.m, S.top, key[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t fl = (R.f >> (8 * k)) & 0xffu;
    const bool ok = (i0 + k < te) && (uint32_t)R.p[k] < (uint32_t)QM_POS_LIMIT_DEV && is_snp(R.r[k], R.a[k]) && !(fl & QMF_NOKEY);
    const int j = fnd[k];
    if (ok && j < S.m && S.keys[j] == key[k]) {
      hits |= 1u << k;
      if (R.p[k] != own_a) slice_update(S, j, fl, R.q[k], nb);
    }
  }
  return hits;
}

// records of later tiles that continue this tile's last run of equal positions
__device__ __forceinline__ void continue_run(const ClassifyParams& P, const Slice& S, int64_t te, int64_t vend, int bpos, int nb, int lane) {
  for (int64_t base = te; base < vend; base += 64) {
    const int64_t i = base + lane;
    bool cont = false;
    if (i < vend) {
      const int p = P.pos[i];
      cont = (p == bpos);
      if (cont && S.m > 0) {
        const int r_ = P.ref[i], a_ = P.alt[i];
        const uint32_t fl = P.flags[i];
        if (is_snp(r_, a_) && (uint32_t)p < (uint32_t)QM_POS_LIMIT_DEV && !(fl & QMF_NOKEY)) {
          const uint32_t key = pack_key(p, r_, a_);
          const int j = lds_lower_bound(S.keys, S.m, S.top, key);
          if (j < S.m && S.keys[j] == key) slice_update(S, j, fl, P.qual[i], nb);
        }
      }
    }
    if (ballot64(cont) != ~0ull) break;
  }
}

__device__ __forceinline__ uint32_t flush_slice(const Slice& S, uint32_t* s_hist, int lane) {
  uint32_t tpr = 0;
  for (int j = lane; j < S.m; j += 64) {
    const uint32_t mx = S.smax[j];
    if (mx) atomicAdd(&s_hist[512 + mx - 1], 1u);
    tpr += (S.srf[j >> 5] >> (j & 31)) & 1u;
  }
  return tpr;
}

__global__ __launch_bounds__(64) void k_classify(ClassifyParams P) {
  __shared__ uint32_t s_keys[2][K1_SLICE];
  __shared__ uint32_t s_max[2][K1_SLICE];
  __shared__ uint32_t s_rf[2][K1_SLICE / 32];
  __shared__ uint32_t s_hist[3 * 256];

  const int lane = (int)threadIdx.x;
  const SpanDesc sp = P.spans[blockIdx.x];
  const VcfDesc vd = P.vcfs[sp.vcf];
  const TruthDev tr = P.truths[vd.truth];
  const int64_t vbegin = vd.off;
  const int64_t vend = vd.off + vd.n;
  const int nb = P.n_bins;

  for (int i = lane; i < 3 * 256; i += 64) s_hist[i] = 0;
  uint32_t span_flags = 0;
  uint32_t acc_pass = 0, acc_tp = 0;  // wave-uniform
  uint32_t acc_tpr = 0, acc_fpr = 0;  // per lane, reduced at the end

  // ---- prologue: first round in flight, first tile's bounds and slice -------------
  int64_t tb = sp.begin;
  int64_t te = (tb + K1_TILE < sp.end) ? tb + K1_TILE : sp.end;
  Rec4 N;
  load_rec4(P, tb + lane * 4, N);
  TileBounds B = tile_bounds(P, tb, te, vbegin, vend);
  int lo, hi;
  slice_range(tr, B.a, B.b, lo, hi);
  int buf = 0;
  Slice S;
  S.keys = s_keys[0]; S.smax = s_max[0]; S.srf = s_rf[0];
  S.m = (hi - lo) < K1_SLICE ? (hi - lo) : K1_SLICE;
  stage_slice(tr, lo, S, lane);
  __syncthreads();

  int tile = sp.tile0;
  for (;;) {
    const bool has_next_tile = te < sp.end;
    const int64_t ntb = tb + K1_TILE;
    const int64_t nte = (ntb + K1_TILE < sp.end) ? ntb + K1_TILE : sp.end;
    const bool started_before = (tb > vbegin) && (B.prevp == B.a);
    const int own_a = started_before ? B.a : INT32_MIN;
    const bool owns_b = !(started_before && B.a == B.b);
    const int nrounds = (int)((te - tb + 255) >> 8);

    // ---- dense truth / sparse VCF: the keys beyond the staged chunk, as a pre-pass ----
    uint32_t prehit = 0;
    if (hi - lo > K1_SLICE && !(P.ablate & 1)) {
      for (int c0 = lo + K1_SLICE; c0 < hi; c0 += K1_SLICE) {
        __syncthreads();
        S.m = (hi - c0) < K1_SLICE ? (hi - c0) : K1_SLICE;
        stage_slice(tr, c0, S, lane);
        __syncthreads();
        for (int r = 0; r < nrounds; ++r) {
          Rec4 T;
          const int64_t i0 = tb + r * 256 + lane * 4;
          load_rec4(P, i0, T);
          prehit |= search_round(S, T, i0, te, own_a, nb) << (4 * r);
        }
        if (B.nextp == B.b && owns_b) continue_run(P, S, te, vend, B.b, nb, lane);
        __syncthreads();
        acc_tpr += flush_slice(S, s_hist, lane);
      }
      __syncthreads();
      S.m = K1_SLICE;
      stage_slice(tr, lo, S, lane);
      __syncthreads();
    }

    // ---- the tile's rounds: next round's loads first, then search + per-record work ----
    TileBounds NB = B;
    uint32_t tile_np = 0, tile_nt = 0;
    for (int r = 0; r < nrounds; ++r) {
      const Rec4 R = N;
      const int64_t rbase = tb + r * 256;
      const int64_t i0 = rbase + lane * 4;
      if (r + 1 < nrounds) {
        load_rec4(P, i0 + 256, N);
      } else if (has_next_tile) {
        load_rec4(P, ntb + lane * 4, N);
        NB = tile_bounds(P, ntb, nte, vbegin, vend);
      }
      uint32_t hits = (prehit >> (4 * r)) & 15u;
      if (!(P.ablate & 1)) hits |= search_round(S, R, i0, te, own_a, nb);

      int prevp = __shfl_up(R.p[3], 1);
      if (lane == 0) prevp = (r == 0) ? B.prevp : P.pos[rbase - 1];
      uint32_t nib_pass = 0, nib_tp = 0;
      bool unsorted = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k;
        const bool valid = i < te;
        const uint32_t fl = (R.f >> (8 * k)) & 0xffu;
        const int p = R.p[k];
        const bool okpos = (uint32_t)p < (uint32_t)QM_POS_LIMIT_DEV;
        if (valid && !okpos) span_flags |= SPANF_BADPOS;
        const bool snp = valid && okpos && is_snp(R.r[k], R.a[k]);
        const bool hit = (hits >> k) & 1u;
        const bool pass = snp && (fl & QMF_PASS);
        const bool tpkey = hit && (fl & QMF_IDDOT);
        nib_pass |= (pass ? 1u : 0u) << k;
        nib_tp |= ((pass && tpkey) ? 1u : 0u) << k;
        const int pp = (k == 0) ? prevp : R.p[k - 1];
        if (valid && p < pp) unsorted = true;
        // ROC histograms: one count per single-base record with a bin
        const int bin = qual_bin(R.q[k], nb);
        const bool counted = snp && bin >= 0;
        if (!(P.ablate & 2)) {
          const bool sat = counted && bin == nb - 1;  // real data piles up in the top bin
          const uint64_t sat_tp = ballot64(sat && tpkey);
          const uint64_t sat_fp = ballot64(sat && !tpkey);
          if (lane == 0) {
            if (sat_tp) atomicAdd(&s_hist[nb - 1], (uint32_t)popc64(sat_tp));
            if (sat_fp) atomicAdd(&s_hist[256 + nb - 1], (uint32_t)popc64(sat_fp));
          }
          if (counted && !sat) atomicAdd(&s_hist[(tpkey ? 0 : 256) + bin], 1u);
        }
        // R path: distinct kept keys outside the truth set.  Only a record whose
        // predecessor has the same position can be a repeat; walk that run backwards
        // (<= 16 distinct single-base keys per position bound the total walk per run).
        if (pass && !hit && !(P.ablate & 8)) {
          bool first = true;
          if (p == pp) {
            const uint32_t nk = fl & QMF_NOKEY;
            for (int64_t j = i - 1; j >= vbegin; --j) {
              if (P.pos[j] != p) break;
              const uint32_t fj = P.flags[j];
              if ((fj & QMF_PASS) && (fj & QMF_NOKEY) == nk && P.ref[j] == R.r[k] && P.alt[j] == R.a[k]) { first = false; break; }
            }
          }
          acc_fpr += first ? 1u : 0u;
        }
      }
      if (unsorted) span_flags |= SPANF_UNSORTED;

      // ---- wave ballots -> natural-order 64-bit mask words ---------------------------
      if (!(P.ablate & 4)) {
        const int sel = lane & 3;
        const int sh = lane >> 2;
        uint64_t wp[4], wt[4];
        {
          const uint64_t b0 = ballot64(nib_pass & 1u), b1 = ballot64(nib_pass & 2u);
          const uint64_t b2 = ballot64(nib_pass & 4u), b3 = ballot64(nib_pass & 8u);
          const uint64_t mine = sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : b3;
#pragma unroll
          for (int w = 0; w < 4; ++w) wp[w] = ballot64((mine >> (16 * w + sh)) & 1ull);
        }
        {
          const uint64_t b0 = ballot64(nib_tp & 1u), b1 = ballot64(nib_tp & 2u);
          const uint64_t b2 = ballot64(nib_tp & 4u), b3 = ballot64(nib_tp & 8u);
          const uint64_t mine = sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : b3;
#pragma unroll
          for (int w = 0; w < 4; ++w) wt[w] = ballot64((mine >> (16 * w + sh)) & 1ull);
        }
        if (lane < 4) {
          const uint64_t vp = lane == 0 ? wp[0] : lane == 1 ? wp[1] : lane == 2 ? wp[2] : wp[3];
          const uint64_t vt = lane == 0 ? wt[0] : lane == 1 ? wt[1] : lane == 2 ? wt[2] : wt[3];
          P.mask_pass[(rbase >> 6) + lane] = vp;
          P.mask_tp[(rbase >> 6) + lane] = vt;
        }
        tile_np += (uint32_t)(popc64(wp[0]) + popc64(wp[1]) + popc64(wp[2]) + popc64(wp[3]));
        tile_nt += (uint32_t)(popc64(wt[0]) + popc64(wt[1]) + popc64(wt[2]) + popc64(wt[3]));
      }
    }

    // ---- tile epilogue: run continuation, per-truth-entry state -> histogram, counts ----
    if (!(P.ablate & 1)) {
      if (B.nextp == B.b && owns_b) continue_run(P, S, te, vend, B.b, nb, lane);
      __syncthreads();
      acc_tpr += flush_slice(S, s_hist, lane);
    }
    if (lane == 0) {
      P.tile_tp[tile] = tile_nt;
      P.tile_fp[tile] = tile_np - tile_nt;
    }
    acc_pass += tile_np;
    acc_tp += tile_nt;

    if (!has_next_tile) break;
    // ---- stage the next tile's slice into the other LDS half ----------------------------
    B = NB;
    tb = ntb;
    te = nte;
    ++tile;
    slice_range(tr, B.a, B.b, lo, hi);
    buf ^= 1;
    S.keys = s_keys[buf]; S.smax = s_max[buf]; S.srf = s_rf[buf];
    S.m = (hi - lo) < K1_SLICE ? (hi - lo) : K1_SLICE;
    stage_slice(tr, lo, S, lane);
    __syncthreads();
  }

  // ---- span epilogue -------------------------------------------------------------------
  for (int o = 32; o > 0; o >>= 1) {
    acc_tpr += __shfl_xor(acc_tpr, o);
    acc_fpr += __shfl_xor(acc_fpr, o);
  }
  const uint64_t any_uns = ballot64(span_flags & SPANF_UNSORTED);
  const uint64_t any_bad = ballot64(span_flags & SPANF_BADPOS);
  __syncthreads();
  uint32_t* oh = P.span_hist + (size_t)blockIdx.x * (3 * 256);
  for (int i = lane; i < 3 * 256; i += 64) oh[i] = s_hist[i];
  if (lane == 0) {
    uint32_t* sc = P.span_scal + (size_t)blockIdx.x * 8;
    sc[0] = acc_pass; sc[1] = acc_tp; sc[2] = acc_pass - acc_tp; sc[3] = acc_tpr; sc[4] = acc_fpr;
    sc[5] = (any_uns ? SPANF_UNSORTED : 0u) | (any_bad ? SPANF_BADPOS : 0u);
    sc[6] = 0; sc[7] = 0;
  }
}

// ---------------------------------------------------------------------------
// k_finalize: one workgroup (256 threads) per VCF
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(FinalizeParams P) {
  __shared__ uint32_t s_h[3][256];
  __shared__ uint32_t s_scan[256];
  __shared__ uint32_t s_carry[2];
  const int v = (int)blockIdx.x;
  const int tid = (int)threadIdx.x;
  const VcfDesc vd = P.vcfs[v];
  const int nb = P.n_bins;

  // flags first: an unsorted VCF's numbers are discarded (redone by the sort path)
  uint32_t fl = 0;
  for (int s = tid; s < vd.nspans; s += 256) fl |= P.span_scal[(size_t)(vd.span0 + s) * 8 + 5];
  const bool unsorted = __syncthreads_or((int)(fl & SPANF_UNSORTED)) != 0;
  const bool badpos = __syncthreads_or((int)(fl & SPANF_BADPOS)) != 0;

  // sum span histograms (thread = bin)
  uint32_t h0 = 0, h1 = 0, h2 = 0;
  for (int s = 0; s < vd.nspans; ++s) {
    const uint32_t* sh = P.span_hist + (size_t)(vd.span0 + s) * (3 * 256);
    h0 += sh[tid]; h1 += sh[256 + tid]; h2 += sh[512 + tid];
  }
  s_h[0][tid] = h0; s_h[1][tid] = h1; s_h[2][tid] = h2;
  __syncthreads();
  if (tid < nb) {
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    for (int b = tid; b < nb; ++b) { c0 += s_h[0][b]; c1 += s_h[1][b]; c2 += s_h[2][b]; }
    uint64_t* roc = P.roc + (size_t)v * 3 * nb;
    roc[tid] = c0; roc[nb + tid] = c1; roc[2 * nb + tid] = c2;
    if (P.global_acc && !unsorted) {
      unsigned long long* g = reinterpret_cast<unsigned long long*>(P.global_acc) + (size_t)vd.truth * 3 * nb;
      if (c0) atomicAdd(&g[tid], (unsigned long long)c0);
      if (c1) atomicAdd(&g[nb + tid], (unsigned long long)c1);
      if (c2) atomicAdd(&g[2 * nb + tid], (unsigned long long)c2);
    }
  }
  // scalars
  if (tid < 8) {
    uint64_t acc = 0;
    for (int s = 0; s < vd.nspans; ++s) acc += P.span_scal[(size_t)(vd.span0 + s) * 8 + tid];
    int64_t* sc = P.scalars + (size_t)v * 8;
    if (tid < 5) sc[tid] = (int64_t)acc;
    else if (tid == 5) { sc[5] = unsorted ? 0 : 1; P.vcf_flags[v] = (unsorted ? SPANF_UNSORTED : 0u) | (badpos ? SPANF_BADPOS : 0u); }
    else if (tid == 6) sc[6] = vd.n;
    else sc[7] = P.truths[vd.truth].n;
  }
  // exclusive scan of the tile counts (TP then FP) over the VCF's tiles
  for (int which = 0; which < 2; ++which) {
    const uint32_t* in = which ? P.tile_fp : P.tile_tp;
    uint32_t* out = which ? P.tile_fp_off : P.tile_tp_off;
    if (tid == 0) s_carry[which] = 0;
    __syncthreads();
    for (int base = 0; base < vd.ntiles; base += 256) {
      const int t = base + tid;
      const uint32_t x = t < vd.ntiles ? in[vd.tile0 + t] : 0u;
      s_scan[tid] = x;
      __syncthreads();
      for (int d = 1; d < 256; d <<= 1) {
        const uint32_t y = tid >= d ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += y;
        __syncthreads();
      }
      const uint32_t incl = s_scan[tid];
      const uint32_t carry = s_carry[which];
      if (t < vd.ntiles) out[vd.tile0 + t] = carry + incl - x;
      __syncthreads();
      if (tid == 255) s_carry[which] = carry + incl;
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// k_compact: ballot masks -> compacted line-index lists.  One wave per 512-record
// tile (4 tiles per workgroup).  idx region of VCF v (vd.n entries at vd.off):
// TP line indices ascending from the front, FP ascending, ending at the back.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(CompactParams P, int n_tiles) {
  const int lane = (int)(threadIdx.x & 63);
  const int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int v = P.tile_vcf[tile];
  const VcfDesc vd = P.vcfs[v];
  const int64_t tb = vd.off + (int64_t)(tile - vd.tile0) * K1_TILE;
  const int64_t vend = vd.off + vd.n;
  const int64_t te = tb + K1_TILE < vend ? tb + K1_TILE : vend;
  const int nwords = (int)((te - tb + 63) >> 6);
  const uint64_t* mp = P.mask_pass + (tb >> 6);
  const uint64_t* mt = P.mask_tp + (tb >> 6);
  // total FP lines of the VCF = offset of its last tile + that tile's count
  const int lastt = vd.tile0 + vd.ntiles - 1;
  const int64_t fp_total = (int64_t)P.tile_fp_off[lastt] + P.tile_fp[lastt];
  int32_t* out = P.idx + vd.off;
  int64_t tp_at = P.tile_tp_off[tile];
  int64_t fp_at = (vd.n - fp_total) + P.tile_fp_off[tile];
  const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
  const int32_t rel0 = (int32_t)(tb - vd.off) + lane;
#pragma unroll
  for (int w = 0; w < K1_TILE / 64; ++w) {
    if (w < nwords) {  // wave-uniform
      const uint64_t wt = mt[w];
      const uint64_t wf = mp[w] & ~wt;
      if ((wt >> lane) & 1ull) out[tp_at + __popcll(wt & below)] = rel0 + w * 64;
      if ((wf >> lane) & 1ull) out[fp_at + __popcll(wf & below)] = rel0 + w * 64;
      tp_at += __popcll(wt);
      fp_at += __popcll(wf);
    }
  }
}

// ---------------------------------------------------------------------------
// expand class masks to one byte per record (host-side consumers)
// ---------------------------------------------------------------------------
__global__ void k_masks_to_cls(const uint64_t* mp, const uint64_t* mt, int64_t off, int64_t n, uint8_t* cls) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g = off + i;
  const uint32_t p = (uint32_t)((mp[g >> 6] >> (g & 63)) & 1ull);
  const uint32_t t = (uint32_t)((mt[g >> 6] >> (g & 63)) & 1ull);
  cls[i] = (uint8_t)(p | (t << 1));
}

// ---------------------------------------------------------------------------
// synthetic workload (DESIGN.md "Synthetic generator")
// ---------------------------------------------------------------------------
__global__ void k_synth(SynthParams S) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // slot in VCF
  const int v = (int)blockIdx.y;
  const VcfDesc vd = S.vcfs[v];
  if (i >= vd.n) return;
  const uint64_t seed = S.seed + (uint64_t)v;
  int64_t src = i;
  if (S.shuffled) src = (int64_t)(((unsigned __int128)(uint64_t)i * S.perm_a + S.perm_b) % (uint64_t)vd.n);
  int32_t p, r, a;
  float q;
  uint8_t f;
  synth_record(S.genome_len, vd.n, S.truth_n, S.truth_seed, seed, src, &p, &r, &a, &q, &f);
  const int64_t g = vd.off + i;
  S.pos[g] = p; S.ref[g] = r; S.alt[g] = a; S.qual[g] = q; S.flags[g] = f;
}

// ---------------------------------------------------------------------------
// radix sort path (unsorted VCFs): stable LSD passes over 8-bit digits of
// key = pos (28 bits), payload = original record index.
// ---------------------------------------------------------------------------
__global__ void k_sort_init(const int32_t* pos, int64_t off, int64_t n, uint32_t* keys, uint32_t* vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = (uint32_t)pos[off + i];
  vals[i] = (uint32_t)i;
}

// per-tile digit histogram: hist[digit * ntiles + tile]
__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t* keys, int64_t n, int shift, uint32_t* hist, int ntiles) {
  __shared__ uint32_t s[256];
  const int tid = (int)threadIdx.x;
  s[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  for (int k = 0; k < SORT_TILE / 256; ++k) {
    const int64_t i = base + k * 256 + tid;
    if (i < n) atomicAdd(&s[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)tid * ntiles + blockIdx.x] = s[tid];
}

// exclusive scan over the digit-major histogram (single workgroup, sequential chunks)
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t* hist, int64_t total) {
  __shared__ uint32_t s_scan[256];
  __shared__ uint32_t s_carry;
  const int tid = (int)threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < total; base += 256) {
    const int64_t t = base + tid;
    const uint32_t x = t < total ? hist[t] : 0u;
    s_scan[tid] = x;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const uint32_t y = tid >= d ? s_scan[tid - d] : 0u;
      __syncthreads();
      s_scan[tid] += y;
      __syncthreads();
    }
    const uint32_t incl = s_scan[tid];
    const uint32_t carry = s_carry;
    if (t < total) hist[t] = carry + incl - x;
    __syncthreads();
    if (tid == 255) s_carry = carry + incl;
    __syncthreads();
  }
}

// stable scatter: wave w of the tile owns SORT_TILE/4 consecutive keys and walks
// them 64 at a time; rank inside a wave step by an 8-ballot multisplit.
__global__ __launch_bounds__(256) void k_sort_scatter(const uint32_t* keys, const uint32_t* vals, int64_t n, int shift,
                                                      const uint32_t* hist, int ntiles, uint32_t* okeys, uint32_t* ovals) {
  __shared__ uint32_t s_cnt[4][256];   // running count of digit d in wave w
  __shared__ uint32_t s_base[4][256];  // start of wave w's digit-d block in the output
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < 4 * 256; i += 256) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * (SORT_TILE / 4);
  constexpr int STEPS = SORT_TILE / 4 / 64;
  uint32_t kk[STEPS], vv[STEPS], rk[STEPS];
  // pass 1: ranks within the wave's chunk
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int64_t i = wbase + s * 64 + lane;
    const bool valid = i < n;
    kk[s] = valid ? keys[i] : 0xffffffffu;
    vv[s] = valid ? vals[i] : 0u;
    const uint32_t d = (kk[s] >> shift) & 255u;
    uint64_t peers = ballot64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t m = ballot64((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t r = (uint32_t)__popcll(peers & below);
    uint32_t basec = 0;
    if (valid) basec = s_cnt[wave][d];   // every peer reads before the leader writes (same wave, lockstep)
    rk[s] = basec + r;
    if (valid && r == 0) s_cnt[wave][d] = basec + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  // digit d (thread d): global offset of this tile + exclusive scan over the 4 waves
  {
    const uint32_t g = hist[(size_t)tid * ntiles + blockIdx.x];
    uint32_t run = g;
#pragma unroll
    for (int w = 0; w < 4; ++w) { s_base[w][tid] = run; run += s_cnt[w][tid]; }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int64_t i = wbase + s * 64 + lane;
    if (i < n) {
      const uint32_t d = (kk[s] >> shift) & 255u;
      const uint32_t o = s_base[wave][d] + rk[s];
      okeys[o] = kk[s];
      ovals[o] = vv[s];
    }
  }
}

// gather the columns of one VCF through the sorted permutation into a scratch VCF
__global__ void k_sort_gather(const int32_t* pos, const int32_t* ref, const int32_t* alt, const float* qual,
                              const uint8_t* flags, int64_t src_off, const uint32_t* perm, int64_t n, int32_t* opos,
                              int32_t* oref, int32_t* oalt, float* oqual, uint8_t* oflags, int64_t dst_off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = src_off + perm[i];
  opos[dst_off + i] = pos[s]; oref[dst_off + i] = ref[s]; oalt[dst_off + i] = alt[s];
  oqual[dst_off + i] = qual[s]; oflags[dst_off + i] = flags[s];
}

// scatter class bits of the sorted scratch VCF back to input order (byte per record)
__global__ void k_sort_scatter_cls(const uint64_t* mp, const uint64_t* mt, int64_t src_off, const uint32_t* perm, int64_t n,
                                   uint8_t* cls) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g = src_off + i;
  const uint32_t p = (uint32_t)((mp[g >> 6] >> (g & 63)) & 1ull);
  const uint32_t t = (uint32_t)((mt[g >> 6] >> (g & 63)) & 1ull);
  cls[perm[i]] = (uint8_t)(p | (t << 1));
}

// byte-per-record classes -> mask words + per-tile counts (input order), one wave per 64 records
__global__ __launch_bounds__(256) void k_cls_to_masks(const uint8_t* cls, int64_t off, int64_t n, uint64_t* mp, uint64_t* mt,
                                                      uint32_t* tile_tp, uint32_t* tile_fp, int tile0) {
  __shared__ uint32_t s_c[2];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  if (tid < 2) s_c[tid] = 0;
  __syncthreads();
  const int64_t tb = (int64_t)blockIdx.x * K1_TILE;
  for (int w = tid >> 6; w < K1_TILE / 64; w += 4) {
    const int64_t i = tb + w * 64 + lane;
    const uint8_t c = i < n ? cls[i] : 0;
    const uint64_t bp = ballot64(c & 1u), bt = ballot64(c & 2u);
    if (tb + w * 64 < n && lane == 0) {
      mp[((off + tb) >> 6) + w] = bp;
      mt[((off + tb) >> 6) + w] = bt;
      atomicAdd(&s_c[0], (uint32_t)__popcll(bt));
      atomicAdd(&s_c[1], (uint32_t)__popcll(bp & ~bt));
    }
  }
  __syncthreads();
  if (tid == 0) { tile_tp[tile0 + blockIdx.x] = s_c[0]; tile_fp[tile0 + blockIdx.x] = s_c[1]; }
}

// ---------------------------------------------------------------------------
// FP overlap (A7): keys tagged with a set bit, sorted by key, grouped.
// ---------------------------------------------------------------------------
__global__ void k_overlap_pack(const int32_t* pos, const int32_t* ref, const int32_t* alt, const int32_t* set_of, int64_t n,
                               uint32_t* keys, uint32_t* vals, uint32_t* bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = is_snp(ref[i], alt[i]) && (uint32_t)pos[i] < (uint32_t)QM_POS_LIMIT_DEV;
  if (!ok) atomicOr(bad, 1u);
  keys[i] = ok ? pack_key(pos[i], ref[i], alt[i]) : 0xffffffffu;
  vals[i] = ok ? (1u << set_of[i]) : 0u;
}
__global__ void k_overlap_count(const uint32_t* keys, const uint32_t* vals, int64_t n, unsigned long long* regions) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keys[i] == 0xffffffffu) return;
  if (i > 0 && keys[i - 1] == keys[i]) return;  // not the head of its group
  uint32_t m = 0;
  for (int64_t j = i; j < n && keys[j] == keys[i]; ++j) m |= vals[j];
  atomicAdd(&regions[m], 1ull);
}

// ---------------------------------------------------------------------------
// launchers (called from qmvt_api.cpp through qmvt_dev.h)
// ---------------------------------------------------------------------------
void launch_classify(const ClassifyParams& P, int n_spans, hipStream_t st) {
  if (n_spans > 0) hipLaunchKernelGGL(k_classify, dim3(n_spans), dim3(64), 0, st, P);
}
void launch_finalize(const FinalizeParams& P, int n_vcf, hipStream_t st) {
  if (n_vcf > 0) hipLaunchKernelGGL(k_finalize, dim3(n_vcf), dim3(256), 0, st, P);
}
void launch_compact(const CompactParams& P, int n_tiles, hipStream_t st) {
  if (n_tiles > 0) hipLaunchKernelGGL(k_compact, dim3((n_tiles + 3) / 4), dim3(256), 0, st, P, n_tiles);
}
void launch_masks_to_cls(const uint64_t* mp, const uint64_t* mt, int64_t off, int64_t n, uint8_t* cls, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_masks_to_cls, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mp, mt, off, n, cls);
}
void launch_synth(const SynthParams& S, int n_vcf, int64_t max_n, hipStream_t st) {
  if (n_vcf > 0 && max_n > 0)
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((max_n + 255) / 256), (unsigned)n_vcf), dim3(256), 0, st, S);
}
void launch_sort_init(const int32_t* pos, int64_t off, int64_t n, uint32_t* keys, uint32_t* vals, hipStream_t st) {
  hipLaunchKernelGGL(k_sort_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, off, n, keys, vals);
}
void launch_sort_pass(const uint32_t* keys, const uint32_t* vals, int64_t n, int shift, uint32_t* hist, uint32_t* okeys,
                      uint32_t* ovals, hipStream_t st) {
  const int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
  hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(256), 0, st, keys, n, shift, hist, ntiles);
  hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(256), 0, st, hist, (int64_t)ntiles * 256);
  hipLaunchKernelGGL(k_sort_scatter, dim3(ntiles), dim3(256), 0, st, keys, vals, n, shift, hist, ntiles, okeys, ovals);
}
void launch_sort_gather(const int32_t* pos, const int32_t* ref, const int32_t* alt, const float* qual, const uint8_t* flags,
                        int64_t src_off, const uint32_t* perm, int64_t n, int32_t* opos, int32_t* oref, int32_t* oalt,
                        float* oqual, uint8_t* oflags, int64_t dst_off, hipStream_t st) {
  hipLaunchKernelGGL(k_sort_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, ref, alt, qual, flags, src_off,
                     perm, n, opos, oref, oalt, oqual, oflags, dst_off);
}
void launch_sort_scatter_cls(const uint64_t* mp, const uint64_t* mt, int64_t src_off, const uint32_t* perm, int64_t n,
                             uint8_t* cls, hipStream_t st) {
  hipLaunchKernelGGL(k_sort_scatter_cls, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mp, mt, src_off, perm, n, cls);
}
void launch_cls_to_masks(const uint8_t* cls, int64_t off, int64_t n, uint64_t* mp, uint64_t* mt, uint32_t* tile_tp,
                         uint32_t* tile_fp, int tile0, hipStream_t st) {
  const int ntiles = (int)((n + K1_TILE - 1) / K1_TILE);
  if (ntiles > 0)
    hipLaunchKernelGGL(k_cls_to_masks, dim3(ntiles), dim3(256), 0, st, cls, off, n, mp, mt, tile_tp, tile_fp, tile0);
}
void launch_overlap_pack(const int32_t* pos, const int32_t* ref, const int32_t* alt, const int32_t* set_of, int64_t n,
                         uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t st) {
  hipLaunchKernelGGL(k_overlap_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, ref, alt, set_of, n, keys, vals, bad);
}
void launch_overlap_count(const uint32_t* keys, const uint32_t* vals, int64_t n, unsigned long long* regions, hipStream_t st) {
  hipLaunchKernelGGL(k_overlap_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, vals, n, regions);
}

}  // namespace qm
