// qmvt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the variant-truth engine.
//
// Pure integer / index work, HBM-bound: no MFMA.  wave = 64 lanes everywhere.
//
//   k_classify   one wave per SPAN (<= SPAN_TILES consecutive 1024-record tiles of one VCF),
//                no barriers.  Streams pos/ref/alt/qual (non-temporal dwordx4 per lane) +
//                flags (dword per lane) one round of 256 records ahead, stages each tile's
//                slice of the sorted truth keys in LDS (the next tile's slice waits in
//                registers) and merge-joins from the sparse side: truth keys bisect the
//                staged record keys.  Emits natural-order class masks (kept / TP, 1 bit per
//                record each; DPP OR over 8 lanes x 4 records), per-tile TP/FP line counts,
//                per-span QUAL-bin histograms (TP, FP, distinct truth keys; u16 pairs) and
//                scalar counters.  Instantiations: <PACKED> for the sorted (key, info)
//                pairs of the radix-sort path, <EXT> for the allele-extended mode.
//   k_finalize   one workgroup per VCF: span histograms -> ROC suffix sums, scalars,
//                exclusive scan of the tile counts, per-truth-set sums.
//   k_compact    one wave per span: expands the masks into the compacted TP / FP
//                line-index lists (popc prefix ranks, LDS rings, 16-byte NT stores).
//   k_sort_*     batched, segmented LSD radix sort (wave multisplit, XCD-aware tile order)
//                for unsorted VCFs; the first pass packs straight from the columns and
//                writes the kept mask in input order; only TP bits travel back.
//   k_synth      on-device generator of the BASELINE.json config-3/4/5 workloads.
//
// Reference stages replaced (file:line in /root/reference):
//   fgrep -wf / -wvf            program/extract_TP_FP_SNPs.py:50-57
//   R intersect/setdiff/length  scripts/caller_performance_compare.R:94-96
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qmvt_dev.h"

namespace qm {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
// streaming accesses (touched once per pass): keep them out of the way of what L2 should hold
template <typename T> __device__ __forceinline__ T ntl(const T* p) {
  return __builtin_nontemporal_load(p);
}
template <typename T> __device__ __forceinline__ void nts(T* p, T v) {
  __builtin_nontemporal_store(v, p);
}
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
//   round = 256 records, 4 consecutive records per lane.  The NEXT round's loads are
//           issued before the current round is touched (register double buffer).
//           Each record is packed to a 32-bit key (pos << 4 | ref << 2 | alt) and a
//           16-bit info (bin + 1, flags, live) and staged in LDS.  The join then runs
//           from the sparse side: every truth key of the round's position range (one
//           per lane) binary-searches the 256 staged record keys, marks its matches
//           in a 256-bit hit set and folds them into its per-truth-entry state.
//           The per-record pass works on 4-bit nibbles in registers (kept, TP, ...);
//           a three-step DPP OR turns 8 lanes' nibbles into a natural-order mask word.
//   tile  = K1_ROUNDS rounds = the unit that owns a slice of the sorted truth keys
//           in LDS (tile t+1's slice is fetched into registers while t runs and staged once t's state is flushed),
//           the per-truth-entry state for U(t)/TP_R, and one TP/FP line count.
// Two input formats: the five SoA columns of include/qmvt.h (PACKED = false), and the
// (key, info) pairs the radix-sort path produces for unsorted VCFs (PACKED = true).
// All record indices are 32-bit and relative to the VCF (n < 2^31).
// ---------------------------------------------------------------------------

// info of a record.  The low 16 bits are what a matching truth key needs (they are what the join sees in LDS):
constexpr uint32_t I_BIN1 = 0x1ffu;      // bin + 1; 0 = passes no threshold or is not a live (single-base, in-range) record
constexpr uint32_t I_PASS = 1u << 9;     // flags bit0
constexpr uint32_t I_IDDOT = 1u << 10;   // flags bit1
constexpr uint32_t I_NOKEY = 1u << 11;   // flags bit2
constexpr uint32_t I_LIVE = 1u << 12;    // position in range, single-base alleles
constexpr uint32_t I_BADPOS = 1u << 13;  // position outside [0, 2^28)
// The per-record pass reads three bits per record, kept four bits apart in the high half (and nothing else there), so
// that `(inf >> 16) << k` OR-ed over the lane's four records yields the three nibbles at once:
constexpr uint32_t I_KEPT = 1u << 16;    // live and PASS: the line is in <x>.filtered.vcf
constexpr uint32_t I_IDDOT4 = 1u << 20;  // copy of I_IDDOT
constexpr uint32_t I_TPLINE = 1u << 24;  // flags bit3: the host found the line selected by fgrep (a TP line whatever its key says)

// key and info of one record from its columns (qmvt_dev.h: the radix-sort path uses the same)
// EXT (allele-extended batches, include/qmvt.h "allele codes"): any valid allele code is live; the
// key's nibble is ref << 2 | alt for single bases and a hash of the two codes otherwise, so key
// equality is necessary and (key, ref, alt) equality is the match.
template <bool EXT = false>
__device__ __forceinline__ void pack_record(int p, int r, int a, float q, uint32_t fl, int nb, uint32_t& key, uint32_t& inf) {
  const bool okpos = (uint32_t)p < (uint32_t)QM_POS_LIMIT_DEV;
  bool live;
  uint32_t nib;
  if (EXT) {
    live = okpos && allele_valid(r) && allele_valid(a);
    nib = allele_nib(r, a);
  } else {
    live = okpos & ((uint32_t)(r | a) < 4u);
    nib = ((uint32_t)r << 2) | (uint32_t)a;
  }
  key = ((uint32_t)p << 4) | (live ? nib : 0u);
  inf = (live ? (uint32_t)(qual_bin(q, nb) + 1) : 0u) | ((fl & 7u) << 9) | (live ? I_LIVE : 0u) | (okpos ? 0u : I_BADPOS) |
        ((live && (fl & QMF_PASS)) ? I_KEPT : 0u) | ((fl & QMF_IDDOT) ? I_IDDOT4 : 0u) | ((live && (fl & QMF_TPLINE)) ? I_TPLINE : 0u);
}

// The flag byte's part of the info word: PASS / IDDOT / NOKEY to bits 9..11, PASS, IDDOT and TPLINE again to the nibble
// lanes of the high half, I_LIVE on top.  k_classify keeps the sixteen values in LDS (one read instead of eight VALU).
__host__ __device__ inline uint32_t flag_info(uint32_t f) {
  return ((f & 7u) << 9) | I_LIVE | ((f & QMF_PASS) ? I_KEPT : 0u) | ((f & QMF_IDDOT) ? I_IDDOT4 : 0u) | ((f & QMF_TPLINE) ? I_TPLINE : 0u);
}

// The same key and info as pack_record<false> for a record that is in range (out-of-range positions are collected by
// the caller in one OR over the round), written for the VALU-bound main loop: 15 vector instructions and one LDS
// read per record.  bin + 1 = min(floor(max(q, -1)), n_bins - 1) + 1 (NaN -> 0); flut = the flag_info table in LDS.
__device__ __forceinline__ void pack_record_fast(int p, int r, int a, float q, uint32_t f4x4 /* flag byte index * 4 */, float nbm1f,
                                                 const uint32_t* flut, uint32_t& key, uint32_t& inf) {
  const uint32_t t = (uint32_t)r | (uint32_t)a;
  const bool live = (((uint32_t)p >> 26) | t) < 4u;          // 0 <= p < 2^28 and both alleles single bases
  const uint32_t nib = ((uint32_t)r << 2) | (uint32_t)a;
  key = ((uint32_t)p << 4) | (live ? nib : 0u);
  const float c = fminf(fmaxf(q, -1.0f), nbm1f);            // NaN -> -1
  const uint32_t b1 = (uint32_t)((int)floorf(c) + 1);
  const uint32_t g = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(flut) + f4x4);
  inf = live ? (b1 | g) : 0u;
}

// The allele-extended counterpart (any valid allele code takes part, include/qmvt.h): the same key and info as
// pack_record<true> for a record whose position is in range.  A code is valid iff it is < 4 or in [2^27, 2^31): by its number
// of leading zeros -- 30 and more, or 1..4 -- which one shift of a constant turns into a bit (c | 1 has c's count for c >= 2 and
// stands for 0 and 1 alike); the nibble is allele_nib's branch-free fold.
__device__ __forceinline__ void pack_record_fast_ext(int p, int r, int a, float q, uint32_t f4x4, float nbm1f, const uint32_t* flut,
                                                     uint32_t& key, uint32_t& inf) {
  const uint32_t ur = (uint32_t)r, ua = (uint32_t)a;
  constexpr uint32_t VALID_BY_CLZ = 0xC000001Eu;            // bit z: a code with z leading zeros is an allele code
  const uint32_t ok = (VALID_BY_CLZ >> __builtin_clz(ur | 1u)) & (VALID_BY_CLZ >> __builtin_clz(ua | 1u)) & 1u;
  const bool live = ((uint32_t)p >> 28) < ok;              // 0 <= p < 2^28 and both codes valid
  const uint32_t nib = ((ur << 2) ^ ua ^ ((ur ^ ua) >> 4)) & 15u;
  key = ((uint32_t)p << 4) | (live ? nib : 0u);
  const float c = fminf(fmaxf(q, -1.0f), nbm1f);            // NaN -> -1
  const uint32_t b1 = (uint32_t)((int)floorf(c) + 1);
  const uint32_t g = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(flut) + f4x4);
  inf = live ? (b1 | g) : 0u;
}

struct Cols {  // bases of one VCF: the five columns, or the packed pair
  const int32_t* pos;
  const int32_t* ref;
  const int32_t* alt;
  const float* qual;
  const uint8_t* flags;
  const uint32_t* pkey;
  const uint32_t* pinf;
};

template <bool PACKED> struct Raw4;   // one round's loads, still in flight
template <> struct Raw4<false> { int4 p, r, a; float4 q; uint32_t f; };
template <> struct Raw4<true> { uint4 k, i; int4 r, a; };   // r / a: allele-extended batches only

template <bool EXT>
__device__ __forceinline__ void load_raw(const Cols& C, int idx, Raw4<false>& R) {
  // streaming data, read once: non-temporal loads keep it from displacing the truth keys and the span
  // outputs in L2 (same-box A/B: +3.5 % on this kernel, k_finalize 10 % faster)
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4i vp = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(C.pos + idx));
  const v4i vr = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(C.ref + idx));
  const v4i va = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(C.alt + idx));
  const v4f vq = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(C.qual + idx));
  R.p = make_int4(vp.x, vp.y, vp.z, vp.w);
  R.r = make_int4(vr.x, vr.y, vr.z, vr.w);
  R.a = make_int4(va.x, va.y, va.z, va.w);
  R.q = make_float4(vq.x, vq.y, vq.z, vq.w);
  R.f = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(C.flags + idx));
}
template <bool EXT>
__device__ __forceinline__ void load_raw(const Cols& C, int idx, Raw4<true>& R) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u vk = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(C.pkey + idx));
  const v4u vi = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(C.pinf + idx));
  R.k = make_uint4(vk.x, vk.y, vk.z, vk.w);
  R.i = make_uint4(vi.x, vi.y, vi.z, vi.w);
  if (EXT) {   // the sorted copies of the allele columns (k_sort_gather_alleles)
    R.r = *reinterpret_cast<const int4*>(C.ref + idx);
    R.a = *reinterpret_cast<const int4*>(C.alt + idx);
  }
}

struct In4 {   // the lane's 4 records of the round, packed
  uint32_t key[4], inf[4];
  int32_t r[4], a[4];   // EXT only: the allele codes behind the key's nibble
  uint32_t posor;       // column input, !EXT: OR of the four positions (anything at or above bit 28 = out of range)
};

template <bool EXT>
__device__ __forceinline__ void unpack_raw(const Raw4<false>& R, int nb, In4& X, const uint32_t* flut) {
  X.posor = 0u;
  if (!EXT) {
    const float nbm1f = (float)(nb - 1);
    pack_record_fast(R.p.x, R.r.x, R.a.x, R.q.x, (R.f << 2) & 0x3cu, nbm1f, flut, X.key[0], X.inf[0]);
    pack_record_fast(R.p.y, R.r.y, R.a.y, R.q.y, (R.f >> 6) & 0x3cu, nbm1f, flut, X.key[1], X.inf[1]);
    pack_record_fast(R.p.z, R.r.z, R.a.z, R.q.z, (R.f >> 14) & 0x3cu, nbm1f, flut, X.key[2], X.inf[2]);
    pack_record_fast(R.p.w, R.r.w, R.a.w, R.q.w, (R.f >> 22) & 0x3cu, nbm1f, flut, X.key[3], X.inf[3]);
    X.posor = (uint32_t)R.p.x | (uint32_t)R.p.y | (uint32_t)R.p.z | (uint32_t)R.p.w;
    return;
  }
  if (EXT) {
    const float nbm1f = (float)(nb - 1);
    pack_record_fast_ext(R.p.x, R.r.x, R.a.x, R.q.x, (R.f << 2) & 0x3cu, nbm1f, flut, X.key[0], X.inf[0]);
    pack_record_fast_ext(R.p.y, R.r.y, R.a.y, R.q.y, (R.f >> 6) & 0x3cu, nbm1f, flut, X.key[1], X.inf[1]);
    pack_record_fast_ext(R.p.z, R.r.z, R.a.z, R.q.z, (R.f >> 14) & 0x3cu, nbm1f, flut, X.key[2], X.inf[2]);
    pack_record_fast_ext(R.p.w, R.r.w, R.a.w, R.q.w, (R.f >> 22) & 0x3cu, nbm1f, flut, X.key[3], X.inf[3]);
    X.posor = (uint32_t)R.p.x | (uint32_t)R.p.y | (uint32_t)R.p.z | (uint32_t)R.p.w;   // anything at or above bit 28: out of range (SPANF_BADPOS)
  } else
  {
  pack_record<EXT>(R.p.x, R.r.x, R.a.x, R.q.x, R.f, nb, X.key[0], X.inf[0]);
  pack_record<EXT>(R.p.y, R.r.y, R.a.y, R.q.y, R.f >> 8, nb, X.key[1], X.inf[1]);
  pack_record<EXT>(R.p.z, R.r.z, R.a.z, R.q.z, R.f >> 16, nb, X.key[2], X.inf[2]);
  pack_record<EXT>(R.p.w, R.r.w, R.a.w, R.q.w, R.f >> 24, nb, X.key[3], X.inf[3]);
  }
  if (EXT) {
    X.r[0] = R.r.x; X.r[1] = R.r.y; X.r[2] = R.r.z; X.r[3] = R.r.w;
    X.a[0] = R.a.x; X.a[1] = R.a.y; X.a[2] = R.a.z; X.a[3] = R.a.w;
  }
}
template <bool EXT>
__device__ __forceinline__ void unpack_raw(const Raw4<true>& R, int, In4& X, const uint32_t*) {
  X.posor = 0u;
  if (EXT) {
    X.r[0] = R.r.x; X.r[1] = R.r.y; X.r[2] = R.r.z; X.r[3] = R.r.w;
    X.a[0] = R.a.x; X.a[1] = R.a.y; X.a[2] = R.a.z; X.a[3] = R.a.w;
  }
  X.key[0] = R.k.x; X.key[1] = R.k.y; X.key[2] = R.k.z; X.key[3] = R.k.w;
  X.inf[0] = R.i.x; X.inf[1] = R.i.y; X.inf[2] = R.i.z; X.inf[3] = R.i.w;
}

// Wave-uniform single values (tile bounds, position-index entries) go through the scalar cache:
// constant-address-space loads become s_load_dword, which count on lgkmcnt and therefore never
// make the wave drain its in-flight vector prefetch (vmcnt is in-order).  The columns and the
// truth set are read-only for the whole launch, so the non-coherent scalar cache is safe.
typedef const __attribute__((address_space(4))) int32_t* ci32p;
typedef const __attribute__((address_space(4))) uint32_t* cu32p;
__device__ __forceinline__ int uload(const int32_t* p, int i) { return ((ci32p)(uintptr_t)p)[i]; }
__device__ __forceinline__ uint32_t uload(const uint32_t* p, int i) { return ((cu32p)(uintptr_t)p)[i]; }
template <bool PACKED> __device__ __forceinline__ int rec_pos_uniform(const Cols& C, int i) {
  if (PACKED) return (int)(uload(C.pkey, i) >> 4);
  return uload(C.pos, i);
}

// single-record accessors for the rare paths (run continuation, repeated keys)
template <bool PACKED> __device__ __forceinline__ int rec_pos(const Cols& C, int i) {
  if (PACKED) return (int)(C.pkey[i] >> 4);
  return C.pos[i];
}
template <bool PACKED, bool EXT = false>
__device__ __forceinline__ void rec_packed(const Cols& C, int i, int nb, uint32_t& key, uint32_t& inf) {
  if (PACKED) { key = C.pkey[i]; inf = C.pinf[i]; return; }
  pack_record<EXT>(C.pos[i], C.ref[i], C.alt[i], C.qual[i], C.flags[i], nb, key, inf);
}

typedef const __attribute__((address_space(1))) uint32_t* gu32p;
typedef const __attribute__((address_space(1))) int32_t* gi32p;
struct TruthG {   // TruthDev with global-address-space pointers: its loads are global_load, never flat_load
  gu32p keys;
  gi32p tidx;
  gi32p ref, alt;   // EXT only
  int32_t shift, nb;
};
template <bool EXT>
__device__ __forceinline__ TruthG truth_global(const TruthDev& t) {
  TruthG g;
  if (EXT) {
    g.keys = (gu32p)t.xkeys; g.tidx = (gi32p)t.xtidx; g.ref = (gi32p)t.xref; g.alt = (gi32p)t.xalt; g.shift = t.xshift; g.nb = t.xnb;
  } else {
    g.keys = (gu32p)t.keys; g.tidx = (gi32p)t.tidx; g.ref = nullptr; g.alt = nullptr; g.shift = t.shift; g.nb = t.nb;
  }
  return g;
}

// LDS layout of the wave (dword offsets into one array, so every access is a ds_ op)
// The three histograms of the span are u16 pairs (a span holds < 65 536 records): slot s lives in the (s & 1) half of dword s >> 1.
// That is the layout span_hist has in memory, and 1.5 KB of LDS less per wave than a dword per slot.
constexpr int H_PACK = 1;
constexpr int L_HTP = 0;                                // [130] TP histogram, slot = bin + 1; slot 0 swallows the uncounted records
constexpr int L_HFP = 130;                              // [130] FP histogram, same indexing
constexpr int L_HU = 260;                               // [128] distinct-truth-key histogram, slot = bin
constexpr int L_KEYS = 388;                             // [K1_SLICE] staged truth keys of the tile
__device__ __forceinline__ void hist_add(uint32_t* lds, int table, uint32_t slot) {
  if (H_PACK) atomicAdd(&lds[table + (slot >> 1)], 1u << (16u * (slot & 1u)));
  else atomicAdd(&lds[table + slot], 1u);
}
__device__ __forceinline__ uint32_t hist_get(const uint32_t* lds, int table, uint32_t slot) {
  return H_PACK ? (lds[table + (slot >> 1)] >> (16u * (slot & 1u))) & 0xffffu : lds[table + slot];
}
constexpr int L_SMAX = L_KEYS + K1_SLICE;               // [K1_SLICE] per key: max(bin + 1) of '.'-ID matches
constexpr int L_SRF = L_SMAX + K1_SLICE;                // [K1_SLICE / 32] per key: matched by a kept record
constexpr int L_RKEY = (L_SRF + K1_SLICE / 32 + 3) & ~3;  // [256 + 28] record keys of the round (16-byte aligned), see rk()
constexpr int L_RINF = L_RKEY + 288;                    // [128] record infos, u16 each
constexpr int L_HITS = L_RINF + 128;                    // [8] one bit per record of the round: matched a truth key
constexpr int L_FLUT = L_HITS + 8;                      // [16] flag_info of the sixteen flag nibbles
constexpr int L_TCNT = L_FLUT + 16;                     // [2][SPAN_TILES] TP / FP line counts of the span's tiles, stored once per span
#ifndef QM_MASK_TILES
#define QM_MASK_TILES 8
#endif
#ifndef QM_MASK_TILES_X
#define QM_MASK_TILES_X 4   // round 4: 4 tiles (8.4 KB of LDS, 19 waves per CU instead of 20) against 2: k_classify<false, true> 3.54 -> 3.51 ms per 1 000 VCFs
#endif                      // (same-box medians of 4 processes each); 8 tiles (9.4 KB, 17 waves): 3.72

// The kept / TP mask words wait in LDS for MASK_TILES tiles and leave as ONE 16-byte-per-lane store per mask (1 KiB
// contiguous for 8 tiles): a 256-byte store per tile in the middle of the read stream cost 8 % of the kernel.
template <bool EXT> __device__ __forceinline__ constexpr int mask_tiles() { return EXT ? QM_MASK_TILES_X : QM_MASK_TILES; }
constexpr int L_MASK = (L_TCNT + 2 * SPAN_TILES + 3) & ~3;   // [2][mask_tiles][32]: kept words of the batch, then TP words (16-byte aligned)
constexpr int L_TOTAL = L_MASK + 64 * QM_MASK_TILES;
// allele-extended instantiation only: the allele codes behind the staged keys
constexpr int L_XRREF = L_MASK + 64 * QM_MASK_TILES_X;  // [256] record REF codes of the round (behind the shorter mask batch of this instantiation)
constexpr int L_XRALT = L_XRREF + 256;                  // [256] record ALT codes
constexpr int L_TOTAL_X = L_XRALT + 256;
// The allele-extended instantiation stages at most SLICE_CAP_X truth entries per tile and keeps their REF / ALT codes
// in the upper halves of the key and state arrays: no extra LDS for the truth side, 16 waves per CU again.
constexpr int SLICE_CAP_X = K1_SLICE / 2;
template <bool EXT> __device__ __forceinline__ constexpr int slice_cap() { return EXT ? SLICE_CAP_X : K1_SLICE; }
static_assert(L_XRREF % 4 == 0, "b128 LDS stores need natural alignment");
static_assert(L_RKEY % 4 == 0 && L_RINF % 2 == 0, "b128 / b64 LDS stores need natural alignment");
static_assert(K1_ROUNDS == 4, "a tile is 32 + 32 mask words");
static_assert(SPAN_TILES % QM_MASK_TILES == 0 && SPAN_TILES % QM_MASK_TILES_X == 0 && (32 * QM_MASK_TILES) % 8 == 0, "mask batches tile the span");
static_assert(K1_ROUNDS >= 3 && K1_SLICE % 64 == 0, "the next tile's slice is fetched over rounds 0..2 of the current one");

// The record keys of a round are bisected by the truth keys: probe k of every lane lands on indices that differ by
// multiples of 256 >> k, i.e. on ONE bank of the 32.  Four dwords of padding after every 32 keys put the eight 32-key
// blocks on eight different banks (and keep every lane's four keys one aligned 16-byte store).
__device__ __forceinline__ int rk(int i) { return L_RKEY + i + ((i >> 5) << 2); }

struct Slice {
  int keys, smax, srf;  // dword offsets of the active buffer
  int ref, alt;         // EXT only
  int m;                // keys staged
};

// One slice buffer is enough: the next tile's keys wait in registers (side chain) until the
// current tile's per-entry state has been flushed.
__device__ __forceinline__ void slice_select(Slice& S) {
  S.keys = L_KEYS;
  S.smax = L_SMAX;
  S.srf = L_SRF;
  S.ref = L_KEYS + SLICE_CAP_X;    // EXT only: upper halves of the key / state arrays
  S.alt = L_SMAX + SLICE_CAP_X;
}

template <bool EXT>
__device__ __forceinline__ void stage_slice(uint32_t* lds, const TruthG& tr, int c0, const Slice& S, int lane) {
  for (int j = lane; j < S.m; j += 64) {
    lds[S.keys + j] = tr.keys[c0 + j];
    lds[S.smax + j] = 0;
    if (EXT) { lds[S.ref + j] = (uint32_t)tr.ref[c0 + j]; lds[S.alt + j] = (uint32_t)tr.alt[c0 + j]; }
  }
  if (lane < K1_SLICE / 32) lds[S.srf + lane] = 0;
}

// truth slice [lo, hi) covering positions a..b, from the coarse position index
__device__ __forceinline__ void slice_range(const TruthG& tr, int a, int b, int& lo, int& hi) {
  uint32_t ba = (uint32_t)a >> tr.shift;
  uint32_t bb = ((uint32_t)b >> tr.shift) + 1u;
  const uint32_t lim = (uint32_t)tr.nb + 1u;
  ba = ba < lim ? ba : lim;
  bb = bb < lim ? bb : lim;
  lo = uload((const int32_t*)(uintptr_t)tr.tidx, (int)ba);
  hi = uload((const int32_t*)(uintptr_t)tr.tidx, (int)bb);
  if (hi < lo) hi = lo;  // only on unsorted input (results discarded)
}

struct SegBounds {   // a run of records that owns truth-entry state (a tile, or a round on the slow path)
  int a, b;          // first / last position
  int prevp, nextp;  // position just before / after it (INT32_MIN at the VCF edge)
};

template <bool PACKED> __device__ __forceinline__ SegBounds seg_bounds(const Cols& C, int sb, int se, int vn) {
  SegBounds t;
  t.a = rec_pos_uniform<PACKED>(C, sb);
  t.b = rec_pos_uniform<PACKED>(C, se - 1);
  t.prevp = (sb > 0) ? rec_pos_uniform<PACKED>(C, sb - 1) : INT32_MIN;
  t.nextp = (se < vn) ? rec_pos_uniform<PACKED>(C, se) : INT32_MIN;
  return t;
}

// ---- phase A: stage the round's keys and infos in LDS ------------------------------------
// records at or beyond `te` become key 0xffffffff (sorts last, matches nothing), info 0.
template <bool EXT>
__device__ __forceinline__ void stage_round(uint32_t* lds, In4& X, int i0, int te, int lane) {
  if (i0 - lane * 4 + 256 > te) {   // wave-uniform: only the last round of a span can be partial
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k >= te) { X.key[k] = 0xffffffffu; X.inf[k] = 0u; }
  }
  uint4 kv;
  kv.x = X.key[0]; kv.y = X.key[1]; kv.z = X.key[2]; kv.w = X.key[3];
  uint2 iv;
  iv.x = (X.inf[0] & 0xffffu) | (X.inf[1] << 16); iv.y = (X.inf[2] & 0xffffu) | (X.inf[3] << 16);
  *reinterpret_cast<uint4*>(&lds[rk(lane * 4)]) = kv;
  *reinterpret_cast<uint2*>(&lds[L_RINF + lane * 2]) = iv;
  if (EXT) {   // records beyond `te` keep whatever codes they loaded: their key matches nothing
    uint4 rv, av;
    rv.x = (uint32_t)X.r[0]; rv.y = (uint32_t)X.r[1]; rv.z = (uint32_t)X.r[2]; rv.w = (uint32_t)X.r[3];
    av.x = (uint32_t)X.a[0]; av.y = (uint32_t)X.a[1]; av.z = (uint32_t)X.a[2]; av.w = (uint32_t)X.a[3];
    *reinterpret_cast<uint4*>(&lds[L_XRREF + lane * 4]) = rv;
    *reinterpret_cast<uint4*>(&lds[L_XRALT + lane * 4]) = av;
  }
  if (lane < 8) lds[L_HITS + lane] = 0;
}

// ---- phase B: the truth keys of the round's position range search the staged records ----
// own_a: INT32_MIN when the segment owns the run at its first position, else that position
// (the run started in an earlier segment, which owns its truth entries).
template <bool EXT>
__device__ __forceinline__ void join_round(uint32_t* lds, const Slice& S, int nrec, int own_a, int lane) {
  if (S.m <= 0 || nrec <= 0) return;
  const uint32_t first_pos = lds[rk(0)] >> 4;
  const uint32_t last_pos = lds[rk(nrec - 1)] >> 4;
  // keys of the slice inside [first_pos, last_pos]: counted with ballots
  int rlo = 0, rhi = 0;
  for (int base = 0; base < S.m; base += 64) {
    const int j = base + lane;
    const uint32_t kp = j < S.m ? lds[S.keys + j] >> 4 : 0xffffffffu;
    rlo += popc64(ballot64(kp < first_pos));
    rhi += popc64(ballot64(kp <= last_pos));
  }
  for (int base = rlo; base < rhi; base += 64) {
    const int j = base + lane;
    if (j < rhi) {
      const uint32_t kkey = lds[S.keys + j];
      const uint32_t kpos = kkey >> 4;
      const uint32_t kfloor = kkey & ~15u;   // smallest key at this position: (rk >> 4) < kpos  <=>  rk < kfloor
      // first staged record with position >= kpos: a branch-free lower bound over the 256 keys
      // ... walked in the PADDED index space (36 dwords per block of 32 keys): three steps pick the block, five the
      // key inside it, none of them crosses padding; three vector instructions and one LDS read per step
      int pp = 0;
#pragma unroll
      for (int st = 144; st >= 36; st >>= 1)
        if (lds[L_RKEY + pp + st - 5] < kfloor) pp += st;   // last key of the block(s) below: logical 32 b - 1 = padded 36 b - 5
#pragma unroll
      for (int st = 16; st > 0; st >>= 1)
        if (lds[L_RKEY + pp + st - 1] < kfloor) pp += st;
      int s = pp - 4 * ((pp * 1821) >> 16);   // back to the logical index: pp / 36 blocks of padding lie below (exact for pp < 288)
      if (s < 256 && lds[L_RKEY + pp] < kfloor) s += 1;   // s == 255 still below (pp = 283: the last key)
      uint32_t mx = 0, rf = 0;
      for (; s < nrec; ++s) {   // the run of records at this position
        const uint32_t rkey = lds[rk(s)];
        if ((rkey & ~15u) != kfloor) break;
        if (rkey != kkey) continue;
        if (EXT) {   // the nibble of an extended key is a hash: the allele codes decide
          if (lds[L_XRREF + s] != lds[S.ref + j] || lds[L_XRALT + s] != lds[S.alt + j]) continue;
        }
        const uint32_t inf = (lds[L_RINF + (s >> 1)] >> (16 * (s & 1))) & 0xffffu;
        if ((inf & (I_LIVE | I_NOKEY)) != I_LIVE) continue;
        atomicOr(&lds[L_HITS + (s >> 5)], 1u << (s & 31));
        if ((int)kpos != own_a) {
          const uint32_t b1 = inf & I_BIN1;
          if ((inf & I_IDDOT) && b1 > mx) mx = b1;
          rf |= (inf & I_PASS) ? 1u : 0u;
        }
      }
      if (mx) atomicMax(&lds[S.smax + j], mx);
      if (rf) atomicOr(&lds[S.srf + (j >> 5)], 1u << (j & 31));
    }
  }
}

// number of staged slice keys < key (records of later tiles look themselves up: rare path)
__device__ __forceinline__ int slice_lower_bound(const uint32_t* lds, const Slice& S, uint32_t key) {
  int lo = 0, n = S.m;
  while (n > 0) {
    const int h = n >> 1;
    if (lds[S.keys + lo + h] < key) { lo += h + 1; n -= h + 1; } else n = h;
  }
  return lo;
}

// records after the segment that continue its last run of equal positions
template <bool PACKED, bool EXT>
__device__ __forceinline__ void continue_run(const Cols& C, uint32_t* lds, const Slice& S, int se, int vn, int bpos, int nb, int lane) {
  for (int base = se; base < vn; base += 64) {
    const int i = base + lane;
    bool cont = false;
    if (i < vn) {
      cont = (rec_pos<PACKED>(C, i) == bpos);
      if (cont && S.m > 0) {
        uint32_t key, inf;
        rec_packed<PACKED, EXT>(C, i, nb, key, inf);
        if ((inf & (I_LIVE | I_NOKEY)) == I_LIVE) {
          int j = slice_lower_bound(lds, S, key);
          if (EXT) {   // several truth entries may share the 32-bit key: find the one with these alleles
            const uint32_t rr = (uint32_t)C.ref[i], aa = (uint32_t)C.alt[i];
            while (j < S.m && lds[S.keys + j] == key && (lds[S.ref + j] != rr || lds[S.alt + j] != aa)) ++j;
          }
          if (j < S.m && lds[S.keys + j] == key) {
            if (inf & I_IDDOT) atomicMax(&lds[S.smax + j], inf & I_BIN1);
            if (inf & I_PASS) atomicOr(&lds[S.srf + (j >> 5)], 1u << (j & 31));
          }
        }
      }
    }
    if (ballot64(cont) != ~0ull) break;
  }
}

__device__ __forceinline__ uint32_t flush_slice(uint32_t* lds, const Slice& S, int lane) {
  uint32_t tpr = 0;
  for (int j = lane; j < S.m; j += 64) {
    const uint32_t mx = lds[S.smax + j];
    if (mx) hist_add(lds, L_HU, mx - 1u);
    tpr += (lds[S.srf + (j >> 5)] >> (j & 31)) & 1u;
  }
  return tpr;
}

// Has a kept record with this key been seen earlier in the VCF?  Only called when the
// predecessor has the same position; walks that run of equal positions backwards.
// <= 16 distinct single-base keys per position bound the total walk per run.
// EXT: the number of distinct keys per position is unbounded, so the walk has a budget; a record
// that exhausts it flags the VCF (SPANF_RUNLIMIT) instead of stalling the wave.
constexpr int X_WALK_LIMIT = 1 << 14;
template <bool PACKED, bool EXT>
__device__ __forceinline__ uint32_t repeated_key(const Cols& C, int i, uint32_t key, uint32_t nokey, int nb, int32_t r, int32_t a,
                                                 uint32_t& bad) {
  for (int j = i - 1; j >= 0; --j) {
    if (EXT && i - j > X_WALK_LIMIT) { bad |= 4u; break; }
    if (rec_pos<PACKED>(C, j) != (int)(key >> 4)) break;
    uint32_t kj, ij;
    rec_packed<PACKED, EXT>(C, j, nb, kj, ij);
    if (kj == key && (ij & (I_LIVE | I_PASS)) == (I_LIVE | I_PASS) && (ij & I_NOKEY) == nokey) {
      if (!EXT || (C.ref[j] == r && C.alt[j] == a)) return 1u;
    }
  }
  return 0u;
}

struct Acc {
  uint32_t bad;              // per lane: bit0 order violated, bit1 position out of range, bit2 EXT walk budget exhausted
  uint32_t fpr;              // per lane: distinct kept keys outside the truth set
  uint32_t n_pass, n_tp;     // per lane: kept / TP lines of the current tile
  uint32_t top_tp, top_fp;   // wave-uniform: records of the saturated top bin (TP, FP)
  uint32_t posor;            // per lane: OR of the positions seen (column input): tells the sort path which position bits are in use
};

// inclusive OR over each group of 8 lanes, valid in lanes 8g+7 (DPP row_shr 1,2,4; rows are 16 lanes)
__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  return v;
}

// ---- phase C: per-record work on the packed registers of the round -----------------------
// prev_last = position of the record before the round (INT32_MIN at the VCF start).
template <bool PACKED, bool EXT>
__device__ __forceinline__ void classify_round(uint32_t* lds, const Cols& C, const In4& X, int rbase, int te, int prev_last, int nb,
                                               int mslot, Acc& A, int lane) {   // mslot: the round's first word inside the batch's kept half
  const uint32_t hit = (lds[L_HITS + (lane >> 3)] >> (4 * (lane & 7))) & 15u;
  uint32_t nib = 0, anyinf = 0;   // nib: kept in bits 0..3, ID-is-'.' in 4..7, host-decided TP line in 8..11 (one bit per record)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    nib |= (X.inf[k] >> 16) << k;
    anyinf |= X.inf[k];
  }
  const uint32_t pass = nib & 15u, iddot = (nib >> 4) & 15u, tpline = (nib >> 8) & 15u;
  A.bad |= ((anyinf & I_BADPOS) | (X.posor >> 28)) ? 2u : 0u;
  A.posor |= X.posor;
  const uint32_t tpkey = (hit & iddot) | tpline;
  const uint32_t tp = pass & tpkey;
  const uint32_t fpkey = pass & ~hit;
  A.n_pass += (uint32_t)__popc(pass);
  A.n_tp += (uint32_t)__popc(tp);
  A.fpr += (uint32_t)__popc(fpkey);
  // natural-order mask words: 8 lanes x 4 records = one 32-bit word; parked in LDS, the tile stores them at once
  {
    const uint32_t sh = 4u * (uint32_t)(lane & 7);
    const uint32_t wp = or_reduce8(pass << sh);
    const uint32_t wt = or_reduce8(tp << sh);
    if ((lane & 7) == 7) {
      lds[L_MASK + mslot + (lane >> 3)] = wp;
      lds[L_MASK + 32 * mask_tiles<EXT>() + mslot + (lane >> 3)] = wt;
    }
  }
  int pp = __shfl_up((int)(X.key[3] >> 4), 1);
  if (lane == 0) pp = prev_last;
  const int i0 = rbase + lane * 4;
  uint32_t cand = 0;   // kept keys outside the truth set whose predecessor has the same position
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = (int)(X.key[k] >> 4);   // records beyond the span carry the maximal key: never below, never kept
    A.bad |= (p < pp) ? 1u : 0u;
    cand |= ((p == pp) ? 1u : 0u) << k;
    // ROC histograms: slot = bin + 1 in the TP or FP table (slot 0 swallows records without a bin).
    // The saturated top bin, where real QUALs pile up, is counted with wave ballots into scalar
    // registers instead: those lanes sit out the LDS add, so they never serialise on one address.
    {
      const uint32_t b1 = X.inf[k] & I_BIN1;
      const uint32_t notp = ((~tpkey) >> k) & 1u;
      const bool top = b1 == (uint32_t)nb;
      const uint64_t mtop = ballot64(top);
      if (mtop) {   // wave-uniform
        const uint32_t ntp = (uint32_t)popc64(ballot64(top && !notp));
        A.top_tp += ntp;
        A.top_fp += (uint32_t)popc64(mtop) - ntp;
      }
      if (!top) hist_add(lds, notp ? L_HFP : L_HTP, b1);
    }
    pp = p;
  }
  // R path: a kept key outside the truth set counts once per VCF
  cand &= fpkey;
  if (cand) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((cand >> k) & 1u) {
        // EXT: the allele codes come back from their LDS staging on this rare path instead of living in registers across the join
        const int32_t rr = EXT ? (int32_t)lds[L_XRREF + lane * 4 + k] : 0, aa = EXT ? (int32_t)lds[L_XRALT + lane * 4 + k] : 0;
        A.fpr -= repeated_key<PACKED, EXT>(C, i0 + k, X.key[k], X.inf[k] & I_NOKEY, nb, rr, aa, A.bad);
      }
  }
}

// sum over the 64 lanes, returned to all of them.  DPP only (no LDS crossbar round trips): xor 1, xor 2
// within quads, half-row and row mirrors, then the gfx9 row broadcasts carry the row sums to lane 63.
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#define QM_DPP_ADD(ctrl, rmask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false)
  QM_DPP_ADD(0xb1, 0xf);    // quad_perm [1,0,3,2]
  QM_DPP_ADD(0x4e, 0xf);    // quad_perm [2,3,0,1]
  QM_DPP_ADD(0x141, 0xf);   // row_half_mirror
  QM_DPP_ADD(0x140, 0xf);   // row_mirror: every lane holds its row's sum
  QM_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  QM_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
#undef QM_DPP_ADD
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

#ifndef K1_WAVES_PER_EU
#define K1_WAVES_PER_EU 5   // the register allocator is told to stay within 96 VGPRs (5 waves per SIMD)
#endif
#ifndef K1_WAVES_EXT_MIN
#define K1_WAVES_EXT_MIN 5   // 96 VGPRs, no scratch: the fifth wave per SIMD is worth 7 % to the allele-extended instantiation (same-box A/B)
#endif
#ifndef K1_WAVES_MAX
#define K1_WAVES_MAX 5
#endif

template <bool PACKED, bool EXT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EXT ? K1_WAVES_EXT_MIN : K1_WAVES_PER_EU, K1_WAVES_MAX))) void k_classify(ClassifyParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[(EXT ? L_TOTAL_X : L_TOTAL)];

  const int lane = (int)threadIdx.x;
  const int bid_ = (int)blockIdx.x;
  const int span_id = bid_ + P.span_base;
  const SpanDesc sp = P.spans[span_id];
  const TruthG tr = truth_global<EXT>(P.truths[sp.truth]);
  Cols C;
  C.pos = P.pos + sp.voff; C.ref = P.ref + sp.voff; C.alt = P.alt + sp.voff; C.qual = P.qual + sp.voff; C.flags = P.flags + sp.voff;
  C.pkey = P.pkey + sp.voff; C.pinf = P.pinf + sp.voff;
  uint32_t* const mpass32 = reinterpret_cast<uint32_t*>(P.mask_pass + (sp.voff >> 6));
  uint32_t* const mtp32 = reinterpret_cast<uint32_t*>(P.mask_tp + (sp.voff >> 6));
  const int vn = sp.vn;
  const int sp_end = (int)(sp.end - sp.voff);
  const int nb = P.n_bins;

  for (int i = lane; i < L_KEYS; i += 64) lds[i] = 0;   // histograms
  if (lane < 16) lds[L_FLUT + lane] = flag_info((uint32_t)lane);
  if (P.zero_acc && blockIdx.x == 0)
    for (int i = lane; i < P.zero_words; i += 64) P.zero_acc[i] = 0ull;
  if (!PACKED && P.known && P.known[sp.vcf]) return;   // found out of order by an earlier run of the same columns: the span's rows still say so
  Acc A = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t acc_pass = 0, acc_tp = 0;  // wave-uniform
  uint32_t acc_tpr = 0;               // per lane, reduced at the end

  // ---- prologue: first round in flight, first tile's bounds and slice -------------
  int tb = (int)(sp.begin - sp.voff);
  int te = (tb + K1_TILE < sp_end) ? tb + K1_TILE : sp_end;
  Raw4<PACKED> N;
  load_raw<EXT>(C, tb + lane * 4, N);
  // (sorted columns only) sixty-four records spread evenly over the WHOLE VCF, one per lane, asked for beside the first round:
  // the same 64 lines for every span of the VCF, so all but the first span find them in L2
  int spos = 0;
  if constexpr (!PACKED) if (vn >= 4096) spos = C.pos[(int)(((int64_t)lane * vn) >> 6)];
  SegBounds B = seg_bounds<PACKED>(C, tb, te, vn);
  if constexpr (!PACKED) {
    // A span whose FIRST round is out of order leaves here, three round trips in (descriptor, positions, this look), instead of
    // after its first tile (the truth side's chain of three more, four rounds of work, the epilogue's rows): the pass over a batch
    // of shuffled VCFs is what a first-seen step pays before its bucket path can start (68 -> 20 us per 256 x 10^6 records).
    // What is left behind is what anybody reads of an unsorted VCF's spans: the flag and the position bits (of the round's
    // records: the buckets' bound, as before an estimate the scatter checks).  Only positions in range count -- a round with a
    // bad one takes the long way and is flagged there -- and the record in front of the span is not looked at.
    const int n0 = te - tb < 256 ? te - tb : 256;
    const int j = lane * 4;
    const int pl = __shfl_up(N.p.w, 1);
    const bool ooo = (j + 1 < n0 && N.p.y < N.p.x) || (j + 2 < n0 && N.p.z < N.p.y) || (j + 3 < n0 && N.p.w < N.p.z) || (lane > 0 && j < n0 && N.p.x < pl);
    uint32_t po = (j < n0 ? (uint32_t)N.p.x : 0u) | (j + 1 < n0 ? (uint32_t)N.p.y : 0u) | (j + 2 < n0 ? (uint32_t)N.p.z : 0u) | (j + 3 < n0 ? (uint32_t)N.p.w : 0u);
    // ... and a span whose VCF is out of order ANYWHERE leaves as well, when the sixty-four samples tell (round 6): a VCF sorted
    // per contig -- two dozen ascending runs -- has its descents inside a third of its spans only; the others found nothing wrong,
    // classified their records (on the slow path: a run is sparse against the truth set) and were thrown away: 2.9 ms per 256 such
    // VCFs before the bucket path started.  Samples out of order prove the VCF unsorted; samples in order prove nothing.
    const int spl = __shfl_up(spos, 1);
    const bool sooo = vn >= 4096 && lane > 0 && spos < spl;
    po |= vn >= 4096 ? (uint32_t)spos : 0u;
    if (ballot64(ooo || sooo) != 0ull && ballot64((po >> 28) != 0u) == 0ull) {
      for (int o = 32; o > 0; o >>= 1) po |= (uint32_t)__shfl_xor((int)po, o);
      if (lane < 8) P.span_scal[(size_t)span_id * 8 + lane] = lane == 5 ? (uint32_t)SPANF_UNSORTED : lane == 6 ? po : 0u;
      return;
    }
  }
  int lo, hi;
  slice_range(tr, B.a, B.b, lo, hi);
  Slice S;
  slice_select(S);
  S.m = (hi - lo) <= slice_cap<EXT>() ? (hi - lo) : 0;   // an oversize slice is handled round by round
  stage_slice<EXT>(lds, tr, lo, S, lane);
  __syncthreads();

  int tile = sp.tile0;
  for (;;) {
    const bool has_next_tile = te < sp_end;
    const int ntb = tb + K1_TILE;
    const int nte = (ntb + K1_TILE < sp_end) ? ntb + K1_TILE : sp_end;
    const bool oversize = (hi - lo) > slice_cap<EXT>();
    const bool started_before = (tb > 0) && (B.prevp == B.a);
    const int own_a = started_before ? B.a : INT32_MIN;
    const bool owns_b = !(started_before && B.a == B.b);
    const int nrounds = (te - tb + 255) >> 8;

    SegBounds NB = B;
    int nlo = 0, nhi = 0;
    constexpr int NSL = (EXT ? SLICE_CAP_X : K1_SLICE) / 64;   // registers per lane that hold the next tile's slice
    uint32_t nkeys[NSL], nref[NSL], nalt[NSL];   // nref / nalt: EXT only
#pragma unroll
    for (int q = 0; q < NSL; ++q) { nkeys[q] = 0u; nref[q] = 0u; nalt[q] = 0u; }
    int prev_last = B.prevp;
    for (int r = 0; r < nrounds; ++r) {
      In4 X;
      unpack_raw<EXT>(N, nb, X, lds + L_FLUT);
      const int rbase = tb + r * 256;
      const int rend = rbase + 256 < te ? rbase + 256 : te;
      // the next tile's slice is fetched as a side chain spread over this tile's rounds, so none of
      // its three dependent global round trips (bounds -> position index -> keys) is exposed
      if (has_next_tile) {
        if (r == 0) {
          NB = seg_bounds<PACKED>(C, ntb, nte, vn);
        } else if (r == 1) {
          slice_range(tr, NB.a, NB.b, nlo, nhi);
        } else if (r == 2) {
          const int nm = (nhi - nlo) <= slice_cap<EXT>() ? (nhi - nlo) : 0;
#pragma unroll
          for (int q = 0; q < NSL; ++q) {
            const bool in = q * 64 + lane < nm;
            nkeys[q] = in ? tr.keys[nlo + q * 64 + lane] : 0u;
            if (EXT) { nref[q] = in ? (uint32_t)tr.ref[nlo + q * 64 + lane] : 0u; nalt[q] = in ? (uint32_t)tr.alt[nlo + q * 64 + lane] : 0u; }
          }
        }
      }
      // then the following rounds' records into flight (rounds are contiguous across the span's tiles)
      if (rbase + 256 < sp_end) load_raw<EXT>(C, rbase + 256 + lane * 4, N);
      stage_round<EXT>(lds, X, rbase + lane * 4, te, lane);
      __syncthreads();
      {
        if (!oversize) {
          join_round<EXT>(lds, S, rend - rbase, own_a, lane);
        } else {
          // dense truth against a sparse VCF -- or an out-of-order round, whose VCF is redone
          // through the radix sort anyway: then there is nothing to join here
          const int pl = __shfl_up((int)(X.key[3] >> 4), 1);
          const bool ooo = (lane > 0 && X.key[0] != 0xffffffffu && (int)(X.key[0] >> 4) < pl) ||
                           (X.key[1] < (X.key[0] & ~15u)) || (X.key[2] < (X.key[1] & ~15u)) || (X.key[3] < (X.key[2] & ~15u));
          if (ballot64(ooo) != 0ull) {
            A.bad |= 1u;
          } else {
            // this round is its own owner of truth state; its slice is walked in chunks staged on the spot
            const SegBounds RB = seg_bounds<PACKED>(C, rbase, rend, vn);
            const bool r_started = (rbase > 0) && (RB.prevp == RB.a);
            const int r_own_a = r_started ? RB.a : INT32_MIN;
            int rlo, rhi;
            slice_range(tr, RB.a, RB.b, rlo, rhi);
            for (int c0 = rlo; c0 < rhi; c0 += slice_cap<EXT>()) {
              S.m = (rhi - c0) < slice_cap<EXT>() ? (rhi - c0) : slice_cap<EXT>();
              stage_slice<EXT>(lds, tr, c0, S, lane);
              __syncthreads();
              join_round<EXT>(lds, S, rend - rbase, r_own_a, lane);
              if (RB.nextp == RB.b && !(r_started && RB.a == RB.b)) continue_run<PACKED, EXT>(C, lds, S, rend, vn, RB.b, nb, lane);
              __syncthreads();
              acc_tpr += flush_slice(lds, S, lane);
              __syncthreads();
            }
            S.m = 0;
          }
        }
      }
      __syncthreads();
      classify_round<PACKED, EXT>(lds, C, X, rbase, te, prev_last, nb, 32 * ((tile - sp.tile0) % mask_tiles<EXT>()) + 8 * r, A, lane);
      prev_last = (int)(lds[rk(255)] >> 4);
      __syncthreads();
    }

    // ---- tile epilogue: run continuation, per-truth-entry state -> histogram, counts ----
    if (!oversize) {
      if (B.nextp == B.b && owns_b) continue_run<PACKED, EXT>(C, lds, S, te, vn, B.b, nb, lane);
      __syncthreads();
      acc_tpr += flush_slice(lds, S, lane);
    }
    {
      const uint32_t tile_np = wave_sum(A.n_pass), tile_nt = wave_sum(A.n_tp);
      A.n_pass = 0; A.n_tp = 0;
      if (lane == 0) {   // the tile counts wait for the end of the span
        lds[L_TCNT + (tile - sp.tile0)] = tile_nt;
        lds[L_TCNT + SPAN_TILES + (tile - sp.tile0)] = tile_np - tile_nt;
      }
      acc_pass += tile_np;
      acc_tp += tile_nt;
    }
    const bool stop_unsorted = !PACKED && ballot64(A.bad & 1u) != 0ull;
    {
      // a full batch of mask words, or the span's last tiles: one 16-byte store per lane and mask (dword stores for a ragged end)
      constexpr int MT = mask_tiles<EXT>();
      const int tin = (tile - sp.tile0) % MT;
      if (tin == MT - 1 || !has_next_tile || stop_unsorted) {
        __syncthreads();
        const int b0 = tb - tin * K1_TILE;                    // first record of the batch (VCF-relative)
        const int nd = ((te - b0 + 255) >> 8) * 8;            // dwords of the rounds that ran (whole rounds: every VCF owns its masks up to a multiple of 256 records, and k_compact reads 64-bit words); the rest of the LDS batch is stale
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint32_t* dst = (half ? mtp32 : mpass32) + (b0 >> 5);
          const int src = L_MASK + half * 32 * MT;
          for (int w = 4 * lane; w < 32 * MT; w += 256) {
            constexpr bool mask_nt = true;   // streaming: 0.8 % of the kernel on every one of 14 allocations (profiles/r04_classify_mask_nt.log)
            if (w + 3 < nd) {
              typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
              if (mask_nt) __builtin_nontemporal_store(*reinterpret_cast<const v4u_*>(&lds[src + w]), reinterpret_cast<v4u_*>(dst + w));
              else *reinterpret_cast<uint4*>(dst + w) = *reinterpret_cast<const uint4*>(&lds[src + w]);
            }
            else for (int k = 0; k < 4; ++k) if (w + k < nd) dst[w + k] = lds[src + w + k];
          }
        }
      }
    }

    if (!has_next_tile) break;
    // Out of order: the VCF will be redone through the radix sort and nothing computed here is used
    // (k_compact skips it, the sort path rewrites its masks, counts and rows) -- stop streaming it.
    if (stop_unsorted) break;
    // ---- stage the next tile's slice into the other LDS half ----------------------------
    // (a full tile has K1_ROUNDS >= 3 rounds, so the side chain above has run to its end)
    B = NB;
    tb = ntb;
    te = nte;
    ++tile;
    lo = nlo;
    hi = nhi;
    S.m = (hi - lo) <= slice_cap<EXT>() ? (hi - lo) : 0;
#pragma unroll
    for (int q = 0; q < NSL; ++q) {
      const int j = q * 64 + lane;
      if (j < S.m) {
        lds[S.keys + j] = nkeys[q]; lds[S.smax + j] = 0;
        if (EXT) { lds[S.ref + j] = nref[q]; lds[S.alt + j] = nalt[q]; }
      }
    }
    if (lane < K1_SLICE / 32) lds[S.srf + lane] = 0;
    __syncthreads();
  }

  // ---- span epilogue -------------------------------------------------------------------
  acc_tpr = wave_sum(acc_tpr);
  A.fpr = wave_sum(A.fpr);
  const uint64_t any_uns = ballot64(A.bad & 1u);
  const uint64_t any_bad = ballot64(A.bad & 2u);
  const uint64_t any_lim = ballot64(A.bad & 4u);
  __syncthreads();
  const uint32_t top_tp = A.top_tp, top_fp = A.top_fp;
  // a span holds at most SPAN_TILES * K1_TILE < 65 536 records: two bins per dword (bin 2i low, 2i + 1 high)
  uint32_t* oh = P.span_hist + (size_t)span_id * SPAN_HIST_WORDS;
  for (int i = lane; i < 128; i += 64) {
    const int b0 = 2 * i, b1 = 2 * i + 1;
#define K1_HST(p, x) (*(p) = (x))
    K1_HST(&oh[i], (hist_get(lds, L_HTP, 1 + b0) + (b0 == nb - 1 ? top_tp : 0u)) | ((hist_get(lds, L_HTP, 1 + b1) + (b1 == nb - 1 ? top_tp : 0u)) << 16));
    K1_HST(&oh[128 + i], (hist_get(lds, L_HFP, 1 + b0) + (b0 == nb - 1 ? top_fp : 0u)) | ((hist_get(lds, L_HFP, 1 + b1) + (b1 == nb - 1 ? top_fp : 0u)) << 16));
    K1_HST(&oh[256 + i], hist_get(lds, L_HU, b0) | (hist_get(lds, L_HU, b1) << 16));
#undef K1_HST
  }
  {   // the span's tile counts: lanes 0..15 TP, 16..31 FP (tiles never reached on an unsorted VCF hold nothing anyone reads)
    const int nt_span = (sp_end - (int)(sp.begin - sp.voff) + K1_TILE - 1) / K1_TILE;
    const int t = lane & (SPAN_TILES - 1);
    if (lane < 2 * SPAN_TILES && t < nt_span) (lane < SPAN_TILES ? P.tile_tp : P.tile_fp)[sp.tile0 + t] = lds[L_TCNT + lane];
  }
  if (lane == 0) {
    uint32_t* sc = P.span_scal + (size_t)span_id * 8;
    sc[0] = acc_pass; sc[1] = acc_tp; sc[2] = acc_pass - acc_tp; sc[3] = acc_tpr; sc[4] = A.fpr;
    sc[5] = (any_uns ? SPANF_UNSORTED : 0u) | (any_bad ? SPANF_BADPOS : 0u) | (any_lim ? SPANF_RUNLIMIT : 0u);
    sc[7] = 0;
  }
  {
    uint32_t po = A.posor;   // OR over the wave (once per span)
    for (int o = 32; o > 0; o >>= 1) po |= (uint32_t)__shfl_xor((int)po, o);
    if (lane == 0) P.span_scal[(size_t)span_id * 8 + 6] = po;
  }
}

// ---------------------------------------------------------------------------
// k_finalize: one workgroup (256 threads) per VCF
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_finalize(FinalizeParams P) {
  __shared__ uint32_t s_h[3][4];
  __shared__ uint32_t s_scan[256];
  const int v = (int)blockIdx.x + P.vcf_base;
  const int tid = (int)threadIdx.x;
  const VcfDesc vd = P.vcfs[v];
  // bucket rows: the rows of the buckets above the segment's highest (of either stream: row s belongs to bucket s & 255) were not written
  const int row_cap = P.row_cap ? ((int)P.row_cap[v] > 1 ? (int)P.row_cap[v] : 1) : HB_BUCKETS;
  const int nb = P.n_bins;

  // the spans' scalar rows (8 words each: five counters, the flags, the OR of the positions), every thread two or three of
  // the words: one round of coalesced loads and LDS atomics.  (Eight threads walking the rows one after the other were the
  // long pole of this kernel: 62 dependent round trips per VCF.)
  __shared__ unsigned long long s_sc[5];
  __shared__ uint32_t s_fl, s_or;
  if (tid < 5) s_sc[tid] = 0ull;
  if (tid == 5) s_fl = 0u;
  if (tid == 6) s_or = 0u;
  __syncthreads();
  for (int i = tid; i < vd.nspans * 8; i += 256) {
    if (P.row_cap && ((i >> 3) & (HB_BUCKETS - 1)) >= row_cap) continue;
    const uint32_t x = P.span_scal[(size_t)vd.span0 * 8 + i];
    const int k = i & 7;
    if (k < 5) { if (x) atomicAdd(&s_sc[k], (unsigned long long)x); }
    else if (k == 5) { if (x) atomicOr(&s_fl, x); }
    else if (k == 6) { if (x) atomicOr(&s_or, x); }
  }
  // P.parts: bit 0 = the part the compaction waits for (per-VCF flags, tile offsets), bit 1 = the rows (ROC, scalars, per-truth
  // sums).  qm_batch_run launches the two apart, the rows on the second stream beside the compaction; everybody else wants both.
  const bool rows = (P.parts & 2) != 0, offsets = (P.parts & 1) != 0;
  // the per-VCF flags, the position bits and their host-mapped mirrors (what qm_batch_finish reads behind the run)
  auto write_flags = [&](uint32_t fl) {
    if (P.vcf_posor) P.vcf_posor[v] = s_or;
    const uint32_t out = fl & (SPANF_UNSORTED | SPANF_BADPOS | SPANF_RUNLIMIT | SPANF_OVERFLOW);
    P.vcf_flags[v] = out;
    if (P.host_flags) {
      __hip_atomic_store(P.host_flags + v, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(P.host_aux + v, P.row_cap ? P.row_cap[v] : s_or, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (P.chunk_bad && ((out & (SPANF_OVERFLOW | SPANF_BADPOS)) || (P.row_cap && P.row_cap[v] > P.row_cap_limit))) atomicOr(P.chunk_bad, 1u);
  };
  // the run's own k_finalize (P.lazy_unsorted): a VCF its spans found out of order is redone from the columns -- rows, scalars,
  // tile counts and offsets are all written again behind that (k_sort_copy_rows, k_tile_counts, the rescan) and the compaction of
  // this run skips it: the flags are all anybody reads, and the workgroup leaves behind them (two of a first-seen step's
  // dependent round trips: the rows' sums, or the tile counts and their scan)
  if (P.lazy_unsorted) {
    __syncthreads();
    if (s_fl & SPANF_UNSORTED) {
      if (offsets && tid == 5) write_flags(s_fl);
      return;
    }
  }
  // sum span histograms.  A row is three histograms of 128 words (two u16 bins per word): thread t takes word t & 127 of the rows
  // t >> 7, t >> 7 + 2, ... -- every word is read once (thread = bin read every word twice) and a VCF's rows are walked by two
  // halves of the workgroup side by side, which is what matters for the few hundred workgroups this kernel has: the loop is bound
  // by its round trips (512 rows of a two-stream bucket segment took 160 us) -- and the halves meet in LDS: thread = bin from there on.
  __shared__ uint32_t s_part[2][4][256];
  uint32_t h0 = 0, h1 = 0, h2 = 0, hsub = 0;
  if (rows) {
    const int w = tid & 127, g = tid >> 7;
    uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0, u0 = 0, u1 = 0;
#pragma unroll 8
    for (int s = g; s < vd.nspans; s += 2) {   // unrolled: the loads of eight spans are in flight together (10 M-record VCFs have 611 spans)
      if (P.row_cap && (s & (HB_BUCKETS - 1)) >= row_cap) continue;
      const uint32_t* sh = P.span_hist + (size_t)(vd.span0 + s) * SPAN_HIST_WORDS;
      const uint32_t x = sh[w], y = sh[128 + w], z = sh[256 + w];
      a0 += x & 0xffffu; a1 += x >> 16; b0 += y & 0xffffu; b1 += y >> 16; c0 += z & 0xffffu; c1 += z >> 16;
      if (s < HB_BUCKETS) { u0 += x & 0xffffu; u1 += x >> 16; }
    }
    s_part[g][0][2 * w] = a0; s_part[g][0][2 * w + 1] = a1; s_part[g][1][2 * w] = b0; s_part[g][1][2 * w + 1] = b1;
    s_part[g][2][2 * w] = c0; s_part[g][2][2 * w + 1] = c1; s_part[g][3][2 * w] = u0; s_part[g][3][2 * w + 1] = u1;
  }
  __syncthreads();
  if (rows) {
    h0 = s_part[0][0][tid] + s_part[1][0][tid]; h1 = s_part[0][1][tid] + s_part[1][1][tid];
    h2 = s_part[0][2][tid] + s_part[1][2][tid]; hsub = s_part[0][3][tid] + s_part[1][3][tid];
    // bucket rows whose scatter counted every first-stream record by bin: their FP histogram is what is left of it
    if (P.all_hist && tid < nb) h1 += P.all_hist[(size_t)v * SEG_HIST_WORDS + 1 + tid] - hsub;
  }
  // ROC = suffix sums over the bins (bins at and above n_bins are empty): shuffles inside the wave, the waves' totals through LDS
  {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t i0 = h0, i1 = h1, i2 = h2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y0 = __shfl_down(i0, o), y1 = __shfl_down(i1, o), y2 = __shfl_down(i2, o);
      if (lane + o < 64) { i0 += y0; i1 += y1; i2 += y2; }
    }
    if (lane == 0) { s_h[0][wave] = i0; s_h[1][wave] = i1; s_h[2][wave] = i2; }
    __syncthreads();   // also: the scalar rows are summed
    uint64_t c0 = i0, c1 = i1, c2 = i2;
    for (int w = wave + 1; w < 4; ++w) { c0 += s_h[0][w]; c1 += s_h[1][w]; c2 += s_h[2][w]; }
    const uint32_t fl = s_fl;
    const bool unsorted = (fl & SPANF_UNSORTED) != 0u;
    if (rows && tid < nb) {
      uint64_t* roc = P.roc + (size_t)v * 3 * nb;
      roc[tid] = c0; roc[nb + tid] = c1; roc[2 * nb + tid] = c2;
      if (P.global_acc && !unsorted) {   // an unsorted VCF's numbers are discarded (redone by the sort path)
        unsigned long long* g = reinterpret_cast<unsigned long long*>(P.global_acc) + (size_t)vd.truth * 3 * nb;
        if (c0) atomicAdd(&g[tid], (unsigned long long)c0);
        if (c1) atomicAdd(&g[nb + tid], (unsigned long long)c1);
        if (c2) atomicAdd(&g[2 * nb + tid], (unsigned long long)c2);
      }
    }
    // scalars
    if (rows && tid < 8) {
      int64_t* sc = P.scalars + (size_t)v * 8;
      if (tid < 5) sc[tid] = (int64_t)s_sc[tid];
      else if (tid == 5) sc[5] = unsorted ? 0 : 1;
      else if (tid == 6) sc[6] = vd.n;
      else sc[7] = P.ext ? P.truths[vd.truth].xn : P.truths[vd.truth].n;
    }
    if (offsets && tid == 5) write_flags(fl);
  }
  if (!offsets) return;
  // exclusive scan of the tile counts (TP and FP together) over the VCF's tiles: four consecutive tiles
  // per thread in registers, the 256 thread totals with wave shuffles -- two barriers per 1 024 tiles
  {
    const int lane = tid & 63, wave = tid >> 6;
    // the FP run ends at the back of the VCF's region: its offsets start at n - (FP lines of the VCF), so that a wave of
    // k_compact finds the place of its tile's entries with one load
    uint32_t fsum = 0;
    for (int i = tid; i < vd.ntiles; i += 256) fsum += P.tile_fp[vd.tile0 + i];
    fsum = wave_sum(fsum);
    if (lane == 0) s_scan[8 + wave] = fsum;
    __syncthreads();
    const uint32_t fp_total = s_scan[8] + s_scan[9] + s_scan[10] + s_scan[11];
    uint32_t carry_t = 0, carry_f = (uint32_t)vd.n - fp_total;
    if (tid == 0 && P.vcf_tot) P.vcf_tot[2 * v + 1] = fp_total;
    for (int base = 0; base < vd.ntiles; base += 1024) {
      const int t0 = base + tid * 4;
      uint32_t xt[4], xf[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool in = t0 + k < vd.ntiles;
        xt[k] = in ? P.tile_tp[vd.tile0 + t0 + k] : 0u;
        xf[k] = in ? P.tile_fp[vd.tile0 + t0 + k] : 0u;
      }
      const uint32_t st = xt[0] + xt[1] + xt[2] + xt[3], sf = xf[0] + xf[1] + xf[2] + xf[3];
      uint32_t it = st, jf = sf;   // inclusive over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t yt = __shfl_up(it, o), yf = __shfl_up(jf, o);
        if (lane >= o) { it += yt; jf += yf; }
      }
      if (lane == 63) { s_scan[wave] = it; s_scan[4 + wave] = jf; }
      __syncthreads();
      uint32_t wt = 0, wf = 0, bt = 0, bf = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t a = s_scan[w], c = s_scan[4 + w];
        wt += w < wave ? a : 0u; wf += w < wave ? c : 0u;
        bt += a; bf += c;
      }
      uint32_t et = carry_t + wt + it - st, ef = carry_f + wf + jf - sf;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t0 + k < vd.ntiles) { P.tile_tp_off[vd.tile0 + t0 + k] = et; P.tile_fp_off[vd.tile0 + t0 + k] = ef; }
        et += xt[k]; ef += xf[k];
      }
      carry_t += bt; carry_f += bf;
      __syncthreads();
    }
    if (tid == 0 && P.vcf_tot) P.vcf_tot[2 * v] = carry_t;   // TP lines of the VCF (k_compact: where the TP list ends)
  }
}

// ---------------------------------------------------------------------------
// k_compact: class masks -> compacted line-index lists (prefix-sum compaction).
// idx region of VCF v (vd.n entries at vd.off): TP line indices ascending from the
// front, FP ascending, ending at the back.  Stage replaced: the shell '>' redirections
// of program/extract_TP_FP_SNPs.py:50-53.
//
// Round 4: one wave per TILE (1 024 records), K3_WAVES tiles of a span per workgroup.
// * What bounded the kernel of rounds 1-3 (one wave per span of 16 tiles, a ring per list)
//   was not its instruction stream: with the stores switched off it took 0.42 ms, with
//   ONLY the stores 0.82 ms (tools/probe/store_shape.hip: waves that each write 60 KB, 1 KiB
//   at a time with work in between, reach 5.1-5.4 TB/s; waves that write 4 KB and leave
//   reach 6.4-6.5).  Short waves keep the chip's write front compact: what is being
//   written at any one time is a few tens of MB of neighbouring lines instead of 5 000
//   separate streams.
// * A lane takes EIGHT records of a pass of 512: one byte of the TP mask and of kept & ~TP
//   (byte loads, 64 B per wave-instruction).  One DPP prefix sum over the lanes' popcounts
//   (TP and FP counts packed into one register) gives every lane the place of its first
//   entry of either list; a 256-entry table in LDS (mask byte -> the positions of its set
//   bits, one per byte) turns the lane's byte into its <= 8 entries, and the lane stores all
//   eight slots UNCONDITIONALLY, highest slot first: a slot beyond the lane's count lands on
//   a place a LATER lane owns (or beyond the list), and that lane's own store of the place
//   is issued after it -- slot j of a later lane starts at a higher address, so it reaches
//   the same place with a SMALLER j -- and the LDS keeps a wave's stores in order.  No
//   per-entry branch, execution mask or rank: two instructions per slot.  (Lanes without an
//   entry sit the eight stores out: their first slot IS the next lane's.)  Both passes of
//   the FP list are stored before the TP list, which lies behind it in the wave's buffer.
// * The tile's two runs leave LDS with 16-byte-per-lane stores.  A run starts wherever the
//   tile's offset says: it is laid out in LDS with the same 16-byte phase as in the output,
//   so every lane moves one aligned quad; the <= 3 entries in front of the first whole quad
//   and behind the last one (of both lists: <= 12) leave with ONE masked dword store.
// * Workgroups are dealt round-robin to the 8 XCDs: the block index is mapped so that every
//   XCD owns a contiguous range of tiles, and the 128-byte lines two neighbouring tiles
//   share meet in ONE L2 instead of reaching HBM as two partial writes.
// k_finalize leaves tile_tp_off / tile_fp_off as places inside the VCF's region (the FP
// offsets already shifted to the back), so a wave needs its span descriptor and then one
// round trip (offsets and mask bytes together).
// ---------------------------------------------------------------------------
#ifndef K3_TILES
#define K3_TILES 4                              // tiles per wave: its stores overlap the next tile's work, and it leaves before the write front widens
#endif
#ifndef K3_WAVES
#define K3_WAVES 4                              // waves per workgroup; the table is loaded once per workgroup
#endif
static_assert(SPAN_TILES % (K3_WAVES * K3_TILES) == 0, "a workgroup's tiles lie in one span");
constexpr int K3_PASS = 512;                    // records per pass: 8 per lane
static_assert(K1_TILE == 2 * K3_PASS, "a tile is two passes");
constexpr int K3_NPASS = 2 * K3_TILES;
// A wave stores whole chunks of a list: the chunk that BEGINS among its entries is completed from the ONE tile behind its own
// (k3_own).  FP list (dense): 256 entries = the 1 KiB of one store instruction.  TP list (a few per cent of the records): 64
// entries = two 128-byte lines.
#ifndef K3_FCH_LOG
#define K3_FCH_LOG 8
#endif
#ifndef K3_TCH_LOG
#define K3_TCH_LOG 6
#endif
static_assert(K3_FCH_LOG >= 4 && K3_FCH_LOG <= 8 && K3_TCH_LOG >= 4 && K3_TCH_LOG <= 8, "a chunk is whole 64-byte pieces, at most one store instruction");
constexpr int K3_EXTRA = 2;                     // passes behind the wave's own whose mask bytes it also reads: ONE tile (k3_own's reach)
static_assert(K3_EXTRA * K3_PASS == K1_TILE, "a wave reaches exactly one tile beyond its own");
constexpr int K3_BUF = 256 + K3_PASS + 16;      // entries per list: the carried partial chunk, one pass, the slack the unconditional stores run into

// mask byte -> positions of its set bits, ascending, one per byte (64 bits per entry)
struct K3Lut {
  uint32_t w[512];
  constexpr K3Lut() : w() {
    for (int v = 0; v < 256; ++v) {
      uint64_t l = 0;
      int n = 0;
      for (int b = 0; b < 8; ++b)
        if ((v >> b) & 1) { l |= (uint64_t)b << (8 * n); ++n; }
      w[2 * v] = (uint32_t)l;
      w[2 * v + 1] = (uint32_t)(l >> 32);
    }
  }
};
__device__ const K3Lut k3_lut = K3Lut();

typedef volatile uint32_t __attribute__((address_space(3)))* k3_ldsp;   // explicit: a generic pointer here turns every access into flat_*
typedef int k3_v4i __attribute__((ext_vector_type(4)));
typedef volatile k3_v4i __attribute__((address_space(3)))* k3_lds4p;

// base | byte K of l in one instruction (SDWA operand select; the compiler finds it for bytes 0 and 3 only)
template <int K> __device__ __forceinline__ uint32_t k3_or_byte(uint32_t base, uint32_t l) {
  uint32_t r;
  if (K == 0) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(l), "v"(base));
  if (K == 1) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(l), "v"(base));
  if (K == 2) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(l), "v"(base));
  if (K == 3) asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(l), "v"(base));
  return r;
}
// the lane's <= 8 entries of one list: slot j at a + j, highest first, all eight whatever the count (see above)
__device__ __forceinline__ void k3_emit(k3_ldsp buf, const uint32_t* lut, uint32_t a, uint32_t m, uint32_t base) {
  if (m) {
    const uint2 l = *reinterpret_cast<const uint2*>(&lut[2 * m]);
    k3_ldsp d = buf + a;
    d[7] = k3_or_byte<3>(base, l.y);
    d[6] = k3_or_byte<2>(base, l.y);
    d[5] = k3_or_byte<1>(base, l.y);
    d[4] = k3_or_byte<0>(base, l.y);
    d[3] = k3_or_byte<3>(base, l.x);
    d[2] = k3_or_byte<2>(base, l.x);
    d[1] = k3_or_byte<1>(base, l.x);
    d[0] = k3_or_byte<0>(base, l.x);
  }
}
// inclusive prefix sum over the wave (DPP only), the wave's total to everybody through `tot`
__device__ __forceinline__ uint32_t k3_scan(uint32_t c, uint32_t& tot) {
  uint32_t x = c;
#define K3_DPP_ADD(ctrl, rmask) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false)
  K3_DPP_ADD(0x111, 0xf);   // row_shr:1
  K3_DPP_ADD(0x112, 0xf);   // row_shr:2
  K3_DPP_ADD(0x114, 0xf);   // row_shr:4
  K3_DPP_ADD(0x118, 0xf);   // row_shr:8: inclusive inside every row of 16
  K3_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
  K3_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef K3_DPP_ADD
  tot = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  return x;
}

struct K3List {
  k3_ldsp buf;              // LDS, K3_BUF entries; buf[i] is entry g0 + i of the VCF's region
  int32_t* out;             // the VCF's index region
  uint32_t lo, hi;          // the wave STORES the entries [lo, hi) of the list: whole 1 KiB chunks, except where the list itself begins or ends
  uint32_t g0;              // entry of buf[0]: a multiple of 256
  uint32_t n;               // entries in buf (the first wave-offset & 255 of them are not there: below lo, never stored)
};
__device__ __forceinline__ void k3_store4(int32_t* p, k3_v4i w) {
  // non-temporal: nobody reads the lists on the device, and plain stores leave their dirty lines to be written back while the next
  // step's k_classify streams (same-box A/B over 5 processes each: k_compact alike, k_classify + 0.19 ms with plain stores)
  __builtin_nontemporal_store(w, reinterpret_cast<k3_v4i*>(p));
}
// entries [lo, hi) of the chunk that starts at buf index c0 (a multiple of 256; entry g0 + c0): whole quads with one
// 16-byte store per lane, the <= 3 + 3 entries around them with one masked dword store.  Only where a LIST begins or ends
// (twice per VCF and list): every other chunk leaves whole.
__device__ __forceinline__ void k3_store_part(const K3List& X, uint32_t c0, uint32_t lo, uint32_t hi, int lane) {
  const uint32_t b = X.g0 + c0;                     // entry of the chunk's first slot
  const uint32_t q = b + 4u * (uint32_t)lane;       // the lane's quad
  if (q >= lo && q + 4u <= hi) k3_store4(X.out + q, *reinterpret_cast<k3_lds4p>(X.buf + (c0 + 4u * lane)));
  if (lane < 6) {
    // lanes 0..2: entries in front of the first whole quad; 3..5: behind the last one
    const uint32_t lo4 = (lo + 3u) & ~3u, hi4 = hi & ~3u;
    uint32_t i = lane < 3 ? lo + (uint32_t)lane : hi4 + (uint32_t)(lane - 3);
    const bool ok = lane < 3 ? (i < lo4 && i < hi) : (i < hi && i >= lo && i >= lo4);
    if (ok) X.out[i] = (int32_t)X.buf[i - X.g0];
  }
}
// Complete chunks of 2^CL entries leave -- those the wave owns --, the partial chunk behind them moves to the front (all
// wave-uniform).  What leaves is a run of whole chunks, i.e. of whole 64-byte pieces of HBM: 256-entry windows of it with one
// 16-byte store per lane; only where the LIST begins or ends inside (twice per VCF) does a window carry ragged ends.
template <int CL>
__device__ __forceinline__ void k3_drain(K3List& X, int lane) {
  const uint32_t nfull = X.n >> CL;
  if (!nfull) return;
  const uint32_t top = X.g0 + (nfull << CL);                                   // entries [g0, top) are complete chunks
  const uint32_t s0 = X.lo > X.g0 ? X.lo : X.g0, e0 = X.hi < top ? X.hi : top;   // ... of which the wave stores [s0, e0)
  for (uint32_t w = X.g0; w < e0; w += 256u) {
    if (w + 256u <= s0) continue;
    if (s0 <= w && w + 256u <= e0) k3_store4(X.out + w + 4u * lane, *reinterpret_cast<k3_lds4p>(X.buf + ((w - X.g0) + 4u * lane)));
    else k3_store_part(X, w - X.g0, s0 > w ? s0 : w, e0 < w + 256u ? e0 : w + 256u, lane);
  }
  const uint32_t left = X.n - (nfull << CL);                                    // < 2^CL <= 256
  if (4u * (uint32_t)lane < left) {
    const k3_v4i v = *reinterpret_cast<k3_lds4p>(X.buf + ((nfull << CL) + 4u * lane));
    *reinterpret_cast<k3_lds4p>(X.buf + 4u * lane) = v;
  }
  X.g0 = top;
  X.n = left;
}
// [lo, hi) of a wave whose own tiles hold the entries [a, a2) of a list that spans [l0, l1): no partial store is left but
// where a list begins or ends, or where it is so sparse that a chunk does not end within a tile (on physically contiguous
// batches whole chunks alone are worth 1.07 -> 0.78 ms; same-box medians of 8 processes against waves that store exactly
// their own entries: 0.726 against 0.766 ms, spread 0.72-0.76 against 0.75-0.80, profiles/r05_compact_form_ab.log).
//   * The wave OWNS the chunks that begin among its entries (the list's ragged head included, wherever its first entry lies)
//     and completes the last of them from the tile behind its own, never further: `reach` is the list's offset behind that
//     tile, so the wave's work is bounded whatever the list holds (a sparse TP list made the first version walk to the end
//     of its VCF: ADVICE round 4).
//   * What an owner leaves -- entries of its last chunk beyond its reach -- is stored by the wave whose tiles hold them: the
//     wave in front (first entry `aprev`, reach `rnext` = the offset behind THIS wave's first tile) owns the chunk that
//     holds `a` exactly when that chunk begins at or behind `aprev`; an owner further back reaches nothing of this wave.
// Every wave derives both ends from tile offsets alone, so neighbours agree without talking: every entry leaves exactly once.
template <int CL>
__device__ __forceinline__ void k3_own(K3List& X, uint32_t a, uint32_t a2, uint32_t aprev, uint32_t rnext, uint32_t reach, uint32_t l0, uint32_t l1) {
  constexpr uint32_t M = (1u << CL) - 1u, C = 1u << CL;
  const uint32_t s = a & ~M;                                   // the chunk that holds entry a
  const uint32_t up = (a + M) & ~M, up2 = (a2 + M) & ~M;
  const bool head = a == l0;                                   // nothing of the list lies in front of this wave (its ragged first chunk begins here, if anywhere)
  const bool begins = up < a2 || (head && a2 > a);             // a chunk begins among the wave's entries
  const uint32_t sb = s > l0 ? s : l0;                         // first entry of the chunk that holds a
  const bool inherited = !head && s != a && sb >= aprev && sb < a;   // ... which the wave in front owns, and stores as far as it reaches
  X.lo = inherited ? (s + C < rnext ? s + C : rnext) : a;
  const uint32_t end = up2 < reach ? up2 : reach;
  X.hi = begins ? (end < l1 ? end : l1) : a2;
  X.g0 = s;
  X.n = a - s;
}

__global__ __launch_bounds__(64 * K3_WAVES) void k_compact(CompactParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[K3_WAVES][2][K3_BUF];
  __shared__ __attribute__((aligned(16))) uint32_t s_lut[512];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int BPS = SPAN_TILES / (K3_WAVES * K3_TILES);   // workgroups per span
  constexpr int NLUT = (256 + 64 * K3_WAVES - 1) / (64 * K3_WAVES);
  constexpr int NPRE = K3_NPASS + K3_EXTRA;                 // passes whose mask bytes are asked for at the start
  const int L = (int)blockIdx.x;   // launch order (an XCD-contiguous tile order was measured with the other ownership form only; not kept)
  // A span that k_classify itself left as out of order belongs to a VCF this launch skips (its flags say so, below): the span's
  // own word, asked for beside its descriptor, lets the workgroup go one round trip earlier -- the launch over a batch of
  // shuffled VCFs is 7 rounds of workgroups that all do just that (28 -> 8 us of device time on a first-seen step)
  const uint32_t own_fl = P.skip_unsorted ? P.span_scal[(size_t)(L / BPS + P.span_base) * 8 + 5] : 0u;
  const SpanDesc sp = P.spans[L / BPS + P.span_base];
  // everything the wave needs is asked for at once, behind the span descriptor: the table, the VCF's flags and list sizes, the
  // offsets of its first tile, of its neighbours' and of the tiles behind them, the mask bytes of its passes and of the tile behind
  uint2 lut[NLUT];
#pragma unroll
  for (int k = 0; k < NLUT; ++k) {
    const int i = tid + k * 64 * K3_WAVES;
    lut[k] = reinterpret_cast<const uint2*>(k3_lut.w)[i < 256 ? i : 0];
  }
  if (own_fl & SPANF_UNSORTED) return;   // (the whole workgroup: one span)
  const int tl = ((L % BPS) * K3_WAVES + wave) * K3_TILES;     // the wave's first tile inside the span
  const int rb = (int)(sp.begin - sp.voff) + tl * K1_TILE;     // its first record inside the VCF (a multiple of K1_TILE)
  const bool live = (int64_t)sp.voff + rb < sp.end;
  const int t = live ? sp.tile0 + tl : sp.tile0;
  const bool has_next = live && rb + K3_TILES * K1_TILE < sp.vn;   // another wave of the same VCF follows
  const bool has_t1 = live && rb + K1_TILE < sp.vn;                // the VCF has a tile behind the wave's first one
  const bool has_reach = live && rb + (K3_TILES + 1) * K1_TILE < sp.vn;   // ... and one behind the successor's first one
  const bool first = rb == 0;
  const uint32_t vflags = P.vcf_flags[sp.vcf];
  const uint32_t tp_total = P.vcf_tot[2 * sp.vcf], fp_total = P.vcf_tot[2 * sp.vcf + 1];
  const uint32_t aT = P.tile_tp_off[t], aF = P.tile_fp_off[t];
  const uint32_t aT2 = has_next ? P.tile_tp_off[t + K3_TILES] : tp_total, aF2 = has_next ? P.tile_fp_off[t + K3_TILES] : (uint32_t)sp.vn;
  const uint32_t pT = first || !live ? 0u : P.tile_tp_off[t - K3_TILES], pF = first || !live ? 0u : P.tile_fp_off[t - K3_TILES];
  const uint32_t nT = has_t1 ? P.tile_tp_off[t + 1] : tp_total, nF = has_t1 ? P.tile_fp_off[t + 1] : (uint32_t)sp.vn;
  const uint32_t rT = has_reach ? P.tile_tp_off[t + K3_TILES + 1] : tp_total, rF = has_reach ? P.tile_fp_off[t + K3_TILES + 1] : (uint32_t)sp.vn;
  // mask bytes from the wave's first record to the end of the VCF: whole 64-bit words (bits beyond the last record are clear)
  const int nb = live ? (((sp.vn - rb) + 63) >> 6) * 8 : 0;
  const uint8_t* mp = reinterpret_cast<const uint8_t*>(P.mask_pass) + ((sp.voff + rb) >> 3);
  const uint8_t* mt = reinterpret_cast<const uint8_t*>(P.mask_tp) + ((sp.voff + rb) >> 3);
  uint32_t mk[NPRE], mq[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int b = u * 64 + lane;
    mk[u] = b < nb ? mp[b] : 0u;
    mq[u] = b < nb ? mt[b] : 0u;
  }
#pragma unroll
  for (int k = 0; k < NLUT; ++k) {
    const int i = tid + k * 64 * K3_WAVES;
    if (i < 256) *reinterpret_cast<uint2*>(&s_lut[2 * i]) = lut[k];
  }
  asm volatile("" :: "s"(aT), "s"(aF), "s"(aT2), "s"(aF2), "s"(vflags), "s"(tp_total), "s"(fp_total));   // here, not behind the barrier where the compiler would sink these loads to
  asm volatile("" :: "s"(pT), "s"(pF), "s"(nT), "s"(nF), "s"(rT), "s"(rF));
  __syncthreads();
  // VCFs found out of order by this run are redone by the sort path: their masks and counts are not meaningful yet
  if (!live || (P.skip_unsorted && (vflags & SPANF_UNSORTED))) return;
  K3List T, F;
  T.buf = (k3_ldsp)s_buf[wave][0]; F.buf = (k3_ldsp)s_buf[wave][1];
  T.out = F.out = P.idx + sp.voff;
  k3_own<K3_TCH_LOG>(T, aT, aT2, pT, nT, rT, 0u, tp_total);
  k3_own<K3_FCH_LOG>(F, aF, aF2, pF, nF, rF, (uint32_t)sp.vn - fp_total, (uint32_t)sp.vn);
  if (T.lo >= T.hi && F.lo >= F.hi) return;                    // nothing of either list is this wave's to store
  const int npass_vcf = (nb + 63) >> 6;                        // passes from here to the end of the VCF
  const int npass_own = npass_vcf < K3_NPASS ? npass_vcf : K3_NPASS;
  // One pass: a prefix sum for both lists (FP count in the low half, TP count in the high half: a pass holds 512 records), the
  // entries of the lists that still want them, complete chunks out
  auto pass = [&](uint32_t kb, uint32_t tb, int p, bool wantF, bool wantT) {
    const uint32_t wt = tb, wf = kb & ~tb;
    const uint32_t c = (uint32_t)__popc(wf) | ((uint32_t)__popc(wt) << 16);
    uint32_t tot;
    const uint32_t ex = k3_scan(c, tot) - c;
    const uint32_t base = (uint32_t)(rb + p * K3_PASS + 8 * lane);
    if (wantF && (tot & 0xffffu)) k3_emit(F.buf, s_lut, F.n + (ex & 0xffffu), wf, base);
    if (wantT && (tot >> 16)) k3_emit(T.buf, s_lut, T.n + (ex >> 16), wt, base);
    asm volatile("" ::: "memory");
    if (wantF) { F.n += tot & 0xffffu; k3_drain<K3_FCH_LOG>(F, lane); }
    if (wantT) { T.n += tot >> 16; k3_drain<K3_TCH_LOG>(T, lane); }
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int u = 0; u < K3_NPASS; ++u) {
    if (u >= npass_own) break;
    pass(mk[u], mq[u], u, true, true);
  }
  // The wave's last chunk of a list is completed from the tile BEHIND its own (the successor leaves the entries in front of
  // its first chunk boundary alone, as far as this wave reaches: k3_own), whose mask bytes are already here.
  int p = npass_own;
  bool wantF = F.g0 + F.n < F.hi, wantT = T.g0 + T.n < T.hi;
#pragma unroll
  for (int u = K3_NPASS; u < NPRE; ++u) {
    if (p == u && u < npass_vcf && (wantF || wantT)) {
      pass(mk[u], mq[u], u, wantF, wantT);
      wantF = wantF && F.g0 + F.n < F.hi; wantT = wantT && T.g0 + T.n < T.hi;
      ++p;
    }
  }
  // what is left below hi: where the list ends, or the wave's reach did (a ragged chunk)
  {
    const uint32_t l = F.lo > F.g0 ? F.lo : F.g0, h = F.hi < F.g0 + F.n ? F.hi : F.g0 + F.n;
    if (h > l) k3_store_part(F, 0u, l, h, lane);
  }
  {
    const uint32_t l = T.lo > T.g0 ? T.lo : T.g0, h = T.hi < T.g0 + T.n ? T.hi : T.g0 + T.n;
    if (h > l) k3_store_part(T, 0u, l, h, lane);
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
  if (S.shuffled == 1) src = (int64_t)(((unsigned __int128)(uint64_t)i * S.perm_a + S.perm_b) % (uint64_t)vd.n);
  else if (S.shuffled > 1) {
    // a VCF of R contigs (CHROM is never compared: SURVEY Q1): R ascending runs one behind the other, run c = the generated
    // records c, c + R, c + 2 R, ... -- every run spans the whole position range, the multiset of records is the sorted VCF's
    const int64_t R = S.shuffled, q = vd.n / R, rem = vd.n % R;
    int64_t c, t;
    if (i < rem * (q + 1)) { c = i / (q + 1); t = i % (q + 1); }
    else { const int64_t i2 = i - rem * (q + 1); c = rem + i2 / q; t = i2 % q; }
    src = t * R + c;
  }
  int32_t p, r, a;
  float q;
  uint8_t f;
  const uint64_t tseed = S.per_vcf_truth ? (uint64_t)(uint32_t)vd.pad : S.truth_seed;
  synth_record(S.genome_len, vd.n, S.truth_n, tseed, seed, src, &p, &r, &a, &q, &f, S.indel_pct);
  const int64_t g = vd.off + i;
  S.pos[g] = p; S.ref[g] = r; S.alt[g] = a; S.qual[g] = q; S.flags[g] = f;
}

// ---------------------------------------------------------------------------
// radix sort path (unsorted VCFs), batched over segments: every unsorted VCF of a chunk is
// one segment; one launch per step serves all of them.  Stable LSD passes over 8-bit digits
// of key = pos, payload = original record index.  Sort tiles are numbered over the chunk;
// tile_seg maps a tile to its segment.
// ---------------------------------------------------------------------------
// allele-extended batches: the sorted copies also need the allele codes behind the keys
__global__ __launch_bounds__(256) void k_sort_gather_alleles(const SortSeg* segs, const int32_t* tile_seg, const uint32_t* perm,
                                                             const int32_t* src_ref, const int32_t* src_alt, int32_t* dst_ref,
                                                             int32_t* dst_alt) {
  const SortSeg sg = segs[tile_seg[blockIdx.x]];
  const int64_t base = (int64_t)((int)blockIdx.x - sg.tile0) * SORT_TILE;
  for (int k = 0; k < SORT_TILE / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < sg.n) {
      const int64_t g = sg.src_off + (int64_t)perm[sg.koff + i];
      dst_ref[sg.dst_off + i] = src_ref[g];
      dst_alt[sg.dst_off + i] = src_alt[g];
    }
  }
}

// XCD-aware tile order for kernels whose neighbouring tiles write neighbouring pieces of the same cache lines:
// workgroups go round-robin to the 8 XCDs, each with its own L2; giving each XCD a CONTIGUOUS range of tiles lets
// one L2 merge the pieces into whole lines before they leave for HBM.
__device__ __forceinline__ int xcd_contiguous_block() {
  const int nblk = (int)gridDim.x, xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
  const int per = nblk >> 3, rem = nblk & 7;
  return xcd * per + (xcd < rem ? xcd : rem) + jx;
}

// the digit of a key in one pass of the LSD sort: 8 bits at `shift`
__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift) { return (key >> shift) & 255u; }

// per-tile digit histogram of a segment: hist[hoff + digit * ntiles + tile]
// FIRST: the first pass reads the position column itself (key bits >= 4 are the position) and also collects the OR of
// all keys, which tells the host how many digits are in use.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_sort_hist(const SortSeg* segs, const int32_t* tile_seg, const uint32_t* keys, int shift,
                                                   uint32_t* hist, const int32_t* pos_col, uint32_t* orbits) {
  __shared__ uint32_t s[256];
  const int bid = xcd_contiguous_block();   // neighbouring tiles write neighbouring 4-byte counters of every digit's row
  const SortSeg sg = segs[tile_seg[bid]];
  const int t = bid - sg.tile0;
  const int tid = (int)threadIdx.x;
  s[tid] = 0;
  uint32_t acc = 0;
  __syncthreads();
  const int64_t base = (int64_t)t * SORT_TILE;
  // koff is a multiple of 64 and the chunk arrays are padded to it: whole uint4 loads, tail masked
  for (int k = 0; k < SORT_TILE / 1024; ++k) {
    const int64_t i = base + (int64_t)k * 1024 + tid * 4;
    if (i < sg.n) {
      uint4 v;
      if (FIRST) {   // every VCF starts on a 256-record boundary of the padded columns: whole uint4 loads here too
        v = *reinterpret_cast<const uint4*>(pos_col + sg.src_off + i);
        v.x <<= 4; v.y <<= 4; v.z <<= 4; v.w <<= 4;
        acc |= v.x | (i + 1 < sg.n ? v.y : 0u) | (i + 2 < sg.n ? v.z : 0u) | (i + 3 < sg.n ? v.w : 0u);
      } else {
        v = *reinterpret_cast<const uint4*>(keys + sg.koff + i);
      }
      atomicAdd(&s[digit_of(v.x, shift)], 1u);
      if (i + 1 < sg.n) atomicAdd(&s[digit_of(v.y, shift)], 1u);
      if (i + 2 < sg.n) atomicAdd(&s[digit_of(v.z, shift)], 1u);
      if (i + 3 < sg.n) atomicAdd(&s[digit_of(v.w, shift)], 1u);
    }
  }
  __syncthreads();
  hist[sg.hoff + (size_t)tid * sg.ntiles + t] = s[tid];
  if (FIRST) {
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o);
    // the OR saturates after a few tiles: only waves that still add a bit touch the shared word
    if ((tid & 63) == 0 && (acc & ~__hip_atomic_load(orbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) atomicOr(orbits, acc);
  }
}

// exclusive scan over each segment's digit-major histogram (one workgroup per segment): every
// thread scans 16 consecutive counters in registers, the 256 thread totals are scanned with wave
// shuffles, so one round of 4096 counters costs two barriers
__global__ __launch_bounds__(256) void k_sort_scan(const SortSeg* segs, uint32_t* hist_all) {
  constexpr int PER = 16;
  __shared__ uint32_t s_wave[4];
  const SortSeg sg = segs[blockIdx.x];
  uint32_t* hist = hist_all + sg.hoff;
  const int64_t total = (int64_t)sg.ntiles * 256;   // multiple of 256, hoff too: 16-counter runs are whole and 16-byte aligned
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t carry = 0;
  for (int64_t base = 0; base < total; base += 256 * PER) {
    const int64_t i0 = base + (int64_t)tid * PER;
    const bool in = i0 < total;
    uint32_t x[PER];
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
      uint4 v = in ? *reinterpret_cast<const uint4*>(hist + i0 + 4 * q) : make_uint4(0u, 0u, 0u, 0u);
      x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const uint32_t t = x[k]; x[k] = sum; sum += t; }   // exclusive inside the run
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, btot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t t = s_wave[w]; woff += w < wave ? t : 0u; btot += t; }
    const uint32_t ex = carry + woff + incl - sum;
    if (in) {
#pragma unroll
      for (int q = 0; q < PER / 4; ++q) {
        uint4 v;
        v.x = ex + x[4 * q]; v.y = ex + x[4 * q + 1]; v.z = ex + x[4 * q + 2]; v.w = ex + x[4 * q + 3];
        *reinterpret_cast<uint4*>(hist + i0 + 4 * q) = v;
      }
    }
    carry += btot;
    __syncthreads();
  }
}

// stable scatter: wave w of the tile owns SORT_TILE/4 consecutive keys and walks them 64 at a
// time; the rank inside a wave step comes from an 8-ballot multisplit (peers = lanes with the
// same digit).  The tile is then reordered by digit in LDS so that each digit's run leaves as
// ONE contiguous, coalesced write per array instead of dribbling out 4 bytes at a time.
// `infs` (second payload) may be null; with `final_dst` the keys and infos of the last pass land at the
// segment's place in the scratch batch (dst_off) instead of the chunk arrays (koff).
// FIRST: the first pass packs its (key, info, index) triples straight from the columns -- there is no separate
// packing pass -- and, having every record's info in hand in INPUT order, writes the VCF's kept mask (kept = live
// and PASS needs no truth set) and clears its TP mask: after the join only the TP bits travel back.
template <bool FIRST>
__global__ __launch_bounds__(256) void k_sort_scatter(const SortSeg* segs, const int32_t* tile_seg, const uint32_t* keys,
                                                      const uint32_t* infs, const uint32_t* vals, int shift, const uint32_t* hist,
                                                      uint32_t* okeys, uint32_t* oinfs, uint32_t* ovals, int final_dst, SortCols src,
                                                      int n_bins, int ext, uint64_t* mask_pass, uint64_t* mask_tp) {
  __shared__ uint32_t s_cnt[4][256];   // count of digit d in wave w's chunk (running during the ranking)
  __shared__ uint32_t s_loc[256];      // tile-local start of digit d's run
  __shared__ uint32_t s_glob[256];     // global start of this tile's digit-d run, minus s_loc[d]
  __shared__ uint32_t s_scan[256];
  __shared__ __attribute__((aligned(16))) uint32_t s_k[SORT_TILE], s_i[SORT_TILE], s_v[SORT_TILE];
  const int bid = xcd_contiguous_block();   // neighbouring tiles write neighbouring 32-byte pieces of every digit's run
  const SortSeg sg = segs[tile_seg[bid]];
  const int t = bid - sg.tile0;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < 4 * 256; i += 256) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t tbase = (int64_t)t * SORT_TILE;
  const int64_t wbase = tbase + (int64_t)wave * (SORT_TILE / 4);
  constexpr int STEPS = SORT_TILE / 4 / 64;
  uint32_t kk[STEPS], vv[STEPS], ii[STEPS], rk[STEPS];
  uint32_t* const stK = s_k + wave * (SORT_TILE / 4);
  uint32_t* const stI = s_i + wave * (SORT_TILE / 4);
  uint32_t* const stV = s_v + wave * (SORT_TILE / 4);
  // Dword loads are bound by the rate of memory instructions, not by bytes: fetch the wave's 512 entries with
  // 16-byte loads (4 consecutive entries per lane), park them in the LDS area the reorder uses later, and take
  // them back one per lane in step order (the stable ranking needs lane = consecutive entry).
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  if (FIRST) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < SORT_TILE / 4 / 256; ++j) {
      const int64_t i4 = wbase + j * 256 + lane * 4;
      v4u k4 = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, f4 = {0u, 0u, 0u, 0u};
      if (i4 < sg.n) {   // whole 16-byte pieces: the columns are padded past every VCF
        const int64_t g = sg.src_off + i4;
        const v4i p = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src.pos + g));
        const v4i r = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src.ref + g));
        const v4i a = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src.alt + g));
        const v4f q = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src.qual + g));
        const uint32_t f = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src.flags + g));
        uint32_t kx[4], ix[4];
        if (ext) {
          pack_record<true>(p.x, r.x, a.x, q.x, f, n_bins, kx[0], ix[0]);
          pack_record<true>(p.y, r.y, a.y, q.y, f >> 8, n_bins, kx[1], ix[1]);
          pack_record<true>(p.z, r.z, a.z, q.z, f >> 16, n_bins, kx[2], ix[2]);
          pack_record<true>(p.w, r.w, a.w, q.w, f >> 24, n_bins, kx[3], ix[3]);
        } else {
          pack_record<false>(p.x, r.x, a.x, q.x, f, n_bins, kx[0], ix[0]);
          pack_record<false>(p.y, r.y, a.y, q.y, f >> 8, n_bins, kx[1], ix[1]);
          pack_record<false>(p.z, r.z, a.z, q.z, f >> 16, n_bins, kx[2], ix[2]);
          pack_record<false>(p.w, r.w, a.w, q.w, f >> 24, n_bins, kx[3], ix[3]);
        }
        k4.x = kx[0]; k4.y = kx[1]; k4.z = kx[2]; k4.w = kx[3];
        f4.x = ix[0]; f4.y = ix[1]; f4.z = ix[2]; f4.w = ix[3];
      }
      *reinterpret_cast<v4u*>(&stK[j * 256 + lane * 4]) = k4;
      *reinterpret_cast<v4u*>(&stI[j * 256 + lane * 4]) = f4;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int j = 0; j < SORT_TILE / 4 / 256; ++j) {
      const int64_t i4 = wbase + j * 256 + lane * 4;
      v4u k4 = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, v4 = {0u, 0u, 0u, 0u}, f4 = {0u, 0u, 0u, 0u};
      if (i4 < sg.n) {   // whole 16-byte pieces: every segment of the chunk arrays is padded to 64 entries
        k4 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(keys + sg.koff + i4));
        v4 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(vals + sg.koff + i4));
        if (infs) f4 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(infs + sg.koff + i4));
      }
      *reinterpret_cast<v4u*>(&stK[j * 256 + lane * 4]) = k4;
      *reinterpret_cast<v4u*>(&stV[j * 256 + lane * 4]) = v4;
      *reinterpret_cast<v4u*>(&stI[j * 256 + lane * 4]) = f4;
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int64_t i = wbase + s * 64 + lane;
    const bool valid = i < sg.n;
    if (FIRST) {
      kk[s] = 0xffffffffu; vv[s] = 0u; ii[s] = 0u;
      if (valid) { kk[s] = stK[s * 64 + lane]; ii[s] = stI[s * 64 + lane]; vv[s] = (uint32_t)i; }
      const uint64_t bp = ballot64(valid && (ii[s] & I_KEPT) != 0u);
      const int64_t i0 = i - lane;   // the wave's 64 consecutive records start at a multiple of 64
      if (lane == 0 && i0 < sg.n) {
        mask_pass[(sg.src_off + i0) >> 6] = bp;
        mask_tp[(sg.src_off + i0) >> 6] = 0ull;
      }
    } else {
      kk[s] = valid ? stK[s * 64 + lane] : 0xffffffffu;
      vv[s] = valid ? stV[s * 64 + lane] : 0u;
      ii[s] = valid ? stI[s * 64 + lane] : 0u;
    }
    const uint32_t d = digit_of(kk[s], shift);
    // peers = valid lanes with my digit.  Kept as two 32-bit halves and accumulated as "differs from me in some bit":
    // per bit one sign-extending bit-field extract (0 / -1), one ballot, two XORs, two ORs -- the 64-bit select form
    // the compiler made of `peers &= bit ? m : ~m` took nine VALU per bit, and this kernel is VALU-bound.
    uint32_t np_lo = 0, np_hi = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int32_t t = ((int32_t)(d << (31 - b))) >> 31;          // all ones if bit b of my digit is set
      const uint64_t m = ballot64(t != 0);
      np_lo |= (uint32_t)m ^ (uint32_t)t;
      np_hi |= (uint32_t)(m >> 32) ^ (uint32_t)t;
    }
    const uint64_t vm = ballot64(valid);
    const uint32_t p_lo = (uint32_t)vm & ~np_lo, p_hi = (uint32_t)(vm >> 32) & ~np_hi;
    const uint32_t r = __builtin_amdgcn_mbcnt_hi(p_hi, __builtin_amdgcn_mbcnt_lo(p_lo, 0u));   // peers in lower lanes
    uint32_t basec = 0;
    if (valid) basec = s_cnt[wave][d];   // every peer reads before the leader writes (same wave, in order)
    rk[s] = basec + r;
    if (valid && r == 0) s_cnt[wave][d] = basec + (uint32_t)__popc(p_lo) + (uint32_t)__popc(p_hi);
  }
  __syncthreads();
  // thread d: digit d's count over the 4 waves -> exclusive scan over digits = tile-local run starts
  uint32_t c0 = s_cnt[0][tid], c1 = s_cnt[1][tid], c2 = s_cnt[2][tid], c3 = s_cnt[3][tid];
  const uint32_t tot = c0 + c1 + c2 + c3;
  uint32_t incl = tot;   // inclusive scan over the 256 digits: shuffles inside the wave, one LDS hop across the four
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(incl, o);
    if (lane >= o) incl += y;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  uint32_t woff = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) woff += w < wave ? s_scan[w] : 0u;
  const uint32_t loc = woff + incl - tot;
  s_loc[tid] = loc;
  s_glob[tid] = hist[sg.hoff + (size_t)tid * sg.ntiles + t] - loc;
  // wave w's block inside digit d's run
  s_cnt[0][tid] = loc; s_cnt[1][tid] = loc + c0; s_cnt[2][tid] = loc + c0 + c1; s_cnt[3][tid] = loc + c0 + c1 + c2;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int64_t i = wbase + s * 64 + lane;
    if (i < sg.n) {
      const uint32_t d = digit_of(kk[s], shift);
      const uint32_t lp = s_cnt[wave][d] + rk[s];
      s_k[lp] = kk[s]; s_i[lp] = ii[s]; s_v[lp] = vv[s];
    }
  }
  __syncthreads();
  const int nvalid = (int)((sg.n - tbase) < SORT_TILE ? (sg.n - tbase) : SORT_TILE);
  const int64_t kbase = final_dst ? sg.dst_off : sg.koff;
  for (int idx = tid; idx < nvalid; idx += 256) {
    const uint32_t k = s_k[idx];
    const uint32_t g = s_glob[digit_of(k, shift)] + (uint32_t)idx;
    okeys[kbase + g] = k;
    if (FIRST || infs) oinfs[kbase + g] = s_i[idx];
    ovals[sg.koff + g] = s_v[idx];
  }
}


// ---------------------------------------------------------------------------
// The bucket path for unsorted VCFs: no sort at all.
//
// k_bucket_scatter: ONE pass over the columns puts every live record, packed to 8 bytes (qmvt_dev.h "bucket entry"), into
// the bucket of its position range (the segment's top eight position bits in use).  Nothing downstream needs an order
// inside a bucket, so there is no histogram pass, no scan and no stable ranking: a tile counts its digits in LDS with one
// returning atomic per record, reserves room for each digit's run with ONE global atomic on the bucket's cursor, reorders
// the tile by digit in LDS and writes every run as one contiguous piece.  Bucket regions have a fixed capacity (what
// k_classify_hash can hold anyway); a bucket has eight sub-regions with a cursor each, and tile g of the launch fills
// sub-region g % 8: workgroups go round-robin to the eight XCDs, so all pieces of a sub-region come from ONE XCD, whose
// L2 merges neighbouring pieces into whole lines before they leave for HBM.  The kernel has every record's info in
// input order in hand: it also writes the VCF's kept mask (kept = live and PASS needs no truth set) and clears its TP mask.
// ---------------------------------------------------------------------------
#ifndef BK_WAVES_PER_EU
#define BK_WAVES_PER_EU 6
#endif
// L2: the segment is one partition of a large VCF (two-level path): its records come as level-1 entries (P.l1_ent + sg.koff,
// sg.n of them, no holes) instead of columns, keys are relative to the partition (sg.key_base), the kept mask was written
// by the first level.
// EXT (allele-extended batches): every record with valid allele codes is live; those whose REF / ALT are not two single bases
// leave in a second stream of 16-byte entries (the ordinary entry with the position's first key, then the two codes), written
// straight to their bucket's second region -- k_join_ext joins them exactly, k_join_direct the single-base ones.
// NB = 512 (VCFs in partitions, SortSeg.part & 4): the tile's segment stands for TWO neighbouring partitions of its VCF -- 512
// buckets whose cursors, regions and rows lie one behind the other -- so that the columns are read once for both.  (Round 6 tried
// eight partitions per pass with a 2 048-digit instantiation, for configs[3]'s shuffled 10 M-record VCFs: a tile then leaves 21
// bytes per bucket, the L2 evicts such pieces before the next tile completes their lines, and 4.1x the entries' bytes reach HBM --
// 3.5 ms per 1.6e8 records against the two levels' 1.34: profiles/r06_pmc_scatter2048_not_kept.json.  Not kept.)
template <bool L2, bool EXT, int NB = 256>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(EXT ? 5 : BK_WAVES_PER_EU, 8))) void k_bucket_scatter(BucketScatterParams P) {
  constexpr int PER = BK_TILE / 512;          // records per thread, in groups of four consecutive ones
  static_assert(BK_TILE % 2048 == 0 && PER >= 4, "whole 16-byte loads, 256 records per wave and group");
  static_assert(NB == 256 || (NB == 512 && !L2), "one digit per thread at most");
  typedef typename std::conditional<NB == 256, uint8_t, uint16_t>::type digit_t;
  __shared__ uint32_t s_cnt[NB];              // records of digit d in the tile (running during the ranking)
  __shared__ uint32_t s_loc[NB];              // tile-local start of digit d's run
  __shared__ int32_t s_glob[NB];              // place of digit d's run in its sub-region, minus s_loc[d]
  __shared__ uint32_t s_scan[NB / 64 + 1];
  __shared__ __attribute__((aligned(16))) uint64_t s_e[BK_TILE];
  __shared__ digit_t s_d[BK_TILE];
  __shared__ uint32_t s_cntx[EXT ? NB : 1];      // second stream: records of digit d in the tile, then where its run starts in the sub-region
  __shared__ uint32_t s_flut[16];                // flag_info of the sixteen flag nibbles (pack_record_fast)
  __shared__ uint32_t s_hall[NB / 256][SEG_HIST_WORDS];   // (P.seg_hist) the tile's first-stream entries by bin + 1, per segment the tile fills
  const int bid = (int)blockIdx.x + P.tile_base;
  const int seg = P.tile_seg[bid];
  const SortSeg sg = P.segs[seg];
  const int sub = bid & (HB_SUBS - 1);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < NB; d += 512) { s_cnt[d] = 0u; if (EXT) s_cntx[d] = 0u; }
  const bool count_all = P.seg_hist != nullptr;
  if (count_all) for (int i = tid; i < (NB / 256) * SEG_HIST_WORDS; i += 512) (&s_hall[0][0])[i] = 0u;
  uint32_t top_all = 0, top_all2 = 0;            // records of the saturated top bin (the lanes of a wave would serialise on its one address)
  const uint32_t nbins = (uint32_t)P.n_bins;
  // one first-stream entry by bin + 1 (b1 <= n_bins <= 256); digit >= 256: the entry belongs to the next segment (NB = 512)
  auto count = [&](uint32_t b1, uint32_t d) {
    if (NB == 512 && d >= 256u) { if (b1 == nbins) ++top_all2; else atomicAdd(&s_hall[NB / 256 - 1][b1], 1u); }
    else { if (b1 == nbins) ++top_all; else atomicAdd(&s_hall[0][b1], 1u); }
  };
  if (!L2 && tid >= 256 && tid < 272) s_flut[tid - 256] = flag_info((uint32_t)(tid - 256));
  const uint32_t dlim = NB >= 512 && (sg.part & 4) ? (uint32_t)HB_BUCKETS * (uint32_t)((sg.part >> 4) & 15) : (uint32_t)HB_BUCKETS;   // buckets this tile's segment stands for (part bits 4..7: partitions of its group)
  // a whole VCF in one segment (part 0): the host joins and sums only the sg.nbk buckets up to the highest position the optimistic
  // pass SAW -- and that pass leaves a span at its first round out of order, so later records can hold higher positions: a record
  // beyond the estimate flags the VCF (the radix sort redoes it) instead of landing in a bucket nobody looks at
  const uint32_t olim = sg.part == 0 ? (uint32_t)sg.nbk : dlim;
  __syncthreads();
  const int64_t tbase = (int64_t)(bid - sg.bk_tile0) * BK_TILE;
  const uint32_t shift = (uint32_t)sg.pad;
  uint64_t ent[PER];
  uint32_t dr[PER];   // digit << 16 | rank inside the tile's digit; 0xffffffff: the record is not live
  uint32_t drx[EXT ? PER : 1];   // the same for the second stream (the entry itself is rebuilt from ent[] and the columns when it is stored)
  uint32_t segfl = 0;
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4i p[PER / 4], r[PER / 4], a[PER / 4];
  v4f q[PER / 4];
  uint32_t f[PER / 4];
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  if (L2) {
    // level-1 entries -> bucket entries: four per thread and group, two 16-byte loads (the partition's region starts on a
    // 16-byte boundary and holds sg.n entries without holes; the array is padded past its end)
    uint32_t hb1 = 0xffffffffu, hb2 = 0xffffffffu, hb3 = 0xffffffffu;   // where the entries of the VCF's 2nd, 3rd, 4th run of 2^24 records begin
    if (P.l1_half) { hb1 = P.l1_half[4 * seg + 1]; hb2 = P.l1_half[4 * seg + 2]; hb3 = P.l1_half[4 * seg + 3]; }
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) {
      const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
      v4u e0 = {0u, 0u, 0u, 0u}, e1 = {0u, 0u, 0u, 0u};
      if (i4 < sg.n) {
        const v4u* src = reinterpret_cast<const v4u*>(P.l1_ent + sg.koff + i4);
        e0 = __builtin_nontemporal_load(src);
        e1 = __builtin_nontemporal_load(src + 1);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = 4 * j + u;
        const uint32_t lo = u == 0 ? e0[0] : u == 1 ? e0[2] : u == 2 ? e1[0] : e1[2];
        const uint32_t hi = u == 0 ? e0[1] : u == 1 ? e0[3] : u == 2 ? e1[1] : e1[3];
        dr[k] = 0xffffffffu;
        ent[k] = 0ull;
        const uint32_t v27 = lo & ((1u << P2_SHIFT) - 1u);
        const uint32_t inf13 = ((lo >> P2_SHIFT) | (hi << (32 - P2_SHIFT))) & 0x1fffu;
        if (i4 + u < sg.n && inf13 != P2_DEAD) {
          const uint32_t d = v27 >> shift;   // shift = DJ_MAX_SHIFT here: < 256
          const uint32_t v = v27 - (d << shift);
          const uint32_t ii = (uint32_t)(i4 + u);
          const uint32_t half = (ii >= hb1 ? 1u : 0u) + (ii >= hb2 ? 1u : 0u) + (ii >= hb3 ? 1u : 0u);
          ent[k] = (uint64_t)v | ((uint64_t)inf13 << 24) | ((uint64_t)((hi >> 8) + (half << P2_INDEX_BITS)) << 37);
          dr[k] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
          if (count_all) count(inf13 & 0x1ffu, d);
        }
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < PER / 4; ++j) {   // all loads first: whole 16-byte pieces, the columns are padded past every VCF
    const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
    f[j] = 0u;
    if (i4 < sg.n) {
      const int64_t g = sg.src_off + i4;
      p[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.pos + g));
      r[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.ref + g));
      a[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.alt + g));
      q[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(P.qual + g));
      f[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(P.flags + g));
    }
  }
  const float nbm1f = (float)(P.n_bins - 1);
#pragma unroll
  for (int j = 0; j < PER / 4; ++j) {
    const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
    uint32_t kept = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * j + u;
      dr[k] = 0xffffffffu;
      if (EXT) drx[k] = 0xffffffffu;
      ent[k] = 0ull;
      if (i4 + u < sg.n) {
        uint32_t key, inf;
        const uint32_t f4x4 = ((f[j] >> (8 * u)) & 15u) << 2;
        if (EXT) pack_record_fast_ext(p[j][u], r[j][u], a[j][u], q[j][u], f4x4, nbm1f, s_flut, key, inf);
        else pack_record_fast(p[j][u], r[j][u], a[j][u], q[j][u], f4x4, nbm1f, s_flut, key, inf);
        segfl |= ((uint32_t)p[j][u] >> 28) ? SPANF_BADPOS : 0u;
        if (inf & I_LIVE) {
          kept |= ((inf >> 16) & 1u) << u;
          const uint32_t rel = key - sg.key_base;   // (key_base = 0 unless the segment is a partition of its VCF)
          const uint32_t d = rel >> shift;
          if (sg.part != 0 && (key < sg.key_base || ((sg.part & 3) == 1 && d >= dlim))) {
            // another partition of the same VCF takes this record (every partition, or pair of partitions, reads all of the VCF's columns)
          } else if (d >= olim) {
            segfl |= SPANF_OVERFLOW;   // a position above what the optimistic pass saw of this VCF: the radix sort redoes it
          } else if (!EXT || (uint32_t)(r[j][u] | a[j][u]) < 4u) {
            const uint32_t v = rel - (d << shift);   // < 2^24: shift <= 24
            ent[k] = (uint64_t)v | ((uint64_t)((inf & 0xfffu) | ((inf >> 12) & 0x1000u)) << 24) | ((uint64_t)(uint32_t)(i4 + u) << 37);
            dr[k] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
            if (count_all) count(inf & I_BIN1, d);
          } else {   // not two single bases: the second stream (the key's nibble is a hash there: the entry carries the position's first key)
            const uint32_t v = (rel & ~15u) - (d << shift);
            ent[k] = (uint64_t)v | ((uint64_t)((inf & 0xfffu) | ((inf >> 12) & 0x1000u)) << 24) | ((uint64_t)(uint32_t)(i4 + u) << 37);
            drx[k] = (d << 16) | atomicAdd(&s_cntx[d], 1u);
          }
        }
      }
    }
    // natural-order mask words: 8 lanes x 4 records = one 32-bit word
    const uint32_t wp = or_reduce8(kept << (4u * (uint32_t)(lane & 7)));
    const int64_t w0 = i4 - 4 * (lane & 7);   // first record of the word
    if ((lane & 7) == 7 && w0 < ((sg.n + 255) & ~(int64_t)255)) {   // to the end of the VCF's padding: the masks are read in 64-bit words
      P.mask_pass[(sg.src_off + w0) >> 5] = wp;
      P.mask_tp[(sg.src_off + w0) >> 5] = 0u;
    }
  }
  }   // !L2
  if (segfl) atomicOr(&P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg], segfl);
  if (count_all) {
    if (ballot64(top_all != 0u)) { top_all = wave_sum(top_all); if (lane == 0) atomicAdd(&s_hall[0][nbins], top_all); }
    if (NB == 512 && ballot64(top_all2 != 0u)) { top_all2 = wave_sum(top_all2); if (lane == 0) atomicAdd(&s_hall[NB / 256 - 1][nbins], top_all2); }
  }
  __syncthreads();
  if (NB <= 512 && P.seg_maxd && tid < NB) {   // 1 + the highest bucket this tile fills, per wave: one atomic each
    uint32_t m = (s_cnt[tid] || (EXT && s_cntx[tid])) ? (uint32_t)(tid & 255) + 1u : 0u;   // (either stream)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)m, o); m = y > m ? y : m; }
    // (digits 256..511: the next segment.)  A look first: hundreds of tiles of one segment are in flight together, and atomics on ONE
    // word queue up at the memory side -- unchecked they doubled the second level's scatter (0.64 -> 1.13 ms per 1.6e8 records);
    // a stale value only costs an atomic that changes nothing
    if (lane == 0 && m > __hip_atomic_load(&P.seg_maxd[seg + (tid >> 8)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&P.seg_maxd[seg + (tid >> 8)], m);
  }
  if (count_all) {   // the tile's counts join its segment's (the next segment's for the digits 256..511)
    for (int i = tid; i < (NB / 256) * SEG_HIST_WORDS; i += 512) {
      const uint32_t c = (&s_hall[0][0])[i];
      if (c) atomicAdd(&P.seg_hist[(size_t)(seg + i / SEG_HIST_WORDS) * SEG_HIST_WORDS + (size_t)(i % SEG_HIST_WORDS)], c);
    }
  }
  // thread d: exclusive scan over the digit counts = tile-local run starts; room for the run in the bucket's sub-region.
  uint32_t cnt = 0, incl = 0;
  if (tid < NB) {
    cnt = s_cnt[tid];
    incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (lane == 63) s_scan[wave] = incl;
  }
  __syncthreads();
  // The room for the runs is asked for (one returning atomic per digit in use) and not waited for: reordering the tile in LDS
  // needs the tile-local run starts only, so the cursors' round trip passes behind it.
  uint32_t g = 0, loc = 0;
  uint32_t cntx2 = 0, gx2 = 0;   // NB = 512: the thread of digit d reserves for both streams
  if (tid < NB) {
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < NB / 64 - 1; ++w) woff += w < wave ? s_scan[w] : 0u;
    loc = woff + incl - cnt;
    s_loc[tid] = loc;
    if (cnt) g = atomicAdd(&P.cursor[((size_t)seg * HB_BUCKETS + tid) * HB_SUBS + sub], cnt);   // (digits 256..511: the next partition's cursors follow)
    if (tid == NB - 1) s_scan[NB / 64] = loc + cnt;
    if (EXT && NB == 512) {
      cntx2 = s_cntx[tid];
      if (cntx2) gx2 = atomicAdd(&P.xcursor[((size_t)seg * HB_BUCKETS + tid) * HB_SUBS + sub], cntx2);
    }
  } else if (EXT) {   // the other four waves: room for the second stream's runs (no reordering in LDS: each entry is stored where it belongs)
    cnt = s_cntx[tid - 256];
    if (cnt) g = atomicAdd(&P.xcursor[((size_t)seg * HB_BUCKETS + (tid - 256)) * HB_SUBS + sub], cnt);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (dr[k] != 0xffffffffu) {
      const uint32_t d = dr[k] >> 16;
      const uint32_t lp = s_loc[d] + (dr[k] & 0xffffu);
      s_e[lp] = ent[k];
      s_d[lp] = (digit_t)d;
    }
  }
  if (tid < NB || EXT) {
    if (tid < NB) {
      s_glob[tid] = (int32_t)g - (int32_t)loc;   // (a run that does not fit its sub-region: below)
    } else {
      if (cnt && g + cnt > (uint32_t)sg.bk_cap) atomicOr(&P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg], SPANF_OVERFLOW);
      s_cntx[tid - 256] = g;
    }
    if (EXT && NB == 512) {
      if (cntx2 && gx2 + cntx2 > (uint32_t)sg.bk_cap) atomicOr(&P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg], SPANF_OVERFLOW);
      s_cntx[tid] = gx2;
    }
  }
  __syncthreads();
  if (EXT) {
    typedef unsigned long long v2ull __attribute__((ext_vector_type(2)));
    v2ull* xout = reinterpret_cast<v2ull*>(P.xent) + sg.bk_off;   // (the second stream's regions are laid out like the first's)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (drx[k] != 0xffffffffu) {
        const uint32_t d = drx[k] >> 16;
        const uint32_t w = s_cntx[d] + (drx[k] & 0xffffu);
        if (w < (uint32_t)sg.bk_cap) {
          v2ull e;
          e.x = ent[k];
          e.y = (unsigned long long)(uint32_t)r[k >> 2][k & 3] | ((unsigned long long)(uint32_t)a[k >> 2][k & 3] << 32);
          xout[((size_t)d * HB_SUBS + sub) * (size_t)sg.bk_cap + (size_t)w] = e;
        }
      }
    }
  }
  const int total = (int)s_scan[NB / 64];
  uint64_t* out = P.ent + sg.bk_off;
  for (int idx = tid; idx < total; idx += 512) {
    const uint32_t d = s_d[idx];
    const int32_t w = s_glob[d] + idx;
    if (w < sg.bk_cap) out[((size_t)d * HB_SUBS + sub) * (size_t)sg.bk_cap + (size_t)w] = s_e[idx];   // beyond: spilled below
  }
  // A run that does not fit its sub-region SPILLS into the bucket's other seven (round 6).  A sub-region takes every eighth tile:
  // even for shuffled records, but a VCF sorted per contig sends a bucket a few LONG pieces -- one per contig -- and three of them
  // on one sub-region were an overflow that sent the whole chunk to the radix sort (2e10 /s instead of 1e11).  The part of the run
  // that lies below the sub-region's capacity is stored above (every slot below min(cursor, capacity) is somebody's: the joins clamp
  // the cursor and no longer read one above the capacity as an overflow); the rest asks the next sub-regions for room, one atomic
  // each, and its owner thread stores it there itself.  Only a run that finds no room in any of the eight flags the VCF.
  if (tid < NB) {
    const uint32_t cap = (uint32_t)sg.bk_cap;
    if (cnt && g + cnt > cap) {
      uint32_t done = g < cap ? cap - g : 0u;                  // entries of the run stored by the loop above
      for (int hop = 1; hop < HB_SUBS && done < cnt; ++hop) {
        const int s2 = (sub + hop) & (HB_SUBS - 1);
        const uint32_t rest = cnt - done;
        const uint32_t g2 = atomicAdd(&P.cursor[((size_t)seg * HB_BUCKETS + (size_t)tid) * HB_SUBS + s2], rest);
        const uint32_t take = g2 < cap ? (rest < cap - g2 ? rest : cap - g2) : 0u;
        uint64_t* o2 = out + ((size_t)tid * HB_SUBS + s2) * (size_t)cap + (size_t)g2;
        for (uint32_t j = 0; j < take; ++j) o2[j] = s_e[loc + done + j];
        done += take;
      }
      if (done < cnt) atomicOr(&P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg], SPANF_OVERFLOW);   // the bucket itself is full: the radix sort redoes the VCF
    }
  }
}

// ---------------------------------------------------------------------------
// Two-level bucket path, first level (VCFs too large for 256 buckets): k_part_hist counts a VCF's records per partition of
// 2^27 keys (and per sub-region: tile g of the launch counts for, and later fills, sub-region g % 8 -- one cursor per XCD);
// the host turns the counts into exact regions; k_part_scatter packs every record to a level-1 entry (qmvt_dev.h) and drops
// it into its partition's region.  EVERY record with a position in range travels, dead ones (alleles that take no part)
// marked as such, so that the counting pass needs the position column only and the regions have no holes.  The kernel sees
// the records in input order: it writes the kept mask and clears the TP mask, as k_bucket_scatter does on the one-level path.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_part_hist(PartParams P) {
  // a tile of a VCF reaches a handful of partitions: sixteen replicas of every counter (one per lane & 15) keep the lanes
  // of a wave off each other's addresses
  constexpr int REP = 16;
  __shared__ uint32_t s_cnt[P2_PARTS * REP];
  const int bid = (int)blockIdx.x;
  const int seg = P.tile_seg[bid];
  const PartSeg sg = P.segs[seg];
  const int sub = bid & (P2_SUBS - 1);
  const int tid = (int)threadIdx.x;
  s_cnt[tid] = 0u;
  static_assert(P2_PARTS * REP == 512, "one counter per thread");
  __syncthreads();
  const int64_t tbase = (int64_t)(bid - sg.tile0) * BK_TILE;
  typedef int v4i __attribute__((ext_vector_type(4)));
  uint32_t bad = 0;
#pragma unroll
  for (int j = 0; j < BK_TILE / 2048; ++j) {
    const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
    if (i4 < sg.n) {
      const v4i p = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.pos + sg.src_off + i4));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i4 + u < sg.n) {
          const uint32_t pp = (uint32_t)p[u];
          if (pp >= (uint32_t)QM_POS_LIMIT_DEV) bad = 1u;
          else atomicAdd(&s_cnt[(pp >> (P2_SHIFT - 4)) * REP + (tid & (REP - 1))], 1u);
        }
      }
    }
  }
  if (ballot64(bad != 0u) && (tid & 63) == 0) atomicOr(&P.segflags[seg], SPANF_BADPOS);
  __syncthreads();
  if (tid < P2_PARTS) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < REP; ++r) c += s_cnt[tid * REP + ((r + tid) & (REP - 1))];
    if (c) atomicAdd(&P.cnt[(size_t)seg * (P2_PARTS * P2_SUBS) + tid * P2_SUBS + sub], c);
  }
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BK_WAVES_PER_EU, 8))) void k_part_scatter(PartParams P) {
  constexpr int PER = BK_TILE / 512;
  __shared__ uint32_t s_cnt[P2_PARTS];
  __shared__ uint32_t s_loc[P2_PARTS];
  __shared__ int64_t s_glob[P2_PARTS];        // place of partition d's run in the level-1 array, minus s_loc[d]
  __shared__ uint32_t s_total;
  __shared__ uint64_t s_e[BK_TILE];
  __shared__ uint8_t s_d[BK_TILE];
  const int bid = (int)blockIdx.x;
  const int seg = P.tile_seg[bid];
  const PartSeg sg = P.segs[seg];
  const int sub = bid & (P2_SUBS - 1);
  const int tid = (int)threadIdx.x, lane = tid & 63;
  __shared__ uint32_t s_flut[16];             // flag_info of the sixteen flag nibbles (pack_record_fast)
  if (tid < P2_PARTS) s_cnt[tid] = 0u;
  if (tid >= 64 && tid < 80) s_flut[tid - 64] = flag_info((uint32_t)(tid - 64));
  __syncthreads();
  const float nbm1f = (float)(P.n_bins - 1);
  const int64_t tbase = (int64_t)(bid - sg.tile0) * BK_TILE;
  uint64_t ent[PER];
  uint32_t dr[PER];   // partition << 16 | rank inside the tile's partition; 0xffffffff: position out of range (the VCF is refused)
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4i p[PER / 4], r[PER / 4], a[PER / 4];
  v4f q[PER / 4];
  uint32_t f[PER / 4];
#pragma unroll
  for (int j = 0; j < PER / 4; ++j) {
    const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
    f[j] = 0u;
    if (i4 < sg.n) {
      const int64_t g = sg.src_off + i4;
      p[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.pos + g));
      r[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.ref + g));
      a[j] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(P.alt + g));
      q[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(P.qual + g));
      f[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(P.flags + g));
    }
  }
#pragma unroll
  for (int j = 0; j < PER / 4; ++j) {
    const int64_t i4 = tbase + (int64_t)j * 2048 + tid * 4;
    uint32_t kept = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * j + u;
      dr[k] = 0xffffffffu;
      ent[k] = 0ull;
      if (i4 + u < sg.n) {
        uint32_t key, inf;
        pack_record_fast(p[j][u], r[j][u], a[j][u], q[j][u], ((f[j] >> (8 * u)) & 15u) << 2, nbm1f, s_flut, key, inf);
        if (!((uint32_t)p[j][u] >> 28)) {   // (the counting pass has flagged the VCF otherwise)
          const bool live = (inf & I_LIVE) != 0u;
          kept |= ((inf >> 16) & 1u) << u;
          const uint32_t d = key >> P2_SHIFT;
          const uint32_t inf13 = live ? ((inf & 0xfffu) | ((inf >> 12) & 0x1000u)) : P2_DEAD;
          ent[k] = (uint64_t)(key & ((1u << P2_SHIFT) - 1u)) | ((uint64_t)inf13 << P2_SHIFT) | ((uint64_t)(uint32_t)(i4 + u) << 40);
          dr[k] = (d << 16) | atomicAdd(&s_cnt[d], 1u);
        }
      }
    }
    const uint32_t wp = or_reduce8(kept << (4u * (uint32_t)(lane & 7)));
    const int64_t w0 = i4 - 4 * (lane & 7);
    if ((lane & 7) == 7 && w0 < ((sg.n + 255) & ~(int64_t)255)) {
      P.mask_pass[(sg.src_off + w0) >> 5] = wp;
      P.mask_tp[(sg.src_off + w0) >> 5] = 0u;
    }
  }
  __syncthreads();
  if (tid < 64) {   // one wave: exclusive scan over the 32 partition counts, room for each run in its (partition, sub-region)
    const uint32_t cnt = tid < P2_PARTS ? s_cnt[tid] : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= o) incl += y;
    }
    if (tid < P2_PARTS) {
      const uint32_t loc = incl - cnt;
      s_loc[tid] = loc;
      const size_t c = (size_t)seg * (P2_PARTS * P2_SUBS) + (size_t)tid * P2_SUBS + sub;
      uint32_t g = 0;
      if (cnt) g = atomicAdd(&P.cursor[c], cnt);
      s_glob[tid] = sg.ent_off + (int64_t)P.off[(size_t)seg * (P2_PARTS * P2_SUBS + 1) + (size_t)tid * P2_SUBS + sub] + (int64_t)g - (int64_t)loc;
    }
    if (tid == P2_PARTS - 1) s_total = incl;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (dr[k] != 0xffffffffu) {
      const uint32_t d = dr[k] >> 16;
      const uint32_t lp = s_loc[d] + (dr[k] & 0xffffu);
      s_e[lp] = ent[k];
      s_d[lp] = (uint8_t)d;
    }
  }
  __syncthreads();
  const int total = (int)s_total;
  for (int idx = tid; idx < total; idx += 512) P.ent[s_glob[s_d[idx]] + idx] = s_e[idx];   // the regions are exact: nothing can overflow
}

// ---------------------------------------------------------------------------
// k_classify_hash -- the join of one bucket, without an order inside it.
//
// Everything the sorted kernel gets from order -- which records sit at a truth key's position, which kept keys repeat --
// a bucket gets from small tables in LDS, because all records of one position (hence all equal keys, and all matches of a
// truth key) are in the same bucket.  The common path of a record is three LDS operations without a loop (the first
// version probed open-addressing tables with compare-and-swap loops per record and spent 2.65 ms of its 3.3 ms per
// 2.56e8 records in dependent LDS round trips):
//   * truth keys: an exact table (key -> slot; state: best bin of a '.'-ID match, matched-by-kept bit) behind a
//     16 K-bit filter -- one LDS read tells 9 records in 10 that they match nothing;
//   * distinct kept keys outside the truth set (FP_R): a 64 K-bit map, one fetch-OR per record.  A key whose bit was
//     clear is new; bits that were hit twice are marked in a second map, and only the keys on marked bits (a few per
//     cent: true repeats and collisions) meet again in a small exact set in a second pass over the thread's own
//     registers.  Keys on unmarked bits are distinct by construction: the count is exact;
//   * kept records without a comparable key: a small exact set of their own (they are keys of their own, as in k_classify).
// One workgroup of 512 threads per (segment, bucket), 16 records per thread at most, 46 KB of LDS, three workgroups per
// CU.  TP bits go straight to the main batch's TP mask in input order (one 64-bit atomic OR per true positive); the
// bucket's histograms and scalars are a "span" row that k_finalize sums exactly like the rows of k_classify.  A bucket
// that does not fit (a full sub-region, 1 024 truth keys, ...) flags the VCF: the radix sort redoes it.
// ---------------------------------------------------------------------------
constexpr uint32_t HB_EMPTY = 0xffffffffu;
__device__ __forceinline__ uint32_t hb_hash(uint32_t v) { return v * 0x9e3779b1u; }
// returns the slot of v, inserting it when absent (*fresh says which); tables are never full (callers bound the load)
__device__ __forceinline__ uint32_t hb_insert(uint32_t* tab, uint32_t log2n, uint32_t v, bool* fresh) {
  const uint32_t mask = (1u << log2n) - 1u;
  uint32_t s = hb_hash(v) >> (32u - log2n);
  for (;;) {
    const uint32_t old = atomicCAS(&tab[s], HB_EMPTY, v);
    if (old == HB_EMPTY) { *fresh = true; return s; }
    if (old == v) { *fresh = false; return s; }
    s = (s + 1u) & mask;
  }
}
__device__ __forceinline__ int hb_find(const uint32_t* tab, uint32_t log2n, uint32_t v) {
  const uint32_t mask = (1u << log2n) - 1u;
  uint32_t s = hb_hash(v) >> (32u - log2n);
  for (;;) {
    const uint32_t cur = tab[s];
    if (cur == v) return (int)s;
    if (cur == HB_EMPTY) return -1;
    s = (s + 1u) & mask;
  }
}

// one descriptor per (segment, bucket) for k_classify_hash (qmvt_dev.h HashRow): grid = segments, thread = bucket
__global__ __launch_bounds__(HB_BUCKETS) void k_bucket_rows(HashParams P) {
  for (uint32_t i = blockIdx.x * HB_BUCKETS + threadIdx.x; i < P.n_zero; i += gridDim.x * HB_BUCKETS) P.zero[i] = 0u;
  const SortSeg sg = P.segs[blockIdx.x];
  const TruthDev tr = P.truths[P.vcfs[sg.main_vcf].truth];
  const uint32_t d = threadIdx.x, shift = (uint32_t)sg.pad;
  const uint32_t kbase = sg.key_base + (d << shift);       // (two-level path: the segment is one partition of a VCF's key space)
  const uint32_t plo = kbase >> 4;
  const uint32_t phi = (kbase + ((1u << shift) - 1u)) >> 4;   // the last bucket of the key space ends at 2^32 - 1: no wrap
  uint32_t ba = plo >> tr.shift, bb = (phi >> tr.shift) + 1u;
  const uint32_t lim = (uint32_t)tr.nb + 1u;
  ba = ba < lim ? ba : lim; bb = bb < lim ? bb : lim;
  const int lo = tr.tidx[ba], hi = tr.tidx[bb];
  HashRow R;
  R.ent = P.ent + sg.bk_off + (size_t)d * HB_SUBS * (size_t)sg.bk_cap;
  R.tkeys = tr.keys + lo;
  R.src_off = sg.src_off;
  R.tn = hi - lo;
  R.cap = (uint32_t)sg.bk_cap;
  R.shift = shift;
  R.kbase = kbase;
  P.rows_out[(size_t)blockIdx.x * HB_BUCKETS + d] = R;
  if (P.xrows) {   // allele-extended batch: the second stream joins against the extended truth table
    uint32_t xa = plo >> tr.xshift, xb = (phi >> tr.xshift) + 1u;
    const uint32_t xlim = (uint32_t)tr.xnb + 1u;
    xa = xa < xlim ? xa : xlim; xb = xb < xlim ? xb : xlim;
    const int xlo = tr.xtidx[xa], xhi = tr.xtidx[xb];
    HashRowX X;
    X.ent = P.xent + 2 * (sg.bk_off + (size_t)d * HB_SUBS * (size_t)sg.bk_cap);
    X.xkeys = tr.xkeys + xlo; X.xref = tr.xref + xlo; X.xalt = tr.xalt + xlo;
    X.src_off = sg.src_off;
    X.tn = xhi - xlo;
    X.cap = (uint32_t)sg.bk_cap;
    X.shift = shift;
    X.kbase = kbase;
    P.xrows[(size_t)blockIdx.x * HB_BUCKETS + d] = X;
  }
}

constexpr int HB_THREADS = 512;
constexpr int HB_FQ_SLOTS = 1024;
// A place in a list (or a budget) for every lane that wants one, with ONE returning atomic per wave.  A wave of returning
// atomics on one LDS address takes ~450 cycles (tools/probe/lds_atomic_probe.hip: 0.14 lane-ops per clock, against 10 on
// random addresses), during which the CU's LDS serves nobody else.  Call with the whole wave active or from a divergent
// branch (the lanes outside simply do not want).
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* ctr, bool want) {
  const uint64_t m = ballot64(want);
  if (m == 0ull) return 0u;
  const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
  uint32_t base = 0u;
  if (want && rank == 0u) base = atomicAdd(ctr, (uint32_t)__popc(lo) + (uint32_t)__popc(hi));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
  return base + rank;
}
// Waves of a bucket's workgroup meet at three barriers only and walk their records independently in between: versions that
// split the record loop into workgroup-wide phases (filter pass, a queue of the filter-positive records settled with full
// waves, ...) removed the divergent steps but put every wave in the same phase at the same time, and lost 20-40 %.
#ifndef HB_WAVES_PER_EU
#define HB_WAVES_PER_EU 6   // three workgroups per CU: 80 VGPRs
#endif
__global__ __launch_bounds__(HB_THREADS) __attribute__((amdgpu_waves_per_eu(HB_WAVES_PER_EU, 8))) void k_classify_hash(HashParams P) {
  constexpr uint32_t LTR = 11, LNK = 9, LX = 11;       // exact tables: truth keys, keyless records, keys on marked bits
  constexpr uint32_t LTB = 14, LFB = 16;                 // bit maps: truth filter, kept keys
  constexpr int PER = 16;                                // records per thread at most: four trips of four
  static_assert((1 << LTR) == HB_TRUTH_SLOTS && (1 << LNK) == HB_NOKEY_SLOTS && HB_MAX_RECORDS <= HB_THREADS * PER, "table sizes");
  __shared__ uint32_t s_tk[1 << LTR];
  __shared__ uint32_t s_ts[1 << LTR];                // best bin + 1 of a '.'-ID match
  __shared__ uint32_t s_tf[(1 << LTR) / 32];         // matched by a kept record (ID ignored)
  __shared__ uint32_t s_tb[(1 << LTB) / 32];         // truth filter
  __shared__ uint32_t s_b1[(1 << LFB) / 32];         // kept keys outside the truth set: seen
  __shared__ uint32_t s_b2[(1 << LFB) / 32];         //                                  seen more than once (or collided)
  __shared__ __attribute__((aligned(8))) uint32_t s_x[1 << LX];   // first the waves' rings of parked records, then the keys on marked bits, exactly
  __shared__ uint32_t s_fq[HB_FQ_SLOTS];             // kept keys outside the truth set that were settled from a ring
  __shared__ uint32_t s_nk[1 << LNK];
  __shared__ uint32_t s_h[3 * 128];                  // TP / FP / U histograms, two u16 bins per dword (a bucket holds < 65 536 records)
  __shared__ uint32_t s_c[9];                        // kept, TP lines, distinct FP keys, matched truth keys, flags, truth keys staged, keyless inserts, marked inserts, listed keys
  const int tid = (int)threadIdx.x;
  const int d = (int)blockIdx.x;
  const int seg_id = (int)blockIdx.y + P.seg_base;
  const size_t row = (size_t)seg_id * HB_BUCKETS + (size_t)d;
  const HashRow R = P.rows[row];                                 // one scalar load beside the cursors: nothing below waits for more than
  const uint32_t* cur = P.cursor + row * HB_SUBS;                // one further round trip (the entries and the truth keys, together)
  const uint32_t segfl = d == 0 ? P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg_id] : 0u;   // the segment's flags travel in its first row
  // pointers that come out of memory are generic to the compiler: loads through them would be flat_load, which counts on
  // the LDS counter too -- every wait for an LDS result would then also wait for the entries in flight
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) v4u* gv4p;
  const gv4p g_ent = (gv4p)R.ent;
  const gu32p g_tkeys = (gu32p)R.tkeys;
  const uint32_t cap = R.cap;
  uint32_t nsub[HB_SUBS];     // wave-uniform (scalar loads)
  uint32_t nrec = 0, over = 0;
#pragma unroll
  for (int k = 0; k < HB_SUBS; ++k) {
    const uint32_t c = cur[k];
    nsub[k] = c < cap ? c : cap;
    nrec += nsub[k];
  }
  if (nrec == 0u) {   // two buckets in five are empty (a genome rarely ends on a power of two): a row of zeros, nothing else
    uint32_t* oh0 = P.row_hist + row * SPAN_HIST_WORDS;
    if (tid < 3 * 128) oh0[tid] = 0u;
    if (tid < 8) P.row_scal[row * 8 + tid] = tid == 5 ? segfl : 0u;
    return;
  }
  // A trip gives every thread four consecutive entries of one sub-region (32 bytes, two 16-byte loads: dword loads are bound
  // by the rate of memory instructions).  The first trip and the thread's truth key are in flight before the tables are
  // even cleared; later trips are fetched one ahead.
  const uint32_t lcap = 31u - (uint32_t)__clz(cap);            // cap is a power of two >= 4
  const uint32_t nslots = (HB_SUBS * cap) >> 2;
  v4u ea[2], eb[2];
  uint32_t nv[2];            // valid entries of the trip's four
  const v4u z4 = {0u, 0u, 0u, 0u};
  auto fetch = [&](int g, int buf) {
    const uint32_t slot = (uint32_t)g * HB_THREADS + (uint32_t)tid;
    const uint32_t e0 = slot << 2, sb = e0 >> lcap, w = e0 & (cap - 1u);
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < HB_SUBS; ++k) n = sb == (uint32_t)k ? nsub[k] : n;
    ea[buf] = z4; eb[buf] = z4; nv[buf] = 0u;
    if (slot < nslots && w < n && !over) {
      nv[buf] = n - w < 4u ? n - w : 4u;
      const gv4p src = g_ent + (e0 >> 1);   // two entries per 16 bytes
      ea[buf] = __builtin_nontemporal_load(src);
      eb[buf] = __builtin_nontemporal_load(src + 1);
    }
  };
  fetch(0, 0);
  const uint32_t shift = R.shift;                            // >= 4: a bucket is a whole range of positions
  const uint32_t kbase = R.kbase;                            // every key of the bucket is >= kbase
  const uint32_t klast = kbase + ((1u << shift) - 1u);
  uint32_t tkey0 = 0u, tkey1 = 0u;
  const int tn = over ? 0 : R.tn;
  if (tid < tn) tkey0 = g_tkeys[tid];
  if (tid + HB_THREADS < tn) tkey1 = g_tkeys[tid + HB_THREADS];   // up to 1 024 keys: all a bucket can take are in flight here
  for (int i = tid; i < (1 << LTR); i += HB_THREADS) { s_tk[i] = HB_EMPTY; s_ts[i] = 0u; }
  for (int i = tid; i < (1 << LFB) / 32; i += HB_THREADS) { s_b1[i] = 0u; s_b2[i] = 0u; }
  if (tid < (1 << LTB) / 32) s_tb[tid] = 0u;
  if (tid < (1 << LTR) / 32) s_tf[tid] = 0u;
  if (tid < (1 << LNK)) s_nk[tid] = HB_EMPTY;
  if (tid < 3 * 128) s_h[tid] = 0u;
  if (tid < 9) s_c[tid] = tid == 4 ? (segfl | (over ? SPANF_OVERFLOW : 0u)) : 0u;
  __syncthreads();
  // ---- the truth keys of the bucket's positions (the coarse position index hands out whole cells) ----
  for (int j0 = 0; j0 < tn; j0 += HB_THREADS) {   // wave-uniform trip count
    const int j = j0 + tid;
    const uint32_t k = j >= tn ? 0u : j0 == 0 ? tkey0 : j0 == HB_THREADS ? tkey1 : g_tkeys[j];
    const bool in = j < tn && k >= kbase && k <= klast;
    const uint32_t at = wave_reserve(&s_c[5], in);
    if (in) {
      if (at >= (1u << LTR) / 2u) {
        atomicOr(&s_c[4], SPANF_OVERFLOW);
      } else {
        bool fresh;
        const uint32_t v = k - kbase;
        (void)hb_insert(s_tk, LTR, v, &fresh);
        const uint32_t h = hb_hash(v) >> (32u - LTB);
        atomicOr(&s_tb[h >> 5], 1u << (h & 31));
      }
    }
  }
  __syncthreads();
  uint32_t n_pass = 0, n_tp = 0, fpr = 0;
  uint32_t vq[PER];          // the thread's records stay in registers for the second pass
  uint32_t cand = 0;         // bit k: record k is a kept key outside the truth set (with a comparable key)
  unsigned long long* mtp = reinterpret_cast<unsigned long long*>(P.mask_tp);
  // What remains of a record once its truth slot is known (t < 0: no truth key): the truth key's state, the ROC histograms,
  // the line counts, the TP bit.  Returns whether the record is a kept key outside the truth set.
  auto settle = [&](uint32_t elo, uint32_t ehi, int t) -> bool {
    const uint32_t inf = (elo >> 24) | ((ehi & 0x1fu) << 8);     // bits 0..11 as in the info word, bit 12 = TP line
    const uint32_t b1 = inf & I_BIN1;
    const bool kept = (inf & I_PASS) != 0u;                       // every entry is a live record
    const bool hit = t >= 0;
    const bool tpl = (hit && (inf & I_IDDOT)) || (inf & 0x1000u);
    if (hit) {
      if ((inf & I_IDDOT) && b1) atomicMax(&s_ts[t], b1);
      if (kept) atomicOr(&s_tf[t >> 5], 1u << (t & 31));
    }
    if (b1) atomicAdd(&s_h[(tpl ? 0 : 128) + ((b1 - 1u) >> 1)], 1u << (16u * ((b1 - 1u) & 1u)));
    if (kept) {
      ++n_pass;
      if (tpl) {
        ++n_tp;
        const int64_t o = R.src_off + (int64_t)(ehi >> 5);
        atomicOr(mtp + (o >> 6), 1ull << (o & 63));
      }
    }
    return kept && !hit;
  };
  // marks a kept key outside the truth set in the maps; a second mark (a repeat or a collision) sends both to the second pass
  auto mark = [&](uint32_t v) {
    const uint32_t h = hb_hash(v) >> (32u - LFB), bt = 1u << (h & 31u);
    if (atomicOr(&s_b1[h >> 5], bt) & bt) atomicOr(&s_b2[h >> 5], bt);
  };
  if (!(s_c[4] & SPANF_OVERFLOW)) {
    // The records whose key passes the truth filter (1 in 10; most of them true matches) are what costs here: their way
    // through the tables is a chain of dependent, divergent steps, and walked where they stand it runs in EVERY step of
    // the wave with a handful of lanes (PMC of that version: 41 % of the VALU lanes in use, SALU 71 % busy with the
    // bookkeeping of the branches).  So a wave only PARKS them -- ballot, rank, one 8-byte LDS store into a ring of its
    // own, no atomics, no barrier -- and settles 64 at a time with every lane busy.  A trip that would overrun the ring
    // (more than a quarter of its records parked twice in a row: a VCF that mostly matches) is walked the old way.
    constexpr uint32_t RING = 128;                                 // entries per wave
    uint2* ring = reinterpret_cast<uint2*>(s_x) + (tid >> 6) * RING;
    const uint32_t lane = (uint32_t)tid & 63u;
    uint32_t head = 0, tail = 0;                                   // wave-uniform
    auto drain = [&](uint32_t n) {                                 // the first n <= 64 parked records, one per lane
      if (lane < n) {
        const uint2 e = ring[(head + lane) & (RING - 1u)];
        const uint32_t v = e.x & 0xffffffu;
        if (settle(e.x, e.y, hb_find(s_tk, LTR, v))) {             // the filter's false positives: kept keys outside the truth set after all
          mark(v);
          const uint32_t at = wave_reserve(&s_c[8], true);         // not this thread's own record: it joins the second pass through a list
          if (at < HB_FQ_SLOTS) s_fq[at] = v; else atomicOr(&s_c[4], SPANF_OVERFLOW);
        }
      }
      head += n;
    };
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
      if (g + 1 < PER / 4) fetch(g + 1, (g + 1) & 1);
      uint32_t elo[4], ehi[4], hv[4], mb = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        elo[u] = u == 0 ? ea[g & 1][0] : u == 1 ? ea[g & 1][2] : u == 2 ? eb[g & 1][0] : eb[g & 1][2];
        ehi[u] = u == 0 ? ea[g & 1][1] : u == 1 ? ea[g & 1][3] : u == 2 ? eb[g & 1][1] : eb[g & 1][3];
        const uint32_t v = elo[u] & 0xffffffu;
        vq[4 * g + u] = v;
        hv[u] = hb_hash(v);
        const uint32_t fw = s_tb[hv[u] >> (37u - LTB)];
        const uint32_t valid = (uint32_t)u < nv[g & 1] ? 1u : 0u;
        mb |= (valid & ~(ehi[u] >> 3) & (fw >> ((hv[u] >> (32u - LTB)) & 31u)) & 1u) << u;   // entry bit 35 = I_NOKEY: no comparable key
      }
      uint64_t m[4];
      uint32_t tot = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) { m[u] = ballot64((mb >> u) & 1u); tot += (uint32_t)popc64(m[u]); }
      while (tail - head >= 64u || (tail != head && tail - head + tot > RING)) drain(tail - head < 64u ? tail - head : 64u);
      const bool park = tot <= RING;                               // wave-uniform
      if (park) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if ((mb >> u) & 1u) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u));
            ring[(tail + r) & (RING - 1u)] = make_uint2(elo[u], ehi[u]);
          }
          tail += (uint32_t)popc64(m[u]);
        }
      }
      // the records that match nothing (and, if the trip is not parked, the others too, the old way).  Making this part
      // branch-free (unconditional atomics with zero operands) was measured twice and lost both times: the extra LDS
      // atomics cost more than the branches they replace.
      uint32_t old[4], bit[4], wd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t valid = (uint32_t)u < nv[g & 1] ? 1u : 0u;
        const uint32_t maybe = (mb >> u) & 1u;
        bit[u] = 0u; wd[u] = 0u;
        bool c = false;
        if (valid & ~maybe) c = settle(elo[u], ehi[u], -1);
        else if (maybe && !park) c = settle(elo[u], ehi[u], hb_find(s_tk, LTR, elo[u] & 0xffffffu));
        if (c) {
          if ((ehi[u] >> 3) & 1u) {   // rare: keyless records are keys of their own, in an exact set, every insertion reserved
            if (wave_reserve(&s_c[6], true) >= (1u << LNK) / 2u) { atomicOr(&s_c[4], SPANF_OVERFLOW); }
            else {
              bool fresh;
              (void)hb_insert(s_nk, LNK, elo[u] & 0xffffffu, &fresh);
              fpr += fresh ? 1u : 0u;
            }
          } else {
            cand |= 1u << (4 * g + u);
            const uint32_t h = hv[u] >> (32u - LFB);
            bit[u] = 1u << (h & 31u); wd[u] = h >> 5;
          }
        }
      }
      // the four fetch-ORs of a trip in flight together (a zero bit changes nothing)
#pragma unroll
      for (int u = 0; u < 4; ++u) old[u] = atomicOr(&s_b1[wd[u]], bit[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (old[u] & bit[u]) atomicOr(&s_b2[wd[u]], bit[u]);   // seen before (or a collision): both meet again below
    }
    while (tail != head) drain(tail - head < 64u ? tail - head : 64u);
  }
  __syncthreads();   // the rings are empty: their room becomes the exact set of the keys on marked bits
  for (int i = tid; i < (1 << LX); i += HB_THREADS) s_x[i] = HB_EMPTY;
  __syncthreads();
  // second pass: keys on unmarked bits are distinct; keys on marked bits are counted exactly -- the thread's own candidates
  // from its registers, then the listed ones (the filter's false positives, settled by some lane of their wave)
  // which candidates sit on marked bits: one LDS read each, no branch; the others are distinct by construction
  uint32_t mm = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const uint32_t h = hb_hash(vq[k]) >> (32u - LFB);
    mm |= ((s_b2[h >> 5] >> (h & 31u)) & (cand >> k) & 1u) << k;
  }
  fpr += (uint32_t)__popc(cand & ~mm);
  {
    // room in the exact set for the marked ones: one reservation per wave (prefix sum of the lanes' counts by DPP-free shuffles)
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t cnt = (uint32_t)__popc(mm);
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(incl, o);
      if (lane >= (uint32_t)o) incl += y;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total) {   // wave-uniform
      uint32_t base = 0u;
      if (lane == 0u) base = atomicAdd(&s_c[7], total);
      uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + incl - cnt;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if ((mm >> k) & 1u) {
          if (at >= (1u << LX) / 2u) {
            atomicOr(&s_c[4], SPANF_OVERFLOW);
          } else {
            bool fresh;
            (void)hb_insert(s_x, LX, vq[k], &fresh);
            fpr += fresh ? 1u : 0u;
          }
          ++at;
        }
      }
    }
  }
  {
    const uint32_t nfq = s_c[8] < (uint32_t)HB_FQ_SLOTS ? s_c[8] : (uint32_t)HB_FQ_SLOTS;
    for (uint32_t i0 = 0; i0 < nfq; i0 += HB_THREADS) {   // wave-uniform trip count
      const uint32_t i = i0 + (uint32_t)tid;
      const uint32_t v = i < nfq ? s_fq[i] : 0u;
      const uint32_t h = hb_hash(v) >> (32u - LFB);
      const bool marked = i < nfq && ((s_b2[h >> 5] >> (h & 31u)) & 1u);
      fpr += (i < nfq && !marked) ? 1u : 0u;
      const uint32_t at = wave_reserve(&s_c[7], marked);
      if (marked) {
        if (at >= (1u << LX) / 2u) {
          atomicOr(&s_c[4], SPANF_OVERFLOW);
        } else {
          bool fresh;
          (void)hb_insert(s_x, LX, v, &fresh);
          fpr += fresh ? 1u : 0u;
        }
      }
    }
  }
  // ---- bucket epilogue: per-entry state -> U histogram and TP_R, counters, the row ----
  n_pass = wave_sum(n_pass); n_tp = wave_sum(n_tp); fpr = wave_sum(fpr);   // (not atomics of per-lane values on one address: hipcc makes a 64-step serial loop of each)
  if ((tid & 63) == 0) { atomicAdd(&s_c[0], n_pass); atomicAdd(&s_c[1], n_tp); atomicAdd(&s_c[2], fpr); }
  __syncthreads();
  uint32_t tpr = 0;
  for (int t = tid; t < (1 << LTR); t += HB_THREADS) {
    if (s_tk[t] == HB_EMPTY) continue;
    const uint32_t mx = s_ts[t];
    if (mx) atomicAdd(&s_h[256 + ((mx - 1u) >> 1)], 1u << (16u * ((mx - 1u) & 1u)));
    tpr += (s_tf[t >> 5] >> (t & 31)) & 1u;
  }
  tpr = wave_sum(tpr);
  if ((tid & 63) == 0 && tpr) atomicAdd(&s_c[3], tpr);
  __syncthreads();
  uint32_t* oh = P.row_hist + row * SPAN_HIST_WORDS;
  if (tid < 3 * 128) oh[tid] = s_h[tid];
  if (tid == 0) {
    uint32_t* sc = P.row_scal + row * 8;
    const uint32_t fl = s_c[4];
    sc[0] = s_c[0]; sc[1] = s_c[1]; sc[2] = s_c[0] - s_c[1]; sc[3] = s_c[3]; sc[4] = s_c[2]; sc[5] = fl; sc[6] = 0u; sc[7] = 0u;
  }
}

// staged truth keys per bucket and the coarse index over them (k_join_lean)
constexpr int DJ_CI_LOG2 = 10;       // coarse index over the staged truth keys: one entry per 2^10 keys (64 positions)

// ---------------------------------------------------------------------------
// k_join_lean -- k_join_direct's join of one bucket, rewritten (round 5).
//
// What bounded k_join_direct (profiles/r04_pmc_shuffled.json: 123 vector and about as many scalar instructions per record -- a CU
// issues one of either per clock --, the memory unit stalled 0.5 % of the time; and, once those are gone, tools/ab_shuffled.sh with
// -DLJ_*: the LIFE of a workgroup -- three dependent memory round trips, five barriers -- with only two workgroups per CU,
// because one bit per KEY of a bucket is 64 KB of LDS):
//   * Maps over POSITIONS, not keys: a bucket has at most 2^15 positions, and kept records rarely share one.  W holds two bits
//     per position -- T (low half of a word, sixteen positions per word): a truth key sits here; S1 (high half): a kept record
//     sits here -- so ONE returning ds_or per record both marks the record's position and brings back "a truth key sits
//     here" and "somebody was here before me"; the latter sets the position's bit in S2.  8 + 4 KB instead of 64: four
//     workgroups per CU hide each other's round trips.
//   * Distinct kept keys, exactly: a position claimed once is one key (popcount(S1) - popcount(S2)); the records of positions
//     claimed more than once (a second look, behind a barrier, at S2: repeats, multi-allelic sites, a random record on a
//     truth position) go through a small exact set.  FP_R = that count - TP_R (the kept keys that are truth keys).
//   * ONE pass over the records without a branch or an execution mask per record except the histogram's: a wave owns one of
//     the bucket's eight sub-regions (wave w = sub-region w: a record's address is its lane's, no select chains); slots
//     beyond the cursor are ZERO entries, inert everywhere (not kept, no bin, no ID '.').  One histogram add per record
//     (all records by bin): the TP histogram is added to by the hits only, the FP histogram is the difference.
//   * The records that need a closer look -- on a truth position (8 %), on a position claimed twice (a few per cent) -- are
//     compacted ONCE per wave (one prefix sum, sixteen masked 8-byte LDS stores) and settled 64 at a time with every lane
//     busy: exact key through the coarse index, best bin / matched-by-kept state, TP histogram, the input-order TP bit (one
//     32-bit atomic OR), the exact set.
//   * Records without a comparable key and host-decided TP lines are rare: a wave that holds one (one OR over its entries'
//     flag words tells) runs the same pass with two extra masks.
// (Round 3's k_join_direct, one bit per KEY of a bucket, is kept as text under tools/probe/.)
// ---------------------------------------------------------------------------
constexpr int LJ_THREADS = 512;
constexpr int LJ_SET_LOG2 = 11;      // exact set of the keys on positions claimed more than once, and of the kept records without a comparable key
constexpr uint32_t LJ_RING = 128;    // compacted records per wave and round
constexpr uint32_t LJ_NOKEY_TAG = 1u << 24;   // (a key inside a bucket is below 2^24)
#ifndef LJ_WAVES
#define LJ_WAVES 6   // 80 VGPRs, three workgroups per CU: with 64 (four per CU) the kernel spills and is 4 % slower (same-box A/B)
#endif
// HIST = false: the scatter counted every record by bin (HashParams.scatter_hist): no histogram here but the true positives'
// BIG = true (round 6): buckets of 2^17 positions and up to 32 768 records -- a shuffled 10 M-record VCF on 50 Mb is TWO partitions
// of 256 such buckets, filled by ONE pass of the 512-digit scatter whose pieces stay whole 64-byte sectors (with 2^15 positions per
// bucket the VCF took a level-1 scatter of its own, or 21-byte pieces: LABNOTES round 6).  The same join with everything four times
// as large: 1 024 threads, 32 records per thread (two waves per sub-region), 4 096 staged truth keys, 48 KB of position maps --
// 120 KB of LDS, one workgroup per CU at four waves per SIMD.
template <int LB, bool HIST, bool BIG = false>
__global__ __launch_bounds__(BIG ? 1024 : LJ_THREADS) __attribute__((amdgpu_waves_per_eu(BIG ? 4 : LJ_WAVES, BIG ? 4 : 8))) void k_join_lean(HashParams P) {
  constexpr int THREADS = BIG ? 1024 : LJ_THREADS;
  constexpr int PER = BIG ? 32 : 16;                     // records per thread at most: trips of four
  constexpr int TMAX = BIG ? 4 * DJ_TRUTH_MAX : DJ_TRUTH_MAX;   // staged truth keys
  constexpr int TPT = TMAX / THREADS;                    // ... per thread
  constexpr int SETL = BIG ? LJ_SET_LOG2 + 1 : LJ_SET_LOG2;
  constexpr uint32_t SUBCAP = BIG ? 4u * HB_SUB_MAX : (uint32_t)HB_SUB_MAX;   // entries per sub-region at most
  constexpr uint32_t WCAP = 64u * PER;                   // entries a wave holds: a whole sub-region, or (BIG) one half of one
  constexpr int W_WORDS = LB >= 10 ? (1 << (LB - 8)) : 4;    // sixteen positions per word: T in bits 0..15, S1 in 16..31
  constexpr int W2_WORDS = LB >= 11 ? (1 << (LB - 9)) : 4;   // S2: a bit per position
  constexpr int CI_N = LB > DJ_CI_LOG2 ? 1 << (LB - DJ_CI_LOG2) : 1;
  static_assert(SUBCAP == WCAP * (BIG ? 2 : 1) && HB_SUBS * 64 * (BIG ? 2 : 1) == THREADS && W_WORDS % 4 == 0 && W2_WORDS % 4 == 0 && TPT * THREADS == TMAX, "a wave (two) per sub-region");
  __shared__ __attribute__((aligned(16))) uint32_t s_W[W_WORDS];
  __shared__ __attribute__((aligned(16))) uint32_t s_W2[W2_WORDS];
  __shared__ uint32_t s_tk[TMAX + 2];        // the staged truth keys, sorted (absolute keys), 0xffffffff behind the last
  __shared__ uint16_t s_ci[CI_N];                    // per block of 2^DJ_CI_LOG2 keys: index of its first staged truth key (written for blocks that hold one)
  __shared__ uint32_t s_ts[TMAX];            // per staged key: best bin + 1 of a '.'-ID match
  __shared__ uint32_t s_tf[TMAX / 32];       // matched by a kept record (ID ignored)
  __shared__ uint32_t s_ha[258];                     // every record by bin + 1 (slot 0: no bin; the top bin is counted in registers)
  __shared__ uint32_t s_htp[130];                    // TP records by bin + 1, two u16 slots per dword
  __shared__ uint32_t s_hu[128];                     // distinct-truth-key histogram, two u16 bins per dword
  __shared__ uint32_t s_set[1 << SETL];
  __shared__ __attribute__((aligned(8))) uint2 s_ring[(THREADS / 64) * LJ_RING];
  __shared__ uint32_t s_c[8];                        // kept, TP lines, fresh keys of the exact set, matched truth keys, flags, top-bin records, set inserts, S1 - S2 bits
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = (int)blockIdx.x;
  const int seg_id = (int)blockIdx.y + P.seg_base;
  const size_t row = (size_t)seg_id * HB_BUCKETS + (size_t)d;
  // The buckets above the segment's highest position hold nothing (40 % of the grid on a 5 Mb genome: the position BITS the
  // optimistic pass hands over bound the buckets in use only by a power of two).  The scatter noted the highest bucket it filled:
  // the workgroups above leave behind ONE scalar load (a workgroup that waits for its descriptor and cursors to find its bucket
  // empty holds a slot of its CU for a whole memory round trip), k_finalize sums no row of theirs (FinalizeParams.row_cap).
  // Bucket 0 always stays: it carries the segment's flags.
  if (P.seg_maxd && d > 0 && (uint32_t)d >= P.seg_maxd[seg_id]) return;
  const HashRow R = P.rows[row];
  const uint32_t* cur = P.cursor + row * HB_SUBS;
  const uint32_t segfl = d == 0 ? P.cursor[(size_t)P.n_seg * HB_BUCKETS * HB_SUBS + seg_id] : 0u;   // the segment's flags travel in its first row
  // maps, truth state and histograms are cleared while the descriptor and the cursors are on their way
  for (int i = tid; i < W_WORDS / 4; i += THREADS) *reinterpret_cast<uint4*>(&s_W[4 * i]) = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < W2_WORDS / 4; i += THREADS) *reinterpret_cast<uint4*>(&s_W2[4 * i]) = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < (1 << SETL) / 4; i += THREADS) *reinterpret_cast<uint4*>(&s_set[4 * i]) = make_uint4(HB_EMPTY, HB_EMPTY, HB_EMPTY, HB_EMPTY);
#pragma unroll
  for (int h = 0; h < TPT; ++h) s_ts[tid + h * THREADS] = 0u;
  if (tid < TMAX / 32) s_tf[tid] = 0u;
  if (tid < 258) s_ha[tid] = 0u;
  if (tid < 130) s_htp[tid] = 0u;
  if (tid < 128) s_hu[tid] = 0u;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) v4u* gv4p;
  const gu32p g_tkeys = (gu32p)R.tkeys;
  const uint32_t cap = R.cap;
  uint32_t nrec = 0, over = 0, nw = 0;     // wave-uniform (scalar loads); nw: entries of this wave's sub-region
#pragma unroll
  for (int k = 0; k < HB_SUBS; ++k) {
    const uint32_t c = cur[k];
    const uint32_t n = c < cap ? c : cap;
    nrec += n;
    if (BIG) { const uint32_t off = (uint32_t)(wave & 1) * WCAP; if ((wave >> 1) == k) nw = n > off ? (n - off < WCAP ? n - off : WCAP) : 0u; }
    else nw = wave == k ? n : nw;
  }
  const size_t orow = (size_t)seg_id * (size_t)P.out_stride + (size_t)d;   // where the bucket's row goes (allele-extended batches: the second stream's rows follow)
  uint32_t* oh = P.row_hist + orow * SPAN_HIST_WORDS;
  if (nrec == 0u) {   // an empty bucket: a row of zeros, nothing else
    if (tid < 3 * 128) oh[tid] = 0u;
    if (tid < 8) P.row_scal[orow * 8 + tid] = tid == 5 ? segfl : 0u;
    return;
  }
  over |= (R.shift > (uint32_t)LB || cap > SUBCAP) ? 1u : 0u;   // (the host never launches this instantiation for such a segment)
  const uint32_t shift = R.shift > (uint32_t)LB ? (uint32_t)LB : R.shift;   // 4 <= shift <= LB: a bucket is a whole range of positions
  const uint32_t kbase = R.kbase;                            // every key of the bucket is >= kbase
  const uint32_t klast = kbase + ((1u << shift) - 1u);
  const int tn_all = R.tn;
  over |= tn_all > TMAX ? 1u : 0u;
  const int tn = over ? 0 : tn_all;
  if (over) nw = 0u;
  // The order of the loads (round 6, from clock stations in the wide join, where a workgroup is alone on its CU: with every trip
  // asked for up front, as rounds 3-5 had it, the workgroup reached its first barrier after 11 of its 31 us -- a CU's memory
  // pipeline takes a few KB of requests at a time, so ISSUING 200 KB of loads is waiting for them): the truth keys first; the
  // entries only behind the truth bits, so that the pass takes the trips one at a time while the later ones are still on their
  // way and no barrier stands behind the loads.  Wide buckets: 0.70 -> 0.65 ms per 1.6e8 records, narrow ones 0.81 -> 0.79 per
  // 2.56e8 (same-box A/B; lightly filled wide buckets lose 2 %: their dummy loads).
  // EVERY load is issued by every lane, whatever the bucket holds -- a lane with nothing to fetch reads a word that is there
  // anyway (the slice's first key, its sub-region's first quad: one request for all such lanes) and throws it away: behind a
  // branch the compiler cannot count the loads in flight and waits for all of them (s_waitcnt vmcnt(0)) at the first use.
  uint32_t tkey[TPT];
  const gu32p tk_safe = tn > 0 ? g_tkeys : (gu32p)P.rows;   // (an empty slice may lie at the end of the truth table)
#pragma unroll
  for (int h = 0; h < TPT; ++h) {
    const int j = tid + h * THREADS;
    tkey[h] = tk_safe[j < tn ? j : 0];
  }
  const int nw4 = (int)(((1u << shift) + 1023u) >> 10);      // 16-byte pieces of W in use (64 positions each); S2 has half as many
  if (tid < 8) s_c[tid] = tid == 4 ? (segfl | (over ? SPANF_OVERFLOW : 0u)) : 0u;
  __syncthreads();
  // ---- the truth keys of the bucket's positions: the sorted slice as it is (its keys outside the bucket match nothing), a bit per position inside ----
#pragma unroll
  for (int h = 0; h < TPT; ++h) {
    const int j = tid + h * THREADS;
    const uint32_t k = j < tn ? tkey[h] : 0xffffffffu;
    const bool in = j < tn && k >= kbase && k <= klast;
    if (j < tn) s_tk[j] = k;
    if (in) atomicOr(&s_W[(k - kbase) >> 8], 1u << (((k - kbase) >> 4) & 15u));
  }
  if (tid < 2) s_tk[tn + tid] = 0xffffffffu;                 // behind the last staged key: a look-up may read past its key
  __syncthreads();
  // ---- the entries: lane l of wave w takes the quads l, l + 64, ... of sub-region w, four consecutive entries = 32 bytes ----
  const gv4p sbase = (gv4p)R.ent + (size_t)(BIG ? wave >> 1 : wave) * (cap >> 1);   // the wave's sub-region; two entries per 16 bytes (cap is even: a power of two)
  const uint32_t qoff = BIG ? (uint32_t)(wave & 1) * (WCAP / 4u) : 0u;             // (BIG: the quads of the sub-region's first half belong to the wave in front)
  v4u ea[PER / 4], eb[PER / 4];
  const int ntrips = (int)((nw + 255u) >> 8);                       // wave-uniform
#pragma unroll
  for (int g = 0; g < PER / 4; ++g) {
    const uint32_t q = (uint32_t)g * 64u + (uint32_t)lane;
    // behind the cursor: the SUB-REGION's first quad again (the pass zeroes it, trips >= ntrips are never looked at) -- not the
    // half's: a sub-region of fewer than 4 096 entries ends in front of its second half, the last one of the last bucket at the
    // end of the allocation (tools/gpu_fuzz_big.py found that one: a fault)
    const uint32_t qq = 4u * q < nw ? qoff + q : 0u;
    ea[g] = __builtin_nontemporal_load(sbase + 2u * qq);
    eb[g] = __builtin_nontemporal_load(sbase + 2u * qq + 1u);
  }
  // the coarse index (read by the settle step, two barriers on): the first key of its block of 2^DJ_CI_LOG2 keys names itself
  // (the slice is sorted; the key in front comes from LDS now that the slice is staged)
#pragma unroll
  for (int h = 0; h < TPT; ++h) {
    const int j = tid + h * THREADS;
    const uint32_t k = tkey[h];
    const bool in = j < tn && k >= kbase && k <= klast;
    const uint32_t kp = j > 0 && j < tn ? s_tk[j - 1] : 0u;   // (0 is below every bucket's first key that matters)
    const bool pin = j > 0 && kp >= kbase;                    // (kp <= k <= klast)
    if (in && (!pin || ((kp - kbase) >> DJ_CI_LOG2) != ((k - kbase) >> DJ_CI_LOG2))) s_ci[(k - kbase) >> DJ_CI_LOG2] = (uint16_t)j;
  }
  const uint32_t nb = (uint32_t)P.n_bins;
  int ttop = 0;                                              // largest power of two <= tn
  if (tn > 0) ttop = 1 << (31 - __clz(tn));
  uint32_t n_tp = 0, fresh = 0;
  uint32_t hitm = 0, keptm = 0, top = 0, nkm = 0;            // bit k: record k sits on a truth position (or is a host-decided TP line) / is kept / (rare) is kept without a key
  // Records without a comparable key and host-decided TP lines are rare.  A wave that holds one in a trip (one OR over the trip's
  // flag words tells) rewrites those entries in its registers so that the ORs below need no mask for them: a keyless record
  // loses its PASS bit (it marks no position and is counted as kept through nkm), which moves to bit 31 of its flag word for
  // the settle step; host-decided TP lines are added to the records to be settled afterwards.
  uint32_t tplm = 0;
  const bool run = !(s_c[4] & SPANF_OVERFLOW);
  auto pass = [&]() {
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
      if (g < ntrips) {                                        // wave-uniform
        // the partly filled quad at the end of the sub-region: what lies behind the cursor there is left over from an earlier run
        if (nw < (uint32_t)(g + 1) * 256u) {                   // wave-uniform (nw > g * 256: g < ntrips)
          const int left = (int)nw - 4 * (g * 64 + lane);      // entries of the lane's quad in front of the cursor
          if (left < 4) { eb[g][2] = 0u; eb[g][3] = 0u; }
          if (left < 3) { eb[g][0] = 0u; eb[g][1] = 0u; }
          if (left < 2) { ea[g][2] = 0u; ea[g][3] = 0u; }
          if (left < 1) { ea[g][0] = 0u; ea[g][1] = 0u; }
        }
        if (ballot64(((ea[g][1] | ea[g][3] | eb[g][1] | eb[g][3]) & 0x18u) != 0u)) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t ehi = u == 0 ? ea[g][1] : u == 1 ? ea[g][3] : u == 2 ? eb[g][1] : eb[g][3];
            const uint32_t nokey = (ehi >> 3) & 1u, kb = (ehi >> 1) & 1u;
            tplm |= ((ehi >> 4) & 1u) << (4 * g + u);
            nkm |= (kb & nokey) << (4 * g + u);
            const uint32_t nhi = nokey ? ((ehi & ~2u) | (kb << 31)) : ehi;
            if (u == 0) ea[g][1] = nhi; else if (u == 1) ea[g][3] = nhi; else if (u == 2) eb[g][1] = nhi; else eb[g][3] = nhi;
          }
        }
        uint32_t old[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                          // the four returning ORs of the trip first, in flight together
          const uint32_t elo = u == 0 ? ea[g][0] : u == 1 ? ea[g][2] : u == 2 ? eb[g][0] : eb[g][2];
          const uint32_t ehi = u == 0 ? ea[g][1] : u == 1 ? ea[g][3] : u == 2 ? eb[g][1] : eb[g][3];
          const uint32_t kb = (ehi >> 1) & 1u;                 // PASS: every entry is a live record, so this is `kept`
          keptm |= kb << (4 * g + u);
          old[u] = atomicOr(&s_W[(elo >> 8) & (uint32_t)(W_WORDS - 1)], kb << (16u + ((elo >> 4) & 15u)));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t elo = u == 0 ? ea[g][0] : u == 1 ? ea[g][2] : u == 2 ? eb[g][0] : eb[g][2];
          const uint32_t ehi = u == 0 ? ea[g][1] : u == 1 ? ea[g][3] : u == 2 ? eb[g][1] : eb[g][3];
          const uint32_t o = old[u] >> ((elo >> 4) & 15u);
          hitm |= (o & 1u) << (4 * g + u);
          if ((o >> 16) & (ehi >> 1) & 1u) atomicOr(&s_W2[(elo >> 9) & (uint32_t)(W2_WORDS - 1)], 1u << ((elo >> 4) & 31u));   // somebody kept was here before (rare: the few lanes it is true for)
          if (HIST) {
            const uint32_t b1 = (elo >> 24) | ((ehi & 1u) << 8);
            // the saturated top bin, where real QUALs pile up, is counted in a register (the lanes of a wave would serialise on its one address)
            const bool istop = b1 == nb;
            top += istop ? 1u : 0u;
            if (!istop) atomicAdd(&s_ha[b1], 1u);
          }
        }
      }
    }
  };
  if (run) pass();
  __syncthreads();   // every kept record has marked its position: S2 says which positions were claimed more than once
  if (run) {
    // ---- the second look: kept records (with a key) on positions claimed more than once ----
    uint32_t collm = 0;
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
      if (g < ntrips) {                                        // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t elo = u == 0 ? ea[g][0] : u == 1 ? ea[g][2] : u == 2 ? eb[g][0] : eb[g][2];
          collm |= ((s_W2[(elo >> 9) & (uint32_t)(W2_WORDS - 1)] >> ((elo >> 4) & 31u)) & 1u) << (4 * g + u);
        }
      }
    }
    // the lane's slots in front of the cursor (a zero entry may sit on a truth position: it is not to be settled)
    uint32_t vmask = 0;
#pragma unroll
    for (int g = 0; g < PER / 4; ++g) {
      if (nw >= (uint32_t)(g + 1) * 256u) vmask |= 15u << (4 * g);             // wave-uniform: the whole trip lies in front of the cursor
      else if (nw > (uint32_t)g * 256u) {
        const int left = (int)nw - 4 * (g * 64 + lane);
        vmask |= (left >= 4 ? 15u : left <= 0 ? 0u : (1u << left) - 1u) << (4 * g);
      }
    }
    const uint32_t postm = (((hitm | tplm) & vmask) | (collm & keptm) | nkm);   // on truth positions (a keyless record there settles as no hit), host-decided TP lines, keys to be counted exactly, keyless kept records
    // ---- one compaction per wave: a prefix sum over the lanes' counts, sixteen masked 8-byte stores; then 64 at a time ----
    uint2* const ring = s_ring + (size_t)wave * LJ_RING;
    uint32_t* const mtp = reinterpret_cast<uint32_t*>(P.mask_tp);
    const uint32_t cnt = (uint32_t)__popc(postm);
    uint32_t tot;
    const uint32_t excl = k3_scan(cnt, tot) - cnt;
    for (uint32_t w0 = 0; w0 < tot; w0 += LJ_RING) {           // (one round unless a wave holds more than LJ_RING such records)
      uint32_t slot = excl - w0;                                // may wrap below zero: unsigned compare keeps it out
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int g = k >> 2, u = k & 3;
        const uint32_t elo = u == 0 ? ea[g][0] : u == 1 ? ea[g][2] : u == 2 ? eb[g][0] : eb[g][2];
        const uint32_t ehi = u == 0 ? ea[g][1] : u == 1 ? ea[g][3] : u == 2 ? eb[g][1] : eb[g][3];
        const bool mine = (postm >> k) & 1u;
        if (mine && slot < LJ_RING) ring[slot] = make_uint2(elo, ehi);
        slot += mine ? 1u : 0u;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own stores, read back by other lanes of the same wave
      const uint32_t n = tot - w0 < LJ_RING ? tot - w0 : LJ_RING;
      for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const bool act = i0 + (uint32_t)lane < n;
        uint2 e = make_uint2(0u, 0u);
        if (act) e = ring[i0 + (uint32_t)lane];
        const uint32_t elo = e.x, ehi = e.y;
        const uint32_t v = elo & 0xffffffu, key = kbase + v;
        const uint32_t b1 = (elo >> 24) | ((ehi & 1u) << 8);
        const bool kept = (ehi & 0x80000002u) != 0u, iddot = (ehi & 4u) != 0u, nokey = (ehi & 8u) != 0u;   // (bit 31: a kept keyless record, see above)
        bool hit = act && !nokey && ((s_W[(v >> 8) & (uint32_t)(W_WORDS - 1)] >> ((v >> 4) & 15u)) & 1u);
        if (hit) {
          // a truth key sits on the record's position: is it the record's key?  At or behind the first key of its block
          int j = (int)s_ci[v >> DJ_CI_LOG2];
          const uint32_t k0 = s_tk[j], k1 = s_tk[j + 1];
          if (k0 != key) {
            if (k1 == key) j = j + 1;
            else if (k1 < key) { j = lds_lower_bound(s_tk, tn, ttop, key); hit = j < tn && s_tk[j] == key; }
            else hit = false;
          }
          if (hit) {
            if (iddot && b1) atomicMax(&s_ts[j], b1);
            if (kept) atomicOr(&s_tf[j >> 5], 1u << (j & 31));
          }
        }
        const bool tpl = act && ((hit && iddot) || (ehi & 0x10u));
        if (tpl) {
          atomicAdd(&s_htp[b1 >> 1], 1u << (16u * (b1 & 1u)));
          if (kept) {
            n_tp += 1u;
            const int64_t o = R.src_off + (int64_t)((ehi >> 5) & 0x3ffffffu);
            atomicOr(mtp + (o >> 5), 1u << (o & 31));
          }
        }
        // the exact set: a kept key on a position claimed more than once, or a kept record without a key (a key of its own)
        const bool coll = act && kept && (nokey || ((s_W2[(v >> 9) & (uint32_t)(W2_WORDS - 1)] >> ((v >> 4) & 31u)) & 1u));
        if (ballot64(coll)) {
          const uint32_t at = wave_reserve(&s_c[6], coll);
          if (coll) {
            if (at >= (1u << SETL) / 2u) atomicOr(&s_c[4], SPANF_OVERFLOW);
            else {
              bool fr;
              (void)hb_insert(s_set, SETL, nokey ? (v | LJ_NOKEY_TAG) : v, &fr);
              fresh += fr ? 1u : 0u;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the ring is read before the next round overwrites it
    }
  }
  // (hipcc turns an LDS atomic add of a per-lane value on ONE address into a serial loop over the 64 lanes: DPP sums instead)
  // one sum for two counts (a wave holds <= 1 024 records: 16 bits each), one for the rarer third
  const uint32_t packed = wave_sum((uint32_t)__popc(keptm | nkm) | (n_tp << 16));
  if (ballot64(fresh != 0u)) fresh = wave_sum(fresh);
  if (HIST) top = wave_sum(top);
  if (lane == 0) {
    atomicAdd(&s_c[0], packed & 0xffffu); atomicAdd(&s_c[1], packed >> 16);
    if (fresh) atomicAdd(&s_c[2], fresh);
    if (HIST && top) atomicAdd(&s_c[5], top);
  }
  // ---- bucket epilogue: positions claimed exactly once, per-entry state -> U histogram and TP_R, counters, the row ----
  // (S1 and S2 are final since the barrier above: only the exact set and the truth state were written behind it)
  {
    int32_t bits = 0;
    for (int i = tid; i < nw4; i += THREADS) {
      const uint4 w = *reinterpret_cast<const uint4*>(&s_W[4 * i]);
      bits += __popc(w.x >> 16) + __popc(w.y >> 16) + __popc(w.z >> 16) + __popc(w.w >> 16);
    }
    for (int i = tid; i < (nw4 + 1) / 2; i += THREADS) {
      const uint4 w = *reinterpret_cast<const uint4*>(&s_W2[4 * i]);
      bits -= __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
    }
    const uint32_t b = wave_sum((uint32_t)bits);
    if (lane == 0 && b) atomicAdd(&s_c[7], b);
  }
  __syncthreads();
  uint32_t tpr = 0;
#pragma unroll
  for (int h = 0; h < TPT; ++h) {
    const int j = tid + h * THREADS;
    if (j < tn) {
      const uint32_t mx = s_ts[j];
      if (mx) atomicAdd(&s_hu[(mx - 1u) >> 1], 1u << (16u * ((mx - 1u) & 1u)));
      tpr += (s_tf[j >> 5] >> (j & 31)) & 1u;
    }
  }
  tpr = wave_sum(tpr);
  if (lane == 0 && tpr) atomicAdd(&s_c[3], tpr);
  __syncthreads();
  if (tid < 128) {
    const uint32_t tall = s_c[5];
    const int b0 = 2 * tid, b1 = 2 * tid + 1;
    auto get = [&](const uint32_t* t, int slot) { return (t[slot >> 1] >> (16 * (slot & 1))) & 0xffffu; };
    const uint32_t tp0 = get(s_htp, 1 + b0), tp1 = get(s_htp, 1 + b1);
    const uint32_t all0 = s_ha[1 + b0] + (b0 == (int)nb - 1 ? tall : 0u), all1 = s_ha[1 + b1] + (b1 == (int)nb - 1 ? tall : 0u);
    oh[tid] = tp0 | (tp1 << 16);
    oh[128 + tid] = HIST ? (all0 - tp0) | ((all1 - tp1) << 16) : 0u;   // (HIST = false: k_finalize takes the difference from the scatter's counts)
    oh[256 + tid] = s_hu[tid];
  }
  if (tid == 0) {
    uint32_t* sc = P.row_scal + orow * 8;
    const uint32_t fl = s_c[4];
    sc[0] = s_c[0]; sc[1] = s_c[1]; sc[2] = s_c[0] - s_c[1]; sc[3] = s_c[3];
    sc[4] = s_c[7] + s_c[2] - s_c[3];   // distinct kept keys outside the truth set: positions claimed once + the exact set's keys, minus those that are truth keys
    sc[5] = fl; sc[6] = 0u; sc[7] = 0u;
  }
}

// ---------------------------------------------------------------------------
// k_join_ext -- the second stream of an allele-extended batch: the records of a bucket whose REF / ALT are not two single bases
// (16-byte entries: the ordinary entry with the position's first key, then the two allele codes), joined EXACTLY on
// (position, REF, ALT) against the extended truth table.  Their 32-bit key cannot stand for them (its nibble is a hash), but a
// bucket has at most 2^15 positions and such records rarely share one, so three bit maps over POSITIONS carry the common case:
//   * P: an extended truth entry sits at this position.  Records there are parked in a ring of their wave and compared with
//     that position's truth entries (bisection of the sorted slice, then REF and ALT) 64 at a time; exact hits are marked in a
//     bit per entry, their truth entry's state and the input-order TP bit follow.
//   * S / S2: a kept record outside the truth set sits here / more than one does.  A record on a position only it claims is a
//     distinct key by construction; the records on positions claimed more than once (true repeats, multi-allelic sites) and
//     the keyless ones go to a list and are compared all against all -- exact, and free of the races a hash set of 12-byte
//     keys would have.
// Histograms, counts, scalars: a row like k_join_direct's (every count of the two streams is additive: a key belongs to one).
// 512 threads, 43 KB of LDS (three workgroups per CU); what does not fit (1 024 truth entries, 1 024 listed records) flags the VCF for the radix sort.
// ---------------------------------------------------------------------------
constexpr int XJ_THREADS = 512;
constexpr int XJ_TRIPS = HB_MAX_RECORDS / XJ_THREADS;   // one entry per thread and trip
__global__ __launch_bounds__(XJ_THREADS) void k_join_ext(HashParams P) {
  constexpr int MAPW = (1 << (DJ_MAX_SHIFT - 4)) / 32;       // 2^15 positions
  __shared__ uint32_t s_p[MAPW], s_s[MAPW], s_s2[MAPW];
  __shared__ uint32_t s_xk[XJ_TRUTH_MAX], s_xr[XJ_TRUTH_MAX], s_xa[XJ_TRUTH_MAX], s_ts[XJ_TRUTH_MAX];
  __shared__ uint32_t s_tf[XJ_TRUTH_MAX / 32];
  __shared__ uint32_t s_hit[HB_MAX_RECORDS / 32];            // entry e of the bucket hit a truth entry exactly
  // the waves' rings (pass 1) and the list (pass 3) take turns in one array: with both, the workgroup's 52.7 KB came to two per
  // CU as the hardware allots LDS, and the kernel's time follows the number of workgroups in flight (0.79 -> 0.58 ms)
  static_assert(3 * XJ_LIST_MAX >= 5 * XJ_THREADS, "the rings fit the list's room");
  __shared__ __attribute__((aligned(16))) uint32_t s_un[3 * XJ_LIST_MAX];
  uint32_t* const s_lp = s_un; uint32_t* const s_lr = s_un + XJ_LIST_MAX; uint32_t* const s_la = s_un + 2 * XJ_LIST_MAX;
  uint4* const s_ring = reinterpret_cast<uint4*>(s_un);        // [(XJ_THREADS / 64) * 64]
  uint32_t* const s_ringe = s_un + 4 * XJ_THREADS;             // [(XJ_THREADS / 64) * 64]
  __shared__ uint32_t s_htp[130], s_hfp[130], s_hu[128];
  __shared__ uint32_t s_c[10];   // kept, TP lines, distinct kept keys outside the truth set, matched truth entries, flags, -, listed records, top-bin TP, top-bin FP, -
  const int tid = (int)threadIdx.x;
  const int d = (int)blockIdx.x;
  const int seg_id = (int)blockIdx.y + P.seg_base;
  const size_t row = (size_t)seg_id * HB_BUCKETS + (size_t)d;
  if (P.seg_maxd && d > 0 && (uint32_t)d >= P.seg_maxd[seg_id]) return;   // no entry of either stream above the segment's highest bucket (k_join_lean)
  const HashRowX R = P.xrows[row];
  const uint32_t* cur = P.xcursor + row * HB_SUBS;
  const size_t orow = (size_t)seg_id * (size_t)P.out_stride + (size_t)HB_BUCKETS + (size_t)d;   // the second stream's rows follow the first's
  for (int i = tid; i < MAPW; i += XJ_THREADS) { s_p[i] = 0u; s_s[i] = 0u; s_s2[i] = 0u; }
  s_ts[tid] = 0u; s_ts[tid + XJ_THREADS] = 0u;
  if (tid < XJ_TRUTH_MAX / 32) s_tf[tid] = 0u;
  if (tid < HB_MAX_RECORDS / 32) s_hit[tid] = 0u;
  if (tid < 130) { s_htp[tid] = 0u; s_hfp[tid] = 0u; }
  if (tid < 128) s_hu[tid] = 0u;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) v4u* gv4p;
  const gv4p g_ent = (gv4p)R.ent;
  const uint32_t cap = R.cap;
  uint32_t nsub[HB_SUBS], pre[HB_SUBS + 1];   // wave-uniform
  uint32_t over = 0;
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < HB_SUBS; ++k) {
    const uint32_t c = cur[k];
    over |= c > cap ? 1u : 0u;
    nsub[k] = c < cap ? c : cap;
    pre[k + 1] = pre[k] + nsub[k];
  }
  const uint32_t nrec = pre[HB_SUBS];
  uint32_t* oh = P.row_hist + orow * SPAN_HIST_WORDS;
  if (nrec == 0u) {
    if (tid < 3 * 128) oh[tid] = 0u;
    if (tid < 8) P.row_scal[orow * 8 + tid] = 0u;
    return;
  }
  over |= R.shift > (uint32_t)DJ_MAX_SHIFT ? 1u : 0u;
  const uint32_t kbase = R.kbase;
  const int tn_all = R.tn;
  over |= tn_all > XJ_TRUTH_MAX ? 1u : 0u;
  const int tn = over ? 0 : tn_all;
  if (tid < 10) s_c[tid] = tid == 4 ? (over ? SPANF_OVERFLOW : 0u) : 0u;
  const int ntrips = over ? 0 : (int)((nrec + XJ_THREADS - 1) / XJ_THREADS);
  // where entry e of the bucket lies: the sub-regions one after the other
  auto entry_at = [&](uint32_t e) -> gv4p {
    uint32_t k = 0;
#pragma unroll
    for (int q = 1; q < HB_SUBS; ++q) k += e >= pre[q] ? 1u : 0u;
    uint32_t b0 = 0;
#pragma unroll
    for (int q = 0; q < HB_SUBS; ++q) b0 = k == (uint32_t)q ? pre[q] : b0;
    return g_ent + ((size_t)k * cap + (size_t)(e - b0));
  };
  // The truth slice is asked for first, then the first four trips' entries (all of them but for the fullest buckets): loads come
  // back in the order they were issued, so the slice is staged while the entries are on their way.  Every lane issues every load
  // -- one with nothing to fetch reads the slice's / the bucket's first entry and ignores it: a load behind a branch made the
  // compiler wait for ALL loads in flight at the first use of any of them (k_join_lean; here it waited for the entries before
  // it had even asked for the slice).
  v4u q4[4];
  auto ask = [&](int t0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t e = (uint32_t)(t0 + u) * XJ_THREADS + (uint32_t)tid;
      q4[u] = __builtin_nontemporal_load(entry_at(t0 + u < ntrips && e < nrec ? e : 0u));   // (nrec > 0 here: entry 0 is there)
    }
  };
  // ---- the truth entries of the bucket's positions: the sorted slice as it is; a bit per position that holds an extended one ----
  {
    const gu32p gk = tn > 0 ? (gu32p)R.xkeys : (gu32p)P.xrows;   // (an empty slice may lie at the end of the table)
    const gi32p gr = tn > 0 ? (gi32p)R.xref : (gi32p)P.xrows, ga = tn > 0 ? (gi32p)R.xalt : (gi32p)P.xrows;
    const uint32_t klast = kbase + ((1u << (R.shift > (uint32_t)DJ_MAX_SHIFT ? (uint32_t)DJ_MAX_SHIFT : R.shift)) - 1u);
    const int j0 = tid < tn ? tid : 0, j1 = tid + XJ_THREADS < tn ? tid + XJ_THREADS : 0;
    const uint32_t k0 = gk[j0], r0 = (uint32_t)gr[j0], a0 = (uint32_t)ga[j0];
    const uint32_t k1 = gk[j1], r1 = (uint32_t)gr[j1], a1 = (uint32_t)ga[j1];
    ask(0);
    __syncthreads();   // the maps are clear
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = tid + h * XJ_THREADS;
      const uint32_t k = h ? k1 : k0, r = h ? r1 : r0, a = h ? a1 : a0;
      if (j < tn) {
        s_xk[j] = k; s_xr[j] = r; s_xa[j] = a;
        if (k >= kbase && k <= klast && (r | a) >= 4u) { const uint32_t pr = (k - kbase) >> 4; atomicOr(&s_p[pr >> 5], 1u << (pr & 31u)); }
      }
    }
  }
  __syncthreads();
  const int nb = P.n_bins;
  int ttop = 0;
  if (tn > 0) ttop = 1 << (31 - __clz(tn));
  unsigned long long* mtp = reinterpret_cast<unsigned long long*>(P.mask_tp);
  uint32_t xs[XJ_TRIPS];   // per trip: position inside the bucket (15 bits) | info << 15 | valid << 28
  // ---- pass 1: records on a position with a truth entry are compared with it ----
  {
    uint4* ring = s_ring + (tid >> 6) * 64;
    uint32_t* ringe = s_ringe + (tid >> 6) * 64;
    const uint32_t lane = (uint32_t)tid & 63u;
    uint32_t nring = 0;   // wave-uniform
    auto drain = [&]() {
      if (lane < nring) {
        const uint4 q = ring[lane];
        const uint32_t e = ringe[lane];
        const uint32_t pr = (q.x & 0xffffffu) >> 4;
        const uint32_t inf = (q.x >> 24) | ((q.y & 0x1fu) << 8);
        const uint32_t b1 = inf & I_BIN1;
        const bool kept = (inf & I_PASS) != 0u;
        bool hit = false;
        if (((s_p[pr >> 5] >> (pr & 31u)) & 1u) && !(inf & I_NOKEY)) {
          const uint32_t key0 = kbase + (pr << 4);
          int j = lds_lower_bound(s_xk, tn, ttop, key0);
          for (; j < tn && (s_xk[j] >> 4) == (key0 >> 4); ++j)
            if (s_xr[j] == q.z && s_xa[j] == q.w) { hit = true; break; }
          if (hit) {
            atomicOr(&s_hit[e >> 5], 1u << (e & 31u));
            if ((inf & I_IDDOT) && b1) atomicMax(&s_ts[j], b1);
            if (kept) atomicOr(&s_tf[j >> 5], 1u << (j & 31));
          }
        }
        if (kept && ((hit && (inf & I_IDDOT)) || (inf & 0x1000u))) {
          const int64_t o = R.src_off + (int64_t)(q.y >> 5);
          atomicOr(mtp + (o >> 6), 1ull << (o & 63));
        }
      }
      nring = 0;
    };
    // four trips' entries are in flight together (trip by trip, every trip paid a memory round trip of its own)
#pragma unroll
    for (int t0 = 0; t0 < XJ_TRIPS; t0 += 4) {
      if (t0 > 0 && t0 < ntrips) ask(t0);   // wave-uniform
#pragma unroll
      for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      xs[t] = 0u;
      if (t < ntrips) {   // wave-uniform
        const uint32_t e = (uint32_t)t * XJ_THREADS + (uint32_t)tid;
        const bool valid = e < nrec;
        const v4u q = q4[u];
        const uint32_t pr = (q[0] & 0xffffffu) >> 4;
        const uint32_t inf = (q[0] >> 24) | ((q[1] & 0x1fu) << 8);
        xs[t] = pr | (inf << 15) | (valid ? 1u << 28 : 0u);
        const bool park = valid && ((((s_p[pr >> 5] >> (pr & 31u)) & 1u) && !(inf & I_NOKEY)) || ((inf & I_PASS) && (inf & 0x1000u)));
        const uint64_t m = ballot64(park);
        const uint32_t cnt = (uint32_t)popc64(m);
        if (nring + cnt > 64u) drain();
        if (park) {
          const uint32_t at = nring + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          ring[at] = make_uint4(q[0], q[1], q[2], q[3]);
          ringe[at] = e;
        }
        nring += cnt;
      }
      }
    }
    if (nring) drain();
  }
  __syncthreads();
  // ---- pass 2: histograms and counts (the hit bits are final); kept records outside the truth set claim their position ----
  uint32_t n_pass = 0, n_tp = 0, top_tp = 0, top_fp = 0;
  uint32_t fpm = 0, nkm = 0;   // bit t: the record of trip t is a kept key outside the truth set / the same without a comparable key
#pragma unroll
  for (int t = 0; t < XJ_TRIPS; ++t) {
    const uint32_t x = xs[t];
    if (t < ntrips && ((x >> 28) & 1u)) {
      const uint32_t e = (uint32_t)t * XJ_THREADS + (uint32_t)tid;
      const uint32_t pr = x & 0x7fffu, inf = (x >> 15) & 0x1fffu;
      const uint32_t hit = (s_hit[e >> 5] >> (e & 31u)) & 1u;
      const uint32_t kept = (inf >> 9) & 1u, keyed = ~(inf >> 11) & 1u;
      const uint32_t tpl = (hit & (inf >> 10)) | ((inf >> 12) & 1u);
      const uint32_t b1 = inf & I_BIN1;
      const uint32_t top = b1 == (uint32_t)nb ? 1u : 0u;
      top_tp += top & tpl;
      top_fp += top & ~tpl;
      if (!top) atomicAdd(tpl ? &s_htp[b1 >> 1] : &s_hfp[b1 >> 1], 1u << (16u * (b1 & 1u)));
      n_pass += kept;
      n_tp += kept & tpl;
      if (kept & ~hit & keyed & 1u) {
        fpm |= 1u << t;
        const uint32_t bit = 1u << (pr & 31u);
        if (atomicOr(&s_s[pr >> 5], bit) & bit) atomicOr(&s_s2[pr >> 5], bit);
      }
      nkm |= (kept & ~hit & ~keyed & 1u) << t;
    }
  }
  __syncthreads();
  // ---- pass 3: a record alone on its position is a distinct key; the others, and the keyless ones, are listed ----
  uint32_t fpr = 0;
#pragma unroll
  for (int t = 0; t < XJ_TRIPS; ++t) {
    if (t < ntrips) {   // wave-uniform
    const uint32_t pr = xs[t] & 0x7fffu;
    const bool mine = (fpm >> t) & 1u;
    const bool shared = mine && ((s_s2[pr >> 5] >> (pr & 31u)) & 1u);
    const bool list = shared || ((nkm >> t) & 1u);
    fpr += (mine && !shared) ? 1u : 0u;
    if (ballot64(list)) {   // rare
      const uint32_t at = wave_reserve(&s_c[6], list);
      if (list) {
        if (at >= (uint32_t)XJ_LIST_MAX) atomicOr(&s_c[4], SPANF_OVERFLOW);
        else {
          const v4u q = *entry_at((uint32_t)t * XJ_THREADS + (uint32_t)tid);   // the allele codes again (the entry is still in L2)
          s_lp[at] = pr | (((nkm >> t) & 1u) << 31);
          s_lr[at] = q[2]; s_la[at] = q[3];
        }
      }
    }
    }
  }
  __syncthreads();
  {
    const uint32_t m = s_c[6] < (uint32_t)XJ_LIST_MAX ? s_c[6] : (uint32_t)XJ_LIST_MAX;
    for (uint32_t i = (uint32_t)tid; i < m; i += XJ_THREADS) {   // all against all: the first of equal records counts
      const uint32_t lp = s_lp[i], lr = s_lr[i], la = s_la[i];
      bool dup = false;
      for (uint32_t j = 0; j < i && !dup; ++j) dup = s_lp[j] == lp && s_lr[j] == lr && s_la[j] == la;
      fpr += dup ? 0u : 1u;
    }
  }
  n_pass = wave_sum(n_pass); n_tp = wave_sum(n_tp); fpr = wave_sum(fpr);
  if (ballot64((top_tp | top_fp) != 0u)) { top_tp = wave_sum(top_tp); top_fp = wave_sum(top_fp); }
  uint32_t tpr = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = tid + h * XJ_THREADS;
    if (j < tn) {
      const uint32_t mx = s_ts[j];
      if (mx) atomicAdd(&s_hu[(mx - 1u) >> 1], 1u << (16u * ((mx - 1u) & 1u)));
      tpr += (s_tf[j >> 5] >> (j & 31)) & 1u;
    }
  }
  tpr = wave_sum(tpr);
  if ((tid & 63) == 0) {
    atomicAdd(&s_c[0], n_pass); atomicAdd(&s_c[1], n_tp); atomicAdd(&s_c[2], fpr);
    if (tpr) atomicAdd(&s_c[3], tpr);
    if (top_tp) atomicAdd(&s_c[7], top_tp);
    if (top_fp) atomicAdd(&s_c[8], top_fp);
  }
  __syncthreads();
  if (tid < 128) {
    const uint32_t ttp = s_c[7], tfp = s_c[8];
    const int b0 = 2 * tid, b1 = 2 * tid + 1;
    auto get = [&](const uint32_t* t, int slot) { return (t[slot >> 1] >> (16 * (slot & 1))) & 0xffffu; };
    oh[tid] = (get(s_htp, 1 + b0) + (b0 == nb - 1 ? ttp : 0u)) | ((get(s_htp, 1 + b1) + (b1 == nb - 1 ? ttp : 0u)) << 16);
    oh[128 + tid] = (get(s_hfp, 1 + b0) + (b0 == nb - 1 ? tfp : 0u)) | ((get(s_hfp, 1 + b1) + (b1 == nb - 1 ? tfp : 0u)) << 16);
    oh[256 + tid] = s_hu[tid];
  }
  if (tid == 0) {
    uint32_t* sc = P.row_scal + orow * 8;
    sc[0] = s_c[0]; sc[1] = s_c[1]; sc[2] = s_c[0] - s_c[1]; sc[3] = s_c[3]; sc[4] = s_c[2]; sc[5] = s_c[4]; sc[6] = 0u; sc[7] = 0u;
  }
}

// TP bits of the sorted scratch VCFs back to input order: only the records that ARE true positives
// (a few per cent) take the random trip, one 64-bit atomic OR each.
__global__ __launch_bounds__(256) void k_sort_scatter_tp(const SortSeg* segs, const int32_t* tile_seg, const uint64_t* sub_mt,
                                                         const uint32_t* perm, uint64_t* mask_tp) {
  const SortSeg sg = segs[tile_seg[blockIdx.x]];
  const int64_t base = (int64_t)((int)blockIdx.x - sg.tile0) * SORT_TILE;
  // one thread per 64-record word of the sorted TP mask (the sorted copies start on 256-record boundaries)
  const int w = (int)threadIdx.x;
  if (w < SORT_TILE / 64) {
    const int64_t i0 = base + (int64_t)w * 64;
    if (i0 < sg.n) {
      uint64_t m = sub_mt[(sg.dst_off + i0) >> 6];
      while (m) {
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        const int64_t i = i0 + bit;
        if (i < sg.n) {
          const int64_t o = sg.src_off + (int64_t)perm[sg.koff + i];
          atomicOr(reinterpret_cast<unsigned long long*>(mask_tp) + (o >> 6), 1ull << (o & 63));
        }
      }
    }
  }
}

// per-tile TP / FP line counts of the redone VCFs from their rebuilt masks (one thread per K1 tile)
__global__ __launch_bounds__(256) void k_tile_counts(const SortSeg* segs, const int32_t* ktile_seg, const int32_t* ktile_local, int nktiles,
                                                     const uint64_t* mp, const uint64_t* mt, uint32_t* tile_tp, uint32_t* tile_fp) {
  // sixteen lanes per tile, one 64-record mask word each (neighbouring lanes read neighbouring words), summed with shuffles
  static_assert(K1_TILE / 64 == 16, "sixteen mask words per tile");
  const int kt = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4);
  const int w = (int)threadIdx.x & 15;
  uint32_t ntp = 0, nfp = 0;
  SortSeg sg;
  int t = 0;
  const bool live = kt < nktiles;
  if (live) {
    sg = segs[ktile_seg[kt]];
    t = ktile_local[kt];
    const int64_t tb = (int64_t)t * K1_TILE;
    if (tb + (int64_t)w * 64 < sg.n) {
      const int64_t i = ((sg.src_off + tb) >> 6) + w;
      const uint64_t bp = mp[i], bt = mt[i];
      ntp = (uint32_t)__popcll(bt);
      nfp = (uint32_t)__popcll(bp & ~bt);
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { ntp += __shfl_xor(ntp, o); nfp += __shfl_xor(nfp, o); }
  if (live && w == 0) {
    tile_tp[sg.main_tile0 + t] = ntp;
    tile_fp[sg.main_tile0 + t] = nfp;
  }
}

// ROC rows and scalars of the sorted scratch VCFs back under the original VCFs
// nparts (or null = one each): two-level bucket path -- the rows of a VCF's partitions, sub_vcf .. sub_vcf + nparts - 1, are summed
// (every count of a row is additive over disjoint ranges of positions; T' is the truth set's, the same in every part)
__global__ __launch_bounds__(256) void k_sort_copy_rows(const SortSeg* segs, const uint64_t* sub_roc, const int64_t* sub_scal,
                                                        uint64_t* roc, int64_t* scal, int n_bins, uint64_t* global_add, const VcfDesc* vcfs,
                                                        const int32_t* nparts, const uint32_t* gate) {
  // gate (bucket path, or null): the chunk's "bad" word of k_finalize, final before this kernel starts -- a bucket overflowed, a
  // position was out of range: the rows are not the VCFs' and the radix sort, which the host starts once it has looked, writes them
  if (gate && *gate) return;
  const SortSeg sg = segs[blockIdx.x];
  const int n = 3 * n_bins;
  const int np = nparts ? nparts[blockIdx.x] : 1;
  for (int i = (int)threadIdx.x; i < n; i += 256) {
    uint64_t v = 0;
    for (int q = 0; q < np; ++q) v += sub_roc[(size_t)(sg.sub_vcf + q) * n + i];
    roc[(size_t)sg.main_vcf * n + i] = v;
    // bucket path: the rows join the per-truth sums only here, once the host knows no bucket overflowed
    if (global_add && v) atomicAdd(reinterpret_cast<unsigned long long*>(global_add) + (size_t)vcfs[sg.main_vcf].truth * n + i, (unsigned long long)v);
  }
  if (threadIdx.x < 8) {
    int64_t v = 0;
    for (int q = 0; q < np; ++q) v += sub_scal[(size_t)(sg.sub_vcf + q) * 8 + threadIdx.x];
    if (threadIdx.x == 5) v = 0;   // QM_S_SORTED: the original was not
    if (threadIdx.x == 6) v = sg.n;   // QM_S_NREC
    if (threadIdx.x == 7 && np > 0) v = sub_scal[(size_t)sg.sub_vcf * 8 + 7];   // QM_S_TRUTH
    scal[(size_t)sg.main_vcf * 8 + threadIdx.x] = v;
  }
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
// bandwidth probes: what this GPU streams with 16 bytes per lane, read only / copy / write only (qm_bw_probe)
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_bw_probe(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, uint32_t* sink) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const int64_t stride = (int64_t)gridDim.x * 256;
  v4u acc = {0u, 0u, 0u, 0u};
  // four independent 16-byte accesses per lane and trip: 4 KiB per wave in flight
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    if (MODE != 2) {
      const v4u x0 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + i));
      const v4u x1 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + i + stride));
      const v4u x2 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + i + 2 * stride));
      const v4u x3 = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + i + 3 * stride));
      if (MODE == 1) {
        __builtin_nontemporal_store(x0, reinterpret_cast<v4u*>(dst + i));
        __builtin_nontemporal_store(x1, reinterpret_cast<v4u*>(dst + i + stride));
        __builtin_nontemporal_store(x2, reinterpret_cast<v4u*>(dst + i + 2 * stride));
        __builtin_nontemporal_store(x3, reinterpret_cast<v4u*>(dst + i + 3 * stride));
      } else {
        acc ^= x0 ^ x1 ^ x2 ^ x3;
      }
    } else {   // write only: plain stores (measured a little faster than non-temporal ones)
      const v4u v = {(unsigned)i, 1u, 2u, 3u};
      *reinterpret_cast<v4u*>(dst + i) = v;
      *reinterpret_cast<v4u*>(dst + i + stride) = v;
      *reinterpret_cast<v4u*>(dst + i + 2 * stride) = v;
      *reinterpret_cast<v4u*>(dst + i + 3 * stride) = v;
    }
  }
  for (; i < n16; i += stride) {
    if (MODE != 2) {
      const v4u x = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + i));
      if (MODE == 1) __builtin_nontemporal_store(x, reinterpret_cast<v4u*>(dst + i)); else acc ^= x;
    } else {
      const v4u v = {(unsigned)i, 1u, 2u, 3u};
      *reinterpret_cast<v4u*>(dst + i) = v;
    }
  }
  if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = 1u;   // keeps the loads alive; never true for the probe's fill pattern
}

// ---------------------------------------------------------------------------
// launchers (called from qmvt_api.cpp through qmvt_dev.h)
// ---------------------------------------------------------------------------
void launch_bw_probe(int mode, const uint8_t* src, uint8_t* dst, int64_t bytes, uint32_t* sink, hipStream_t st) {
  const int64_t n16 = bytes / 16;
  // many short blocks stream best on this part (tools/probe/bw_sweep.hip: 65 536 blocks read 7.0 TB/s where 2 048 read 6.1)
  const dim3 grid(65536), block(256);
  if (mode == 0) hipLaunchKernelGGL((k_bw_probe<0>), grid, block, 0, st, (const uint4*)src, (uint4*)dst, n16, sink);
  else if (mode == 1) hipLaunchKernelGGL((k_bw_probe<1>), grid, block, 0, st, (const uint4*)src, (uint4*)dst, n16, sink);
  else hipLaunchKernelGGL((k_bw_probe<2>), grid, block, 0, st, (const uint4*)src, (uint4*)dst, n16, sink);
}
void launch_classify(const ClassifyParams& P, int n_spans, hipStream_t st) {   // spans P.span_base .. + n_spans
  if (n_spans <= 0) return;
  if (P.pkey && P.ext) hipLaunchKernelGGL((k_classify<true, true>), dim3(n_spans), dim3(64), 0, st, P);
  else if (P.pkey) hipLaunchKernelGGL((k_classify<true, false>), dim3(n_spans), dim3(64), 0, st, P);
  else if (P.ext) hipLaunchKernelGGL((k_classify<false, true>), dim3(n_spans), dim3(64), 0, st, P);
  else hipLaunchKernelGGL((k_classify<false, false>), dim3(n_spans), dim3(64), 0, st, P);
}
void launch_finalize(const FinalizeParams& P, int n_vcf, hipStream_t st) {
  if (n_vcf > 0) hipLaunchKernelGGL(k_finalize, dim3(n_vcf), dim3(256), 0, st, P);
}
void launch_compact(const CompactParams& P, int n_spans, hipStream_t st) {
  if (n_spans > 0) hipLaunchKernelGGL(k_compact, dim3(n_spans * (SPAN_TILES / (K3_WAVES * K3_TILES))), dim3(64 * K3_WAVES), 0, st, P);
}
void launch_masks_to_cls(const uint64_t* mp, const uint64_t* mt, int64_t off, int64_t n, uint8_t* cls, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_masks_to_cls, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mp, mt, off, n, cls);
}
void launch_synth(const SynthParams& S, int n_vcf, int64_t max_n, hipStream_t st) {
  if (n_vcf > 0 && max_n > 0)
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((max_n + 255) / 256), (unsigned)n_vcf), dim3(256), 0, st, S);
}
// first pass of the batch's sort, in two steps because the host needs the OR of the keys in between
void launch_sort_first_hist(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const int32_t* pos_col, uint32_t* hist, uint32_t* orbits,
                            hipStream_t st) {
  if (ntiles > 0)
    hipLaunchKernelGGL((k_sort_hist<true>), dim3(ntiles), dim3(256), 0, st, segs, tile_seg, (const uint32_t*)nullptr, 4, hist, pos_col, orbits);
}
void launch_sort_first_scatter(const SortSeg* segs, const int32_t* tile_seg, int nseg, int ntiles, const SortCols& src, int n_bins, int ext,
                               uint32_t* hist, uint32_t* okeys, uint32_t* oinfs, uint32_t* ovals, int final_dst, uint64_t* mask_pass,
                               uint64_t* mask_tp, hipStream_t st) {
  if (ntiles <= 0) return;
  hipLaunchKernelGGL(k_sort_scan, dim3(nseg), dim3(256), 0, st, segs, hist);
  hipLaunchKernelGGL((k_sort_scatter<true>), dim3(ntiles), dim3(256), 0, st, segs, tile_seg, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                     (const uint32_t*)nullptr, 4, hist, okeys, oinfs, ovals, final_dst, src, n_bins, ext, mask_pass, mask_tp);
}
void launch_sort_gather_alleles(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const uint32_t* perm, const int32_t* src_ref,
                                const int32_t* src_alt, int32_t* dst_ref, int32_t* dst_alt, hipStream_t st) {
  if (ntiles > 0)
    hipLaunchKernelGGL(k_sort_gather_alleles, dim3(ntiles), dim3(256), 0, st, segs, tile_seg, perm, src_ref, src_alt, dst_ref, dst_alt);
}
void launch_sort_pass(const SortSeg* segs, const int32_t* tile_seg, int nseg, int ntiles, const uint32_t* keys, const uint32_t* infs,
                      const uint32_t* vals, int shift, uint32_t* hist, uint32_t* okeys, uint32_t* oinfs, uint32_t* ovals, int final_dst,
                      hipStream_t st) {
  if (ntiles <= 0) return;
  hipLaunchKernelGGL((k_sort_hist<false>), dim3(ntiles), dim3(256), 0, st, segs, tile_seg, keys, shift, hist, (const int32_t*)nullptr,
                     (uint32_t*)nullptr);
  hipLaunchKernelGGL(k_sort_scan, dim3(nseg), dim3(256), 0, st, segs, hist);
  const SortCols none = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL((k_sort_scatter<false>), dim3(ntiles), dim3(256), 0, st, segs, tile_seg, keys, infs, vals, shift, hist, okeys, oinfs, ovals,
                     final_dst, none, 0, 0, (uint64_t*)nullptr, (uint64_t*)nullptr);
}
void launch_sort_scatter_tp(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const uint64_t* sub_mt, const uint32_t* perm,
                            uint64_t* mask_tp, hipStream_t st) {
  if (ntiles > 0) hipLaunchKernelGGL(k_sort_scatter_tp, dim3(ntiles), dim3(256), 0, st, segs, tile_seg, sub_mt, perm, mask_tp);
}
void launch_tile_counts(const SortSeg* segs, const int32_t* ktile_seg, const int32_t* ktile_local, int nktiles, const uint64_t* mp,
                        const uint64_t* mt, uint32_t* tile_tp, uint32_t* tile_fp, hipStream_t st) {
  if (nktiles > 0)
    hipLaunchKernelGGL(k_tile_counts, dim3((unsigned)((nktiles + 15) / 16)), dim3(256), 0, st, segs, ktile_seg, ktile_local, nktiles, mp, mt,
                       tile_tp, tile_fp);
}
void launch_sort_copy_rows(const SortSeg* segs, int nseg, const uint64_t* sub_roc, const int64_t* sub_scal, uint64_t* roc,
                           int64_t* scal, int n_bins, hipStream_t st, uint64_t* global_add, const VcfDesc* vcfs, const int32_t* nparts,
                           const uint32_t* gate) {
  if (nseg > 0) hipLaunchKernelGGL(k_sort_copy_rows, dim3(nseg), dim3(256), 0, st, segs, sub_roc, sub_scal, roc, scal, n_bins, global_add, vcfs, nparts, gate);
}
void launch_classify_hash(const HashParams& P, int nseg, hipStream_t st) {
  if (nseg > 0) hipLaunchKernelGGL(k_classify_hash, dim3(HB_BUCKETS, nseg), dim3(HB_THREADS), 0, st, P);
}
// lb: log2 of the widest bucket key range among the launch's segments (SortSeg.pad), <= DJ_MAX_SHIFT
// nbk: buckets in use at most among the launch's segments (the grid; buckets above a VCF's highest position hold nothing)
void launch_join_lean(const HashParams& P, int nseg, int lb, int nbk, hipStream_t st) {
  if (nseg <= 0) return;
  if (P.scatter_hist) {
    if (lb <= 16) hipLaunchKernelGGL((k_join_lean<16, false>), dim3(nbk, nseg), dim3(LJ_THREADS), 0, st, P);
    else hipLaunchKernelGGL((k_join_lean<DJ_MAX_SHIFT, false>), dim3(nbk, nseg), dim3(LJ_THREADS), 0, st, P);
  } else {
    if (lb <= 16) hipLaunchKernelGGL((k_join_lean<16, true>), dim3(nbk, nseg), dim3(LJ_THREADS), 0, st, P);
    else hipLaunchKernelGGL((k_join_lean<DJ_MAX_SHIFT, true>), dim3(nbk, nseg), dim3(LJ_THREADS), 0, st, P);
  }
}
void launch_join_big(const HashParams& P, int nseg, int nbk, hipStream_t st) {   // buckets of 2^17 positions (SortSeg.pad = DJ_BIG_SHIFT), the scatter's histogram
  if (nseg > 0) hipLaunchKernelGGL((k_join_lean<DJ_BIG_SHIFT, false, true>), dim3(nbk, nseg), dim3(1024), 0, st, P);
}
void launch_join_ext(const HashParams& P, int nseg, int nbk, hipStream_t st) {
  if (nseg > 0) hipLaunchKernelGGL(k_join_ext, dim3(nbk, nseg), dim3(XJ_THREADS), 0, st, P);
}
void launch_bucket_rows(const HashParams& P, int nseg, hipStream_t st) {
  if (nseg > 0) hipLaunchKernelGGL(k_bucket_rows, dim3(nseg), dim3(HB_BUCKETS), 0, st, P);
}
void launch_bucket_scatter(const BucketScatterParams& P, int ntiles, hipStream_t st) {
  if (ntiles <= 0) return;
  if (P.l1_ent) hipLaunchKernelGGL((k_bucket_scatter<true, false>), dim3(ntiles), dim3(512), 0, st, P);
  else if (P.ext && P.pairs) hipLaunchKernelGGL((k_bucket_scatter<false, true, 512>), dim3(ntiles), dim3(512), 0, st, P);
  else if (P.pairs) hipLaunchKernelGGL((k_bucket_scatter<false, false, 512>), dim3(ntiles), dim3(512), 0, st, P);
  else if (P.ext) hipLaunchKernelGGL((k_bucket_scatter<false, true>), dim3(ntiles), dim3(512), 0, st, P);
  else hipLaunchKernelGGL((k_bucket_scatter<false, false>), dim3(ntiles), dim3(512), 0, st, P);
}
void launch_part_hist(const PartParams& P, int ntiles, hipStream_t st) {
  if (ntiles > 0) hipLaunchKernelGGL(k_part_hist, dim3(ntiles), dim3(512), 0, st, P);
}
void launch_part_scatter(const PartParams& P, int ntiles, hipStream_t st) {
  if (ntiles > 0) hipLaunchKernelGGL(k_part_scatter, dim3(ntiles), dim3(512), 0, st, P);
}
__global__ void k_add_u64(unsigned long long* dst, const unsigned long long* src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
void launch_add_u64(uint64_t* dst, const uint64_t* src, int64_t n, hipStream_t st) {
  if (n > 0) hipLaunchKernelGGL(k_add_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned long long*)dst, (const unsigned long long*)src, n);
}
void launch_overlap_pack(const int32_t* pos, const int32_t* ref, const int32_t* alt, const int32_t* set_of, int64_t n,
                         uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t st) {
  hipLaunchKernelGGL(k_overlap_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pos, ref, alt, set_of, n, keys, vals, bad);
}
void launch_overlap_count(const uint32_t* keys, const uint32_t* vals, int64_t n, unsigned long long* regions, hipStream_t st) {
  hipLaunchKernelGGL(k_overlap_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, vals, n, regions);
}

}  // namespace qm
