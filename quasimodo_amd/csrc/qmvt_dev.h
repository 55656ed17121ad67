// qmvt_dev.h -- structures shared by the kernels (qmvt_kernels.hip) and the
// C-ABI host side (qmvt_api.cpp).  Internal; the public surface is include/qmvt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qm {

#ifndef QM_K1_ROUNDS
#define QM_K1_ROUNDS 4
#endif
#ifndef QM_K1_SLICE
#define QM_K1_SLICE 256
#endif
#ifndef QM_SPAN_TILES
#define QM_SPAN_TILES 16
#endif
constexpr int K1_ROUNDS = QM_K1_ROUNDS;      // rounds of 256 records (4 consecutive per lane) per tile
constexpr int K1_TILE = 256 * K1_ROUNDS;     // 1024 records: one LDS truth slice, one TP/FP line count
constexpr int K1_SLICE = QM_K1_SLICE;        // truth keys per LDS slice buffer (two buffers per wave)
constexpr int SPAN_TILES = QM_SPAN_TILES;    // tiles per wave = per workgroup (one histogram flush per span)
constexpr int SPAN_HIST_WORDS = 3 * 128;            // per-span TP / FP / U histograms, two u16 bins per dword
static_assert(QM_SPAN_TILES * 256 * QM_K1_ROUNDS < 65536, "span histograms are u16");
constexpr int VCF_ALIGN = 256;                     // device start of every VCF (records)
#ifndef QM_SORT_TILE
#define QM_SORT_TILE 2048
#endif
constexpr int SORT_TILE = QM_SORT_TILE;      // keys per radix-sort workgroup
constexpr int QM_POS_LIMIT_DEV = 1 << 28;

constexpr uint32_t QMF_PASS = 1u;
constexpr uint32_t QMF_IDDOT = 2u;
constexpr uint32_t QMF_NOKEY = 4u;
constexpr uint32_t QMF_TPLINE = 8u;   // decided on the host: a TP line whatever the key says (include/qmvt.h)
constexpr uint32_t SPANF_UNSORTED = 1u;
constexpr uint32_t SPANF_BADPOS = 2u;
constexpr uint32_t SPANF_RUNLIMIT = 4u;
constexpr uint32_t SPANF_OVERFLOW = 8u;   // bucket path: a bucket does not fit its LDS tables (the VCF is redone by the radix sort)   // allele-extended batch: more records at one position than the dedupe walk allows

// ---- allele codes (include/qmvt.h): 0..3 single base, >= QM_ALLELE_EXT_MIN and non-negative = an
// inline-packed (2..13 bases) or dictionary-interned longer allele; everything else takes no part.
constexpr uint32_t ALLELE_EXT_MIN = 0x08000000u;
__host__ __device__ inline bool allele_valid(int32_t c) {
  const uint32_t u = (uint32_t)c;
  return u < 4u || (u - ALLELE_EXT_MIN) < (0x80000000u - ALLELE_EXT_MIN);
}
// low nibble of the 32-bit key: exact for single-base pairs (ref << 2 | alt: the XOR of two disjoint bit pairs, nothing above
// them), a fold of both codes for the others -- ONE branch-free expression (round 6: the rotate-and-fold of rounds 2-5 behind a
// select was 13 vector instructions and an exec-mask branch per record of an allele-extended batch, which is bound by exactly
// those: profiles/r06_pmc_per_launch_alleles.json).  Bits 2..7 of either code take part: the bases behind an indel's anchor.
__host__ __device__ inline uint32_t allele_nib(int32_t r, int32_t a) {
  const uint32_t ur = (uint32_t)r, ua = (uint32_t)a;
  return ((ur << 2) ^ ua ^ ((ur ^ ua) >> 4)) & 15u;
}

struct TruthDev {
  const uint32_t* keys;  // sorted distinct pos<<4 | ref<<2 | alt
  const int32_t* tidx;   // tidx[b] = first key index with pos >= b << shift; nb + 2 entries
  int32_t shift;
  int32_t nb;
  int64_t n;
  // allele-extended table: every entry (single-base ones included), distinct (pos, ref, alt),
  // sorted by (pos << 4 | allele_nib, ref, alt); same coarse index
  const uint32_t* xkeys;
  const int32_t* xref;
  const int32_t* xalt;
  const int32_t* xtidx;
  int32_t xshift;
  int32_t xnb;
  int64_t xn;
};

struct VcfDesc {
  int64_t off;  // first record (device index, multiple of VCF_ALIGN)
  int64_t n;    // records
  int32_t truth;
  int32_t tile0, ntiles;
  int32_t span0, nspans;
  int32_t pad;     // qm_batch_synth with QM_SYNTH_TRUTH_PER_VCF: the seed of the VCF's synthetic truth set
};

struct SpanDesc {
  int64_t begin, end;  // device record indices, one VCF
  int32_t vcf;
  int32_t tile0;       // global index of the first tile
  // copies of the VCF's off / n / truth: k_classify starts streaming after ONE descriptor load
  int64_t voff;
  int32_t vn;
  int32_t truth;
};

struct ClassifyParams {
  const int32_t* pos;
  const int32_t* ref;
  const int32_t* alt;
  const float* qual;
  const uint8_t* flags;
  const uint32_t* pkey;   // packed input (radix-sort path): key / info pairs instead of the five columns
  const uint32_t* pinf;
  const SpanDesc* spans;
  const VcfDesc* vcfs;
  const TruthDev* truths;
  uint64_t* mask_pass;
  uint64_t* mask_tp;
  uint32_t* tile_tp;
  uint32_t* tile_fp;
  uint32_t* span_hist;  // [n_spans][SPAN_HIST_WORDS]
  uint32_t* span_scal;  // [n_spans][8]
  int32_t n_bins;
  int32_t ext;     // allele-extended batch: k_classify<false, true> against the truth sets' extended tables
  int32_t span_base;   // first span of this launch (the batch is run in a few span ranges so that compaction overlaps classification)
  uint64_t* zero_acc;  // or null: the per-truth sums k_finalize adds to, cleared by the first wave of this launch (saves a memset node per run)
  int32_t zero_words;
  const uint8_t* known;   // or null: known[v] != 0 = an earlier run of this batch found VCF v out of order and its columns have not changed since:
                          // its spans return at once (their rows still say so) and qm_batch_finish sends it down the bucket path without asking
};

struct FinalizeParams {
  const VcfDesc* vcfs;
  const TruthDev* truths;
  const uint32_t* span_hist;
  const uint32_t* span_scal;
  const uint32_t* tile_tp;
  const uint32_t* tile_fp;
  uint32_t* tile_tp_off;
  uint32_t* tile_fp_off;
  uint32_t* vcf_tot;  // [n_vcf][2]: TP lines, FP lines of the VCF (where k_compact's lists end)
  uint64_t* roc;      // [n_vcf][3][n_bins]
  int64_t* scalars;   // [n_vcf][8]
  uint32_t* vcf_flags;
  uint32_t* vcf_posor;   // or null: OR of the positions the optimistic pass saw in each VCF (which position bits are in use)
  uint64_t* global_acc;  // [n_truth][3][n_bins] or null
  int32_t n_bins;
  int32_t ext;           // allele-extended batch: T' is the size of the extended truth table
  int32_t vcf_base;      // first VCF of this launch
  int32_t parts;          // 1 = flags and tile offsets (what k_compact needs), 2 = ROC / scalars / per-truth sums, 3 = both
  const uint8_t* known;   // or null (ClassifyParams.known): a VCF known to be out of order does not raise the summary for being out of order
  // bucket rows only, or null: [n_vcf][SEG_HIST_WORDS] every first-stream record of the "VCF" (a segment of the bucket path) by
  // bin + 1, counted by the scatter.  The rows of the first stream (the first HB_BUCKETS rows of the VCF) then hold no FP
  // histogram: FP = this - the sum of their TP histograms.
  const uint32_t* all_hist;
  const uint32_t* row_cap;   // bucket rows only, or null: [n_vcf] 1 + the highest bucket of the segment that took an entry (the scatter's seg_maxd): the rows
                             // above it were not written (k_join_lean leaves at once for them), and hold nothing
  // or null: host-mapped mirrors [n_vcf] of vcf_flags and of vcf_posor (bucket rows: of row_cap), written beside the device copies so
  // that the host reads them behind its wait for the stream without a copy of its own (two round trips of a first-seen step)
  uint32_t* host_flags;
  uint32_t* host_aux;
  // bucket rows only, or null: ONE word of the chunk, raised when a "VCF" of the launch carries SPANF_OVERFLOW / SPANF_BADPOS or
  // reached a bucket at or above row_cap_limit (the join was launched for the buckets below it): the kernels queued behind
  // (k_sort_copy_rows) then leave the chunk's rows alone -- the host looks at the mirrors only at the end of the step and sends
  // such a chunk through the radix sort
  uint32_t* chunk_bad;
  uint32_t row_cap_limit;
  int32_t lazy_unsorted;   // 1 (qm_batch_run): a VCF found out of order gets its flags written and nothing else -- everything else of it is redone
};

struct CompactParams {
  const VcfDesc* vcfs;
  const SpanDesc* spans;
  const uint64_t* mask_pass;
  const uint64_t* mask_tp;
  const uint32_t* tile_fp;
  const uint32_t* tile_tp_off;
  const uint32_t* tile_fp_off;
  const uint32_t* vcf_tot;     // [n_vcf][2] (k_finalize)
  int32_t* idx;
  const uint32_t* vcf_flags;   // written by k_finalize of the same run
  int32_t skip_unsorted;       // 1: leave VCFs flagged unsorted alone (they are redone); 0: compact everything
  int32_t span_base;           // first span of this launch
  const uint32_t* span_scal;   // (skip_unsorted) the spans' scalar rows of k_classify: word 5 = the span's own flags
};

// one unsorted VCF inside a sort chunk
struct SortSeg {
  int64_t src_off;   // first record of the VCF in the main batch (device index)
  int64_t dst_off;   // first record of its sorted copy in the scratch batch
  int64_t koff;      // offset of the segment in the chunk's key / payload / class arrays
  int64_t hoff;      // offset of the segment's [256][ntiles] digit histogram
  int64_t n;
  int64_t bk_off;    // bucket path: first entry of the segment's [256][8][bk_cap] bucket regions
  int32_t tile0, ntiles;     // sort tiles (SORT_TILE records), numbered over the chunk
  int32_t main_vcf, sub_vcf;
  int32_t main_tile0, pad;   // first K1 tile of the VCF in the main batch; pad: the bucket path's shift (key >> pad = bucket)
  int32_t bk_tile0, bk_cap;  // bucket path: first scatter tile (BK_TILE records) of the segment; entries per sub-region (a power of two)
  int32_t nbk;               // bucket path: buckets in use (those above the VCF's highest position hold nothing)
  uint32_t key_base;         // two-level path: first key of the segment's partition (0 on the one-level path)
  int32_t part;              // one-level scatter over a PARTITION of a VCF's key range, read from the columns (allele-extended VCFs too
                             // large or too wide for 256 buckets): 0 = the whole VCF; 1 = keys outside [key_base, key_base + 256 << pad)
                             // belong to another segment of the same VCF; 2 = the VCF's last partition: keys above it flag the VCF;
                             // + 4: the segment's tiles fill the buckets of the segments that FOLLOW it too (bits 4..7: partitions of the group,
                             //      itself included: 2 with the 512-digit scatter, up to 8 with the 2 048-digit one -- one read of the columns)
};
// bucket path (k_bucket_scatter + k_classify_hash): one workgroup per (segment, bucket of the one scatter pass)
#ifndef QM_BK_TILE
#define QM_BK_TILE 4096
#endif
constexpr int BK_TILE = QM_BK_TILE;      // records per scatter workgroup (512 threads)
constexpr int HB_BUCKETS = 256;
constexpr int HB_SUBS = 8;               // sub-regions of a bucket: tile g of the scatter launch fills sub-region g % 8 (one per XCD)
constexpr int HB_SUB_MAX = 1024;         // entries per sub-region at most
constexpr int HB_MAX_RECORDS = HB_SUBS * HB_SUB_MAX;   // records per bucket (16 per thread of the bucket's workgroup, kept in registers between its two passes)
constexpr int HB_TRUTH_SLOTS = 2048;     // truth keys of the bucket's positions (<= 50 % full)
constexpr int HB_NOKEY_SLOTS = 512;      // kept records without a comparable key
constexpr int HB_MIN_RECORDS = 16384;    // smaller unsorted VCFs take the radix sort
constexpr int HB_INDEX_BITS = 21;        // a bucket entry holds the record's index inside its VCF
constexpr int DJ_TRUTH_MAX = 1024;   // k_join_lean: staged truth keys per bucket (whole cells of the coarse position index); four times as many per wide bucket
constexpr int DJ_BIG_SHIFT = 21;         // the wide buckets of shuffled 10 M-record VCFs: 2^17 positions (k_join_lean<.., BIG>: 48 KB of position maps)
constexpr int DJ_MAX_SHIFT = 19;         // k_join_direct: a bucket's key range (2^shift keys) as ONE bit map in LDS, 64 KB at most
// a bucket entry (8 bytes): key - (bucket << shift) in bits 0..23, info bits 0..11 (bin + 1, PASS, IDDOT, NOKEY) in 24..35,
// the host-decided TP-line bit in 36, the record's index inside the VCF in 37..57
// ---- two-level bucket path (VCFs too large for 256 buckets of 8 192 records): a first scatter deals the records of a VCF to
// the <= 32 PARTITIONS of 2^27 keys (= 256 buckets of 2^19 keys, the widest k_join_direct takes) its positions reach, into
// regions sized exactly by a counting pass; every partition then is a segment of the one-level path, read from the level-1
// entries instead of the columns.
constexpr int P2_SHIFT = DJ_MAX_SHIFT + 8;    // 27
constexpr int P2_PARTS = 32;                  // 2^32 keys / 2^27
constexpr int P2_SUBS = 8;                    // sub-regions of a partition: tile g of the launch fills sub-region g % 8 (one per XCD), each with a cursor
constexpr int P2_INDEX_BITS = 24;             // a level-1 entry holds the record's index inside its half of the VCF (the whole VCF up to 2^24 records)
constexpr int P2_MAX_HALVES = 4;              // ... of which a VCF has at most four: 2^26 records (bit 31 of a bucket entry's second word is k_join_lean's)
constexpr uint32_t P2_DEAD = 0x1fffu;         // info of a record that takes no part (it travels so that the counting pass can count by position alone)
// a level-1 entry (8 bytes): key & (2^27 - 1) in bits 0..26, info (12 bits + the host-decided TP-line bit) in 27..39, index in 40..63
struct PartSeg {
  int64_t src_off;   // first record of the VCF in the main batch
  int64_t n;
  int64_t ent_off;   // first level-1 entry of the VCF
  int32_t tile0;     // first level-1 tile (BK_TILE records) of the VCF in the launch
  int32_t main_vcf;
};
struct PartParams {
  const PartSeg* segs;
  const int32_t* tile_seg;
  const int32_t* pos;
  const int32_t* ref;
  const int32_t* alt;
  const float* qual;
  const uint8_t* flags;
  uint32_t* cnt;              // [n_seg][P2_PARTS * P2_SUBS] records per (partition, sub-region): the counting pass
  const uint32_t* off;        // [n_seg][P2_PARTS * P2_SUBS + 1] where each sub-region starts (entries from ent_off): the host's prefix sums
  uint32_t* cursor;           // [n_seg][P2_PARTS * P2_SUBS] entries written so far; zeroed before the launch
  uint32_t* segflags;         // [n_seg] SPANF_*
  uint64_t* ent;
  uint32_t* mask_pass;        // main batch, as 32-bit words: the kept mask is written here, the TP mask cleared
  uint32_t* mask_tp;
  int32_t n_seg;
  int32_t n_bins;
};
void launch_part_hist(const PartParams& P, int ntiles, hipStream_t st);
void launch_part_scatter(const PartParams& P, int ntiles, hipStream_t st);

struct BucketScatterParams {
  const SortSeg* segs;
  const int32_t* tile_seg;    // segment of every scatter tile
  const int32_t* pos;
  const int32_t* ref;
  const int32_t* alt;
  const float* qual;
  const uint8_t* flags;
  uint32_t* cursor;           // [n_seg][256][8] entries written so far, then [n_seg] flag words (SPANF_*); zeroed before the launch
  uint64_t* ent;
  uint32_t* mask_pass;        // main batch, as 32-bit words: the kept mask is written here, the TP mask cleared
  uint32_t* mask_tp;
  int32_t n_seg;
  int32_t n_bins;
  int32_t tile_base;          // first scatter tile of this launch (the chunk is scattered in a few segment ranges)
  const uint64_t* l1_ent;     // two-level path: the level-1 entries the segments are read from (SortSeg.koff = first entry) instead of the columns
  // allele-extended batches: records whose REF / ALT are not two single bases go to a second stream of 16-byte entries (the
  // ordinary entry with the position's first key, then REF | ALT << 32), regions [n_seg][256][8][bk_cap] with cursors of their own
  uint64_t* xent;
  uint32_t* xcursor;
  int32_t ext;
  int32_t pairs;              // the launch holds tiles whose segment stands for two partitions (SortSeg.part & 4): the 512-digit instantiation
  // or null.  [n_seg][SEG_HIST_WORDS]: every record that leaves as an ordinary (first-stream) entry counted by bin + 1, per segment
  // (zeroed with the cursors).  The scatter waits for memory with its SIMDs and its LDS half idle; k_join_lean is bound by
  // instruction issue: the one histogram that needs no truth set is taken here, the join adds the true positives' and
  // k_finalize takes the difference (FinalizeParams.all_hist).
  uint32_t* seg_hist;
  // two-level path, VCFs above 2^P2_INDEX_BITS records: or null.  [n_seg][4]: a level-1 entry's index is relative to its HALF of the VCF (a run of
  // 2^24 records, level-1 segments of their own whose entries lie one behind the other inside every partition); entry i of the
  // segment's run belongs to half (i >= l1_half[4 seg + 1]) + (i >= l1_half[4 seg + 2]) + (i >= l1_half[4 seg + 3])
  const uint32_t* l1_half;
  uint32_t* seg_maxd;         // or null.  [n_seg] 1 + the highest bucket of the segment that took an entry (zeroed with the cursors): the join need not
                              // launch a workgroup for the buckets above it (the position bits the optimistic pass saw only bound them by a power of two)
};
constexpr int SEG_HIST_WORDS = 260;   // slot = bin + 1 (slot 0: records without a bin), up to 256 bins
// everything k_classify_hash needs to know about one (segment, bucket), laid out by k_bucket_rows before it runs: the
// workgroup of a bucket lives only a few microseconds, and every dependent load on its way to the data (segment table ->
// VCF -> truth set -> position index -> keys) would cost it one memory round trip with nothing else to do
struct HashRow {
  const uint64_t* ent;        // the bucket's [8][cap] entries
  const uint32_t* tkeys;      // the truth keys of the bucket's positions (a superset: whole cells of the coarse position index)
  int64_t src_off;            // first record of the VCF in the main batch
  int32_t tn;                 // how many
  uint32_t cap;               // entries per sub-region
  uint32_t shift;             // (key - the segment's key_base) >> shift = bucket
  uint32_t kbase;             // first key of the bucket
};
// the same for the second stream of an allele-extended batch (k_join_ext): the truth side is the extended table
struct HashRowX {
  const uint64_t* ent;        // the bucket's [8][cap] 16-byte entries
  const uint32_t* xkeys;      // the truth entries of the bucket's positions: keys, REF codes, ALT codes (sorted by key, REF, ALT)
  const int32_t* xref;
  const int32_t* xalt;
  int64_t src_off;
  int32_t tn;
  uint32_t cap;
  uint32_t shift;
  uint32_t kbase;
};
constexpr int XJ_TRUTH_MAX = 1024;   // staged truth entries per bucket
#ifndef XJ_LIST_MAX_
#define XJ_LIST_MAX_ 1024
#endif
constexpr int XJ_LIST_MAX = XJ_LIST_MAX_;    // records of a bucket that need the exact comparison (positions claimed more than once, keyless records)
struct HashParams {
  const SortSeg* segs;
  const HashRow* rows;        // [n_seg * 256]
  HashRow* rows_out;          // the same, for k_bucket_rows
  HashRowX* xrows;            // allele-extended batches: [n_seg * 256] descriptors of the second stream (null otherwise)
  const uint64_t* xent;
  const uint32_t* xcursor;
  int32_t out_stride;         // rows of histograms / scalars per segment: 256, or 512 when the second stream's rows follow the first's
  int32_t ext;
  const uint64_t* ent;
  const uint32_t* cursor;
  const TruthDev* truths;
  const VcfDesc* vcfs;        // main batch (truth set of a segment's VCF)
  uint64_t* mask_tp;          // main batch: TP bits in input order (cleared by the scatter pass)
  uint32_t* row_hist;         // [n_seg * 256][SPAN_HIST_WORDS]
  uint32_t* row_scal;         // [n_seg * 256][8]
  int32_t n_seg;
  int32_t n_bins;
  int32_t seg_base;           // first segment of this launch of k_classify_hash
  int32_t scatter_hist;       // 1: the scatter counted every record by bin (BucketScatterParams.seg_hist): k_join_lean adds no histogram of its own but the true positives'
  const uint32_t* seg_maxd;   // or null: BucketScatterParams.seg_maxd of the scatter in front -- the workgroups of the buckets above leave at once (k_join_lean)
  uint32_t* zero = nullptr;   // k_bucket_rows only, or null: n_zero words it clears on the way (the cursors, flags and counts of the scatter behind it:
  uint32_t n_zero = 0;        // one dispatch instead of a memset's two in front of every chunk)
};
struct SortCols { const int32_t* pos; const int32_t* ref; const int32_t* alt; const float* qual; const uint8_t* flags; };

struct SynthParams {
  const VcfDesc* vcfs;
  int32_t* pos;
  int32_t* ref;
  int32_t* alt;
  float* qual;
  uint8_t* flags;
  int64_t genome_len;
  int64_t truth_n;
  uint64_t truth_seed;
  uint64_t seed;
  uint64_t perm_a, perm_b;
  int32_t shuffled;
  int32_t indel_pct;
  int32_t per_vcf_truth;   // 1: VCF v is generated against the truth set seeded VcfDesc.pad
};

// ---- synthetic workload: the same arithmetic on host and device -------------
__host__ __device__ inline uint64_t mix64(uint64_t x) {  // splitmix64 finalizer
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b) {
  return mix64(mix64(seed ^ 0x51ed270b7f3c9a11ull) + a * 0x9e3779b97f4a7c15ull + b * 0xc2b2ae3d27d4eb4full);
}
__host__ __device__ inline int32_t synth_refbase(int64_t p) { return (int32_t)(hash3(3, (uint64_t)p, 0) & 3u); }
// A synthetic allele (config 5, mixed SNP + indel): length minlen + Geom(0.5) capped at 32.  Up to
// 13 bases it is the inline code (len << 26 | 2-bit bases); longer ones are an id of the synthetic
// dictionary (SYNTH_POOL strings of 14..32 bases, spelled from the id alone, tests/ restate the spelling).
constexpr uint32_t SYNTH_POOL = 1u << 20;
__host__ __device__ inline int32_t synth_allele(uint64_t h, int minlen) {
  int len = minlen;
  uint64_t g = h;
  while (len < 32 && (g & 1u)) { ++len; g >>= 1; }
  const uint64_t bits = mix64(h ^ 0xa11e1e5ull);
  if (len == 1) return (int32_t)(bits & 3u);
  if (len <= 13) return (int32_t)(((uint32_t)len << 26) | (uint32_t)(bits & ((1ull << (2 * len)) - 1ull)));
  return (int32_t)(0x40000000u | (uint32_t)(bits % SYNTH_POOL));
}
// truth entry j of T over genome L: one position per stratum of width L / T; indel_pct % of the
// entries carry longer alleles (0 = the single-base workload of configs 3 / 4)
__host__ __device__ inline void synth_truth(int64_t L, int64_t T, uint64_t tseed, int64_t j, int32_t* p, int32_t* r,
                                             int32_t* a, int indel_pct = 0) {
  const int64_t wt = L / T;
  const uint64_t h = hash3(tseed, (uint64_t)j, 1);
  const int64_t pp = j * wt + 1 + (int64_t)(h % (uint64_t)wt);
  const int32_t rr = synth_refbase(pp);
  *p = (int32_t)pp;
  *r = rr;
  *a = (int32_t)((rr + 1 + (int32_t)((h >> 32) % 3u)) & 3);
  if (indel_pct > 0 && (int)(hash3(tseed, (uint64_t)j, 2) % 100u) < indel_pct) {
    *r = synth_allele(hash3(tseed, (uint64_t)j, 3), 1);
    *a = synth_allele(hash3(tseed, (uint64_t)j, 4), 2);
  }
}
// record i of a VCF with N records: one position per stratum of width L / N;
// takes the truth entry that falls into its stratum with probability 0.8.
__host__ __device__ inline void synth_record(int64_t L, int64_t N, int64_t T, uint64_t tseed, uint64_t seed, int64_t i,
                                              int32_t* p, int32_t* r, int32_t* a, float* q, uint8_t* f, int indel_pct = 0) {
  const int64_t w = L / N;
  const int64_t wt = L / T;
  const int64_t s0 = i * w + 1;  // stratum [s0, s0 + w)
  const int64_t j = (s0 - 1) / wt;
  int32_t tp = 0, tr = 0, ta = 0;
  bool take = false;
  if (j < T) {
    synth_truth(L, T, tseed, j, &tp, &tr, &ta, indel_pct);
    take = tp >= s0 && tp < s0 + w && (hash3(seed, (uint64_t)i, 10) % 10u) < 8u;
  }
  if (take) {
    *p = tp; *r = tr; *a = ta;
  } else {
    const int64_t pp = s0 + (int64_t)(hash3(seed, (uint64_t)i, 11) % (uint64_t)w);
    const int32_t rr = synth_refbase(pp);
    *p = (int32_t)pp;
    *r = rr;
    *a = (int32_t)((rr + 1 + (int32_t)(hash3(seed, (uint64_t)i, 12) % 3u)) & 3);
    if (indel_pct > 0 && (int)(hash3(seed, (uint64_t)i, 14) % 100u) < indel_pct) {
      *r = synth_allele(hash3(seed, (uint64_t)i, 15), 1);
      *a = synth_allele(hash3(seed, (uint64_t)i, 16), 2);
    }
  }
  const uint32_t qi = (uint32_t)(hash3(seed, (uint64_t)i, 13) & 255u);
  *q = (float)qi;
  *f = (uint8_t)(QMF_IDDOT | (qi >= 20u ? QMF_PASS : 0u));
}

// ---- launchers ---------------------------------------------------------------
void launch_classify(const ClassifyParams& P, int n_spans, hipStream_t st);
void launch_finalize(const FinalizeParams& P, int n_vcf, hipStream_t st);
void launch_compact(const CompactParams& P, int n_spans, hipStream_t st);
void launch_masks_to_cls(const uint64_t* mp, const uint64_t* mt, int64_t off, int64_t n, uint8_t* cls, hipStream_t st);
void launch_synth(const SynthParams& S, int n_vcf, int64_t max_n, hipStream_t st);
void launch_classify_hash(const HashParams& P, int nseg, hipStream_t st);   // segments P.seg_base .. + nseg
void launch_bucket_rows(const HashParams& P, int nseg, hipStream_t st);
void launch_join_ext(const HashParams& P, int nseg, int nbk, hipStream_t st);     // the second stream of an allele-extended batch
void launch_join_lean(const HashParams& P, int nseg, int lb, int nbk, hipStream_t st);
void launch_join_big(const HashParams& P, int nseg, int nbk, hipStream_t st);   // k_join_lean<DJ_BIG_SHIFT, false, BIG>: buckets of 2^17 positions, up to 32 768 records   // segments P.seg_base .. + nseg, every bucket shift <= lb <= DJ_MAX_SHIFT
void launch_bucket_scatter(const BucketScatterParams& P, int ntiles, hipStream_t st);   // tiles P.tile_base .. + ntiles; P.l1_ent: from level-1 entries
void launch_sort_first_hist(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const int32_t* pos_col, uint32_t* hist, uint32_t* orbits,
                            hipStream_t st);
void launch_sort_first_scatter(const SortSeg* segs, const int32_t* tile_seg, int nseg, int ntiles, const SortCols& src, int n_bins, int ext,
                               uint32_t* hist, uint32_t* okeys, uint32_t* oinfs, uint32_t* ovals, int final_dst, uint64_t* mask_pass,
                               uint64_t* mask_tp, hipStream_t st);
void launch_sort_gather_alleles(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const uint32_t* perm, const int32_t* src_ref,
                                const int32_t* src_alt, int32_t* dst_ref, int32_t* dst_alt, hipStream_t st);
void launch_sort_pass(const SortSeg* segs, const int32_t* tile_seg, int nseg, int ntiles, const uint32_t* keys, const uint32_t* infs,
                      const uint32_t* vals, int shift, uint32_t* hist, uint32_t* okeys, uint32_t* oinfs, uint32_t* ovals, int final_dst,
                      hipStream_t st);
void launch_sort_scatter_tp(const SortSeg* segs, const int32_t* tile_seg, int ntiles, const uint64_t* sub_mt, const uint32_t* perm,
                            uint64_t* mask_tp, hipStream_t st);
void launch_tile_counts(const SortSeg* segs, const int32_t* ktile_seg, const int32_t* ktile_local, int nktiles, const uint64_t* mp,
                        const uint64_t* mt, uint32_t* tile_tp, uint32_t* tile_fp, hipStream_t st);
void launch_sort_copy_rows(const SortSeg* segs, int nseg, const uint64_t* sub_roc, const int64_t* sub_scal, uint64_t* roc,
                           int64_t* scal, int n_bins, hipStream_t st, uint64_t* global_add = nullptr, const VcfDesc* vcfs = nullptr,
                           const int32_t* nparts = nullptr, const uint32_t* gate = nullptr);
void launch_bw_probe(int mode, const uint8_t* src, uint8_t* dst, int64_t bytes, uint32_t* sink, hipStream_t st);
void launch_add_u64(uint64_t* dst, const uint64_t* src, int64_t n, hipStream_t st);
void launch_overlap_pack(const int32_t* pos, const int32_t* ref, const int32_t* alt, const int32_t* set_of, int64_t n,
                         uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t st);
void launch_overlap_count(const uint32_t* keys, const uint32_t* vals, int64_t n, unsigned long long* regions, hipStream_t st);

}  // namespace qm
