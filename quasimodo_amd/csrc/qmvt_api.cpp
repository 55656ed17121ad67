// qmvt_api.cpp -- C ABI of libqmvt.so (include/qmvt.h): contexts, truth sets,
// resident batches, the sort path for unsorted VCFs, FP overlap.  Host side of
// the HIP engine; the text tokenizer/writers live in qmvt_host.cpp.
//
// There is deliberately no CPU implementation of the classification here: every
// compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "../../include/qmvt.h"
#include "qmvt_dev.h"

using namespace qm;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

void qm_set_error(const char* msg) { g_err = msg ? msg : ""; }   // for the other translation units of the library

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(QM_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct Truth {
  uint32_t* d_keys = nullptr;
  int32_t* d_tidx = nullptr;
  int32_t shift = 0, nb = 0;
  int64_t n = 0;
  // allele-extended table (every valid entry, single-base ones included)
  uint32_t* d_xkeys = nullptr;
  int32_t *d_xref = nullptr, *d_xalt = nullptr, *d_xtidx = nullptr;
  int32_t xshift = 0, xnb = 0;
  int64_t xn = 0;
  // how a synthetic truth set was generated (qm_truth_synth*): lets qm_batch_synth generate every VCF of a batch
  // against the truth set it is assigned to
  bool synthetic = false;
  int64_t syn_L = 0, syn_T = 0;
  uint64_t syn_seed = 0;
  int syn_pct = 0;
  // qm_truth_release: the slot stays (ids are stable) and may be reused by a later load; `gen` tells a batch that was
  // created against an earlier tenant of the slot that its truth set is gone
  bool released = false;
  uint32_t gen = 0;
};

struct qm_ctx {
  int dev = 0;
  hipStream_t stream = nullptr;
  hipStream_t aux = nullptr;     // compaction of one span range runs here, beside the classification of the next
  std::vector<Truth> truths;
  TruthDev* d_truths = nullptr;  // device copy of the descriptors
  int d_truths_cap = 0;
  int64_t path_total[QM_N_PATH_STATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // qm_path_stats_total: every finish of every batch of this context
};

template <typename T>
static int dalloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  const size_t bytes = count * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes);
  if (e != hipSuccess) return fail(QM_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  return QM_OK;
}
#define DALLOC(p, n)                 \
  do {                               \
    int rc_ = dalloc(&(p), (n));     \
    if (rc_ != QM_OK) return rc_;    \
  } while (0)

extern "C" int qm_abi_version(void) { return QM_ABI_VERSION; }
#ifndef QM_KERNELS_ID
#define QM_KERNELS_ID "unknown"
#endif
#ifndef QM_BUILD_ID
#define QM_BUILD_ID "unknown"
#endif
// also findable in the file without loading it (quasimodo_amd/_lib.py: embedded_ids)
extern "C" const char qm_build_marker[] = "@(#)qmvt-ids kernels=" QM_KERNELS_ID " build=" QM_BUILD_ID ";";
extern "C" const char* qm_kernels_id(void) { return QM_KERNELS_ID; }
extern "C" const char* qm_build_id(void) { return QM_BUILD_ID; }
extern "C" const char* qm_last_error(qm_ctx*) { return g_err.c_str(); }

extern "C" int qm_init(int device_id, qm_ctx** out) {
  if (!out) return fail(QM_E_INVAL, "qm_init: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(QM_E_NODEVICE, "qm_init: no HIP device (%s); the engine has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(QM_E_INVAL, "qm_init: device %d out of range (have %d)", device_id, n);
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(QM_E_NODEVICE, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
  qm_ctx* c = new qm_ctx();
  c->dev = device_id;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking);
  if (e != hipSuccess) { if (c->stream) (void)hipStreamDestroy(c->stream); delete c; return fail(QM_E_NODEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
  *out = c;
  return QM_OK;
}

void qm_pipeline_ctx_destroyed(qm_ctx* ctx);   // qmvt_pipeline.cpp: the context's page-locked buffers

extern "C" void qm_destroy(qm_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  qm_pipeline_ctx_destroyed(c);
  for (auto& t : c->truths) {
    (void)hipFree(t.d_keys); (void)hipFree(t.d_tidx);
    (void)hipFree(t.d_xkeys); (void)hipFree(t.d_xref); (void)hipFree(t.d_xalt); (void)hipFree(t.d_xtidx);
  }
  (void)hipFree(c->d_truths);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->aux) (void)hipStreamDestroy(c->aux);
  delete c;
}

// ---------------------------------------------------------------------------
// truth sets
// ---------------------------------------------------------------------------
static int upload_truth_table(qm_ctx* c) {
  const int n = (int)c->truths.size();
  if (n > c->d_truths_cap) {
    (void)hipFree(c->d_truths);
    c->d_truths_cap = std::max(16, n * 2);
    DALLOC(c->d_truths, (size_t)c->d_truths_cap);
  }
  std::vector<TruthDev> h((size_t)n);
  for (int i = 0; i < n; ++i) {
    h[i].keys = c->truths[i].d_keys; h[i].tidx = c->truths[i].d_tidx;
    h[i].shift = c->truths[i].shift; h[i].nb = c->truths[i].nb; h[i].n = c->truths[i].n;
    h[i].xkeys = c->truths[i].d_xkeys; h[i].xref = c->truths[i].d_xref; h[i].xalt = c->truths[i].d_xalt;
    h[i].xtidx = c->truths[i].d_xtidx; h[i].xshift = c->truths[i].xshift; h[i].xnb = c->truths[i].xnb; h[i].xn = c->truths[i].xn;
  }
  HIPCHK(hipMemcpy(c->d_truths, h.data(), sizeof(TruthDev) * (size_t)n, hipMemcpyHostToDevice));
  return QM_OK;
}

// coarse position index over sorted keys (pos << 4 | nibble): tidx[b] = first key with pos >= b << shift
static void coarse_index(const std::vector<uint32_t>& keys, int32_t* shift_out, int32_t* nb_out, std::vector<int32_t>& tidx) {
  const uint32_t maxpos = keys.empty() ? 0u : (keys.back() >> 4);
  int shift = 0;
  while ((maxpos >> shift) >= (1u << 16)) ++shift;  // <= 65536 buckets
  const int32_t nb = (int32_t)(maxpos >> shift);
  tidx.assign((size_t)nb + 2, 0);
  size_t j = 0;
  for (int32_t b = 0; b <= nb + 1; ++b) {
    const uint64_t lim = ((uint64_t)b << shift) << 4;
    while (j < keys.size() && (uint64_t)keys[j] < lim) ++j;
    tidx[(size_t)b] = (int32_t)j;
  }
  tidx[(size_t)nb + 1] = (int32_t)keys.size();
  *shift_out = shift;
  *nb_out = nb;
}

struct XEntry { uint32_t key; int32_t ref, alt; };

// keys: single-base entries; xe: every valid entry (allele-extended batches join against these)
static void truth_free(Truth& t) {
  (void)hipFree(t.d_keys); (void)hipFree(t.d_tidx);
  (void)hipFree(t.d_xkeys); (void)hipFree(t.d_xref); (void)hipFree(t.d_xalt); (void)hipFree(t.d_xtidx);
  t = Truth();
}

static int truth_build(std::vector<uint32_t>& keys, std::vector<XEntry>& xe, Truth& t) {
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  t.n = (int64_t)keys.size();
  std::vector<int32_t> tidx;
  coarse_index(keys, &t.shift, &t.nb, tidx);
  DALLOC(t.d_keys, keys.size());
  DALLOC(t.d_tidx, tidx.size());
  if (!keys.empty()) HIPCHK(hipMemcpy(t.d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(t.d_tidx, tidx.data(), tidx.size() * 4, hipMemcpyHostToDevice));
  {
    auto lt = [](const XEntry& a, const XEntry& b) {
      if (a.key != b.key) return a.key < b.key;
      if (a.ref != b.ref) return a.ref < b.ref;
      return a.alt < b.alt;
    };
    auto eq = [](const XEntry& a, const XEntry& b) { return a.key == b.key && a.ref == b.ref && a.alt == b.alt; };
    std::sort(xe.begin(), xe.end(), lt);
    xe.erase(std::unique(xe.begin(), xe.end(), eq), xe.end());
    t.xn = (int64_t)xe.size();
    std::vector<uint32_t> xk(xe.size());
    std::vector<int32_t> xr(xe.size()), xa(xe.size()), xtidx;
    for (size_t i = 0; i < xe.size(); ++i) { xk[i] = xe[i].key; xr[i] = xe[i].ref; xa[i] = xe[i].alt; }
    coarse_index(xk, &t.xshift, &t.xnb, xtidx);
    DALLOC(t.d_xkeys, xk.size());
    DALLOC(t.d_xref, xk.size());
    DALLOC(t.d_xalt, xk.size());
    DALLOC(t.d_xtidx, xtidx.size());
    if (!xk.empty()) {
      HIPCHK(hipMemcpy(t.d_xkeys, xk.data(), xk.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(t.d_xref, xr.data(), xr.size() * 4, hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(t.d_xalt, xa.data(), xa.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(t.d_xtidx, xtidx.data(), xtidx.size() * 4, hipMemcpyHostToDevice));
  }
  return QM_OK;
}

static int truth_from_keys(qm_ctx* c, std::vector<uint32_t>& keys, std::vector<XEntry>& xe, int* truth_id) {
  Truth t;
  int rc = truth_build(keys, xe, t);
  if (rc != QM_OK) { truth_free(t); return rc; }   // nothing half-built stays behind
  int slot = -1;
  for (size_t i = 0; i < c->truths.size(); ++i) if (c->truths[i].released) { slot = (int)i; break; }
  if (slot >= 0) {   // a released slot is reused: long-lived contexts do not grow without bound
    t.gen = c->truths[(size_t)slot].gen + 1;
    c->truths[(size_t)slot] = t;
  } else {
    c->truths.push_back(t);
    slot = (int)c->truths.size() - 1;
  }
  rc = upload_truth_table(c);
  if (rc != QM_OK) {
    const uint32_t g = c->truths[(size_t)slot].gen;
    truth_free(c->truths[(size_t)slot]);
    if (slot == (int)c->truths.size() - 1 && g == 0) c->truths.pop_back();
    else { c->truths[(size_t)slot].released = true; c->truths[(size_t)slot].gen = g; }
    return rc;
  }
  if (truth_id) *truth_id = slot;
  return QM_OK;
}

extern "C" int qm_truth_release(qm_ctx* c, int truth_id) {
  if (!c || truth_id < 0 || truth_id >= (int)c->truths.size() || c->truths[(size_t)truth_id].released)
    return fail(QM_E_INVAL, "qm_truth_release: no live truth set %d", truth_id);
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->stream));   // nothing of this context still reads it
  const uint32_t g = c->truths[(size_t)truth_id].gen;
  truth_free(c->truths[(size_t)truth_id]);
  c->truths[(size_t)truth_id].released = true;
  c->truths[(size_t)truth_id].gen = g;
  return upload_truth_table(c);
}

extern "C" int qm_truth_load(qm_ctx* c, const int32_t* pos, const int32_t* ref, const int32_t* alt, int64_t n, int* truth_id) {
  if (!c) return fail(QM_E_INVAL, "qm_truth_load: ctx is NULL");
  if (n < 0 || (n > 0 && (!pos || !ref || !alt))) return fail(QM_E_INVAL, "qm_truth_load: bad arguments");
  HIPCHK(hipSetDevice(c->dev));
  std::vector<uint32_t> keys;
  std::vector<XEntry> xe;
  keys.reserve((size_t)n);
  xe.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    if (!allele_valid(ref[i]) || !allele_valid(alt[i])) continue;  // can never match a kept line
    if ((uint32_t)pos[i] >= (uint32_t)QM_POS_LIMIT)
      return fail(QM_E_RANGE, "qm_truth_load: position %d of row %lld outside [0, 2^28)", pos[i], (long long)i);
    const uint32_t key = ((uint32_t)pos[i] << 4) | allele_nib(ref[i], alt[i]);
    xe.push_back(XEntry{key, ref[i], alt[i]});
    if ((uint32_t)(ref[i] | alt[i]) < 4u) keys.push_back(key);   // single-base: what the reference's pattern list holds
  }
  return truth_from_keys(c, keys, xe, truth_id);
}

extern "C" int qm_truth_synth_ext(qm_ctx* c, int64_t L, int64_t T, uint64_t tseed, int indel_pct, int* truth_id) {
  if (!c) return fail(QM_E_INVAL, "qm_truth_synth: ctx is NULL");
  if (T <= 0 || L <= 0 || L % T != 0 || L >= QM_POS_LIMIT) return fail(QM_E_INVAL, "qm_truth_synth: need T | L and L < 2^28");
  if (indel_pct < 0 || indel_pct > 100) return fail(QM_E_INVAL, "qm_truth_synth: indel_pct must be 0..100");
  HIPCHK(hipSetDevice(c->dev));
  std::vector<uint32_t> keys;
  std::vector<XEntry> xe((size_t)T);
  keys.reserve((size_t)T);
  for (int64_t j = 0; j < T; ++j) {
    int32_t p, r, a;
    synth_truth(L, T, tseed, j, &p, &r, &a, indel_pct);
    const uint32_t key = ((uint32_t)p << 4) | allele_nib(r, a);
    xe[(size_t)j] = XEntry{key, r, a};
    if ((uint32_t)(r | a) < 4u) keys.push_back(key);
  }
  int tid = -1;
  int rc = truth_from_keys(c, keys, xe, &tid);
  if (rc != QM_OK) return rc;
  Truth& t = c->truths[(size_t)tid];
  t.synthetic = true; t.syn_L = L; t.syn_T = T; t.syn_seed = tseed; t.syn_pct = indel_pct;
  if (truth_id) *truth_id = tid;
  return QM_OK;
}
extern "C" int qm_truth_synth(qm_ctx* c, int64_t L, int64_t T, uint64_t tseed, int* truth_id) {
  return qm_truth_synth_ext(c, L, T, tseed, 0, truth_id);
}

extern "C" int qm_truth_size(qm_ctx* c, int truth_id, int64_t* n_unique) {
  if (!c || truth_id < 0 || truth_id >= (int)c->truths.size() || c->truths[(size_t)truth_id].released || !n_unique) return fail(QM_E_INVAL, "qm_truth_size: bad arguments");
  *n_unique = c->truths[(size_t)truth_id].n;
  return QM_OK;
}
extern "C" int qm_truth_size_ext(qm_ctx* c, int truth_id, int64_t* n_unique) {
  if (!c || truth_id < 0 || truth_id >= (int)c->truths.size() || c->truths[(size_t)truth_id].released || !n_unique) return fail(QM_E_INVAL, "qm_truth_size_ext: bad arguments");
  *n_unique = c->truths[(size_t)truth_id].xn;
  return QM_OK;
}
extern "C" int qm_truth_count(qm_ctx* c) { return c ? (int)c->truths.size() : 0; }

// ---------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------
struct Layout {
  std::vector<VcfDesc> vcfs;
  std::vector<SpanDesc> spans;
  std::vector<int32_t> tile_vcf;
  int64_t n_pad = 0;     // records allocated
  int64_t n_total = 0;   // records present
  int64_t max_n = 0;
};

static void build_layout(const int64_t* n_records, const int32_t* truth_ids, int n_vcf, Layout& L) {
  L.vcfs.assign((size_t)n_vcf, VcfDesc());
  L.spans.clear();
  L.tile_vcf.clear();
  int64_t off = 0;
  L.n_total = 0;
  L.max_n = 0;
  for (int v = 0; v < n_vcf; ++v) {
    VcfDesc& d = L.vcfs[(size_t)v];
    d.off = off;
    d.n = n_records[v];
    d.truth = truth_ids[v];
    d.tile0 = (int32_t)L.tile_vcf.size();
    d.ntiles = (int32_t)((d.n + K1_TILE - 1) / K1_TILE);
    d.span0 = (int32_t)L.spans.size();
    d.nspans = (d.ntiles + SPAN_TILES - 1) / SPAN_TILES;
    d.pad = 0;
    for (int t = 0; t < d.ntiles; ++t) L.tile_vcf.push_back(v);
    for (int s = 0; s < d.nspans; ++s) {
      SpanDesc sd;
      sd.vcf = v;
      sd.tile0 = d.tile0 + s * SPAN_TILES;
      sd.begin = d.off + (int64_t)s * SPAN_TILES * K1_TILE;
      sd.end = std::min(d.off + d.n, sd.begin + (int64_t)SPAN_TILES * K1_TILE);
      sd.voff = d.off; sd.vn = (int32_t)d.n; sd.truth = d.truth;
      L.spans.push_back(sd);
    }
    off += (d.n + VCF_ALIGN - 1) / VCF_ALIGN * VCF_ALIGN;
    L.n_total += d.n;
    L.max_n = std::max(L.max_n, d.n);
  }
  L.n_pad = off + K1_TILE;  // the last tile's vector loads may run past its VCF
}

struct qm_batch {
  qm_ctx* ctx = nullptr;
  int n_vcf = 0, n_bins = 256;
  Layout L;
  size_t cap_spans = 0, cap_tiles = 0;
  // columns (or, for the radix-sort scratch batch, the packed key / info pairs)
  uint32_t *pkey = nullptr, *pinf = nullptr;
  int32_t *pos = nullptr, *ref = nullptr, *alt = nullptr;
  float* qual = nullptr;
  uint8_t* flags = nullptr;
  // outputs / workspace
  uint64_t *mask_pass = nullptr, *mask_tp = nullptr;
  int32_t* idx = nullptr;
  uint32_t *tile_tp = nullptr, *tile_fp = nullptr, *tile_tp_off = nullptr, *tile_fp_off = nullptr, *vcf_tot = nullptr;
  uint32_t *span_hist = nullptr, *span_scal = nullptr, *vcf_flags = nullptr, *vcf_posor = nullptr;
  // bucket path of the unsorted VCFs: one "span" row per (segment, bucket), fake VCF descriptors for k_finalize
  uint32_t *bk_hist = nullptr, *bk_scal = nullptr;
  VcfDesc* d_bk_vcfs = nullptr;
  int64_t cap_bk_rows = 0, cap_bk_vcfs = 0;
  uint64_t* bk_xent = nullptr;      // allele-extended batches: the second stream's regions (16-byte entries), its cursors and row descriptors
  uint32_t* bk_xcursor = nullptr;
  HashRowX* bk_xrows = nullptr;
  int64_t cap_bk_xent = 0, cap_bk_xcursor = 0, cap_bk_xrows = 0;
  uint64_t* bk_ent = nullptr;       // the bucket regions: [segment][256][8][SortSeg.bk_cap] packed records
  uint32_t* bk_cursor = nullptr;    // [segment][256][8] fill counts, then one flag word per segment
  int32_t* d_bk_tile_seg = nullptr;
  HashRow* bk_rows = nullptr;       // one descriptor per (segment, bucket)
  int64_t cap_bk_rowdesc = 0;
  int64_t cap_bk_ent = 0, cap_bk_cursor = 0, cap_bk_tiles = 0;
  bool bk_tiles_valid = false;      // d_bk_tile_seg holds the tile map of last_segs
  bool bk_fake_valid = false;       // d_bk_vcfs holds the row descriptors of last_segs
  // rows (ROC, scalars, flags) of the VCFs a bucket chunk redoes, before they are copied under the original VCFs
  uint64_t* bk_roc = nullptr;
  int64_t* bk_rscal = nullptr;
  uint32_t* bk_vflags = nullptr;
  int64_t cap_bk_roc = 0, cap_bk_rscal = 0, cap_bk_vflags = 0;
  // two-level bucket path (large VCFs): level-1 descriptors, counts, offsets, cursors, entries; VCF-level segment table
  PartSeg* p_segs = nullptr;
  int32_t* p_tile_seg = nullptr;
  uint32_t *p_cnt = nullptr, *p_off = nullptr, *p_cursor = nullptr, *p_flags = nullptr;
  uint64_t* p_ent = nullptr;
  SortSeg* d_vsegs = nullptr;
  int32_t* d_vparts = nullptr;          // partitions in use per VCF
  int64_t cap_vparts = 0;
  std::vector<int> last2_seg_vcf;       // VCF of every level-2 segment of that chunk
  std::vector<int> last2_vs;            // the chunk these tables were built for, and its counts: a batch run again with the same
  std::vector<uint32_t> last2_cnt;      // VCFs out of order keeps them on the device (as last_segs does on the one-level path)
  int last2_nseg = 0;
  bool last2_halves = false;          // a VCF of the chunk has more than 2^24 records (BucketScatterParams.l1_half)
  uint32_t* p_half = nullptr;
  int64_t cap_p_half = 0;
  int64_t last2_nbt = 0, last2_nkt = 0;
  int64_t cap_p_segs = 0, cap_p_tiles = 0, cap_p_cnt = 0, cap_p_off = 0, cap_p_cursor = 0, cap_p_flags = 0, cap_p_ent = 0, cap_vsegs = 0;
  // throw-away outputs of the rescan after a sort (kept: an allocation per finish costs more than the rescan)
  uint64_t* rs_roc = nullptr;
  int64_t* rs_scal = nullptr;
  uint32_t* rs_flags = nullptr;
  // the segment tables of the last chunk, as uploaded: a batch that is run again with the same VCFs unsorted skips the upload
  std::vector<SortSeg> last_segs;
  uint64_t *roc = nullptr, *global_acc = nullptr;
  int64_t* scalars = nullptr;
  VcfDesc* d_vcfs = nullptr;
  SpanDesc* d_spans = nullptr;
  int32_t* d_tile_vcf = nullptr;
  uint8_t* cls_scratch = nullptr;  // max_n bytes
  int64_t dev_bytes = 0;
  // sort path scratch (lazy): one chunk of unsorted VCFs at a time
  qm_batch* sub = nullptr;               // sorted copies of the chunk's VCFs
  std::vector<int64_t> sub_sig;          // record counts the scratch batch was built for
  std::vector<int32_t> sub_tids;         // truth ids its layout was uploaded with
  uint32_t *sk[2] = {nullptr, nullptr}, *sv[2] = {nullptr, nullptr}, *si[2] = {nullptr, nullptr}, *shist = nullptr, *sorbits = nullptr;
  SortSeg* d_segs = nullptr;
  int32_t *d_tile_seg = nullptr, *d_ktile_seg = nullptr, *d_ktile_local = nullptr;
  int64_t cap_sort_n = 0, cap_sort_hist = 0;
  int cap_segs = 0, cap_stiles = 0, cap_ktiles = 0;
  // The run is pipelined over a few ranges of VCFs: k_compact of range i (issue-bound, writes) runs on the context's
  // second stream beside k_classify of range i + 1 (latency-bound, reads).
  static constexpr int MAX_CHUNKS = 8;
  struct Chunk { int v0, v1, s0, s1; };
  std::vector<Chunk> chunks;
  hipEvent_t ev_sync[MAX_CHUNKS + 2] = {};   // ordering between the two streams (no timing)
  // behind the k_finalize part of the run that writes the per-VCF flags (and their host-mapped mirrors): qm_batch_finish waits for
  // THIS, not for the whole stream -- what it then queues (the bucket path of the VCFs found out of order) is built and launched
  // while the run's compaction and row sums are still going, and starts right behind them
  hipEvent_t ev_flags = nullptr;
  bool flags_recorded = false;
  // bucket chunks whose flags (overflow, bad position, highest bucket) nobody has looked at yet: their last kernels were queued
  // without a round trip through the host (k_sort_copy_rows looks at the chunk's "bad" word itself); qm_batch_finish settles them
  // behind its last wait and sends a chunk that did not fit through the radix sort then
  struct Pending { std::vector<int> vs; int off; int nbk_launch; bool tight, direct; };
  std::vector<Pending> pend;
  int pend_segs = 0;   // mirror words handed out to the chunks of this finish
  // timing
  bool timing = false;
  static constexpr int EV_RING = 32;   // per-kernel events of the latest runs
  static constexpr int EV_PER_RUN = 2 + 5 * MAX_CHUNKS;   // [0] start [1] end, then per chunk: classify start / end, finalize end (main stream); compact start / end (second stream)
  hipEvent_t ev[EV_RING][EV_PER_RUN] = {};
  int ev_chunks[EV_RING] = {};
  int64_t n_timed = 0;
  bool ran = false, finished = false;
  bool ext = false;   // allele-extended: any valid allele code takes part (build-defined widening, config 5)
  uint64_t* last_global = nullptr;
  // The per-truth sums are laid out for the truth sets that existed when the batch was created: later loads on the
  // same context change neither the size of global_acc nor what qm_batch_get_global copies.
  int n_truth = 1;
  std::vector<std::pair<int, uint32_t>> truth_gens;   // (truth id, generation) of every truth set a VCF names
  // host memory the device can write: [16 words, unused since the summary word went][n_vcf flags][n_vcf position bits][n_vcf
  // bucket-row flags][n_vcf highest buckets] -- k_finalize's mirrors of the per-VCF words qm_batch_finish looks at (no copies)
  uint32_t* h_summary = nullptr;
  uint32_t* d_summary = nullptr;
  // where the unsorted VCFs of the last qm_batch_finish went (qm_batch_path_stats)
  int64_t path_stats[QM_N_PATH_STATS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // What a finish found stays known while the columns stay the same (only qm_batch_upload* / qm_batch_synth write them): a VCF found out
  // of order is not streamed by the optimistic pass again, and qm_batch_finish queues its bucket path behind the run without first
  // waiting for the flags (a host round trip of ~ 0.15 ms per step).  QM_MEMO=0: off.
  std::vector<int> lastx_vs;          // bucketx_chunk: the chunk whose tables are on the device (a batch run again)
  std::vector<uint32_t> lastx_por;
  bool lastx_wide = false;            // ... with the wide buckets of 2^17 positions (bucketw_takes)
  std::vector<int> lastx_seg_vcf;     // VCF of every segment
  int lastx_nseg = 0;
  int64_t lastx_nbt = 0, lastx_nkt = 0;
  std::vector<uint8_t> known;         // per VCF: 1 = out of order, as the last finish found it
  std::vector<uint32_t> known_posor;  // its position bits (vcf_posor of that finish)
  // The position bits a VCF is remembered with come from what the optimistic pass SAW before it left (the first 256 records of a
  // span): an estimate that can lie below the VCF's highest position -- the scatter then flags the chunk, the radix sort redoes it,
  // and with the batch's memory on the same would happen on every run.  The radix sort's first pass ORs ALL keys of its chunk
  // (sorbits): those bits are kept here and join the estimate of the chunk's VCFs, so the next finish sizes their buckets for
  // what they really hold.  Cleared with the batch's memory.
  std::vector<uint32_t> posor_seen;
  std::vector<uint32_t> known_nbk;    // 1 + the highest bucket its records reached on the one-level bucket path (0: not known)
  int n_known = 0;
  bool known_dirty = false;           // the device copy is stale
  uint8_t* d_known = nullptr;
  bool run_used_known = false;        // the run in flight was launched with d_known
};

static bool memo_on() {   // read at every run / finish: bench.py times a batch with and without its memory in one process
  const char* e = getenv("QM_MEMO");
  return !e || atoi(e) != 0;
}
static bool flags_event_on() {   // QM_FLAGS_WAIT=stream: qm_batch_finish waits for the whole stream before it looks at the flags (rounds 1-4)
  const char* e = getenv("QM_FLAGS_WAIT");
  return !(e && strcmp(e, "stream") == 0);
}
static void forget_known(qm_batch* b, int v) {   // v < 0: every VCF
  if (!b->posor_seen.empty()) { if (v < 0) std::fill(b->posor_seen.begin(), b->posor_seen.end(), 0u); else b->posor_seen[(size_t)v] = 0u; }
  if (!b->known_nbk.empty()) { if (v < 0) std::fill(b->known_nbk.begin(), b->known_nbk.end(), 0u); else b->known_nbk[(size_t)v] = 0u; }
  if (b->known.empty() || b->n_known == 0) return;
  if (v < 0) { std::fill(b->known.begin(), b->known.end(), (uint8_t)0); b->n_known = 0; b->known_dirty = true; return; }
  if (b->known[(size_t)v]) { b->known[(size_t)v] = 0; --b->n_known; b->known_dirty = true; }
}

static void batch_free(qm_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->dev);
  if (b->sub) { batch_free(b->sub); b->sub = nullptr; }
  void* ptrs[] = {b->pkey, b->pinf, b->pos, b->ref, b->alt, b->qual, b->flags, b->mask_pass, b->mask_tp, b->idx, b->tile_tp, b->tile_fp,
                  b->tile_tp_off, b->tile_fp_off, b->vcf_tot, b->span_hist, b->span_scal, b->vcf_flags, b->vcf_posor, b->bk_hist, b->bk_scal, b->d_bk_vcfs, b->bk_ent, b->bk_cursor, b->d_bk_tile_seg, b->bk_rows, b->bk_xent, b->bk_xcursor, b->bk_xrows, b->bk_roc, b->bk_rscal, b->bk_vflags, b->p_segs, b->p_tile_seg, b->p_cnt, b->p_off, b->p_half, b->p_cursor, b->p_flags, b->p_ent, b->d_vsegs, b->d_vparts, b->rs_roc, b->rs_scal, b->rs_flags, b->roc, b->global_acc,
                  b->scalars, b->d_vcfs, b->d_spans, b->d_tile_vcf, b->cls_scratch, b->sk[0], b->sk[1], b->sv[0],
                  b->sv[1], b->si[0], b->si[1], b->shist, b->sorbits, b->d_segs, b->d_tile_seg, b->d_ktile_seg, b->d_ktile_local, b->d_known};
  for (void* p : ptrs) (void)hipFree(p);
  if (b->h_summary) (void)hipHostFree(b->h_summary);
  for (auto& r : b->ev) for (auto& e : r) if (e) (void)hipEventDestroy(e);
  for (auto& e : b->ev_sync) if (e) (void)hipEventDestroy(e);
  if (b->ev_flags) (void)hipEventDestroy(b->ev_flags);
  delete b;
}

static int upload_layout(qm_batch* b) {
  const Layout& L = b->L;
  HIPCHK(hipMemcpy(b->d_vcfs, L.vcfs.data(), sizeof(VcfDesc) * L.vcfs.size(), hipMemcpyHostToDevice));
  if (!L.spans.empty()) HIPCHK(hipMemcpy(b->d_spans, L.spans.data(), sizeof(SpanDesc) * L.spans.size(), hipMemcpyHostToDevice));
  if (!L.tile_vcf.empty()) HIPCHK(hipMemcpy(b->d_tile_vcf, L.tile_vcf.data(), 4 * L.tile_vcf.size(), hipMemcpyHostToDevice));
  return QM_OK;
}

static int batch_alloc(qm_ctx* c, int n_vcf, const int64_t* n_records, const int32_t* truth_ids, int n_bins, qm_batch** out,
                       bool packed = false, bool packed_alleles = false) {
  qm_batch* b = new qm_batch();
  b->ctx = c;
  b->n_vcf = n_vcf;
  b->n_bins = n_bins;
  build_layout(n_records, truth_ids, n_vcf, b->L);
  const Layout& L = b->L;
  b->cap_spans = L.spans.size();
  b->cap_tiles = L.tile_vcf.size();
  const size_t np = (size_t)L.n_pad;
  const size_t nt = std::max<size_t>(1, (size_t)c->truths.size());
  b->n_truth = (int)nt;
  for (int v = 0; v < n_vcf; ++v) {
    const std::pair<int, uint32_t> tg(truth_ids[v], c->truths[(size_t)truth_ids[v]].gen);
    if (std::find(b->truth_gens.begin(), b->truth_gens.end(), tg) == b->truth_gens.end()) b->truth_gens.push_back(tg);
  }
  int rc = QM_OK;
#define A_(p, n) if (rc == QM_OK) { rc = dalloc(&(p), (n)); if (rc == QM_OK) b->dev_bytes += (int64_t)((n) * sizeof(*(p))); }
  if (packed) {
    A_(b->pkey, np) A_(b->pinf, np)
    if (packed_alleles) { A_(b->ref, np) A_(b->alt, np) }   // sorted copies of the allele codes
  }
  else { A_(b->pos, np) A_(b->ref, np) A_(b->alt, np) A_(b->qual, np) A_(b->flags, np) }
  A_(b->mask_pass, np / 64 + 64) A_(b->mask_tp, np / 64 + 64) A_(b->idx, np)
  A_(b->tile_tp, b->cap_tiles) A_(b->tile_fp, b->cap_tiles) A_(b->tile_tp_off, b->cap_tiles + SPAN_TILES) A_(b->tile_fp_off, b->cap_tiles + SPAN_TILES) A_(b->vcf_tot, (size_t)n_vcf * 2)
  A_(b->span_hist, b->cap_spans * SPAN_HIST_WORDS) A_(b->span_scal, b->cap_spans * 8) A_(b->vcf_flags, (size_t)n_vcf) A_(b->vcf_posor, (size_t)n_vcf)
  A_(b->roc, (size_t)n_vcf * 3 * (size_t)n_bins) A_(b->global_acc, nt * 3 * (size_t)n_bins) A_(b->scalars, (size_t)n_vcf * 8)
  A_(b->d_vcfs, (size_t)n_vcf) A_(b->d_spans, b->cap_spans) A_(b->d_tile_vcf, b->cap_tiles) A_(b->cls_scratch, (size_t)L.max_n)
#undef A_
  if (rc == QM_OK) rc = upload_layout(b);
  if (rc == QM_OK) {
    // ranges of whole VCFs with about equal numbers of spans; small batches stay in one piece
    int want = 1;   // measured on MI355X: more than one range is SLOWER (the two kernels slow each other down by more than the overlap gains; profiles/README.md)
    int min_spans = 4096;   // a range should fill the chip (5 120 waves) a few times over
    if (const char* e = getenv("QM_PIPE_CHUNKS")) { want = atoi(e); min_spans = 1; }   // (tests, tools/gpu_fuzz.py: the ranges asked for, however small the batch)
    want = std::max(1, std::min(want, (int)qm_batch::MAX_CHUNKS));
    const int ns = (int)L.spans.size();
    if (ns < min_spans * want) want = std::max(1, ns / min_spans);
    int v = 0;
    for (int k = 0; k < want && v < n_vcf; ++k) {
      qm_batch::Chunk ck;
      ck.v0 = v; ck.s0 = L.vcfs[(size_t)v].span0;
      const int target = (int)((int64_t)ns * (k + 1) / want);
      while (v < n_vcf && (k == want - 1 || L.vcfs[(size_t)v].span0 + L.vcfs[(size_t)v].nspans <= target || v == ck.v0)) ++v;
      ck.v1 = v; ck.s1 = v < n_vcf ? L.vcfs[(size_t)v].span0 : ns;
      b->chunks.push_back(ck);
    }
    if (b->chunks.empty()) b->chunks.push_back(qm_batch::Chunk{0, n_vcf, 0, ns});
    b->chunks.back().v1 = n_vcf; b->chunks.back().s1 = ns;
    for (auto& e : b->ev_sync) {
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = fail(QM_E_HIP, "hipEventCreate failed"); break; }
    }
    if (rc == QM_OK && hipEventCreateWithFlags(&b->ev_flags, hipEventDisableTiming) != hipSuccess) rc = fail(QM_E_HIP, "hipEventCreate failed");
  }
  if (rc == QM_OK && !packed) {
    // padding lanes are masked in the kernels, but keep the columns defined.  On the context's
    // stream (hipMemset on the null stream would not be ordered before work on a non-blocking stream).
    hipError_t e = hipMemsetAsync(b->flags, 0, np, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b->pos, 0, np * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b->ref, 0, np * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b->alt, 0, np * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b->qual, 0, np * 4, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(QM_E_HIP, "hipMemset: %s", hipGetErrorString(e));
  }
  if (rc == QM_OK && !packed) {   // (the scratch batches of the sort path are finalized without it)
    void* h = nullptr;
    void* d = nullptr;
    // (QM_NO_MIRRORS=1, tests: as if the mapping had failed -- the flags come back by copies, every chunk is looked at before its last kernels)
    if (!getenv("QM_NO_MIRRORS") && hipHostMalloc(&h, 64 + 16 * (size_t)std::max(n_vcf, 1), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      b->h_summary = static_cast<uint32_t*>(h);
      b->d_summary = static_cast<uint32_t*>(d);
      memset(h, 0, 64 + 16 * (size_t)std::max(n_vcf, 1));   // [16 words: the summary][n_vcf flags][n_vcf position bits][n_vcf bucket-row flags][n_vcf highest buckets]: k_finalize's host-mapped mirrors
    } else {   // not fatal: qm_batch_finish then reads the flags back every time
      if (h) (void)hipHostFree(h);
      (void)hipGetLastError();
    }
  }
  if (rc != QM_OK) { batch_free(b); return rc; }
  *out = b;
  return QM_OK;
}

extern "C" int qm_batch_create_ext(qm_ctx* c, int n_vcf, const int64_t* n_records, const int32_t* truth_id_per_vcf, int n_bins,
                                   unsigned mode, qm_batch** out) {
  if (!c || !out || n_vcf <= 0 || !n_records || !truth_id_per_vcf) return fail(QM_E_INVAL, "qm_batch_create: bad arguments");
  if (mode & ~(unsigned)QM_BATCH_ALLELES) return fail(QM_E_INVAL, "qm_batch_create_ext: unknown mode bits 0x%x", mode);
  if (n_bins < 1 || n_bins > QM_MAX_BINS) return fail(QM_E_INVAL, "qm_batch_create: n_bins must be 1..256");
  *out = nullptr;
  for (int v = 0; v < n_vcf; ++v) {
    if (n_records[v] < 0 || n_records[v] > 0x7fffff00ll) return fail(QM_E_INVAL, "qm_batch_create: VCF %d has %lld records", v, (long long)n_records[v]);
    if (truth_id_per_vcf[v] < 0 || truth_id_per_vcf[v] >= (int)c->truths.size() || c->truths[(size_t)truth_id_per_vcf[v]].released)
      return fail(QM_E_INVAL, "qm_batch_create: VCF %d names truth set %d (have %zu)", v, truth_id_per_vcf[v], c->truths.size());
  }
  HIPCHK(hipSetDevice(c->dev));
  int rc = batch_alloc(c, n_vcf, n_records, truth_id_per_vcf, n_bins, out);
  if (rc == QM_OK) (*out)->ext = (mode & QM_BATCH_ALLELES) != 0;
  return rc;
}
extern "C" int qm_batch_create(qm_ctx* c, int n_vcf, const int64_t* n_records, const int32_t* truth_id_per_vcf, int n_bins,
                               qm_batch** out) {
  return qm_batch_create_ext(c, n_vcf, n_records, truth_id_per_vcf, n_bins, 0u, out);
}

extern "C" void qm_batch_destroy(qm_batch* b) { batch_free(b); }
extern "C" int64_t qm_batch_device_bytes(qm_batch* b) { return b ? b->dev_bytes : 0; }
extern "C" int qm_batch_n_truth(qm_batch* b) { return b ? b->n_truth : 0; }

extern "C" int qm_batch_upload(qm_batch* b, int v, const int32_t* pos, const int32_t* ref, const int32_t* alt, const float* qual,
                               const uint8_t* flags) {
  if (!b || v < 0 || v >= b->n_vcf) return fail(QM_E_INVAL, "qm_batch_upload: bad arguments");
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  if (d.n == 0) return QM_OK;
  if (!pos || !ref || !alt || !qual || !flags) return fail(QM_E_INVAL, "qm_batch_upload: NULL column");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const size_t n = (size_t)d.n;
  HIPCHK(hipMemcpy(b->pos + d.off, pos, n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->ref + d.off, ref, n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->alt + d.off, alt, n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->qual + d.off, qual, n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(b->flags + d.off, flags, n, hipMemcpyHostToDevice));
  b->ran = b->finished = false;
  forget_known(b, v);
  return QM_OK;
}

// The same from page-locked host memory (hipHostMalloc / hipHostRegister), asynchronous on `stream` (NULL = the
// context's own stream): the copies of one VCF overlap the tokenising of the next.
extern "C" int qm_batch_upload_async(qm_batch* b, int v, const int32_t* pos, const int32_t* ref, const int32_t* alt, const float* qual,
                                     const uint8_t* flags, void* stream) {
  if (!b || v < 0 || v >= b->n_vcf) return fail(QM_E_INVAL, "qm_batch_upload_async: bad arguments");
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  if (d.n == 0) return QM_OK;
  if (!pos || !ref || !alt || !qual || !flags) return fail(QM_E_INVAL, "qm_batch_upload_async: NULL column");
  HIPCHK(hipSetDevice(b->ctx->dev));
  hipStream_t st = stream ? (hipStream_t)stream : b->ctx->stream;
  const size_t n = (size_t)d.n;
  HIPCHK(hipMemcpyAsync(b->pos + d.off, pos, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->ref + d.off, ref, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->alt + d.off, alt, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->qual + d.off, qual, n * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->flags + d.off, flags, n, hipMemcpyHostToDevice, st));
  b->ran = b->finished = false;
  forget_known(b, v);
  return QM_OK;
}

static uint64_t gcd64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }

extern "C" int qm_batch_synth(qm_batch* b, const qm_synth_cfg* cfg) {
  if (!b || !cfg) return fail(QM_E_INVAL, "qm_batch_synth: bad arguments");
  const Layout& L = b->L;
  for (const VcfDesc& d : L.vcfs) {
    if (d.n <= 0 || cfg->genome_len % d.n != 0) return fail(QM_E_INVAL, "qm_batch_synth: records per VCF must divide genome_len");
    if ((cfg->genome_len / cfg->truth_n) % (cfg->genome_len / d.n) != 0)
      return fail(QM_E_INVAL, "qm_batch_synth: VCF stratum width must divide the truth stratum width");
  }
  if (cfg->truth_n <= 0 || cfg->genome_len % cfg->truth_n != 0 || cfg->genome_len >= QM_POS_LIMIT)
    return fail(QM_E_INVAL, "qm_batch_synth: need truth_n | genome_len < 2^28");
  HIPCHK(hipSetDevice(b->ctx->dev));
  SynthParams S;
  S.vcfs = b->d_vcfs; S.pos = b->pos; S.ref = b->ref; S.alt = b->alt; S.qual = b->qual; S.flags = b->flags;
  S.genome_len = cfg->genome_len; S.truth_n = cfg->truth_n; S.truth_seed = cfg->truth_seed; S.seed = cfg->seed;
  S.per_vcf_truth = 0;
  if (cfg->truth_seed == QM_SYNTH_TRUTH_PER_VCF) {
    // every VCF against the synthetic truth set it was assigned at qm_batch_create (configs[4]: VCF v uses truth v mod 3)
    for (const VcfDesc& d : L.vcfs) {
      const Truth& t = b->ctx->truths[(size_t)d.truth];
      if (!t.synthetic || t.syn_L != cfg->genome_len || t.syn_T != cfg->truth_n || t.syn_pct != cfg->indel_pct || t.syn_seed > 0x7fffffffull)
        return fail(QM_E_INVAL, "qm_batch_synth: truth set %d was not made by qm_truth_synth_ext with this genome_len / truth_n / indel_pct", d.truth);
    }
    for (VcfDesc& d : b->L.vcfs) d.pad = (int32_t)b->ctx->truths[(size_t)d.truth].syn_seed;
    int rc = upload_layout(b);
    if (rc != QM_OK) return rc;
    S.per_vcf_truth = 1;
  }
  S.shuffled = cfg->shuffled;
  if (cfg->shuffled < 0) return fail(QM_E_INVAL, "qm_batch_synth: shuffled must be 0, 1 or a number of runs");
  if (cfg->shuffled > 1)
    for (const VcfDesc& d : L.vcfs) if (d.n > 0 && d.n < cfg->shuffled) return fail(QM_E_INVAL, "qm_batch_synth: %d runs need at least as many records per VCF", cfg->shuffled);
  S.indel_pct = cfg->indel_pct;
  if (cfg->indel_pct < 0 || cfg->indel_pct > 100) return fail(QM_E_INVAL, "qm_batch_synth: indel_pct must be 0..100");
  if (cfg->indel_pct > 0 && !b->ext) return fail(QM_E_INVAL, "qm_batch_synth: indel_pct > 0 needs an allele-extended batch");
  // slot i holds generated record (i * a + b) mod n, a coprime to every n in the batch
  uint64_t a = 2654435761ull;
  for (;;) {
    bool ok = true;
    for (const VcfDesc& d : L.vcfs) if (gcd64(a, (uint64_t)d.n) != 1) { ok = false; break; }
    if (ok) break;
    a += 2;
  }
  S.perm_a = a; S.perm_b = 12345;
  launch_synth(S, b->n_vcf, L.max_n, b->ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(b->ctx->stream));
  b->ran = b->finished = false;
  forget_known(b, -1);
  return QM_OK;
}

static ClassifyParams classify_params(qm_batch* b) {
  ClassifyParams P;
  P.pos = b->pos; P.ref = b->ref; P.alt = b->alt; P.qual = b->qual; P.flags = b->flags;
  P.pkey = b->pkey; P.pinf = b->pinf;
  P.spans = b->d_spans; P.vcfs = b->d_vcfs; P.truths = b->ctx->d_truths;
  P.mask_pass = b->mask_pass; P.mask_tp = b->mask_tp; P.tile_tp = b->tile_tp; P.tile_fp = b->tile_fp;
  P.span_hist = b->span_hist; P.span_scal = b->span_scal; P.n_bins = b->n_bins;
  P.ext = b->ext ? 1 : 0;
  P.span_base = 0;
  P.zero_acc = nullptr; P.zero_words = 0;
  P.known = nullptr;
  return P;
}
static FinalizeParams finalize_params(qm_batch* b, uint64_t* global) {
  FinalizeParams F;
  F.vcfs = b->d_vcfs; F.truths = b->ctx->d_truths; F.span_hist = b->span_hist; F.span_scal = b->span_scal;
  F.tile_tp = b->tile_tp; F.tile_fp = b->tile_fp; F.tile_tp_off = b->tile_tp_off; F.tile_fp_off = b->tile_fp_off; F.vcf_tot = b->vcf_tot;
  F.roc = b->roc; F.scalars = b->scalars; F.vcf_flags = b->vcf_flags; F.vcf_posor = b->vcf_posor; F.global_acc = global; F.n_bins = b->n_bins; F.ext = b->ext ? 1 : 0;
  F.vcf_base = 0;
  F.parts = 3;
  F.known = nullptr;
  F.all_hist = nullptr;
  F.row_cap = nullptr;
  F.host_flags = nullptr; F.host_aux = nullptr;
  F.chunk_bad = nullptr; F.row_cap_limit = 0u; F.lazy_unsorted = 0;
  return F;
}
static CompactParams compact_params(qm_batch* b) {
  CompactParams C;
  C.vcfs = b->d_vcfs; C.spans = b->d_spans; C.mask_pass = b->mask_pass; C.mask_tp = b->mask_tp;
  C.tile_fp = b->tile_fp; C.tile_tp_off = b->tile_tp_off; C.tile_fp_off = b->tile_fp_off; C.vcf_tot = b->vcf_tot; C.idx = b->idx;
  C.vcf_flags = b->vcf_flags; C.skip_unsorted = 1; C.span_base = 0; C.span_scal = b->span_scal;
  return C;
}

extern "C" int qm_batch_set_timing(qm_batch* b, int on) {
  if (!b) return fail(QM_E_INVAL, "qm_batch_set_timing: NULL batch");
  HIPCHK(hipSetDevice(b->ctx->dev));
  b->timing = on != 0;
  b->n_timed = 0;   // averages restart
  if (b->timing && !b->ev[0][0]) for (auto& r : b->ev) for (auto& e : r) HIPCHK(hipEventCreate(&e));
  return QM_OK;
}

extern "C" int qm_batch_run(qm_batch* b, void* stream, void* global_dev) {
  if (!b) return fail(QM_E_INVAL, "qm_batch_run: NULL batch");
  qm_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  uint64_t* g = global_dev ? (uint64_t*)global_dev : b->global_acc;
  for (const auto& tg : b->truth_gens)
    if (tg.first >= (int)c->truths.size() || c->truths[(size_t)tg.first].released || c->truths[(size_t)tg.first].gen != tg.second)
      return fail(QM_E_STATE, "qm_batch_run: truth set %d was released after the batch was created", tg.first);
  const size_t gbytes = (size_t)b->n_truth * 3 * (size_t)b->n_bins * 8;   // as allocated at batch creation
  hipEvent_t* ev = b->ev[b->n_timed % qm_batch::EV_RING];
  const int nch = (int)b->chunks.size();
  const bool T = b->timing;
  if (T) { HIPCHK(hipEventRecord(ev[0], st)); b->ev_chunks[b->n_timed % qm_batch::EV_RING] = nch; }
  // main stream: classify and finalize of every range, in order; second stream: the compaction of a range as soon as its
  // tile offsets exist.  The second stream starts behind everything queued on the main one so far (an earlier run's
  // compaction reads the masks this run rewrites) and the main stream ends behind the last compaction.
  hipStream_t aux = nch > 1 ? c->aux : st;
  // VCFs an earlier finish found out of order, columns unchanged since: their spans return at once (qm_batch: known)
  const bool use_known = memo_on() && b->n_known > 0;
  if (use_known && (b->known_dirty || !b->d_known)) {
    if (!b->d_known) { DALLOC(b->d_known, (size_t)b->n_vcf); b->dev_bytes += b->n_vcf; }
    HIPCHK(hipMemcpy(b->d_known, b->known.data(), (size_t)b->n_vcf, hipMemcpyHostToDevice));
    b->known_dirty = false;
  }
  b->run_used_known = use_known;
  b->flags_recorded = false;
  if (nch > 1) {
    HIPCHK(hipEventRecord(b->ev_sync[qm_batch::MAX_CHUNKS], st));
    HIPCHK(hipStreamWaitEvent(aux, b->ev_sync[qm_batch::MAX_CHUNKS], 0));
  }
  // every VCF known to be out of order: the optimistic pass, its k_finalize and its compaction would find and write what the
  // earlier run left (flags, position bits, rows nobody reads) -- only the per-truth sums are cleared
  const bool all_known = use_known && b->n_known == b->n_vcf;
  for (int k = 0; k < nch; ++k) {
    const qm_batch::Chunk& ck = b->chunks[(size_t)k];
    hipEvent_t* e5 = ev + 2 + 5 * k;
    if (all_known) {
      if (k == 0) HIPCHK(hipMemsetAsync(g, 0, gbytes, st));
      if (T) for (int q = 0; q < 5; ++q) HIPCHK(hipEventRecord(e5[q], st));
      continue;
    }
    ClassifyParams P = classify_params(b);
    P.span_base = ck.s0;
    if (k == 0) {   // k_finalize, behind this launch, adds to the per-truth sums: the first wave of the launch clears them
      if (ck.s1 > ck.s0) { P.zero_acc = g; P.zero_words = (int32_t)(gbytes / 8); }
      else HIPCHK(hipMemsetAsync(g, 0, gbytes, st));   // a batch of empty VCFs launches nothing
    }
    if (use_known) P.known = b->d_known;
    if (T) HIPCHK(hipEventRecord(e5[0], st));
    launch_classify(P, ck.s1 - ck.s0, st);
    if (T) HIPCHK(hipEventRecord(e5[1], st));
    FinalizeParams F = finalize_params(b, g);
    F.vcf_base = ck.v0;
    // (no summary word: with the mirrors, qm_batch_finish looks through the VCFs' flag words itself -- and every workgroup of a
    // batch of shuffled VCFs storing to that ONE word of host memory made this kernel 50 us instead of 15: same-address stores
    // to system memory queue up.)
    if (b->d_summary) { F.host_flags = b->d_summary + 16; F.host_aux = b->d_summary + 16 + b->n_vcf; }
    if (use_known) F.known = b->d_known;
    F.lazy_unsorted = 1;
    // In one piece (the default), the compaction waits only for what it needs of k_finalize -- per-VCF flags and tile offsets --
    // and the rows (ROC, scalars, per-truth sums: 96 MB of span histograms to sum) go to the second stream beside it.
    const bool split = nch == 1 && b->ev_sync[0] != nullptr;
    // (host order: what the compaction -- and a first-seen step's look at the flags -- waits for is queued first, the rows' part
    // behind it; it starts beside the compaction either way)
    if (split) F.parts = 1;
    launch_finalize(F, ck.v1 - ck.v0, st);
    if (k == nch - 1 && b->ev_flags && flags_event_on()) { HIPCHK(hipEventRecord(b->ev_flags, st)); b->flags_recorded = true; }   // every VCF's flags are written
    if (split) HIPCHK(hipEventRecord(b->ev_sync[0], st));
    if (T) HIPCHK(hipEventRecord(e5[2], st));
    CompactParams K = compact_params(b);
    K.span_base = ck.s0;
    if (nch > 1) {
      HIPCHK(hipEventRecord(b->ev_sync[k], st));
      HIPCHK(hipStreamWaitEvent(aux, b->ev_sync[k], 0));
    }
    if (T) HIPCHK(hipEventRecord(e5[3], aux));
    launch_compact(K, ck.s1 - ck.s0, aux);
    if (split) {
      HIPCHK(hipStreamWaitEvent(c->aux, b->ev_sync[0], 0));
      F.parts = 2;
      launch_finalize(F, ck.v1 - ck.v0, c->aux);
      HIPCHK(hipEventRecord(b->ev_sync[1], c->aux));
    }
    if (T) HIPCHK(hipEventRecord(e5[4], aux));
  }
  if (all_known) {
  } else if (nch > 1) {
    HIPCHK(hipEventRecord(b->ev_sync[qm_batch::MAX_CHUNKS + 1], aux));
    HIPCHK(hipStreamWaitEvent(st, b->ev_sync[qm_batch::MAX_CHUNKS + 1], 0));
  } else if (b->ev_sync[0] != nullptr) {
    HIPCHK(hipStreamWaitEvent(st, b->ev_sync[1], 0));   // the rows of k_finalize
  }
  if (T) { HIPCHK(hipEventRecord(ev[1], st)); b->n_timed++; }
  HIPCHK(hipGetLastError());
  b->ran = true;
  b->finished = false;
  b->last_global = g;
  return QM_OK;
}

extern "C" int qm_batch_timings(qm_batch* b, float* ms4) {
  if (!b || !ms4 || !b->timing || b->n_timed == 0) return fail(QM_E_STATE, "qm_batch_timings: timing is off or nothing ran");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const int n = (int)std::min<int64_t>(b->n_timed, qm_batch::EV_RING);
  double acc[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {   // averages over the latest runs since qm_batch_set_timing(1)
    const int slot = (int)((b->n_timed - 1 - i) % qm_batch::EV_RING);
    hipEvent_t* ev = b->ev[slot];
    HIPCHK(hipEventSynchronize(ev[1]));
    float t;
    for (int k = 0; k < b->ev_chunks[slot]; ++k) {   // kernel times add up over the ranges; compaction overlaps the next range's classification
      hipEvent_t* e5 = ev + 2 + 5 * k;
      HIPCHK(hipEventElapsedTime(&t, e5[0], e5[1])); acc[0] += t;
      HIPCHK(hipEventElapsedTime(&t, e5[1], e5[2])); acc[1] += t;
      HIPCHK(hipEventElapsedTime(&t, e5[3], e5[4])); acc[2] += t;
    }
    HIPCHK(hipEventElapsedTime(&t, ev[0], ev[1])); acc[3] += t;
  }
  for (int k = 0; k < 4; ++k) ms4[k] = (float)(acc[k] / n);
  return QM_OK;
}

// ---- sort path: unsorted VCFs are redone in chunks; every step of a chunk is one launch ----
constexpr int64_t SORT_CHUNK_RECORDS = 1ll << 28;
static int64_t sort_chunk_records() {   // QM_SORT_CHUNK_RECORDS (tests, tools/gpu_fuzz.py): small chunks, several of them per finish
  if (const char* e = getenv("QM_SORT_CHUNK_RECORDS")) { const long long v = atoll(e); if (v > 0) return std::min<int64_t>(v, SORT_CHUNK_RECORDS); }
  return SORT_CHUNK_RECORDS;
}

// (Growing an array frees the old one.  A speculative bucket chunk returns with its last kernels still queued -- they read d_segs,
// the cursors' "bad" word, the rows -- and the NEXT chunk of the same finish may grow exactly those: the device is drained first,
// explicitly.  hipFree would do so implicitly; the code must not depend on that.  ADVICE round 5.)
template <typename T>
static int regrow(T** p, int64_t* cap, int64_t need, int64_t* bytes) {
  if (need <= *cap) return QM_OK;
  if (*p) (void)hipDeviceSynchronize();
  (void)hipFree(*p);
  *p = nullptr;
  *bytes -= *cap * (int64_t)sizeof(T);
  *cap = 0;
  int rc = dalloc(p, (size_t)need);
  if (rc != QM_OK) return rc;
  *bytes += need * (int64_t)sizeof(T);
  *cap = need;
  return QM_OK;
}

// Which unsorted VCFs go through the bucket path (k_bucket_scatter + k_classify_hash): default-mode batches; large enough to be
// worth 256 workgroups and 256 histogram rows (smaller ones are sorted in no time); small enough for 256 buckets x 8 sub-regions
// x 1 024 entries at five eighths full and for the 21 index bits of an entry.  QM_SORT_PATH=radix keeps everything on the sort,
// QM_BUCKET_MIN moves the lower limit.
// The knobs that say which path an unsorted VCF takes, read from the environment ONCE per qm_batch_finish (redo_unsorted): the
// predicates below run once per VCF, and ten getenv calls per VCF were 22 us of a first-seen step's host time (256 VCFs) -- between
// the flags' arrival and the first launch of the bucket path, with the device waiting.
struct PathEnv {
  bool radix_only = false;     // QM_SORT_PATH=radix
  int bucket_ext = -1;         // QM_BUCKET_EXT (-1: unset)
  int bucket2 = -1;            // QM_BUCKET2
  int bucketx = -1;            // QM_BUCKETX
  int64_t bucket_min = HB_MIN_RECORDS;   // QM_BUCKET_MIN: tests and tools/gpu_fuzz.py send their small VCFs through the buckets too
};
static thread_local PathEnv g_penv;
static void refresh_path_env() {
  PathEnv e;
  if (const char* v = getenv("QM_SORT_PATH")) e.radix_only = strcmp(v, "radix") == 0;
  if (const char* v = getenv("QM_BUCKET_EXT")) e.bucket_ext = atoi(v);
  if (const char* v = getenv("QM_BUCKET2")) e.bucket2 = atoi(v);
  if (const char* v = getenv("QM_BUCKETX")) e.bucketx = atoi(v);
  if (const char* v = getenv("QM_BUCKET_MIN")) e.bucket_min = atoll(v);
  g_penv = e;
}
static bool bucket_path_takes(const qm_batch* b, int64_t n) {
  if (b->ext && g_penv.bucket_ext == 0) return false;   // allele-extended batches on the radix sort only
  if (g_penv.radix_only) return false;
  return n >= g_penv.bucket_min && n <= (int64_t)HB_BUCKETS * HB_MAX_RECORDS * 5 / 8 && n <= ((int64_t)1 << HB_INDEX_BITS);
}

static bool join_hash_forced() {
  static const bool on = getenv("QM_JOIN") && strcmp(getenv("QM_JOIN"), "hash") == 0;
  return on;
}

// the scratch batch of a chunk of unsorted VCFs: their sorted copies on the radix path, the rows (ROC, scalars, flags) of every
// path; rebuilt only when the chunk's shape changes
static int ensure_sub(qm_batch* b, const std::vector<int>& vs, std::vector<int32_t>& tids) {
  const int nseg = (int)vs.size();
  std::vector<int64_t> sig((size_t)nseg);
  tids.assign((size_t)nseg, 0);
  for (int i = 0; i < nseg; ++i) { sig[(size_t)i] = b->L.vcfs[(size_t)vs[(size_t)i]].n; tids[(size_t)i] = b->L.vcfs[(size_t)vs[(size_t)i]].truth; }
  if (!b->sub || b->sub_sig != sig) {
    if (b->sub) { b->dev_bytes -= b->sub->dev_bytes; batch_free(b->sub); b->sub = nullptr; }
    int rc = batch_alloc(b->ctx, nseg, sig.data(), tids.data(), b->n_bins, &b->sub, true, b->ext);
    if (rc != QM_OK) return rc;
    b->sub->ext = b->ext;
    b->dev_bytes += b->sub->dev_bytes;
    b->sub_sig = sig;
    b->sub_tids = tids;
    b->bk_fake_valid = false;
  } else if (b->sub_tids != tids) {
    b->sub_tids = tids;
    b->bk_fake_valid = false;   // the row descriptors name the truth sets
    for (int i = 0; i < nseg; ++i) b->sub->L.vcfs[(size_t)i].truth = tids[(size_t)i];
    for (SpanDesc& sd : b->sub->L.spans) sd.truth = b->sub->L.vcfs[(size_t)sd.vcf].truth;
    int rc = upload_layout(b->sub);
    if (rc != QM_OK) return rc;
  }
  return QM_OK;
}

// the rows of a bucket chunk's VCFs (k_finalize over the buckets' rows writes them; k_sort_copy_rows takes them home): small
// arrays of their own, so that a bucket chunk needs no scratch batch with room for sorted copies
static int ensure_bucket_rows(qm_batch* b, int nv) {
  int rc = regrow(&b->bk_roc, &b->cap_bk_roc, (int64_t)nv * 3 * b->n_bins, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->bk_rscal, &b->cap_bk_rscal, (int64_t)nv * 8, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->bk_vflags, &b->cap_bk_vflags, (int64_t)nv, &b->dev_bytes);
  return rc;
}
static FinalizeParams bucket_rows_finalize(qm_batch* b, const uint32_t* all_hist) {
  FinalizeParams F = finalize_params(b, nullptr);   // the rows join the per-truth sums only once no bucket is known to have overflowed
  F.all_hist = all_hist;
  F.vcfs = b->d_bk_vcfs; F.span_hist = b->bk_hist; F.span_scal = b->bk_scal; F.vcf_posor = nullptr;
  F.roc = b->bk_roc; F.scalars = b->bk_rscal; F.vcf_flags = b->bk_vflags;
  F.vcf_tot = nullptr;   // (these "VCFs" are rows of buckets: no tiles, no lists -- and not the main batch's numbering)
  return F;
}

// posor[v]: OR of the positions the optimistic pass saw in VCF v (which position bits are in use)
static thread_local double g_ftrace[4];   // QM_FINISH_TRACE: host clock inside the latest sort_chunk (entered, tables ready, scatter queued)
static double ftrace_now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
static int sort_chunk(qm_batch* b, const std::vector<int>& vs, hipStream_t st, uint64_t* global, const std::vector<uint32_t>& posor, bool buckets, bool may_speculate);
// A one-level bucket chunk one of whose VCFs did not fit: `bad` (the VCFs with the flag) through the radix sort, `good` (the others)
// through the buckets once more, looked at before they are handed over.  (Through round 5 the whole chunk took the radix sort: one
// VCF with a crowd of records on a few positions sent up to 4 096 others onto the path that is four times slower.)
static int redo_overflowed(qm_batch* b, const std::vector<int>& bad, const std::vector<int>& good, hipStream_t st, uint64_t* global, const std::vector<uint32_t>& posor) {
  int rc = QM_OK;
  if (!good.empty()) rc = sort_chunk(b, good, st, global, posor, true, false);
  if (rc == QM_OK && !bad.empty()) {
    b->path_stats[QM_PATH_RADIX_AFTER_OVERFLOW] += (int64_t)bad.size();
    rc = sort_chunk(b, bad, st, global, posor, false, false);
    b->path_stats[QM_PATH_RADIX] -= (int64_t)bad.size();
  }
  return rc;
}

// may_speculate = false: the chunk's flags are looked at before its results are handed over (the re-run of the VCFs that fitted, above)
static int sort_chunk(qm_batch* b, const std::vector<int>& vs, hipStream_t st, uint64_t* global, const std::vector<uint32_t>& posor, bool buckets, bool may_speculate = true) {
  g_ftrace[0] = ftrace_now();
  const int nseg = (int)vs.size();
  std::vector<int32_t> tids((size_t)nseg);
  for (int i = 0; i < nseg; ++i) tids[(size_t)i] = b->L.vcfs[(size_t)vs[(size_t)i]].truth;
  if (!buckets) {   // the scratch batch of the sorted copies: only the radix sort needs it (a bucket chunk that overflows makes it below)
    int rc = ensure_sub(b, vs, tids);
    if (rc != QM_OK) return rc;
  }
  // --- segment table and tile maps
  std::vector<SortSeg> segs((size_t)nseg);
  std::vector<int> nbk_used((size_t)nseg, HB_BUCKETS);
  int64_t koff = 0, hoff = 0, bk_ents = 0, nst64 = 0, nkt64 = 0, nbt64 = 0, dst_off = 0;
  for (int i = 0; i < nseg; ++i) {
    const VcfDesc& d = b->L.vcfs[(size_t)vs[(size_t)i]];
    SortSeg& g = segs[(size_t)i];
    g.src_off = d.off; g.dst_off = dst_off; g.koff = koff; g.hoff = hoff; g.n = d.n;   // dst_off: as build_layout lays the scratch batch out
    dst_off += (d.n + VCF_ALIGN - 1) / VCF_ALIGN * VCF_ALIGN;
    g.tile0 = (int32_t)nst64; g.ntiles = (int32_t)((d.n + SORT_TILE - 1) / SORT_TILE);
    g.main_vcf = vs[(size_t)i]; g.sub_vcf = i; g.main_tile0 = d.tile0;
    {   // bucket path: the shift that makes the top eight key bits in use the bucket number (key = pos << 4 | nibble)
      const uint32_t kor = (posor[(size_t)vs[(size_t)i]] << 4) | 15u;
      int msb = 31;
      while (msb > 0 && !((kor >> msb) & 1u)) --msb;
      g.pad = std::max(4, msb - 7);
      nbk_used[(size_t)i] = (int)(kor >> g.pad) + 1;   // buckets above the VCF's highest position are empty by construction: <= 256
      g.nbk = std::min(nbk_used[(size_t)i], (int)HB_BUCKETS); g.key_base = 0u;
      // room per sub-region: between 128 and 256 buckets are in use, a sub-region takes every eighth tile; half as much again on top
      int64_t want = d.n / (128 * HB_SUBS) * 3 / 2 + 16, cap2 = 16;
      while (cap2 < want && cap2 < HB_SUB_MAX) cap2 *= 2;
      g.bk_cap = (int32_t)cap2;
      g.bk_off = bk_ents;
      g.bk_tile0 = (int32_t)nbt64;
      bk_ents += (int64_t)HB_BUCKETS * HB_SUBS * cap2;
      nbt64 += (d.n + BK_TILE - 1) / BK_TILE;
    }
    nst64 += g.ntiles;
    nkt64 += d.ntiles;
    koff += (d.n + 63) / 64 * 64;
    hoff += (int64_t)g.ntiles * 256;
  }
  if (nst64 > INT32_MAX || nkt64 > INT32_MAX || nbt64 > INT32_MAX) return fail(QM_E_LIMIT, "sort chunk: too many tiles");
  const int nst = (int)nst64, nkt = (int)nkt64, nbt = (int)nbt64;
  // the tile maps (which segment a tile belongs to) follow from the segment table and are only spelled out when it changed:
  // a batch that is run again and again with the same VCFs out of order keeps them on the device
  std::vector<int32_t> tile_seg, ktile_seg, ktile_local, bk_tile_seg;
  auto build_tile_maps = [&]() {
    tile_seg.reserve((size_t)nst); ktile_seg.reserve((size_t)nkt); ktile_local.reserve((size_t)nkt); bk_tile_seg.reserve((size_t)nbt);
    for (int i = 0; i < nseg; ++i) {
      const VcfDesc& d = b->L.vcfs[(size_t)vs[(size_t)i]];
      for (int t = 0; t < segs[(size_t)i].ntiles; ++t) tile_seg.push_back(i);
      for (int t = 0; t < d.ntiles; ++t) { ktile_seg.push_back(i); ktile_local.push_back(t); }
      for (int64_t t = 0; t < (d.n + BK_TILE - 1) / BK_TILE; ++t) bk_tile_seg.push_back(i);
    }
  };
  int rc = QM_OK;
  int64_t cap;
  if (rc == QM_OK) { cap = b->cap_segs; rc = regrow(&b->d_segs, &cap, (int64_t)nseg, &b->dev_bytes); b->cap_segs = (int)cap; }
  if (rc == QM_OK) { cap = b->cap_stiles; rc = regrow(&b->d_tile_seg, &cap, (int64_t)nst, &b->dev_bytes); b->cap_stiles = (int)cap; }
  if (rc == QM_OK) {
    cap = b->cap_ktiles; rc = regrow(&b->d_ktile_seg, &cap, (int64_t)nkt, &b->dev_bytes);
    if (rc == QM_OK) { cap = b->cap_ktiles; rc = regrow(&b->d_ktile_local, &cap, (int64_t)nkt, &b->dev_bytes); }
    if (rc == QM_OK) b->cap_ktiles = std::max(b->cap_ktiles, nkt);
  }
  if (rc == QM_OK && !b->sorbits) rc = dalloc(&b->sorbits, 1);
  // The bucket path (k_classify_hash): ONE scatter pass, no sort.  For batches of the default mode whose VCFs are small enough
  // for 256 buckets of at most HB_MAX_RECORDS records; QM_SORT_PATH=radix keeps everything on the radix sort.
  bool try_buckets = buckets;   // the caller chose the chunk's VCFs by size (bucket_path_takes)
  int lb_all = 0, nbk_all = 1;
  for (int i = 0; i < nseg; ++i) { lb_all = std::max(lb_all, (int)segs[(size_t)i].pad); nbk_all = std::max(nbk_all, std::min(nbk_used[(size_t)i], (int)HB_BUCKETS)); }
  const bool direct = lb_all <= DJ_MAX_SHIFT && !join_hash_forced();
  // allele-extended batches: two entry streams and two joins per bucket (k_join_lean for the single-base records, k_join_ext
  // for the others), both of which need the bucket's key range to fit the bit maps; wider key ranges take the radix sort
  const bool xstream = b->ext;
  if (xstream && !direct) try_buckets = false;
  const int out_stride = xstream ? 2 * HB_BUCKETS : HB_BUCKETS;
  if (xstream) nbk_all = HB_BUCKETS;   // every row of a segment is written (the rows of the second stream follow at a fixed distance)
  if (rc == QM_OK && !try_buckets && buckets) rc = ensure_sub(b, vs, tids);
  if (rc == QM_OK && try_buckets) {
    const int64_t rows = (int64_t)nseg * HB_BUCKETS;   // cap_bk_rows counts rows for both arrays
    const int64_t orows = (int64_t)nseg * out_stride;
    int64_t c1 = b->cap_bk_rows * SPAN_HIST_WORDS, c2 = b->cap_bk_rows * 8;
    rc = regrow(&b->bk_hist, &c1, orows * SPAN_HIST_WORDS, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_scal, &c2, orows * 8, &b->dev_bytes);
    if (rc == QM_OK) b->cap_bk_rows = std::max(b->cap_bk_rows, orows);
    if (rc == QM_OK && xstream) {
      rc = regrow(&b->bk_xent, &b->cap_bk_xent, 2 * bk_ents, &b->dev_bytes);
      if (rc == QM_OK) rc = regrow(&b->bk_xcursor, &b->cap_bk_xcursor, rows * HB_SUBS, &b->dev_bytes);
      if (rc == QM_OK) rc = regrow(&b->bk_xrows, &b->cap_bk_xrows, rows, &b->dev_bytes);
    }
    if (rc == QM_OK) rc = regrow(&b->d_bk_vcfs, &b->cap_bk_vcfs, (int64_t)nseg, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_ent, &b->cap_bk_ent, bk_ents, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_rows, &b->cap_bk_rowdesc, rows, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_cursor, &b->cap_bk_cursor, rows * HB_SUBS + nseg + 32 + 16 * 65 + (int64_t)nseg * (SEG_HIST_WORDS + 1) + 1, &b->dev_bytes);   // + 32 + 16 * 65: phase clocks of a profiling build; + 1: the chunk's "bad" word
    if (rc == QM_OK) rc = ensure_bucket_rows(b, nseg);
    if (rc == QM_OK && (int64_t)nbt > b->cap_bk_tiles) {
      b->bk_tiles_valid = false;
      rc = regrow(&b->d_bk_tile_seg, &b->cap_bk_tiles, (int64_t)nbt, &b->dev_bytes);
    }
  }
  if (rc != QM_OK) return rc;
  b->last2_vs.clear(); b->lastx_vs.clear();   // (the two-level path and the partitions path keep their tables in the same arrays)
  const bool same_tables = b->last_segs.size() == segs.size() && memcmp(b->last_segs.data(), segs.data(), sizeof(SortSeg) * segs.size()) == 0;
  if (!same_tables || (try_buckets && !b->bk_tiles_valid)) {
    build_tile_maps();
    if (!same_tables) {
      HIPCHK(hipMemcpyAsync(b->d_segs, segs.data(), sizeof(SortSeg) * segs.size(), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_tile_seg, tile_seg.data(), 4 * tile_seg.size(), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_ktile_seg, ktile_seg.data(), 4 * ktile_seg.size(), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(b->d_ktile_local, ktile_local.data(), 4 * ktile_local.size(), hipMemcpyHostToDevice, st));
      b->bk_tiles_valid = false;
      b->bk_fake_valid = false;
    }
    if (try_buckets && !b->bk_tiles_valid) HIPCHK(hipMemcpyAsync(b->d_bk_tile_seg, bk_tile_seg.data(), 4 * bk_tile_seg.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // the host vectors die with this call
    b->last_segs = segs;
    b->bk_tiles_valid = try_buckets;
  }
  SortCols src = {b->pos, b->ref, b->alt, b->qual, b->flags};
  if (try_buckets) {
    // --- bucket path: ONE scatter on each VCF's top eight key bits into fixed-size bucket regions, then the hash-join per bucket
    if (!b->bk_fake_valid) {   // the buckets of a VCF as the "spans" of a VCF, for k_finalize
      std::vector<VcfDesc> fake((size_t)nseg);
      for (int i = 0; i < nseg; ++i) {
        VcfDesc& f = fake[(size_t)i];
        f = VcfDesc();
        f.off = 0; f.n = segs[(size_t)i].n; f.truth = tids[(size_t)i]; f.tile0 = 0; f.ntiles = 0; f.span0 = i * out_stride;
        f.nspans = xstream ? out_stride : std::min(nbk_used[(size_t)i], (int)HB_BUCKETS); f.pad = 0;
      }
      HIPCHK(hipMemcpyAsync(b->d_bk_vcfs, fake.data(), sizeof(VcfDesc) * fake.size(), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
      b->bk_fake_valid = true;
    }
    const size_t nhist0 = (size_t)nseg * HB_BUCKETS * HB_SUBS + (size_t)nseg + 32 + 16 * 65;   // the scatter's per-segment histograms lie behind the cursors, flags and phase clocks
  const size_t ncur = (nhist0 + (size_t)nseg * (SEG_HIST_WORDS + 1) + 1) * 4;   // (+ 1: seg_maxd behind the histograms; + 1: the chunk's "bad" word behind them)
    BucketScatterParams S;
    S.segs = b->d_segs; S.tile_seg = b->d_bk_tile_seg; S.pos = b->pos; S.ref = b->ref; S.alt = b->alt; S.qual = b->qual; S.flags = b->flags;
    S.cursor = b->bk_cursor; S.ent = b->bk_ent; S.mask_pass = reinterpret_cast<uint32_t*>(b->mask_pass); S.mask_tp = reinterpret_cast<uint32_t*>(b->mask_tp);
    S.n_seg = nseg; S.n_bins = b->n_bins; S.tile_base = 0; S.l1_ent = nullptr;
    S.xent = xstream ? b->bk_xent : nullptr; S.xcursor = xstream ? b->bk_xcursor : nullptr; S.ext = xstream ? 1 : 0; S.pairs = 0;
    uint32_t* const seg_hist = direct ? b->bk_cursor + nhist0 : nullptr;   // k_join_lean follows: the scatter counts every record by bin
    S.seg_hist = seg_hist; S.l1_half = nullptr;
    uint32_t* const seg_maxd = b->bk_cursor + nhist0 + (size_t)nseg * SEG_HIST_WORDS;
    uint32_t* const chunk_bad = seg_maxd + nseg;
    S.seg_maxd = seg_maxd;
    if (xstream) HIPCHK(hipMemsetAsync(b->bk_xcursor, 0, (size_t)nseg * HB_BUCKETS * HB_SUBS * 4, st));
    HashParams H;
    H.segs = b->d_segs; H.rows = b->bk_rows; H.rows_out = b->bk_rows; H.ent = b->bk_ent; H.cursor = b->bk_cursor; H.truths = b->ctx->d_truths; H.vcfs = b->d_vcfs;
    H.mask_tp = b->mask_tp; H.row_hist = b->bk_hist; H.row_scal = b->bk_scal; H.n_seg = nseg; H.n_bins = b->n_bins; H.seg_base = 0;
    H.xrows = xstream ? b->bk_xrows : nullptr; H.xent = b->bk_xent; H.xcursor = b->bk_xcursor; H.out_stride = out_stride; H.ext = xstream ? 1 : 0;
    H.scatter_hist = seg_hist ? 1 : 0;
    H.seg_maxd = nullptr;   // (set below once the launch shape is known)
    H.zero = b->bk_cursor; H.n_zero = (uint32_t)(ncur / 4);   // the scatter's cursors, flags, counts: cleared by the kernel that writes the rows
    g_ftrace[1] = ftrace_now();
    launch_bucket_rows(H, nseg, st);
    H.zero = nullptr; H.n_zero = 0u;
    // The scatter streams (memory-bound, its SIMDs half idle), the join issues instructions (and hardly waits for memory): in a
    // few segment ranges, the join of one range on the second stream beside the scatter of the next, they fill each other's gaps.
    // k_classify_hash issues instructions where the scatter waits for memory: a few segment ranges, the join of one on the second
    // stream beside the scatter of the next, fill each other's gaps (- 7 %).  The bit-map join (then k_join_direct) was bound by the latency of a
    // workgroup's serial steps and wants every LDS slot of the chip: beside a scatter it only loses (3.06 ms in one piece
    // against 3.11 - 3.22 in 2 - 8 ranges, same box)
    int nbk_launch = nbk_all;
    int parts = nseg >= 8 && b->ev_sync[0] && !direct ? 4 : 1;
    if (!b->ev_sync[0]) parts = 1;
    const bool tight_nbk = direct && parts == 1;
    if (tight_nbk) H.seg_maxd = seg_maxd;
    hipStream_t aux = parts > 1 ? b->ctx->aux : st;
    int i0 = 0;
    for (int p = 0; p < parts; ++p) {
      // ranges of about equal record counts
      int i1 = p + 1 == parts ? nseg : i0;
      if (p + 1 < parts) {
        const int64_t want = (int64_t)nbt * (p + 1) / parts;
        while (i1 < nseg && segs[(size_t)i1].bk_tile0 < want) ++i1;
        i1 = std::max(i1, i0);
      }
      if (i1 > i0) {
        const int t0 = segs[(size_t)i0].bk_tile0, t1 = i1 < nseg ? segs[(size_t)i1].bk_tile0 : nbt;
        S.tile_base = t0;
        launch_bucket_scatter(S, t1 - t0, st);
        g_ftrace[2] = ftrace_now();
        if (parts > 1) {
          HIPCHK(hipEventRecord(b->ev_sync[p], st));
          HIPCHK(hipStreamWaitEvent(aux, b->ev_sync[p], 0));
        }
        H.seg_base = i0;
        // The buckets above a VCF's highest position hold nothing, and a workgroup that finds its bucket empty has still held a
        // slot of its CU for a memory round trip: 40 % of the grid on a 5 Mb genome, whose position BITS (all the optimistic pass
        // hands over) bound the buckets in use only by 256 -- 0.09 of the step's 2.65 ms (same box).  The scatter notes the
        // highest bucket it filled per segment (seg_maxd): the join's workgroups above it leave at once and k_finalize sums no
        // row of theirs -- no round trip through the host.
        if (tight_nbk) {   // a batch that ran before remembers the highest bucket of every VCF: nothing is launched above
          bool have = memo_on() && !b->known_nbk.empty();
          uint32_t m = 0;
          if (have) for (int i = 0; i < nseg && have; ++i) { const uint32_t k = b->known_nbk[(size_t)vs[(size_t)i]]; have = k != 0u; m = std::max(m, k); }
          if (have && !xstream) nbk_launch = (int)std::min<uint32_t>(std::max(m, 1u), (uint32_t)nbk_all);   // (two streams: every bucket is launched, the rows of the second follow at a fixed distance)
        }
        // the join: one bit per key of the bucket in LDS where a bucket's key range allows it (k_join_lean: two bits per position), the hashed
        // tables of k_classify_hash otherwise (QM_JOIN=hash: always)
        if (direct) launch_join_lean(H, i1 - i0, lb_all, nbk_launch, aux);
        else launch_classify_hash(H, i1 - i0, aux);
        if (xstream) launch_join_ext(H, i1 - i0, nbk_all, aux);
      }
      i0 = i1;
    }
    if (parts > 1) {
      HIPCHK(hipEventRecord(b->ev_sync[qm_batch::MAX_CHUNKS + 1], aux));
      HIPCHK(hipStreamWaitEvent(st, b->ev_sync[qm_batch::MAX_CHUNKS + 1], 0));
    }
    const bool mirrors = b->d_summary != nullptr && nseg <= b->n_vcf;   // (one segment per VCF on this path)
    // No round trip through the host between the rows' k_finalize and the kernels that hand the chunk's results over: they are queued
    // at once, k_sort_copy_rows looks at the chunk's "bad" word on the device, and qm_batch_finish reads the mirrors behind its last
    // wait (settle_pending) -- 15-25 us per chunk of 2.5 ms.  Every chunk of a finish has mirror words of its own (b->pend_segs).
    const bool speculate = may_speculate && mirrors && b->pend_segs + nseg <= b->n_vcf && !(getenv("QM_SPECULATE") && atoi(getenv("QM_SPECULATE")) == 0);
    const int moff = speculate ? b->pend_segs : 0;
    {
      FinalizeParams F = bucket_rows_finalize(b, seg_hist);
      if (tight_nbk) F.row_cap = seg_maxd;   // (the rows above were not written by this run)
      if (mirrors) { F.host_flags = b->d_summary + 16 + 2 * b->n_vcf + moff; F.host_aux = b->d_summary + 16 + 3 * b->n_vcf + moff; }   // (a place of their own: the run's flags may still be unread)
      if (speculate) { F.chunk_bad = chunk_bad; F.row_cap_limit = (uint32_t)nbk_launch; }
      launch_finalize(F, nseg, st);
    }
    if (speculate) {
      launch_sort_copy_rows(b->d_segs, nseg, b->bk_roc, b->bk_rscal, b->roc, b->scalars, b->n_bins, st, global, b->d_vcfs, nullptr, chunk_bad);
      launch_tile_counts(b->d_segs, b->d_ktile_seg, b->d_ktile_local, nkt, b->mask_pass, b->mask_tp, b->tile_tp, b->tile_fp, st);
      HIPCHK(hipGetLastError());
      b->path_stats[QM_PATH_BUCKET_CHUNKS] += 1;
      b->pend.push_back(qm_batch::Pending{vs, moff, nbk_launch, tight_nbk, direct});
      b->pend_segs += nseg;
      return QM_OK;
    }
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> hfl((size_t)nseg), hmd((size_t)nseg, 0u);
    if (!mirrors) {
      HIPCHK(hipMemcpyAsync(hfl.data(), b->bk_vflags, 4 * hfl.size(), hipMemcpyDeviceToHost, st));
      if (tight_nbk) HIPCHK(hipMemcpyAsync(hmd.data(), seg_maxd, 4 * hmd.size(), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));   // also makes the host tables above safe to free
    if (mirrors) {   // k_finalize's host-mapped mirrors: the flags and (row_cap) the highest buckets, no copy
      memcpy(hfl.data(), b->h_summary + 16 + 2 * b->n_vcf, 4 * hfl.size());
      if (tight_nbk) memcpy(hmd.data(), b->h_summary + 16 + 3 * b->n_vcf, 4 * hmd.size());
    }
    if (tight_nbk) {
      for (int i = 0; i < nseg; ++i) if (hmd[(size_t)i] > (uint32_t)nbk_launch) hfl[(size_t)i] |= SPANF_OVERFLOW;   // (a remembered bound that no longer holds: cannot happen while the columns stay the same)
      if (memo_on()) {
        if (b->known_nbk.empty()) b->known_nbk.assign((size_t)b->n_vcf, 0u);
        for (int i = 0; i < nseg; ++i) b->known_nbk[(size_t)vs[(size_t)i]] = std::max(hmd[(size_t)i], 1u);
      }
    }
    bool overflow = false;
    for (int i = 0; i < nseg; ++i) {
      if (hfl[(size_t)i] & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", vs[(size_t)i]);
      overflow = overflow || (hfl[(size_t)i] & SPANF_OVERFLOW);
    }
    b->path_stats[QM_PATH_BUCKET_CHUNKS] += 1;
    if (!overflow) {
      b->path_stats[direct ? QM_PATH_DIRECT : QM_PATH_HASHED] += nseg;
      launch_sort_copy_rows(b->d_segs, nseg, b->bk_roc, b->bk_rscal, b->roc, b->scalars, b->n_bins, st, global, b->d_vcfs);
      launch_tile_counts(b->d_segs, b->d_ktile_seg, b->d_ktile_local, nkt, b->mask_pass, b->mask_tp, b->tile_tp, b->tile_fp, st);
      HIPCHK(hipGetLastError());
      return QM_OK;   // no wait: the rescan that follows is on the same stream and ends with one
    }
    // a bucket did not fit its tables (dense positions, a dense truth set): the radix sort redoes THAT VCF from the columns; the
    // VCFs of the chunk that fitted take the buckets again, among themselves (nothing of a chunk with a flag is handed over)
    b->path_stats[QM_PATH_OVERFLOW_CHUNKS] += 1;
    std::vector<int> bad, good;
    for (int i = 0; i < nseg; ++i) ((hfl[(size_t)i] & SPANF_OVERFLOW) ? bad : good).push_back(vs[(size_t)i]);
    return redo_overflowed(b, bad, good, st, global, posor);
  } else {
    b->path_stats[QM_PATH_RADIX] += nseg;
  }
  b->path_stats[QM_PATH_RADIX_CHUNKS] += 1;
  qm_batch* s = b->sub;
  // --- 1. + 2. stable LSD radix sort by position (key bits 4..31), only the digits in use.  The first pass packs the
  //        records to (key, info, original index) on the fly; the last pass drops keys and infos straight into the
  //        scratch batch.  (Its ping-pong arrays exist only once a chunk has come this way.)
  for (int i = 0; i < 2 && rc == QM_OK; ++i) {
    cap = b->cap_sort_n; rc = regrow(&b->sk[i], &cap, koff, &b->dev_bytes);
    if (rc == QM_OK) { cap = b->cap_sort_n; rc = regrow(&b->sv[i], &cap, koff, &b->dev_bytes); }
    if (rc == QM_OK) { cap = b->cap_sort_n; rc = regrow(&b->si[i], &cap, koff, &b->dev_bytes); }
  }
  if (rc == QM_OK) b->cap_sort_n = std::max(b->cap_sort_n, koff);
  if (rc == QM_OK) rc = regrow(&b->shist, &b->cap_sort_hist, hoff, &b->dev_bytes);
  if (rc != QM_OK) return rc;
  HIPCHK(hipMemsetAsync(b->sorbits, 0, 4, st));
  launch_sort_first_hist(b->d_segs, b->d_tile_seg, nst, b->pos, b->shist, b->sorbits, st);
  uint32_t orbits = 0;
  HIPCHK(hipMemcpyAsync(&orbits, b->sorbits, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));   // also makes the host tables above safe to free
  if (memo_on()) {   // every position bit in use in this chunk: corrects an under-estimate the VCFs were (or will be) remembered with
    if (b->posor_seen.empty()) b->posor_seen.assign((size_t)b->n_vcf, 0u);
    for (int v : vs) {
      b->posor_seen[(size_t)v] |= orbits >> 4;
      if (!b->known_posor.empty() && b->known[(size_t)v]) b->known_posor[(size_t)v] |= orbits >> 4;
    }
  }
  int npass = 1;
  while (4 + 8 * npass < 32 && (orbits >> (4 + 8 * npass)) != 0) ++npass;
  int cur = 0;
  {
    const bool last = npass == 1;
    launch_sort_first_scatter(b->d_segs, b->d_tile_seg, nseg, nst, src, b->n_bins, b->ext ? 1 : 0, b->shist, last ? s->pkey : b->sk[0],
                              last ? s->pinf : b->si[0], b->sv[0], last ? 1 : 0, b->mask_pass, b->mask_tp, st);
  }
  for (int ps = 1; ps < npass; ++ps) {
    const bool last = ps == npass - 1;
    launch_sort_pass(b->d_segs, b->d_tile_seg, nseg, nst, b->sk[cur], b->si[cur], b->sv[cur], 4 + 8 * ps, b->shist,
                     last ? s->pkey : b->sk[cur ^ 1], last ? s->pinf : b->si[cur ^ 1], b->sv[cur ^ 1], last ? 1 : 0, st);
    cur ^= 1;
  }
  const uint32_t* perm = b->sv[cur];
  if (b->ext) launch_sort_gather_alleles(b->d_segs, b->d_tile_seg, nst, perm, b->ref, b->alt, s->ref, s->alt, st);
  // --- 3. the normal path on the sorted copies (their ROC rows go into the caller's per-truth sums)
  launch_classify(classify_params(s), (int)s->L.spans.size(), st);
  launch_finalize(finalize_params(s, global), nseg, st);
  // --- 4. results back under the original VCFs: ROC + scalar rows, class bits in input order
  launch_sort_copy_rows(b->d_segs, nseg, s->roc, s->scalars, b->roc, b->scalars, b->n_bins, st);
  launch_sort_scatter_tp(b->d_segs, b->d_tile_seg, nst, s->mask_tp, perm, b->mask_tp, st);
  launch_tile_counts(b->d_segs, b->d_ktile_seg, b->d_ktile_local, nkt, b->mask_pass, b->mask_tp, b->tile_tp, b->tile_fp, st);
  HIPCHK(hipGetLastError());
  std::vector<uint32_t> sfl((size_t)nseg);
  HIPCHK(hipMemcpyAsync(sfl.data(), s->vcf_flags, 4 * sfl.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < nseg; ++i) {
    if (sfl[(size_t)i] & SPANF_UNSORTED) return fail(QM_E_HIP, "internal: VCF %d still unsorted after the radix sort", vs[(size_t)i]);
    // the optimistic pass stops streaming at the first out-of-order tile: a bad position behind it shows up only here
    if (sfl[(size_t)i] & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", vs[(size_t)i]);
    if (sfl[(size_t)i] & SPANF_RUNLIMIT)
      return fail(QM_E_LIMIT, "VCF %d repeats one position more than %d times between equal alleles (allele-extended de-duplication limit)",
                  vs[(size_t)i], 1 << 14);
  }
  return QM_OK;
}

// ---- two-level bucket path (qmvt_dev.h): VCFs too large for 256 buckets of 8 192 records ----
static bool bucket2_takes(const qm_batch* b, int64_t n) {
  if (b->ext) return false;
  if (g_penv.radix_only) return false;
  if (g_penv.bucket2 == 0) return false;
  if (join_hash_forced()) return false;
  if (n > (int64_t)P2_MAX_HALVES << P2_INDEX_BITS) return false;   // (VCFs above 2^24 records: in runs of 2^24, level-1 segments of their own)
  if (g_penv.bucket2 == 2) return n > 0;   // tests and tools/gpu_fuzz.py: every unsorted VCF takes the two levels
  return !bucket_path_takes(b, n) && n >= HB_MIN_RECORDS;
}

// *taken = false: the chunk is not for this path after all (a partition too dense for its buckets, a bucket overflowed): the
// caller sends it through the radix sort; nothing the path wrote is kept in that case (the sort rewrites masks, counts and rows)
// bad (when the chunk is not taken): the VCF with a partition too dense for its buckets, or the VCFs whose buckets overflowed
static int bucket2_chunk(qm_batch* b, const std::vector<int>& vs, hipStream_t st, uint64_t* global, bool* taken, std::vector<int>* bad = nullptr) {
  if (bad) bad->clear();
  *taken = false;
  const int nv = (int)vs.size();
  std::vector<int32_t> tids((size_t)nv);
  for (int i = 0; i < nv; ++i) tids[(size_t)i] = b->L.vcfs[(size_t)vs[(size_t)i]].truth;
  int rc = ensure_bucket_rows(b, nv);
  if (rc != QM_OK) return rc;
  // --- level 1: descriptors, tile map, counting pass.  A level-1 entry has 24 index bits: a VCF above 2^24 records is dealt out in
  //     HALVES -- runs of 2^24 records, each a level-1 segment of its own (own counts, own regions, indices relative to its first
  //     record) -- whose regions lie one behind the other inside every partition of the VCF, so that a partition still is ONE run of
  //     entries for the second level, which adds the half's base back (BucketScatterParams.l1_half).
  constexpr int64_t HALF = (int64_t)1 << P2_INDEX_BITS;
  std::vector<PartSeg> ps;
  std::vector<int> half_vcf, vcf_half0((size_t)nv + 1, 0);   // VCF (index into vs) of every half; first half of every VCF
  std::vector<int32_t> ptile;
  int64_t ent_total = 0, nt1 = 0;
  const bool build1 = b->last2_vs != vs;   // the level-1 tables depend on the chunk's VCFs only
  bool any_halves = false;
  for (int i = 0; i < nv; ++i) {
    const VcfDesc& d = b->L.vcfs[(size_t)vs[(size_t)i]];
    vcf_half0[(size_t)i] = (int)ps.size();
    const int nh = (int)std::max<int64_t>(1, (d.n + HALF - 1) / HALF);
    any_halves = any_halves || nh > 1;
    for (int h = 0; h < nh; ++h) {
      PartSeg g;
      g.src_off = d.off + (int64_t)h * HALF; g.n = std::min<int64_t>(HALF, d.n - (int64_t)h * HALF); g.ent_off = ent_total; g.tile0 = (int32_t)nt1; g.main_vcf = vs[(size_t)i];
      const int64_t t = (g.n + BK_TILE - 1) / BK_TILE;
      if (build1) ptile.insert(ptile.end(), (size_t)t, (int32_t)ps.size());
      nt1 += t;
      ps.push_back(g);
      half_vcf.push_back(i);
    }
    ent_total += d.n + 2 * P2_PARTS;   // every partition starts on a 16-byte boundary: at most one entry of padding each
  }
  vcf_half0[(size_t)nv] = (int)ps.size();
  const int nh_all = (int)ps.size();
  if (nt1 > INT32_MAX) return fail(QM_E_LIMIT, "bucket path: too many tiles");
  constexpr int NC = P2_PARTS * P2_SUBS;
  if (rc == QM_OK) rc = regrow(&b->p_segs, &b->cap_p_segs, (int64_t)nh_all, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_tile_seg, &b->cap_p_tiles, nt1, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_cnt, &b->cap_p_cnt, (int64_t)nh_all * NC, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_off, &b->cap_p_off, (int64_t)nh_all * (NC + 1), &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_cursor, &b->cap_p_cursor, (int64_t)nh_all * NC, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_flags, &b->cap_p_flags, (int64_t)nh_all, &b->dev_bytes);
  if (rc == QM_OK) rc = regrow(&b->p_ent, &b->cap_p_ent, ent_total + 64, &b->dev_bytes);
  if (rc != QM_OK) return rc;
  const bool same_vs = b->last2_vs == vs;
  if (!same_vs) {
    b->last2_cnt.clear();
    HIPCHK(hipMemcpyAsync(b->p_segs, ps.data(), sizeof(PartSeg) * ps.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->p_tile_seg, ptile.data(), 4 * ptile.size(), hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemsetAsync(b->p_cnt, 0, (size_t)nh_all * NC * 4, st));
  HIPCHK(hipMemsetAsync(b->p_cursor, 0, (size_t)nh_all * NC * 4, st));
  HIPCHK(hipMemsetAsync(b->p_flags, 0, (size_t)nh_all * 4, st));
  PartParams PP;
  PP.segs = b->p_segs; PP.tile_seg = b->p_tile_seg; PP.pos = b->pos; PP.ref = b->ref; PP.alt = b->alt; PP.qual = b->qual; PP.flags = b->flags;
  PP.cnt = b->p_cnt; PP.off = b->p_off; PP.cursor = b->p_cursor; PP.segflags = b->p_flags; PP.ent = b->p_ent;
  PP.mask_pass = reinterpret_cast<uint32_t*>(b->mask_pass); PP.mask_tp = reinterpret_cast<uint32_t*>(b->mask_tp); PP.n_seg = nh_all; PP.n_bins = b->n_bins;
  launch_part_hist(PP, (int)nt1, st);
  std::vector<uint32_t> cnt((size_t)nh_all * NC), pfl((size_t)nh_all);
  HIPCHK(hipMemcpyAsync(cnt.data(), b->p_cnt, 4 * cnt.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(pfl.data(), b->p_flags, 4 * pfl.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));   // (also: the host tables above may go)
  for (int hh = 0; hh < nh_all; ++hh)
    if (pfl[(size_t)hh] & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", vs[(size_t)half_vcf[(size_t)hh]]);
  // --- exact regions; one level-2 segment per (VCF, partition in use).  The same VCFs with the same counts as last time (a batch
  //     run again): every table below is still on the device
  const bool same_cnt = same_vs && b->last2_cnt == cnt && !b->last2_cnt.empty();
  if (!same_cnt) {
  std::vector<uint32_t> off((size_t)nh_all * (NC + 1));
  std::vector<uint32_t> half_off;   // [segment][4]: where the entries of the VCF's 2nd, 3rd, 4th half begin inside the segment's run
  std::vector<SortSeg> segs;
  std::vector<VcfDesc> fake;
  std::vector<SortSeg> vsegs((size_t)nv);
  std::vector<int32_t> bk_tile_seg, ktile_seg, ktile_local, vparts((size_t)nv);
  int64_t bk_ents = 0, nbt = 0, nkt = 0;
  b->last2_vs.clear();
  for (int i = 0; i < nv; ++i) {
    const VcfDesc& d = b->L.vcfs[(size_t)vs[(size_t)i]];
    uint32_t run = 0;
    const int seg0 = (int)segs.size();
    const int h0 = vcf_half0[(size_t)i], h1 = vcf_half0[(size_t)i + 1];
    for (int p = 0; p < P2_PARTS; ++p) {
      run = (run + 1u) & ~1u;
      const uint32_t start = run;
      uint32_t hb[4] = {0u, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      for (int hh = h0; hh < h1; ++hh) {   // the halves' regions of the partition one behind the other
        hb[hh - h0] = run - start;
        for (int k = 0; k < P2_SUBS; ++k) { off[(size_t)hh * (NC + 1) + (size_t)p * P2_SUBS + k] = run; run += cnt[(size_t)hh * NC + (size_t)p * P2_SUBS + k]; }
      }
      const int64_t np = (int64_t)run - start;
      if (np == 0) continue;
      half_off.insert(half_off.end(), hb, hb + 4);
      if (np > (int64_t)HB_BUCKETS * HB_MAX_RECORDS * 13 / 16) { if (bad) bad->push_back(vs[(size_t)i]); return QM_OK; }   // all 256 buckets of a partition are in use: more than 6 656 records per bucket on average will not fit 8 x 1 024 (not for this path: *taken stays false)
      SortSeg g;
      memset(&g, 0, sizeof g);
      g.src_off = d.off; g.koff = ps[(size_t)h0].ent_off + start; g.n = np;   // (the halves of a VCF share its level-1 region)
      g.main_vcf = vs[(size_t)i]; g.sub_vcf = i; g.main_tile0 = d.tile0;
      g.pad = DJ_MAX_SHIFT; g.nbk = HB_BUCKETS; g.key_base = (uint32_t)p << P2_SHIFT;
      int64_t want = np / (128 * HB_SUBS) * 3 / 2 + 16, cap2 = 16;
      while (cap2 < want && cap2 < HB_SUB_MAX) cap2 *= 2;
      g.bk_cap = (int32_t)cap2; g.bk_off = bk_ents; g.bk_tile0 = (int32_t)nbt;
      bk_ents += (int64_t)HB_BUCKETS * HB_SUBS * cap2;
      const int64_t t = (np + BK_TILE - 1) / BK_TILE;
      bk_tile_seg.insert(bk_tile_seg.end(), (size_t)t, (int32_t)segs.size());
      nbt += t;
      // the buckets of a PARTITION as the "spans" of a VCF for k_finalize: one workgroup per partition (one per VCF summed
      // 1 536 rows by itself); k_sort_copy_rows adds a VCF's partitions up when the rows go home
      VcfDesc f = VcfDesc();
      f.off = 0; f.n = np; f.truth = tids[(size_t)i]; f.tile0 = 0; f.ntiles = 0; f.span0 = (int32_t)segs.size() * HB_BUCKETS; f.nspans = HB_BUCKETS; f.pad = 0;
      fake.push_back(f);
      segs.push_back(g);
    }
    for (int hh = h0; hh < h1; ++hh) off[(size_t)hh * (NC + 1) + NC] = run;
    SortSeg& vg = vsegs[(size_t)i];
    memset(&vg, 0, sizeof vg);
    vg.src_off = d.off; vg.n = d.n; vg.main_vcf = vs[(size_t)i]; vg.sub_vcf = seg0; vg.main_tile0 = d.tile0;
    vparts[(size_t)i] = (int32_t)segs.size() - seg0;
    const size_t k0 = ktile_seg.size();
    ktile_seg.insert(ktile_seg.end(), (size_t)d.ntiles, (int32_t)i);
    ktile_local.resize(k0 + (size_t)d.ntiles);
    for (int t = 0; t < d.ntiles; ++t) ktile_local[k0 + (size_t)t] = t;
    nkt += d.ntiles;
  }
  const int nseg = (int)segs.size();
  if (nbt > INT32_MAX || nkt > INT32_MAX) return fail(QM_E_LIMIT, "bucket path: too many tiles");
  // --- the arrays of the one-level path, sized for the level-2 segments (its cached tables are gone after this)
  b->last_segs.clear(); b->bk_tiles_valid = false; b->bk_fake_valid = false; b->lastx_vs.clear();
  int64_t cap;
  if (rc == QM_OK) { cap = b->cap_segs; rc = regrow(&b->d_segs, &cap, (int64_t)nseg, &b->dev_bytes); b->cap_segs = (int)cap; }
  if (rc == QM_OK) rc = regrow(&b->d_vsegs, &b->cap_vsegs, (int64_t)nv, &b->dev_bytes);
  if (rc == QM_OK) {
    cap = b->cap_ktiles; rc = regrow(&b->d_ktile_seg, &cap, nkt, &b->dev_bytes);
    if (rc == QM_OK) { cap = b->cap_ktiles; rc = regrow(&b->d_ktile_local, &cap, nkt, &b->dev_bytes); }
    if (rc == QM_OK) b->cap_ktiles = std::max(b->cap_ktiles, (int)nkt);
  }
  {
    const int64_t rows = (int64_t)nseg * HB_BUCKETS;
    int64_t c1 = b->cap_bk_rows * SPAN_HIST_WORDS, c2 = b->cap_bk_rows * 8;
    if (rc == QM_OK) rc = regrow(&b->bk_hist, &c1, rows * SPAN_HIST_WORDS, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_scal, &c2, rows * 8, &b->dev_bytes);
    if (rc == QM_OK) b->cap_bk_rows = std::max(b->cap_bk_rows, rows);
    if (rc == QM_OK) rc = regrow(&b->d_bk_vcfs, &b->cap_bk_vcfs, (int64_t)nseg, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_ent, &b->cap_bk_ent, bk_ents, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_rows, &b->cap_bk_rowdesc, rows, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_cursor, &b->cap_bk_cursor, rows * HB_SUBS + nseg + 32 + 16 * 65 + (int64_t)nseg * (SEG_HIST_WORDS + 1), &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->d_bk_tile_seg, &b->cap_bk_tiles, nbt, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->d_vparts, &b->cap_vparts, (int64_t)nv, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->p_half, &b->cap_p_half, (int64_t)nseg * 4, &b->dev_bytes);
    if (rc == QM_OK) rc = ensure_bucket_rows(b, nseg);
  }
  if (rc != QM_OK) return rc;
  b->last2_halves = any_halves;
  HIPCHK(hipMemcpyAsync(b->p_half, half_off.data(), 4 * half_off.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->p_off, off.data(), 4 * off.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_segs, segs.data(), sizeof(SortSeg) * segs.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_vsegs, vsegs.data(), sizeof(SortSeg) * vsegs.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_bk_tile_seg, bk_tile_seg.data(), 4 * bk_tile_seg.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_ktile_seg, ktile_seg.data(), 4 * ktile_seg.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_ktile_local, ktile_local.data(), 4 * ktile_local.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_bk_vcfs, fake.data(), sizeof(VcfDesc) * fake.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_vparts, vparts.data(), 4 * vparts.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // the host tables die with this block
  b->last2_vs = vs; b->last2_cnt = cnt; b->last2_nseg = nseg; b->last2_nbt = nbt; b->last2_nkt = nkt;
  b->last2_seg_vcf.resize(segs.size());
  for (size_t i = 0; i < segs.size(); ++i) b->last2_seg_vcf[i] = segs[i].main_vcf;
  }   // !same_cnt
  const int nseg = b->last2_nseg;
  const int64_t nbt = b->last2_nbt, nkt = b->last2_nkt;
  const size_t nhist0 = (size_t)nseg * HB_BUCKETS * HB_SUBS + (size_t)nseg + 32 + 16 * 65;   // the scatter's per-segment histograms lie behind the cursors, flags and phase clocks
  const size_t ncur = (nhist0 + (size_t)nseg * (SEG_HIST_WORDS + 1)) * 4;   // (+ 1: seg_maxd behind the histograms)
  HIPCHK(hipMemsetAsync(b->bk_cursor, 0, ncur, st));
  // --- level 1 scatter, then the one-level path over the partitions: rows, scatter from the level-1 entries, join, rows summed per VCF
  launch_part_scatter(PP, (int)nt1, st);
  BucketScatterParams S;
  S.segs = b->d_segs; S.tile_seg = b->d_bk_tile_seg; S.pos = b->pos; S.ref = b->ref; S.alt = b->alt; S.qual = b->qual; S.flags = b->flags;
  S.cursor = b->bk_cursor; S.ent = b->bk_ent; S.mask_pass = reinterpret_cast<uint32_t*>(b->mask_pass); S.mask_tp = reinterpret_cast<uint32_t*>(b->mask_tp);
  S.n_seg = nseg; S.n_bins = b->n_bins; S.tile_base = 0; S.l1_ent = b->p_ent; S.xent = nullptr; S.xcursor = nullptr; S.ext = 0; S.pairs = 0;
  uint32_t* const seg_hist = b->bk_cursor + nhist0;
  // (no look at the highest bucket here: every partition but a VCF's last fills its 256 buckets, and the look costs the scatter more than
  // the few empty workgroups cost the join -- 2.81 against 2.71 ms per 16 x 10 M records)
  uint32_t* const seg_maxd = nullptr;   // (the look at the highest filled bucket was measured on this path and lost: 2.08 against 2.02 ms per 64 x 2 M)
  S.seg_hist = seg_hist; S.seg_maxd = seg_maxd; S.l1_half = b->last2_halves ? b->p_half : nullptr;
  HashParams H;
  H.segs = b->d_segs; H.rows = b->bk_rows; H.rows_out = b->bk_rows; H.ent = b->bk_ent; H.cursor = b->bk_cursor; H.truths = b->ctx->d_truths; H.vcfs = b->d_vcfs;
  H.mask_tp = b->mask_tp; H.row_hist = b->bk_hist; H.row_scal = b->bk_scal; H.n_seg = nseg; H.n_bins = b->n_bins; H.seg_base = 0;
  H.xrows = nullptr; H.xent = nullptr; H.xcursor = nullptr; H.out_stride = HB_BUCKETS; H.ext = 0; H.scatter_hist = seg_hist ? 1 : 0; H.seg_maxd = seg_maxd;
  launch_bucket_rows(H, nseg, st);
  launch_bucket_scatter(S, (int)nbt, st);
  launch_join_lean(H, nseg, DJ_MAX_SHIFT, HB_BUCKETS, st);
  {
    FinalizeParams F = bucket_rows_finalize(b, seg_hist);
    F.row_cap = seg_maxd;
    launch_finalize(F, nseg, st);
  }
  HIPCHK(hipGetLastError());
  std::vector<uint32_t> hfl((size_t)nseg);
  HIPCHK(hipMemcpyAsync(hfl.data(), b->bk_vflags, 4 * hfl.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  b->path_stats[QM_PATH_BUCKET_CHUNKS] += 1;
  bool overflow = false;
  for (int i = 0; i < nseg; ++i)
    if (hfl[(size_t)i] & SPANF_OVERFLOW) {   // a bucket did not fit: nothing of the chunk is handed over
      overflow = true;
      if (bad && (size_t)i < b->last2_seg_vcf.size() && (bad->empty() || bad->back() != b->last2_seg_vcf[(size_t)i])) bad->push_back(b->last2_seg_vcf[(size_t)i]);
    }
  if (overflow) { b->path_stats[QM_PATH_OVERFLOW_CHUNKS] += 1; b->last2_vs.clear(); return QM_OK; }
  launch_sort_copy_rows(b->d_vsegs, nv, b->bk_roc, b->bk_rscal, b->roc, b->scalars, b->n_bins, st, global, b->d_vcfs, b->d_vparts);
  launch_tile_counts(b->d_vsegs, b->d_ktile_seg, b->d_ktile_local, (int)nkt, b->mask_pass, b->mask_tp, b->tile_tp, b->tile_fp, st);
  HIPCHK(hipGetLastError());
  b->path_stats[QM_PATH_DIRECT2] += nv;
  *taken = true;
  return QM_OK;
}

// ---- VCFs too large or too wide for 256 buckets (allele-extended ones; default-mode ones on references of 8.4 ... 16.8 M positions):
//      partitions of the key space, read from the columns ----
// The one-level path needs <= 8 192 records and <= 2^19 keys per bucket: configs[4]'s VCFs (2 M records on a 10 Mb genome) have
// neither.  The two-level path's first scatter packs level-1 entries that have no room for allele codes, so for these batches
// every PARTITION of 2^27 keys (256 buckets of 2^19) is a segment of the one-level scatter that reads ALL of its VCF's columns
// and keeps the records of its own key range (SortSeg.part).  Two neighbouring partitions share one pass (the 512-digit
// instantiation of the scatter: the first one's tiles fill both), so a 10 Mb genome costs ONE read of the columns, four partitions
// two; nothing else is new -- bucket rows, the two joins, the rows per segment and their sum per VCF are the one-level and
// two-level paths'.
constexpr int PX_MAX_PARTS = 4;
static int ext_parts_of(uint32_t posor, int pshift = P2_SHIFT) { return (int)((((uint64_t)posor << 4) | 15u) >> pshift) + 1; }
// WIDE buckets (round 6): 2^17 positions and up to 32 768 records per bucket, 256 of them per partition of 2^29 keys.  A shuffled
// default-mode VCF of 1.3 ... 67 M records on a reference of up to 67 M positions -- configs[3]'s 10 M records on 50 Mb -- is then
// TWO partitions = ONE pass of the 512-digit scatter over its columns (pieces of 64 bytes and more: whole sectors) and
// k_join_lean<.., BIG> per bucket, where the two-level path moves every record twice.
constexpr int PW_SHIFT = DJ_BIG_SHIFT + 8;   // 29
static bool join_hash_forced();
static bool bucketw_takes(const qm_batch* b, int v, int64_t n, uint32_t posor) {
  if (b->ext || g_penv.radix_only || join_hash_forced() || g_penv.bucketx == 0) return false;
  if (n > ((int64_t)1 << 26) || ext_parts_of(posor, PW_SHIFT) > 2) return false;    // (26 index bits of an entry; wider references: two levels)
  if (g_penv.bucketx == 3) return n >= g_penv.bucket_min;                          // 3: every default-mode unsorted VCF that fits (tests, fuzz)
  if (n < HB_MIN_RECORDS || ext_parts_of(posor) < 3) return false;                 // narrower references have cheaper paths
  const int64_t kor = ((int64_t)posor << 4) | 15;
  const int64_t buckets = (kor >> DJ_BIG_SHIFT) + 1;
  const int64_t truth = b->ctx->truths[(size_t)b->L.vcfs[(size_t)v].truth].n;
  // not fuller than eight sub-regions of 4 096 take; a bucket stages 4 096 truth keys (three quarters of that on average: the rest
  // is room for an uneven truth set)
  if (n > buckets * (4 * HB_MAX_RECORDS * 13 / 16) || truth > buckets * (4 * DJ_TRUTH_MAX * 3 / 4)) return false;
  if (!bucket_path_takes(b, n)) return true;   // (above the one-level path's 1.31 M records.  Measured on 50 Mb, 1.6e8 records per step: 2 M-record VCFs 4.3e10 /s against the two levels' 2.8e10, 5 M 6.7e10 against 3.6e10)
  // Up to 1.31 M records the one-level path takes a VCF of such a reference with the hashed join (7.5e10 /s where it fits, the
  // wide buckets 3.9e10: their workgroups have a fixed cost and few records each) -- but a one-level bucket is 2^18+ positions
  // here and the hashed join stages 1 024 truth keys: a denser truth set flags the chunk and the radix sort redoes it
  // (1.3-1.7e10 /s; wide buckets: 2.7-3.4e10).
  int msb = 31;
  while (msb > 0 && !((kor >> msb) & 1)) --msb;
  const int64_t one_level_buckets = (kor >> std::max(4, msb - 7)) + 1;
  return truth > one_level_buckets * (HB_TRUTH_SLOTS / 2 * 5 / 8);
}
static bool bucketx_takes(const qm_batch* b, int64_t n, uint32_t posor) {
  if (b->ext && g_penv.bucket_ext == 0) return false;
  if (g_penv.radix_only) return false;
  if (join_hash_forced()) return false;
  if (n < HB_MIN_RECORDS || n > ((int64_t)1 << (b->ext ? HB_INDEX_BITS : 26))) return false;   // (the second stream's entries hold 21 index bits, the first stream's 26)
  const int parts = ext_parts_of(posor);
  if (parts > PX_MAX_PARTS) return false;
  if (g_penv.bucketx == 0) return false;
  if (g_penv.bucketx == 2) return true;   // 2: every unsorted VCF that fits (tests, fuzz)
  if (b->ext) return parts > 1 || !bucket_path_takes(b, n);
  // default mode: a reference of 8.4 ... 16.8 M positions is ONE pair of partitions = one pass of the 512-digit scatter and the
  // bit-map join, where the one-level path would need the hashed join (bucket key ranges of 2^20) and larger VCFs two levels
  // (more partitions per pass were tried in round 6 -- eight, a 2 048-digit scatter -- and lost to the two levels by 2.6x: the
  // pieces a tile leaves per bucket fall below an L2 line and are evicted half written, profiles/r06_pmc_scatter2048_not_kept.json)
  return parts == 2;
}

// bad (when the chunk is not taken): the VCFs whose buckets overflowed -- the others would have fitted among themselves
static int bucketx_chunk(qm_batch* b, const std::vector<int>& vs, hipStream_t st, uint64_t* global, const std::vector<uint32_t>& posor, bool* taken, bool wide = false,
                         std::vector<int>* bad = nullptr) {
  *taken = false;
  if (bad) bad->clear();
  const int nv = (int)vs.size();
  const int pshift = wide ? PW_SHIFT : P2_SHIFT, bshift = wide ? DJ_BIG_SHIFT : DJ_MAX_SHIFT;
  const int64_t sub_max = wide ? 4 * HB_SUB_MAX : HB_SUB_MAX;
  std::vector<uint32_t> por((size_t)nv);
  for (int i = 0; i < nv; ++i) por[(size_t)i] = posor[(size_t)vs[(size_t)i]];
  const bool same = !b->lastx_vs.empty() && b->lastx_vs == vs && b->lastx_por == por && b->lastx_wide == wide;   // the tables of this chunk are still on the device
  int nseg = b->lastx_nseg;
  int64_t nbt = b->lastx_nbt, nkt = b->lastx_nkt;
  const bool xs = b->ext;                                  // two entry streams (allele-extended batches)
  const int out_stride = xs ? 2 * HB_BUCKETS : HB_BUCKETS;
  constexpr int G = 2;                                     // partitions per pass over the columns
  if (!same) {
  nbt = 0; nkt = 0;
  std::vector<SortSeg> segs, vsegs((size_t)nv);
  std::vector<VcfDesc> fake;
  std::vector<int32_t> bk_tile_seg, ktile_seg, ktile_local, vparts((size_t)nv);
  int64_t bk_ents = 0;
  for (int i = 0; i < nv; ++i) {
    const VcfDesc& d = b->L.vcfs[(size_t)vs[(size_t)i]];
    const uint32_t kor = (posor[(size_t)vs[(size_t)i]] << 4) | 15u;
    const int parts = ext_parts_of(posor[(size_t)vs[(size_t)i]], pshift);
    const int seg0 = (int)segs.size();
    for (int p = 0; p < parts; ++p) {
      SortSeg g;
      memset(&g, 0, sizeof g);
      g.src_off = d.off; g.n = d.n; g.main_vcf = vs[(size_t)i]; g.sub_vcf = i; g.main_tile0 = d.tile0;
      g.pad = bshift; g.key_base = (uint32_t)p << pshift;
      // two neighbouring partitions share ONE pass over the columns: the tiles belong to the first of the pair, whose digits (512
      // of them) reach into the second's cursors, regions and rows (same capacity, laid out one behind the other)
      const int gfirst = p / G * G, gsize = std::min(G, parts - gfirst);
      const bool lead = p == gfirst && gsize > 1, follow = p != gfirst;
      const bool pair_last = gfirst + gsize == parts;
      g.part = (pair_last ? 2 : 1) | (lead ? 4 | (gsize << 4) : 0);
      g.nbk = p + 1 == parts ? std::min<int>(HB_BUCKETS, (int)((kor - g.key_base) >> bshift) + 1) : HB_BUCKETS;
      int64_t want = d.n / (128 * HB_SUBS) * 3 / 2 + 16, cap2 = 16;   // (how the records spread over the partitions is not known: room as for all of them)
      while (cap2 < want && cap2 < sub_max) cap2 *= 2;
      g.bk_cap = (int32_t)cap2; g.bk_off = bk_ents; g.bk_tile0 = (int32_t)nbt;
      bk_ents += (int64_t)HB_BUCKETS * HB_SUBS * cap2;
      const int64_t t = follow ? 0 : (d.n + BK_TILE - 1) / BK_TILE;   // (the second of a pair has no tiles of its own)
      bk_tile_seg.insert(bk_tile_seg.end(), (size_t)t, (int32_t)segs.size());
      nbt += t;
      VcfDesc f = VcfDesc();   // the two streams' rows of a partition as the "spans" of a VCF for k_finalize
      f.off = 0; f.n = d.n; f.truth = d.truth; f.tile0 = 0; f.ntiles = 0; f.span0 = (int32_t)segs.size() * out_stride; f.nspans = out_stride; f.pad = 0;
      fake.push_back(f);
      segs.push_back(g);
    }
    SortSeg& vg = vsegs[(size_t)i];
    memset(&vg, 0, sizeof vg);
    vg.src_off = d.off; vg.n = d.n; vg.main_vcf = vs[(size_t)i]; vg.sub_vcf = seg0; vg.main_tile0 = d.tile0;
    vparts[(size_t)i] = parts;
    const size_t k0 = ktile_seg.size();
    ktile_seg.insert(ktile_seg.end(), (size_t)d.ntiles, (int32_t)i);
    ktile_local.resize(k0 + (size_t)d.ntiles);
    for (int t = 0; t < d.ntiles; ++t) ktile_local[k0 + (size_t)t] = t;
    nkt += d.ntiles;
  }
  nseg = (int)segs.size();
  if (nbt > INT32_MAX || nkt > INT32_MAX) return fail(QM_E_LIMIT, "bucket path: too many tiles");
  // the arrays of the one-level path, sized for these segments (its cached tables, and the two-level path's, are gone after this)
  b->last_segs.clear(); b->bk_tiles_valid = false; b->bk_fake_valid = false; b->last2_vs.clear(); b->lastx_vs.clear();
  int rc = QM_OK;
  int64_t cap;
  { cap = b->cap_segs; rc = regrow(&b->d_segs, &cap, (int64_t)nseg, &b->dev_bytes); b->cap_segs = (int)cap; }
  if (rc == QM_OK) rc = regrow(&b->d_vsegs, &b->cap_vsegs, (int64_t)nv, &b->dev_bytes);
  if (rc == QM_OK) {
    cap = b->cap_ktiles; rc = regrow(&b->d_ktile_seg, &cap, nkt, &b->dev_bytes);
    if (rc == QM_OK) { cap = b->cap_ktiles; rc = regrow(&b->d_ktile_local, &cap, nkt, &b->dev_bytes); }
    if (rc == QM_OK) b->cap_ktiles = std::max(b->cap_ktiles, (int)nkt);
  }
  {
    const int64_t rows = (int64_t)nseg * HB_BUCKETS, orows = (int64_t)nseg * out_stride;
    int64_t c1 = b->cap_bk_rows * SPAN_HIST_WORDS, c2 = b->cap_bk_rows * 8;
    if (rc == QM_OK) rc = regrow(&b->bk_hist, &c1, orows * SPAN_HIST_WORDS, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_scal, &c2, orows * 8, &b->dev_bytes);
    if (rc == QM_OK) b->cap_bk_rows = std::max(b->cap_bk_rows, orows);
    if (rc == QM_OK && xs) rc = regrow(&b->bk_xent, &b->cap_bk_xent, 2 * bk_ents, &b->dev_bytes);
    if (rc == QM_OK && xs) rc = regrow(&b->bk_xcursor, &b->cap_bk_xcursor, rows * HB_SUBS, &b->dev_bytes);
    if (rc == QM_OK && xs) rc = regrow(&b->bk_xrows, &b->cap_bk_xrows, rows, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->d_bk_vcfs, &b->cap_bk_vcfs, (int64_t)nseg, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_ent, &b->cap_bk_ent, bk_ents, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_rows, &b->cap_bk_rowdesc, rows, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->bk_cursor, &b->cap_bk_cursor, rows * HB_SUBS + nseg + 32 + 16 * 65 + (int64_t)nseg * (SEG_HIST_WORDS + 1), &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->d_bk_tile_seg, &b->cap_bk_tiles, nbt, &b->dev_bytes);
    if (rc == QM_OK) rc = regrow(&b->d_vparts, &b->cap_vparts, (int64_t)nv, &b->dev_bytes);
    if (rc == QM_OK) rc = ensure_bucket_rows(b, nseg);
  }
  if (rc != QM_OK) return rc;
  HIPCHK(hipMemcpyAsync(b->d_segs, segs.data(), sizeof(SortSeg) * segs.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_vsegs, vsegs.data(), sizeof(SortSeg) * vsegs.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_bk_tile_seg, bk_tile_seg.data(), 4 * bk_tile_seg.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_ktile_seg, ktile_seg.data(), 4 * ktile_seg.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_ktile_local, ktile_local.data(), 4 * ktile_local.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_bk_vcfs, fake.data(), sizeof(VcfDesc) * fake.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(b->d_vparts, vparts.data(), 4 * vparts.size(), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // the host tables die with this block
  b->lastx_vs = vs; b->lastx_por = por; b->lastx_wide = wide; b->lastx_nseg = nseg; b->lastx_nbt = nbt; b->lastx_nkt = nkt;
  b->lastx_seg_vcf.resize((size_t)nseg);
  for (int i = 0; i < nseg; ++i) b->lastx_seg_vcf[(size_t)i] = segs[(size_t)i].main_vcf;
  }   // !same
  const size_t nhist0 = (size_t)nseg * HB_BUCKETS * HB_SUBS + (size_t)nseg + 32 + 16 * 65;   // the scatter's per-segment histograms lie behind the cursors, flags and phase clocks
  const size_t ncur = (nhist0 + (size_t)nseg * (SEG_HIST_WORDS + 1)) * 4;   // (+ 1: seg_maxd behind the histograms)
  if (xs) HIPCHK(hipMemsetAsync(b->bk_xcursor, 0, (size_t)nseg * HB_BUCKETS * HB_SUBS * 4, st));
  BucketScatterParams S;
  S.segs = b->d_segs; S.tile_seg = b->d_bk_tile_seg; S.pos = b->pos; S.ref = b->ref; S.alt = b->alt; S.qual = b->qual; S.flags = b->flags;
  S.cursor = b->bk_cursor; S.ent = b->bk_ent; S.mask_pass = reinterpret_cast<uint32_t*>(b->mask_pass); S.mask_tp = reinterpret_cast<uint32_t*>(b->mask_tp);
  S.n_seg = nseg; S.n_bins = b->n_bins; S.tile_base = 0; S.l1_ent = nullptr; S.xent = xs ? b->bk_xent : nullptr; S.xcursor = xs ? b->bk_xcursor : nullptr; S.ext = xs ? 1 : 0;
  S.pairs = 1;   // (tiles of single partitions run through the 512-digit instantiation as well)
  uint32_t* const seg_hist = b->bk_cursor + nhist0;
  uint32_t* const seg_maxd = nullptr;   // (the look at the highest filled bucket was measured on this path and lost: 2.08 against 2.02 ms per 64 x 2 M)
  S.seg_hist = seg_hist; S.seg_maxd = seg_maxd; S.l1_half = nullptr;
  HashParams H;
  H.segs = b->d_segs; H.rows = b->bk_rows; H.rows_out = b->bk_rows; H.ent = b->bk_ent; H.cursor = b->bk_cursor; H.truths = b->ctx->d_truths; H.vcfs = b->d_vcfs;
  H.mask_tp = b->mask_tp; H.row_hist = b->bk_hist; H.row_scal = b->bk_scal; H.n_seg = nseg; H.n_bins = b->n_bins; H.seg_base = 0;
  H.xrows = xs ? b->bk_xrows : nullptr; H.xent = b->bk_xent; H.xcursor = b->bk_xcursor; H.out_stride = out_stride; H.ext = xs ? 1 : 0; H.scatter_hist = seg_hist ? 1 : 0; H.seg_maxd = seg_maxd;
  H.zero = b->bk_cursor; H.n_zero = (uint32_t)(ncur / 4);   // the scatter's cursors, flags and counts: cleared by the kernel that writes the rows (one dispatch instead of a memset's two)
  launch_bucket_rows(H, nseg, st);
  H.zero = nullptr; H.n_zero = 0u;
  launch_bucket_scatter(S, (int)nbt, st);
  if (wide) launch_join_big(H, nseg, HB_BUCKETS, st);
  else launch_join_lean(H, nseg, DJ_MAX_SHIFT, HB_BUCKETS, st);
  if (xs) launch_join_ext(H, nseg, HB_BUCKETS, st);
  {
    FinalizeParams F = bucket_rows_finalize(b, seg_hist);
    F.row_cap = seg_maxd;
    launch_finalize(F, nseg, st);
  }
  HIPCHK(hipGetLastError());
  std::vector<uint32_t> hfl((size_t)nseg);
  HIPCHK(hipMemcpyAsync(hfl.data(), b->bk_vflags, 4 * hfl.size(), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  b->path_stats[QM_PATH_BUCKET_CHUNKS] += 1;
  bool overflow = false;
  for (int i = 0; i < nseg; ++i) {
    if (hfl[(size_t)i] & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", b->lastx_seg_vcf[(size_t)i]);
    if (hfl[(size_t)i] & SPANF_OVERFLOW) {   // a bucket did not fit: nothing of the chunk is handed over
      overflow = true;
      if (bad && (bad->empty() || bad->back() != b->lastx_seg_vcf[(size_t)i])) bad->push_back(b->lastx_seg_vcf[(size_t)i]);   // (a VCF's segments lie one behind the other)
    }
  }
  if (overflow) { b->path_stats[QM_PATH_OVERFLOW_CHUNKS] += 1; b->lastx_vs.clear(); return QM_OK; }
  launch_sort_copy_rows(b->d_vsegs, nv, b->bk_roc, b->bk_rscal, b->roc, b->scalars, b->n_bins, st, global, b->d_vcfs, b->d_vparts);
  launch_tile_counts(b->d_vsegs, b->d_ktile_seg, b->d_ktile_local, (int)nkt, b->mask_pass, b->mask_tp, b->tile_tp, b->tile_fp, st);
  HIPCHK(hipGetLastError());
  b->path_stats[QM_PATH_PARTITIONS] += nv;
  *taken = true;
  return QM_OK;
}

// re-derive tile offsets + compaction for every VCF (cheap: masks only)
static int rescan_and_compact(qm_batch* b, hipStream_t st);

// The VCFs found out of order, redone: by size on the two-level bucket path (kind 2), the bucket path (1) or the radix sort (0),
// in chunks of <= 2^28 records.  posor[v]: the position bits the optimistic pass saw in VCF v.
static int redo_unsorted(qm_batch* b, const std::vector<int>& todo, const std::vector<uint32_t>& posor, hipStream_t st) {
  // the VCFs the bucket path takes (by size: a VCF costs it 256 workgroups and 256 rows whatever it holds, and its
  // buckets hold 8 192 records at most) in chunks of their own, the others on the radix sort
  // kind 3: allele-extended VCFs in partitions read from the columns (bucketx_chunk)
  refresh_path_env();
  std::vector<int> part[5];   // 4: partitions of WIDE buckets (bucketw_takes)
  for (int v : todo) {
    const int64_t n = b->L.vcfs[(size_t)v].n;
    const bool force2 = g_penv.bucket2 == 2 && bucket2_takes(b, n);
    const bool wide = bucketw_takes(b, v, n, posor[(size_t)v]);
    // A default-mode VCF with more records than the narrow buckets of its reference hold (8 192 per 2^15 positions, the
    // sub-regions 13/16 full on average: 0.2 records per position) overflows the partitions and the two levels alike, and a
    // chunk falls back as a whole: such a VCF goes to the radix sort alone instead of taking its neighbours with it.  (The
    // position bits are an upper bound of up to twice the highest position: this errs towards trying the buckets.)
    const int64_t narrow_buckets = ((((int64_t)posor[(size_t)v] << 4) | 15) >> DJ_MAX_SHIFT) + 1;
    const bool dense = !b->ext && g_penv.bucketx != 2 && g_penv.bucket2 != 2 && n > narrow_buckets * (HB_MAX_RECORDS * 13 / 16);
    part[wide && g_penv.bucketx == 3 ? 4 : !dense && bucketx_takes(b, n, posor[(size_t)v]) ? 3 : force2 ? 2 : wide ? 4 : bucket_path_takes(b, n) ? 1
         : !dense && bucket2_takes(b, n) ? 2 : 0].push_back(v);
  }
  const int64_t chunk_records = sort_chunk_records();
  // the two levels for a chunk; the VCFs of it that do not fit (a partition too dense, a bucket that overflowed) go through the
  // radix sort, the others through the two levels again among themselves
  auto two_levels_or_sort = [&](const std::vector<int>& chunk) -> int {
    std::vector<int> pending = chunk, bad;   // pending: the VCFs without a result so far
    for (int round = 0; round < 3; ++round) {   // (a partition too dense names one VCF at a time: three tries, then everything left is sorted)
      std::vector<int> good, named;
      for (int v : pending) if (std::find(bad.begin(), bad.end(), v) == bad.end()) good.push_back(v);
      if (good.empty()) break;
      bool taken = false;
      const int rc = bucket2_chunk(b, good, st, b->last_global, &taken, &named);
      if (rc != QM_OK) return rc;
      if (taken) { pending = bad; break; }
      if (named.empty() || named.size() >= good.size()) break;
      bad.insert(bad.end(), named.begin(), named.end());
    }
    if (pending.empty()) return QM_OK;
    b->path_stats[QM_PATH_RADIX_AFTER_OVERFLOW] += (int64_t)pending.size();
    const int rc = sort_chunk(b, pending, st, b->last_global, posor, false);
    b->path_stats[QM_PATH_RADIX] -= (int64_t)pending.size();
    return rc;
  };
  for (int kind = 4; kind >= 0; --kind) {
    std::vector<int> chunk;
    int64_t chunk_n = 0;
    for (size_t i = 0; i <= part[kind].size(); ++i) {
      // (kind 3: every PAIR of partitions reads its whole VCF)
      const int64_t wgt = i < part[kind].size() ? b->L.vcfs[(size_t)part[kind][i]].n * (kind == 3 ? (ext_parts_of(posor[(size_t)part[kind][i]]) + 1) / 2 : 1) : 0;
      const bool flush = i == part[kind].size() || (!chunk.empty() && chunk_n + wgt > chunk_records) ||
                         (kind >= 1 && chunk.size() >= (size_t)(kind == 3 ? 4096 / PX_MAX_PARTS : kind == 4 ? 64 : 4096));   // (kind 4: 2 x 67 MB of bucket regions per VCF)   // 256 rows of 1.5 KB and >= 1 MB of bucket regions per VCF: bounded per chunk
      if (flush && !chunk.empty()) {
        int rc = QM_OK;
        bool taken = false;
        if (kind == 4 || kind == 3) {
          // a chunk one of whose VCFs overflowed is not handed over: the VCFs that fitted take the same buckets again, among
          // themselves; the others go on to the next path (wide buckets: two levels, then the radix sort)
          std::vector<int> bad, rest = chunk;
          rc = bucketx_chunk(b, chunk, st, b->last_global, posor, &taken, kind == 4, &bad);
          if (rc == QM_OK && !taken && !bad.empty() && bad.size() < chunk.size()) {
            std::vector<int> good;
            for (int v : chunk) if (std::find(bad.begin(), bad.end(), v) == bad.end()) good.push_back(v);
            bool taken_good = false;
            rc = bucketx_chunk(b, good, st, b->last_global, posor, &taken_good, kind == 4);
            if (taken_good) rest = bad;
          }
          if (rc == QM_OK && !taken) {
            if (kind == 4) rc = two_levels_or_sort(rest);
            else { b->path_stats[QM_PATH_RADIX_AFTER_OVERFLOW] += (int64_t)rest.size(); rc = sort_chunk(b, rest, st, b->last_global, posor, false); b->path_stats[QM_PATH_RADIX] -= (int64_t)rest.size(); }
          }
        } else if (kind == 2) {
          rc = two_levels_or_sort(chunk);
        } else {
          rc = sort_chunk(b, chunk, st, b->last_global, posor, kind == 1);
        }
        if (rc != QM_OK) return rc;
        chunk.clear();
        chunk_n = 0;
      }
      if (i < part[kind].size()) { chunk.push_back(part[kind][i]); chunk_n += wgt; }
    }
  }
  return QM_OK;
}

// The bucket chunks queued without a look at their flags (sort_chunk: speculate), behind the wait that ended the step: the mirrors
// of their rows' k_finalize say whether a chunk fitted.  One that did not was left alone by k_sort_copy_rows (nothing of it joined
// the per-truth sums) and goes through the radix sort now, followed by another rescan.
static int settle_pending(qm_batch* b, const std::vector<uint32_t>& posor, hipStream_t st) {
  std::vector<qm_batch::Pending> pend;
  pend.swap(b->pend);
  b->pend_segs = 0;
  // every chunk's mirror words first: a re-run below writes its own over them
  std::vector<std::vector<uint32_t>> flags(pend.size());
  for (size_t k = 0; k < pend.size(); ++k) {
    const qm_batch::Pending& p = pend[k];
    const int nseg = (int)p.vs.size();
    flags[k].resize((size_t)nseg);
    for (int i = 0; i < nseg; ++i) {
      uint32_t fl = b->h_summary[16 + 2 * (size_t)b->n_vcf + (size_t)p.off + (size_t)i];
      if (p.tight) {
        const uint32_t md = b->h_summary[16 + 3 * (size_t)b->n_vcf + (size_t)p.off + (size_t)i];
        if (md > (uint32_t)p.nbk_launch) fl |= SPANF_OVERFLOW;   // (a remembered bound that no longer holds: cannot happen while the columns stay the same)
        else if (memo_on()) {
          if (b->known_nbk.empty()) b->known_nbk.assign((size_t)b->n_vcf, 0u);
          b->known_nbk[(size_t)p.vs[(size_t)i]] = std::max(md, 1u);
        }
      }
      if (fl & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", p.vs[(size_t)i]);
      flags[k][(size_t)i] = fl;
    }
  }
  bool again = false;
  for (size_t k = 0; k < pend.size(); ++k) {
    const qm_batch::Pending& p = pend[k];
    const int nseg = (int)p.vs.size();
    std::vector<int> bad, good;
    for (int i = 0; i < nseg; ++i) ((flags[k][(size_t)i] & SPANF_OVERFLOW) ? bad : good).push_back(p.vs[(size_t)i]);
    if (bad.empty()) { b->path_stats[p.direct ? QM_PATH_DIRECT : QM_PATH_HASHED] += nseg; continue; }
    // a bucket of a VCF did not fit its tables (dense positions, a dense truth set): nothing of the chunk was handed over
    // (k_sort_copy_rows saw its "bad" word).  The radix sort redoes that VCF from the columns, the others take the buckets again.
    if (p.tight && !b->known_nbk.empty()) for (int v : p.vs) b->known_nbk[(size_t)v] = 0u;
    // (what the radix sort learns about its chunk's positions corrects the estimate such a VCF is remembered with: posor_seen)
    b->path_stats[QM_PATH_OVERFLOW_CHUNKS] += 1;
    const int rc = redo_overflowed(b, bad, good, st, b->last_global, posor);
    if (rc != QM_OK) return rc;
    again = true;
  }
  return again ? rescan_and_compact(b, st) : QM_OK;
}

extern "C" int qm_batch_finish(qm_batch* b, void* stream) {
  if (!b || !b->ran) return fail(QM_E_STATE, "qm_batch_finish: nothing was run");
  qm_ctx* c = b->ctx;
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  if (b->finished) { HIPCHK(hipStreamSynchronize(st)); return QM_OK; }
  for (auto& x : b->path_stats) x = 0;
  b->pend.clear(); b->pend_segs = 0;
  // QM_FINISH_TRACE=1: the host's clock at the stations of this call, in us since it was entered (stderr)
  static const bool ftrace = getenv("QM_FINISH_TRACE") != nullptr;
  double ft[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now_us = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
  if (ftrace) ft[0] = now_us();
  std::vector<uint32_t> posor((size_t)b->n_vcf, 0u);
  bool redone = false;
  if (b->run_used_known) {
    // the VCFs known to be out of order: their bucket path goes onto the stream behind the run, before anything is waited for
    std::vector<int> kn;
    for (int v = 0; v < b->n_vcf; ++v) if (b->known[(size_t)v]) { kn.push_back(v); posor[(size_t)v] = b->known_posor[(size_t)v]; }
    b->path_stats[QM_PATH_UNSORTED] += (int64_t)kn.size();
    const int rc = redo_unsorted(b, kn, posor, st);
    if (rc != QM_OK) { (void)hipStreamSynchronize(st); b->pend.clear(); b->pend_segs = 0; forget_known(b, -1); return rc; }
    redone = !kn.empty();
  }
  // every VCF known to be out of order: the run launched no optimistic pass, nothing can have raised a flag, and the rescan below
  // goes onto the stream behind the bucket path without a round trip through the host in between
  const bool nothing_ran = b->run_used_known && b->n_known == b->n_vcf && b->h_summary != nullptr;
  // the flags (and their mirrors) are complete behind the k_finalize part that writes them: the wait is for that kernel, while the
  // run's compaction and row sums go on -- the bucket path of what is found out of order is built and queued beside them
  // (QM_FLAGS_WAIT=stream: wait for everything, as rounds 1-4 did)
  bool idle = nothing_ran;   // has the stream been waited for?
  if (!nothing_ran) {
    if (b->flags_recorded && b->h_summary && flags_event_on()) HIPCHK(hipEventSynchronize(b->ev_flags));
    else { HIPCHK(hipStreamSynchronize(st)); idle = true; }
  }
  if (ftrace) ft[1] = now_us();
  std::vector<int> todo;
  if (!nothing_ran) {
    std::vector<uint32_t> fl((size_t)b->n_vcf);
    if (b->h_summary) memcpy(fl.data(), b->h_summary + 16, 4 * fl.size());   // k_finalize's host-mapped mirror: complete behind the wait above
    else HIPCHK(hipMemcpy(fl.data(), b->vcf_flags, 4 * fl.size(), hipMemcpyDeviceToHost));
    for (int v = 0; v < b->n_vcf; ++v) {
      if (fl[(size_t)v] & (SPANF_BADPOS | SPANF_RUNLIMIT)) {
        (void)hipStreamSynchronize(st);   // nothing of this run is left in flight behind the error
        b->pend.clear(); b->pend_segs = 0;
        if (fl[(size_t)v] & SPANF_BADPOS) return fail(QM_E_RANGE, "VCF %d holds a position outside [0, 2^28)", v);
        return fail(QM_E_LIMIT, "VCF %d repeats one position more than %d times between equal alleles (allele-extended de-duplication limit)", v, 1 << 14);
      }
      if ((fl[(size_t)v] & SPANF_UNSORTED) && !(b->run_used_known && b->known[(size_t)v])) todo.push_back(v);
    }
    if (!todo.empty()) {
      std::vector<uint32_t> po((size_t)b->n_vcf);
      if (b->h_summary) memcpy(po.data(), b->h_summary + 16 + b->n_vcf, 4 * po.size());
      else HIPCHK(hipMemcpy(po.data(), b->vcf_posor, 4 * po.size(), hipMemcpyDeviceToHost));
      for (int v : todo) posor[(size_t)v] = po[(size_t)v];
      b->path_stats[QM_PATH_UNSORTED] += (int64_t)todo.size();
      if (ftrace) ft[2] = now_us();
      const int rc = redo_unsorted(b, todo, posor, st);
      if (rc != QM_OK) { (void)hipStreamSynchronize(st); b->pend.clear(); b->pend_segs = 0; return rc; }
      if (ftrace) ft[3] = now_us();
      redone = true;
    }
  }
  if (redone) {
    int rc = rescan_and_compact(b, st);
    if (rc == QM_OK && !b->pend.empty()) rc = settle_pending(b, posor, st);
    if (rc != QM_OK) { (void)hipStreamSynchronize(st); b->pend.clear(); b->pend_segs = 0; if (memo_on()) forget_known(b, -1); return rc; }
  } else if (nothing_ran || !idle) {
    HIPCHK(hipStreamSynchronize(st));
  }
  if (memo_on() && !todo.empty()) {
    if (b->known.empty()) { b->known.assign((size_t)b->n_vcf, (uint8_t)0); b->known_posor.assign((size_t)b->n_vcf, 0u); }
    for (int v : todo) if (!b->known[(size_t)v]) {
      b->known[(size_t)v] = 1; b->known_posor[(size_t)v] = posor[(size_t)v] | (b->posor_seen.empty() ? 0u : b->posor_seen[(size_t)v]);
      ++b->n_known; b->known_dirty = true;
    }
  }
  for (int k = 0; k < QM_N_PATH_STATS; ++k) c->path_total[k] += b->path_stats[k];
  b->finished = true;
  if (ftrace) {
    ft[4] = now_us();
    fprintf(stderr, "finish trace: flags waited for %.1f us, read %.1f, bucket path queued %.1f, everything done %.1f (the latest chunk: entered %.1f, tables %.1f, scatter queued %.1f)\n", ft[1] - ft[0],
            ft[2] ? ft[2] - ft[0] : 0.0, ft[3] ? ft[3] - ft[0] : 0.0, ft[4] - ft[0], g_ftrace[0] - ft[0], g_ftrace[1] - ft[0], g_ftrace[2] - ft[0]);
  }
  return QM_OK;
}

// k_finalize recomputes the offsets of every tile; to keep ROC rows and per-truth
// sums untouched it runs on a throw-away output set.
static int rescan_and_compact(qm_batch* b, hipStream_t st) {
  if (!b->rs_roc) {
    DALLOC(b->rs_roc, (size_t)b->n_vcf * 3 * (size_t)b->n_bins);
    DALLOC(b->rs_scal, (size_t)b->n_vcf * 8);
    DALLOC(b->rs_flags, (size_t)b->n_vcf);
    b->dev_bytes += (int64_t)b->n_vcf * (3 * b->n_bins * 8 + 64 + 4);
  }
  FinalizeParams F = finalize_params(b, nullptr);
  F.roc = b->rs_roc; F.scalars = b->rs_scal; F.vcf_flags = b->rs_flags; F.vcf_posor = nullptr;
  F.parts = 1;   // the tile offsets only: the rows it would sum are thrown away (611 span rows per 10 M-record VCF: 0.3 ms for sixteen of them)
  launch_finalize(F, b->n_vcf, st);
  CompactParams CP = compact_params(b);
  CP.skip_unsorted = 0;   // the flags still say 'unsorted' for the VCFs just redone: compact them too
  launch_compact(CP, (int)b->L.spans.size(), st);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(QM_E_HIP, "rescan/compact: %s", hipGetErrorString(e));
  return QM_OK;
}

// ---- getters ------------------------------------------------------------------
#define NEED_FINISHED(b, name) \
  if (!(b) || !(b)->finished) return fail(QM_E_STATE, name ": call qm_batch_run + qm_batch_finish first")

extern "C" int qm_batch_get_cls(qm_batch* b, int v, uint8_t* out) {
  NEED_FINISHED(b, "qm_batch_get_cls");
  if (v < 0 || v >= b->n_vcf || !out) return fail(QM_E_INVAL, "qm_batch_get_cls: bad arguments");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  if (d.n == 0) return QM_OK;
  launch_masks_to_cls(b->mask_pass, b->mask_tp, d.off, d.n, b->cls_scratch, b->ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, b->cls_scratch, (size_t)d.n, hipMemcpyDeviceToHost, b->ctx->stream));
  HIPCHK(hipStreamSynchronize(b->ctx->stream));
  return QM_OK;
}
// class masks of one VCF as they sit in HBM: bit r of word r / 64 = record r (1 bit per record and mask instead of the
// byte per record of qm_batch_get_cls).  (n + 63) / 64 words each.
extern "C" int qm_batch_get_masks(qm_batch* b, int v, uint64_t* kept, uint64_t* tp) {
  NEED_FINISHED(b, "qm_batch_get_masks");
  if (v < 0 || v >= b->n_vcf || !kept || !tp) return fail(QM_E_INVAL, "qm_batch_get_masks: bad arguments");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  const size_t nw = (size_t)((d.n + 63) / 64);
  if (nw) {
    HIPCHK(hipMemcpyAsync(kept, b->mask_pass + (d.off >> 6), nw * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipMemcpyAsync(tp, b->mask_tp + (d.off >> 6), nw * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
    if (d.n & 63) {   // bits beyond the VCF's last record are not defined on the device
      const uint64_t m = (1ull << (d.n & 63)) - 1ull;
      kept[nw - 1] &= m; tp[nw - 1] &= m;
    }
  }
  return QM_OK;
}
extern "C" int qm_batch_get_idx(qm_batch* b, int v, int32_t* out) {
  NEED_FINISHED(b, "qm_batch_get_idx");
  if (v < 0 || v >= b->n_vcf || !out) return fail(QM_E_INVAL, "qm_batch_get_idx: bad arguments");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  if (d.n) HIPCHK(hipMemcpy(out, b->idx + d.off, (size_t)d.n * 4, hipMemcpyDeviceToHost));
  return QM_OK;
}
extern "C" int qm_batch_get_roc(qm_batch* b, uint64_t* out) {
  NEED_FINISHED(b, "qm_batch_get_roc");
  HIPCHK(hipSetDevice(b->ctx->dev));
  HIPCHK(hipMemcpy(out, b->roc, (size_t)b->n_vcf * 3 * (size_t)b->n_bins * 8, hipMemcpyDeviceToHost));
  return QM_OK;
}
extern "C" int qm_batch_get_scalars(qm_batch* b, int64_t* out) {
  NEED_FINISHED(b, "qm_batch_get_scalars");
  HIPCHK(hipSetDevice(b->ctx->dev));
  HIPCHK(hipMemcpy(out, b->scalars, (size_t)b->n_vcf * 8 * 8, hipMemcpyDeviceToHost));
  return QM_OK;
}
extern "C" int qm_batch_get_global(qm_batch* b, uint64_t* out) {
  NEED_FINISHED(b, "qm_batch_get_global");
  HIPCHK(hipSetDevice(b->ctx->dev));
  HIPCHK(hipMemcpy(out, b->last_global, (size_t)b->n_truth * 3 * (size_t)b->n_bins * 8, hipMemcpyDeviceToHost));
  return QM_OK;
}
// dst[i] += src[i] for device arrays (qm_extract_files_ex adds the per-truth sums of its groups into the caller's buffer)
int qm_device_add_u64(qm_ctx* c, uint64_t* dst, const uint64_t* src, int64_t n) {
  if (!c || !dst || !src || n < 0) return fail(QM_E_INVAL, "qm_device_add_u64: bad arguments");
  HIPCHK(hipSetDevice(c->dev));
  launch_add_u64(dst, src, n, c->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return QM_OK;
}
// bytes of a device buffer cleared on the context's device (qm_extract_files_ex: the caller's per-truth-file sums)
int qm_device_zero(qm_ctx* c, void* dst, size_t bytes) {
  if (!c || (!dst && bytes)) return fail(QM_E_INVAL, "qm_device_zero: bad arguments");
  HIPCHK(hipSetDevice(c->dev));
  if (bytes) HIPCHK(hipMemsetAsync(dst, 0, bytes, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return QM_OK;
}
extern "C" int qm_path_stats_total(qm_ctx* c, int64_t* out) {
  if (!c || !out) return fail(QM_E_INVAL, "qm_path_stats_total: NULL");
  memcpy(out, c->path_total, sizeof c->path_total);
  return QM_OK;
}

extern "C" int qm_batch_path_stats(qm_batch* b, int64_t* out) {
  NEED_FINISHED(b, "qm_batch_path_stats");
  if (!out) return fail(QM_E_INVAL, "qm_batch_path_stats: NULL");
  memcpy(out, b->path_stats, sizeof b->path_stats);
  return QM_OK;
}
extern "C" int qm_batch_global_device(qm_batch* b, void** dev) {
  NEED_FINISHED(b, "qm_batch_global_device");
  if (!dev) return fail(QM_E_INVAL, "qm_batch_global_device: NULL");
  *dev = b->last_global;
  return QM_OK;
}
extern "C" int qm_batch_get_columns(qm_batch* b, int v, int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags) {
  if (!b || v < 0 || v >= b->n_vcf) return fail(QM_E_INVAL, "qm_batch_get_columns: bad arguments");
  HIPCHK(hipSetDevice(b->ctx->dev));
  const VcfDesc& d = b->L.vcfs[(size_t)v];
  const size_t n = (size_t)d.n;
  if (!n) return QM_OK;
  if (pos) HIPCHK(hipMemcpy(pos, b->pos + d.off, n * 4, hipMemcpyDeviceToHost));
  if (ref) HIPCHK(hipMemcpy(ref, b->ref + d.off, n * 4, hipMemcpyDeviceToHost));
  if (alt) HIPCHK(hipMemcpy(alt, b->alt + d.off, n * 4, hipMemcpyDeviceToHost));
  if (qual) HIPCHK(hipMemcpy(qual, b->qual + d.off, n * 4, hipMemcpyDeviceToHost));
  if (flags) HIPCHK(hipMemcpy(flags, b->flags + d.off, n, hipMemcpyDeviceToHost));
  return QM_OK;
}

// ---------------------------------------------------------------------------
// one-shot host-buffer entry point
// ---------------------------------------------------------------------------
extern "C" int qm_classify_batch(qm_ctx* c, int n_vcf, const int64_t* rec_offsets, const int32_t* pos, const int32_t* ref,
                                 const int32_t* alt, const float* qual, const uint8_t* flags, const int32_t* truth_id_per_vcf,
                                 int n_bins, uint8_t* out_cls, uint64_t* out_roc, int64_t* out_scalars, int32_t* out_idx,
                                 uint64_t* out_global) {
  return qm_classify_batch_ext(c, n_vcf, rec_offsets, pos, ref, alt, qual, flags, truth_id_per_vcf, n_bins, 0u, out_cls, out_roc,
                               out_scalars, out_idx, out_global);
}

extern "C" int qm_classify_batch_ext(qm_ctx* c, int n_vcf, const int64_t* rec_offsets, const int32_t* pos, const int32_t* ref,
                                     const int32_t* alt, const float* qual, const uint8_t* flags, const int32_t* truth_id_per_vcf,
                                     int n_bins, unsigned mode, uint8_t* out_cls, uint64_t* out_roc, int64_t* out_scalars,
                                     int32_t* out_idx, uint64_t* out_global) {
  if (!c || n_vcf <= 0 || !rec_offsets || !truth_id_per_vcf) return fail(QM_E_INVAL, "qm_classify_batch: bad arguments");
  std::vector<int64_t> n((size_t)n_vcf);
  for (int v = 0; v < n_vcf; ++v) {
    n[(size_t)v] = rec_offsets[v + 1] - rec_offsets[v];
    if (n[(size_t)v] < 0) return fail(QM_E_INVAL, "qm_classify_batch: rec_offsets must be non-decreasing");
  }
  qm_batch* b = nullptr;
  int rc = qm_batch_create_ext(c, n_vcf, n.data(), truth_id_per_vcf, n_bins, mode, &b);
  if (rc != QM_OK) return rc;
  for (int v = 0; v < n_vcf && rc == QM_OK; ++v) {
    const int64_t o = rec_offsets[v];
    rc = qm_batch_upload(b, v, pos + o, ref + o, alt + o, qual + o, flags + o);
  }
  if (rc == QM_OK) rc = qm_batch_run(b, nullptr, nullptr);
  if (rc == QM_OK) rc = qm_batch_finish(b, nullptr);
  for (int v = 0; v < n_vcf && rc == QM_OK; ++v) {
    if (out_cls) rc = qm_batch_get_cls(b, v, out_cls + rec_offsets[v]);
    if (rc == QM_OK && out_idx) rc = qm_batch_get_idx(b, v, out_idx + rec_offsets[v]);
  }
  if (rc == QM_OK && out_roc) rc = qm_batch_get_roc(b, out_roc);
  if (rc == QM_OK && out_scalars) rc = qm_batch_get_scalars(b, out_scalars);
  if (rc == QM_OK && out_global) rc = qm_batch_get_global(b, out_global);
  std::string keep = g_err;
  qm_batch_destroy(b);
  g_err = keep;
  return rc;
}

// ---------------------------------------------------------------------------
// self-contained synthetic run (for harnesses that are not Python)
// ---------------------------------------------------------------------------
extern "C" int qm_bench_synth(qm_ctx* c, const qm_synth_cfg* cfg, int n_vcf, int64_t records_per_vcf, int n_bins, int steps,
                              qm_bench_result* out) {
  if (!c || !cfg || !out || n_vcf <= 0 || records_per_vcf <= 0 || steps <= 0) return fail(QM_E_INVAL, "qm_bench_synth: bad arguments");
  int tid = -1;
  int rc = qm_truth_synth_ext(c, cfg->genome_len, cfg->truth_n, cfg->truth_seed, cfg->indel_pct, &tid);
  if (rc != QM_OK) return rc;
  std::vector<int64_t> n((size_t)n_vcf, records_per_vcf);
  std::vector<int32_t> tids((size_t)n_vcf, tid);
  qm_batch* b = nullptr;
  rc = qm_batch_create_ext(c, n_vcf, n.data(), tids.data(), n_bins, cfg->indel_pct > 0 ? QM_BATCH_ALLELES : 0u, &b);
  if (rc != QM_OK) { std::string keep = g_err; (void)qm_truth_release(c, tid); g_err = keep; return rc; }
  rc = qm_batch_synth(b, cfg);
  if (rc == QM_OK) rc = qm_batch_run(b, nullptr, nullptr);      // warm-up, also sorts what needs sorting once
  if (rc == QM_OK) rc = qm_batch_finish(b, nullptr);
  if (rc == QM_OK) rc = qm_batch_set_timing(b, 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < steps && rc == QM_OK; ++i) {
    rc = qm_batch_run(b, nullptr, nullptr);
    if (rc == QM_OK && cfg->shuffled) rc = qm_batch_finish(b, nullptr);
  }
  if (rc == QM_OK) rc = qm_batch_finish(b, nullptr);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  float ms[4] = {0, 0, 0, 0};
  if (rc == QM_OK) rc = qm_batch_timings(b, ms);
  if (rc == QM_OK) {
    std::vector<int64_t> sc((size_t)n_vcf * QM_N_SCALARS);
    rc = qm_batch_get_scalars(b, sc.data());
    memset(out, 0, sizeof *out);
    out->records = (int64_t)n_vcf * records_per_vcf;
    out->seconds_per_step = sec / steps;
    out->classifications_per_s = (double)out->records * steps / sec;
    out->classify_ms = ms[0]; out->finalize_ms = ms[1]; out->compact_ms = ms[2];
    for (int v = 0; v < n_vcf; ++v) {
      out->kept += sc[(size_t)v * QM_N_SCALARS + QM_S_NPASS];
      out->tp_lines += sc[(size_t)v * QM_N_SCALARS + QM_S_TP_LINES];
      out->fp_lines += sc[(size_t)v * QM_N_SCALARS + QM_S_FP_LINES];
    }
    out->device_bytes = qm_batch_device_bytes(b);
  }
  std::string keep = g_err;
  qm_batch_destroy(b);
  (void)qm_truth_release(c, tid);   // the synthetic truth set belonged to this run only
  g_err = keep;
  return rc;
}

// ---------------------------------------------------------------------------
// what this GPU streams: the measured denominators beside the 8 TB/s of the data sheet
// ---------------------------------------------------------------------------
extern "C" int qm_bw_probe(qm_ctx* c, int64_t bytes, int reps, double* gbps3) {
  if (!c || !gbps3 || bytes < (1 << 20) || reps < 1) return fail(QM_E_INVAL, "qm_bw_probe: bad arguments");
  HIPCHK(hipSetDevice(c->dev));
  bytes &= ~(int64_t)4095;
  uint8_t *a = nullptr, *d = nullptr;
  uint32_t* sink = nullptr;
  int rc = dalloc(&a, (size_t)bytes);
  if (rc == QM_OK) rc = dalloc(&d, (size_t)bytes);
  if (rc == QM_OK) rc = dalloc(&sink, 1);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto done = [&](int r) { (void)hipFree(a); (void)hipFree(d); (void)hipFree(sink); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); return r; };
  if (rc != QM_OK) return done(rc);
  hipError_t e = hipMemsetAsync(a, 1, (size_t)bytes, c->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d, 2, (size_t)bytes, c->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e != hipSuccess) return done(fail(QM_E_HIP, "qm_bw_probe: %s", hipGetErrorString(e)));
  for (int mode = 0; mode < 3; ++mode) {   // 0 read only, 1 copy (bytes read + bytes written), 2 write only
    launch_bw_probe(mode, a, d, bytes, sink, c->stream);   // warm-up
    (void)hipEventRecord(e0, c->stream);
    for (int r = 0; r < reps; ++r) launch_bw_probe(mode, a, d, bytes, sink, c->stream);
    (void)hipEventRecord(e1, c->stream);
    e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return done(fail(QM_E_HIP, "qm_bw_probe: %s", hipGetErrorString(e)));
    gbps3[mode] = (mode == 1 ? 2.0 : 1.0) * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;
  }
  return done(QM_OK);
}

// ---------------------------------------------------------------------------
// FP overlap (A7)
// ---------------------------------------------------------------------------
extern "C" int qm_fp_overlap(qm_ctx* c, int n_sets, const int64_t* set_offsets, const int32_t* pos, const int32_t* ref,
                             const int32_t* alt, int64_t* regions) {
  if (!c || n_sets < 1 || n_sets > 5 || !set_offsets || !regions) return fail(QM_E_INVAL, "qm_fp_overlap: bad arguments");
  HIPCHK(hipSetDevice(c->dev));
  const int64_t n = set_offsets[n_sets] - set_offsets[0];
  const int nreg = 1 << n_sets;
  for (int i = 0; i < nreg; ++i) regions[i] = 0;
  if (n <= 0) return QM_OK;
  if (n > 0x7fffff00ll) return fail(QM_E_INVAL, "qm_fp_overlap: too many keys");
  std::vector<int32_t> set_of((size_t)n);
  for (int s = 0; s < n_sets; ++s)
    for (int64_t i = set_offsets[s]; i < set_offsets[s + 1]; ++i) set_of[(size_t)(i - set_offsets[0])] = s;
  const int64_t o = set_offsets[0];
  int32_t *dp = nullptr, *dr = nullptr, *da = nullptr, *ds = nullptr;
  uint32_t *k[2] = {nullptr, nullptr}, *v[2] = {nullptr, nullptr}, *hist = nullptr, *bad = nullptr;
  unsigned long long* dreg = nullptr;
  SortSeg* dseg = nullptr;
  int32_t* dts = nullptr;
  int rc = QM_OK;
  auto cleanup = [&]() {
    void* ps[] = {dp, dr, da, ds, k[0], k[1], v[0], v[1], hist, bad, dreg, dseg, dts};
    for (void* p : ps) (void)hipFree(p);
  };
#define A_(p, cnt) if (rc == QM_OK) rc = dalloc(&(p), (size_t)(cnt));
  A_(dp, n) A_(dr, n) A_(da, n) A_(ds, n) A_(k[0], n + 64) A_(k[1], n + 64) A_(v[0], n + 64) A_(v[1], n + 64)   /* the sort kernels read whole 16-byte pieces */
  A_(hist, (size_t)((n + SORT_TILE - 1) / SORT_TILE) * 256) A_(bad, 1) A_(dreg, nreg) A_(dseg, 1)
  A_(dts, (n + SORT_TILE - 1) / SORT_TILE)
#undef A_
  if (rc != QM_OK) { cleanup(); return rc; }
  hipStream_t st = c->stream;
  hipError_t e = hipMemcpy(dp, pos + o, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dr, ref + o, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(da, alt + o, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ds, set_of.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemsetAsync(bad, 0, 4, st);
  if (e == hipSuccess) e = hipMemsetAsync(dreg, 0, sizeof(unsigned long long) * (size_t)nreg, st);
  if (e != hipSuccess) { cleanup(); return fail(QM_E_HIP, "qm_fp_overlap: %s", hipGetErrorString(e)); }
  launch_overlap_pack(dp, dr, da, ds, n, k[0], v[0], bad, st);
  {
    SortSeg g;
    memset(&g, 0, sizeof g);
    g.n = n; g.ntiles = (int32_t)((n + SORT_TILE - 1) / SORT_TILE);
    std::vector<int32_t> ts((size_t)g.ntiles, 0);
    e = hipMemcpy(dseg, &g, sizeof g, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dts, ts.data(), 4 * ts.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { cleanup(); return fail(QM_E_HIP, "qm_fp_overlap: %s", hipGetErrorString(e)); }
  }
  int cur = 0;
  for (int shift = 0; shift < 32; shift += 8) {
    launch_sort_pass(dseg, dts, 1, (int)((n + SORT_TILE - 1) / SORT_TILE), k[cur], nullptr, v[cur], shift, hist, k[cur ^ 1], nullptr, v[cur ^ 1], 0, st);
    cur ^= 1;
  }
  launch_overlap_count(k[cur], v[cur], n, dreg, st);
  std::vector<unsigned long long> hreg((size_t)nreg);
  uint32_t hbad = 0;
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipMemcpy(hreg.data(), dreg, sizeof(unsigned long long) * (size_t)nreg, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(QM_E_HIP, "qm_fp_overlap: %s", hipGetErrorString(e));
  if (hbad) return fail(QM_E_INVAL, "qm_fp_overlap: a key is not a single-base variant with 0 <= pos < 2^28");
  for (int i = 0; i < nreg; ++i) regions[i] = (int64_t)hreg[(size_t)i];
  return QM_OK;
}

// ---------------------------------------------------------------------------
// The path's one exchange behind the C ABI (round 6; include/qmvt.h "the path's one exchange"): RCCL itself, opened at run
// time -- the library links without it, and a process that never asks for a communicator never loads it.  Only the six entry
// points below are used; their prototypes are RCCL's (rccl.h), restated here so that the build needs no RCCL header either.
// ---------------------------------------------------------------------------
namespace {
typedef struct ncclComm* rcclComm_t;
struct RcclId { char internal[128]; };
enum { RCCL_UINT64 = 5, RCCL_SUM = 0 };   // ncclUint64, ncclSum
struct Rccl {
  void* h = nullptr;
  int (*CommInitAll)(rcclComm_t*, int, const int*) = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};
Rccl* rccl() {
  static Rccl R;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (R.h) return &R;
  const char* names[] = {getenv("QM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (R.h) break;
    R.err = dlerror();
  }
  if (!R.h) return &R;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p) { ok = false; R.err = std::string("librccl lacks ") + n; } return p; };
  R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
  R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
  R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(R.h); R.h = nullptr; }
  return &R;
}
}  // namespace

struct qm_comm {
  std::vector<qm_ctx*> ctxs;          // members this process holds (one for qm_comm_create_rank)
  std::vector<rcclComm_t> comms;      // theirs, in the same order
  std::vector<int64_t> issued;        // collectives per member
  int n_ranks = 0;
};
static_assert(sizeof(qm_comm_id) == sizeof(RcclId), "qm_comm_id carries an ncclUniqueId");

#define RCCLCHK(R, expr)                                                                                        \
  do {                                                                                                          \
    const int e_ = (expr);                                                                                      \
    if (e_ != 0) return fail(QM_E_COMM, "%s failed: %s", #expr, (R)->GetErrorString ? (R)->GetErrorString(e_) : "?"); \
  } while (0)

extern "C" int qm_comm_create(qm_ctx* const* ctxs, int n, qm_comm** out) {
  if (!ctxs || n < 1 || !out) return fail(QM_E_INVAL, "qm_comm_create: bad arguments");
  *out = nullptr;
  std::vector<int> devs;
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(QM_E_INVAL, "qm_comm_create: context %d is NULL", i);
    for (int d : devs) if (d == ctxs[i]->dev) return fail(QM_E_INVAL, "qm_comm_create: two contexts on device %d (RCCL refuses two ranks on one card)", d);
    devs.push_back(ctxs[i]->dev);
  }
  Rccl* R = rccl();
  if (!R->h) return fail(QM_E_COMM, "qm_comm_create: RCCL could not be opened: %s", R->err.c_str());
  std::unique_ptr<qm_comm> c(new qm_comm);
  c->ctxs.assign(ctxs, ctxs + n);
  c->comms.assign((size_t)n, nullptr);
  c->issued.assign((size_t)n, 0);
  c->n_ranks = n;
  RCCLCHK(R, R->CommInitAll(c->comms.data(), n, devs.data()));
  *out = c.release();
  return QM_OK;
}

extern "C" int qm_comm_make_id(qm_comm_id* out) {
  if (!out) return fail(QM_E_INVAL, "qm_comm_make_id: NULL");
  Rccl* R = rccl();
  if (!R->h) return fail(QM_E_COMM, "qm_comm_make_id: RCCL could not be opened: %s", R->err.c_str());
  RcclId id;
  RCCLCHK(R, R->GetUniqueId(&id));
  memcpy(out->bytes, id.internal, sizeof id.internal);
  return QM_OK;
}

extern "C" int qm_comm_create_rank(qm_ctx* ctx, int rank, int n_ranks, const qm_comm_id* id, qm_comm** out) {
  if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(QM_E_INVAL, "qm_comm_create_rank: bad arguments");
  *out = nullptr;
  Rccl* R = rccl();
  if (!R->h) return fail(QM_E_COMM, "qm_comm_create_rank: RCCL could not be opened: %s", R->err.c_str());
  HIPCHK(hipSetDevice(ctx->dev));
  std::unique_ptr<qm_comm> c(new qm_comm);
  c->ctxs.assign(1, ctx);
  c->comms.assign(1, nullptr);
  c->issued.assign(1, 0);
  c->n_ranks = n_ranks;
  RcclId rid;
  memcpy(rid.internal, id->bytes, sizeof rid.internal);
  RCCLCHK(R, R->CommInitRank(&c->comms[0], n_ranks, rid, rank));
  *out = c.release();
  return QM_OK;
}

extern "C" int qm_allreduce_counters(qm_batch* b, qm_comm* comm, void* stream) {
  NEED_FINISHED(b, "qm_allreduce_counters");
  if (!comm) return fail(QM_E_INVAL, "qm_allreduce_counters: no communicator");
  size_t m = 0;
  while (m < comm->ctxs.size() && comm->ctxs[m] != b->ctx) ++m;
  if (m == comm->ctxs.size()) return fail(QM_E_INVAL, "qm_allreduce_counters: the batch's context is not a member of this communicator");
  Rccl* R = rccl();
  if (!R->h) return fail(QM_E_COMM, "qm_allreduce_counters: RCCL is gone");
  HIPCHK(hipSetDevice(b->ctx->dev));
  hipStream_t st = stream ? (hipStream_t)stream : b->ctx->stream;
  const size_t count = (size_t)b->n_truth * 3 * (size_t)b->n_bins;
  // in place on the per-truth sums of the last run (the caller's global_dev when the run was given one); the one collective of a step
  RCCLCHK(R, R->AllReduce(b->last_global, b->last_global, count, RCCL_UINT64, RCCL_SUM, comm->comms[m], st));
  comm->issued[m] += 1;
  HIPCHK(hipStreamSynchronize(st));
  return QM_OK;
}

extern "C" int64_t qm_comm_collectives(const qm_comm* comm, const qm_ctx* ctx) {
  if (!comm) return -1;
  for (size_t m = 0; m < comm->ctxs.size(); ++m) if (comm->ctxs[m] == ctx) return comm->issued[m];
  return -1;
}

extern "C" void qm_comm_destroy(qm_comm* comm) {
  if (!comm) return;
  Rccl* R = rccl();
  for (size_t m = 0; m < comm->comms.size(); ++m)
    if (comm->comms[m] && R->h) { (void)hipSetDevice(comm->ctxs[m]->dev); (void)R->CommDestroy(comm->comms[m]); }
  delete comm;
}
