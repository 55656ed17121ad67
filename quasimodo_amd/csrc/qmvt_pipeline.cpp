// qmvt_pipeline.cpp -- qm_extract_files: the whole per-VCF worker of the reference
// (program/extract_TP_FP_SNPs.py:12-57 hcmv, :60-105 custom) for MANY VCFs in one call, files in, files out:
//
//   map the inputs  ->  tokenise (host threads, straight into page-locked column buffers)  ->  host path for the
//   lines the columns cannot describe  ->  asynchronous upload of each VCF as soon as it is tokenised  ->  ONE
//   engine batch (classify, finalize, compact)  ->  class masks back (2 bits per record)  ->  the three output
//   files of every VCF, gathered from the mapped input with writev (host threads).
//
// Replaces, per VCF: three awk passes + grep over the input, two awk passes over the truth file, fgrep -wf and
// fgrep -wvf (extract_TP_FP_SNPs.py:24-32,47-57), and the `>` redirections.  Nothing here classifies on the CPU:
// without a HIP device qm_init has already failed.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qmvt.h"

// from qmvt_host.cpp (same library, not part of the public ABI)
void qm_host_count_lines(const uint8_t* text, size_t len, int64_t* n_lines, int64_t* n_data);
int qm_host_scan_threads(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off, uint8_t* line_kind, int32_t* pos,
                         int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_vcf_cols* info, qm_dict* dict, int nthreads);
int qm_host_write_masks(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                        const uint8_t* line_kind, const uint64_t* kept, const uint64_t* tp, const uint8_t* flags, int select);
int qm_host_threads(void);
void qm_set_error(const char* msg);   // qmvt_api.cpp: what qm_last_error returns
int qm_device_add_u64(qm_ctx* ctx, uint64_t* dst, const uint64_t* src, int64_t n);   // qmvt_api.cpp: dst[i] += src[i] on the device, blocking
int qm_device_zero(qm_ctx* ctx, void* dst, size_t bytes);                              // qmvt_api.cpp: on the context's device, blocking

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// user + system seconds of the whole process so far (QM_FILES_TRACE: what a phase cost in CPU time, the currency of a box with a CPU quota)
double cpu_now() {
  struct rusage u;
  if (getrusage(RUSAGE_SELF, &u) != 0) return 0.0;
  return (double)u.ru_utime.tv_sec + (double)u.ru_stime.tv_sec + 1e-6 * ((double)u.ru_utime.tv_usec + (double)u.ru_stime.tv_usec);
}

struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
  bool ok = false;
  Mapped() = default;
  Mapped(const Mapped&) = delete;
  Mapped& operator=(const Mapped&) = delete;
  Mapped(Mapped&& o) noexcept : p(o.p), n(o.n), ok(o.ok) { o.p = nullptr; o.n = 0; o.ok = false; }
  void open_file(const char* path) {
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return;
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); return; }
    n = (size_t)st.st_size;
    if (n == 0) { ok = true; ::close(fd); return; }
    void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { n = 0; return; }
    (void)madvise(m, n, MADV_SEQUENTIAL | MADV_WILLNEED);
    p = (const uint8_t*)m;
    ok = true;
  }
  ~Mapped() { if (p) munmap((void*)p, n); }
};

template <typename F> void parallel_for(int n, int nthreads, F f) {
  if (n <= 0) return;
  nthreads = std::max(1, std::min(nthreads, n));
  if (nthreads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::atomic<int> next{0};
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); });
  for (auto& x : th) x.join();
}

// page-locked column buffers, kept by the process across calls (pinning memory costs more than filling it)
struct PinnedArena {
  uint8_t* base = nullptr;
  size_t cap = 0;
  bool pinned = false;
  uint8_t* get(size_t need) {
    if (need <= cap) return base;
    release();
    need = need + need / 4 + (1 << 20);
    void* p = nullptr;
    if (hipHostMalloc(&p, need, hipHostMallocDefault) == hipSuccess) { base = (uint8_t*)p; cap = need; pinned = true; return base; }
    (void)hipGetLastError();
    base = (uint8_t*)malloc(need);   // pageable: the copies still work, only slower
    cap = base ? need : 0;
    pinned = false;
    return base;
  }
  void release() {
    if (base) { if (pinned) (void)hipHostFree(base); else free(base); }
    base = nullptr; cap = 0;
  }
  ~PinnedArena() { release(); }
};
// One arena per context (= per device): calls on different contexts -- a thread and a context per GPU, examples/qm_multi.c --
// tokenise, upload and write side by side; two calls on ONE context take turns.  The table itself is never destroyed: at
// process exit the HIP runtime may be gone before static destructors run; qm_destroy releases a context's arena.
struct CtxArena { PinnedArena arena[2]; std::mutex mu; };
std::map<qm_ctx*, CtxArena*>& g_arenas = *new std::map<qm_ctx*, CtxArena*>();
std::mutex* g_arenas_mu = new std::mutex();
CtxArena* arena_of(qm_ctx* ctx) {
  std::lock_guard<std::mutex> g(*g_arenas_mu);
  CtxArena*& a = g_arenas[ctx];
  if (!a) a = new CtxArena();
  return a;
}

struct JobState {
  Mapped vcf;
  int64_t n_lines = 0, n_data = 0;
  std::vector<int64_t> line_off;
  std::vector<uint8_t> line_kind;
  int32_t *pos = nullptr, *ref = nullptr, *alt = nullptr;
  float* qual = nullptr;
  uint8_t* flags = nullptr;
  uint64_t *kept = nullptr, *tp = nullptr;   // class masks, (n_data + 63) / 64 words each
  qm_vcf_cols info{};
  int64_t ex[5] = {0, 0, 0, 0, 0};
  int truth = -1;      // index into the call's distinct truth files
  int batch_v = -1;    // VCF index inside the engine batch (mixed samples only)
  int rc = QM_OK;
  std::string err;     // the message that belongs to rc (qm_last_error is per thread: copied on the worker that failed)
};

struct TruthState {
  std::string path;
  int mode = 0;
  Mapped file;
  qm_patterns* pats = nullptr;
  int64_t info[4] = {0, 0, 0, 0}, counts[5] = {0, 0, 0, 0, 0};
  int tid = -1;
  int rc = QM_OK;
};

int fail(int code, const std::string& msg) { qm_set_error(msg.c_str()); return code; }

}  // namespace

// called by qm_destroy: the page-locked buffers of a context go with it
void qm_pipeline_ctx_destroyed(qm_ctx* ctx) {
  CtxArena* a = nullptr;
  {
    std::lock_guard<std::mutex> g(*g_arenas_mu);
    auto it = g_arenas.find(ctx);
    if (it != g_arenas.end()) { a = it->second; g_arenas.erase(it); }
  }
  if (a) { { std::lock_guard<std::mutex> g(a->mu); a->arena[0].release(); a->arena[1].release(); } delete a; }
}

extern "C" int qm_extract_files(qm_ctx* ctx, int n_jobs, const qm_file_job* jobs, int n_bins, unsigned mode, int strict,
                                qm_file_stats* stats, uint64_t* roc_out, double* phase_seconds) {
  return qm_extract_files_ex(ctx, n_jobs, jobs, n_bins, mode, strict, stats, roc_out, phase_seconds, nullptr, 0, nullptr);
}

extern "C" int qm_extract_files_ex(qm_ctx* ctx, int n_jobs, const qm_file_job* jobs, int n_bins, unsigned mode, int strict,
                                   qm_file_stats* stats, uint64_t* roc_out, double* phase_seconds, const int32_t* truth_slot,
                                   int n_slots, void* global_dev) {
  if (!ctx || n_jobs < 0 || (n_jobs && !jobs) || n_bins < 1 || n_bins > QM_MAX_BINS || (mode & ~(unsigned)QM_BATCH_ALLELES))
    return fail(QM_E_INVAL, "qm_extract_files: bad arguments");
  if (global_dev && (n_slots < 1 || (n_jobs && !truth_slot))) return fail(QM_E_INVAL, "qm_extract_files_ex: global_dev needs truth_slot and n_slots >= 1");
  const size_t gbytes = global_dev ? (size_t)n_slots * 3 * (size_t)n_bins * sizeof(uint64_t) : 0;
  if (global_dev) { const int rc0 = qm_device_zero(ctx, global_dev, gbytes); if (rc0 != QM_OK) return rc0; }   // (on the context's device, whatever the caller's thread had current)
  if (n_jobs == 0) {   // nothing to do is not an error (a rank of a sharded run may hold no VCF)
    if (phase_seconds) memset(phase_seconds, 0, 8 * sizeof(double));
    return QM_OK;
  }
  const bool ext = (mode & QM_BATCH_ALLELES) != 0;
  // map + count, truth sets (beside the former), batch layout, tokenise + host path (+ uploads beside it), engine, masks back,
  // write, release -- summed over the groups of the pipeline below (stages of different groups overlap, so the sum of the
  // phases exceeds the wall time of the call)
  double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, phc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::mutex ph_mu;
  const bool trace = getenv("QM_FILES_TRACE") != nullptr;
  auto add_ph = [&](int k, double dt, double dc = 0.0) { std::lock_guard<std::mutex> g(ph_mu); ph[k] += dt; phc[k] += dc; };
  const int nthr = qm_host_threads();
  std::vector<JobState> J((size_t)n_jobs);
  for (int j = 0; j < n_jobs; ++j) {
    const qm_file_job& q = jobs[j];
    if (!q.vcf_path || !q.filtered_out || !q.fp_out || (!q.pure && (!q.truth_path || !q.tp_out)) || (q.mode != 0 && q.mode != 1))
      return fail(QM_E_INVAL, "qm_extract_files: job " + std::to_string(j) + " is incomplete");
    if (ext && !q.pure && q.mode != 0) return fail(QM_E_INVAL, "qm_extract_files: the allele-extended mode needs VCF truth sets (mode 0)");
  }
  qm_dict* dict = ext ? qm_dict_create() : nullptr;
  std::vector<TruthState> T;

  // ---- groups: the VCFs CAN go through a two-stage pipeline, group after group -- while group g is on the engine and its
  //      files are being written (a thread of its own, fanning out over the host threads), group g + 1 is mapped, tokenised
  //      and uploaded.  Measured (round 3, 16 VCFs of 10^6 lines, profiles/r03_e2e_groups.log): 83 ms in one group, 86 / 108 /
  //      142 ms in 2 / 4 / 6 -- every stage parallelises over FILES (one thread counts a file, one thread writes an output
  //      file), so a stage takes as long as its slowest file however few files it holds, and smaller groups only idle
  //      threads.  The default is therefore ONE group (QM_FILES_GROUPS / QM_FILES_GROUP_MB for experiments); what would make
  //      groups pay is parallelism INSIDE a file in the count and in the writers.
  struct Group {
    std::vector<int> jobs;             // indices into `jobs`, ascending
    qm_batch* batch = nullptr;
    std::vector<int64_t> nrec;
    std::vector<int32_t> tids;
    hipStream_t copy_stream = nullptr;
    std::vector<int64_t> scal;
    std::vector<uint64_t> roc;
    int rc = QM_OK;
    std::string err;
  };
  std::vector<Group> G;
  {
    int64_t total = 0;
    std::vector<int64_t> sz((size_t)n_jobs, 0);
    for (int j = 0; j < n_jobs; ++j) { struct stat st; if (stat(jobs[j].vcf_path, &st) == 0) sz[(size_t)j] = (int64_t)st.st_size; total += sz[(size_t)j]; }
    int want = 1;
    if (const char* e = getenv("QM_FILES_GROUPS")) want = std::max(1, atoi(e));
    want = std::min(want, n_jobs);
    G.resize((size_t)want);
    int64_t acc = 0;
    int g = 0;
    for (int j = 0; j < n_jobs; ++j) {
      while (g + 1 < want && acc >= total * (g + 1) / want && !G[(size_t)g].jobs.empty()) ++g;
      G[(size_t)g].jobs.push_back(j);
      acc += sz[(size_t)j];
    }
    while (!G.empty() && G.back().jobs.empty()) G.pop_back();
  }
  // The context is not made for two threads: with more than one group (QM_FILES_GROUPS / QM_FILES_GROUP_MB; experimental, the
  // default is one) stage one of group g + 1 runs beside stage two of group g, so whatever touches the context's streams and
  // truth tables -- batch creation there, run / finish / row read-back here -- takes turns.  (The uploads of the tokenizer
  // threads write their own batch through a stream of its own.)
  std::mutex engine_mu;
  std::thread truth_thread;   // ends with the patterns of the truth files
  std::thread stage2;         // engine + masks + files of the group before the one being tokenised
  auto cleanup = [&]() {
    if (stage2.joinable()) stage2.join();
    if (truth_thread.joinable()) truth_thread.join();
    for (auto& g : G) { if (g.batch) { qm_batch_destroy(g.batch); g.batch = nullptr; } if (g.copy_stream) { (void)hipStreamDestroy(g.copy_stream); g.copy_stream = nullptr; } }
    for (auto& t : T) { if (t.pats) qm_patterns_destroy(t.pats); if (t.tid >= 0) (void)qm_truth_release(ctx, t.tid); }
    if (dict) qm_dict_destroy(dict);
  };

  // ---- 1. the distinct truth files (keys for the device, patterns as text for the host path), on a thread of their own ...
  {
    std::map<std::pair<std::string, int>, int> seen;
    for (int j = 0; j < n_jobs; ++j) {
      if (jobs[j].pure) continue;
      const auto key = std::make_pair(std::string(jobs[j].truth_path), (int)jobs[j].mode);
      auto it = seen.find(key);
      if (it == seen.end()) { it = seen.emplace(key, (int)T.size()).first; T.emplace_back(); T.back().path = key.first; T.back().mode = key.second; }
      J[(size_t)j].truth = it->second;
    }
  }
  std::vector<int32_t> slot_of_truth(T.size(), -1);
  if (global_dev) {
    for (int j = 0; j < n_jobs; ++j) {
      if (jobs[j].pure) continue;
      const int32_t sl = truth_slot[j];
      int32_t& have = slot_of_truth[(size_t)J[(size_t)j].truth];
      if (sl < 0 || sl >= n_slots || (have >= 0 && have != sl)) {
        if (dict) qm_dict_destroy(dict);
        return fail(QM_E_INVAL, "qm_extract_files_ex: job " + std::to_string(j) + " names row " + std::to_string(sl) + " for a truth file that another job puts elsewhere (or outside [0, n_slots))");
      }
      have = sl;
    }
  }
  int truth_rc = QM_OK;
  std::string truth_msg;
  // The keys go first (the batch layout needs the truth ids as soon as the VCFs are counted); the patterns as text -- a hash
  // set of every row, wanted only by the host path and by the decision whether a VCF needs it -- are built on a thread of
  // their own and waited for by the first VCF that has been tokenised.
  std::mutex pats_mu;
  std::condition_variable pats_cv;
  bool pats_ready = false, keys_ready = false;
  truth_thread = std::thread([&]() {
    const double tt0 = now(), tc0 = trace ? cpu_now() : 0.0;
    parallel_for((int)T.size(), std::max(1, nthr / 4), [&](int k) {
      TruthState& t = T[(size_t)k];
      t.file.open_file(t.path.c_str());
      if (!t.file.ok) t.rc = QM_E_IO;
    });
    std::thread pats_thread([&]() {
      parallel_for((int)T.size(), std::max(1, nthr / 4), [&](int k) {
        TruthState& t = T[(size_t)k];
        if (t.rc != QM_OK) return;
        t.pats = qm_patterns_create(t.file.p, t.file.n, t.mode, ext ? 1 : 0);
        if (!t.pats) { t.rc = QM_E_INVAL; return; }
        (void)qm_patterns_info(t.pats, t.info);
      });
      { std::lock_guard<std::mutex> g(pats_mu); pats_ready = true; }
      pats_cv.notify_all();
    });
    for (auto& t : T) {
      if (truth_rc != QM_OK) break;
      if (t.rc == QM_E_IO) { truth_rc = t.rc; truth_msg = "cannot read truth file " + t.path; break; }
      const int64_t cap = qm_vcf_count_lines(t.file.p, t.file.n) + 1;
      std::vector<int32_t> tp((size_t)cap), tr((size_t)cap), ta((size_t)cap);
      const int64_t k = qm_truth_scan_ext(t.file.p, t.file.n, t.mode, cap, tp.data(), tr.data(), ta.data(), t.counts, dict);
      if (k < 0) { truth_rc = (int)k; truth_msg = "qm_truth_scan failed for " + t.path; break; }
      const int rc = qm_truth_load(ctx, tp.data(), tr.data(), ta.data(), k, &t.tid);
      if (rc != QM_OK) { truth_rc = rc; truth_msg = qm_last_error(ctx); break; }
    }
    add_ph(1, now() - tt0, trace ? cpu_now() - tc0 : 0.0);
    { std::lock_guard<std::mutex> g(pats_mu); keys_ready = true; }
    pats_cv.notify_all();
    pats_thread.join();   // (the VCFs do not wait for this thread but for pats_ready)
  });
  auto wait_patterns = [&]() { std::unique_lock<std::mutex> g(pats_mu); pats_cv.wait(g, [&] { return pats_ready; }); };

  CtxArena* const ca = arena_of(ctx);
  std::unique_lock<std::mutex> arena_lock(ca->mu);   // one call at a time per context uses its page-locked arenas

  // ---- stage 1 of a group: map + count, batch layout, tokenise + host path + uploads ----
  auto stage_one = [&](Group& gr, int gi) -> int {
    const int ng = (int)gr.jobs.size();
    double t0 = now(), c0 = trace ? cpu_now() : 0.0;
    parallel_for(ng, nthr, [&](int k) {
      const int j = gr.jobs[(size_t)k];
      JobState& s = J[(size_t)j];
      s.vcf.open_file(jobs[j].vcf_path);
      if (!s.vcf.ok) { s.rc = QM_E_IO; return; }
      int64_t nl = 0, nd = 0;
      qm_host_count_lines(s.vcf.p, s.vcf.n, &nl, &nd);
      s.n_lines = nl; s.n_data = nd;
    });
    add_ph(0, now() - t0, trace ? cpu_now() - c0 : 0.0);
    { std::unique_lock<std::mutex> g(pats_mu); pats_cv.wait(g, [&] { return keys_ready; }); }   // the thread itself ends with the patterns
    for (int j : gr.jobs)
      if (J[(size_t)j].rc != QM_OK) return fail(QM_E_IO, std::string("cannot read ") + jobs[j].vcf_path);
    if (truth_rc != QM_OK) return fail(truth_rc, truth_msg);
    t0 = now(); c0 = trace ? cpu_now() : 0.0;
    for (int j : gr.jobs)
      if (!jobs[j].pure) { J[(size_t)j].batch_v = (int)gr.nrec.size(); gr.nrec.push_back(J[(size_t)j].n_data); gr.tids.push_back(T[(size_t)J[(size_t)j].truth].tid); }
    if (!gr.nrec.empty()) {
      std::lock_guard<std::mutex> eg(engine_mu);
      const int rc = qm_batch_create_ext(ctx, (int)gr.nrec.size(), gr.nrec.data(), gr.tids.data(), n_bins, mode, &gr.batch);
      if (rc != QM_OK) return rc;
    }
    size_t need = 0;
    std::vector<size_t> aoff((size_t)ng);
    for (int k = 0; k < ng; ++k) {
      aoff[(size_t)k] = need;
      const size_t cap = (size_t)J[(size_t)gr.jobs[(size_t)k]].n_lines + 1;
      need += ((cap * 17 + 255) & ~(size_t)255) + ((((cap + 63) / 64) * 16 + 255) & ~(size_t)255);
    }
    uint8_t* arena = ca->arena[gi & 1].get(need);   // two arenas take turns: group g - 2 has been written by now
    if (!arena) return fail(QM_E_NOMEM, "qm_extract_files: no memory for the column buffers");
    if (hipStreamCreateWithFlags(&gr.copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(QM_E_HIP, "hipStreamCreate failed");
    add_ph(2, now() - t0, trace ? cpu_now() - c0 : 0.0);
    t0 = now(); c0 = trace ? cpu_now() : 0.0;
    const int per_file_threads = std::max(1, nthr / std::max(1, std::min(ng, nthr)));
    parallel_for(ng, nthr, [&](int k) {
      const int j = gr.jobs[(size_t)k];
      JobState& s = J[(size_t)j];
      const size_t cap = (size_t)s.n_lines + 1;
      uint8_t* a = arena + aoff[(size_t)k];
      s.pos = (int32_t*)a; s.ref = s.pos + cap; s.alt = s.ref + cap; s.qual = (float*)(s.alt + cap); s.flags = (uint8_t*)(s.qual + cap);
      s.kept = (uint64_t*)(a + ((cap * 17 + 255) & ~(size_t)255)); s.tp = s.kept + (cap + 63) / 64;
      s.line_off.resize(cap + 1);
      s.line_kind.resize(cap);
      s.rc = qm_host_scan_threads(s.vcf.p, s.vcf.n, (int64_t)cap, s.line_off.data(), s.line_kind.data(), s.pos, s.ref, s.alt, s.qual, s.flags,
                                  &s.info, dict, per_file_threads > 1 ? per_file_threads : -1);   // -1: one thread, lines counted above
      if (s.rc != QM_OK || s.info.n_data != s.n_data) {
        // (the file changed between the count and the scan: the scan stops at the room it was given)
        s.err = s.rc == QM_OK || s.rc == QM_E_INVAL ? "the file changed while it was being read" : "tokenising failed";
        if (s.rc == QM_OK) s.rc = QM_E_INVAL;
        return;
      }
      if (jobs[j].pure) return;
      wait_patterns();
      const TruthState& t = T[(size_t)s.truth];
      if (t.rc != QM_OK) { s.rc = t.rc; return; }
      if (s.info.n_host || s.info.n_nokey_kept || t.info[1] > 0 || t.info[2] > 0) {
        s.rc = qm_vcf_hostpath(t.pats, s.vcf.p, s.vcf.n, s.info.n_lines, s.line_off.data(), s.line_kind.data(), s.pos, s.ref, s.alt, s.flags, s.ex);
        if (s.rc != QM_OK) s.err = "the host path (fgrep -w on the text) failed";
      }
      if (s.rc == QM_OK && !(strict && s.info.n_refused)) {
        s.rc = qm_batch_upload_async(gr.batch, s.batch_v, s.pos, s.ref, s.alt, s.qual, s.flags, gr.copy_stream);
        if (s.rc != QM_OK) s.err = qm_last_error(ctx);   // this thread's message: the caller's thread would not see it
      }
    });
    int rc = QM_OK;
    wait_patterns();
    for (const auto& t : T) {
      if (rc != QM_OK) break;
      if (t.rc != QM_OK) rc = fail(t.rc, "cannot take the patterns of truth file " + t.path);
      else if (strict && t.info[3] > 0) rc = fail(QM_E_NONCANON, t.path + ": " + std::to_string(t.info[3]) + " truth rows hold NUL or non-ASCII bytes");
    }
    for (int k = 0; k < ng && rc == QM_OK; ++k) {
      const int j = gr.jobs[(size_t)k];
      const JobState& s = J[(size_t)j];
      if (s.rc != QM_OK) rc = fail(s.rc, std::string("tokenising / uploading failed for ") + jobs[j].vcf_path + (s.err.empty() ? "" : ": " + s.err));
      else if (strict && s.info.n_refused)
        rc = fail(QM_E_NONCANON, std::string(jobs[j].vcf_path) + " line " + std::to_string(s.info.first_refused_line) +
                                     ": a kept line holds NUL or non-ASCII bytes -- the reference's answer for it depends on the locale "
                                     "Python exports to grep; set QM_LENIENT=1 to classify it by its columns");
    }
    if (hipStreamSynchronize(gr.copy_stream) != hipSuccess && rc == QM_OK) rc = fail(QM_E_HIP, "upload failed");
    add_ph(3, now() - t0, trace ? cpu_now() - c0 : 0.0);
    return rc;
  };

  // ---- stage 2 of a group (a thread of its own): the engine, the class masks back, the three files of every VCF, the rows ----
  auto stage_two = [&](Group& gr) {
    int rc = QM_OK;
    double t0 = now(), c0 = trace ? cpu_now() : 0.0;
    if (gr.batch) {
      std::lock_guard<std::mutex> eg(engine_mu);
      rc = qm_batch_run(gr.batch, nullptr, nullptr);
      if (rc == QM_OK) rc = qm_batch_finish(gr.batch, nullptr);
      gr.scal.resize(gr.nrec.size() * QM_N_SCALARS);
      gr.roc.resize(gr.nrec.size() * 3 * (size_t)n_bins);
      if (rc == QM_OK) rc = qm_batch_get_scalars(gr.batch, gr.scal.data());
      if (rc == QM_OK) rc = qm_batch_get_roc(gr.batch, gr.roc.data());
      if (rc == QM_OK && global_dev) {
        // the per-truth-set sums as the engine left them in HBM, row by row ADDED into the caller's layout: what a multi-GPU
        // caller all-reduces (device to device: the counters never visit the host)
        void* src = nullptr;
        rc = qm_batch_global_device(gr.batch, &src);
        const size_t roww = 3 * (size_t)n_bins;
        for (size_t k = 0; k < T.size() && rc == QM_OK; ++k) {
          if (slot_of_truth[k] < 0 || T[k].tid < 0) continue;
          rc = qm_device_add_u64(ctx, (uint64_t*)global_dev + (size_t)slot_of_truth[k] * roww, (const uint64_t*)src + (size_t)T[k].tid * roww, (int64_t)roww);
        }
      }
      if (rc != QM_OK) gr.err = qm_last_error(ctx);
    }
    add_ph(4, now() - t0, trace ? cpu_now() - c0 : 0.0);
    t0 = now(); c0 = trace ? cpu_now() : 0.0;
    for (size_t k = 0; k < gr.jobs.size() && rc == QM_OK; ++k) {
      const int j = gr.jobs[k];
      JobState& s = J[(size_t)j];
      if (jobs[j].pure) continue;
      rc = qm_batch_get_masks(gr.batch, s.batch_v, s.kept, s.tp);
      if (rc != QM_OK) gr.err = qm_last_error(ctx);
    }
    add_ph(5, now() - t0, trace ? cpu_now() - c0 : 0.0);
    if (gr.copy_stream) { (void)hipStreamDestroy(gr.copy_stream); gr.copy_stream = nullptr; }
    if (rc != QM_OK) { gr.rc = rc; return; }
    t0 = now(); c0 = trace ? cpu_now() : 0.0;
    struct WTask { int j, select; const char* path; bool pure; };
    std::vector<WTask> W;
    for (int j : gr.jobs) {
      if (jobs[j].pure) { W.push_back({j, 0, jobs[j].filtered_out, true}); W.push_back({j, 0, jobs[j].fp_out, true}); }   // cp filtered fp (:33-36)
      else { W.push_back({j, 0, jobs[j].filtered_out, false}); W.push_back({j, 1, jobs[j].tp_out, false}); W.push_back({j, 2, jobs[j].fp_out, false}); }
    }
    std::vector<int> wrc(W.size(), QM_OK);
    parallel_for((int)W.size(), nthr, [&](int k) {
      const WTask& w = W[(size_t)k];
      const JobState& s = J[(size_t)w.j];
      // pure-strain samples never reach the device: kept = the A2 filter's verdict, which the tokenizer left in the flags
      wrc[(size_t)k] = qm_host_write_masks(w.path, s.vcf.p, s.vcf.n, s.info.n_lines, s.line_off.data(), s.line_kind.data(),
                                           w.pure ? nullptr : s.kept, w.pure ? nullptr : s.tp, s.flags, w.select);
    });
    for (size_t k = 0; k < W.size(); ++k)
      if (wrc[k] != QM_OK) { gr.rc = wrc[k]; gr.err = std::string("cannot write ") + W[k].path; return; }
    add_ph(6, now() - t0, trace ? cpu_now() - c0 : 0.0);
    // per-VCF rows
    for (int j : gr.jobs) {
      const JobState& s = J[(size_t)j];
      int64_t hk = 0, hk_tp = 0;
      for (int64_t i = 0; i < s.info.n_lines; ++i) { hk += s.line_kind[(size_t)i] == QM_LINE_HEADER_KEPT || s.line_kind[(size_t)i] == QM_LINE_HEADER_KEPT_TP; hk_tp += s.line_kind[(size_t)i] == QM_LINE_HEADER_KEPT_TP; }
      if (stats) {
        qm_file_stats& o = stats[j];
        memset(&o, 0, sizeof o);
        o.n_lines = s.info.n_lines; o.n_refused = s.info.n_refused; o.header_kept = hk; o.header_kept_tp = hk_tp; o.host_decided = s.ex[0]; o.r_hostile = s.info.n_r_hostile;
        if (jobs[j].pure) {
          int64_t np = 0;
          for (int64_t r = 0; r < s.n_data; ++r) np += s.flags[r] & QM_F_PASS;
          o.scalars[QM_S_NPASS] = np; o.scalars[QM_S_FP_LINES] = np; o.scalars[QM_S_SORTED] = 1; o.scalars[QM_S_NREC] = s.n_data;
        } else {
          memcpy(o.scalars, &gr.scal[(size_t)s.batch_v * QM_N_SCALARS], sizeof o.scalars);
          // R keys a line by the TEXT of POS / REF / ALT; for lines without a comparable key the device counted distinct
          // (carried pos, ref, alt) instead: swap those for the text keys (qm_vcf_hostpath)
          o.scalars[QM_S_FP_R] += s.ex[4] - s.ex[2];
          o.scalars[QM_S_TP_R] += s.ex[3];
          o.genomediff = T[(size_t)s.truth].counts[0];
        }
      }
      if (roc_out) {
        uint64_t* dst = roc_out + (size_t)j * 3 * (size_t)n_bins;
        if (jobs[j].pure) memset(dst, 0, sizeof(uint64_t) * 3 * (size_t)n_bins);
        else memcpy(dst, &gr.roc[(size_t)s.batch_v * 3 * (size_t)n_bins], sizeof(uint64_t) * 3 * (size_t)n_bins);
      }
    }
    // the group is done: its batch, its mappings and its line tables go while the next group is at work
    t0 = now(); c0 = trace ? cpu_now() : 0.0;
    if (gr.batch) { qm_batch_destroy(gr.batch); gr.batch = nullptr; }
    const double t1 = now();
    parallel_for((int)gr.jobs.size(), nthr, [&](int k) { JobState tmp = std::move(J[(size_t)gr.jobs[(size_t)k]]); (void)tmp; });   // unmap / free in parallel
    if (getenv("QM_FILES_TRACE")) fprintf(stderr, "release: batch destroy %.2f ms, unmap + free %.2f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
    add_ph(7, now() - t0, trace ? cpu_now() - c0 : 0.0);
  };

  int rc = QM_OK;
  std::string msg;
  for (size_t g = 0; g < G.size() && rc == QM_OK; ++g) {
    rc = stage_one(G[g], (int)g);
    if (rc != QM_OK) msg = qm_last_error(ctx);
    if (stage2.joinable()) stage2.join();          // group g - 1 is written (and its arena free for group g + 1)
    if (g > 0 && rc == QM_OK && G[g - 1].rc != QM_OK) { rc = G[g - 1].rc; msg = G[g - 1].err; }
    if (rc == QM_OK) stage2 = std::thread(stage_two, std::ref(G[g]));
  }
  if (stage2.joinable()) stage2.join();
  if (rc == QM_OK) for (auto& g : G) if (g.rc != QM_OK) { rc = g.rc; msg = g.err; break; }
  cleanup();
  if (rc != QM_OK) return fail(rc, msg);
  if (phase_seconds) memcpy(phase_seconds, ph, sizeof ph);
  if (trace)   // (process-wide CPU time between a phase's two clock readings: phases that run beside each other share it)
    fprintf(stderr, "cpu seconds: map_count %.3f truth_beside %.3f batch_layout %.3f tokenise_upload %.3f engine %.3f masks_back %.3f write %.3f release %.3f\n",
            phc[0], phc[1], phc[2], phc[3], phc[4], phc[5], phc[6], phc[7]);
  return QM_OK;
}
