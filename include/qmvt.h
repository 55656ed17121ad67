/*
 * qmvt.h -- C ABI of libqmvt.so, the MI355X (gfx950) variant-truth engine.
 *
 * The reference (hzi-bifo/Quasimodo v0.4.2) has no FFI for this path: its
 * boundary is a per-VCF process,
 *     python program/extract_TP_FP_SNPs.py <vcf> <truth> {hcmv,custom} <outdir> <caller>
 * (program/extract_TP_FP_SNPs.py:124-140, invoked by rules/extract_TP.smk:20 and
 * eval_variant_custom.smk:73) that shells out to awk/grep.  Each entry point
 * below names the reference stage it replaces.  All signatures are plain C:
 * pointers + sizes, no C++/torch types.  Return value: 0 = QM_OK, negative =
 * error (text via qm_last_error).  The library never falls back to a CPU
 * implementation of the classification: without a usable HIP device every
 * compute entry point fails with QM_E_NODEVICE.
 *
 * Column encoding (one record = one VCF data line):
 *   pos   int32  POS, 0 <= pos < 2^28 (canonical decimal spelling on the text side)
 *   ref   int32  0..3 = A,C,G,T; any other value = not a single-base allele
 *   alt   int32  same
 *   qual  float  "effective QUAL": floor(qual) >= t  <=>  the record passes the
 *                awk test `$6>=t` (t = 0..n_bins-1); '.' and passing non-numeric
 *                spellings are +inf, failing ones -inf
 *   flags uint8  bit0 QM_F_PASS  = line kept by the A2 filter (extract_TP_FP_SNPs.py:24)
 *                bit1 QM_F_IDDOT = ID column is exactly "." (a key match makes the line a TP line);
 *                                  cleared by qm_vcf_hostpath for a line it found NOT selected
 *                bit2 QM_F_NOKEY = the line has no comparable key (POS is not a canonical
 *                                  decimal): it can never be in the truth set; the packer
 *                                  gives it the previous record's pos so order is kept
 *                bit3 QM_F_TPLINE = the text side (qm_vcf_hostpath) found the line selected by
 *                                  `fgrep -wf` where the columns cannot tell (pattern aligned at
 *                                  other columns, POS spelled non-canonically, ...): a TP LINE
 *                                  whatever its key says; unique-key (R path) counts ignore it
 */
#ifndef QMVT_H
#define QMVT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QM_ABI_VERSION 6

#define QM_OK 0
#define QM_E_INVAL (-1)     /* bad argument */
#define QM_E_NODEVICE (-2)  /* no HIP device / HIP runtime failure at init */
#define QM_E_HIP (-3)       /* a HIP call failed */
#define QM_E_NOMEM (-4)
#define QM_E_RANGE (-5)     /* position outside [0, 2^28) */
#define QM_E_STATE (-6)     /* call order violated */
#define QM_E_IO (-7)
#define QM_E_NONCANON (-8)  /* text input the engine refuses to guess about (strict mode) */
#define QM_E_LIMIT (-9)     /* allele-extended batch: too many records at one position */
#define QM_E_UNSORTED (-10) /* qm_bgzf_write_tbi: sequences not in blocks or positions stepping backwards (tabix refuses such a VCF too) */
#define QM_E_COMM (-11)     /* the collective library (RCCL) is missing or one of its calls failed */

/* ---- allele-extended mode (QM_BATCH_ALLELES) ---------------------------------------------
 * BASELINE.json configs[4] (mixed SNP + indel, variable-length alleles).  The reference drops
 * every non-single-base record at its A2 filter, so this mode is a build-defined widening and is
 * OFF by default: `$4~/^[ACGT]$/&&$5~/^[ACGT]$/` becomes `$4~/^[ACGT]+$/&&$5~/^[ACGT]+$/`, in the
 * caller filter and in the truth pattern list alike; everything else (ID == ".", QUAL clause,
 * line / unique-key counts, ROC) is unchanged, and records with single-base alleles are
 * classified exactly as without the mode.
 * Allele codes (ref / alt columns, int32): two alleles are the same string iff their codes are equal.
 *   0..3                          one base A,C,G,T
 *   len << 26 | bases             2..13 bases inline: base k (0-based, A=0 C=1 G=2 T=3) in bits 2k+1..2k
 *   0x40000000 | id               longer alleles: id from a qm_dict (interned strings)
 *   negative, or 4..0x07ffffff    not an allele that takes part (as without the mode) */
#define QM_BATCH_ALLELES 1u
#define QM_ALLELE_NONE (-1)
#define QM_ALLELE_INLINE_MAX 13
#define QM_ALLELE_DICT 0x40000000

#define QM_F_PASS 1u
#define QM_F_IDDOT 2u
#define QM_F_NOKEY 4u
#define QM_F_TPLINE 8u

#define QM_CLS_KEPT 1u /* out_cls bit0: line is in <x>.filtered.vcf */
#define QM_CLS_TP 2u   /* out_cls bit1: line is in tp/<x>.tp.vcf (else, if kept, fp/) */

#define QM_MAX_BINS 256
#define QM_POS_LIMIT (1 << 28)

/* per-VCF scalar slots (int64 each) */
enum {
  QM_S_NPASS = 0,    /* kept lines = R `calleridentify` (caller_performance_compare.R:82) */
  QM_S_TP_LINES = 1, /* data lines of tp.vcf  (fgrep -wf,  extract_TP_FP_SNPs.py:50) */
  QM_S_FP_LINES = 2, /* data lines of fp.vcf  (fgrep -wvf, extract_TP_FP_SNPs.py:52) */
  QM_S_TP_R = 3,     /* |unique(snp) & truth|  (caller_performance_compare.R:94) */
  QM_S_FP_R = 4,     /* |unique(snp) \ truth|  (caller_performance_compare.R:95) */
  QM_S_SORTED = 5,   /* 1 = positions non-decreasing as given, 0 = went through the radix sort */
  QM_S_NREC = 6,     /* records in the VCF */
  QM_S_TRUTH = 7,    /* distinct truth keys T' of the truth set used (FN_R = T' - TP_R) */
  QM_N_SCALARS = 8
};

typedef struct qm_ctx qm_ctx;
typedef struct qm_batch qm_batch;

/* ---- lifecycle ------------------------------------------------------------ */
int qm_abi_version(void);
/* Which sources this library was built from: first 16 hex digits of the sha256 over the device code's sources
 * (qmvt_kernels.hip, qmvt_dev.h: what a kernel profile belongs to) / over every source of the library, taken by the Makefile
 * at build time.  "unknown" for a build that did not go through it (A/B builds).  The host package compares them with the
 * sources in the tree, so that a stale binary is rebuilt -- and can never be quoted with another build's profile. */
const char* qm_kernels_id(void);
const char* qm_build_id(void);
/* One context per process and GPU.  device_id indexes HIP devices (cuda:N in torch). */
int qm_init(int device_id, qm_ctx** out);
void qm_destroy(qm_ctx* ctx);
/* Last error text of this thread (ctx may be NULL for errors raised before a ctx exists). */
const char* qm_last_error(qm_ctx* ctx);

/* ---- truth sets ------------------------------------------------------------
 * Replaces the `awk ... {print $2,".",$4,$5}` pattern list fed to fgrep
 * (extract_TP_FP_SNPs.py:47-48 hcmv, :92-93 custom): the engine keeps the set of
 * single-base (pos,ref,alt) keys, sorted and de-duplicated, in HBM together
 * with a coarse position index.  Host arrays are copied; rows whose ref/alt is
 * not 0..3 are ignored (they can never match a kept line). */
int qm_truth_load(qm_ctx* ctx, const int32_t* pos, const int32_t* ref, const int32_t* alt, int64_t n,
                  int* truth_id);
int qm_truth_size(qm_ctx* ctx, int truth_id, int64_t* n_unique);
/* distinct valid (pos, ref, alt) entries of any allele length: T' of allele-extended batches */
int qm_truth_size_ext(qm_ctx* ctx, int truth_id, int64_t* n_unique);
/* number of truth-set slots of the context (released ones included: ids are stable) */
int qm_truth_count(qm_ctx* ctx);
/* Frees a truth set's device memory.  Its id may be handed out again by a later load; a batch created
 * against the released set refuses to run (QM_E_STATE). */
int qm_truth_release(qm_ctx* ctx, int truth_id);

/* ---- one-shot, host buffers -------------------------------------------------
 * What n_vcf invocations of the reference script compute (A2 flags in, A4/A5
 * split + A6 counts + ROC out).  VCF v owns records rec_offsets[v]..rec_offsets[v+1].
 * Unsorted VCFs are radix-sorted on the GPU transparently.  Blocking.
 *   out_cls      [N]                QM_CLS_* per record, input order        (may be NULL)
 *   out_roc      [n_vcf][3][n_bins] cumulative TP(t), FP(t), U(t); FN(t) = T' - U(t)
 *   out_scalars  [n_vcf][QM_N_SCALARS]
 *   out_idx      [N]  per VCF region: TP line indices ascending from the front,
 *                     FP line indices ascending ending at the back           (may be NULL)
 *   out_global   [n_truth_sets][3][n_bins] sums over the VCFs of each truth set (may be NULL)
 */
int qm_classify_batch(qm_ctx* ctx, int n_vcf, const int64_t* rec_offsets, const int32_t* pos,
                      const int32_t* ref, const int32_t* alt, const float* qual, const uint8_t* flags,
                      const int32_t* truth_id_per_vcf, int n_bins, uint8_t* out_cls, uint64_t* out_roc,
                      int64_t* out_scalars, int32_t* out_idx, uint64_t* out_global);
/* The same with a batch mode (0 or QM_BATCH_ALLELES). */
int qm_classify_batch_ext(qm_ctx* ctx, int n_vcf, const int64_t* rec_offsets, const int32_t* pos,
                          const int32_t* ref, const int32_t* alt, const float* qual, const uint8_t* flags,
                          const int32_t* truth_id_per_vcf, int n_bins, unsigned mode, uint8_t* out_cls,
                          uint64_t* out_roc, int64_t* out_scalars, int32_t* out_idx, uint64_t* out_global);

/* ---- resident batches (columns live in HBM across runs) -------------------- */
int qm_batch_create(qm_ctx* ctx, int n_vcf, const int64_t* n_records, const int32_t* truth_id_per_vcf,
                    int n_bins, qm_batch** out);
int qm_batch_create_ext(qm_ctx* ctx, int n_vcf, const int64_t* n_records, const int32_t* truth_id_per_vcf,
                        int n_bins, unsigned mode, qm_batch** out);
void qm_batch_destroy(qm_batch* b);
int qm_batch_upload(qm_batch* b, int vcf, const int32_t* pos, const int32_t* ref, const int32_t* alt,
                    const float* qual, const uint8_t* flags);

#define QM_SYNTH_TRUTH_PER_VCF 0xffffffffffffffffull
typedef struct qm_synth_cfg {
  int64_t genome_len;   /* L: positions 1..L                                  */
  uint64_t seed;        /* VCF v uses seed + v                                 */
  uint64_t truth_seed;  /* must equal the seed passed to qm_truth_synth, or QM_SYNTH_TRUTH_PER_VCF: every VCF is
                           generated against the synthetic truth set it was assigned at qm_batch_create */
  int64_t truth_n;      /* T of that truth set                                  */
  int32_t shuffled;     /* 0 = position sorted, 1 = records permuted, R >= 2 = R ascending runs one behind the other
                           (a VCF of R contigs: run c holds the generated records c, c + R, c + 2 R, ...; needs R <= records) */
  int32_t indel_pct;    /* 0 = single-base records only (configs 3/4); > 0: that share of the
                           generated records / truth entries carries longer alleles (config 5;
                           allele-extended batches only; must equal qm_truth_synth_ext's) */
} qm_synth_cfg;
/* Synthetic workload of BASELINE.json configs 3/4, generated on the device
 * (DESIGN.md "Synthetic generator").  Every VCF of the batch is filled. */
int qm_truth_synth(qm_ctx* ctx, int64_t genome_len, int64_t truth_n, uint64_t truth_seed, int* truth_id);
int qm_truth_synth_ext(qm_ctx* ctx, int64_t genome_len, int64_t truth_n, uint64_t truth_seed, int indel_pct, int* truth_id);
int qm_batch_synth(qm_batch* b, const qm_synth_cfg* cfg);

/* One self-contained synthetic run for harnesses that are not Python: truth set +
 * batch of n_vcf x records_per_vcf generated on the device, one warm-up, `steps` timed runs of the
 * whole path on the context's stream (wall clock around them; per-kernel times from HIP events). */
typedef struct qm_bench_result {
  int64_t records;               /* per step */
  double seconds_per_step;
  double classifications_per_s;
  float classify_ms, finalize_ms, compact_ms;
  int32_t reserved;
  int64_t kept, tp_lines, fp_lines;   /* sums over the batch (same every step) */
  int64_t device_bytes;
} qm_bench_result;
int qm_bench_synth(qm_ctx* ctx, const qm_synth_cfg* cfg, int n_vcf, int64_t records_per_vcf, int n_bins, int steps,
                   qm_bench_result* out);

/* Enqueue the whole path on `stream` (a hipStream_t; NULL = the context's own
 * stream): classify -> finalize (ROC suffix sums, tile offsets, per-truth sums)
 * -> compaction of TP/FP line indices.  Asynchronous.  If global_dev is not
 * NULL it must be a device buffer of [qm_batch_n_truth(b)][3][n_bins] uint64 that
 * receives the per-truth sums (the caller all-reduces it across ranks). */
int qm_batch_run(qm_batch* b, void* stream, void* global_dev);
/* Wait for the stream, then redo VCFs found unsorted through the radix-sort path. */
int qm_batch_finish(qm_batch* b, void* stream);
/* Per-kernel device time, averaged over the (up to 32 latest) qm_batch_run calls made
 * since qm_batch_set_timing(b, 1); HIP events recorded on the run's stream:
 * ms[0]=classify ms[1]=finalize ms[2]=compact ms[3]=whole run. */
int qm_batch_set_timing(qm_batch* b, int on);
int qm_batch_timings(qm_batch* b, float* ms4);

/* qm_batch_upload from page-locked host memory, asynchronous on `stream` (NULL = the context's own). */
int qm_batch_upload_async(qm_batch* b, int vcf, const int32_t* pos, const int32_t* ref, const int32_t* alt,
                          const float* qual, const uint8_t* flags, void* stream);
int qm_batch_get_cls(qm_batch* b, int vcf, uint8_t* out_cls);
/* the class masks as they sit in HBM: bit r of word r / 64 = record r; (n + 63) / 64 words per mask */
int qm_batch_get_masks(qm_batch* b, int vcf, uint64_t* kept, uint64_t* tp);
int qm_batch_get_idx(qm_batch* b, int vcf, int32_t* out_idx);
int qm_batch_get_roc(qm_batch* b, uint64_t* out_roc /*[n_vcf][3][n_bins]*/);
int qm_batch_get_scalars(qm_batch* b, int64_t* out /*[n_vcf][QM_N_SCALARS]*/);
int qm_batch_get_global(qm_batch* b, uint64_t* out /*[qm_batch_n_truth(b)][3][n_bins]*/);
int qm_batch_get_columns(qm_batch* b, int vcf, int32_t* pos, int32_t* ref, int32_t* alt, float* qual,
                         uint8_t* flags);
/* Where the VCFs that the last qm_batch_finish found out of order went (a sorted batch reports zeros).  The bucket path
 * has capacity limits (a bucket's records, the truth keys of its positions, the VCF's size); a VCF beyond them is redone by
 * the radix sort -- correct, several times slower -- and these counters say how often that happened. */
enum {
  QM_PATH_UNSORTED = 0,              /* VCFs found out of order */
  QM_PATH_DIRECT = 1,                /* ... joined bucket by bucket with two bits per position in LDS (k_join_lean) */
  QM_PATH_HASHED = 2,                /* ... joined bucket by bucket through hashed tables (k_classify_hash: wide key ranges) */
  QM_PATH_RADIX = 3,                 /* ... radix-sorted because the bucket path does not take them (size, allele-extended batch) */
  QM_PATH_RADIX_AFTER_OVERFLOW = 4,  /* ... radix-sorted after a bucket of their chunk overflowed */
  QM_PATH_BUCKET_CHUNKS = 5, QM_PATH_OVERFLOW_CHUNKS = 6, QM_PATH_RADIX_CHUNKS = 7,
  QM_PATH_DIRECT2 = 8,               /* ... too large for 256 buckets: dealt to partitions of 2^27 keys by a first scatter (two levels), then as QM_PATH_DIRECT */
  QM_PATH_PARTITIONS = 9,            /* ... too large or too wide for 256 buckets: every partition (256 buckets of 2^15 positions, or of 2^17: up to 32 768 records each) a segment of ONE scatter that reads the columns */
  QM_N_PATH_STATS = 10
};
int qm_batch_path_stats(qm_batch* b, int64_t* out /*[QM_N_PATH_STATS]*/);
/* The same counters summed over every qm_batch_finish of every batch of the context since qm_init, the batches that
 * qm_extract_files / qm_classify_batch make for themselves included: callers take the difference around a call. */
int qm_path_stats_total(qm_ctx* ctx, int64_t* out /*[QM_N_PATH_STATS]*/);
/* Device address of the per-truth sums of the last run ([qm_batch_n_truth(b)][3][n_bins] uint64; the caller's global_dev when
 * qm_batch_run was given one): valid until the batch runs again or is destroyed.  For callers that hand the counters to a
 * collective without a trip through the host. */
int qm_batch_global_device(qm_batch* b, void** dev);
/* ---- the path's one exchange: the all-reduce of the confusion counters (SURVEY.md 8b's all-reduced `out_global`, 8e) ----
 * Replaces nothing in the reference (its per-VCF processes never add anything up: R does, from files); it is what makes the
 * sums of a batch sharded over GPUs one number.  RCCL directly -- ncclCommInitAll / ncclCommInitRank, ONE
 * ncclAllReduce(ncclUint64, ncclSum) in place on the batch's per-truth sums ([qm_batch_n_truth][3][n_bins] uint64: the buffer
 * qm_batch_global_device names) -- so that a host that is not Python has the collective too (the Python host may keep
 * torch.distributed on the same buffer).  librccl is opened when a communicator is first asked for (QM_RCCL_LIB names another
 * file); a build or a box without it fails there with QM_E_COMM and nowhere else.  Unmeasured beyond one device (no multi-GPU
 * box was available to any round; two ranks on one card are refused by RCCL).
 *
 * qm_comm_create: one process, n contexts on n DISTINCT devices (one host thread per context: examples/qm_multi.c).
 * qm_comm_create_rank: one process per GPU; rank 0 makes the id with qm_comm_make_id and hands its 128 bytes to the others
 *   (a file, the environment, MPI: the caller's).
 * qm_allreduce_counters: every member calls it once per step, behind qm_batch_finish; the all-reduce is enqueued on `stream`
 *   (NULL = the context's own) behind the batch's run, and the call returns when THIS member's copy of the sums is complete
 *   (qm_batch_get_global / qm_batch_global_device then hold the sums over all members).  Batches of all members must have the same
 *   n_truth and n_bins.  From one process, call it from one thread per member (a member blocks until all have arrived).
 * qm_comm_collectives: collectives this communicator has issued for `ctx`'s member (tests: exactly one per step). */
typedef struct qm_comm qm_comm;
typedef struct qm_comm_id { char bytes[128]; } qm_comm_id;
int qm_comm_create(qm_ctx* const* ctxs, int n, qm_comm** out);
int qm_comm_make_id(qm_comm_id* out);
int qm_comm_create_rank(qm_ctx* ctx, int rank, int n_ranks, const qm_comm_id* id, qm_comm** out);
int qm_allreduce_counters(qm_batch* b, qm_comm* comm, void* stream);
int64_t qm_comm_collectives(const qm_comm* comm, const qm_ctx* ctx);
void qm_comm_destroy(qm_comm* comm);

/* Bytes the engine holds in HBM for this batch. */
int64_t qm_batch_device_bytes(qm_batch* b);
/* Rows of the per-truth sums ([n][3][n_bins]; qm_batch_get_global, qm_batch_run's global_dev): the number of
 * truth-set slots the context had when the batch was created.  Truth sets loaded later do not change it. */
int qm_batch_n_truth(qm_batch* b);

/* What this GPU streams with 16-byte accesses per lane over `bytes` of HBM (>= 1 MiB; use a size far beyond the 256 MiB
 * Infinity Cache), `reps` passes each: gbps[0] = read only, gbps[1] = copy (bytes read + bytes written per second),
 * gbps[2] = write only.  The measured denominators bench.py quotes beside the data sheet's 8 TB/s. */
int qm_bw_probe(qm_ctx* ctx, int64_t bytes, int reps, double* gbps /*[3]*/);

/* ---- FP overlap (rules/compare_FP.smk + scripts/snpcaller_fp_compare.R:36-47) -
 * n_sets (<= 5) key lists (fp.vcf rows as packed columns); regions[m] = number
 * of distinct keys whose membership mask is m.  regions has 1 << n_sets slots. */
int qm_fp_overlap(qm_ctx* ctx, int n_sets, const int64_t* set_offsets, const int32_t* pos,
                  const int32_t* ref, const int32_t* alt, int64_t* regions);

/* ---- host text side (no GPU): tokenizer / packer / writers ------------------
 * Replaces the three awk passes + `grep -E "^#"` per VCF
 * (extract_TP_FP_SNPs.py:24-32,50,52) with one scan.  qm_vcf_scan fills, for
 * every line of the text (header lines included), its byte offset; for data
 * lines the packed columns.  Returns QM_OK or a negative code.  line_kind: */
#define QM_LINE_DATA 0          /* data line, described completely by its columns */
#define QM_LINE_HEADER 1        /* begins with '#' */
#define QM_LINE_DATA_HOST 2     /* single-base data line whose fgrep answer the columns cannot give (POS not a
                                   canonical decimal, or a "\t.\t" that a pattern could sit on after the ALT column):
                                   qm_vcf_hostpath decides it and writes the decision into its flags */
#define QM_LINE_HEADER_KEPT 3   /* '#' line that also satisfies the A2 filter: awk does not skip it, so the reference
                                   emits it in the header block AND among the kept lines (and then in tp or fp) */
#define QM_LINE_HEADER_KEPT_TP 4 /* the same once qm_vcf_hostpath found it selected by fgrep -wf */
#define QM_LINE_REFUSED 5       /* kept data line holding a NUL or an INVALID UTF-8 sequence: grep answers "binary file matches" under the
                                   locale Python exports to it (PEP 538); strict callers stop with QM_E_NONCANON.  Valid UTF-8 is text: such
                                   a single-base line is QM_LINE_DATA_HOST (word characters by iswalnum on C.UTF-8, as grep -w asks) */
#define QM_LINE_HEADER_REFUSED 6 /* the same for a '#' line that satisfies the A2 filter */
typedef struct qm_vcf_cols {
  int64_t n_lines;      /* all lines */
  int64_t n_data;       /* data lines = records */
  int64_t n_host;       /* lines for qm_vcf_hostpath (kinds 2 and 3) */
  int64_t n_refused;    /* kinds 5 and 6 */
  int64_t first_refused_line; /* 1-based, 0 = none */
  int64_t n_nokey_kept; /* kept data lines without a comparable key (QM_F_NOKEY): qm_vcf_hostpath counts their keys as text */
  /* (ABI 4) kept data lines that hold '#', ' or ": the reference's counting step reads <x>.filtered.vcf with R's read.table
   * (scripts/caller_performance_compare.R:29-39, custom_snp_benchmark.R:45-48: comment.char = "#", quote = "\"'"), which cuts a
   * line at a '#', lets a quote swallow tabs and newlines up to the next one, and turns a file it then cannot parse into an
   * EMPTY one (tryCatch -> NA row).  The three output files are unaffected (they are the shell pipeline's bytes); the R-path
   * counts (QM_S_TP_R / FP_R / NPASS) assume lines split at their tabs alone.  Strict table writers refuse such a file. */
  int64_t n_r_hostile;
  int64_t first_r_hostile_line; /* 1-based, 0 = none */
} qm_vcf_cols;
typedef struct qm_dict qm_dict;
int64_t qm_vcf_count_lines(const uint8_t* text, size_t len);
int qm_vcf_scan(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off /*cap+1*/,
                uint8_t* line_kind, int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags,
                qm_vcf_cols* info);
/* Truth text -> key columns.  mode 0 = VCF as written by mummer2vcf.py
 * (columns 2,4,5), mode 1 = 12-column show-snps TSV (columns 1,2,3).
 * out_counts[0] = rows R counts as `genomediff`, [1] = keys emitted (rows that are not comments, with a
 * canonical position and single-base alleles: the only patterns a line can match through its columns),
 * [2] = rows whose pattern can match no such line (they live in qm_patterns only), [3] = rows refused
 * (always 0 since round 6: a canonical pattern is ASCII, the row's other columns never reach it), [4] = '#' rows that awk turns into a pattern all the same. */
int64_t qm_truth_scan(const uint8_t* text, size_t len, int mode, int64_t cap, int32_t* pos, int32_t* ref,
                      int32_t* alt, int64_t* out_counts /*[5]*/);
/* Allele-extended tokenising (QM_BATCH_ALLELES): the filter's `^[ACGT]$` becomes `^[ACGT]+$`, ref / alt
 * carry allele codes, alleles longer than QM_ALLELE_INLINE_MAX bases are interned in `dict`, which the
 * truth set and every VCF of a batch must share.  dict == NULL is qm_vcf_scan / qm_truth_scan.
 * qm_truth_scan_ext takes mode 0 (VCF) only.  qm_dict is thread safe. */
qm_dict* qm_dict_create(void);
void qm_dict_destroy(qm_dict* d);
int64_t qm_dict_size(qm_dict* d);
/* code of an allele string; QM_ALLELE_NONE unless it is [ACGT]+ (and, beyond 13 bases, d != NULL) */
int32_t qm_allele_code(qm_dict* d, const uint8_t* s, size_t n);
/* spelling of a code into out[cap]; returns its length, -1 if the code is no allele / does not fit */
int64_t qm_allele_spell(qm_dict* d, int32_t code, uint8_t* out, size_t cap);
int qm_vcf_scan_ext(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off, uint8_t* line_kind,
                    int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_vcf_cols* info,
                    qm_dict* dict);
int64_t qm_truth_scan_ext(const uint8_t* text, size_t len, int mode, int64_t cap, int32_t* pos, int32_t* ref,
                          int32_t* alt, int64_t* out_counts /*[5]*/, qm_dict* dict);

/* ---- host path for what the columns cannot describe -----------
 * qm_patterns is the pattern list the reference feeds to `fgrep -wf` (extract_TP_FP_SNPs.py:47-53, :92-98), kept as
 * TEXT: one pattern X \t . \t Y \t Z per truth row that satisfies the awk program, '#' rows included.  mode as
 * qm_truth_scan; ext != 0 widens `^[ACGT]$` to `^[ACGT]+$` (allele-extended mode, mode 0 only).
 * info[0] = distinct patterns, [1] = patterns whose Y / Z are not one character each, [2] = canonical keys that only
 * '#' rows carry (fgrep sees them, R does not), [3] = rows refused (a NUL or an invalid UTF-8 sequence inside the pattern's fields).  When [1] or [2] is non-zero the
 * columns alone cannot reproduce the reference for ANY line compared with this truth set: qm_vcf_hostpath then
 * decides every single-base data line from the text. */
typedef struct qm_patterns qm_patterns;
qm_patterns* qm_patterns_create(const uint8_t* truth_text, size_t len, int mode, int ext);
void qm_patterns_destroy(qm_patterns* p);
int qm_patterns_info(const qm_patterns* p, int64_t* info /*[4]*/);
/* Exact `fgrep -w` (GNU grep 3.7: a pattern occurrence with a non-word character or the line edge on both sides) for
 * the lines of one scanned VCF that need it: kind QM_LINE_DATA_HOST lines get QM_F_TPLINE set or QM_F_IDDOT cleared
 * in `flags` (indexed by data line), QM_LINE_HEADER_KEPT lines become QM_LINE_HEADER_KEPT_TP when selected.  Run it
 * between qm_vcf_scan and the upload of the columns; the device then counts and lists these lines like any other.
 * R's unique-key counts (caller_performance_compare.R:84-99) take the key of a line as TEXT; for kept lines without
 * a comparable key the device counts distinct (carried pos, ref, alt) instead, so out[2..4] carry the exchange:
 * out[0] = lines decided here, [1] = of them selected, [2] = what the device will add to FP_R for the QM_F_NOKEY
 * lines, [3] = their distinct text keys found in the truth file as R reads it (add to TP_R), [4] = the other
 * distinct text keys (add to FP_R after subtracting [2]).  pos / ref / alt: the scan's columns. */
int qm_vcf_hostpath(const qm_patterns* p, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                    uint8_t* line_kind, const int32_t* pos, const int32_t* ref, const int32_t* alt, uint8_t* flags,
                    int64_t* out /*[5]*/);

/* Writes header lines + selected data lines, verbatim, newline-terminated
 * (SURVEY Q7).  select: 0 = kept (filtered.vcf), 1 = TP, 2 = FP.  QM_LINE_HEADER_KEPT(_TP) lines appear in the
 * header block and again, in input order, among the selected lines.  Atomic (temp file + rename). */
int qm_vcf_write(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                 const uint8_t* line_kind, const uint8_t* cls /*per data line*/, int select);
/* SNP / indel splitters of rules/vis_eval_vcf.smk:35,50,66,81 (`extract_snp`, `extract_indel`,
 * `extract_nucmer_snp`, `extract_nucmer_indel`): every '#' line, plus every line whose REF/ALT
 * satisfy the rule's awk pattern, in input order ('#' lines that also satisfy it appear twice,
 * as awk prints them).  mode 0 = xsnp, 1 = xindel.  flavour 0 reads `{2,}` as a POSIX interval
 * (gawk), flavour 1 as literal text (mawk 1.3.4 20200120).  Atomic (temp file + rename). */
int qm_vcf_split_write(const char* path, const uint8_t* text, size_t len, int mode, int flavour, int64_t* n_written);
/* ---- files in, files out: the reference's per-VCF worker for MANY VCFs in one call ------------------------------
 * What n_jobs invocations of `python program/extract_TP_FP_SNPs.py <vcf> <truth> {hcmv,custom} <outdir> <caller>`
 * (extract_TP_FP_SNPs.py:124-140; rules/extract_TP.smk:20, eval_variant_custom.smk:73) do: inputs are mapped,
 * tokenised by host threads straight into page-locked buffers and uploaded while the next file is tokenised, the
 * host path decides what the columns cannot describe, ONE engine batch classifies every mixed-sample VCF, the class
 * masks come back (2 bits per record) and the three files of every VCF are gathered from the mapped input with
 * writev.  Output paths are the caller's (the reference derives them, :19-22,39-41 / :71-72,91); their directories
 * must exist.  pure != 0: pure-strain sample (:33-36): fp is a copy of filtered, no tp file, the truth is never read.
 * strict != 0: QM_E_NONCANON for kept lines / truth rows holding a NUL or an invalid UTF-8 sequence (their reference answer depends
 * on the locale); 0: such lines are classified by their columns.  mode: 0 or QM_BATCH_ALLELES.
 * stats[j] / roc[j][3][n_bins] (either may be NULL): the per-VCF rows; phase_seconds[8] (may be NULL): map + count,
 * truth sets (on a thread beside the former), batch layout, tokenise + host path + uploads, engine, masks back, write,
 * release. */
typedef struct qm_file_job {
  const char* vcf_path;
  const char* truth_path;   /* may be NULL when pure */
  int32_t mode;             /* 0: truth VCF (hcmv), 1: show-snps table (custom) */
  int32_t pure;
  const char* filtered_out;
  const char* tp_out;       /* may be NULL when pure */
  const char* fp_out;
} qm_file_job;
typedef struct qm_file_stats {
  int64_t scalars[QM_N_SCALARS];   /* QM_S_*; TP_R / FP_R with the text keys of QM_F_NOKEY lines exchanged in */
  int64_t n_lines, n_refused, genomediff;
  int64_t header_kept, header_kept_tp;   /* '#' lines that pass the A2 filter (awk emits them among the kept lines too) */
  int64_t host_decided;                  /* lines decided by the host path */
  int64_t r_hostile;                     /* (ABI 4) qm_vcf_cols.n_r_hostile: kept lines R's read.table would not read as tab-split text */
} qm_file_stats;
int qm_extract_files(qm_ctx* ctx, int n_jobs, const qm_file_job* jobs, int n_bins, unsigned mode, int strict,
                     qm_file_stats* stats, uint64_t* roc, double* phase_seconds);
/* The same for one rank of a multi-GPU run (rules/extract_TP.smk:17-20 runs one process per VCF; here one process per GPU
 * takes a share of them): global_dev, a DEVICE buffer of [n_slots][3][n_bins] uint64, receives the per-truth-file sums of this
 * call's VCFs -- cleared first; VCF j adds to row truth_slot[j] (the caller's layout: every rank uses the same row for the
 * same truth file; pure-strain jobs are ignored; jobs that name the same truth file must name the same row) -- ready for the
 * one all-reduce of the confusion counters (RCCL).  n_jobs == 0 is allowed: the buffer is cleared, nothing else happens.
 * truth_slot / global_dev NULL: qm_extract_files. */
int qm_extract_files_ex(qm_ctx* ctx, int n_jobs, const qm_file_job* jobs, int n_bins, unsigned mode, int strict,
                        qm_file_stats* stats, uint64_t* roc, double* phase_seconds, const int32_t* truth_slot, int n_slots,
                        void* global_dev);

/* `bgzip -c` (the *.vcf.gz outputs the same rules declare, rules/vis_eval_vcf.smk:29,36 ...): BGZF = gzip members of at
 * most 64 KiB with a 'BC' extra field + the EOF member; zcat and tabix / htslib read it.  level -1 = zlib's default (6,
 * bgzip's default).  Atomic. */
int qm_bgzf_write(const char* path, const uint8_t* data, size_t len, int level);

/* `bgzip -c x.vcf > x.vcf.gz && tabix -p vcf x.vcf.gz` (rules/vis_eval_vcf.smk:36-37, 51-52, 67-68, 82-83): <path> as
 * qm_bgzf_write writes it and <path>.tbi, the tabix index of its data lines (sequence = column 1, begin = POS - 1, end =
 * begin + len(REF) or INFO's END=; bins of the UCSC scheme with a 16 kb linear index over BGZF virtual offsets; htslib's
 * pseudo-bin 37450 per sequence; itself BGZF-compressed).  QM_E_UNSORTED when the sequences do not come in blocks or a
 * position steps backwards -- `tabix` stops on such a file as well --, QM_E_RANGE for coordinates beyond 2^29, QM_E_INVAL
 * for a data line without CHROM / POS; nothing is written then.  Atomic per file. */
int qm_bgzf_write_tbi(const char* path, const uint8_t* data, size_t len, int level);

/* ---- truth-set builder ------------------------------------------------------------------------------
 * `mummer2vcf.py -s <table> [--input-header] [-n] [-t SNP|INDEL] [--output-header] -g <ref.fa>` of rules/genome_diff.smk:22-24
 * (program/mummer2vcf.py:69-357): a MUMmer `show-snps -T` table and the reference FASTA in, the truth VCF out (SNVs of one
 * position collapsed into a multi-allelic ALT, runs of insertions / deletions merged and anchored on the base in front of them,
 * rows ordered by contig and position).  Restated from the reference's text -- Biopython is not in the build image, the
 * reference cannot run there: parity unpinned, hand-derived cases only.  table / fasta: the files' bytes; reference_name: what
 * `##reference=` shall say (the path given to -g); file_date: "YYYYMMDD" for `##fileDate=` or NULL for today; *out: the VCF,
 * malloc'ed, to be released with qm_free.  QM_E_INVAL: a row with fewer than 12 columns or a P1 that is no integer, an indel on
 * a contig the FASTA does not hold; QM_E_RANGE: an indel beyond its contig's end. */
#define QM_M2V_NO_NS 1u          /* -n */
#define QM_M2V_OUTPUT_HEADER 2u  /* --output-header */
#define QM_M2V_INPUT_HEADER 4u   /* --input-header: the table's first four lines are its header */
#define QM_M2V_ONLY_SNP 8u       /* -t SNP */
#define QM_M2V_ONLY_INDEL 16u    /* -t INDEL */
int qm_mummer2vcf(const uint8_t* table, size_t table_len, const uint8_t* fasta, size_t fasta_len, const char* reference_name,
                  unsigned flags, const char* file_date, uint8_t** out, size_t* out_len);
void qm_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* QMVT_H */
