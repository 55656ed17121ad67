/* examples/qm_multi.c -- several GPUs from ONE plain-C process, through include/qmvt.h only.
 *
 *   gcc -O2 -pthread -Iinclude -o qm_multi examples/qm_multi.c -Lquasimodo_amd/csrc -lqmvt -Wl,-rpath,$PWD/quasimodo_amd/csrc
 *   ./qm_multi <devices, e.g. 0,1,2,3> [n_vcf] [records_per_vcf] [steps] [check]
 *
 * The path shards by VCF with no exchange between the shards: one host thread and one qm_ctx per device, each
 * with the truth set and a contiguous range of the synthetic VCFs (BASELINE config 3 shape; VCF v has seed 3000 + v wherever
 * it lands).  The only thing that crosses devices is the sum of the [n_truth][3][n_bins] confusion counters: ONE RCCL
 * all-reduce per step through the library's own collective (qm_comm_create over the contexts, qm_allreduce_counters from
 * every shard's thread, in place on the batch's device buffer) -- what a Python host does with torch.distributed
 * (quasimodo_amd/multigpu.py, bench.py).  A device may be named more than once ("0,0": two contexts on one card) to rehearse
 * the threading without a second GPU; RCCL refuses two ranks on one card, so such a run adds the vectors up on the host
 * and says so ("exchange": "host-sum").
 * check = 1: the same VCFs once more in one batch on the first device; the summed counters must be identical.
 * Exit status 2 without a usable HIP device: there is no CPU path. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qmvt.h"

enum { N_BINS = 256, MAX_DEV = 16 };

typedef struct shard {
  int device, v0, n_vcf, steps;
  qm_ctx* ctx;                    /* made by main: the communicator wants all of them */
  qm_comm* comm;                  /* NULL: host sum */
  long long collectives;
  long long records;
  int rc;
  char err[256];
  double seconds;                 /* wall clock of the timed steps on this shard's thread */
  uint64_t glob[3 * N_BINS];      /* per-truth sums of the shard (one truth set) */
  long long kept, tp, fp;
} shard;

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static pthread_barrier_t g_start, g_stop;

static void* run_shard(void* arg) {
  shard* s = (shard*)arg;
  qm_ctx* ctx = s->ctx;
  qm_batch* b = NULL;
  int64_t* nrec = NULL;
  int32_t* tids = NULL;
  int64_t* scal = NULL;
  int tid = -1, armed = 0;
  const int own_ctx = ctx == NULL;   /* the check run brings none: a fresh context (one truth set, so one row of sums) */
  if (own_ctx) {
    s->rc = qm_init(s->device, &ctx);
    if (s->rc != QM_OK) { snprintf(s->err, sizeof s->err, "qm_init(%d): %s", s->device, qm_last_error(NULL)); goto out; }
  }
  s->rc = qm_truth_synth(ctx, 5000000, 100000, 3, &tid);
  if (s->rc != QM_OK) goto fail;
  nrec = (int64_t*)malloc(sizeof(int64_t) * (size_t)s->n_vcf);
  tids = (int32_t*)malloc(sizeof(int32_t) * (size_t)s->n_vcf);
  scal = (int64_t*)malloc(sizeof(int64_t) * (size_t)s->n_vcf * QM_N_SCALARS);
  if (!nrec || !tids || !scal) { s->rc = QM_E_NOMEM; snprintf(s->err, sizeof s->err, "out of host memory"); goto out; }
  for (int v = 0; v < s->n_vcf; ++v) { nrec[v] = s->records; tids[v] = tid; }
  s->rc = qm_batch_create(ctx, s->n_vcf, nrec, tids, N_BINS, &b);
  if (s->rc != QM_OK) goto fail;
  {
    qm_synth_cfg cfg;
    cfg.genome_len = 5000000; cfg.seed = 3000u + (uint64_t)s->v0; cfg.truth_seed = 3; cfg.truth_n = 100000; cfg.shuffled = 0; cfg.indel_pct = 0;
    s->rc = qm_batch_synth(b, &cfg);
    if (s->rc != QM_OK) goto fail;
  }
  s->rc = qm_batch_run(b, NULL, NULL);                 /* warm-up */
  if (s->rc == QM_OK) s->rc = qm_batch_finish(b, NULL);
  if (s->rc != QM_OK) goto fail;
  armed = 1;
  pthread_barrier_wait(&g_start);                      /* all shards start their timed steps together */
  {
    const double t0 = now();
    for (int k = 0; k < s->steps && s->rc == QM_OK; ++k) {
      s->rc = qm_batch_run(b, NULL, NULL);
      if (s->rc == QM_OK) s->rc = qm_batch_finish(b, NULL);
      if (s->rc == QM_OK && s->comm) s->rc = qm_allreduce_counters(b, s->comm, NULL);   /* the step's one collective */
    }
    s->seconds = now() - t0;
  }
  pthread_barrier_wait(&g_stop);
  if (s->rc != QM_OK) goto fail;
  if (s->comm) s->collectives = (long long)qm_comm_collectives(s->comm, ctx);
  s->rc = qm_batch_get_global(b, s->glob);            /* with a communicator: the sums over ALL shards, on every shard */
  if (s->rc == QM_OK) s->rc = qm_batch_get_scalars(b, scal);
  if (s->rc != QM_OK) goto fail;
  for (int v = 0; v < s->n_vcf; ++v) { s->kept += scal[v * QM_N_SCALARS + QM_S_NPASS]; s->tp += scal[v * QM_N_SCALARS + QM_S_TP_LINES]; s->fp += scal[v * QM_N_SCALARS + QM_S_FP_LINES]; }
  goto out;
fail:
  snprintf(s->err, sizeof s->err, "%s", qm_last_error(ctx));
out:
  if (!armed) { pthread_barrier_wait(&g_start); pthread_barrier_wait(&g_stop); }   /* nobody waits for a shard that failed early */
  if (b) qm_batch_destroy(b);
  if (own_ctx && ctx) qm_destroy(ctx);
  free(nrec); free(tids); free(scal);
  return NULL;
}

int main(int argc, char** argv) {
  int dev[MAX_DEV], n_dev = 0;
  const char* list = argc > 1 ? argv[1] : "0";
  const int n_vcf = argc > 2 ? atoi(argv[2]) : 64;
  const long long records = argc > 3 ? atoll(argv[3]) : 1000000;
  const int steps = argc > 4 ? atoi(argv[4]) : 5;
  const int check = argc > 5 ? atoi(argv[5]) : 0;
  for (const char* p = list; *p && n_dev < MAX_DEV;) {
    dev[n_dev++] = (int)strtol(p, (char**)&p, 10);
    if (*p == ',') ++p; else if (*p) { fprintf(stderr, "bad device list %s\n", list); return 1; }
  }
  if (qm_abi_version() != QM_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 3; }
  if (n_dev < 1 || n_vcf < n_dev || records < 1 || steps < 1) { fprintf(stderr, "usage: qm_multi <devices> [n_vcf >= devices] [records] [steps] [check]\n"); return 1; }
  shard sh[MAX_DEV + 1];
  pthread_t th[MAX_DEV];
  memset(sh, 0, sizeof sh);
  pthread_barrier_init(&g_start, NULL, (unsigned)n_dev);
  pthread_barrier_init(&g_stop, NULL, (unsigned)n_dev);
  /* one context per device, then the communicator over them (distinct devices only) */
  qm_ctx* ctxs[MAX_DEV];
  qm_comm* comm = NULL;
  int distinct = 1;
  for (int r = 0; r < n_dev; ++r) for (int q = 0; q < r; ++q) if (dev[q] == dev[r]) distinct = 0;
  for (int r = 0; r < n_dev; ++r) {
    const int rc = qm_init(dev[r], &ctxs[r]);
    if (rc != QM_OK) { fprintf(stderr, "qm_init(%d) failed (%d): %s\n", dev[r], rc, qm_last_error(NULL)); return rc == QM_E_NODEVICE ? 2 : 1; }
  }
  if (distinct) {
    const int rc = qm_comm_create(ctxs, n_dev, &comm);
    if (rc != QM_OK) { fprintf(stderr, "qm_comm_create failed (%d): %s\n", rc, qm_last_error(NULL)); return 5; }
  }
  for (int r = 0, v0 = 0; r < n_dev; ++r) {   /* equal VCFs: contiguous, near-equal ranges (unequal ones: longest-processing-time first, as quasimodo_amd/sharding.py) */
    const int n = n_vcf / n_dev + (r < n_vcf % n_dev ? 1 : 0);
    sh[r].ctx = ctxs[r]; sh[r].comm = comm;
    sh[r].device = dev[r]; sh[r].v0 = v0; sh[r].n_vcf = n; sh[r].records = records; sh[r].steps = steps;
    v0 += n;
    pthread_create(&th[r], NULL, run_shard, &sh[r]);
  }
  uint64_t glob[3 * N_BINS] = {0};
  long long kept = 0, tp = 0, fp = 0;
  double seconds = 0;
  int bad = 0;
  for (int r = 0; r < n_dev; ++r) {
    pthread_join(th[r], NULL);
    if (sh[r].rc != QM_OK) { fprintf(stderr, "shard %d on device %d failed (%d): %s\n", r, sh[r].device, sh[r].rc, sh[r].err); bad = sh[r].rc == QM_E_NODEVICE ? 2 : 1; }
    if (!comm) for (int i = 0; i < 3 * N_BINS; ++i) glob[i] += sh[r].glob[i];      /* (a device named twice: the exchange on the host) */
    else if (r == 0) memcpy(glob, sh[0].glob, sizeof glob);                          /* all-reduced on the devices: every shard holds the sums */
    else if (memcmp(glob, sh[r].glob, sizeof glob) != 0 || sh[r].collectives != steps) { fprintf(stderr, "shard %d disagrees after the all-reduce\n", r); bad = 1; }
    kept += sh[r].kept; tp += sh[r].tp; fp += sh[r].fp;
    if (sh[r].seconds > seconds) seconds = sh[r].seconds;                /* the slowest shard is the job */
  }
  if (bad) return bad;
  int same = -1;
  if (check) {   /* the same VCFs in one batch on the first device */
    pthread_barrier_destroy(&g_start); pthread_barrier_destroy(&g_stop);
    pthread_barrier_init(&g_start, NULL, 1); pthread_barrier_init(&g_stop, NULL, 1);
    shard* one = &sh[MAX_DEV];
    one->ctx = NULL; one->comm = NULL;
    one->device = dev[0]; one->v0 = 0; one->n_vcf = n_vcf; one->records = records; one->steps = 1;
    run_shard(one);
    if (one->rc != QM_OK) { fprintf(stderr, "reference run failed (%d): %s\n", one->rc, one->err); return 1; }
    same = memcmp(one->glob, glob, sizeof glob) == 0 && one->kept == kept && one->tp == tp && one->fp == fp;
  }
  if (comm) qm_comm_destroy(comm);
  for (int r = 0; r < n_dev; ++r) qm_destroy(ctxs[r]);
  printf("{\"devices\": %d, \"exchange\": \"%s\", \"vcfs\": %d, \"records_per_vcf\": %lld, \"steps\": %d, \"classifications_per_s\": %.6g, \"ms_per_step\": %.4f, "
         "\"kept\": %lld, \"tp_lines\": %lld, \"fp_lines\": %lld, \"roc_tp_at_20\": %llu, \"equals_one_batch\": %s}\n",
         n_dev, distinct ? "rccl all-reduce (qm_allreduce_counters), one per step" : "host-sum", n_vcf, records, steps, (double)n_vcf * (double)records * steps / seconds, seconds / steps * 1e3, kept, tp, fp,
         (unsigned long long)glob[20], same < 0 ? "null" : same ? "true" : "false");
  return same == 0 ? 4 : 0;
}
