/* examples/qm_bench.c -- the engine from plain C, through include/qmvt.h only.
 *
 *   gcc -O2 -Iinclude -o qm_bench examples/qm_bench.c -Lquasimodo_amd/csrc -lqmvt -Wl,-rpath,$PWD/quasimodo_amd/csrc
 *   ./qm_bench [n_vcf] [records_per_vcf] [steps] [shuffled] [indel_pct]
 *
 * Runs BASELINE.json's synthetic workload (config 3 shape by default) inside the library and
 * prints classifications/s and the per-kernel times.  Without a usable HIP device qm_init fails
 * with QM_E_NODEVICE and the program exits with status 2: there is no CPU path to fall back to. */
#include <stdio.h>
#include <stdlib.h>

#include "qmvt.h"

int main(int argc, char** argv) {
  const int n_vcf = argc > 1 ? atoi(argv[1]) : 64;
  const long long records = argc > 2 ? atoll(argv[2]) : 1000000;
  const int steps = argc > 3 ? atoi(argv[3]) : 5;
  const int shuffled = argc > 4 ? atoi(argv[4]) : 0;
  const int indel_pct = argc > 5 ? atoi(argv[5]) : 0;
  if (qm_abi_version() != QM_ABI_VERSION) {
    fprintf(stderr, "header / library ABI mismatch: %d vs %d\n", QM_ABI_VERSION, qm_abi_version());
    return 3;
  }
  qm_ctx* ctx = NULL;
  int rc = qm_init(0, &ctx);
  if (rc != QM_OK) {
    fprintf(stderr, "qm_init failed (%d): %s\n", rc, qm_last_error(NULL));
    return rc == QM_E_NODEVICE ? 2 : 1;
  }
  qm_synth_cfg cfg;
  cfg.genome_len = 5000000;
  cfg.seed = indel_pct ? 5000 : 3000;
  cfg.truth_seed = indel_pct ? 5 : 3;
  cfg.truth_n = 100000;
  cfg.shuffled = shuffled;
  cfg.indel_pct = indel_pct;
  qm_bench_result r;
  rc = qm_bench_synth(ctx, &cfg, n_vcf, records, 256, steps, &r);
  if (rc != QM_OK) {
    fprintf(stderr, "qm_bench_synth failed (%d): %s\n", rc, qm_last_error(ctx));
    qm_destroy(ctx);
    return 1;
  }
  printf("{\"vcfs\": %d, \"records_per_vcf\": %lld, \"steps\": %d, \"classifications_per_s\": %.6g, \"ms_per_step\": %.4f, "
         "\"classify_ms\": %.4f, \"finalize_ms\": %.4f, \"compact_ms\": %.4f, \"kept\": %lld, \"tp_lines\": %lld, \"fp_lines\": %lld, "
         "\"device_bytes\": %lld}\n",
         n_vcf, records, steps, r.classifications_per_s, r.seconds_per_step * 1e3, r.classify_ms, r.finalize_ms, r.compact_ms,
         (long long)r.kept, (long long)r.tp_lines, (long long)r.fp_lines, (long long)r.device_bytes);
  qm_destroy(ctx);
  return 0;
}
