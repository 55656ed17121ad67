/* Sanitizer harness for the oracle's text functions (oracle/qm_oracle.c) -- the checker deserves checking too.
 * usage: oracle_asan <truth> <vcf>...   (every VCF against the truth file, plus mutations) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint8_t* data; size_t len; size_t cap; } qmo_buf;
int qmo_extract_text(const uint8_t* vcf, size_t vlen, const uint8_t* truth, size_t tlen, int custom, int pure, qmo_buf* f, qmo_buf* t,
                     qmo_buf* p, int64_t* st);
int qmo_count_text(const uint8_t* filtered, size_t flen, const uint8_t* truth, size_t tlen, int custom, int64_t* out);
void qmo_buf_free(qmo_buf* b);

static uint8_t* slurp(const char* p, size_t* n) {
  FILE* f = fopen(p, "rb");
  if (!f) { *n = 0; return (uint8_t*)malloc(1); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz > (2 << 20)) sz = 2 << 20;
  uint8_t* b = (uint8_t*)malloc(sz ? (size_t)sz : 1);
  *n = fread(b, 1, (size_t)sz, f);
  fclose(f);
  return b;
}

static long calls = 0;
static void run(const uint8_t* v, size_t vn, const uint8_t* t, size_t tn) {
  uint8_t* vv = (uint8_t*)malloc(vn ? vn : 1);
  uint8_t* tt = (uint8_t*)malloc(tn ? tn : 1);
  memcpy(vv, v, vn); memcpy(tt, t, tn);
  for (int custom = 0; custom < 2; ++custom)
    for (int pure = 0; pure < 2; ++pure) {
      qmo_buf f = {0, 0, 0}, tp = {0, 0, 0}, fp = {0, 0, 0};
      int64_t st[6], cnt[6];
      qmo_extract_text(vv, vn, tt, tn, custom, pure, &f, &tp, &fp, st);
      qmo_count_text(f.data ? f.data : vv, f.data ? f.len : 0, tt, tn, custom, cnt);
      qmo_buf_free(&f); qmo_buf_free(&tp); qmo_buf_free(&fp);
      ++calls;
    }
  free(vv); free(tt);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  size_t tn;
  uint8_t* t = slurp(argv[1], &tn);
  unsigned rng = 777;
  static const uint8_t alphabet[] = {'\t', '\n', '\r', '.', 'A', '9', '#', ',', ' ', 'e', '-', '+', 'x'};
  for (int a = 2; a < argc; ++a) {
    size_t vn;
    uint8_t* v = slurp(argv[a], &vn);
    run(v, vn, t, tn);
    for (int k = 0; k < 12 && vn; ++k) {
      rng = rng * 1664525u + 1013904223u;
      run(v, rng % (vn + 1), t, (rng >> 7) % (tn + 1));
      uint8_t* w = (uint8_t*)malloc(vn);
      memcpy(w, v, vn);
      for (int j = 0; j < 16; ++j) { rng = rng * 1664525u + 1013904223u; w[(rng >> 8) % vn] = alphabet[rng % sizeof alphabet]; }
      run(w, vn, t, tn);
      free(w);
    }
    free(v);
  }
  free(t);
  printf("oracle sanitizer run: %ld calls, clean\n", calls);
  return 0;
}
