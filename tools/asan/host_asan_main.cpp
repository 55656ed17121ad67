// Sanitizer harness for the host text side (qmvt_host.cpp): every host entry point of include/qmvt.h on the files
// given on the command line, plus truncations and byte flips of them.  Build + run: bash tools/asan/run.sh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qmvt.h"

static std::vector<uint8_t> slurp(const char* p) {
  std::vector<uint8_t> v;
  FILE* f = fopen(p, "rb");
  if (!f) return v;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

static long g_calls = 0;

// internal to the library (qmvt_pipeline.cpp calls it): the writer from the device's class masks
int qm_host_write_masks(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                        const uint8_t* line_kind, const uint64_t* kept, const uint64_t* tp, const uint8_t* flags, int select);
static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return v;
  uint8_t b[65536]; size_t n;
  while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n);
  fclose(f);
  return v;
}

static void exercise(const std::vector<uint8_t>& t, qm_dict* dict, const char* outdir) {
  // exact-size heap copy: any read past the end trips the sanitizer
  uint8_t* text = (uint8_t*)malloc(t.size() ? t.size() : 1);
  if (t.size()) memcpy(text, t.data(), t.size());
  const size_t len = t.size();
  const int64_t cap = qm_vcf_count_lines(text, len) + 1;
  std::vector<int64_t> off((size_t)cap + 1);
  std::vector<uint8_t> kind((size_t)cap), flags((size_t)cap), cls((size_t)cap, 3);
  std::vector<int32_t> pos((size_t)cap), ref((size_t)cap), alt((size_t)cap);
  std::vector<float> qual((size_t)cap);
  for (int ext = 0; ext < 2; ++ext) {
    qm_vcf_cols info;
    int rc = qm_vcf_scan_ext(text, len, cap, off.data(), kind.data(), pos.data(), ref.data(), alt.data(), qual.data(), flags.data(), &info,
                             ext ? dict : nullptr);
    if (rc != QM_OK) { fprintf(stderr, "scan rc=%d\n", rc); exit(1); }
    for (int sel = 0; sel < 3; ++sel) {
      const std::string o = std::string(outdir) + "/w.vcf";
      qm_vcf_write(o.c_str(), text, len, info.n_lines, off.data(), kind.data(), cls.data(), sel);
    }
    {
      // the writer from class MASKS (64 lines at a time where every '#' line precedes the data) against the line-by-line writer
      // from class BYTES, on a pseudo-random classification of this file's records: the same bytes, all three files
      std::vector<uint8_t> c2((size_t)cap, 0);
      std::vector<uint64_t> kept((size_t)(info.n_data + 63) / 64 + 1, 0), tpm(kept.size(), 0);
      uint64_t x = 0x9e3779b97f4a7c15ull + (uint64_t)len;
      for (int64_t r = 0; r < info.n_data; ++r) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const unsigned m = (unsigned)(x >> 33) % 100;
        const uint8_t c = m < 10 ? 0 : m < 25 ? 3 : m < 30 ? 2 : 1;   // (2: the TP bit without the kept bit selects nothing)
        c2[(size_t)r] = c;
        if (c & 1) kept[(size_t)(r >> 6)] |= 1ull << (r & 63);
        if (c & 2) tpm[(size_t)(r >> 6)] |= 1ull << (r & 63);
      }
      if (len % 3 == 0) for (int64_t r = 0; r < info.n_data; ++r) { c2[(size_t)r] = 3; kept[(size_t)(r >> 6)] |= 1ull << (r & 63); tpm[(size_t)(r >> 6)] |= 1ull << (r & 63); }   // long runs too
      for (int sel = 0; sel < 3; ++sel) {
        const std::string a = std::string(outdir) + "/wa.vcf", b = std::string(outdir) + "/wb.vcf";
        const int ra = qm_vcf_write(a.c_str(), text, len, info.n_lines, off.data(), kind.data(), c2.data(), sel);
        const int rb = qm_host_write_masks(b.c_str(), text, len, info.n_lines, off.data(), kind.data(), kept.data(), tpm.data(), flags.data(), sel);
        if (ra != rb || (ra == QM_OK && slurp(a) != slurp(b))) { fprintf(stderr, "writer mismatch: masks vs bytes, select %d, %zu bytes of text\n", sel, len); exit(1); }
        g_calls += 2;
      }
    }
    int64_t counts[5];
    std::vector<int32_t> tpos((size_t)cap), tref((size_t)cap), talt((size_t)cap);
    for (int mode = 0; mode < 2; ++mode) {
      if (ext && mode == 1) continue;
      qm_truth_scan_ext(text, len, mode, cap, tpos.data(), tref.data(), talt.data(), counts, ext ? dict : nullptr);
      // the host path with the file as its own truth text: every pattern shape the file can produce, against every line
      qm_patterns* pt = qm_patterns_create(text, len, mode, ext);
      if (!pt) { fprintf(stderr, "qm_patterns_create failed\n"); exit(1); }
      int64_t pinfo[4], ex[5];
      qm_patterns_info(pt, pinfo);
      std::vector<uint8_t> k2 = kind, f2 = flags;
      if (qm_vcf_hostpath(pt, text, len, info.n_lines, off.data(), k2.data(), pos.data(), ref.data(), alt.data(), f2.data(), ex) != QM_OK) {
        fprintf(stderr, "hostpath failed\n"); exit(1);
      }
      const std::string o = std::string(outdir) + "/h.vcf";
      qm_vcf_write(o.c_str(), text, len, info.n_lines, off.data(), k2.data(), cls.data(), 1);
      qm_patterns_destroy(pt);
      g_calls += 3;
    }
    g_calls += 6;
  }
  // the truth-set builder on whatever this file is (as the table AND as the FASTA): refusals are fine, reads past the end are not
  for (unsigned fl : {0u, 1u, 2u, 3u, 4u, 9u, 18u, 7u}) {
    uint8_t* o = nullptr; size_t on = 0;
    const int rc = qm_mummer2vcf(text, len, text, len, "ref.fa", fl, "20261004", &o, &on);
    if (rc == QM_OK) qm_free(o);
    ++g_calls;
  }
  {
    static const char tab[] = "5\t.\tA\t50\t5\t100\t20\t30\t1\t1\tc\tq\n5\t.\tC\t50\t5\t100\t20\t30\t1\t1\tc\tq\r\n1\tA\t.\t9\t5\t100\t20\t30\t1\t1\tc\tq\n2\tC\tG\t9\t5\t100\t20\t30\t1\t1\tc\tq";
    static const char fa[] = ">c x\nACGT\nAC\n>d\n";
    uint8_t* o = nullptr; size_t on = 0;
    if (qm_mummer2vcf((const uint8_t*)tab, sizeof tab - 1, (const uint8_t*)fa, sizeof fa - 1, nullptr, 3u, nullptr, &o, &on) != QM_OK || on == 0) { fprintf(stderr, "qm_mummer2vcf failed on its own example\n"); exit(1); }
    qm_free(o);
    ++g_calls;
  }
  for (int mode = 0; mode < 2; ++mode)
    for (int fl = 0; fl < 2; ++fl) {
      int64_t n = 0;
      const std::string o = std::string(outdir) + "/s.vcf";
      qm_vcf_split_write(o.c_str(), text, len, mode, fl, &n);
      ++g_calls;
    }
  {   // bgzip + tabix index: any return code is fine (mutated files are out of order or lack columns), no fault is
    const std::string o = std::string(outdir) + "/t.vcf.gz";
    (void)qm_bgzf_write_tbi(o.c_str(), text, len, 1);
    ++g_calls;
  }
  free(text);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: host_asan <outdir> <file>...\n"); return 2; }
  qm_dict* dict = qm_dict_create();
  unsigned rng = 12345;
  for (int a = 2; a < argc; ++a) {
    std::vector<uint8_t> t = slurp(argv[a]);
    if (t.size() > (4u << 20)) t.resize(4u << 20);
    exercise(t, dict, argv[1]);
    // truncations at awkward places and random byte flips
    for (int k = 0; k < 24 && !t.empty(); ++k) {
      rng = rng * 1664525u + 1013904223u;
      std::vector<uint8_t> u(t.begin(), t.begin() + (rng % (t.size() + 1)));
      exercise(u, dict, argv[1]);
      std::vector<uint8_t> w = t;
      for (int j = 0; j < 16; ++j) {
        rng = rng * 1664525u + 1013904223u;
        static const uint8_t alphabet[] = {'\t', '\n', '\r', 0, 0xff, '.', 'A', '9', '#', ',', ' ', 'e', '-', '+'};
        w[(rng >> 8) % w.size()] = alphabet[rng % sizeof alphabet];
      }
      exercise(w, dict, argv[1]);
    }
  }
  uint8_t buf[64];
  for (int32_t c : {0, 3, 4, -1, (2 << 26) | 5, 0x40000000, 0x40000001, 0x7fffffff, (int32_t)0x80000000}) qm_allele_spell(dict, c, buf, sizeof buf);
  printf("host sanitizer run: %ld calls, dictionary %lld entries, clean\n", g_calls, (long long)qm_dict_size(dict));
  qm_dict_destroy(dict);
  return 0;
}
