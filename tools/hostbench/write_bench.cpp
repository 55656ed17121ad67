// CPU time of the three output files of one VCF (qm_host_write_masks: header block + the selected lines, gathered from the
// input text) on a LoFreq-like VCF of N lines, 92 % of them kept and 8 % of those TP, as the pipeline calls it.
// build + run: g++ -O3 -std=c++17 -pthread -I include -o /tmp/write_bench tools/hostbench/write_bench.cpp quasimodo_amd/csrc/qmvt_host.cpp -lz && /tmp/write_bench [N] [dir]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/resource.h>
#include "qmvt.h"
int qm_host_write_masks(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                        const uint8_t* line_kind, const uint64_t* kept, const uint64_t* tp, const uint8_t* flags, int select);
static double cpu_s() { rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const std::string dir = argc > 2 ? argv[2] : "/dev/shm";
  std::string text = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
  uint64_t x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  char buf[128];
  for (int64_t i = 0; i < n; ++i) { const int k = snprintf(buf, sizeof buf, "chr1\t%lld\t.\tA\tG\t%d\tPASS\tDP=%d;AF=0.01\n", (long long)(i * 5 + 1), (int)(rnd() & 255), (int)(rnd() % 1000)); text.append(buf, (size_t)k); }
  const uint8_t* t = (const uint8_t*)text.data();
  const int64_t cap = qm_vcf_count_lines(t, text.size()) + 1;
  std::vector<int64_t> off((size_t)cap + 1);
  std::vector<uint8_t> kind((size_t)cap), flags((size_t)cap);
  std::vector<int32_t> pos((size_t)cap), ref((size_t)cap), alt((size_t)cap);
  std::vector<float> q((size_t)cap);
  qm_vcf_cols info;
  if (qm_vcf_scan(t, text.size(), cap, off.data(), kind.data(), pos.data(), ref.data(), alt.data(), q.data(), flags.data(), &info) != QM_OK) return 1;
  std::vector<uint64_t> kept((size_t)(info.n_data + 63) / 64 + 1, 0), tp(kept.size(), 0);
  int64_t nk = 0, nt = 0;
  for (int64_t r = 0; r < info.n_data; ++r) {
    if (flags[(size_t)r] & QM_F_PASS) { kept[(size_t)(r >> 6)] |= 1ull << (r & 63); ++nk; if (rnd() % 100 < 8) { tp[(size_t)(r >> 6)] |= 1ull << (r & 63); ++nt; } }
  }
  printf("%lld lines, %.1f MB, kept %lld, tp %lld\n", (long long)info.n_lines, text.size() / 1e6, (long long)nk, (long long)nt);
  const char* names[3] = {"filtered", "tp", "fp"};
  for (int rep = 0; rep < 3; ++rep) {
    double tot_w = 0, tot_c = 0;
    for (int sel = 0; sel < 3; ++sel) {
      const std::string path = dir + "/qm_write_bench." + names[sel] + ".vcf";
      const double c0 = cpu_s();
      const auto t0 = std::chrono::steady_clock::now();
      const int rc = qm_host_write_masks(path.c_str(), t, text.size(), info.n_lines, off.data(), kind.data(), kept.data(), tp.data(), flags.data(), sel);
      const double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), c = cpu_s() - c0;
      if (rc != QM_OK) { printf("write failed %d\n", rc); return 1; }
      if (rep == 2) printf("  %-8s wall %.4f s cpu %.4f s\n", names[sel], w, c);
      tot_w += w; tot_c += c;
    }
    printf("rep %d: three files wall %.4f s cpu %.4f s = %.3f CPU-s per GB of input\n", rep, tot_w, tot_c, tot_c / (text.size() / 1e9));
  }
  for (int sel = 0; sel < 3; ++sel) { FILE* f = fopen((dir + "/qm_write_bench." + names[sel] + ".vcf").c_str(), "rb"); fseek(f, 0, SEEK_END); printf("  %s: %ld bytes\n", names[sel], ftell(f)); fclose(f); remove((dir + "/qm_write_bench." + names[sel] + ".vcf").c_str()); }
  return 0;
}
