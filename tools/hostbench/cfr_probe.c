// copy_file_range(2) against writev(2) from the mapped input: CPU seconds to copy the kept runs of a VCF-like text (92 % of the lines, in runs of
// ~12 lines) into a new file -- is it worth carrying the input's descriptor to the writers?   cc -O2 -o cfr_probe cfr_probe.c && ./cfr_probe <dir>
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <sys/sendfile.h>
#include <time.h>
#include <unistd.h>
static double cpu_s(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6; }
static double wall_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "/tmp";
  char in[512], out[512];
  snprintf(in, sizeof in, "%s/cfr_in.vcf", dir); snprintf(out, sizeof out, "%s/cfr_out.vcf", dir);
  const size_t nlines = 1000000, ll = 64, len = nlines * ll;
  char* buf = malloc(len);
  for (size_t i = 0; i < nlines; ++i) { memset(buf + i * ll, 'a' + (int)(i % 26), ll - 1); buf[i * ll + ll - 1] = '\n'; }
  int fd = open(in, O_WRONLY | O_CREAT | O_TRUNC, 0644); if (write(fd, buf, len) != (ssize_t)len) return 1; close(fd); free(buf);
  // runs: 12 kept lines, 1 dropped
  size_t nruns = 0; size_t* rb = malloc(sizeof(size_t) * nlines); size_t* rn = malloc(sizeof(size_t) * nlines);
  for (size_t i = 0; i < nlines; i += 13) { rb[nruns] = i * ll; rn[nruns] = (i + 12 <= nlines ? 12 : nlines - i) * ll; ++nruns; }
  for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 3; ++rep) {
    int fi = open(in, O_RDONLY), fo = open(out, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    const char* map = mmap(NULL, len, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fi, 0);
    const double c0 = cpu_s(), w0 = wall_s();
    int err = 0;
    if (mode == 0) {   // writev from the mapping, 1024 ranges a call
      struct iovec iov[1024]; size_t k = 0;
      for (size_t r = 0; r < nruns; ++r) { iov[k].iov_base = (void*)(map + rb[r]); iov[k].iov_len = rn[r]; if (++k == 1024 || r + 1 == nruns) { if (writev(fo, iov, (int)k) < 0) err = errno; k = 0; } }
    } else if (mode == 1) {   // one copy_file_range per run
      for (size_t r = 0; r < nruns && !err; ++r) { off_t o = (off_t)rb[r]; size_t left = rn[r]; while (left) { ssize_t w = copy_file_range(fi, &o, fo, NULL, left, 0); if (w <= 0) { err = errno ? errno : -1; break; } left -= (size_t)w; } }
    } else {   // sendfile per run
      for (size_t r = 0; r < nruns && !err; ++r) { off_t o = (off_t)rb[r]; size_t left = rn[r]; while (left) { ssize_t w = sendfile(fo, fi, &o, left); if (w <= 0) { err = errno ? errno : -1; break; } left -= (size_t)w; } }
    }
    const double c1 = cpu_s(), w1 = wall_s();
    printf("%-16s rep %d: cpu %.3f s wall %.3f s%s%s\n", mode == 0 ? "writev(mmap)" : mode == 1 ? "copy_file_range" : "sendfile", rep, c1 - c0, w1 - w0, err ? " ERROR " : "", err ? strerror(err) : "");
    munmap((void*)map, len); close(fi); close(fo);
  }
  unlink(in); unlink(out);
  return 0;
}
