// qmvt_host.cpp -- host text side of libqmvt.so: VCF / truth tokenizer-packer
// and the output writers.  No GPU code, no classification: this file turns
// text into the packed columns of include/qmvt.h and class bits back into files.
//
// Replaces (file:line in /root/reference):
//   awk -F"\t" '$4~/^[ACGT]$/&&$5~/^[ACGT]$/&&($6>=20||$6==".")'   program/extract_TP_FP_SNPs.py:24
//   grep -E "^#"                                                    :27,50,52
//   awk ... {print $2,".",$4,$5}  /  {print $1,".",$2,$3}           :47, :92
//   shell `>` redirection of the selected lines                     :50-53
#include <algorithm>
#include <array>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <locale.h>
#include <wctype.h>
#include <sys/uio.h>
#include <unistd.h>
#include <sched.h>
#include <zlib.h>

#include "../../include/qmvt.h"

// Interned long alleles of the allele-extended mode (include/qmvt.h): id <-> string, thread safe.
struct qm_dict {
  std::mutex mu;
  std::unordered_map<std::string, int32_t> ids;
  std::vector<std::string> strings;
};

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

struct Span { const uint8_t* p; size_t n; };

inline bool acgt1(Span f) { return f.n == 1 && (f.p[0] == 'A' || f.p[0] == 'C' || f.p[0] == 'G' || f.p[0] == 'T'); }
inline int base_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
inline bool is_dot(Span f) { return f.n == 1 && f.p[0] == '.'; }
inline bool acgt_all(Span f) {
  if (f.n == 0) return false;
  for (size_t i = 0; i < f.n; ++i)
    if (base_code(f.p[i]) > 3) return false;
  return true;
}
// allele code of an [ACGT]+ field (caller checked): one base, 2..13 inline, longer ones interned
int32_t allele_code(qm_dict* d, Span f) {
  if (f.n == 1) return base_code(f.p[0]);
  if (f.n <= (size_t)QM_ALLELE_INLINE_MAX) {
    uint32_t c = (uint32_t)f.n << 26;
    for (size_t k = 0; k < f.n; ++k) c |= (uint32_t)base_code(f.p[k]) << (2 * k);
    return (int32_t)c;
  }
  if (!d) return QM_ALLELE_NONE;
  std::string key((const char*)f.p, f.n);
  std::lock_guard<std::mutex> g(d->mu);
  auto it = d->ids.find(key);
  if (it != d->ids.end()) return (int32_t)(QM_ALLELE_DICT | it->second);
  if (d->strings.size() >= (size_t)0x3fffffff) return QM_ALLELE_NONE;
  const int32_t id = (int32_t)d->strings.size();
  d->strings.push_back(key);
  d->ids.emplace(std::move(key), id);
  return (int32_t)(QM_ALLELE_DICT | id);
}
inline bool is_word(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }
// Non-ASCII text (round 6).  The reference's grep runs under the locale CPython exports -- LC_CTYPE=C.UTF-8 when the caller's is
// C / POSIX (PEP 538; tests/golden/PROVENANCE.md) -- where `grep -w` asks iswalnum() about the CHARACTER next to a match and a line
// that is not valid UTF-8 turns the output into "binary file matches".  The same question to the same glibc locale here: a kept line
// of valid UTF-8 is decided from its text on the host (qm_vcf_hostpath), one with a NUL or an invalid sequence is still refused --
// and so is every non-ASCII line on a host without that locale.  Pinned by tests/golden/utf8/ (written by the reference).
inline locale_t utf8_locale() {
  static locale_t loc = newlocale(LC_CTYPE_MASK, "C.UTF-8", (locale_t)0);
  return loc;
}
// length of the valid UTF-8 sequence at p (1..4), 0 if there is none
inline int utf8_at(const uint8_t* p, size_t n, uint32_t* cp) {
  if (!n) return 0;
  const uint8_t c = p[0];
  if (c < 0x80) { *cp = c; return 1; }
  int len; uint32_t v, min;
  if (c >= 0xc2 && c <= 0xdf) { len = 2; v = c & 0x1fu; min = 0x80; }
  else if (c >= 0xe0 && c <= 0xef) { len = 3; v = c & 0x0fu; min = 0x800; }
  else if (c >= 0xf0 && c <= 0xf4) { len = 4; v = c & 0x07u; min = 0x10000; }
  else return 0;
  if ((size_t)len > n) return 0;
  for (int k = 1; k < len; ++k) { if ((p[k] & 0xc0) != 0x80) return 0; v = (v << 6) | (p[k] & 0x3fu); }
  if (v < min || v > 0x10ffff || (v >= 0xd800 && v <= 0xdfff)) return 0;
  *cp = v;
  return len;
}
inline bool utf8_text(const uint8_t* p, size_t n) {   // valid UTF-8 without a NUL, on a host that has the locale to read it with
  if (!utf8_locale()) return false;
  for (size_t i = 0; i < n;) {
    uint32_t cp;
    const int l = p[i] ? utf8_at(p + i, n - i, &cp) : 0;
    if (!l) return false;
    i += (size_t)l;
  }
  return true;
}
inline bool word_cp(uint32_t cp) {
  if (cp < 0x80) return is_word((uint8_t)cp);
  const locale_t loc = utf8_locale();
  return loc && iswalnum_l((wint_t)cp, loc) != 0;
}
// is the character that ends right before s[i] (i > lo) / that starts at s[i] (i < hi) a word character?
inline bool word_before(const uint8_t* s, size_t lo, size_t i) {
  if (s[i - 1] < 0x80) return is_word(s[i - 1]);
  size_t b = i - 1;
  while (b > lo && (s[b] & 0xc0) == 0x80 && i - b < 4) --b;
  uint32_t cp = 0;
  return utf8_at(s + b, i - b, &cp) == (int)(i - b) && word_cp(cp);
}
inline bool word_at(const uint8_t* s, size_t i, size_t hi) {
  if (s[i] < 0x80) return is_word(s[i]);
  uint32_t cp = 0;
  return utf8_at(s + i, hi - i, &cp) != 0 && word_cp(cp);
}
// begins with [ACGT]+ that ends at the field's end or before a non-word character (what `grep -w` needs)
inline bool acgt_prefix_word(Span f) {
  size_t k = 0;
  while (k < f.n && base_code(f.p[k]) <= 3) ++k;
  return k > 0 && (k == f.n || !is_word(f.p[k]));
}

// canonical decimal POS: "0" or [1-9][0-9]*, value < 2^28
inline bool canon_pos(Span f, int32_t* out) {
  if (f.n == 0 || f.n > 9) return false;
  if (f.n > 1 && f.p[0] == '0') return false;
  uint32_t v = 0;
  for (size_t i = 0; i < f.n; ++i) {
    if (f.p[i] < '0' || f.p[i] > '9') return false;
    v = v * 10 + (uint32_t)(f.p[i] - '0');
  }
  if (v >= (uint32_t)QM_POS_LIMIT) return false;
  *out = (int32_t)v;
  return true;
}

// mawk 1.3.4 numeric-string test (tests/golden/PROVENANCE.md): blanks stripped,
// last char digit or '.', first char digit/+/-/., glibc strtod consumes all,
// any ERANGE -> plain string.
bool awk_strnum(Span f, double* d) {
  const uint8_t* s = f.p;
  const uint8_t* q = f.p + f.n;
  while (s < q && (*s == ' ' || *s == '\t')) ++s;
  if (s == q) return false;
  while (q[-1] == ' ' || q[-1] == '\t') --q;
  const uint8_t last = q[-1], first = *s;
  if (!((last >= '0' && last <= '9') || last == '.')) return false;
  if (!((first >= '0' && first <= '9') || first == '+' || first == '-' || first == '.')) return false;
  const size_t m = (size_t)(q - s);
  if (memchr(s, 0, m)) return false;
  char tmp[256];
  std::string heap;
  char* z = tmp;
  if (m + 1 > sizeof tmp) { heap.assign((const char*)s, m); z = &heap[0]; } else { memcpy(tmp, s, m); tmp[m] = 0; }
  char* endp = nullptr;
  errno = 0;
  const double v = strtod(z, &endp);
  if (endp != z + m || errno != 0) return false;
  *d = v;
  return true;
}

// Plain decimals -- optional sign, <= 15 significant digits, optional fraction of <= 15 digits, no
// exponent -- convert exactly with one correctly rounded division (Clinger's fast path), which is
// what strtod returns for them; everything else goes through awk_strnum / strtod.
static const double kPow10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
inline bool fast_decimal(Span f, double* d) {
  const uint8_t* s = f.p;
  const uint8_t* q = f.p + f.n;
  if (s == q) return false;
  bool neg = false;
  if (*s == '+' || *s == '-') { neg = *s == '-'; ++s; }
  uint64_t mant = 0;
  int digits = 0, frac = 0;
  bool seen_dot = false, any = false;
  for (; s < q; ++s) {
    if (*s >= '0' && *s <= '9') {
      if (digits >= 15) return false;
      mant = mant * 10 + (uint64_t)(*s - '0');
      digits += (mant != 0 || digits != 0) ? 1 : 0;
      frac += seen_dot ? 1 : 0;
      any = true;
    } else if (*s == '.' && !seen_dot) {
      seen_dot = true;
    } else {
      return false;
    }
  }
  if (!any || frac > 15) return false;
  const double v = (double)mant / kPow10[frac];
  *d = neg ? -v : v;
  return true;
}

// Effective QUAL: floor(q) >= t  <=>  awk `$6>=t` for integer t, numeric fields;
// other spellings collapse to +-inf by their answer at t = 20.
float effective_qual(Span f, bool* ge20) {
  double d;
  if (fast_decimal(f, &d) || awk_strnum(f, &d)) {
    *ge20 = d >= 20.0;
    float q = (float)d;
    if ((double)q > d) q = nextafterf(q, -INFINITY);  // round toward -inf: never crosses an integer upward
    return q;
  }
  if (is_dot(f)) { *ge20 = true; return INFINITY; }  // `||$6=="."`
  const size_t m = f.n < 2 ? f.n : 2;
  const int c = memcmp(f.p, "20", m);
  const bool ge = c != 0 ? c > 0 : f.n >= 2;
  *ge20 = ge;
  return ge ? INFINITY : -INFINITY;
}

inline int split_tabs(const uint8_t* line, size_t n, Span* f, int maxf) {
  if (n == 0) return 0;
  int nf = 0;
  const uint8_t* s = line;
  const uint8_t* end = line + n;
  for (;;) {
    const uint8_t* t = (const uint8_t*)memchr(s, '\t', (size_t)(end - s));
    if (nf < maxf) { f[nf].p = s; f[nf].n = (size_t)((t ? t : end) - s); }
    ++nf;
    if (!t) break;
    s = t + 1;
  }
  return nf;
}
// the first maxf fields only (the tokenizer needs CHROM..QUAL); returns how many it found
inline int split_head(const uint8_t* line, size_t n, Span* f, int maxf) {
  if (n == 0) return 0;
  int nf = 0;
  const uint8_t* s = line;
  const uint8_t* end = line + n;
  while (nf < maxf) {
    const uint8_t* t = (const uint8_t*)memchr(s, '\t', (size_t)(end - s));
    f[nf].p = s; f[nf].n = (size_t)((t ? t : end) - s);
    ++nf;
    if (!t) break;
    s = t + 1;
  }
  return nf;
}
// could a truth pattern X\t.\tY\tZ also sit on a "\t.\t" behind the ID column?  Walks the fields from
// ALT (field 4) to the end of the line with a window of four: (anything, ".", Y-like, Z-like then a non-word character).
// Y-like / Z-like = one character each (allele-extended: [ACGT]+); truth sets holding patterns of other shapes send
// every line to the host path anyway (qm_patterns_info).
bool pattern_at_later_fields(const uint8_t* s, const uint8_t* end, bool ext) {
  Span w[4] = {{s, 0}, {s, 0}, {s, 0}, {s, 0}};
  int have = 0;
  for (;;) {
    const uint8_t* t = (const uint8_t*)memchr(s, '\t', (size_t)(end - s));
    w[0] = w[1]; w[1] = w[2]; w[2] = w[3];
    w[3].p = s; w[3].n = (size_t)((t ? t : end) - s);
    if (++have >= 4 && is_dot(w[1]) &&
        (ext ? acgt_all(w[2]) && acgt_prefix_word(w[3]) : w[2].n == 1 && w[3].n >= 1 && (w[3].n == 1 || !is_word(w[3].p[1]))))
      return true;
    if (!t) return false;
    s = t + 1;
  }
}


// Lines and header lines ('#' first) of a piece of text without looking at it line by line: newlines are counted with a
// population count over compare masks, header lines are the line starts (offset 0, every byte behind a newline) that hold '#'.
// A last line without a newline counts; an empty line is a data line (as everywhere in the tokenizer).
#if defined(__x86_64__)
__attribute__((target("avx2"))) void count_lines_avx2(const uint8_t* p, size_t len, int64_t* nl, int64_t* nh, uint32_t* carry) {
  const __m256i vn = _mm256_set1_epi8('\n'), vh = _mm256_set1_epi8('#');
  int64_t l = 0, h = 0;
  uint32_t c = *carry;
  size_t i = 0;
  for (; i + 32 <= len; i += 32) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + i));
    const uint32_t mn = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vn)), mh = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vh));
    l += __builtin_popcount(mn);
    h += __builtin_popcount(((mn << 1) | c) & mh);
    c = mn >> 31;
  }
  for (; i < len; ++i) { h += (c && p[i] == '#') ? 1 : 0; c = p[i] == '\n'; l += c; }
  *nl += l; *nh += h; *carry = c;
}
bool cpu_has_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#endif
void count_lines_fast(const uint8_t* p, size_t len, int64_t* n_lines, int64_t* n_data) {
  int64_t l = 0, h = 0;
  uint32_t c = 1;   // offset 0 starts a line
  size_t i = 0;
#if defined(__x86_64__)
  if (cpu_has_avx2()) {
    count_lines_avx2(p, len, &l, &h, &c);
    i = len;
  } else {
    const __m128i vn = _mm_set1_epi8('\n'), vh = _mm_set1_epi8('#');
    for (; i + 16 <= len; i += 16) {
      const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i));
      const uint32_t mn = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vn)), mh = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vh));
      l += __builtin_popcount(mn);
      h += __builtin_popcount(((mn << 1) | c) & mh & 0xffffu);
      c = (mn >> 15) & 1u;
    }
  }
#endif
  for (; i < len; ++i) { h += (c && p[i] == '#') ? 1 : 0; c = p[i] == '\n'; l += c; }
  if (len && p[len - 1] != '\n') ++l;
  *n_lines = l;
  *n_data = l - h;
}

// One pass over a line, 16 or 32 bytes per compare (SURVEY f-1: "SIMD tab/newline scan"): its length, where its tabs are,
// whether it holds a NUL or a non-ASCII byte.  The tokenizer used to find the same things with one memchr per field, a second
// walk for the byte check and a third for pattern_at_later_fields; fields are a few bytes long, so the calls cost more than
// the bytes.
constexpr int LINE_MAXT = 48;   // tab positions kept (a line with more falls back to the walks above)
struct LineIndex {
  size_t n;        // length without the newline
  int ntab;        // all tabs of the line
  bool dirty;      // NUL or non-ASCII byte
  bool punct;      // a byte in 0x20..0x27 (space ! " # $ % & '): the line MAY hold one of the three that R's read.table reads its own way
  uint32_t tab[LINE_MAXT];
};
#define QM_INDEX_TABS(mt, off) while (mt) { const uint32_t i_ = (uint32_t)__builtin_ctz(mt); mt &= mt - 1u; if (nt < LINE_MAXT) L.tab[nt] = (off) + i_; ++nt; }
inline void index_line_tail(const uint8_t* s, const uint8_t* p, const uint8_t* lim, int nt, bool dirty, bool punct, LineIndex& L) {
  for (; p < lim && *p != '\n'; ++p) {
    if (*p == '\t') { if (nt < LINE_MAXT) L.tab[nt] = (uint32_t)(p - s); ++nt; }
    dirty = dirty || *p == 0 || *p >= 0x80;
    punct = punct || (*p & 0xf8) == 0x20;
  }
  L.n = (size_t)(p - s); L.ntab = nt; L.dirty = dirty; L.punct = punct;
}
// a quote character anywhere, or a '#' in front of the last column: R's read.table (comment.char = "#", quote = "\"'") does not split
// such a line at its tabs alone.  (A '#' INSIDE the last column only shortens that column: the field count stays, and the
// counting scripts never read it.)
inline bool r_hostile_line(const uint8_t* s, size_t n) {
  size_t hash = n;                       // first '#'
  size_t last_tab = 0; bool any_tab = false;
  for (size_t i = 0; i < n; ++i) {
    if (s[i] == '\'' || s[i] == '"') return true;
    if (s[i] == '#' && hash == n) hash = i;
    if (s[i] == '\t') { last_tab = i; any_tab = true; }
  }
  return hash < n && !(any_tab && hash > last_tab);
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) void index_line_avx2(const uint8_t* s, const uint8_t* lim, LineIndex& L) {
  const uint8_t* p = s;
  int nt = 0;
  bool dirty = false, punct = false;
  const __m256i vt = _mm256_set1_epi8('\t'), vn = _mm256_set1_epi8('\n'), vz = _mm256_setzero_si256();
  const __m256i vf8 = _mm256_set1_epi8((char)0xf8), v20 = _mm256_set1_epi8(0x20);
  while (p + 32 <= lim) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
    uint32_t mt = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vt));
    const uint32_t mn = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vn));
    uint32_t md = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vz)) | (uint32_t)_mm256_movemask_epi8(x);   // zero bytes | high bits
    uint32_t mq = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(x, vf8), v20));              // 0x20..0x27
    const uint32_t off = (uint32_t)(p - s);
    if (mn) {
      const uint32_t e = (uint32_t)__builtin_ctz(mn), below = e ? (0xffffffffu >> (32u - e)) : 0u;
      mt &= below; md &= below; mq &= below;
      dirty = dirty || md != 0;
      punct = punct || mq != 0;
      QM_INDEX_TABS(mt, off)
      L.n = (size_t)off + e; L.ntab = nt; L.dirty = dirty; L.punct = punct;
      return;
    }
    dirty = dirty || md != 0;
    punct = punct || mq != 0;
    QM_INDEX_TABS(mt, off)
    p += 32;
  }
  index_line_tail(s, p, lim, nt, dirty, punct, L);
}
#endif
inline void index_line(const uint8_t* s, const uint8_t* lim, LineIndex& L) {   // lim: end of the chunk (whole lines; inside the mapping)
#if defined(__x86_64__)
  if (cpu_has_avx2()) { index_line_avx2(s, lim, L); return; }
  const uint8_t* p = s;
  int nt = 0;
  bool dirty = false, punct = false;
  const __m128i vt = _mm_set1_epi8('\t'), vn = _mm_set1_epi8('\n'), vz = _mm_setzero_si128();
  const __m128i vf8 = _mm_set1_epi8((char)0xf8), v20 = _mm_set1_epi8(0x20);
  while (p + 16 <= lim) {
    const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
    uint32_t mt = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vt));
    const uint32_t mn = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vn));
    uint32_t md = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, vz)) | (uint32_t)_mm_movemask_epi8(x);   // zero bytes | high bits
    uint32_t mq = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(x, vf8), v20));              // 0x20..0x27
    const uint32_t off = (uint32_t)(p - s);
    if (mn) {
      const uint32_t e = (uint32_t)__builtin_ctz(mn), below = (1u << e) - 1u;
      mt &= below; md &= below; mq &= below;
      dirty = dirty || md != 0;
      punct = punct || mq != 0;
      QM_INDEX_TABS(mt, off)
      L.n = (size_t)off + e; L.ntab = nt; L.dirty = dirty; L.punct = punct;
      return;
    }
    dirty = dirty || md != 0;
    punct = punct || mq != 0;
    QM_INDEX_TABS(mt, off)
    p += 16;
  }
  index_line_tail(s, p, lim, nt, dirty, punct, L);
#else
  index_line_tail(s, s, lim, 0, false, false, L);
#endif
}
#undef QM_INDEX_TABS
// field k of an indexed line (k <= ntab <= LINE_MAXT)
inline Span line_field(const uint8_t* s, const LineIndex& L, int k) {
  const size_t b = k ? (size_t)L.tab[k - 1] + 1 : 0, e = k < L.ntab ? (size_t)L.tab[k] : L.n;
  return Span{s + b, e - b};
}
// pattern_at_later_fields on an indexed line: windows (field k-3, ".", Y, Z) for k >= 7
inline bool pattern_at_later_fields_indexed(const uint8_t* s, const LineIndex& L, bool ext) {
  for (int k = 7; k <= L.ntab; ++k) {
    const Span d = line_field(s, L, k - 2);
    if (!is_dot(d)) continue;
    const Span y = line_field(s, L, k - 1), z = line_field(s, L, k);
    if (ext ? acgt_all(y) && acgt_prefix_word(z) : y.n == 1 && z.n >= 1 && (z.n == 1 || !is_word(z.p[1]))) return true;
  }
  return false;
}

}  // namespace

extern "C" int64_t qm_vcf_count_lines(const uint8_t* text, size_t len) {
  int64_t nl = 0, nd = 0;
  if (text && len) count_lines_fast(text, len, &nl, &nd);
  return nl;
}
// internal (qmvt_pipeline.cpp): lines and data lines of a whole file
void qm_host_count_lines(const uint8_t* text, size_t len, int64_t* n_lines, int64_t* n_data) {
  *n_lines = 0; *n_data = 0;
  if (text && len) count_lines_fast(text, len, n_lines, n_data);
}

// One chunk of the text (whole lines).  count_only: just the number of lines / data lines.
struct ScanChunk {
  size_t begin = 0, end = 0;       // byte range
  int64_t nl = 0, nd = 0;          // lines / data lines in the chunk
  int64_t l0 = 0, d0 = 0;          // global index of its first line / data line
  int64_t nhost = 0, nref = 0, first_ref = 0, nnokey = 0;   // host-path lines; refused lines, 1-based global line of the first; kept NOKEY lines
  int64_t nrq = 0, first_rq = 0;   // kept data lines holding '#', ' or ": R's read.table does not read them the way the counts assume; the first one's line
  int32_t last_pos = 0;            // last canonical POS seen in the chunk (0 if none)
  int64_t lead_nokey = 0;          // data lines before the chunk's first canonical POS
  bool any_pos = false;
  int64_t cap_lines = INT64_MAX;   // room in the caller's line / column arrays (global line index)
  bool overflow = false;           // the text holds more lines than that (a file that grew between the count and the scan)
};

// dict != nullptr: allele-extended tokenising -- the filter's `^[ACGT]$` becomes `^[ACGT]+$`
static void scan_chunk(const uint8_t* text, ScanChunk& c, bool count_only, int64_t* line_off, uint8_t* line_kind, int32_t* pos,
                       int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_dict* dict) {
  auto allele_ok = [dict](Span f) { return dict ? acgt_all(f) : acgt1(f); };
  int64_t nl = 0, nd = 0;
  int32_t last_pos = 0;
  bool any_pos = false;
  size_t off = c.begin;
  enum { MAXF = 6 };   // CHROM POS ID REF ALT QUAL
  Span f[MAXF];
  if (count_only) {
    count_lines_fast(text + c.begin, c.end - c.begin, &c.nl, &c.nd);
    return;
  }
  LineIndex L;
  while (off < c.end) {
    const uint8_t* s = text + off;
    index_line(s, text + c.end, L);
    const size_t n = L.n;
    const bool header = n && s[0] == '#';
    {
      const int64_t gl = c.l0 + nl, gd = c.d0 + nd;
      if (gl >= c.cap_lines) { c.overflow = true; break; }   // never write past the caller's arrays, whoever counted the lines
      line_off[gl] = (int64_t)off;
      const bool indexed = L.ntab <= LINE_MAXT;   // every tab of the line is in the index
      auto dirty = [&]() { return L.dirty; };     // NUL / non-ASCII: the reference's answer depends on the locale its grep runs under
      auto head = [&](Span* f) -> int {           // the first MAXF fields (CHROM..QUAL); returns how many the line has of them
        if (n == 0) return 0;
        if (!indexed) return split_head(s, n, f, MAXF);
        const int nf = L.ntab + 1 < (int)MAXF ? L.ntab + 1 : (int)MAXF;
        for (int k = 0; k < nf; ++k) f[k] = line_field(s, L, k);
        return nf;
      };
      if (header) {
        // awk does not skip '#': a header line that also satisfies the A2 filter is emitted twice by the reference
        // (once by grep, once by awk, then in tp or fp as fgrep decides) while R's read.table ignores it
        uint8_t kind = QM_LINE_HEADER;
        const int nf = head(f);
        if (nf >= 5 && allele_ok(f[3]) && allele_ok(f[4])) {
          bool ge20 = false;
          const Span empty = {(const uint8_t*)"", 0};
          (void)effective_qual(nf > 5 ? f[5] : empty, &ge20);
          if (ge20) {
            if (dirty() && !utf8_text(s, n)) { kind = QM_LINE_HEADER_REFUSED; ++c.nref; if (!c.first_ref) c.first_ref = gl + 1; }
            else { kind = QM_LINE_HEADER_KEPT; ++c.nhost; }
          }
        }
        line_kind[gl] = kind;
      } else {
        uint8_t kind = QM_LINE_DATA;
        const int nf = head(f);
        const Span empty = {(const uint8_t*)"", 0};
        const Span fpos = nf > 1 ? f[1] : empty, fid = nf > 2 ? f[2] : empty, fref = nf > 3 ? f[3] : empty,
                   falt = nf > 4 ? f[4] : empty, fq = nf > 5 ? f[5] : empty;
        const bool snp = allele_ok(fref) && allele_ok(falt);
        bool ge20 = false;
        const float q = effective_qual(fq, &ge20);
        const bool pass = snp && ge20;
        int32_t p = -1;
        const bool cpos = canon_pos(fpos, &p);
        if (cpos) { last_pos = p; any_pos = true; } else { p = last_pos; if (!any_pos) c.lead_nokey++; }
        const bool high = dirty() && snp;                 // NUL or non-ASCII bytes in a single-base line
        const bool text_ok = high && utf8_text(s, n);     // ... that is valid UTF-8: grep compares it as text, and so does the host path
        if (pass && high && !text_ok) {
          kind = QM_LINE_REFUSED; ++c.nref; if (!c.first_ref) c.first_ref = gl + 1;
        } else if (snp && (text_ok || !cpos || (nf > 5 && (indexed ? pattern_at_later_fields_indexed(s, L, dict != nullptr) : pattern_at_later_fields(f[4].p, s + n, dict != nullptr))))) {
          // every single-base line, whatever its QUAL: the ROC sweep moves the threshold.  fgrep compares POS as
          // text, so only canonical spellings are safe on the device
          kind = QM_LINE_DATA_HOST; ++c.nhost;
        }
        if (pass && !cpos) ++c.nnokey;
        if (pass && L.punct && r_hostile_line(s, n)) { ++c.nrq; if (!c.first_rq) c.first_rq = gl + 1; }
        line_kind[gl] = kind;
        if (pos) {
          pos[gd] = p;
          if (dict) {
            ref[gd] = acgt_all(fref) ? allele_code(dict, fref) : QM_ALLELE_NONE;
            alt[gd] = acgt_all(falt) ? allele_code(dict, falt) : QM_ALLELE_NONE;
          } else {
            ref[gd] = fref.n == 1 ? base_code(fref.p[0]) : 4;
            alt[gd] = falt.n == 1 ? base_code(falt.p[0]) : 4;
          }
          qual[gd] = q;
          flags[gd] = (uint8_t)((pass ? QM_F_PASS : 0u) | (is_dot(fid) ? QM_F_IDDOT : 0u) | (cpos ? 0u : QM_F_NOKEY));
        }
      }
      if (!header) ++nd;
    }
    ++nl;
    off += n + 1;
  }
  c.nl = nl;
  c.nd = nd;
  c.last_pos = last_pos;
  c.any_pos = any_pos;
}

static int host_threads() {
  // QM_HOST_THREADS says it; otherwise the CPUs this process may run on (its affinity mask: a container's share of a
  // large host), at most 32 -- more threads than that gain nothing for the memory traffic of tokenising and writing
  // (profiles/r03_e2e_threads.log)
  if (const char* e = getenv("QM_HOST_THREADS")) { const int n = atoi(e); return n < 1 ? 1 : n > 256 ? 256 : n; }
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n < 1) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  return n > 32 ? 32 : n;
}

// Two passes over whole-line chunks, each pass with one thread per chunk: count, then fill at
// known offsets.  The only cross-chunk state, the position carried onto records without a
// comparable POS, is patched for the (few) records that precede a chunk's first canonical POS.
int qm_host_threads(void) { return host_threads(); }

// internal: qm_vcf_scan_ext with the number of threads given (qmvt_pipeline.cpp spreads its threads over files first)
int qm_host_scan_threads(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off, uint8_t* line_kind,
                         int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_vcf_cols* info,
                         qm_dict* dict, int nt) {
  // nt < 0: one thread AND the caller has counted the lines itself (cap_lines is exact or larger): no counting pass
  const bool counted = nt < 0;
  if ((!text && len) || !line_off || !line_kind || !info) return QM_E_INVAL;
  if (nt < 1) nt = 1;
  if (len < (size_t)(1 << 20)) nt = 1;
  std::vector<ScanChunk> ch((size_t)nt);
  size_t b = 0;
  for (int t = 0; t < nt; ++t) {
    size_t e = t == nt - 1 ? len : len * (size_t)(t + 1) / (size_t)nt;
    if (e < b) e = b;
    if (e < len) {   // extend to the end of the line
      const uint8_t* nlp = (const uint8_t*)memchr(text + e, '\n', len - e);
      e = nlp ? (size_t)(nlp - text) + 1 : len;
    }
    ch[(size_t)t].begin = b;
    ch[(size_t)t].end = e;
    b = e;
  }
  auto run = [&](bool count_only) {
    if (nt == 1) { scan_chunk(text, ch[0], count_only, line_off, line_kind, pos, ref, alt, qual, flags, dict); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t]() { scan_chunk(text, ch[(size_t)t], count_only, line_off, line_kind, pos, ref, alt, qual, flags, dict); });
    for (auto& x : th) x.join();
  };
  int64_t nl = 0, nd = 0;
  if (!counted) {
    run(true);
    for (auto& c : ch) { c.l0 = nl; c.d0 = nd; nl += c.nl; nd += c.nd; }
    if (nl > cap_lines) return QM_E_INVAL;
  }
  for (auto& c : ch) c.cap_lines = cap_lines;
  run(false);
  for (auto& c : ch) if (c.overflow) return QM_E_INVAL;
  if (counted) { nl = ch[0].nl; nd = ch[0].nd; }
  int64_t nhost = 0, nref = 0, first_ref = 0, nnokey = 0, nrq = 0, first_rq = 0;
  int32_t carry = 0;
  for (auto& c : ch) {
    if (pos && carry != 0)
      for (int64_t i = 0; i < c.lead_nokey; ++i) pos[c.d0 + i] = carry;   // leading records had no position of their own
    if (c.any_pos) carry = c.last_pos;
    nhost += c.nhost; nref += c.nref; nnokey += c.nnokey; nrq += c.nrq;
    if (!first_ref && c.first_ref) first_ref = c.first_ref;
    if (!first_rq && c.first_rq) first_rq = c.first_rq;
  }
  line_off[nl] = (int64_t)len;
  info->n_lines = nl;
  info->n_data = nd;
  info->n_host = nhost;
  info->n_refused = nref;
  info->first_refused_line = first_ref;
  info->n_nokey_kept = nnokey;
  info->n_r_hostile = nrq;
  info->first_r_hostile_line = first_rq;
  return QM_OK;
}

extern "C" int qm_vcf_scan_ext(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off, uint8_t* line_kind,
                               int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_vcf_cols* info,
                               qm_dict* dict) {
  return qm_host_scan_threads(text, len, cap_lines, line_off, line_kind, pos, ref, alt, qual, flags, info, dict, host_threads());
}

extern "C" int qm_vcf_scan(const uint8_t* text, size_t len, int64_t cap_lines, int64_t* line_off, uint8_t* line_kind,
                           int32_t* pos, int32_t* ref, int32_t* alt, float* qual, uint8_t* flags, qm_vcf_cols* info) {
  return qm_vcf_scan_ext(text, len, cap_lines, line_off, line_kind, pos, ref, alt, qual, flags, info, nullptr);
}

extern "C" qm_dict* qm_dict_create(void) { return new qm_dict(); }
extern "C" void qm_dict_destroy(qm_dict* d) { delete d; }
extern "C" int64_t qm_dict_size(qm_dict* d) {
  if (!d) return 0;
  std::lock_guard<std::mutex> g(d->mu);
  return (int64_t)d->strings.size();
}
extern "C" int32_t qm_allele_code(qm_dict* d, const uint8_t* s, size_t n) {
  const Span f = {s, n};
  if (!s || !acgt_all(f)) return QM_ALLELE_NONE;
  return allele_code(d, f);
}
extern "C" int64_t qm_allele_spell(qm_dict* d, int32_t code, uint8_t* out, size_t cap) {
  static const char B[] = "ACGT";
  const uint32_t u = (uint32_t)code;
  if (u < 4u) {
    if (cap < 1) return -1;
    out[0] = (uint8_t)B[u];
    return 1;
  }
  if (u >= 0x08000000u && u < (uint32_t)QM_ALLELE_DICT) {
    const size_t n = u >> 26;
    if (n < 2 || n > (size_t)QM_ALLELE_INLINE_MAX || cap < n) return -1;
    for (size_t k = 0; k < n; ++k) out[k] = (uint8_t)B[(u >> (2 * k)) & 3u];
    return (int64_t)n;
  }
  if (u >= (uint32_t)QM_ALLELE_DICT && u < 0x80000000u && d) {
    std::lock_guard<std::mutex> g(d->mu);
    const size_t id = u & 0x3fffffffu;
    if (id >= d->strings.size() || cap < d->strings[id].size()) return -1;
    memcpy(out, d->strings[id].data(), d->strings[id].size());
    return (int64_t)d->strings[id].size();
  }
  return -1;
}

extern "C" int64_t qm_truth_scan(const uint8_t* text, size_t len, int mode, int64_t cap, int32_t* pos, int32_t* ref, int32_t* alt,
                                 int64_t* out_counts) {
  return qm_truth_scan_ext(text, len, mode, cap, pos, ref, alt, out_counts, nullptr);
}

extern "C" int64_t qm_truth_scan_ext(const uint8_t* text, size_t len, int mode, int64_t cap, int32_t* pos, int32_t* ref, int32_t* alt,
                                     int64_t* out_counts, qm_dict* dict) {
  if ((!text && len) || (mode != 0 && mode != 1)) return QM_E_INVAL;
  if (dict && mode != 0) return QM_E_INVAL;   // show-snps tables spell gaps as '.', not as VCF alleles
  auto allele_ok = [dict](Span f) { return dict ? acgt_all(f) : acgt1(f); };
  int64_t genomediff = 0, nkeys = 0, never = 0, refused = 0, ncomment = 0;
  size_t off = 0;
  Span f[6];
  const Span empty = {(const uint8_t*)"", 0};
  while (off < len) {
    const uint8_t* s = text + off;
    const uint8_t* e = (const uint8_t*)memchr(s, '\n', len - off);
    const size_t n = e ? (size_t)(e - s) : len - off;
    off += n + 1;
    const int nf = split_tabs(s, n, f, 6);
    for (int i = nf < 6 ? nf : 6; i < 6; ++i) f[i] = empty;
    const bool comment = n && s[0] == '#';
    Span X, Y, Z;
    bool pattern;
    if (mode == 0) {  // extract_TP_FP_SNPs.py:47 ; R: caller_performance_compare.R:29-55
      X = f[1]; Y = f[3]; Z = f[4];
      pattern = allele_ok(Y) && allele_ok(Z);
      if (pattern && !comment) ++genomediff;
    } else {          // extract_TP_FP_SNPs.py:92 ; R: custom_snp_benchmark.R:23-27
      X = f[0]; Y = f[1]; Z = f[2];
      pattern = !is_dot(Y) && !is_dot(Z);
      if (pattern && !comment && n) ++genomediff;
    }
    if (!pattern) continue;
    if (comment) { ++ncomment; continue; }  // awk makes a pattern of it, R skips it: never a device key (qm_patterns keeps it)
    int32_t p;
    if (!allele_ok(Y) || !allele_ok(Z)) { ++never; continue; }   // Y must equal a single-base REF field
    if (!canon_pos(X, &p)) {
      // digits with a non-canonical spelling or out of range could match a non-canonical line
      ++never;
      continue;
    }
    // (a canonical pattern is ASCII by construction; what the row's OTHER columns hold -- a non-ASCII INFO text, say -- never
    // reaches the pattern list the reference's awk prints: such rows were refused through round 5)
    if (pos) {
      if (nkeys >= cap) return QM_E_INVAL;
      pos[nkeys] = p;
      ref[nkeys] = dict ? allele_code(dict, Y) : base_code(Y.p[0]);
      alt[nkeys] = dict ? allele_code(dict, Z) : base_code(Z.p[0]);
    }
    ++nkeys;
  }
  if (out_counts) { out_counts[0] = genomediff; out_counts[1] = nkeys; out_counts[2] = never; out_counts[3] = refused; out_counts[4] = ncomment; }
  return nkeys;
}

// ---------------------------------------------------------------------------
// Host path for the lines the columns cannot describe.
//
// The reference decides TP / FP with `fgrep -wf <(awk patterns) <(awk filter)` (program/extract_TP_FP_SNPs.py:47-53,
// custom :92-98): a line is selected when some pattern X\t.\tY\tZ occurs in it with a non-word character (or the line
// edge) on both sides.  For a line with a canonical POS and no other "\t.\t" a pattern could sit on, that is the key
// lookup the device does.  For the rest -- POS spelled "01000" or "1-1000", the pattern found at the QUAL/FILTER/INFO
// columns, ALT followed by ",T" there, '#' lines that pass the filter -- the answer needs the TEXT of the patterns.
// ---------------------------------------------------------------------------
namespace {

// set of byte strings: open addressing over one arena (lookups take a pointer + length, nothing is copied)
struct ByteSet {
  std::string arena;
  std::vector<std::pair<uint32_t, uint32_t>> items;   // (offset, length)
  std::vector<int32_t> slots;                         // index into items, -1 = empty
  static uint64_t hash(const uint8_t* p, size_t n) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)n;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; h ^= h >> 29; }
    return h;
  }
  void grow() {
    const size_t cap = slots.empty() ? 64 : slots.size() * 2;
    slots.assign(cap, -1);
    for (size_t k = 0; k < items.size(); ++k) {
      size_t i = (size_t)hash((const uint8_t*)arena.data() + items[k].first, items[k].second) & (cap - 1);
      while (slots[i] >= 0) i = (i + 1) & (cap - 1);
      slots[i] = (int32_t)k;
    }
  }
  bool has(const uint8_t* p, size_t n) const {
    if (slots.empty()) return false;
    const size_t cap = slots.size();
    for (size_t i = (size_t)hash(p, n) & (cap - 1); slots[i] >= 0; i = (i + 1) & (cap - 1)) {
      const auto& it = items[(size_t)slots[i]];
      if (it.second == n && memcmp(arena.data() + it.first, p, n) == 0) return true;
    }
    return false;
  }
  bool add(const uint8_t* p, size_t n) {   // false: was there already
    if (has(p, n)) return false;
    if ((items.size() + 1) * 2 > slots.size()) grow();
    items.emplace_back((uint32_t)arena.size(), (uint32_t)n);
    arena.append((const char*)p, n);
    const size_t cap = slots.size();
    size_t i = (size_t)hash(p, n) & (cap - 1);
    while (slots[i] >= 0) i = (i + 1) & (cap - 1);
    slots[i] = (int32_t)(items.size() - 1);
    return true;
  }
  size_t size() const { return items.size(); }
};

inline void join3(std::string& out, Span a, const char* sep1, Span b, const char* sep2, Span c) {
  out.assign((const char*)a.p, a.n);
  out.append(sep1);
  out.append((const char*)b.p, b.n);
  out.append(sep2);
  out.append((const char*)c.p, c.n);
}

}  // namespace

struct qm_patterns {
  ByteSet pats;        // X \t . \t Y \t Z of every truth row the awk program accepts
  ByteSet r_nokey;     // X \t Y \t Z of the rows R reads (no comments) whose X is no canonical position: the truth side of
                       // the text keys of QM_F_NOKEY lines
  int64_t n_exotic = 0, n_comment_only = 0, n_refused = 0;
  bool ext = false;
};

extern "C" qm_patterns* qm_patterns_create(const uint8_t* text, size_t len, int mode, int ext) {
  if ((!text && len) || (mode != 0 && mode != 1) || (ext && mode != 0)) return nullptr;
  qm_patterns* P = new qm_patterns();
  P->ext = ext != 0;
  auto allele_ok = [ext](Span f) { return ext ? acgt_all(f) : acgt1(f); };
  ByteSet device_keys, comment_keys;   // canonical patterns from rows R reads / from '#' rows
  std::string pat, rk;
  size_t off = 0;
  Span f[6];
  const Span empty = {(const uint8_t*)"", 0};
  while (off < len) {
    const uint8_t* s = text + off;
    const uint8_t* e = (const uint8_t*)memchr(s, '\n', len - off);
    const size_t n = e ? (size_t)(e - s) : len - off;
    off += n + 1;
    const int nf = split_tabs(s, n, f, 6);
    for (int i = nf < 6 ? nf : 6; i < 6; ++i) f[i] = empty;
    Span X, Y, Z;
    if (mode == 0) {   // awk -F"\t" '$4~/^[ACGT]$/&&$5~/^[ACGT]$/{print $2,".",$4,$5}'
      X = f[1]; Y = f[3]; Z = f[4];
      if (!allele_ok(Y) || !allele_ok(Z)) continue;
    } else {           // awk -F"\t" '$2!="."&&$3!="."{print $1,".",$2,$3}'
      X = f[0]; Y = f[1]; Z = f[2];
      if (is_dot(Y) || is_dot(Z)) continue;
    }
    join3(pat, X, "\t.\t", Y, "\t", Z);
    {   // the PATTERN's bytes: valid UTF-8 is text that grep compares bytewise; a NUL or an invalid sequence is refused
      bool high = false;
      for (size_t i = 0; i < pat.size() && !high; ++i) high = pat[i] == 0 || (uint8_t)pat[i] >= 0x80;
      if (high && !utf8_text((const uint8_t*)pat.data(), pat.size())) { ++P->n_refused; continue; }
    }
    const bool fresh = P->pats.add((const uint8_t*)pat.data(), pat.size());
    if (fresh && (ext ? !(acgt_all(Y) && acgt_all(Z)) : (Y.n != 1 || Z.n != 1))) ++P->n_exotic;
    const bool comment = n && s[0] == '#';
    int32_t p;
    const bool canon = canon_pos(X, &p) && allele_ok(Y) && allele_ok(Z);
    if (canon) (comment ? comment_keys : device_keys).add((const uint8_t*)pat.data(), pat.size());
    if (!comment && !canon_pos(X, &p) && (mode == 1 ? n > 0 : true)) {
      join3(rk, X, "\t", Y, "\t", Z);
      P->r_nokey.add((const uint8_t*)rk.data(), rk.size());
    }
  }
  for (const auto& it : comment_keys.items)
    if (!device_keys.has((const uint8_t*)comment_keys.arena.data() + it.first, it.second)) ++P->n_comment_only;
  return P;
}
extern "C" void qm_patterns_destroy(qm_patterns* p) { delete p; }
extern "C" int qm_patterns_info(const qm_patterns* p, int64_t* info) {
  if (!p || !info) return QM_E_INVAL;
  info[0] = (int64_t)p->pats.size(); info[1] = p->n_exotic; info[2] = p->n_comment_only; info[3] = p->n_refused;
  return QM_OK;
}

namespace {

// Does `fgrep -w` with the pattern list select the line?  Every pattern is X \t . \t Y \t Z with tab-free X, Y, Z and
// Y, Z != ".", so its first tab sits on a "\t.\t" of the line: X is a suffix of the field before it, Y the whole field
// after it, Z a prefix of the one after that.
bool fgrep_w_selects(const qm_patterns& P, const uint8_t* s, size_t n) {
  if (P.pats.size() == 0) return false;
  for (size_t j = 0; j + 2 < n; ++j) {
    if (s[j] != '\t' || s[j + 1] != '.' || s[j + 2] != '\t') continue;
    if (j == 0) continue;   // the pattern's own first tab follows X inside the line: a "." in the first field has no tab before it
    size_t fb = j;
    while (fb > 0 && s[fb - 1] != '\t') --fb;
    const size_t yb = j + 3;
    const uint8_t* t = yb < n ? (const uint8_t*)memchr(s + yb, '\t', n - yb) : nullptr;
    if (!t) break;   // no field after Y: no pattern fits here, nor on any later "\t.\t"
    const size_t zb = (size_t)(t - s) + 1;
    size_t ze = zb;
    while (ze < n && s[ze] != '\t') ++ze;
    for (size_t i = fb; i <= j; ++i) {
      if (i > fb && word_before(s, fb, i)) continue;    // i == fb: a tab or the line start precedes
      if (i < j && (s[i] & 0xc0) == 0x80) continue;     // (a match starts and ends on a character boundary)
      for (size_t e = zb; e <= ze; ++e) {
        if (e < ze && (word_at(s, e, ze) || (s[e] & 0xc0) == 0x80)) continue;      // e == ze: a tab or the line end follows
        if (P.pats.has(s + i, e - i)) return true;
      }
    }
  }
  return false;
}

}  // namespace

extern "C" int qm_vcf_hostpath(const qm_patterns* P, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                               uint8_t* line_kind, const int32_t* pos, const int32_t* ref, const int32_t* alt, uint8_t* flags,
                               int64_t* out) {
  if (!P || (!text && len) || !line_off || !line_kind || (!flags && n_lines)) return QM_E_INVAL;
  const bool full = P->n_exotic > 0 || P->n_comment_only > 0;   // the columns cannot stand in for this pattern list at all
  int64_t decided = 0, selected = 0;
  ByteSet text_keys;                         // X \t Y \t Z of kept QM_F_NOKEY lines
  std::vector<std::array<int32_t, 3>> trip;  // what the device tells such lines apart by: (carried pos, ref, alt)
  int64_t tp_r = 0, fp_r = 0;
  std::string rk;
  int64_t r = 0;
  for (int64_t i = 0; i < n_lines; ++i) {
    const uint8_t kind = line_kind[i];
    if (kind == QM_LINE_HEADER || kind == QM_LINE_HEADER_REFUSED) continue;
    size_t b = (size_t)line_off[i], e = (size_t)line_off[i + 1];
    if (e > len) e = len;
    if (e > b && text[e - 1] == '\n') --e;
    if (kind == QM_LINE_HEADER_KEPT || kind == QM_LINE_HEADER_KEPT_TP) {
      const bool sel = fgrep_w_selects(*P, text + b, e - b);
      line_kind[i] = sel ? QM_LINE_HEADER_KEPT_TP : QM_LINE_HEADER_KEPT;
      ++decided; selected += sel ? 1 : 0;
      continue;
    }
    const int64_t d = r++;
    if (kind == QM_LINE_DATA_HOST || (full && kind == QM_LINE_DATA)) {
      const bool sel = fgrep_w_selects(*P, text + b, e - b);
      flags[d] = (uint8_t)(sel ? (flags[d] | QM_F_TPLINE) : (flags[d] & ~(QM_F_IDDOT | QM_F_TPLINE)));
      ++decided; selected += sel ? 1 : 0;
    }
    if ((flags[d] & (QM_F_PASS | QM_F_NOKEY)) == (QM_F_PASS | QM_F_NOKEY)) {
      // R keys a line by the text of its POS / REF / ALT fields (caller_performance_compare.R:29-55)
      Span f[5];
      const Span empty = {(const uint8_t*)"", 0};
      const int nf = split_head(text + b, e - b, f, 5);
      for (int k = nf; k < 5; ++k) f[k] = empty;
      join3(rk, f[1], "\t", f[3], "\t", f[4]);
      if (text_keys.add((const uint8_t*)rk.data(), rk.size())) {
        if (P->r_nokey.has((const uint8_t*)rk.data(), rk.size())) ++tp_r; else ++fp_r;
      }
      if (pos && ref && alt) trip.push_back({pos[d], ref[d], alt[d]});
    }
  }
  std::sort(trip.begin(), trip.end());
  const int64_t dev_unique = (int64_t)(std::unique(trip.begin(), trip.end()) - trip.begin());
  if (out) { out[0] = decided; out[1] = selected; out[2] = dev_unique; out[3] = tp_r; out[4] = fp_r; }
  return QM_OK;
}

namespace {

// Gathers byte ranges of the input text into a file: neighbouring selected lines leave as ONE range, ranges go out in
// writev batches straight from the (mapped) input -- no staging copy in user space.  Small scattered ranges (the TP
// file: a few per cent of the lines, one line each) are packed into a buffer first, since a writev entry per 40-byte
// line costs more than copying it.
class RangeWriter {
 public:
  explicit RangeWriter(int fd) : fd_(fd) { iov_.reserve(kMaxIov); small_.reserve(kSmallCap); }
  bool add(const uint8_t* p, size_t n) {
    if (n == 0) return true;
    if (n < kSmallRange) {
      // entries pointing into small_ hold offsets until flush() resolves them (the buffer may still move)
      if (small_.size() + n > kSmallCap && !flush()) return false;
      const size_t at = small_.size();
      small_.insert(small_.end(), p, p + n);
      if (!iov_.empty() && small_tail_ && (size_t)(uintptr_t)iov_.back().iov_base + iov_.back().iov_len == at) {
        iov_.back().iov_len += n;
        return true;
      }
      push((void*)(uintptr_t)at, n, true);
    } else {
      push((void*)p, n, false);
    }
    return iov_.size() < kMaxIov || flush();
  }
  bool flush() {
    for (size_t k = 0; k < iov_.size(); ++k)
      if (is_small_[k]) iov_[k].iov_base = small_.data() + (size_t)(uintptr_t)iov_[k].iov_base;
    size_t k = 0;
    while (k < iov_.size()) {
      const ssize_t w = writev(fd_, &iov_[k], (int)std::min<size_t>(iov_.size() - k, kMaxIov));
      if (w < 0) { if (errno == EINTR) continue; return false; }
      size_t left = (size_t)w;
      while (left > 0 && k < iov_.size()) {
        if (left >= iov_[k].iov_len) { left -= iov_[k].iov_len; ++k; }
        else { iov_[k].iov_base = (char*)iov_[k].iov_base + left; iov_[k].iov_len -= left; left = 0; }
      }
    }
    iov_.clear(); is_small_.clear(); small_.clear(); small_tail_ = false;
    return true;
  }

 private:
  static constexpr size_t kMaxIov = 1024, kSmallRange = 512, kSmallCap = 1 << 20;
  void push(void* p, size_t n, bool small) {
    iovec v; v.iov_base = p; v.iov_len = n;
    iov_.push_back(v); is_small_.push_back(small); small_tail_ = small;
  }
  int fd_;
  std::vector<iovec> iov_;
  std::vector<char> is_small_;
  std::vector<uint8_t> small_;
  bool small_tail_ = false;
};

// header block, then the selected lines in input order; a missing final newline is added.
// cls_of(r) = QM_CLS_* bits of data line r.
template <typename ClsOf>
int write_selected(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off, const uint8_t* line_kind,
                   ClsOf cls_of, int select) {
  static std::atomic<unsigned> serial{0};
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
  if (fd < 0) return QM_E_IO;
  RangeWriter W(fd);
  static const uint8_t NL = '\n';
  bool ok = true;
  size_t run_b = 0, run_e = 0;   // pending contiguous range [run_b, run_e) of whole lines
  auto put = [&](int64_t i) {
    size_t b = (size_t)line_off[i], e = (size_t)line_off[i + 1];
    if (e > len) e = len;
    if (b >= e) { b = e; }
    const bool has_nl = e > b && text[e - 1] == '\n';
    if (run_e == b && run_e > run_b) run_e = e;
    else { if (run_e > run_b) ok = ok && W.add(text + run_b, run_e - run_b); run_b = b; run_e = e; }
    if (!has_nl) {   // only the last line of a text can lack its newline (or be empty without one)
      if (run_e > run_b) ok = ok && W.add(text + run_b, run_e - run_b);
      run_b = run_e = 0;
      ok = ok && W.add(&NL, 1);
    }
  };
  for (int64_t i = 0; i < n_lines && ok; ++i) {
    const uint8_t k = line_kind[i];
    if (k == QM_LINE_HEADER || k == QM_LINE_HEADER_KEPT || k == QM_LINE_HEADER_KEPT_TP || k == QM_LINE_HEADER_REFUSED) put(i);
  }
  int64_t r = 0;
  for (int64_t i = 0; i < n_lines && ok; ++i) {
    const uint8_t k = line_kind[i];
    if (k == QM_LINE_HEADER || k == QM_LINE_HEADER_REFUSED) continue;
    bool sel;
    if (k == QM_LINE_HEADER_KEPT || k == QM_LINE_HEADER_KEPT_TP) {   // awk printed it among the kept lines too
      sel = select == 0 || (select == 1) == (k == QM_LINE_HEADER_KEPT_TP);
    } else {
      const uint8_t c = cls_of(r);
      ++r;
      sel = select == 0 ? (c & QM_CLS_KEPT) : select == 1 ? ((c & 3u) == 3u) : ((c & 3u) == 1u);
    }
    if (sel) put(i);
  }
  if (ok && run_e > run_b) ok = W.add(text + run_b, run_e - run_b);
  ok = ok && W.flush();
  ok = (close(fd) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return QM_E_IO; }
  if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return QM_E_IO; }
  return QM_OK;
}

// The same from the class masks, 64 data lines at a time, for the plain shape of a VCF: every '#' line in front of the first
// data line.  Data line r is line h0 + r then, the selection of 64 lines is one word (kept, kept & tp, kept & ~tp), and the
// RUNS of selected lines fall out of it with count-trailing-zeros -- one range per run instead of a test per line (the per-line
// loop above was two thirds of this function's time: 20 ns a line, three files a VCF).  Returns QM_E_STATE for any other shape.
int write_selected_words(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off, const uint8_t* line_kind,
                         const uint64_t* kept, const uint64_t* tp, int select) {
  int64_t h0 = 0;
  while (h0 < n_lines && line_kind[h0] == QM_LINE_HEADER) ++h0;
  for (int64_t i = h0; i < n_lines; ++i) {
    const uint8_t k = line_kind[i];
    if (!(k == QM_LINE_DATA || k == QM_LINE_DATA_HOST || k == QM_LINE_REFUSED)) return QM_E_STATE;
  }
  const int64_t nd = n_lines - h0;
  static std::atomic<unsigned> serial{0};
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid()) + ".w" + std::to_string(serial.fetch_add(1));
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
  if (fd < 0) return QM_E_IO;
  RangeWriter W(fd);
  static const uint8_t NL = '\n';
  bool ok = true;
  const bool last_open = len > 0 && text[len - 1] != '\n';   // only the last line of a text can lack its newline
  // lines [a, b) of the text as one range; a missing final newline is added
  auto put_lines = [&](int64_t a, int64_t b) {
    if (b <= a) return;
    size_t pb = (size_t)line_off[a], pe = (size_t)line_off[b];
    if (pe > len) pe = len;
    if (pb < pe) ok = ok && W.add(text + pb, pe - pb);
    if (b == n_lines && last_open) ok = ok && W.add(&NL, 1);
  };
  put_lines(0, h0);   // the header block
  int64_t rb = 0, re = 0;   // pending run of selected data lines [rb, re)
  const int64_t nw = (nd + 63) >> 6;
  for (int64_t k = 0; k < nw && ok; ++k) {
    uint64_t w = select == 0 ? kept[k] : select == 1 ? (kept[k] & tp[k]) : (kept[k] & ~tp[k]);
    if (k == nw - 1 && (nd & 63)) w &= (1ull << (nd & 63)) - 1ull;   // (bits beyond the VCF are clear anyway)
    while (w) {
      const int s0 = __builtin_ctzll(w);
      const uint64_t inv = ~(w >> s0);
      const int n1 = inv ? __builtin_ctzll(inv) : 64 - s0;       // ones from s0 on (inv == 0: they fill the shifted word)
      const int64_t start = (k << 6) + s0;
      if (start == re && re > rb) re += n1;
      else { put_lines(h0 + rb, h0 + re); rb = start; re = start + n1; }
      if (s0 + n1 >= 64) break;
      w &= ~(((1ull << n1) - 1ull) << s0);
    }
  }
  if (ok) put_lines(h0 + rb, h0 + re);
  ok = ok && W.flush();
  ok = (close(fd) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return QM_E_IO; }
  if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return QM_E_IO; }
  return QM_OK;
}

}  // namespace

extern "C" int qm_vcf_write(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                            const uint8_t* line_kind, const uint8_t* cls, int select) {
  if (!path || !line_off || !line_kind || select < 0 || select > 2 || (!text && len)) return QM_E_INVAL;
  return write_selected(path, text, len, n_lines, line_off, line_kind, [cls](int64_t r) -> uint8_t { return cls ? cls[r] : 0; }, select);
}

// internal (qmvt_pipeline.cpp): the same from the class masks as they come back from the device (bit r of word r / 64),
// or, without masks (pure-strain samples never reach the device), from the A2 verdict in the flags column
int qm_host_write_masks(const char* path, const uint8_t* text, size_t len, int64_t n_lines, const int64_t* line_off,
                        const uint8_t* line_kind, const uint64_t* kept, const uint64_t* tp, const uint8_t* flags, int select) {
  if (!path || !line_off || !line_kind || select < 0 || select > 2 || (!text && len)) return QM_E_INVAL;
  if (kept && tp) {
    const int rc = write_selected_words(path, text, len, n_lines, line_off, line_kind, kept, tp, select);
    if (rc != QM_E_STATE) return rc;   // (QM_E_STATE: not the plain shape -- header lines behind the first data line: line by line)
    return write_selected(path, text, len, n_lines, line_off, line_kind, [kept, tp](int64_t r) -> uint8_t {
      return (uint8_t)(((kept[r >> 6] >> (r & 63)) & 1u) | (((tp[r >> 6] >> (r & 63)) & 1u) << 1)); }, select);
  }
  return write_selected(path, text, len, n_lines, line_off, line_kind, [flags](int64_t r) -> uint8_t { return flags ? (flags[r] & QM_F_PASS) : 0; }, select);
}

// ---------------------------------------------------------------------------
// BGZF (what `bgzip -c` writes in rules/vis_eval_vcf.smk:36,51,67,82): a series of gzip members of at most 64 KiB, each
// carrying its compressed size in a 'BC' extra field, closed by the fixed empty EOF member.  Readable by zcat / gzip and
// by tabix / htslib.  (SAM/BAM specification, section 4.1.)
// ---------------------------------------------------------------------------
namespace {

const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// one member from at most 0xff00 input bytes; returns the member size, 0 on error
size_t bgzf_block(const uint8_t* in, size_t n, int level, uint8_t* out /* >= 0x10000 */) {
  static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
  memcpy(out, head, 16);
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
  zs.next_in = const_cast<Bytef*>(in);
  zs.avail_in = (uInt)n;
  zs.next_out = out + 18;
  zs.avail_out = 0x10000 - 18 - 8;
  const int rc = deflate(&zs, Z_FINISH);
  const size_t clen = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return 0;   // cannot happen for <= 0xff00 bytes: deflate's worst case fits
  const size_t total = 18 + clen + 8;
  const uint16_t bsize = (uint16_t)(total - 1);
  out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n);
  const uint32_t isz = (uint32_t)n;
  for (int k = 0; k < 4; ++k) { out[18 + clen + k] = (uint8_t)(crc >> (8 * k)); out[22 + clen + k] = (uint8_t)(isz >> (8 * k)); }
  return total;
}

}  // namespace

// ---------------------------------------------------------------------------
// Tabix index of a BGZF-compressed VCF (`tabix -p vcf`, rules/vis_eval_vcf.smk:37,52,68,83; format: the tabix paper's
// supplement / htslib's tbx.c + hts.c, restated): UCSC binning with a 16 kb linear index (min_shift 14, 5 levels) over
// BGZF virtual offsets (compressed offset of the block << 16 | offset inside the block).  Per data line: sequence = column 1,
// begin = POS - 1, end = begin + len(REF), or INFO's END= when that lies behind begin.  Records of one bin that follow each
// other make one chunk [offset of the first, offset behind the last); a window of the linear index holds the offset of the
// first record that overlaps it, empty windows take their successor's.  Every sequence also carries htslib's pseudo-bin
// 37450 (file range of its records; record count).  The index itself is BGZF-compressed.
// Like tabix, the writer refuses a VCF whose sequences do not come in blocks or whose positions step backwards.
// ---------------------------------------------------------------------------
namespace {

inline uint32_t tbi_reg2bin(int64_t beg, int64_t end) {
  int l, s = 14;
  int64_t t = ((1 << 15) - 1) / 7;
  for (--end, l = 5; l > 0; --l, s += 3, t -= (int64_t)1 << (3 * l))
    if ((beg >> s) == (end >> s)) return (uint32_t)(t + (beg >> s));
  return 0;
}

struct TbiRef {
  std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
  std::vector<uint64_t> lin;
  uint64_t off_beg = 0, off_end = 0, n = 0;
};

void put32(std::vector<uint8_t>& o, uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put64(std::vector<uint8_t>& o, uint64_t v) { for (int k = 0; k < 8; ++k) o.push_back((uint8_t)(v >> (8 * k))); }

// the index of `data` as it lies in a BGZF file cut into members of 0xff00 input bytes whose compressed offsets are coff[]
// (coff[number of members] = where the EOF member starts).  QM_OK, QM_E_UNSORTED, QM_E_RANGE (a coordinate tabix cannot bin).
int tbi_build(const uint8_t* data, size_t len, const std::vector<uint64_t>& coff, std::vector<uint8_t>& out) {
  auto voff = [&](size_t u) { return (coff[u / 0xff00] << 16) | (uint64_t)(u % 0xff00); };
  std::vector<std::string> names;
  std::vector<TbiRef> refs;
  std::string last_name;
  int tid = -1;
  int64_t last_beg = -1;
  uint32_t last_bin = 0xffffffffu;
  size_t b = 0;
  while (b < len) {
    const uint8_t* nl = (const uint8_t*)memchr(data + b, '\n', len - b);
    const size_t e = nl ? (size_t)(nl - data) : len;          // line = [b, e), the record ends behind its newline
    const size_t next = nl ? e + 1 : len;
    if (e > b && data[b] != '#') {
      Span f[8] = {};
      const int nf = split_head(data + b, e - b, f, 8);
      if (nf < 2 || f[0].n == 0) return QM_E_INVAL;            // tabix: "failed to parse" a line without its columns
      int64_t pos = 0;
      {
        size_t k = 0;
        for (; k < f[1].n && f[1].p[k] >= '0' && f[1].p[k] <= '9' && pos < ((int64_t)1 << 40); ++k) pos = pos * 10 + (f[1].p[k] - '0');
        if (k == 0) return QM_E_INVAL;
      }
      const int64_t beg = pos - 1;
      int64_t end = beg + (nf >= 4 && f[3].n ? (int64_t)f[3].n : 1);
      if (nf >= 8) {   // INFO: END= at its start or behind a ';'
        const uint8_t* s = f[7].p;
        const size_t n = f[7].n;
        for (size_t k = 0; k + 4 <= n; ++k) {
          if ((k == 0 || s[k - 1] == ';') && memcmp(s + k, "END=", 4) == 0) {
            int64_t v = 0;
            size_t q = k + 4;
            for (; q < n && s[q] >= '0' && s[q] <= '9' && v < ((int64_t)1 << 40); ++q) v = v * 10 + (s[q] - '0');
            if (q > k + 4 && (q == n || s[q] == ';') && v > beg) end = v;
            break;
          }
        }
      }
      if (beg < 0 || end > ((int64_t)1 << 29)) return QM_E_RANGE;   // 2^(14 + 3 * 5): the reach of this binning scheme
      const std::string name((const char*)f[0].p, f[0].n);
      if (tid < 0 || name != last_name) {
        for (const auto& nm : names) if (nm == name) return QM_E_UNSORTED;   // "chromosome blocks not continuous"
        names.push_back(name); refs.emplace_back();
        tid = (int)names.size() - 1; last_name = name; last_beg = -1; last_bin = 0xffffffffu;
        refs[(size_t)tid].off_beg = voff(b);
      }
      if (beg < last_beg) return QM_E_UNSORTED;                                // "unsorted positions"
      last_beg = beg;
      TbiRef& R = refs[(size_t)tid];
      const uint64_t o0 = voff(b), o1 = voff(next);
      const uint32_t bin = tbi_reg2bin(beg, end);
      if (bin == last_bin) R.bins[bin].back().second = o1;
      else { R.bins[bin].emplace_back(o0, o1); last_bin = bin; }
      const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
      if (R.lin.size() < w1 + 1) R.lin.resize(w1 + 1, ~0ull);
      for (size_t w = w0; w <= w1; ++w) if (R.lin[w] == ~0ull) R.lin[w] = o0;
      R.off_end = o1; R.n += 1;
    }
    b = next;
  }
  out.clear();
  const uint8_t magic[4] = {'T', 'B', 'I', 1};
  out.insert(out.end(), magic, magic + 4);
  put32(out, (uint32_t)names.size());
  put32(out, 2u);          // format: VCF
  put32(out, 1u); put32(out, 2u); put32(out, 0u);   // sequence, begin, end columns
  put32(out, (uint32_t)'#'); put32(out, 0u);        // comment character, lines to skip
  size_t l_nm = 0;
  for (const auto& nm : names) l_nm += nm.size() + 1;
  put32(out, (uint32_t)l_nm);
  for (const auto& nm : names) { out.insert(out.end(), nm.begin(), nm.end()); out.push_back(0); }
  for (TbiRef& R : refs) {
    for (size_t w = R.lin.size(); w-- > 1;) if (R.lin[w - 1] == ~0ull) R.lin[w - 1] = R.lin[w];
    put32(out, (uint32_t)R.bins.size() + 1u);
    for (const auto& kv : R.bins) {
      put32(out, kv.first); put32(out, (uint32_t)kv.second.size());
      for (const auto& c : kv.second) { put64(out, c.first); put64(out, c.second); }
    }
    put32(out, 37450u); put32(out, 2u);               // htslib's pseudo-bin: the file range of the records, their number
    put64(out, R.off_beg); put64(out, R.off_end); put64(out, R.n); put64(out, 0ull);
    put32(out, (uint32_t)R.lin.size());
    for (uint64_t v : R.lin) put64(out, v);
  }
  put64(out, 0ull);        // records without coordinates
  return QM_OK;
}

// `data` as a series of BGZF members + the EOF member into a file; coff (optional) receives the members' offsets
bool bgzf_stream(FILE* fh, const uint8_t* data, size_t len, int level, std::vector<uint64_t>* coff) {
  std::vector<uint8_t> blk(0x10000);
  uint64_t at = 0;
  for (size_t off = 0; off < len; off += 0xff00) {
    const size_t n = std::min<size_t>(0xff00, len - off);
    const size_t m = bgzf_block(data + off, n, level, blk.data());
    if (coff) coff->push_back(at);
    if (m == 0 || fwrite(blk.data(), 1, m, fh) != m) return false;
    at += m;
  }
  if (coff) coff->push_back(at);
  return fwrite(kBgzfEof, 1, sizeof kBgzfEof, fh) == sizeof kBgzfEof;
}

int bgzf_file(const std::string& path, const uint8_t* data, size_t len, int level, std::vector<uint64_t>* coff) {
  static std::atomic<unsigned> serial{0};
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
  FILE* fh = fopen(tmp.c_str(), "wb");
  if (!fh) return QM_E_IO;
  bool ok = bgzf_stream(fh, data, len, level, coff);
  ok = (fclose(fh) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return QM_E_IO; }
  if (rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); return QM_E_IO; }
  return QM_OK;
}

}  // namespace

extern "C" int qm_bgzf_write(const char* path, const uint8_t* data, size_t len, int level) {
  if (!path || (!data && len) || level < -1 || level > 9) return QM_E_INVAL;
  return bgzf_file(path, data, len, level, nullptr);
}

// bgzip + tabix -p vcf in one call: <path> and <path>.tbi.  The index is built (and the order of the VCF checked) before
// anything is written: QM_E_UNSORTED leaves no file behind.
extern "C" int qm_bgzf_write_tbi(const char* path, const uint8_t* data, size_t len, int level) {
  if (!path || (!data && len) || level < -1 || level > 9) return QM_E_INVAL;
  // both files are complete under names of their own (process id + a serial number: two threads may write the same path) before
  // either appears: the .gz first, then its index -- a reader never finds an index beside an older or missing .gz; if the
  // second rename fails, the index that no longer belongs to the new .gz is removed
  static std::atomic<unsigned> serial{0};
  const std::string gz(path), tag = "." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
  const std::string tmp = gz + ".tbitmp" + tag, tmpi = gz + ".tbi.tbitmp" + tag;
  std::vector<uint64_t> coff;
  int rc = bgzf_file(tmp, data, len, level, &coff);   // (the compressed sizes are what the virtual offsets are made of)
  if (rc != QM_OK) return rc;
  std::vector<uint8_t> idx;
  rc = tbi_build(data, len, coff, idx);
  if (rc == QM_OK) rc = bgzf_file(tmpi, idx.data(), idx.size(), level, nullptr);
  if (rc != QM_OK) { remove(tmp.c_str()); remove(tmpi.c_str()); return rc; }
  if (rename(tmp.c_str(), gz.c_str()) != 0) { remove(tmp.c_str()); remove(tmpi.c_str()); return QM_E_IO; }
  if (rename(tmpi.c_str(), (gz + ".tbi").c_str()) != 0) { remove(tmpi.c_str()); remove((gz + ".tbi").c_str()); return QM_E_IO; }
  return QM_OK;
}

// ---------------------------------------------------------------------------
// SNP / indel splitters (rules/vis_eval_vcf.smk:35,50,66,81): awk programs with two rules,
//   /^#.*/{print}   and   <allele pattern>     (no `next`, so a '#' line that also satisfies
// the allele pattern is printed twice), output in input order.
//   mode 0 (xsnp):   $4 ~ /^[actgACTG]$/ && $5 ~ /^[actgACTG]$/
//   mode 1 (xindel): $4 ~ /^[actgACTG]{2,}/ || $5 ~ /^[actgACTG]{2,}/
// flavour 0: `{2,}` is a POSIX interval (gawk, mawk >= 1.3.4-2020xxxx built with repetitions);
// flavour 1: `{2,}` is the literal text (mawk 1.3.4 20200120, the awk of the build image).
// ---------------------------------------------------------------------------
static inline bool base8(uint8_t c) {
  switch (c) { case 'a': case 'c': case 't': case 'g': case 'A': case 'C': case 'T': case 'G': return true; default: return false; }
}
static inline bool indel_field(const uint8_t* f, size_t n, int flavour) {
  if (flavour == 0) return n >= 2 && base8(f[0]) && base8(f[1]);
  return n >= 5 && base8(f[0]) && f[1] == '{' && f[2] == '2' && f[3] == ',' && f[4] == '}';
}

extern "C" int qm_vcf_split_write(const char* path, const uint8_t* text, size_t len, int mode, int flavour, int64_t* n_written) {
  if (!path || (!text && len) || mode < 0 || mode > 1 || flavour < 0 || flavour > 1) return QM_E_INVAL;
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  FILE* fh = fopen(tmp.c_str(), "wb");
  if (!fh) return QM_E_IO;
  std::vector<char> buf;
  buf.reserve(1 << 20);
  bool ok = true;
  int64_t nw = 0;
  size_t b = 0;
  while (b < len && ok) {
    const uint8_t* nl = (const uint8_t*)memchr(text + b, '\n', len - b);
    const size_t e = nl ? (size_t)(nl - text) : len;
    // fields 4 and 5 of the tab-split record
    const uint8_t* f[2] = {nullptr, nullptr};
    size_t fl[2] = {0, 0};
    size_t q = b;
    for (int k = 1; k <= 5 && q <= e; ++k) {
      const uint8_t* t = (q < e) ? (const uint8_t*)memchr(text + q, '\t', e - q) : nullptr;
      const size_t fe = t ? (size_t)(t - text) : e;
      if (k >= 4) { f[k - 4] = text + q; fl[k - 4] = fe - q; }
      if (!t) break;
      q = fe + 1;
    }
    int times = (e > b && text[b] == '#') ? 1 : 0;
    if (mode == 0) times += (fl[0] == 1 && base8(f[0][0]) && fl[1] == 1 && base8(f[1][0])) ? 1 : 0;
    else times += ((f[0] && indel_field(f[0], fl[0], flavour)) || (f[1] && indel_field(f[1], fl[1], flavour))) ? 1 : 0;
    for (int t = 0; t < times; ++t) {
      buf.insert(buf.end(), (const char*)text + b, (const char*)text + e);
      buf.push_back('\n');
      ++nw;
    }
    if (buf.size() >= (1u << 20)) { ok = fwrite(buf.data(), 1, buf.size(), fh) == buf.size(); buf.clear(); }
    b = e + 1;
  }
  if (ok && !buf.empty()) ok = fwrite(buf.data(), 1, buf.size(), fh) == buf.size();
  ok = (fclose(fh) == 0) && ok;
  if (!ok) { remove(tmp.c_str()); return QM_E_IO; }
  if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return QM_E_IO; }
  if (n_written) *n_written = nw;
  return QM_OK;
}

// ---------------------------------------------------------------------------
// Truth-set builder: MUMmer `show-snps -CTHlr` table -> truth VCF, what rules/genome_diff.smk:22-24
// runs as `mummer2vcf.py -s <table> --output-header -n -g <ref.fa>`.  Restated from the text of the reference's
// program/mummer2vcf.py (Biopython is not in the build image, so the reference itself cannot run: PARITY UNPINNED, hand-derived
// cases only; quasimodo_amd/mummer2vcf.py holds the same restatement in Python and the tests compare the two on random tables):
//   * one VCF row per table row: CHROM = reference tag (column 11), POS = P1 (column 1), REF / ALT = the two SUB columns, QUAL 30,
//     FILTER PASS, INFO DP=30;REF1=..;REF2=..                                                            (mummer2vcf.py:69-85)
//   * -n: rows with an N / n in REF or ALT are dropped                                                   (:88-100)
//   * single-base, dot-free REF and ALT -> SNV, anything else INDEL                                      (:103-118)
//   * SNVs ordered by position (stable); a row at the position of the row before it folds its ALT into the kept row's comma
//     list (no duplicates)                                                                               (:122-144, :242-252)
//   * indels keep input order; a row continues the one before it when it is an insertion at the same position or a deletion at
//     the next position: inserted bases are appended to every allele (another query position) or add an alternative allele whose
//     last character is replaced (the same query position); deleted bases are appended to REF           (:147-185)
//   * every indel row gets the reference base in front of it as anchor and its POS moves one to the left (:188-210)
//   * rows ordered by POS as TEXT, then (stable) by (CHROM, POS as a number): SNVs in front of indels at equal keys; INFO gains
//     ;ORIG=<query tag>:<P2>;TYPE=SNV|INDEL; eight columns                                               (:277-313)
//   * --output-header: VCFv4.2 header with the contigs that carry variants, in FASTA order               (:320-357)
// ---------------------------------------------------------------------------
namespace {

struct M2vRow {
  std::string chrom, ref, alt, info, orig;
  long long pos = 0;
  bool snv = false;
};

// text -> lines the way Python's universal-newlines text mode hands them out (\n, \r\n and a lone \r end a line)
void m2v_lines(const uint8_t* t, size_t n, std::vector<std::string>& out) {
  size_t b = 0;
  for (size_t i = 0; i < n; ++i) {
    if (t[i] == '\n' || t[i] == '\r') {
      out.emplace_back((const char*)t + b, i - b);
      if (t[i] == '\r' && i + 1 < n && t[i + 1] == '\n') ++i;
      b = i + 1;
    }
  }
  if (b < n) out.emplace_back((const char*)t + b, n - b);
}
std::vector<std::string> m2v_split(const std::string& s, char sep) {
  std::vector<std::string> v;
  size_t b = 0;
  for (;;) {
    const size_t e = s.find(sep, b);
    if (e == std::string::npos) { v.push_back(s.substr(b)); break; }
    v.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return v;
}
std::string m2v_join(const std::vector<std::string>& v, const char* sep) {
  std::string o;
  for (size_t i = 0; i < v.size(); ++i) { if (i) o += sep; o += v[i]; }
  return o;
}
bool m2v_int(const std::string& s, long long* out) {   // a decimal integer with an optional sign, blanks around it allowed (Python's int())
  size_t b = 0, e = s.size();
  while (b < e && (s[b] == ' ' || s[b] == '\t')) ++b;
  while (e > b && (s[e - 1] == ' ' || s[e - 1] == '\t')) --e;
  if (b == e) return false;
  bool neg = false;
  if (s[b] == '+' || s[b] == '-') { neg = s[b] == '-'; ++b; }
  if (b == e || e - b > 18) return false;
  long long v = 0;
  for (size_t i = b; i < e; ++i) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (s[i] - '0'); }
  *out = neg ? -v : v;
  return true;
}

}  // namespace

extern "C" int qm_mummer2vcf(const uint8_t* table, size_t table_len, const uint8_t* fasta, size_t fasta_len, const char* reference_name,
                             unsigned flags, const char* file_date, uint8_t** out, size_t* out_len) {
  if (!out || !out_len || (!table && table_len) || (!fasta && fasta_len) || (flags & ~31u) || ((flags & 8u) && (flags & 16u))) return QM_E_INVAL;
  *out = nullptr; *out_len = 0;
  const bool no_ns = flags & 1u, header = flags & 2u, in_header = flags & 4u, only_snp = flags & 8u, only_indel = flags & 16u;
  // ---- the FASTA: first word of every '>' line -> its sequence, in file order (a name seen again keeps its place, takes the new sequence)
  std::vector<std::pair<std::string, std::string>> seqs;
  std::map<std::string, size_t> seq_of;
  {
    std::vector<std::string> L;
    if (fasta_len) m2v_lines(fasta, fasta_len, L);
    long cur = -1;
    auto strip = [](const std::string& s) {
      size_t b = 0, e = s.size();
      auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
      while (b < e && ws(s[b])) ++b;
      while (e > b && ws(s[e - 1])) --e;
      return s.substr(b, e - b);
    };
    for (const std::string& ln : L) {
      if (!ln.empty() && ln[0] == '>') {
        const std::string h = strip(ln.substr(1));
        size_t e = 0;
        while (e < h.size() && !(h[e] == ' ' || h[e] == '\t' || h[e] == '\f' || h[e] == '\v')) ++e;
        const std::string name = h.substr(0, e);
        auto it = seq_of.find(name);
        if (it == seq_of.end()) { seq_of[name] = seqs.size(); cur = (long)seqs.size(); seqs.emplace_back(name, std::string()); }
        else { cur = (long)it->second; seqs[(size_t)cur].second.clear(); }
      } else if (cur >= 0) {
        seqs[(size_t)cur].second += strip(ln);
      }
    }
  }
  // ---- the table
  std::vector<M2vRow> snvs, indels;
  {
    std::vector<std::string> L;
    if (table_len) m2v_lines(table, table_len, L);
    for (size_t i = in_header ? 4 : 0; i < L.size(); ++i) {
      std::string ln = L[i];
      if (ln.empty()) continue;
      while (!ln.empty() && (ln.back() == '\n' || ln.back() == '\r' || ln.back() == '\b')) ln.pop_back();
      const std::vector<std::string> c = m2v_split(ln, '\t');
      if (c.size() < 12) return QM_E_INVAL;
      M2vRow r;
      if (!m2v_int(c[0], &r.pos)) return QM_E_INVAL;
      r.ref = c[1]; r.alt = c[2]; r.chrom = c[10];
      if (no_ns && (r.ref.find_first_of("Nn") != std::string::npos || r.alt.find_first_of("Nn") != std::string::npos)) continue;
      r.snv = r.ref.size() == 1 && r.ref != "." && r.alt.size() == 1 && r.alt != ".";
      r.info = "DP=30;REF1=" + c[10] + ";REF2=" + c[11];
      r.orig = c[11] + ":" + c[3];
      (r.snv ? snvs : indels).push_back(std::move(r));
    }
  }
  // ---- SNVs: by position (stable), rows at the position of the row before them folded into it
  {
    std::stable_sort(snvs.begin(), snvs.end(), [](const M2vRow& a, const M2vRow& b) { return a.pos < b.pos; });
    std::vector<M2vRow> o;
    long long prev = 0;
    bool have = false;
    for (M2vRow& r : snvs) {
      if (have && !o.empty() && r.pos == prev) {
        std::vector<std::string> alts = m2v_split(o.back().alt, ',');
        if (std::find(alts.begin(), alts.end(), r.alt) == alts.end()) { alts.push_back(r.alt); o.back().alt = m2v_join(alts, ","); }
      } else {
        o.push_back(r);
      }
      prev = r.pos; have = true;
    }
    snvs.swap(o);
  }
  // ---- indels: runs merged in input order, then the anchor base
  if (!indels.empty()) {
    std::vector<M2vRow> o;
    long long prev_pos = 0;
    std::string prev_orig;
    bool have = false;
    for (M2vRow& r : indels) {
      const bool cont = have && !o.empty() && ((r.pos == prev_pos && r.ref == ".") || (r.pos == prev_pos + 1 && r.alt == "."));
      if (cont) {
        M2vRow& last = o.back();
        if (r.ref == ".") {
          if (r.orig != prev_orig) {
            std::vector<std::string> alts = m2v_split(last.alt, ',');
            for (std::string& a : alts) a += r.alt;
            last.alt = m2v_join(alts, ",");
          } else {
            last.alt = last.alt + "," + last.alt.substr(0, last.alt.size() - 1) + r.alt;
          }
        } else if (r.alt == ".") {
          last.ref += r.ref;
        }
      } else {
        o.push_back(r);
      }
      prev_pos = r.pos; prev_orig = r.orig; have = true;
    }
    for (M2vRow& r : o) {
      auto it = seq_of.find(r.chrom);
      if (it == seq_of.end()) return QM_E_INVAL;              // (the reference: KeyError)
      const std::string& sq = seqs[it->second].second;
      long long i = r.pos - 2;                                // the base in front of the variant; Python's index: -1 wraps to the last base
      if (i < 0) i += (long long)sq.size();
      if (i < 0 || i >= (long long)sq.size()) return QM_E_RANGE;
      const std::string base(1, sq[(size_t)i]);
      if (r.ref == ".") {
        std::vector<std::string> alts = m2v_split(r.alt, ',');
        for (std::string& a : alts) a = base + a;
        r.ref = base; r.alt = m2v_join(alts, ",");
      } else if (r.alt == ".") {
        r.ref = base + r.ref; r.alt = base;
      }
      r.pos -= 1;
    }
    indels.swap(o);
  }
  // ---- order: POS as text, then (stable) (CHROM, POS); the type filter between the two as in the reference
  std::vector<const M2vRow*> all;
  for (const M2vRow& r : snvs) all.push_back(&r);
  for (const M2vRow& r : indels) all.push_back(&r);
  std::stable_sort(all.begin(), all.end(), [](const M2vRow* a, const M2vRow* b) { return std::to_string(a->pos) < std::to_string(b->pos); });
  if (only_snp || only_indel) {
    std::vector<const M2vRow*> f;
    for (const M2vRow* r : all) if (r->snv == (bool)only_snp) f.push_back(r);
    all.swap(f);
  }
  std::stable_sort(all.begin(), all.end(), [](const M2vRow* a, const M2vRow* b) { return a->chrom != b->chrom ? a->chrom < b->chrom : a->pos < b->pos; });
  std::string o;
  if (header) {
    char date[16] = "";
    if (file_date) snprintf(date, sizeof date, "%.8s", file_date);
    else { const time_t now = time(nullptr); struct tm tmv; localtime_r(&now, &tmv); strftime(date, sizeof date, "%Y%m%d", &tmv); }
    o += "##fileformat=VCFv4.2\n##fileDate=" + std::string(date) + "\n##source=mummer2vcf.py\n##reference=" + std::string(reference_name ? reference_name : "None") + "\n";
    std::set<std::string> used;
    for (const M2vRow* r : all) used.insert(r->chrom);
    for (const auto& sq : seqs) if (used.count(sq.first)) o += "##contig=<ID=" + sq.first + ",length=" + std::to_string(sq.second.size()) + ">\n";
    o += "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total depth of quality bases\">\n"
         "##INFO=<ID=REF1,Number=1,Type=String,Description=\"The name of the 1st reference sequence\">\n"
         "##INFO=<ID=REF2,Number=1,Type=String,Description=\"The name of the 2nd reference sequence\">\n"
         "##INFO=<ID=ORIG,Number=1,Type=String,Description=\"The original position of variant at 2nd reference sequence\">\n"
         "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"Indicates that the variant is an INDEL or SNV.\">\n"
         "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
  }
  for (const M2vRow* r : all)
    o += r->chrom + "\t" + std::to_string(r->pos) + "\t.\t" + r->ref + "\t" + r->alt + "\t30\tPASS\t" + r->info + ";ORIG=" + r->orig + ";TYPE=" + (r->snv ? "SNV" : "INDEL") + "\n";
  uint8_t* buf = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!buf) return QM_E_NOMEM;
  memcpy(buf, o.data(), o.size());
  *out = buf; *out_len = o.size();
  return QM_OK;
}
extern "C" void qm_free(void* p) { free(p); }
