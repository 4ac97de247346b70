// Host side of the input pipeline (plain C ABI, no GPU): include/geeco_host.h.
//
// What tf.data's worker threads do when the reference reads an episode (src/data/geeco_gym.py:442-445
// TFRecordDataset(ZLIB, num_parallel_reads); :291-315 _parse_v4 = tf.parse_single_sequence_example + rgb / 255):
//   file -> inflate -> record framing + masked CRC-32C -> SequenceExample field scan -> dense arrays,
// here without a Python lock held, so `num_threads` readers run side by side (geeco_amd/input_fn.py).
// The writer of the format is src/data/data_recorder.py:37-59,134-156 + src/data/utils/tfrecord.py:42-81
// (uint8 images stored as FLOAT lists: an RGB frame is 196 608 floats).
#include "../../include/geeco_host.h"
#include "host_inflate.h"

#include <errno.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#include <nmmintrin.h>
#endif

// ---------------------------------------------------------------------------------------------------------------------
// errors (per calling thread)
// ---------------------------------------------------------------------------------------------------------------------
static thread_local char t_err[512] = "";

static void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* geeco_host_last_error(void) { return t_err; }
extern "C" int geeco_host_abi_version(void) { return GEECO_HOST_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------------------------------
// CRC-32C
// ---------------------------------------------------------------------------------------------------------------------
static uint32_t g_tab[8][256];
static volatile int g_init = 0;

static void init_tables() {
  const uint32_t poly = 0x82f63b78u;   // reflected CRC-32C polynomial
  uint32_t tab[8][256];
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ poly : c >> 1;
    tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = tab[0][i];
    for (int t = 1; t < 8; ++t) {
      c = tab[0][c & 0xff] ^ (c >> 8);
      tab[t][i] = c;
    }
  }
  memcpy(g_tab, tab, sizeof(tab));   // racing threads write identical values
  __sync_synchronize();
  g_init = 1;
}

static uint32_t crc32c_sw(const uint8_t* p, size_t n, uint32_t crc) {
  if (!g_init) init_tables();
  while (n && ((uintptr_t)p & 7)) {
    crc = g_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
    --n;
  }
  while (n >= 8) {   // slicing-by-8
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= crc;
    crc = g_tab[7][v & 0xff] ^ g_tab[6][(v >> 8) & 0xff] ^ g_tab[5][(v >> 16) & 0xff] ^ g_tab[4][(v >> 24) & 0xff] ^
          g_tab[3][(v >> 32) & 0xff] ^ g_tab[2][(v >> 40) & 0xff] ^ g_tab[1][(v >> 48) & 0xff] ^ g_tab[0][v >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) crc = g_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
  return crc;
}

#if defined(__x86_64__)
// the SSE4.2 crc32 instruction implements exactly this polynomial; three independent streams hide its 3-cycle latency
// only with a carry-less-multiply recombination, which a 105 MB episode does not need: one stream runs at ~8 B / 3 cycles
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const uint8_t* p, size_t n, uint32_t crc) {
  uint64_t c = crc;
  while (n && ((uintptr_t)p & 7)) {
    c = _mm_crc32_u8((uint32_t)c, *p++);
    --n;
  }
  while (n >= 32) {
    uint64_t a, b, d, e;
    memcpy(&a, p, 8); memcpy(&b, p + 8, 8); memcpy(&d, p + 16, 8); memcpy(&e, p + 24, 8);
    c = _mm_crc32_u64(c, a);
    c = _mm_crc32_u64(c, b);
    c = _mm_crc32_u64(c, d);
    c = _mm_crc32_u64(c, e);
    p += 32;
    n -= 32;
  }
  while (n >= 8) {
    uint64_t a;
    memcpy(&a, p, 8);
    c = _mm_crc32_u64(c, a);
    p += 8;
    n -= 8;
  }
  while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
  return (uint32_t)c;
}
#endif

extern "C" uint32_t geeco_crc32c(const uint8_t* p, size_t n, uint32_t crc) {
  crc = ~crc;
#if defined(__x86_64__)
  static const int hw = __builtin_cpu_supports("sse4.2");
  crc = hw ? crc32c_hw(p, n, crc) : crc32c_sw(p, n, crc);
#else
  crc = crc32c_sw(p, n, crc);
#endif
  return ~crc;
}

// TFRecord stores masked CRCs: ((crc >> 15) | (crc << 17)) + 0xa282ead8
extern "C" uint32_t geeco_masked_crc32c(const uint8_t* p, size_t n) {
  uint32_t c = geeco_crc32c(p, n, 0);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ---------------------------------------------------------------------------------------------------------------------
// inflate
// ---------------------------------------------------------------------------------------------------------------------
struct Buf {
  uint8_t* p = nullptr;
  size_t n = 0, cap = 0;
  ~Buf() { free(p); }
  bool reserve(size_t want) {
    if (want <= cap) return true;
    uint8_t* q = (uint8_t*)realloc(p, want);    // large blocks are mmap'ed: growing them is an mremap, not a copy
    if (!q) return false;
    p = q;
    cap = want;
    return true;
  }
};

static int g_fast_inflate = 1;          // geeco_host_set_fast_inflate(0): zlib only (A/B, tests)
extern "C" void geeco_host_set_fast_inflate(int on) { __atomic_store_n(&g_fast_inflate, on ? 1 : 0, __ATOMIC_RELAXED); }

static uint8_t* grow_buf(void* ctx, size_t want) {
  Buf* b = (Buf*)ctx;
  return b->reserve(want) ? b->p : nullptr;
}

// zlib / gzip stream -> growable buffer.  `padded`: src is readable for GEECO_FI_PAD bytes past n, so a zlib stream goes through
// the table-driven decoder of host_inflate.cpp first; whatever that declines (and every gzip stream) is zlib's.
static int inflate_into(const uint8_t* src, size_t n, int format, Buf* out, size_t fixed_cap, bool padded = false) {
  if (padded && format == 1 && !fixed_cap && __atomic_load_n(&g_fast_inflate, __ATOMIC_RELAXED)) {
    if (out->reserve(n * 5 + (1u << 20))) {
      int64_t got = geeco_fast_inflate(src, n, grow_buf, out, out->p, out->cap);
      if (got >= 0) {
        out->n = (size_t)got;
        return 0;
      }
    }
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, format == 2 ? 16 + MAX_WBITS : MAX_WBITS) != Z_OK) {
    set_err("inflateInit2 failed");
    return -1;
  }
  zs.next_in = const_cast<Bytef*>(src);
  size_t in_left = n;
  out->n = 0;
  for (;;) {
    if (out->n == out->cap) {
      if (fixed_cap) {
        // full: the stream may still have its end-of-block / checksum bytes left and produce nothing more
        Bytef scratch;
        zs.avail_in = (uInt)(in_left > (1u << 30) ? (1u << 30) : in_left);
        zs.next_out = &scratch;
        zs.avail_out = 1;
        int rc = inflate(&zs, Z_NO_FLUSH);
        bool done = rc == Z_STREAM_END && zs.avail_out == 1;
        bool more = zs.avail_out == 0;
        inflateEnd(&zs);
        if (done) return 0;
        if (more) {
          set_err("inflate: destination too small (%zu bytes)", fixed_cap);
          return -2;
        }
        set_err("inflate: malformed or truncated stream (zlib rc %d)", rc);
        return -1;
      }
      if (!out->reserve(out->cap ? out->cap * 2 : (n * 8 + (1u << 20)))) {
        inflateEnd(&zs);
        set_err("inflate: out of memory at %zu bytes", out->cap);
        return -1;
      }
    }
    // zlib counts in 32-bit uInt: feed and drain in < 4 GiB pieces
    uInt in_now = (uInt)(in_left > (1u << 30) ? (1u << 30) : in_left);
    size_t room = out->cap - out->n;
    uInt out_now = (uInt)(room > (1u << 30) ? (1u << 30) : room);
    zs.avail_in = in_now;
    zs.next_out = out->p + out->n;
    zs.avail_out = out_now;
    int rc = inflate(&zs, Z_NO_FLUSH);
    in_left -= in_now - zs.avail_in;
    out->n += out_now - zs.avail_out;
    if (rc == Z_STREAM_END) break;
    if (rc == Z_BUF_ERROR && zs.avail_out == 0) continue;      // needs more room
    if (rc != Z_OK || (in_left == 0 && zs.avail_out != 0)) {
      set_err("inflate: malformed or truncated stream (zlib rc %d, %s)", rc, zs.msg ? zs.msg : "-");
      inflateEnd(&zs);
      return -1;
    }
  }
  inflateEnd(&zs);
  return 0;
}

// The table-driven decoder alone (tests, benchmarks): zlib stream -> dst; -2 = dst too small, -3 = declined (host_inflate.h).
extern "C" int64_t geeco_inflate_fast(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  if (!src || (!dst && cap)) {
    set_err("geeco_inflate_fast: bad arguments");
    return -1;
  }
  Buf in, out;
  if (!in.reserve(n + GEECO_FI_PAD) || !out.reserve(n * 5 + (1u << 20))) {
    set_err("geeco_inflate_fast: out of memory");
    return -1;
  }
  memcpy(in.p, src, n);
  memset(in.p + n, 0, GEECO_FI_PAD);
  int64_t got = geeco_fast_inflate(in.p, n, grow_buf, &out, out.p, out.cap);
  if (got < 0) {
    set_err("geeco_inflate_fast: the stream is not one the table-driven decoder takes (zlib decides what is wrong with it)");
    return GEECO_FI_FALLBACK;
  }
  if ((size_t)got > cap) {
    set_err("geeco_inflate_fast: destination too small (%zu bytes for %lld)", cap, (long long)got);
    return -2;
  }
  memcpy(dst, out.p, (size_t)got);
  return got;
}

extern "C" int64_t geeco_inflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int format) {
  if (!src || !dst || (format != 1 && format != 2)) {
    set_err("geeco_inflate: bad arguments");
    return -1;
  }
  Buf b;
  b.p = dst;
  b.cap = cap;
  int rc = inflate_into(src, n, format, &b, cap ? cap : 1);
  int64_t got = (int64_t)b.n;
  b.p = nullptr;     // caller's memory
  return rc == 0 ? got : rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// SequenceExample index
//   SequenceExample { Features context = 1; FeatureLists feature_lists = 2; }
//   FeatureLists { map<string, FeatureList> feature_list = 1; }   map entry { key = 1; value = 2; }
//   FeatureList  { repeated Feature feature = 1; }
//   Feature      { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
//   FloatList { repeated float value = 1 [packed] }   Int64List { repeated int64 value = 1 [packed] }
// ---------------------------------------------------------------------------------------------------------------------
struct Span {
  const uint8_t* p;
  size_t n;
};

struct FeatureList {
  std::string name;
  std::vector<Span> frames;    // the Feature message of every frame
};

// Spare output buffers of closed episodes.  An inflated episode is ~105 MB: beyond glibc's mmap threshold, so every malloc of one is
// a fresh mapping whose 26 k pages fault in one by one while the decoder writes (~40 ms per episode, 10 % of reading it); a buffer a
// closed episode hands back is already mapped.  Bounded: at most g_spare_max buffers (one per reader thread + 1) / kSpareBytes
// bytes are kept, and the reader frees them all when an epoch's last episode has been read (geeco_host_release_buffers).
static const size_t kSpareBytes = (size_t)6 << 30;
static size_t g_spare_max = 8;            // geeco_host_set_buffer_limit: the reader sets it to its thread count + 1
static pthread_mutex_t g_spare_mu = PTHREAD_MUTEX_INITIALIZER;
static std::vector<std::pair<uint8_t*, size_t>> g_spare;
static size_t g_spare_bytes = 0;

static void take_spare(Buf* b, size_t want) {
  pthread_mutex_lock(&g_spare_mu);
  int best = -1;
  for (int i = 0; i < (int)g_spare.size(); ++i)
    if (g_spare[i].second >= want && (best < 0 || g_spare[i].second < g_spare[best].second)) best = i;
  if (best < 0 && !g_spare.empty()) {       // none is large enough: the largest one grows by mremap, most of it already mapped
    best = 0;
    for (int i = 1; i < (int)g_spare.size(); ++i)
      if (g_spare[i].second > g_spare[best].second) best = i;
  }
  if (best >= 0) {
    free(b->p);
    b->p = g_spare[best].first;
    b->cap = g_spare[best].second;
    b->n = 0;
    g_spare_bytes -= b->cap;
    g_spare.erase(g_spare.begin() + best);
  }
  pthread_mutex_unlock(&g_spare_mu);
}

static void give_spare(Buf* b) {
  if (!b->p) return;
  pthread_mutex_lock(&g_spare_mu);
  if (g_spare.size() < g_spare_max && g_spare_bytes + b->cap <= kSpareBytes) {
    g_spare.emplace_back(b->p, b->cap);
    g_spare_bytes += b->cap;
    b->p = nullptr;
    b->cap = b->n = 0;
  }
  pthread_mutex_unlock(&g_spare_mu);
}

extern "C" void geeco_host_set_buffer_limit(int max_buffers) {
  pthread_mutex_lock(&g_spare_mu);
  g_spare_max = (size_t)(max_buffers < 0 ? 0 : max_buffers > 64 ? 64 : max_buffers);
  while (g_spare.size() > g_spare_max) {
    g_spare_bytes -= g_spare.back().second;
    free(g_spare.back().first);
    g_spare.pop_back();
  }
  pthread_mutex_unlock(&g_spare_mu);
}

extern "C" int geeco_host_spare_buffers(void) {
  pthread_mutex_lock(&g_spare_mu);
  int n = (int)g_spare.size();
  pthread_mutex_unlock(&g_spare_mu);
  return n;
}

extern "C" void geeco_host_release_buffers(void) {
  pthread_mutex_lock(&g_spare_mu);
  for (auto& s : g_spare) free(s.first);
  g_spare.clear();
  g_spare_bytes = 0;
  pthread_mutex_unlock(&g_spare_mu);
}

struct geeco_episode {
  Buf raw;                                // inflated stream (or the file itself)
  ~geeco_episode() { give_spare(&raw); }
  int64_t num_records = 0;
  std::vector<FeatureList> lists;
  const FeatureList* find(const char* name) const {
    for (const FeatureList& l : lists)
      if (l.name == name) return &l;
    return nullptr;
  }
};

static bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  uint64_t r = 0;
  for (int shift = 0; shift < 64 && p < end; shift += 7) {
    uint8_t b = *p++;
    r |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *v = r;
      return true;
    }
  }
  return false;
}

// next field of a message: number, wire type, and for length-delimited fields the body; fixed / varint values in *val
static bool next_field(const uint8_t*& p, const uint8_t* end, uint32_t* fnum, uint32_t* wt, Span* body, uint64_t* val) {
  uint64_t key;
  if (!get_varint(p, end, &key)) return false;
  *fnum = (uint32_t)(key >> 3);
  *wt = (uint32_t)(key & 7);
  switch (*wt) {
    case 0:
      return get_varint(p, end, val);
    case 1:
      if ((size_t)(end - p) < 8) return false;
      memcpy(val, p, 8);
      p += 8;
      return true;
    case 2: {
      uint64_t len;
      if (!get_varint(p, end, &len) || len > (uint64_t)(end - p)) return false;
      body->p = p;
      body->n = (size_t)len;
      p += len;
      return true;
    }
    case 5:
      if ((size_t)(end - p) < 4) return false;
      *val = 0;
      memcpy(val, p, 4);
      p += 4;
      return true;
    default:
      return false;
  }
}

static bool index_feature_lists(Span payload, std::vector<FeatureList>* lists) {
  const uint8_t *p = payload.p, *end = payload.p + payload.n;
  uint32_t f, wt;
  Span body;
  uint64_t val;
  while (p < end) {
    if (!next_field(p, end, &f, &wt, &body, &val)) return false;
    if (f != 2 || wt != 2) continue;                          // context (= the meta file, geeco_gym.py:304) is not decoded
    const uint8_t *q = body.p, *qend = body.p + body.n;
    while (q < qend) {                                        // FeatureLists: map entries
      Span entry;
      if (!next_field(q, qend, &f, &wt, &entry, &val)) return false;
      if (f != 1 || wt != 2) continue;
      FeatureList fl;
      Span value = {nullptr, 0};
      const uint8_t *e = entry.p, *eend = entry.p + entry.n;
      while (e < eend) {
        Span s;
        if (!next_field(e, eend, &f, &wt, &s, &val)) return false;
        if (f == 1 && wt == 2) fl.name.assign((const char*)s.p, s.n);
        else if (f == 2 && wt == 2) value = s;
      }
      const uint8_t *v = value.p, *vend = value.p + value.n;
      while (v < vend) {                                      // FeatureList: repeated Feature
        Span feat;
        if (!next_field(v, vend, &f, &wt, &feat, &val)) return false;
        if (f == 1 && wt == 2) fl.frames.push_back(feat);
      }
      lists->push_back(std::move(fl));
    }
  }
  return true;
}

// the value list inside a Feature: kind (1 bytes, 2 float, 3 int64; 0 = empty feature) and its body
static bool feature_kind(Span feat, int* kind, Span* list) {
  const uint8_t *p = feat.p, *end = feat.p + feat.n;
  *kind = 0;
  list->p = nullptr;
  list->n = 0;
  uint32_t f, wt;
  Span body;
  uint64_t val;
  while (p < end) {
    if (!next_field(p, end, &f, &wt, &body, &val)) return false;
    if (wt == 2 && f >= 1 && f <= 3) {
      *kind = (int)f;
      *list = body;
    }
  }
  return true;
}

// FloatList body -> dst[count]; packed (one length-delimited run) or unpacked fixed32 entries, in any mix
static int64_t decode_floats(Span list, float* dst, int64_t cap) {
  const uint8_t *p = list.p, *end = list.p + list.n;
  int64_t n = 0;
  uint32_t f, wt;
  Span body;
  uint64_t val;
  while (p < end) {
    if (!next_field(p, end, &f, &wt, &body, &val)) return -1;
    if (f != 1) continue;
    if (wt == 2) {
      if (body.n % 4) return -1;
      int64_t k = (int64_t)(body.n / 4);
      if (n + k > cap) return -2;
      if (dst) memcpy(dst + n, body.p, body.n);
      n += k;
    } else if (wt == 5) {
      if (n + 1 > cap) return -2;
      if (dst) {
        uint32_t u = (uint32_t)val;
        memcpy(dst + n, &u, 4);
      }
      n += 1;
    } else {
      return -1;
    }
  }
  return n;
}

static int64_t count_values(int kind, Span list) {
  if (kind == 2) return decode_floats(list, nullptr, INT64_MAX);
  const uint8_t *p = list.p, *end = list.p + list.n;
  int64_t n = 0;
  uint32_t f, wt;
  Span body;
  uint64_t val;
  while (p < end) {
    if (!next_field(p, end, &f, &wt, &body, &val)) return -1;
    if (f != 1) continue;
    if (kind == 3 && wt == 2) {
      const uint8_t *q = body.p, *qend = body.p + body.n;
      while (q < qend) {
        if (!get_varint(q, qend, &val)) return -1;
        ++n;
      }
    } else {
      ++n;
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// episode handle
// ---------------------------------------------------------------------------------------------------------------------
static bool read_file(const char* path, Buf* out) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    set_err("%s: %s", path, strerror(errno));
    return false;
  }
  bool ok = false;
  if (fseek(f, 0, SEEK_END) == 0) {
    long sz = ftell(f);
    if (sz >= 0 && fseek(f, 0, SEEK_SET) == 0 && out->reserve((size_t)sz + GEECO_FI_PAD)) {
      out->n = fread(out->p, 1, (size_t)sz, f);
      ok = out->n == (size_t)sz;
      if (ok) memset(out->p + out->n, 0, GEECO_FI_PAD);      // the inflate's bit buffer refills 8 bytes at a time
    }
  }
  if (!ok) set_err("%s: read failed", path);
  fclose(f);
  return ok;
}

extern "C" geeco_episode* geeco_episode_open(const char* path, int compression, int verify_crc) {
  if (!path || compression < 0 || compression > 2) {
    set_err("geeco_episode_open: bad arguments");
    return nullptr;
  }
  geeco_episode* ep = new geeco_episode();
  {
    // the compressed file goes through a per-thread buffer that stays mapped between episodes (a fresh 27 MB block is 6.6 k page
    // faults); released when it has grown beyond 256 MB
    static thread_local Buf file;
    if (file.cap > ((size_t)256 << 20)) {
      free(file.p);
      file.p = nullptr;
      file.cap = file.n = 0;
    }
    if (!read_file(path, compression ? &file : &ep->raw)) {
      delete ep;
      return nullptr;
    }
    if (compression) take_spare(&ep->raw, file.n * 4);
    if (compression && inflate_into(file.p, file.n, compression, &ep->raw, 0, /*padded=*/true) != 0) {
      std::string why = t_err;
      set_err("%s: %s", path, why.c_str());
      delete ep;
      return nullptr;
    }
  }
  // record framing: u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)   [TF1.15 record format]
  const uint8_t* raw = ep->raw.p;
  size_t pos = 0, n = ep->raw.n;
  Span first = {nullptr, 0};
  while (pos < n) {
    if (pos + 12 > n) {
      set_err("%s: truncated record header at byte %zu", path, pos);
      delete ep;
      return nullptr;
    }
    uint64_t len;
    uint32_t lcrc, dcrc;
    memcpy(&len, raw + pos, 8);
    memcpy(&lcrc, raw + pos + 8, 4);
    if (verify_crc && geeco_masked_crc32c(raw + pos, 8) != lcrc) {
      set_err("%s: corrupted record length at byte %zu", path, pos);
      delete ep;
      return nullptr;
    }
    if (len > n - pos - 12 || n - pos - 12 - len < 4) {
      set_err("%s: truncated record at byte %zu", path, pos);
      delete ep;
      return nullptr;
    }
    const uint8_t* payload = raw + pos + 12;
    memcpy(&dcrc, payload + len, 4);
    if (verify_crc && geeco_masked_crc32c(payload, (size_t)len) != dcrc) {
      set_err("%s: corrupted record payload at byte %zu", path, pos);
      delete ep;
      return nullptr;
    }
    if (ep->num_records == 0) first = Span{payload, (size_t)len};
    ep->num_records += 1;
    pos += 12 + (size_t)len + 4;
  }
  if (ep->num_records == 0) {
    set_err("%s: no record", path);
    delete ep;
    return nullptr;
  }
  if (!index_feature_lists(first, &ep->lists)) {
    set_err("%s: malformed SequenceExample", path);
    delete ep;
    return nullptr;
  }
  return ep;
}

extern "C" void geeco_episode_close(geeco_episode* ep) { delete ep; }
extern "C" int64_t geeco_episode_num_records(const geeco_episode* ep) { return ep ? ep->num_records : -1; }
extern "C" int64_t geeco_episode_inflated_bytes(const geeco_episode* ep) { return ep ? (int64_t)ep->raw.n : -1; }
extern "C" int geeco_episode_num_lists(const geeco_episode* ep) { return ep ? (int)ep->lists.size() : -1; }

extern "C" const char* geeco_episode_list_name(const geeco_episode* ep, int i) {
  if (!ep || i < 0 || i >= (int)ep->lists.size()) return nullptr;
  return ep->lists[i].name.c_str();
}

extern "C" int64_t geeco_episode_list_frames(const geeco_episode* ep, const char* name) {
  const FeatureList* l = (ep && name) ? ep->find(name) : nullptr;
  return l ? (int64_t)l->frames.size() : -1;
}

extern "C" int geeco_episode_list_kind(const geeco_episode* ep, const char* name, int64_t* values) {
  const FeatureList* l = (ep && name) ? ep->find(name) : nullptr;
  if (!l) {
    set_err("feature list '%s' missing", name ? name : "(null)");
    return -1;
  }
  int kind = 0;
  Span list = {nullptr, 0};
  if (!l->frames.empty() && !feature_kind(l->frames[0], &kind, &list)) {
    set_err("feature list '%s': malformed feature", name);
    return -1;
  }
  if (values) *values = kind ? count_values(kind, list) : 0;
  return kind;
}

// common checks of the read calls; returns the list or NULL
static const FeatureList* checked_list(const geeco_episode* ep, const char* name, const void* dst, int64_t frames,
                                       int64_t vpf) {
  if (!ep || !name || !dst || frames < 0 || vpf < 0) {
    set_err("geeco_episode_read: bad arguments");
    return nullptr;
  }
  const FeatureList* l = ep->find(name);
  if (!l) {
    set_err("feature list '%s' missing", name);
    return nullptr;
  }
  if ((int64_t)l->frames.size() != frames) {
    set_err("feature list '%s' holds %zu frames, caller expects %lld", name, l->frames.size(), (long long)frames);
    return nullptr;
  }
  return l;
}

extern "C" int geeco_episode_read_f32(const geeco_episode* ep, const char* name, float* dst, int64_t frames, int64_t vpf) {
  const FeatureList* l = checked_list(ep, name, dst, frames, vpf);
  if (!l) return -1;
  for (int64_t t = 0; t < frames; ++t) {
    int kind;
    Span list;
    if (!feature_kind(l->frames[t], &kind, &list) || (kind != 2 && !(kind == 0 && vpf == 0))) {
      set_err("feature list '%s' frame %lld: not a float list", name, (long long)t);
      return -1;
    }
    if (decode_floats(list, dst + t * vpf, vpf) != vpf) {
      set_err("feature list '%s' frame %lld: value count differs from %lld", name, (long long)t, (long long)vpf);
      return -1;
    }
  }
  return 0;
}

extern "C" int geeco_episode_read_i64(const geeco_episode* ep, const char* name, int64_t* dst, int64_t frames, int64_t vpf) {
  const FeatureList* l = checked_list(ep, name, dst, frames, vpf);
  if (!l) return -1;
  for (int64_t t = 0; t < frames; ++t) {
    int kind;
    Span list;
    if (!feature_kind(l->frames[t], &kind, &list) || (kind != 3 && !(kind == 0 && vpf == 0))) {
      set_err("feature list '%s' frame %lld: not an int64 list", name, (long long)t);
      return -1;
    }
    const uint8_t *p = list.p, *end = list.p + list.n;
    int64_t n = 0;
    uint32_t f, wt;
    Span body;
    uint64_t val;
    bool ok = true;
    while (ok && p < end) {
      if (!next_field(p, end, &f, &wt, &body, &val)) ok = false;
      else if (f != 1) continue;
      else if (wt == 2) {
        const uint8_t *q = body.p, *qend = body.p + body.n;
        while (ok && q < qend) {
          if (!get_varint(q, qend, &val) || n >= vpf) ok = false;
          else dst[t * vpf + n++] = (int64_t)val;
        }
      } else if (wt == 0) {
        if (n >= vpf) ok = false;
        else dst[t * vpf + n++] = (int64_t)val;
      } else {
        ok = false;
      }
    }
    if (!ok || n != vpf) {
      set_err("feature list '%s' frame %lld: value count differs from %lld", name, (long long)t, (long long)vpf);
      return -1;
    }
  }
  return 0;
}

// floats -> uint8 with the integrality test in the same pass; returns 1 iff every value was an integer in [0, 255]
static int floats_to_u8_scalar(const uint8_t* src, uint8_t* dst, int64_t n) {
  uint32_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    float v;
    memcpy(&v, src + 4 * i, 4);
    // NaN and out-of-range values fail the comparisons; the clamp keeps the int conversion defined
    float c = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
    int k = (int)c;
    bad |= (uint32_t)!((float)k == v);
    dst[i] = (uint8_t)k;
  }
  return bad ? 0 : 1;
}

#if defined(__x86_64__)
// 32 floats -> 32 bytes per step (the scalar loop runs at 1.6 GB/s of floats: 48 ms of an episode's 0.27 s)
__attribute__((target("avx2"))) static int floats_to_u8_avx2(const uint8_t* src, uint8_t* dst, int64_t n) {
  const __m256i lanes = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7), hi = _mm256_set1_epi32(~255);
  __m256i bad = _mm256_setzero_si256();
  int64_t i = 0;
  for (; i + 32 <= n; i += 32) {
    __m256i k[4];
    for (int j = 0; j < 4; ++j) {
      const __m256 v = _mm256_loadu_ps(reinterpret_cast<const float*>(src + 4 * (i + 8 * j)));
      k[j] = _mm256_cvttps_epi32(v);                                       // NaN / out of int range -> 0x80000000
      const __m256 same = _mm256_cmp_ps(_mm256_cvtepi32_ps(k[j]), v, _CMP_EQ_OQ);
      bad = _mm256_or_si256(bad, _mm256_or_si256(_mm256_and_si256(k[j], hi), _mm256_xor_si256(_mm256_castps_si256(same), _mm256_set1_epi32(-1))));
    }
    const __m256i w0 = _mm256_packus_epi32(k[0], k[1]), w1 = _mm256_packus_epi32(k[2], k[3]);   // saturating: < 0 -> 0, > 65535 -> 65535
    const __m256i b = _mm256_permutevar8x32_epi32(_mm256_packus_epi16(w0, w1), lanes);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), b);
  }
  int ok = _mm256_testz_si256(bad, bad);
  if (i < n) ok &= floats_to_u8_scalar(src + 4 * i, dst + i, n - i);
  return ok;
}
#endif

static int floats_to_u8(const uint8_t* src, uint8_t* dst, int64_t n) {
#if defined(__x86_64__)
  static const int avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return floats_to_u8_avx2(src, dst, n);
#endif
  return floats_to_u8_scalar(src, dst, n);
}

extern "C" int geeco_episode_read_u8(const geeco_episode* ep, const char* name, uint8_t* dst, int64_t frames, int64_t vpf,
                                     int* exact) {
  const FeatureList* l = checked_list(ep, name, dst, frames, vpf);
  if (!l || !exact) {
    if (l) set_err("geeco_episode_read_u8: exact is NULL");
    return -1;
  }
  int all = 1;
  std::vector<float> tmp;
  for (int64_t t = 0; t < frames; ++t) {
    int kind;
    Span list;
    if (!feature_kind(l->frames[t], &kind, &list) || (kind != 2 && !(kind == 0 && vpf == 0))) {
      set_err("feature list '%s' frame %lld: not a float list", name, (long long)t);
      return -1;
    }
    // fast path: one packed run (what every writer produces)
    const uint8_t *p = list.p, *end = list.p + list.n;
    uint32_t f, wt;
    Span body;
    uint64_t val;
    if (p < end && next_field(p, end, &f, &wt, &body, &val) && p == end && f == 1 && wt == 2) {
      if ((int64_t)body.n != 4 * vpf) {
        set_err("feature list '%s' frame %lld: value count differs from %lld", name, (long long)t, (long long)vpf);
        return -1;
      }
      all &= floats_to_u8(body.p, dst + t * vpf, vpf);
    } else {
      tmp.resize((size_t)vpf);
      if (decode_floats(list, tmp.data(), vpf) != vpf) {
        set_err("feature list '%s' frame %lld: value count differs from %lld", name, (long long)t, (long long)vpf);
        return -1;
      }
      all &= floats_to_u8((const uint8_t*)tmp.data(), dst + t * vpf, vpf);
    }
  }
  *exact = all;
  return 0;
}
