// A table-driven DEFLATE decoder for the episode reader (RFC 1950 / 1951; plain C ABI, no GPU).
//
// Why: reading an episode (src/data/geeco_gym.py:442-445: TFRecordDataset(compression_type='ZLIB')) is 97 % inflate: 105 MB
// out of a 27 MB file, 0.43 s per episode and thread with the system zlib (242 MB/s), which bounds epoch 1 of real-data
// training at 40 % of the GPU's step rate on the box's 16 usable cores.  The recorder's frames are uint8 values stored as
// float lists (src/data/utils/tfrecord.py:73-74): the stream is almost all short matches at distances that are multiples of 4,
// so a decoder wins by (1) a 64-bit bit buffer refilled with one unaligned 8-byte load, (2) 11-bit root tables (one
// lookup for nearly every literal / length code of these streams), (3) match copies in 8-byte words with replicated patterns
// for distances 1, 2 and 4 instead of byte loops.  Nothing here is format-specific: any zlib stream decodes.
//
// Contract: geeco_fast_inflate() either returns the exact output zlib's inflate() would (the Adler-32 trailer is verified)
// or a negative code having written nothing the caller may use; host_io.cpp then runs the stream through zlib, which also
// produces the authoritative error text for malformed input.  The input buffer must be readable for GEECO_FI_PAD bytes past its
// end (the refill loads 8 bytes at a time and the position is checked once per stretch of symbols); the output buffer grows
// through the callback.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "host_inflate.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// Adler-32 of the output (RFC 1950 trailer).  zlib's scalar loop runs at ~1.7 GB/s: 60 ms for the 105 MB of an episode, a third of
// the decode itself; 32 bytes per step with AVX2 (byte sums by psadbw, position-weighted sums by pmaddubsw) leaves ~10 ms.
#if defined(__x86_64__)
__attribute__((target("avx2"))) uint32_t adler32_avx2(uint32_t adler, const uint8_t* p, size_t n) {
  uint64_t s1 = adler & 0xffff, s2 = adler >> 16;
  const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9,
                                           8, 7, 6, 5, 4, 3, 2, 1);
  const __m256i ones16 = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
  while (n >= 32) {
    size_t blocks = n / 32;
    if (blocks > 173) blocks = 173;          // 5536 bytes: the 32-bit lanes below cannot overflow
    n -= blocks * 32;
    __m256i v_s1 = zero, v_ps = zero, v_s2 = zero;
    for (size_t k = 0; k < blocks; ++k, p += 32) {
      const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
      v_ps = _mm256_add_epi32(v_ps, v_s1);                               // byte sums of the blocks before this one
      v_s1 = _mm256_add_epi32(v_s1, _mm256_sad_epu8(v, zero));
      v_s2 = _mm256_add_epi32(v_s2, _mm256_madd_epi16(_mm256_maddubs_epi16(v, weights), ones16));
    }
    uint32_t a[8], b[8], c[8];
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(a), v_s1);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(b), v_ps);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(c), v_s2);
    uint64_t h1 = 0, hp = 0, h2 = 0;
    for (int i = 0; i < 8; ++i) {
      h1 += a[i]; hp += b[i]; h2 += c[i];
    }
    s2 = (s2 + 32 * blocks * s1 + 32 * hp + h2) % 65521;
    s1 = (s1 + h1) % 65521;
  }
  while (n--) {
    s1 += *p++;
    s2 += s1;
  }
  return (uint32_t)(((s2 % 65521) << 16) | (s1 % 65521));
}
#endif

uint32_t adler32_of(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const int avx2 = __builtin_cpu_supports("avx2");
  if (avx2) return adler32_avx2(1, p, n);
#endif
  uint32_t a = 1;
  for (size_t off = 0; off < n;) {                      // zlib counts in 32-bit pieces
    const size_t piece = n - off > (1u << 30) ? (1u << 30) : n - off;
    a = (uint32_t)adler32(a, p + off, (uInt)piece);
    off += piece;
  }
  return a;
}

struct Entry {
  uint16_t val;     // literal, length / distance base, or subtable offset
  uint8_t op;       // see below
  uint8_t bits;     // bits this entry consumes: the code (its part behind the root bits in a subtable) + the extra bits of a length /
                    // distance; the root bits for a subtable pointer
};
// op: 0 = literal; OP_BASE | extra bits = length / distance; OP_EOB; OP_SUB | index bits of the subtable; OP_BAD
constexpr uint8_t OP_BASE = 0x10, OP_EOB = 0x20, OP_SUB = 0x40, OP_BAD = 0x80;

constexpr int LROOT = 11, DROOT = 8, PROOT = 7;
constexpr int LSIZE = (1 << LROOT) + 288 * (1 << (15 - LROOT));   // every long code could open its own subtable
constexpr int DSIZE = (1 << DROOT) + 32 * (1 << (15 - DROOT));

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPrecodeOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum Kind { LITLEN, DIST, PRECODE };

inline uint32_t bitrev(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) {
    r = (r << 1) | (code & 1);
    code >>= 1;
  }
  return r;
}

inline Entry symbol_entry(Kind kind, int sym, int bits) {
  Entry e;
  e.bits = (uint8_t)bits;
  if (kind == PRECODE) {
    e.val = (uint16_t)sym; e.op = 0;
  } else if (kind == LITLEN) {
    if (sym < 256) { e.val = (uint16_t)sym; e.op = 0; }
    else if (sym == 256) { e.val = 0; e.op = OP_EOB; }
    else if (sym < 286) { e.val = kLenBase[sym - 257]; e.op = (uint8_t)(OP_BASE | kLenExtra[sym - 257]); e.bits = (uint8_t)(bits + kLenExtra[sym - 257]); }
    else { e.val = 0; e.op = OP_BAD; }
  } else {
    if (sym < 30) { e.val = kDistBase[sym]; e.op = (uint8_t)(OP_BASE | kDistExtra[sym]); e.bits = (uint8_t)(bits + kDistExtra[sym]); }
    else { e.val = 0; e.op = OP_BAD; }
  }
  return e;
}

// Canonical Huffman code -> lookup table with `root` index bits and fixed-size subtables for longer codes.
// Returns false for an over-subscribed or (other than the single-code case deflate itself emits) incomplete code.
bool build_table(Kind kind, const uint8_t* lens, int n, int root, Entry* table, int table_cap) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  int maxlen = 15;
  while (maxlen > 0 && count[maxlen] == 0) --maxlen;
  const Entry bad = {0, OP_BAD, 1};
  if (maxlen == 0) {                      // no codes at all (a block without matches has an empty distance code)
    for (int i = 0; i < (1 << root); ++i) table[i] = bad;
    return kind == DIST;
  }
  int left = 1;
  for (int len = 1; len <= 15; ++len) {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;
  }
  int ncodes = n - count[0];
  if (left > 0 && !(ncodes == 1 && count[1] == 1)) return false;     // incomplete: only "one code of one bit" is legal
  // symbols in canonical order
  uint16_t sorted[288];
  int offs[17];
  offs[1] = 0;
  for (int len = 1; len <= 15; ++len) offs[len + 1] = offs[len] + count[len];
  for (int i = 0; i < n; ++i)
    if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
  for (int i = 0; i < (1 << root); ++i) table[i] = bad;
  const int subbits = maxlen > root ? maxlen - root : 0;
  int next_sub = 1 << root;
  uint32_t code = 0;
  int k = 0;
  for (int len = 1; len <= maxlen; ++len) {
    for (int c = 0; c < count[len]; ++c, ++k, ++code) {
      const int sym = sorted[k];
      const uint32_t rev = bitrev(code, len);
      if (len <= root) {
        const Entry e = symbol_entry(kind, sym, len);
        for (uint32_t i = rev; i < (1u << root); i += 1u << len) table[i] = e;
      } else {
        const uint32_t prefix = rev & ((1u << root) - 1);
        if (!(table[prefix].op & OP_SUB)) {
          if (next_sub + (1 << subbits) > table_cap) return false;
          table[prefix].val = (uint16_t)next_sub;
          table[prefix].op = (uint8_t)(OP_SUB | subbits);
          table[prefix].bits = (uint8_t)root;
          for (int i = 0; i < (1 << subbits); ++i) table[next_sub + i] = bad;
          next_sub += 1 << subbits;
        }
        const Entry e = symbol_entry(kind, sym, len - root);
        Entry* sub = table + table[prefix].val;
        for (uint32_t i = rev >> root; i < (1u << subbits); i += 1u << (len - root)) sub[i] = e;
      }
    }
    code <<= 1;
  }
  return true;
}

struct Tables {
  Entry lit[LSIZE];
  Entry dist[DSIZE];
};

const Tables* fixed_tables() {
  static Tables* t = [] {
    Tables* f = (Tables*)malloc(sizeof(Tables));
    uint8_t lens[288];
    for (int i = 0; i < 144; ++i) lens[i] = 8;
    for (int i = 144; i < 256; ++i) lens[i] = 9;
    for (int i = 256; i < 280; ++i) lens[i] = 7;
    for (int i = 280; i < 288; ++i) lens[i] = 8;
    build_table(LITLEN, lens, 288, LROOT, f->lit, LSIZE);
    uint8_t dl[32];
    for (int i = 0; i < 32; ++i) dl[i] = 5;
    build_table(DIST, dl, 32, DROOT, f->dist, DSIZE);
    return f;
  }();
  return t;
}

inline uint64_t load64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;       // x86-64 / little endian hosts only (checked at compile time below)
}
inline void store64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }

#if !defined(__BYTE_ORDER__) || __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "host_inflate.cpp assumes a little-endian host"
#endif

}  // namespace

// see host_inflate.h
int64_t geeco_fast_inflate(const uint8_t* src, size_t n, geeco_grow_fn grow, void* grow_ctx, uint8_t* dst, size_t cap) {
  if (n < 6) return GEECO_FI_FALLBACK;
  // zlib header: CM = 8, window <= 32 K, check bits, no preset dictionary
  if ((src[0] & 0x0f) != 8 || (src[0] >> 4) > 7 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return GEECO_FI_FALLBACK;
  const uint8_t* in = src + 2;
  const uint8_t* const in_end = src + n;
  uint64_t bb = 0;
  int bc = 0;
  size_t out = 0;
  constexpr size_t MARGIN = 320;            // one iteration writes <= 2 literals + a match of 258 + 15 bytes of word-copy overshoot
  Tables* dyn = (Tables*)malloc(sizeof(Tables));
  if (!dyn) return GEECO_FI_FALLBACK;
  int64_t result = GEECO_FI_FALLBACK;

#define REFILL()                           \
  do {                                     \
    bb |= load64(in) << bc;                \
    in += (63 - bc) >> 3;                  \
    bc |= 56;                              \
  } while (0)
#define TAKE(k) (bb >>= (k), bc -= (k))
// the refill runs ahead of the data by design (padding bytes enter the bit buffer): what counts is the position of the first
// bit not yet consumed
#define OVERRUN() (in - (bc >> 3) > in_end)
#define ROOM(need)                                              \
  do {                                                          \
    if (out + (need) > cap) {                                   \
      size_t want = cap * 2 > out + (need) ? cap * 2 : out + (need) * 2; \
      dst = grow ? grow(grow_ctx, want) : nullptr;              \
      if (!dst) goto done;                                      \
      cap = want;                                               \
    }                                                           \
  } while (0)

  for (;;) {
    if (OVERRUN()) goto done;             // consumed more than there is: truncated stream
    REFILL();
    const int bfinal = (int)(bb & 1);
    const int btype = (int)((bb >> 1) & 3);
    TAKE(3);
    if (btype == 0) {
      // stored: drop to the byte boundary, give the unread whole bytes back
      TAKE(bc & 7);
      in -= bc >> 3;
      bb = 0; bc = 0;
      if (in + 4 > in_end) goto done;
      const unsigned len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
      if ((len ^ nlen) != 0xffffu) goto done;
      in += 4;
      if ((size_t)(in_end - in) < len) goto done;
      ROOM(len + MARGIN);
      memcpy(dst + out, in, len);
      in += len;
      out += len;
    } else if (btype == 3) {
      goto done;
    } else {
      const Tables* t;
      if (btype == 1) {
        t = fixed_tables();
      } else {
        const int hlit = (int)(bb & 31) + 257, hdist = (int)((bb >> 5) & 31) + 1, hclen = (int)((bb >> 10) & 15) + 4;
        TAKE(14);
        if (hlit > 286 || hdist > 30) goto done;
        uint8_t pl[19] = {0};
        for (int i = 0; i < hclen; ++i) {
          if (bc < 3) REFILL();
          pl[kPrecodeOrder[i]] = (uint8_t)(bb & 7);
          TAKE(3);
        }
        Entry pre[1 << PROOT];
        if (!build_table(PRECODE, pl, 19, PROOT, pre, 1 << PROOT)) goto done;
        uint8_t lens[320];
        int i = 0;
        while (i < hlit + hdist) {
          if (OVERRUN()) goto done;
          REFILL();
          const Entry e = pre[bb & ((1 << PROOT) - 1)];
          if (e.op) goto done;
          TAKE(e.bits);
          if (e.val < 16) {
            lens[i++] = (uint8_t)e.val;
          } else {
            int rep, v = 0;
            if (e.val == 16) {
              if (i == 0) goto done;
              v = lens[i - 1];
              rep = 3 + (int)(bb & 3); TAKE(2);
            } else if (e.val == 17) {
              rep = 3 + (int)(bb & 7); TAKE(3);
            } else {
              rep = 11 + (int)(bb & 127); TAKE(7);
            }
            if (i + rep > hlit + hdist) goto done;
            while (rep--) lens[i++] = (uint8_t)v;
          }
        }
        if (lens[256] == 0) goto done;       // no end-of-block code
        if (!build_table(LITLEN, lens, hlit, LROOT, dyn->lit, LSIZE)) goto done;
        if (!build_table(DIST, lens + hlit, hdist, DROOT, dyn->dist, DSIZE)) goto done;
        t = dyn;
      }
      const Entry* lt = t->lit;
      const Entry* dt = t->dist;
      // Invariant at the top of an iteration: >= 56 bits in the buffer, `e` = the entry of the next code (looked up, not yet
      // consumed).  One iteration = up to three literals, or up to two literals and a match; the lookup of the NEXT iteration is
      // issued before the match is copied, so its latency hides behind the copy.
      REFILL();
      Entry e = lt[bb & ((1u << LROOT) - 1)];
      for (;;) {
        if (in > in_end + 8) goto done;        // the buffer runs <= 8 bytes ahead of the consumed position: this is past the data
        ROOM(MARGIN);
        if (e.op & OP_SUB) {
          TAKE(e.bits);
          e = lt[e.val + (bb & ((1u << (e.op & 15)) - 1))];
        }
        uint64_t saved = bb;                           // a length's extra bits are read from here, off the TAKE chain
        TAKE(e.bits);
        if (e.op == 0) {
          dst[out++] = (uint8_t)e.val;
          e = lt[bb & ((1u << LROOT) - 1)];            // >= 41 bits left
          if (e.op & OP_SUB) {
            TAKE(e.bits);
            e = lt[e.val + (bb & ((1u << (e.op & 15)) - 1))];
          }
          saved = bb;
          TAKE(e.bits);
          if (e.op == 0) {
            dst[out++] = (uint8_t)e.val;
            e = lt[bb & ((1u << LROOT) - 1)];          // >= 26 bits left (a length code + its extra bits: <= 20)
            if (e.op & OP_SUB) {
              TAKE(e.bits);
              e = lt[e.val + (bb & ((1u << (e.op & 15)) - 1))];
            }
            saved = bb;
            TAKE(e.bits);
            if (e.op == 0) {
              dst[out++] = (uint8_t)e.val;
              REFILL();
              e = lt[bb & ((1u << LROOT) - 1)];
              continue;
            }
          }
        }
        // `e` (consumed, extra bits included) is a length, the end of the block or invalid
        if (e.op & OP_BASE) {
          const int xb = e.op & 15;
          const unsigned len = e.val + (unsigned)((saved >> (e.bits - xb)) & ((1u << xb) - 1));
          REFILL();
          Entry d = dt[bb & ((1u << DROOT) - 1)];
          if (d.op & OP_SUB) {
            TAKE(d.bits);
            d = dt[d.val + (bb & ((1u << (d.op & 15)) - 1))];
          }
          if (!(d.op & OP_BASE)) goto done;
          const int db = d.op & 15;
          const size_t dist = d.val + (size_t)((bb >> (d.bits - db)) & ((1u << db) - 1));
          TAKE(d.bits);
          if (dist > out) goto done;
          REFILL();
          e = lt[bb & ((1u << LROOT) - 1)];            // next iteration's code, in flight during the copy
          uint8_t* o = dst + out;
          uint8_t* const oend = o + len;
          const uint8_t* s = o - dist;
          out += len;
          if (dist >= 8) {
            store64(o, load64(s));                     // two words unconditionally: 9 of 10 matches of these streams end here
            store64(o + 8, load64(s + 8));
            if (len > 16) {
              o += 16; s += 16;
              do {
                store64(o, load64(s));
                o += 8; s += 8;
              } while (o < oend);
            }
          } else if (dist == 4 || dist == 2 || dist == 1) {
            uint64_t pat;
            if (dist == 4) {
              uint32_t w;
              memcpy(&w, s, 4);
              pat = (uint64_t)w | ((uint64_t)w << 32);
            } else if (dist == 2) {
              uint16_t w;
              memcpy(&w, s, 2);
              pat = 0x0001000100010001ull * w;
            } else {
              pat = 0x0101010101010101ull * s[0];
            }
            do {
              store64(o, pat);
              o += 8;
            } while (o < oend);
          } else {
            do {
              *o++ = *s++;
            } while (o < oend);
          }
        } else if (e.op & OP_EOB) {
          break;
        } else {
          goto done;       // invalid code
        }
      }
      if (OVERRUN()) goto done;
    }
    if (bfinal) break;
  }
  {
    // trailer: give back the whole bytes still in the bit buffer, then the big-endian Adler-32 of the output
    TAKE(bc & 7);
    in -= bc >> 3;
    if (in + 4 > in_end) goto done;
    const uint32_t want = ((uint32_t)in[0] << 24) | ((uint32_t)in[1] << 16) | ((uint32_t)in[2] << 8) | in[3];
    const uint32_t a = adler32_of(dst, out);
    if (a != want) goto done;
    result = (int64_t)out;
  }
done:
#undef REFILL
#undef TAKE
#undef OVERRUN
#undef ROOM
  free(dyn);
  return result;
}
