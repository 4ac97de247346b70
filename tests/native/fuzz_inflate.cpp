// Sanitizer harness of csrc/host_inflate.cpp (built by tests/test_input_pipeline_cpu.py with -fsanitize=address,undefined):
// valid streams of several shapes must decode to their source; damaged ones (bit flips, truncations, garbage, spliced headers)
// must be declined or - when a flip happens to leave a valid stream - decode to what zlib decodes; never read or write outside
// the buffers (the input is an exact-size heap block + the documented GEECO_FI_PAD bytes).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "host_inflate.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

struct Out {
  std::vector<uint8_t> v;
};
static uint8_t* grow(void* ctx, size_t want) {
  Out* o = (Out*)ctx;
  o->v.resize(want);
  return o->v.data();
}

static int64_t fast(const std::vector<uint8_t>& comp, Out* out) {
  uint8_t* in = (uint8_t*)malloc(comp.size() + GEECO_FI_PAD);       // exact size: ASan sees any read past the padding
  memcpy(in, comp.data(), comp.size());
  memset(in + comp.size(), 0, GEECO_FI_PAD);
  out->v.resize(64);
  int64_t got = geeco_fast_inflate(in, comp.size(), grow, out, out->v.data(), out->v.size());
  free(in);
  return got;
}

static bool zlib_ref(const std::vector<uint8_t>& comp, std::vector<uint8_t>* out) {
  uLongf cap = 1 << 24;
  out->resize(cap);
  int rc = uncompress(out->data(), &cap, comp.data(), comp.size());
  if (rc != Z_OK) return false;
  out->resize(cap);
  return true;
}

int main() {
  std::vector<std::vector<uint8_t>> sources;
  sources.push_back({});
  sources.push_back({'a'});
  {
    std::vector<uint8_t> v(200000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)(rnd() & 255);
    sources.push_back(v);                                            // incompressible: stored blocks
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)(rnd() & 3);
    sources.push_back(v);                                            // low entropy: short codes
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)("abcdefg"[i % 7]);
    sources.push_back(v);                                            // period 7: byte-wise overlapping copies
    for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)(i % 3 == 0 ? rnd() & 255 : 0);
    sources.push_back(v);
    std::vector<uint8_t> f(400000);
    for (size_t i = 0; i + 4 <= f.size(); i += 4) {
      float x = (float)(rnd() & 255);
      memcpy(&f[i], &x, 4);
    }
    sources.push_back(f);                                            // the recorder's uint8-as-float lists
    std::vector<uint8_t> w(300000);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (uint8_t)((rnd() % 100 < 97) ? (i * 7) & 255 : rnd() & 255);
    sources.push_back(w);                                            // all 256 literals in use: long codes, subtables
  }
  long accepted = 0, declined = 0, valid = 0;
  for (const auto& src : sources) {
    for (int level : {0, 1, 4, 6, 9}) {
      uLongf cl = compressBound(src.size());
      std::vector<uint8_t> comp(cl);
      static const uint8_t none = 0;
      if (compress2(comp.data(), &cl, src.empty() ? &none : src.data(), src.size(), level) != Z_OK) return 2;
      comp.resize(cl);
      Out out;
      int64_t got = fast(comp, &out);
      if (got != (int64_t)src.size() || (got > 0 && memcmp(out.v.data(), src.data(), src.size()) != 0)) {
        fprintf(stderr, "valid stream (%zu bytes, level %d) not decoded: %lld\n", src.size(), level, (long long)got);
        return 3;
      }
      ++valid;
      // damage
      const int trials = src.size() > 1000 ? 120 : 30;
      for (int t = 0; t < trials; ++t) {
        std::vector<uint8_t> bad = comp;
        const int kind = (int)(rnd() % 4);
        if (kind == 0 && !bad.empty()) {
          bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
        } else if (kind == 1) {
          bad.resize(rnd() % (bad.size() + 1));
        } else if (kind == 2 && bad.size() > 8) {
          size_t at = 2 + rnd() % (bad.size() - 2), len = 1 + rnd() % 64;
          for (size_t i = at; i < bad.size() && i < at + len; ++i) bad[i] = (uint8_t)(rnd() & 255);
        } else if (bad.size() > 2) {
          for (int k = 0; k < 3; ++k) bad[2 + rnd() % (bad.size() < 40 ? bad.size() - 2 : 38)] ^= (uint8_t)(rnd() & 255);   // block header / code lengths
        }
        Out o2;
        int64_t g2 = fast(bad, &o2);
        if (g2 >= 0) {
          std::vector<uint8_t> ref;
          if (!zlib_ref(bad, &ref) || (int64_t)ref.size() != g2 || memcmp(ref.data(), o2.v.data(), ref.size()) != 0) {
            fprintf(stderr, "accepted a damaged stream that zlib rejects or decodes differently (%zu bytes, level %d, kind %d)\n", src.size(), level, kind);
            return 4;
          }
          ++accepted;
        } else {
          ++declined;
        }
      }
    }
  }
  printf("ok: %ld valid streams, damaged: %ld declined, %ld accepted (= zlib)\n", valid, declined, accepted);
  return 0;
}
