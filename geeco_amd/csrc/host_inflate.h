// Internal interface of host_inflate.cpp (not part of the C ABI; include/geeco_host.h exports geeco_inflate_fast for tests).
#pragma once
#include <stddef.h>
#include <stdint.h>

#define GEECO_FI_FALLBACK (-3)   // the fast decoder declined (odd header, malformed / truncated data, checksum, no room): ask zlib
#define GEECO_FI_PAD 128         // bytes that must be readable past the end of the input (the bit buffer refills 8 at a time)

// Grows the output buffer to at least `want` bytes keeping its contents; returns the (possibly moved) buffer or NULL.
typedef uint8_t* (*geeco_grow_fn)(void* ctx, size_t want);

// zlib stream src[0, n) -> dst (capacity cap, grown through `grow` when that is not NULL).  Returns the number of bytes
// produced -- then they are exactly what zlib's inflate() produces and the Adler-32 trailer matched -- or GEECO_FI_FALLBACK.
int64_t geeco_fast_inflate(const uint8_t* src, size_t n, geeco_grow_fn grow, void* grow_ctx, uint8_t* dst, size_t cap);
