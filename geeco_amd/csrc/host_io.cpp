// Host-side helpers for the TFRecord input pipeline (plain C ABI, no GPU):
// CRC-32C (Castagnoli) with TFRecord masking.  Replaces the checksum that
// tf.data.TFRecordDataset verifies when the reference reads its episodes
// (reference src/data/geeco_gym.py:442-445; writer src/data/data_recorder.py:154-155).
#include <stddef.h>
#include <stdint.h>

static uint32_t g_tab[8][256];
static int g_init = 0;

static void init_tables() {
  const uint32_t poly = 0x82f63b78u;   // reflected CRC-32C polynomial
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ poly : c >> 1;
    g_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = g_tab[0][i];
    for (int t = 1; t < 8; ++t) {
      c = g_tab[0][c & 0xff] ^ (c >> 8);
      g_tab[t][i] = c;
    }
  }
  g_init = 1;
}

extern "C" uint32_t geeco_crc32c(const uint8_t* p, size_t n, uint32_t crc) {
  if (!g_init) init_tables();
  crc = ~crc;
  while (n && ((uintptr_t)p & 7)) {
    crc = g_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
    --n;
  }
  while (n >= 8) {   // slicing-by-8
    uint64_t v = *(const uint64_t*)p ^ crc;
    crc = g_tab[7][v & 0xff] ^ g_tab[6][(v >> 8) & 0xff] ^ g_tab[5][(v >> 16) & 0xff] ^ g_tab[4][(v >> 24) & 0xff] ^
          g_tab[3][(v >> 32) & 0xff] ^ g_tab[2][(v >> 40) & 0xff] ^ g_tab[1][(v >> 48) & 0xff] ^ g_tab[0][v >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) crc = g_tab[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
  return ~crc;
}

// TFRecord stores masked CRCs: ((crc >> 15) | (crc << 17)) + 0xa282ead8
extern "C" uint32_t geeco_masked_crc32c(const uint8_t* p, size_t n) {
  uint32_t c = geeco_crc32c(p, n, 0);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
