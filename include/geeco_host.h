/* libgeeco_host.so -- host side of the input pipeline (plain C ABI, no GPU, no torch types).
 *
 * Replaces what tf.data does on the reference's worker threads when it reads an episode
 * (src/data/geeco_gym.py:442-445 TFRecordDataset(compression_type='ZLIB', num_parallel_reads=num_threads);
 *  :291-315 _parse_v4 = tf.parse_single_sequence_example + reshape + rgb / 255):
 * zlib inflate, TFRecord framing with masked CRC-32C, the SequenceExample field scan and the
 * float-list -> array copies, in native code that holds no Python lock, so that `num_threads` reader threads
 * (geeco_amd/input_fn.py) really run side by side.
 *
 * Thread safety: every function is re-entrant; an episode handle may be used by one thread at a time.
 * Errors: functions return NULL / a negative value and leave a message for the calling THREAD in
 * geeco_host_last_error().  No global mutable state apart from the CRC tables (filled once, idempotent).
 */
#ifndef GEECO_HOST_H_
#define GEECO_HOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEECO_HOST_ABI_VERSION 4

int geeco_host_abi_version(void);
const char* geeco_host_last_error(void);

/* CRC-32C (Castagnoli), `crc` = running value (0 to start); TFRecord's masked form
 * ((crc >> 15 | crc << 17) + 0xa282ead8) [TF1.15 lib/hash/crc32c.h]. */
uint32_t geeco_crc32c(const uint8_t* p, size_t n, uint32_t crc);
uint32_t geeco_masked_crc32c(const uint8_t* p, size_t n);

/* ---- one episode file = one (zlib-compressed) TFRecord stream holding a tf.train.SequenceExample ---------------
 * geeco_episode_open: reads the file, inflates it (compression: 0 none, 1 zlib, 2 gzip), walks the record framing
 * (verify_crc != 0: length and payload checksums of EVERY record are checked, as TFRecordDataset does) and indexes the
 * feature lists of the FIRST record (the reference writes one record per file, data_recorder.py:134-156).
 * Nothing is decoded until a geeco_episode_read_* call asks for a list. */
typedef struct geeco_episode geeco_episode;

geeco_episode* geeco_episode_open(const char* path, int compression, int verify_crc);
void geeco_episode_close(geeco_episode* ep);

int64_t geeco_episode_num_records(const geeco_episode* ep);
int64_t geeco_episode_inflated_bytes(const geeco_episode* ep);
int geeco_episode_num_lists(const geeco_episode* ep);
/* name of feature list i (NUL terminated, owned by the handle) */
const char* geeco_episode_list_name(const geeco_episode* ep, int i);
/* frames of a feature list; -1 when the record has no such list (tf.parse_single_sequence_example would raise) */
int64_t geeco_episode_list_frames(const geeco_episode* ep, const char* name);
/* kind of frame 0: 1 bytes list, 2 float list, 3 int64 list, 0 empty feature; values = its value count */
int geeco_episode_list_kind(const geeco_episode* ep, const char* name, int64_t* values);

/* Copy a whole feature list into a dense [frames, values_per_frame] array.  Fails (< 0) when the list is missing,
 * has another frame count, holds another kind, or any frame has another number of values.
 *  _f32: FloatList  -> float32
 *  _i64: Int64List  -> int64
 *  _u8 : FloatList  -> uint8; *exact = 1 iff EVERY value was an integer in [0, 255] (uint8 images are recorded as float
 *        lists, src/data/utils/tfrecord.py:73-74); with *exact == 0 the destination holds garbage and the caller reads
 *        the list as float32 instead. */
int geeco_episode_read_f32(const geeco_episode* ep, const char* name, float* dst, int64_t frames, int64_t values_per_frame);
int geeco_episode_read_i64(const geeco_episode* ep, const char* name, int64_t* dst, int64_t frames, int64_t values_per_frame);
int geeco_episode_read_u8(const geeco_episode* ep, const char* name, uint8_t* dst, int64_t frames, int64_t values_per_frame,
                          int* exact);

/* Raw DEFLATE / zlib / gzip decoder used by geeco_episode_open (exposed for tests and the reader benchmark):
 * inflates `src` into `dst` (capacity `cap`); returns the number of bytes produced, -1 on a malformed stream,
 * -2 when `cap` is too small.  format: 1 zlib (RFC 1950, Adler-32 checked), 2 gzip (RFC 1952, CRC-32 checked). */
int64_t geeco_inflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int format);
/* The reader's own table-driven zlib decoder alone (csrc/host_inflate.cpp; geeco_episode_open tries it first and hands whatever
 * it declines to zlib): same result as geeco_inflate(format 1) for every stream it accepts, -2 when `cap` is too small,
 * -3 when it declines the stream (malformed, truncated, checksum mismatch, preset dictionary). */
int64_t geeco_inflate_fast(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
/* 0: geeco_episode_open inflates with zlib only (A/B measurements, tests); 1 (default): table-driven decoder first. */
void geeco_host_set_fast_inflate(int on);
/* geeco_episode_close keeps up to `max_buffers` inflate buffers (default 8, <= 6 GiB in all) mapped for the next
 * geeco_episode_open (the reader sets its thread count + 1); geeco_host_release_buffers frees the ones kept now (the reader
 * calls it when an epoch's last episode has been read: once every episode sits in the HBM cache no reader runs again). */
void geeco_host_set_buffer_limit(int max_buffers);
int geeco_host_spare_buffers(void);     /* how many are kept right now */
void geeco_host_release_buffers(void);

#ifdef __cplusplus
}
#endif
#endif  /* GEECO_HOST_H_ */
