/* Development-only entry point (libgeeco_hip_dev.so, scripts/dev/build_dev_lib.sh); not part of include/geeco_hip.h. */
#ifndef GEECO_CONV_BOTTOM_FWD_H_
#define GEECO_CONV_BOTTOM_FWD_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* conv1 -> conv2 of the encoder (graph.py:76-85) in ONE launch: geeco_conv1_fwd_relu_bits[_rgb] followed by
 * geeco_conv2_fwd_relu_fields, bitwise the same y1 / bits / y2 / fields, without conv2 reading y1 back from memory (the
 * y1 halo of a conv2 tile is produced in LDS by conv1 waves beside the conv2 waves; y1 is still written: conv2's filter
 * gradient reads it).  x [G][N][H][W][4] (channel-padded), w1 [G][9][w_cin][32] with w_cin = 3 (the RGB variable as stored)
 * or 4, w2 [G][9][32][48]; bits / fields may be NULL (evaluation / prediction).  H, W even. */
int geeco_conv1_conv2_fwd(const float* x, const float* w1, const float* b1, float* y1, uint32_t* bits, const float* w2,
                          const float* b2, float* y2, uint16_t* fields, int groups, int64_t gs_x, int64_t gs_w1,
                          int64_t gs_b1, int64_t gs_y1, int64_t gs_bits, int64_t gs_w2, int64_t gs_b2, int64_t gs_y2,
                          int64_t gs_fields, int N, int H, int W, int w_cin, void* stream);
#ifdef __cplusplus
}
#endif
#endif
