/*
 * geeco_hip.h -- C ABI of the MI355X (gfx950) kernels behind GEECO's e2evmc training hot path.
 *
 * The reference (ogroth/geeco) has no FFI: every op below replaces a stock TensorFlow-1.15 op that
 * the reference's Python graph instantiates.  Each entry point cites the reference call site it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all tensors are float32, device pointers, NHWC activations, HWIO conv kernels (TF layout);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only enqueues
 *     work on that stream: no allocation, no synchronisation, no host<->device copies, so a caller
 *     may capture a sequence of calls into a hipGraph;
 *   - return value: 0 on success, a negative GEECO_E* code on bad arguments, or a positive
 *     hipError_t if the launch failed; geeco_last_error() gives a thread-local message;
 *   - "groups" let one launch serve several independent instances with identical shapes (the
 *     three encoders of geeco-f): instance g uses ptr + g * group_stride (strides in elements).
 */
#ifndef GEECO_HIP_H_
#define GEECO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEECO_ABI_VERSION 6   /* = the build round that last changed the entry points or their calling conventions */

#define GEECO_EINVAL  (-1)   /* bad shape / alignment / null pointer */
#define GEECO_ENOSUP  (-2)   /* shape outside what the kernels were built for */

int geeco_abi_version(void);
const char* geeco_last_error(void);
/* 0: the product library (csrc/build.sh): only the kernels the measured-best path launches, no environment variable is ever
 * read.  1: the development build (scripts/dev/build_dev_lib.sh, -DGEECO_DEV_KERNELS), which also holds every A/B kernel
 * variant and reads the GEECO_* switches of scripts/dev/SWITCHES.md when GEECO_DEV=1 is set. */
int geeco_has_dev_kernels(void);

/* Diagnostics (bench.py's per-layer table): between _begin and _end on one host thread every
 * conv entry point records the names of the kernels it dispatched; _end returns them ';'-separated
 * (thread-local buffer, valid until the next _begin on that thread). */
void geeco_debug_kernel_trace_begin(void);
const char* geeco_debug_kernel_trace_end(void);

/* ---- dynamic image: src/models/e2evmc/graph.py:30-55 (dynimg), :17-28 (_H/_alpha) -------------
 * frames [N][K][H][W][C] (or, when frame_stride/sample_stride are given, any strided stack of
 * HWC frames) -> out [N][H][W][Cpad] = (D - min_n) / (max_n - min_n + 1e-6), D = sum_t alpha_t X_t.
 * Channels C..Cpad-1 of `out` are written as zero (Cpad is 4 for RGB so conv1 reads float4 pixels).
 * `alpha` is a HOST pointer to K coefficients (geeco_dynimg_alpha fills it).
 * `ws` is a device workspace of geeco_dynimg_ws_bytes(N, H*W*C) bytes.
 * Two-frame form (dyndiff, graph.py:397-400): pass K = 2, frames = cur, frames2 = tgt,
 * frame_stride ignored; otherwise frames2 = NULL. */
void geeco_dynimg_alpha(int K, float* alpha_host);
int64_t geeco_dynimg_ws_bytes(int N, int64_t hwc);
int geeco_dynimg_fwd(const float* frames, const float* frames2, int64_t sample_stride,
                     int64_t frame_stride, const float* alpha_host, int N, int K, int64_t HW,
                     int C, int Cpad, float* out, void* ws, void* stream);

/* RGB-D form of geeco_dynimg_fwd without packing: `rgb` [N][K][HW][3] and `depth` [N][K][HW] stay separate tensors
 * (each with its own sample / frame strides, 16-byte aligned, HW % 4 == 0) and tf.concat([rgb, depth], -1)
 * (estimator.py:169,172) happens in registers; out [N][HW][4] normalised over all four channels (graph.py:47-54).
 * Two-frame form: K = 2, rgb2 / depth2 = the second frame ([N][HW][3], [N][HW]). */
int geeco_dynimg_rgbd_fwd(const float* rgb, const float* rgb2, int64_t sample_stride, int64_t frame_stride,
                          const float* depth, const float* depth2, int64_t dsample_stride, int64_t dframe_stride,
                          const float* alpha_host, int N, int K, int64_t HW, float* out, void* ws, void* stream);
/* All three conv1 inputs of the goal model's dynimg branch (graph.py:386-401) in ONE launch and one pass over the window:
 * buf_out = dynimg of the K-frame stack, diff_out = dynimg of (last frame, target) with the 2-frame coefficients alpha2,
 * cur_out = the last frame channel-padded.  Both images stay in registers across their per-sample min / max (the blocks of
 * a sample meet at a per-sample arrival counter in `ws`) and are stored once, normalised; bitwise the images of
 * geeco_dynimg_fwd / geeco_dynimg_rgbd_fwd.  depth / tgt_depth NULL: RGB ((R, G, B, 0) pixels), else RGB-D from separate
 * tensors.  HW % 4 == 0, 16-byte aligned strides.
 * ws: geeco_goal_dynimgs_ws_bytes(N, HW) bytes that the caller ZERO-FILLS ONCE (per-sample arrival counters, which every call
 * finds and leaves zero, + per-block min / max slots); calls that share a `ws` must be stream-ordered. */
int64_t geeco_goal_dynimgs_ws_bytes(int N, int64_t HW);
int geeco_goal_dynimgs_fwd(const float* rgb, int64_t sample_stride, int64_t frame_stride, const float* tgt_rgb,
                           const float* depth, int64_t dsample_stride, int64_t dframe_stride, const float* tgt_depth,
                           const float* alpha_host, const float* alpha2_host, int N, int K, int64_t HW, float* cur_out,
                           float* buf_out, float* diff_out, void* ws, void* stream);
/* The blocks of a sample wait for each other (bounded: seconds).  A wait that expires -- only if the device did not start the
 * blocks of a launch in index order, see csrc/dynimg.hip -- is never silent: that sample's two images are written as NaN (never
 * with stale min / max; NB a ReLU turns NaN into 0, so the loss behind them can be finite) and the block counts itself into a
 * sticky per-sample word in `ws`.  This entry sums those words
 * over the N samples into *count_host (0 = every image ever produced through this ws was normalised with its sample's true
 * min / max).  It copies N x 64 bytes to the host and SYNCHRONISES `stream`: call it where the host waits for the device
 * anyway (loss read-out, end of an epoch; the reference reads its loss in the same places, estimator.py:263-269).  A workspace
 * that has seen a timeout must be zero-filled again before its next use. */
int geeco_goal_dynimgs_timeouts(const void* ws, int N, void* stream, int64_t* count_host);
/* Process-wide number of polls a block waits before it reports (default 2^22); returns the previous value.  0 makes every block
 * that is not the last of its sample to arrive report at once -- the tests provoke the error path with it. */
unsigned geeco_goal_dynimgs_set_wait_polls(unsigned polls);
/* The same stage read straight from the episodes' resident uint8 frames: replaces the host-side window + division of the
 * reference's input pipeline (_window_v3, src/data/geeco_gym.py:615-631; rgb / 255.0, _parse_v4 :312) AND the fp32 window
 * tensor they produce.  win_ptrs_dev / tgt_ptrs_dev: DEVICE arrays of N addresses; window n = K consecutive [HW][3] uint8
 * frames starting at win_ptrs_dev[n] (4-byte aligned), its target frame [HW][3] uint8 at tgt_ptrs_dev[n].  The tables are
 * read when the kernel RUNS, so a captured graph follows the host repointing them between replays; the frames they point
 * at must stay allocated until that run has finished.  depth / tgt_depth: dense float32 [N][K][HW] / [N][HW] or NULL.
 * Outputs are bitwise those of geeco_gather_windows(divisor 255) followed by geeco_goal_dynimgs_fwd. */
int geeco_goal_dynimgs_u8_fwd(const void* const* win_ptrs_dev, const void* const* tgt_ptrs_dev, const float* depth,
                              int64_t dsample_stride, int64_t dframe_stride, const float* tgt_depth, const float* alpha_host,
                              const float* alpha2_host, int N, int K, int64_t HW, float* cur_out, float* buf_out,
                              float* diff_out, void* ws, void* stream);


/* Copy [npix][C] -> [npix][Cpad] (zero-filled tail).  Used for the "current frame" view
 * rgb[:, -1] (graph.py:387) and for RGB||depth concat (estimator.py:169,172) via src2. */
int geeco_pack_pixels(const float* src, int64_t src_sample_stride, const float* src2,
                      int64_t src2_sample_stride, int N, int64_t HW, int C1, int C2, int Cpad,
                      float* dst, void* stream);

/* On-device window builder (replaces the host-side _window_v3 of src/data/geeco_gym.py:615-631):
 * out[n][k][:] = float(src[starts_dev[n] + k][:]) / divisor.  `src` = an episode's frames resident in
 * HBM ([T][frame_elems], uint8 if src_is_u8 else float32); `starts_dev` = N int32 frame indices on the
 * device; divisor 255 reproduces `rgb /= 255.0` (geeco_gym.py:312) bit-exactly, 1 copies. */
int geeco_gather_windows(const void* src, int src_is_u8, const int* starts_dev, int N, int K,
                         int64_t frame_elems, float divisor, float* out, void* stream);

/* ---- conv encoder: graph.py:76-115 (tf.layers.conv2d 3x3, padding='SAME', bias, ReLU) ----------
 * x [G][N][H][W][Cin], w [G][3][3][Cin][Cout] (HWIO), b [G][Cout], y [G][N][Ho][Wo][Cout],
 * Ho = ceil(H/stride); TF SAME padding (pad_before = pad_total/2, i.e. 0 top/left for stride 2 on
 * even sizes).  Requires Cin % 4 == 0, Cout % 16 == 0. */
int geeco_conv3x3_fwd(const float* x, const float* w, const float* b, float* y, int groups,
                      int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int N, int H, int W,
                      int Cin, int Cout, int stride, int relu, void* ws, void* stream);
/* The encoders' TOP layer (conv8 + bias + ReLU, all `groups` encoders over the same N frames) together with the one-step
 * decoder's state concat (representation_concatenation_v2, graph.py:169-192; jnt_state_list[-1], :388): the launch that sums
 * the split-K slabs also writes state[n][cell * Ctot + feat_off[g] + c] = y[g][n][cell][c] and copies jnt[n][0..J) into the
 * columns [jnt_off, jnt_off + J) of every cell -- the values of geeco_conv3x3_fwd + geeco_state_concat_fwd, one dependent launch
 * fewer.  Returns GEECO_ENOSUP, having launched nothing, when the shape does not take the split-K path (ws NULL, no split for
 * this M, a halo-kernel shape): the caller then runs the two entry points. */
int geeco_conv3x3_fwd_state(const float* x, const float* w, const float* b, float* y, int groups, int64_t gs_x, int64_t gs_w,
                            int64_t gs_b, int64_t gs_y, int N, int H, int W, int Cin, int Cout, int stride, void* ws,
                            const int* feat_off, int Ctot, const float* jnt, int64_t jnt_stride, int jnt_off, int J,
                            float* state, int64_t state_stride, void* stream);

/* `ws` (may be NULL): device workspace of geeco_conv3x3_fwd_ws_bytes(...) bytes; layers whose
 * output is too small to fill 256 CUs (conv6..8) split their K loop over blocks through it. */
int64_t geeco_conv3x3_fwd_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride);

/* Conv2DBackpropInput fused with the ReluGrad of the layer below (autodiff of graph.py:76-115 via
 * estimator.py:243-244):  dx = conv3x3_transpose(dz, w) * (ymask > 0).
 * dz [G][N][Ho][Wo][Cout], wt [G][3][3][Cout][Cin] (= geeco_transpose_hwio(w)), w (optional) the
 * HWIO kernel itself: the LDS-halo kernel of the 32->48 stride-2 layer reads it directly,
 * ymask [G][N][H][W][Cin] = forward output of the layer below (NULL: no mask), dx like ymask.
 * stride 1 or 2; all stride*stride parity classes run in one launch; `ws` as for the forward. */
int geeco_conv3x3_dgrad(const float* dz, const float* w, const float* wt, const float* ymask,
                        float* dx, int groups, int64_t gs_dz, int64_t gs_w, int64_t gs_wt,
                        int64_t gs_dx, int N, int H, int W, int Cin, int Cout, int stride, void* ws,
                        void* stream);
int64_t geeco_conv3x3_dgrad_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride);
/* 0 if geeco_conv3x3_dgrad, GIVEN the HWIO kernel `w`, runs a kernel that reads `w` itself for this shape (the LDS-staged
 * kernels, and the gather GEMM whenever Cout % 16 == 0: it transposes the kernel tile on its way into LDS); the caller then
 * need not keep the transposed copy `wt` up to date and may pass NULL for it.  1 if `wt` is read. */
int geeco_conv3x3_dgrad_needs_wt(int H, int W, int Cin, int Cout, int stride);
int geeco_conv3x3_dgrad_relu_fields_supported(int H, int W, int Cin, int Cout, int stride);

/* Conv2DBackpropFilter + BiasAddGrad:  dw[ky][kx][ci][co] = sum_m x[pix(m,ky,kx)][ci] dz[m][co],
 * db[co] = sum_m dz[m][co].  dw/db are OVERWRITTEN (not accumulated).
 * `ws`: device workspace of geeco_conv3x3_wgrad_ws_bytes(...) bytes (split-K partial slabs). */
int64_t geeco_conv3x3_wgrad_ws_bytes(int groups, int N, int H, int W, int Cin, int Cout, int stride);
int geeco_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, int groups,
                        int64_t gs_x, int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H,
                        int W, int Cin, int Cout, int stride, void* ws, void* stream);

/* Encoder bottom, backward, fused (autodiff of graph.py:76-85 via estimator.py:243-244): conv2's input gradient and
 * conv1's filter/bias gradient in one kernel - conv1's input is data, so dz1 = (y1 > 0) * conv2_dgrad(dz2) has a
 * single consumer and never has to reach HBM:
 *   dw1[g][3][3][real_channels][32], db1[g][32]  <-  x [G][N][H][W][4], dz1
 * real_channels = 3: x is RGB zero-padded to 4 channels; dw1 has the reference's own [3][3][3][32] layout (the pad
 * channel has no row); real_channels = 4: RGB-D, all four input channels are real.
 * dz2 [G][N][H/2][W/2][48], w2 [G][3][3][32][48] (HWIO), y1 [G][N][H][W][32] (conv1's output: the ReLU mask).
 * dz1 (optional, may be NULL): if given, dz1 is ALSO written ([G][N][H][W][32], group stride gs_y1).
 * Shapes are fixed to the reference encoder's conv1 (4 -> 32, stride 1) / conv2 (32 -> 48, stride 2); H, W even.
 * Equivalent to geeco_conv3x3_dgrad(conv2) followed by geeco_conv3x3_wgrad(conv1); dw1/db1 are OVERWRITTEN. */
int64_t geeco_conv2_dgrad_conv1_wgrad_ws_bytes(int groups);
int geeco_conv2_dgrad_conv1_wgrad(const float* dz2, const float* w2, const float* y1, const float* x,
                                  float* dw1, float* db1, float* dz1, int groups, int64_t gs_dz2,
                                  int64_t gs_w2, int64_t gs_y1, int64_t gs_x, int64_t gs_dw1,
                                  int64_t gs_db1, int N, int H, int W, int real_channels, void* ws,
                                  void* stream);

/* Deferred slab sums.  The filter-gradient kernels leave one partial slab per split in `ws` and finish with a small
 * slab-sum launch.  The *_partial forms skip that launch and describe it in *pending instead (pending->S == 0: the
 * kernel wrote dw/db itself, nothing is pending); geeco_slab_reduce_batch then finishes up to GEECO_SLAB_REDUCE_MAX
 * of them in ONE launch (a training step: one per part of the backward instead of one per layer).  The `ws` of a
 * pending item must stay untouched until the batch has run.  Results are bitwise those of the plain calls (same
 * summation order).
 * reserved_cus (0..128; the forms below and geeco_conv2_dgrad_conv1_wgrad_bits): data parallel -- the two persistent
 * one-block-per-CU kernels at the bottom of the backward (conv2's filter gradient, the fused conv2-dgrad + conv1-wgrad)
 * normally occupy every CU while the early gradient bucket is all-reduced beside them; k > 0 makes THIS launch leave k CUs
 * to the collective's workgroups ((256 - k) / groups blocks per encoder; workspaces are sized for k = 0 and fit any k; other
 * kernels ignore it).  A launch argument since ABI 4 (a process-wide setter before): nothing outlives the call.  The
 * reference has no counterpart (no distributed code: SURVEY.md 2). */
typedef struct geeco_slab_reduce {
  const float* part;      /* [groups][S][KC + Cout] partial slabs */
  float* dw;              /* [groups] x KC floats, group stride gs_dw */
  float* db;              /* [groups] x Cout floats, group stride gs_db (may be NULL) */
  int64_t gs_dw, gs_db, KC;
  int32_t S, Cout, groups, reserved;
} geeco_slab_reduce;
#define GEECO_SLAB_REDUCE_MAX 8
int geeco_conv3x3_wgrad_partial(const float* x, const float* dz, float* dw, float* db, int groups,
                                int64_t gs_x, int64_t gs_dz, int64_t gs_dw, int64_t gs_db, int N, int H,
                                int W, int Cin, int Cout, int stride, void* ws, void* stream,
                                geeco_slab_reduce* pending, int reserved_cus);
int geeco_conv2_dgrad_conv1_wgrad_partial(const float* dz2, const float* w2, const float* y1, const float* x,
                                          float* dw1, float* db1, float* dz1, int groups, int64_t gs_dz2,
                                          int64_t gs_w2, int64_t gs_y1, int64_t gs_x, int64_t gs_dw1,
                                          int64_t gs_db1, int N, int H, int W, int real_channels, void* ws,
                                          void* stream, geeco_slab_reduce* pending, int reserved_cus);
int geeco_slab_reduce_batch(const geeco_slab_reduce* items, int n, void* stream);
/* ... and, as one more block of the same launch, geeco_adam_prepare (below): the step counter and lr_t depend on nothing the
 * slab sums touch and only have to be in place before geeco_adam_tf -- the training step's last slab-sum launch carries them
 * instead of a dependent launch of their own.  n may be 0. */
int geeco_slab_reduce_batch_prepare(const geeco_slab_reduce* items, int n, int64_t* global_step_dev, float lr, float beta1,
                                    float beta2, float* scal_dev, void* stream);
/* TWO independent filter gradients of the 64 x 64-tile generic kernel as ONE grid (the model: conv7's + conv8's, both ready
 * once conv8's input gradient exists; problem 0 = the longer one).  Arguments per problem as geeco_conv3x3_wgrad; common
 * groups / stride; pending2: NULL or TWO items (deferred slab sums as geeco_conv3x3_wgrad_partial; S = 0 where the kernel wrote
 * the gradient directly).  Bitwise the results of two geeco_conv3x3_wgrad calls.  GEECO_ENOSUP for shapes outside the paired
 * kernel (Cin = 256, Cout % 64 == 0, stride 2): launch twice then. */
int geeco_conv3x3_wgrad_pair(const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0, int64_t gs_dz0,
                             int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                             const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                             int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1,
                             int groups, int stride, void* stream, geeco_slab_reduce* pending2);
/* ... and with conv7's INPUT gradient beside them (all three need only dz7; each alone fills the 256 CUs badly): ONE
 * heterogeneous grid, the filter-gradient blocks first.  Leading arguments as geeco_conv3x3_dgrad (stride 2; ws = its split-K
 * workspace, required), then the two problems of geeco_conv3x3_wgrad_pair.  Bitwise the separate calls.  GEECO_ENOSUP when
 * any of the three is outside the kernels this launch combines (the gather GEMM's 64 x 64 x 16 tiles; the paired 64 x 64
 * filter-gradient tiles): nothing has been launched, use the separate entry points.  x1 == NULL: ONE filter gradient beside
 * the input gradient (the model: conv8's pair, then conv7's). */
int geeco_conv_top_bwd(const float* dz, const float* w, const float* wt, const float* ymask, float* dx, int64_t gs_dz,
                       int64_t gs_w, int64_t gs_wt, int64_t gs_dx, int N, int H, int W, int Cin, int Cout, void* ws,
                       const float* x0, const float* dz0, float* dw0, float* db0, int64_t gs_x0, int64_t gs_dz0,
                       int64_t gs_dw0, int64_t gs_db0, int N0, int H0, int W0, int Cin0, int Cout0, void* ws0,
                       const float* x1, const float* dz1, float* dw1, float* db1, int64_t gs_x1, int64_t gs_dz1,
                       int64_t gs_dw1, int64_t gs_db1, int N1, int H1, int W1, int Cin1, int Cout1, void* ws1,
                       int groups, int stride, void* stream, geeco_slab_reduce* pending2);

/* ReLU sign bits as the ReluGrad mask of the encoder bottom.  conv1's output y1 (805 MB at the bench shape) is read by
 * the fused bottom backward only for its sign; geeco_conv1_fwd_relu_bits is conv1's forward (4 -> 32, stride 1, bias,
 * ReLU: geeco_conv3x3_fwd on those shapes) that ALSO writes one uint32 per pixel,
 *   bits[g][n][y][x]: Hp = geeco_relu_bits_rows(H) rows of Wp = geeco_relu_bits_pitch(W) words per image (whole 8 x 64
 *   tiles), group stride gs_bits words; bit (c & 3) * 8 + (c >> 2) set iff y1[g][n][y][x][c] > 0.
 * Only the words of real pixels are written: the caller zero-fills the array ONCE, the backward relies on zero padding.
 * geeco_conv2_dgrad_conv1_wgrad_bits is geeco_conv2_dgrad_conv1_wgrad taking those words instead of y1 (no dz1
 * output; pending: NULL = finish the slab sum, else defer it as geeco_conv2_dgrad_conv1_wgrad_partial does). */
int64_t geeco_relu_bits_pitch(int W);
int64_t geeco_relu_bits_rows(int H);
int geeco_conv1_fwd_relu_bits(const float* x, const float* w, const float* b, float* y, uint32_t* bits, int groups,
                              int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_bits, int N, int H,
                              int W, void* stream);
/* ... with the RGB model's kernel variable w3 [G][3][3][3][32] as it is stored (x stays channel-padded; the pad channel's
 * kernel rows count as zero): bitwise the same y / bits as with the padded copy, which then need not be re-derived after
 * every optimiser step. */
int geeco_conv1_fwd_relu_bits_rgb(const float* x, const float* w3, const float* b, float* y, uint32_t* bits, int groups,
                                  int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_bits, int N, int H,
                                  int W, void* stream);
/* The same idea one layer up: conv2's forward (32 -> 48, stride 2, bias, ReLU; x [G][N][H][W][32]) that also writes the
 * sign fields of its output y2, and conv3's input gradient (48 -> 64, stride 2; dz [G][N][H/2][W/2][64], dx = d(y2)
 * [G][N][H][W][48]) masked by those fields instead of by y2 itself (302 MB at the bench shape).
 *   fields[g][n][y][x][q], q = 0..3: uint16, bit 4 i + j set iff y2[g][n][y][x][16 i + 4 q + j] > 0; rows / columns
 *   padded to whole 8 x 64 tiles (geeco_relu_fields_elems uint16 per encoder), group stride gs_fields elements.
 * reserved_cus (round 5): as for the two launches behind it in the data-parallel step (geeco_conv3x3_wgrad_partial): this persistent
 *   one-block-per-CU kernel is the first of part 2 and leaves that many CUs to the collective running beside it; 0 otherwise. */
int64_t geeco_relu_fields_elems(int N, int H, int W);
int geeco_conv2_fwd_relu_fields(const float* x, const float* w, const float* b, float* y, uint16_t* fields, int groups,
                                int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_fields, int N, int H,
                                int W, void* stream);
int geeco_conv3_dgrad_relu_fields(const float* dz, const float* w, const uint16_t* y2_fields, float* dx, int groups,
                                  int64_t gs_dz, int64_t gs_w, int64_t gs_fields, int64_t gs_dx, int N, int H, int W,
                                  void* stream, int reserved_cus);
/* ... and one more layer up: conv3's forward (48 -> 64, stride 2) writing byte sign fields of its output y3, and the
 * LDS-staged input-gradient kernel of the next layer (the shapes geeco_conv3x3_dgrad_relu_fields_supported reports,
 * e.g. conv4: 64 -> 128) masked by them instead of by y3 (100 MB at the bench shape):
 *   fields[g][n][y][x][Cin / 8] bytes: byte (T >> 1) * 4 + q, bit 4 (T & 1) + j set iff y[g][n][y][x][16 T + 4 q + j] > 0
 *   (T = 16-channel tile, q = channel quad inside it); group stride gs_fields bytes; no padding. */
int geeco_conv3_fwd_relu_fields(const float* x, const float* w, const float* b, float* y, uint8_t* fields, int groups,
                                int64_t gs_x, int64_t gs_w, int64_t gs_b, int64_t gs_y, int64_t gs_fields, int N, int H,
                                int W, void* stream);
int geeco_conv3x3_dgrad_relu_fields(const float* dz, const float* w, const uint8_t* y_fields, float* dx, int groups,
                                    int64_t gs_dz, int64_t gs_w, int64_t gs_fields, int64_t gs_dx, int N, int H, int W,
                                    int Cin, int Cout, int stride, void* stream);
int geeco_conv2_dgrad_conv1_wgrad_bits(const float* dz2, const float* w2, const uint32_t* y1_bits, const float* x,
                                       float* dw1, float* db1, int groups, int64_t gs_dz2, int64_t gs_w2,
                                       int64_t gs_bits, int64_t gs_x, int64_t gs_dw1, int64_t gs_db1, int N, int H, int W,
                                       int real_channels, void* ws, void* stream, geeco_slab_reduce* pending,
                                       int reserved_cus);

/* [G][9][A][B] -> [G][9][B][A] per-tap transpose (HWIO -> HWOI) feeding geeco_conv3x3_dgrad. */
int geeco_transpose_hwio(const float* w, float* wt, int groups, int64_t gs_w, int64_t gs_wt, int Cin,
                         int Cout, void* stream);

/* All derived weight copies of an encoder stack in one launch (the per-step chain after Adam):
 * for l < nlayers: wt[l] = geeco_transpose_hwio(w[l]) with (Cin[l], Cout[l]), group strides gs_w (kernels,
 * common) and gs_wt[l]; and, if pad_src/pad_dst are given, conv1's kernel [G][9][pad_Cin][pad_Cout]
 * (stride gs_w) zero-padded to [G][9][pad_Cin_padded][pad_Cout] (stride gs_pad) = geeco_pad_mid per group. */
int geeco_derive_conv_weights(int nlayers, const float* const* w, float* const* wt, const int* Cin,
                              const int* Cout, const int64_t* gs_wt, int groups, int64_t gs_w,
                              const float* pad_src, float* pad_dst, int pad_Cin, int pad_Cin_padded,
                              int pad_Cout, int64_t gs_pad, void* stream);

/* src [A][B][C] -> dst [A][Bd][C], copying min(B,Bd) middle rows and zero-filling the rest: pads
 * conv1's RGB kernel [9][3][Co] to [9][4][Co] and un-pads its gradient. */
int geeco_pad_mid(const float* src, float* dst, int64_t A, int B, int Bd, int C, void* stream);

/* ---- state concat: graph.py:123-192 (tile jnt 2x2, concat along channels, flatten) -------------
 * Builds state [N][cells*(sum Cf + J)] from up to 3 feature maps [N][cells][Cf_i] and jnt [N][J];
 * `jnt_pos` = number of feature maps placed BEFORE the joint block (1: state_concatenation and
 * representation_concatenation, 2: representation_concatenation_v2).  `sub_from` (optional,
 * graph.py:369 'residual'): feature 0 is written as sub_from - feat0.
 * jnt rows are read at jnt + n * jnt_stride. */
int geeco_state_concat_fwd(const float* const* feats, const int* feat_ch, int nfeat, int jnt_pos,
                           const float* jnt, int64_t jnt_stride, int J, const float* sub_from,
                           int N, int cells, float* state, int64_t state_stride, void* stream);
/* Scatter scale * d(state) back into d(feat_i), applying the ReLU mask of the encoder's last layer
 * (feats_fwd_i > 0).  dfeats[i] may be NULL to skip.  accumulate != 0 adds into dfeats.  The
 * 'residual' target mode (graph.py:369: tgt_feat - feat) calls it with scale = -1 for feat and once
 * more with feats_fwd = tgt_feat, scale = +1, accumulate over the window. */
int geeco_state_concat_bwd(const float* dstate, int64_t dstate_stride, const float* const* feats_fwd,
                           float* const* dfeats, const int* feat_ch, int nfeat, int jnt_pos, int J,
                           int N, int cells, int accumulate, float scale, void* stream);

/* ---- dense GEMM used by the LSTM gate matmul and its backward (graph.py:217-225) ---------------
 * C[M][N] (ldc) = op(A) op(B) (+ C if accumulate).  ta/tb: 0 = as stored, 1 = transposed.
 * A is [M][K] (lda) or, if ta, [K][M]; B is [K][N] (ldb) or, if tb, [N][K].
 * `ws`: geeco_gemm_ws_bytes(M,N,K) bytes (split-K slabs). */
int64_t geeco_gemm_ws_bytes(int M, int N, int K);
int geeco_gemm_f32(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb, float* C,
                   int64_t ldc, int M, int N, int K, int accumulate, void* ws, void* stream);

/* ---- LSTM cell: tf.nn.rnn_cell.LSTMCell(num_units=H, state_is_tuple=False), graph.py:217-225 ----
 * z [N][4H] = [x|h_prev] W (from geeco_gemm_f32), gate order i, j, f, o, forget_bias 1.0:
 *   c = sigmoid(f + 1) c_prev + sigmoid(i) tanh(j);  h = sigmoid(o) tanh(c).
 * `gates` [N][4H] receives the activated gates (si, tj, sf, so) for the backward pass.
 * c_prev may be NULL (zero state, graph.py:218-220). */
int geeco_lstm_gates_fwd(const float* z, const float* bias, const float* c_prev, float* c, float* h,
                         float* gates, int N, int H, void* stream);
/* The FIRST step of the cell (zero state, graph.py:218-220: z = x wx alone) as two launches instead of three: the gate GEMM
 * (geeco_gemm_f32's kernel and K split) and the gate math with the split-K slab sum inside it.  x [N][D] (ldx), wx [D][4H]
 * (ldw); z, c, h, gates as geeco_lstm_gates_fwd; ws: geeco_gemm_ws_bytes(N, 4H, D).  Bitwise equal to geeco_gemm_f32 +
 * geeco_lstm_gates_fwd(c_prev = NULL). */
int geeco_lstm_input_step_fwd(const float* x, int64_t ldx, const float* wx, int64_t ldw, const float* bias, float* z, float* c,
                              float* h, float* gates, int N, int H, int D, void* ws, void* stream);
/* dz [N][4H] from dh, dc (either may be NULL = 0); dc_prev (optional) out. */
int geeco_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c, const float* dh,
                         const float* dc, float* dz, float* dc_prev, int N, int H, void* stream);
/* TWO launches for everything a single LSTM step's backward needs after the gate gradients dz [N][4H] (the goal model's
 * dynimg branch runs one step: graph.py:405-407; autodiff of :217-225 + :169-192): dwx [D][4H] = x^T dz, db [4H] = column
 * sums of dz, dx [N][D] = dz wx^T and - if nfeat > 0 - the state-concat backward of dx (geeco_state_concat_bwd with
 * scale 1, accumulate 0: dfeats[i][n][cell][c] = (feats_fwd[i] > 0) * dx[n][cell * Ctot + off_i + c]).  Grid 1: the
 * tiles of dwx, the split-K tiles of dx and the column sums side by side; grid 2: dx's slab sum with the scatter in its
 * epilogue.  Replaces the five launches geeco_gemm_f32 (ta) + geeco_colsum + geeco_gemm_f32 (tb: split-K + reduce) +
 * geeco_state_concat_bwd.  dx: same K / slab order as geeco_gemm_f32, bitwise the same for every N.  dwx: ONE K loop over
 * the N rows (no split), bitwise equal to geeco_gemm_f32 where that does not split K either, i.e. N < 128 (the batch sizes
 * of the reference: params.py:26); for N >= 128 geeco_gemm_f32 splits the batch dimension and the two differ in rounding
 * (tests/test_kernels_gpu.py: tolerance there, bitwise below).  ws: geeco_lstm_step_bwd_ws_bytes bytes.
 * `pending` (may be NULL): the batch sums geeco_lstm_step_heads_fwd_bwd left behind (weight / bias gradients of fc1 and the
 * heads, loss means); they run as the first blocks of grid 1, beside the tiles of dwx / dx. */
typedef struct geeco_heads_finish { unsigned char opaque[1024]; } geeco_heads_finish;    /* filled and read by the library only */
int64_t geeco_lstm_step_bwd_ws_bytes(int N, int D, int H4);
int geeco_lstm_step_bwd(const float* x, int64_t ldx, const float* dz, int64_t ldz, const float* wx, int64_t ldw, float* dwx,
                        int64_t lddw, float* db, float* dx, int64_t lddx, int N, int D, int H4,
                        const float* const* feats_fwd, float* const* dfeats, const int* feat_ch, int nfeat, int jnt_pos,
                        int J, int cells, void* ws, const geeco_heads_finish* pending, void* stream);
/* column sums: out[j] = sum_i a[i][j] (bias gradients). */
int geeco_colsum(const float* a, int64_t lda, int M, int N, float* out, int accumulate, void* stream);

/* ---- fc1 + heads + losses, forward and backward ------------------------------------------------
 * graph.py:229-259 (fc1 ReLU, linear heads), :430-500 (losses), estimator.py:206-239 (targets,
 * loss composition).  `nheads` <= 5 linear heads [Hfc][size_i] are described by parallel arrays:
 *   head_kind 0: tf.losses.mean_squared_error against target rows (size_i floats at
 *                targets[i] + n * target_stride[i]);
 *   head_kind 1: softmax cross-entropy against one_hot(rint(targets[i][n * stride]) + 1)
 *                (the gripper command, estimator.py:213-215);
 *   loss = sum_i head_weight[i] * L_i (cartesian: 1, 1, lambda_aux, lambda_aux; velocity: all 1),
 *   every L_i a mean over the LOCAL batch N; gradients are scaled by `loss_scale`.
 * Cartesian mode heads: pred_cmd_ee (MSE vs cmd[:, :3]), logits_cmd_grp (CE vs cmd[:, 3]),
 * pred_aux_ee, pred_aux_obj (MSE vs features[...][:, -1, :3]).
 * Outputs: preds [N][sum size_i], losses[1 + nheads] = {total, L_0, ...} (unscaled local means); when
 * `backward` != 0 also dh [N][H] and d_fc1_w, d_fc1_b, d_heads_w[i], d_heads_b[i] (overwritten).
 * Two launches (round 5): one workgroup PER SAMPLE for everything that is independent per sample (fc1, heads, loss terms, and
 * back to dh), then a few blocks for what sums over the batch (weight / bias gradients, loss means; sums over n ascending).
 * H <= 128 and Hfc in {64, 128}; other sizes run a single-workgroup kernel.  N <= 4096, sum size_i <= 32. */
int geeco_heads_loss_fwd_bwd(const float* h, const float* fc1_w, const float* fc1_b, int nheads,
                             const float* const* heads_w, const float* const* heads_b,
                             const int* head_size, const int* head_kind, const float* head_weight,
                             const float* const* targets, const int64_t* target_stride,
                             float loss_scale, int N, int H, int Hfc, float* preds, float* losses,
                             int backward, float* dh, float* d_fc1_w, float* d_fc1_b,
                             float* const* d_heads_w, float* const* d_heads_b, float* ws, void* stream);
int64_t geeco_heads_ws_bytes(int N, int H, int Hfc);
/* The decoder of the models that run ONE LSTM step from the zero state (goal model, dynimg branch: graph.py:405-407, 217-260):
 * geeco_lstm_input_step_fwd + geeco_heads_loss_fwd_bwd + (backward) geeco_lstm_gates_bwd as TWO launches: the gate GEMM
 * (x [N][D] wx [D][4H], split-K slabs in gemm_ws), then one workgroup per sample that sums its slabs, runs the gate math
 * (z, c, h, gates written as geeco_lstm_input_step_fwd does), fc1, the heads and the loss terms and -- `backward` != 0 -- returns
 * through d(h) to the gate gradients dz [N][4H] (no dh output: nothing else reads it).  Arguments as the three entry points.
 * `pending` NULL: the batch sums (loss means; weight / bias gradients) run here as a third small launch; non-NULL: they are
 * described in *pending and geeco_lstm_step_bwd runs them inside its own first grid -- losses and the heads' / fc1's
 * gradients are valid only after that call.  GEECO_ENOSUP (nothing launched) outside H <= 128, Hfc in {64, 128}. */
int geeco_lstm_step_heads_fwd_bwd(const float* x, int64_t ldx, const float* wx, int64_t ldw, const float* bias, float* z,
                                  float* c, float* h, float* gates, int N, int H, int D, void* gemm_ws, const float* fc1_w,
                                  const float* fc1_b, int nheads, const float* const* heads_w, const float* const* heads_b,
                                  const int* head_size, const int* head_kind, const float* head_weight,
                                  const float* const* targets, const int64_t* target_stride, float loss_scale, int Hfc,
                                  float* preds, float* losses, int backward, float* dz, float* d_fc1_w, float* d_fc1_b,
                                  float* const* d_heads_w, float* const* d_heads_b, float* heads_ws,
                                  geeco_heads_finish* pending, void* stream);

/* ---- optimiser: tf.train.AdamOptimizer(lr).minimize, estimator.py:105,243-244 ------------------
 * TF semantics (epsilon outside the bias correction):
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps),
 *   lr_t = lr sqrt(1-b2^t)/(1-b1^t).
 * geeco_adam_prepare increments the DEVICE-resident global step t (tf.train.get_global_step) and
 * writes lr_t to scal_dev[0]; keeping both on the device lets a captured hipGraph replay the step.
 * geeco_adam_tf is one fused pass over the flat parameter arena: grad_scale multiplies g first
 * (1/world after an all-reduce SUM); l2 adds l2 * p to g (tf.contrib.layers.l2_regularizer,
 * graph.py:13-15).  Arenas must be 16-byte aligned. */
int geeco_adam_prepare(int64_t* global_step_dev, float lr, float beta1, float beta2, float* scal_dev,
                       void* stream);
int geeco_adam_tf(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_t_dev,
                  float beta1, float beta2, float eps, float grad_scale, float l2, void* stream);
/* The same update over up to GEECO_ADAM_SEGMENTS_MAX pieces of the arena (offsets and counts in floats, multiples of 4), each piece
 * with its own gradient source; g_out != NULL: every piece's (unscaled) gradients are also stored at its place in that arena.
 * Any partition of the arena gives bitwise geeco_adam_tf's result.  Data parallel (round 5): the variables whose gradients came
 * with the early bucket are updated while the late bucket is still on the wire; the encoders' conv1 / conv2 follow, their
 * gradients read straight from the late bucket's staging buffer (no unpack copies).  The reference has one optimiser op per
 * variable and no distributed code (estimator.py:243-244). */
#define GEECO_ADAM_SEGMENTS_MAX 8
typedef struct geeco_adam_segment {
  const float* g;       /* gradients of the piece (16-byte aligned) */
  int64_t p_off, count; /* its place in the parameter / moment arenas */
} geeco_adam_segment;
int geeco_adam_tf_segments(float* p, float* g_out, float* m, float* v, const geeco_adam_segment* segs, int nseg,
                           const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale, float l2,
                           void* stream);
/* sum of squares of an arena (for the L2 regularisation loss term); out[0] overwritten. */
int geeco_sumsq(const float* p, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* GEECO_HIP_H_ */
