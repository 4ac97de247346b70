"""Per-kernel parity: HIP (through the C-ABI) vs the CPU oracle on the same seeded inputs.

fp32 tolerances are written next to each check; summation order differs from the oracle's
(MFMA k-ordered fmaf chains, split-K slabs), so equality is to rounding, not bitwise.
"""
import numpy as np
import pytest
import torch

from oracle import geeco_oracle as O

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol, what=''):
  a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
  b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
  err = np.abs(a - b)
  tol = atol + rtol * np.abs(b)
  assert a.shape == b.shape, (what, a.shape, b.shape)
  assert np.all(err <= tol), '%s: max err %.3e (tol %.3e) at %s' % (
      what, err.max(), tol.flat[err.argmax()], np.unravel_index(err.argmax(), err.shape))


CONV_CASES = [
    # N, H, W, Cin, Cout, stride
    (2, 16, 16, 4, 32, 1),      # conv1-like (RGB padded to 4)
    (2, 16, 16, 32, 48, 2),     # conv2-like
    (1, 12, 20, 48, 64, 2),     # conv3-like, non-square
    (2, 40, 136, 48, 64, 2),    # conv3 chunked-halo dgrad: several 8x64 tiles, ragged last column of tiles
    (3, 8, 8, 64, 128, 2),
    (2, 9, 7, 16, 16, 2),       # odd sizes: SAME pads (1,1)
    (2, 4, 4, 192, 256, 2),     # small-M path
    (4, 2, 2, 256, 256, 2),     # conv8-like: 2x2 -> 1x1
    (1, 10, 10, 16, 64, 1),     # stride 1, 16 channels
    (2, 64, 64, 32, 48, 2),     # conv2 halo kernel, exact tiles
    (3, 40, 72, 32, 48, 2),     # conv2 halo kernel, ragged tiles (Ho = 20, Wo = 36)
    (16, 64, 64, 64, 128, 2),   # conv4-like at a size that takes the 128x128 tile path
    (8, 32, 32, 128, 192, 2),   # conv5-like (128-row wgrad tiles)
    (8, 64, 64, 128, 192, 2),   # conv5-like with 128 M tiles: the forward takes the 64 x 96 tiles (256 blocks instead of 384)
    (2, 32, 32, 4, 32, 1),      # conv1 shape
    (1, 20, 44, 4, 32, 1),      # conv1 shape, ragged
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', CONV_CASES)
def test_conv3x3_fwd(dev, N, H, W, Cin, Cout, stride):
  from geeco_amd import ops
  r = np.random.default_rng(7)
  x = r.standard_normal([N, H, W, Cin]).astype(np.float32)
  w = (r.standard_normal([3, 3, Cin, Cout]) / np.sqrt(9 * Cin)).astype(np.float32)
  b = r.standard_normal([Cout]).astype(np.float32)
  ref = O.conv2d_same(torch.tensor(x, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                      torch.tensor(b, dtype=torch.float64), stride, relu=True)
  y = ops.conv3x3(torch.tensor(x, device=dev), torch.tensor(w, device=dev), torch.tensor(b, device=dev), stride, True)
  torch.cuda.synchronize()
  _close(y, ref, 2e-5, 2e-5, 'conv fwd')


# LDS-staged input gradient of the middle layers (conv_dgrad_lds.hip): both tile variants, several ci blocks, odd frame
# counts for the two-frame tiles, more items than one round of blocks
DGRAD_LDS_CASES = [
    (3, 32, 64, 64, 128, 2),    # 1x16 groups: Ho = 16, Wo = 32 -> 2 x 2 tiles per frame
    (2, 16, 32, 128, 64, 2),    # two ci blocks, one tile per frame
    (5, 16, 16, 192, 256, 2),   # conv6 type: two-frame tiles, odd frame count, three ci blocks
    (4, 16, 16, 64, 32, 2),     # two-frame tiles, two co chunks only
    (40, 32, 32, 64, 128, 2),   # 80 tiles x ... enough items for several per block
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [c for c in CONV_CASES if c[3] % 16 == 0] + DGRAD_LDS_CASES +
                         [(3, 128, 192, 32, 48, 2)])
def test_conv3x3_dgrad(dev, N, H, W, Cin, Cout, stride):
  from geeco_amd import ops
  r = np.random.default_rng(8)
  Ho, Wo = -(-H // stride), -(-W // stride)
  dz = r.standard_normal([N, Ho, Wo, Cout]).astype(np.float32)
  w = (r.standard_normal([3, 3, Cin, Cout]) / np.sqrt(9 * Cout)).astype(np.float32)
  ymask = r.standard_normal([N, H, W, Cin]).astype(np.float32)
  x = torch.zeros(N, H, W, Cin, dtype=torch.float64, requires_grad=True)
  y = O.conv2d_same(x, torch.tensor(w, dtype=torch.float64), torch.zeros(Cout, dtype=torch.float64), stride, relu=False)
  y.backward(torch.tensor(dz, dtype=torch.float64))
  ref = x.grad * (torch.tensor(ymask) > 0)
  dx = ops.conv3x3_dgrad(torch.tensor(dz, device=dev), torch.tensor(w, device=dev), torch.tensor(ymask, device=dev),
                         (H, W), stride)
  torch.cuda.synchronize()
  _close(dx, ref, 2e-5, 2e-5, 'conv dgrad')


# LDS-staged filter gradient of the middle layers (conv_wgrad_halo.hip): every tile variant, ragged tile rows / columns,
# several ci / co blocks, more slices than tiles
WGRAD_LDS_CASES = [
    (2, 64, 64, 48, 64, 2),     # conv3 type (CIB 48, 2x16 tiles), exact tiles
    (3, 36, 72, 48, 64, 2),     # conv3 type, Ho = 18, Wo = 36: ragged last tile column
    (5, 20, 24, 64, 64, 2),     # CIB 64, 4x8 tiles: Ho = 10, Wo = 12 ragged in both directions
    (4, 16, 16, 192, 256, 2),   # conv6 type: 3 ci blocks x 4 co blocks, 4x8 tiles
    (1, 4, 32, 64, 128, 2),     # 2x16 tiles with fewer tiles (1) than slices
    (2, 34, 66, 128, 64, 2),    # Ho = 17 (odd): last tile row half empty; Wo = 33
]


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', CONV_CASES + WGRAD_LDS_CASES +
                         [(2, 64, 64, 4, 32, 1), (3, 32, 32, 32, 48, 2), (5, 128, 128, 32, 48, 2)])
def test_conv3x3_wgrad(dev, N, H, W, Cin, Cout, stride):
  from geeco_amd import ops
  r = np.random.default_rng(9)
  Ho, Wo = -(-H // stride), -(-W // stride)
  x = r.standard_normal([N, H, W, Cin]).astype(np.float32)
  dz = r.standard_normal([N, Ho, Wo, Cout]).astype(np.float32)
  wt = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
  bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
  y = O.conv2d_same(torch.tensor(x, dtype=torch.float64), wt, bt, stride, relu=False)
  y.backward(torch.tensor(dz, dtype=torch.float64))
  dw, db = ops.conv3x3_wgrad(torch.tensor(x, device=dev), torch.tensor(dz, device=dev), stride)
  torch.cuda.synchronize()
  scale = np.sqrt(N * Ho * Wo)
  _close(dw, wt.grad, 2e-5, 2e-5 * scale, 'conv wgrad')
  _close(db, bt.grad, 2e-5, 2e-5 * scale, 'conv bias grad')


def test_same_padding_probe(dev):
  """SURVEY 8c KAT: with TF SAME / stride 2 on an even input a delta at (1,1) lights only output (0,0)."""
  from geeco_amd import ops
  x = torch.zeros(1, 8, 8, 4, device=dev)
  x[0, 1, 1, 0] = 1.0
  w = torch.zeros(3, 3, 4, 16, device=dev)
  w[:, :, 0, 0] = 1.0
  y = ops.conv3x3(x, w, torch.zeros(16, device=dev), 2, relu=False)[0, :, :, 0].cpu().numpy()
  expect = np.zeros([4, 4], np.float32)
  expect[0, 0] = 1.0
  np.testing.assert_array_equal(y, expect)


@pytest.mark.parametrize('N,K,H,W,C,Cpad', [(3, 16, 16, 24, 3, 4), (2, 4, 8, 8, 3, 4), (2, 5, 6, 10, 4, 4),
                                            (2, 2, 7, 9, 3, 4), (1, 32, 16, 16, 4, 4)])
def test_dynimg(dev, N, K, H, W, C, Cpad):
  from geeco_amd import ops
  r = np.random.default_rng(3)
  fr = r.random([N, K, H, W, C], dtype=np.float32)
  ref = O.dynimg(torch.tensor(fr))            # fp32 restatement
  ref64 = O.dynimg(torch.tensor(fr, dtype=torch.float64))
  out = ops.dynimg(torch.tensor(fr, device=dev), Cpad)
  torch.cuda.synchronize()
  assert out.shape == (N, H, W, Cpad)
  _close(out[..., :C], ref64, 0, 5e-6, 'dynimg vs fp64')   # values in [0,1]; alpha up to ~50 x fp32 eps
  _close(out[..., :C], ref, 0, 1e-5, 'dynimg vs fp32')
  if Cpad > C:
    assert float(out[..., C:].abs().max()) == 0.0


@pytest.mark.parametrize('N,K,H,W', [(2, 5, 40, 52), (3, 32, 136, 136), (2, 2, 64, 64)])
def test_dynimg_rgbd_unpacked(dev, N, K, H, W):
  """RGB-D dynamic image with rgb and depth in separate tensors (concat in registers) == the oracle on the packed frames;
  also the two-frame form used for the diff image."""
  from geeco_amd import ops
  r = np.random.default_rng(12)
  rgb = r.random([N, K, H, W, 3], dtype=np.float32)
  dep = (0.5 + 2.5 * r.random([N, K, H, W, 1], dtype=np.float32))
  ws = ops.dynimg_ws(N, H * W * 4, dev)
  out = torch.full((N, H, W, 4), float('nan'), device=dev)
  rd, dd = torch.tensor(rgb, device=dev), torch.tensor(dep, device=dev)
  ops.dynimg_rgbd_into(out, rd, dd, K, N, H * W, ws, K * H * W * 3, H * W * 3, K * H * W, H * W)
  torch.cuda.synchronize()
  ref = O.dynimg(torch.tensor(np.concatenate([rgb, dep], -1), dtype=torch.float64))
  _close(out, ref, 0, 5e-6, 'rgbd buffer image')
  tr, td = r.random([N, H, W, 3], dtype=np.float32), (0.5 + 2.5 * r.random([N, H, W, 1], dtype=np.float32))
  ops.dynimg_rgbd_into(out, rd[:, K - 1], dd[:, K - 1], 2, N, H * W, ws, K * H * W * 3, 0, K * H * W, 0,
                       rgb2=torch.tensor(tr, device=dev), depth2=torch.tensor(td, device=dev))
  torch.cuda.synchronize()
  cur = np.concatenate([rgb[:, K - 1], dep[:, K - 1]], -1)
  ref2 = O.dynimg(torch.tensor(np.stack([cur, np.concatenate([tr, td], -1)], 1), dtype=torch.float64))
  _close(out, ref2, 0, 5e-6, 'rgbd diff image')


def test_dynimg_known_answers(dev):
  """Constant sequence -> D == 0 -> normalised image == 0 (sum alpha = 0); dyndiff(cur == tgt) == 0."""
  from geeco_amd import ops
  fr = torch.full((2, 4, 8, 8, 3), 0.37, device=dev)
  out = ops.dynimg(fr, 4)
  assert float(out.abs().max()) <= 1e-6 / 1e-6 * 1.0   # |D| <= fp32 rounding of sum(alpha)*0.37, / 1e-6 range
  # alpha tables (graph.py:17-28)
  np.testing.assert_allclose(ops.dynimg_alpha(2), [-0.5, 0.5], atol=1e-7)
  np.testing.assert_allclose(ops.dynimg_alpha(4), [-2.416667, 0.583333, 1.083333, 0.75], atol=2e-6)
  np.testing.assert_allclose(ops.dynimg_alpha(16), O.dynimg_alpha(16), rtol=0, atol=0)


@pytest.mark.parametrize('M,N,K,ta,tb', [(32, 512, 3100, False, False), (3100, 512, 32, True, False),
                                         (32, 3100, 512, False, True), (5, 7, 9, False, False),
                                         (70, 130, 260, True, True)])
def test_gemm(dev, M, N, K, ta, tb):
  from geeco_amd import ops
  r = np.random.default_rng(5)
  A = r.standard_normal([K, M] if ta else [M, K]).astype(np.float32)
  B = r.standard_normal([N, K] if tb else [K, N]).astype(np.float32)
  ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
  C = ops.gemm(torch.tensor(A, device=dev), torch.tensor(B, device=dev), ta, tb)
  torch.cuda.synchronize()
  _close(C, ref, 1e-5, 2e-5 * np.sqrt(K), 'gemm')


def test_lstm_cell_tf_published_vector(dev):
  """The vector TF 1.15's own rnn_cell_test.py::testBasicLSTMCell publishes (tests/test_oracle_kat.py has the literals and what
  they do and do not pin), through the HIP entry points of the cell: geeco_gemm_f32 ([x | h] W) + geeco_lstm_gates_fwd with a
  NON-zero incoming state, twice (two stacked cells of 2 units)."""
  from geeco_amd import ops
  from test_oracle_kat import TF_BASIC_LSTM_OUTPUT, TF_BASIC_LSTM_STATE, TF_BASIC_LSTM_TOL, tf_basic_lstm_case
  x, state, kernel, bias = (t.to(dev) for t in tf_basic_lstm_case(torch.float32))
  H = 2

  def cell(inp, c_prev, h_prev):
    z = ops.gemm(torch.cat([inp, h_prev], 1).contiguous(), kernel)
    c, h, gates = (torch.full((1, H), float('nan'), device=dev), torch.full((1, H), float('nan'), device=dev),
                   torch.full((1, 4 * H), float('nan'), device=dev))
    ops.lstm_gates_fwd_into(c, h, gates, z, bias, c_prev.contiguous(), 1, H)
    return c, h
  c1, h1 = cell(x, state[:, 0:2], state[:, 2:4])
  c2, h2 = cell(h1, state[:, 4:6], state[:, 6:8])
  torch.cuda.synchronize()
  new_state = torch.cat([c1, h1, c2, h2], 1).cpu().numpy()
  np.testing.assert_allclose(h2.cpu().numpy(), TF_BASIC_LSTM_OUTPUT, atol=TF_BASIC_LSTM_TOL, rtol=0)
  np.testing.assert_allclose(new_state, TF_BASIC_LSTM_STATE, atol=TF_BASIC_LSTM_TOL, rtol=0)
  np.testing.assert_allclose(new_state, TF_BASIC_LSTM_STATE, atol=2e-6, rtol=0)       # fp32 against the eight-digit literals


def test_conv3x3_same_padding_tf_published_vectors(dev):
  """TF 1.15's own conv_ops_test.py vectors for padding='SAME' with strides (tests/test_oracle_kat.py has the literals) through
  geeco_conv3x3_fwd.  The entry point is 3 x 3 only, so TF's 2 x 2 filters sit inside a 3 x 3 kernel whose other taps are zero, at the
  position where the 3 x 3 SAME window covers the pixels TF's 2 x 2 SAME window covers:
    * stride 3 on 4 x 4 (testConv2DKernelSmallerThanStrideSame): 3 x 3 pads (1, 1), its window for output o starts at 3 o - 1; TF's
      2 x 2 pads (0, 1), window starts at 3 o -> the filter goes to kernel rows / columns 1..2;
    * stride 2 on 2 x 3 (testConv2D2x2FilterStride2Same): rows -- 3 x 3 pads (0, 1), 2 x 2 pads (0, 0): kernel rows 0..1; columns --
      3 x 3 pads (1, 1), 2 x 2 pads (0, 1): kernel columns 1..2.
  Channels are zero-padded to the entry point's multiples (Cin 4, Cout 16).  Exact: small integers in fp32."""
  from geeco_amd import ops
  from test_oracle_kat import TF_CONV_SAME_CASES, tf_running_numbers
  for (in_sizes, f_sizes, stride, expected), (r0, c0) in ((TF_CONV_SAME_CASES[3], (1, 1)), (TF_CONV_SAME_CASES[0], (0, 1))):
    _, H, W, C = in_sizes
    kh, kw, _, Co = f_sizes
    x = torch.zeros(1, H, W, 4)
    x[..., :C] = tf_running_numbers(in_sizes, torch.float32)
    w = torch.zeros(3, 3, 4, 16)
    w[r0:r0 + kh, c0:c0 + kw, :C, :Co] = tf_running_numbers(f_sizes, torch.float32)
    Ho, Wo = -(-H // stride), -(-W // stride)
    y = torch.full((1, Ho, Wo, 16), float('nan'), device=dev)
    xd, wd, bd = x.to(dev), w.to(dev), torch.zeros(16, device=dev)
    ws = ops._ws(ops.conv3x3_fwd_ws_bytes(1, 1, H, W, 4, 16, stride), dev)
    ops.conv3x3_fwd_into(y, xd, wd, bd, 1, 0, 0, 0, 0, 1, H, W, 4, 16, stride, relu=False, ws=ws)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y[..., :Co].reshape(-1).cpu().numpy(), expected)
    assert not y[..., Co:].any()


def test_adam_follows_the_protocol_of_tfs_own_test(dev):
  """geeco_adam_prepare + geeco_adam_tf through TF 1.15's adam_test.py::testBasic protocol (var [1, 2] / [3, 4], constant gradients 0.1 /
  0.01, three steps, lr 0.001) against the numpy reference that test carries (tests/test_oracle_kat.py: tf_adam_update_numpy): epsilon
  beside sqrt(v) ("epsilon hat"), lr_t from the step counter kept in device memory."""
  from geeco_amd import ops
  from test_oracle_kat import TF_ADAM_GRADS, TF_ADAM_VARS, tf_adam_update_numpy
  p = torch.tensor(TF_ADAM_VARS[0] + TF_ADAM_VARS[1], device=dev)
  g = torch.tensor(TF_ADAM_GRADS[0] + TF_ADAM_GRADS[1], device=dev)
  m, v = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
  step = torch.zeros(1, dtype=torch.int64, device=dev)
  scal = torch.zeros(2, device=dev)
  ref, rm, rv = np.array(TF_ADAM_VARS[0] + TF_ADAM_VARS[1]), np.zeros(4), np.zeros(4)
  gn = np.array(TF_ADAM_GRADS[0] + TF_ADAM_GRADS[1])
  for t in (1, 2, 3):
    ops.adam_prepare(step, 0.001, scal)
    ops.adam_tf(p, g, m, v, 4, scal)
    torch.cuda.synchronize()
    ref, rm, rv = tf_adam_update_numpy(ref, gn, t, rm, rv)
    assert int(step.item()) == t
    np.testing.assert_allclose(p.cpu().numpy(), ref, rtol=3e-7, atol=0)      # (TF's test itself asserts 1e-6 on float32 variables)
    # the slots carry float32(0.9) / float32(0.999), as TF's float32 kernel does: 1 - float32(0.999) differs from 0.001 by 1.3e-5 relative
    np.testing.assert_allclose(m.cpu().numpy(), rm, rtol=1e-6)
    np.testing.assert_allclose(v.cpu().numpy(), rv, rtol=2e-5)


def test_losses_tf_published_values(dev):
  """tf.losses.mean_squared_error / softmax_cross_entropy values TF 1.15's own losses_test.py publishes (49.5; 10.0 to three places)
  through the HIP decoder tail (geeco_heads_loss_fwd_bwd): fc1 = identity, head kernels = the first rows of the identity, zero biases,
  so that the heads' predictions ARE the first entries of a non-negative h.  Pins "mean over all elements" (MSE) and "mean over the
  batch" of one-hot cross entropy (labels = rint(cmd[:, 3]) + 1, estimator.py:208-210) in the HIP path itself."""
  from geeco_amd import ops
  from test_oracle_kat import (TF_MSE_LABELS, TF_MSE_LOSS, TF_MSE_PREDICTIONS, TF_XENT_LOGITS, TF_XENT_WRONG_CLASSES, TF_XENT_WRONG_LOSS)
  H = F = 128
  for kind, preds_in, target, expected, tol in (
      (0, TF_MSE_PREDICTIONS, torch.tensor(TF_MSE_LABELS), TF_MSE_LOSS, 1e-5),
      (1, TF_XENT_LOGITS, torch.tensor(TF_XENT_WRONG_CLASSES, dtype=torch.float32).reshape(-1, 1) - 1.0, TF_XENT_WRONG_LOSS, 5e-4)):
    N = len(preds_in)
    h = torch.zeros(N, H)
    h[:, :3] = torch.tensor(preds_in)
    w1, b1 = torch.eye(H, F), torch.zeros(F)
    hw, hb = torch.zeros(F, 3), torch.zeros(3)
    hw[:3, :3] = torch.eye(3)
    d = lambda t: t.to(dev).contiguous()
    preds, losses = torch.empty(N, 3, device=dev), torch.zeros(8, device=dev)
    ws = torch.empty(ops.heads_ws_bytes(N, H, F) // 4 + 4, device=dev)
    ops.heads_loss_into(preds, losses, d(h), d(w1), d(b1), [d(hw)], [d(hb)], [3], [kind], [1.0], [d(target)], [target.shape[1]], 1.0, N, H,
                        F, ws)
    torch.cuda.synchronize()
    np.testing.assert_allclose(preds.cpu().numpy(), preds_in, rtol=0, atol=1e-6)
    assert abs(float(losses[1]) - expected) <= tol and abs(float(losses[0]) - expected) <= tol, (kind, losses[:2].tolist())
    if kind == 1:
      assert abs(float(losses[1]) - 10.0000908) < 2e-5


# fc1 + heads + losses (+ gradients): one workgroup per sample + a few blocks for the sums over the batch (H <= 128, Hfc 64 / 128);
# other widths run the single-workgroup kernel; both against the oracle's decoder tail differentiated by autograd in fp64.
@pytest.mark.parametrize('N,mode,H,F', [(1, 'cartesian', 128, 128), (7, 'cartesian', 128, 128), (32, 'cartesian', 128, 128),
                                        (33, 'cartesian', 128, 128), (64, 'cartesian', 128, 128), (32, 'velocity', 128, 128),
                                        (40, 'velocity', 128, 128), (9, 'cartesian', 100, 64), (6, 'cartesian', 64, 96),
                                        (5, 'velocity', 160, 128)])
def test_heads_loss(dev, N, mode, H, F):
  from geeco_amd import ops
  names_expected = ['heads_sample_kernel<false>', 'heads_finish_kernel'] if (H <= 128 and F in (64, 128)) else ['heads_loss_kernel']
  g = torch.Generator().manual_seed(100 + N)
  if mode == 'cartesian':   # (size, kind, weight): graph.py:233-239; lambda_aux = 0.5
    heads = [(3, 0, 1.0), (3, 1, 1.0), (3, 0, 0.5), (3, 0, 0.5)]
  else:                     # velocity heads graph.py:240-249, all MSE, unit weights
    heads = [(7, 0, 1.0), (3, 0, 1.0), (2, 0, 1.0), (3, 0, 1.0), (3, 0, 1.0)]
  h = torch.randn(N, H, generator=g)
  w1 = torch.randn(H, F, generator=g) * 0.1
  b1 = torch.randn(F, generator=g) * 0.1
  hw = [torch.randn(F, sz, generator=g) * 0.1 for sz, _, _ in heads]
  hb = [torch.randn(sz, generator=g) * 0.1 for sz, _, _ in heads]
  tg = [torch.randn(N, sz, generator=g) if kind == 0 else torch.randint(-1, 2, (N, 1), generator=g).float()
        for sz, kind, _ in heads]
  # oracle (fp64 autograd)
  leaves = [t.double().requires_grad_(True) for t in [h, w1, b1] + hw + hb]
  hd, w1d, b1d = leaves[:3]
  hwd, hbd = leaves[3:3 + len(heads)], leaves[3 + len(heads):]
  a1 = torch.relu(hd @ w1d + b1d)
  total, parts, preds_ref = 0.0, [], []
  for (sz, kind, wt), w_, b_, t_ in zip(heads, hwd, hbd, tg):
    pr = a1 @ w_ + b_
    preds_ref.append(pr)
    l = O.mse(pr, t_.double()) if kind == 0 else O.softmax_xent(pr, torch.round(t_[:, 0]).long() + 1, sz)
    parts.append(l)
    total = total + wt * l
  total.backward()
  # HIP
  d = lambda t: t.to(dev).contiguous()
  OT = sum(sz for sz, _, _ in heads)
  preds = torch.empty(N, OT, device=dev)
  losses = torch.zeros(8, device=dev)
  ws = torch.empty(ops.heads_ws_bytes(N, H, F) // 4 + 4, device=dev)
  dh, dw1, db1 = torch.empty(N, H, device=dev), torch.empty(H, F, device=dev), torch.empty(F, device=dev)
  dhw = [torch.empty(F, sz, device=dev) for sz, _, _ in heads]
  dhb = [torch.empty(sz, device=dev) for sz, _, _ in heads]
  tgd = [d(t) for t in tg]
  dev_in = (d(h), d(w1), d(b1), [d(t) for t in hw], [d(t) for t in hb])
  names = ops.kernel_trace(lambda: ops.heads_loss_into(
      preds, losses, *dev_in, [sz for sz, _, _ in heads], [k for _, k, _ in heads], [w for _, _, w in heads], tgd,
      [t.shape[1] for t in tg], 1.0, N, H, F, ws, dh=dh, d_fc1_w=dw1, d_fc1_b=db1, d_heads_w=dhw, d_heads_b=dhb))
  assert names == names_expected, names
  torch.cuda.synchronize()
  _close(preds, torch.cat(preds_ref, 1), 1e-5, 1e-5, 'preds')
  _close(losses[0], total, 1e-5, 1e-6, 'loss')
  _close(losses[1:1 + len(heads)], torch.stack(parts), 1e-5, 1e-6, 'per-head losses')
  scale = lambda t: float(t.abs().max()) * 2e-5 + 1e-8
  _close(dh, hd.grad, 0, scale(hd.grad), 'dh')
  _close(dw1, w1d.grad, 0, scale(w1d.grad), 'd fc1/kernel')
  _close(db1, b1d.grad, 0, scale(b1d.grad), 'd fc1/bias')
  for i in range(len(heads)):
    _close(dhw[i], hwd[i].grad, 0, scale(hwd[i].grad), 'd head kernel %d' % i)
    _close(dhb[i], hbd[i].grad, 0, scale(hbd[i].grad), 'd head bias %d' % i)


def test_derive_conv_weights(dev):
  """One-launch derive (per-tap transposes + padded conv1 kernel) vs torch permute / pad, bit-exact copies."""
  from geeco_amd import ops
  G = 3
  shapes = [(32, 48), (48, 64), (64, 128), (20, 36)]     # (Cin, Cout), the last one with ragged 32x32 tiles
  g = torch.Generator().manual_seed(5)
  arena = torch.randn(G, 200000, generator=g).to(dev)      # kernels of one encoder at a common group stride
  gs_w, off, ws = arena.shape[1], 0, []
  for ci, co in shapes:
    ws.append(arena[:, off:off + 9 * ci * co])
    off += 9 * ci * co + 12
  w1 = arena[:, off:off + 9 * 3 * 32]
  wts = [torch.empty(G, 9, co, ci, device=dev) for ci, co in shapes]
  w1p = torch.full((G, 9, 4, 32), 7.0, device=dev)
  ops.derive_conv_weights([w[0] for w in ws], wts, [s[0] for s in shapes], [s[1] for s in shapes], G, gs_w,
                          pad_src=w1[0], pad_dst=w1p, pad_cin=3, pad_cin_padded=4, pad_cout=32)
  torch.cuda.synchronize()
  for (ci, co), w, wt in zip(shapes, ws, wts):
    assert torch.equal(wt, w.reshape(G, 9, ci, co).permute(0, 1, 3, 2).contiguous())
  ref = torch.zeros(G, 9, 4, 32, device=dev)
  ref[:, :, :3] = w1.reshape(G, 9, 3, 32)
  assert torch.equal(w1p, ref)


# Grouped launches (G encoders at a stride, as the training step issues them) of the persistent LDS-halo
# kernels, sized so that some block's tile range CROSSES an encoder boundary: that block has to swap the
# kernel it keeps resident in LDS mid-flight.  (The single-group cases above never take that path.)
@pytest.mark.parametrize('Cin,Cout,Nf,Nd', [(32, 48, 7, 5), (48, 64, 7, 5)])
def test_conv3x3_grouped_halo_kernels(dev, Cin, Cout, Nf, Nd):
  from geeco_amd import ops
  G, H, W, stride = 3, 72, 136, 2
  Ho, Wo = H // 2, W // 2
  r = np.random.default_rng(21)
  gs_w = 9 * Cin * Cout + 16           # kernels of the G encoders at a common (padded) stride, like the arena
  warena = torch.zeros(G, gs_w, device=dev)
  w = (r.standard_normal([G, 3, 3, Cin, Cout]) / np.sqrt(9 * Cin)).astype(np.float32)
  warena[:, :9 * Cin * Cout] = torch.tensor(w.reshape(G, -1), device=dev)
  barena = torch.tensor(r.standard_normal([G, Cout + 16]).astype(np.float32), device=dev)
  # forward: tiles per encoder (7 x 9 x 5 = 315) are not a multiple of the tiles per block (4)
  x = r.standard_normal([G, Nf, H, W, Cin]).astype(np.float32)
  xd = torch.tensor(x, device=dev)
  y = torch.empty(G, Nf, Ho, Wo, Cout, device=dev)
  ws = torch.empty(ops.conv3x3_fwd_ws_bytes(G, Nf, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
  ops.conv3x3_fwd_into(y, xd, warena, barena, G, xd[0].numel(), gs_w, Cout + 16, y[0].numel(), Nf, H, W, Cin, Cout,
                       stride, relu=True, ws=ws)
  torch.cuda.synchronize()
  for g in range(G):
    ref = O.conv2d_same(torch.tensor(x[g], dtype=torch.float64), torch.tensor(w[g], dtype=torch.float64),
                        barena[g, :Cout].double().cpu(), stride, relu=True)
    _close(y[g], ref, 2e-5, 2e-5, 'grouped fwd, encoder %d' % g)
  # input gradient: 5 x 9 x 3 = 135 tiles per encoder, 2 tiles per block
  dz = r.standard_normal([G, Nd, Ho, Wo, Cout]).astype(np.float32)
  mask = r.standard_normal([G, Nd, H, W, Cin]).astype(np.float32)
  dzd, md = torch.tensor(dz, device=dev), torch.tensor(mask, device=dev)
  wt = torch.empty(G, 3, 3, Cout, Cin, device=dev)
  ops.transpose_hwio_into(wt, warena, G, gs_w, wt[0].numel(), Cin, Cout)
  dx = torch.empty(G, Nd, H, W, Cin, device=dev)
  dws = torch.empty(ops.conv3x3_dgrad_ws_bytes(G, Nd, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
  ops.conv3x3_dgrad_into(dx, dzd, wt, md, G, dzd[0].numel(), wt[0].numel(), dx[0].numel(), Nd, H, W, Cin, Cout, stride,
                         ws=dws, w=warena, gs_w=gs_w)
  torch.cuda.synchronize()
  for g in range(G):
    xg = torch.zeros(Nd, H, W, Cin, dtype=torch.float64, requires_grad=True)
    yy = O.conv2d_same(xg, torch.tensor(w[g], dtype=torch.float64), torch.zeros(Cout, dtype=torch.float64), stride, relu=False)
    yy.backward(torch.tensor(dz[g], dtype=torch.float64))
    _close(dx[g], xg.grad * (torch.tensor(mask[g]) > 0), 2e-5, 2e-5, 'grouped dgrad, encoder %d' % g)


# Fused encoder bottom backward (conv2 dgrad + conv1 wgrad, dz1 on chip) vs the oracle's conv1 -> ReLU -> conv2
# differentiated by autograd in fp64; grouped, ragged tiles (H = 24 -> 3 tile rows, W = 72 -> 2 tile columns, the
# second one partial) and a block whose tile range crosses an image boundary.
# (3, 8, 64, 256): 256 tiles per encoder = 85 blocks x 3 + 1: the launch takes the remainder-block form (one more block
# walks the tile each encoder leaves over: three segments, three slabs)
@pytest.mark.parametrize('G,N,H,W,C', [(1, 2, 16, 64, 3), (3, 3, 24, 72, 3), (2, 40, 32, 64, 3), (2, 3, 24, 72, 4),
                                       (3, 8, 64, 256, 3)])
def test_conv2_dgrad_conv1_wgrad_fused(dev, G, N, H, W, C):
  from geeco_amd import ops
  r = np.random.default_rng(31)
  x3 = r.standard_normal([G, N, H, W, C]).astype(np.float32)      # C real input channels (3: RGB, 4: RGB-D)
  x4 = np.concatenate([x3, np.zeros([G, N, H, W, 4 - C], np.float32)], -1)
  w1 = (r.standard_normal([G, 3, 3, C, 32]) / np.sqrt(9 * C)).astype(np.float32)
  b1 = (0.1 * r.standard_normal([G, 32])).astype(np.float32)
  w2 = (r.standard_normal([G, 3, 3, 32, 48]) / np.sqrt(288)).astype(np.float32)
  dz2 = r.standard_normal([G, N, H // 2, W // 2, 48]).astype(np.float32)
  dw_ref, db_ref, y1s, dz1_ref = [], [], [], []
  for g in range(G):
    w1t = torch.tensor(w1[g], dtype=torch.float64, requires_grad=True)
    b1t = torch.tensor(b1[g], dtype=torch.float64, requires_grad=True)
    y1 = O.conv2d_same(torch.tensor(x3[g], dtype=torch.float64), w1t, b1t, 1, relu=True)
    y1.retain_grad()
    z2 = O.conv2d_same(y1, torch.tensor(w2[g], dtype=torch.float64), torch.zeros(48, dtype=torch.float64), 2, relu=False)
    z2.backward(torch.tensor(dz2[g], dtype=torch.float64))
    dw_ref.append(w1t.grad); db_ref.append(b1t.grad); y1s.append(y1.detach().float().numpy())
    dz1_ref.append((y1.grad * (y1.detach() > 0)))
  y1d = torch.tensor(np.stack(y1s), device=dev)
  dw1p = torch.full((G, 9, C, 32), 3.0, device=dev)     # the variable's own layout [3][3][C][32]
  db1 = torch.full((G, 32), 3.0, device=dev)
  dz1 = torch.empty(G, N, H, W, 32, device=dev)
  ws = torch.empty(ops.conv2_dgrad_conv1_wgrad_ws_bytes(G) // 4 + 4, device=dev)
  xd, w2d, dz2d = torch.tensor(x4, device=dev), torch.tensor(w2, device=dev), torch.tensor(dz2, device=dev)
  for with_dz1 in (True, False):
    ops.conv2_dgrad_conv1_wgrad_into(dw1p, db1, dz2d, w2d, y1d, xd, G, dz2d[0].numel(), w2d[0].numel(), y1d[0].numel(),
                                     xd[0].numel(), dw1p[0].numel(), 32, N, H, W, ws, dz1=dz1 if with_dz1 else None,
                                     real_channels=C)
    torch.cuda.synchronize()
    scale = np.sqrt(N * H * W)
    for g in range(G):
      _close(dw1p[g], dw_ref[g].reshape(9, C, 32), 2e-5, 2e-5 * scale, 'fused dw1, encoder %d' % g)
      _close(db1[g], db_ref[g], 2e-5, 2e-5 * scale, 'fused db1, encoder %d' % g)
      if with_dz1:
        _close(dz1[g], dz1_ref[g], 2e-5, 2e-5, 'fused dz1, encoder %d' % g)


@pytest.mark.parametrize('G,N,H,W,C', [(1, 2, 16, 64, 3), (3, 3, 24, 72, 3), (2, 3, 24, 72, 4)])
def test_conv1_relu_bits_and_fused_bits(dev, G, N, H, W, C):
  """conv1's forward with the sign-bit side output (bit (c & 3) * 8 + (c >> 2) of a pixel's word <-> y1[c] > 0) and the
  fused bottom backward fed by those words: y1 equals the plain forward, the words equal the packed signs of y1, and
  dw1 / db1 equal (bitwise) the gradients of the kernel that reads y1 itself; ragged tiles, row pitch > W."""
  from geeco_amd import ops
  r = np.random.default_rng(37)
  x3 = r.standard_normal([G, N, H, W, C]).astype(np.float32)
  x4 = np.concatenate([x3, np.zeros([G, N, H, W, 4 - C], np.float32)], -1)
  w1 = np.zeros([G, 3, 3, 4, 32], np.float32)
  w1[:, :, :, :C] = (r.standard_normal([G, 3, 3, C, 32]) / np.sqrt(9 * C)).astype(np.float32)
  b1 = (0.1 * r.standard_normal([G, 32])).astype(np.float32)
  w2 = (r.standard_normal([G, 3, 3, 32, 48]) / np.sqrt(288)).astype(np.float32)
  dz2 = r.standard_normal([G, N, H // 2, W // 2, 48]).astype(np.float32)
  xd, w1d, b1d = torch.tensor(x4, device=dev), torch.tensor(w1, device=dev), torch.tensor(b1, device=dev)
  w2d, dz2d = torch.tensor(w2, device=dev), torch.tensor(dz2, device=dev)
  Wp, Hp = ops.relu_bits_pitch(W), ops.relu_bits_rows(H)
  assert Wp % 64 == 0 and W <= Wp < W + 64 and Hp % 8 == 0 and H <= Hp < H + 8
  y_plain = torch.empty(G, N, H, W, 32, device=dev)
  ws = torch.empty(ops.conv3x3_fwd_ws_bytes(G, N, H, W, 4, 32, 1) // 4 + 4, device=dev)
  ops.conv3x3_fwd_into(y_plain, xd, w1d, b1d, G, xd[0].numel(), w1d[0].numel(), 32, y_plain[0].numel(), N, H, W, 4, 32, 1,
                       relu=True, ws=ws)
  y = torch.empty_like(y_plain)
  bits = torch.zeros(G, N, Hp, Wp, dtype=torch.int32, device=dev)     # zero-filled once: the padding stays zero
  ops.conv1_fwd_relu_bits_into(y, bits, xd, w1d, b1d, G, xd[0].numel(), w1d[0].numel(), 32, y[0].numel(), bits[0].numel(),
                               N, H, W)
  torch.cuda.synchronize()
  assert torch.equal(y, y_plain)
  c = np.arange(32)
  weights = (1 << ((c & 3) * 8 + (c >> 2))).astype(np.int64)
  want = ((y.cpu().numpy() > 0).astype(np.int64) * weights).sum(-1).astype(np.uint32)
  got = bits.cpu().numpy().view(np.uint32)
  assert np.array_equal(got[:, :, :H, :W], want)
  assert (got[:, :, :, W:] == 0).all() and (got[:, :, H:] == 0).all()      # the padding is never written
  wsf = torch.empty(ops.conv2_dgrad_conv1_wgrad_ws_bytes(G) // 4 + 4, device=dev)
  outs = []
  for use_bits in (False, True):
    dw1 = torch.full((G, 9, C, 32), float('nan'), device=dev)
    db1 = torch.full((G, 32), float('nan'), device=dev)
    if use_bits:
      names = ops.kernel_trace(lambda: ops.conv2_dgrad_conv1_wgrad_bits_into(
          dw1, db1, dz2d, w2d, bits, xd, G, dz2d[0].numel(), w2d[0].numel(), bits[0].numel(), xd[0].numel(), dw1[0].numel(),
          32, N, H, W, wsf, real_channels=C))
      assert names[0] == 'conv2_dgrad_conv1_wgrad_kernel<%d, true>' % C, names
    else:
      ops.conv2_dgrad_conv1_wgrad_into(dw1, db1, dz2d, w2d, y, xd, G, dz2d[0].numel(), w2d[0].numel(), y[0].numel(),
                                       xd[0].numel(), dw1[0].numel(), 32, N, H, W, wsf, real_channels=C)
    torch.cuda.synchronize()
    outs.append((dw1, db1))
  assert not torch.isnan(outs[1][0]).any()
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('G,N,H,W', [(1, 2, 16, 64), (3, 2, 24, 72)])
def test_conv2_relu_fields_and_conv3_dgrad_fields(dev, G, N, H, W):
  """conv2's forward with the sign-field side output (uint16 per (pixel, quad q): bit 4 i + j <-> channel 16 i + 4 q + j)
  and conv3's input gradient masked by those fields: y2 equals the plain forward, the fields equal the packed signs of
  y2, dx equals (bitwise) the gradient masked by y2 itself; ragged tiles (H x W = conv2's input, y2 is H/2 x W/2)."""
  from geeco_amd import ops
  r = np.random.default_rng(39)
  x = torch.tensor(r.standard_normal([G, N, H, W, 32]).astype(np.float32), device=dev)
  w2 = torch.tensor((r.standard_normal([G, 3, 3, 32, 48]) / 17).astype(np.float32), device=dev)
  b2 = torch.tensor((0.1 * r.standard_normal([G, 48])).astype(np.float32), device=dev)
  H2, W2 = H // 2, W // 2
  y_plain = torch.empty(G, N, H2, W2, 48, device=dev)
  ws = torch.empty(ops.conv3x3_fwd_ws_bytes(G, N, H, W, 32, 48, 2) // 4 + 4, device=dev)
  ops.conv3x3_fwd_into(y_plain, x, w2, b2, G, x[0].numel(), w2[0].numel(), 48, y_plain[0].numel(), N, H, W, 32, 48, 2,
                       relu=True, ws=ws)
  y = torch.empty_like(y_plain)
  ne = ops.relu_fields_elems(N, H2, W2)
  Hp, Wp = (H2 + 7) // 8 * 8, (W2 + 63) // 64 * 64
  assert ne == N * Hp * Wp * 4
  fields = torch.zeros(G, ne, dtype=torch.int16, device=dev)
  names = ops.kernel_trace(lambda: ops.conv2_fwd_relu_fields_into(y, fields, x, w2, b2, G, x[0].numel(), w2[0].numel(), 48,
                                                                  y[0].numel(), ne, N, H, W))
  torch.cuda.synchronize()
  assert names[0].startswith('conv_s2_halo_fwd_ws_kernel'), names
  assert torch.equal(y, y_plain)
  pos = (y.cpu().numpy() > 0).reshape(G, N, H2, W2, 3, 4, 4)          # [.., i, q, j]
  want = np.zeros([G, N, H2, W2, 4], np.uint16)
  for i in range(3):
    for j in range(4):
      want |= (pos[..., i, :, j].astype(np.uint16) << (4 * i + j))
  got = fields.cpu().numpy().view(np.uint16).reshape(G, N, Hp, Wp, 4)
  assert np.array_equal(got[:, :, :H2, :W2], want)
  assert (got[:, :, H2:] == 0).all() and (got[:, :, :, W2:] == 0).all()
  # conv3's input gradient (48 -> 64, stride 2) on the y2 grid
  w3 = torch.tensor((r.standard_normal([G, 3, 3, 48, 64]) / 20).astype(np.float32), device=dev)
  dz3 = torch.tensor(r.standard_normal([G, N, H2 // 2, W2 // 2, 64]).astype(np.float32), device=dev)
  wt = torch.empty(G, 3, 3, 64, 48, device=dev)
  ops.transpose_hwio_into(wt, w3, G, w3[0].numel(), wt[0].numel(), 48, 64)
  dx_ref = torch.full((G, N, H2, W2, 48), float('nan'), device=dev)
  dws = torch.empty(ops.conv3x3_dgrad_ws_bytes(G, N, H2, W2, 48, 64, 2) // 4 + 4, device=dev)
  ops.conv3x3_dgrad_into(dx_ref, dz3, wt, y, G, dz3[0].numel(), wt[0].numel(), dx_ref[0].numel(), N, H2, W2, 48, 64, 2,
                         ws=dws, w=w3, gs_w=w3[0].numel())
  dx = torch.full_like(dx_ref, float('nan'))
  names = ops.kernel_trace(lambda: ops.conv3_dgrad_relu_fields_into(dx, dz3, w3, fields, G, dz3[0].numel(), w3[0].numel(), ne,
                                                                    dx[0].numel(), N, H2, W2))
  torch.cuda.synchronize()
  assert names == ['conv_s2_halo_dgrad_chunked_kernel<48, 64, true>'], names
  assert not torch.isnan(dx).any() and torch.equal(dx, dx_ref)
  # data parallel: the persistent kernel leaves CUs to the collective beside it (fewer blocks, longer tile ranges): same bits
  for k in (16, 100):
    dx2 = torch.full_like(dx_ref, float('nan'))
    ops.conv3_dgrad_relu_fields_into(dx2, dz3, w3, fields, G, dz3[0].numel(), w3[0].numel(), ne, dx2[0].numel(), N, H2, W2, reserved_cus=k)
    assert torch.equal(dx2, dx_ref), k
  with pytest.raises(Exception):
    ops.conv3_dgrad_relu_fields_into(dx, dz3, w3, fields, G, dz3[0].numel(), w3[0].numel(), ne, dx[0].numel(), N, H2, W2, reserved_cus=200)


@pytest.mark.parametrize('N,K,H,W,C', [(2, 4, 16, 24, 3), (3, 16, 136, 136, 3), (2, 3, 16, 24, 4), (1, 1, 8, 8, 3), (2, 2, 40, 36, 4),
                                       (25, 3, 256, 256, 3), (26, 2, 128, 256, 4), (5, 2, 100, 164, 3)])
def test_goal_dynimgs_one_pass(dev, N, K, H, W, C):
  """The goal model's input stage (graph.py:386-401) as the step runs it: ONE launch, one pass over the window: the buffer image
  and the pair image of (current frame, target) stay in registers across their per-sample min / max (the blocks of a sample meet
  at an arrival counter) and are stored once, normalised; the current frame's padded copy comes from the same pass.
  Bitwise equal to the separate launches (buffer image; two-frame image with the target as second frame), and both images
  against the fp64 oracle; K = 1 (alpha = [0]: the buffer image is identically 0), rgb and rgb + depth; both block shapes
  (from 192 blocks on, 1024 threads carry TWO samples each, the first one's stores inside the second one's frame loop: the 25-
  sample case has an odd count (the last block pair holds one sample), the 26-sample one is even; 256 x 4 pixels below), a
  ragged last block (100 x 164); run three times on one control block (every call must leave it zero-filled)."""
  from geeco_amd import ops
  r = np.random.default_rng(57)
  HW = H * W
  rgb = torch.tensor(r.random([N, K, H, W, 3]).astype(np.float32), device=dev)
  tgt = torch.tensor(r.random([N, H, W, 3]).astype(np.float32), device=dev)
  ws = ops.dynimg_ws(N, HW * 4, dev)
  ws2 = ops.goal_dynimgs_ws(N, HW, dev)
  cur, buf, dif = (torch.full((N, H, W, 4), float('nan'), device=dev) for _ in range(3))
  buf_ref, dif_ref, cur_ref = (torch.empty(N, H, W, 4, device=dev) for _ in range(3))
  if C == 3:
    ops.dynimg_into(buf_ref, rgb, K, N, HW, 3, 4, ws, K * HW * 3, HW * 3)
    ops.dynimg_into(dif_ref, rgb[:, K - 1], 2, N, HW, 3, 4, ws, K * HW * 3, 0, frames2=tgt)
    ops.pack_pixels_into(cur_ref, rgb[:, K - 1], K * HW * 3, N, HW, 3, 4)
    names = ops.kernel_trace(lambda: ops.goal_dynimgs_into(cur, buf, dif, rgb, tgt, K, N, HW, ws2, K * HW * 3, HW * 3))
    for _ in range(2):
      ops.goal_dynimgs_into(cur, buf, dif, rgb, tgt, K, N, HW, ws2, K * HW * 3, HW * 3)
    frames64 = rgb.cpu().double()
    tgt64 = tgt.cpu().double()
  else:
    dep = torch.tensor((0.5 + 2.5 * r.random([N, K, H, W, 1])).astype(np.float32), device=dev)
    tdep = torch.tensor((0.5 + 2.5 * r.random([N, H, W, 1])).astype(np.float32), device=dev)
    ops.dynimg_rgbd_into(buf_ref, rgb, dep, K, N, HW, ws, K * HW * 3, HW * 3, K * HW, HW)
    ops.dynimg_rgbd_into(dif_ref, rgb[:, K - 1], dep[:, K - 1], 2, N, HW, ws, K * HW * 3, 0, K * HW, 0, rgb2=tgt, depth2=tdep)
    ops.pack_pixels_into(cur_ref, rgb[:, K - 1], K * HW * 3, N, HW, 3, 4, dep[:, K - 1], K * HW, 1)
    run = lambda: ops.goal_dynimgs_into(cur, buf, dif, rgb, tgt, K, N, HW, ws2, K * HW * 3, HW * 3, depth=dep, tgt_depth=tdep,
                                        dsample_stride=K * HW, dframe_stride=HW)
    names = ops.kernel_trace(run)
    for _ in range(2):
      run()
    frames64 = torch.cat([rgb, dep], -1).cpu().double()
    tgt64 = torch.cat([tgt, tdep], -1).cpu().double()
  torch.cuda.synchronize()
  big = N * -(-(HW // 4) // 2048) >= 192
  assert names == [('dynimg_goal_onepass2_kernel<%s, false, 1024>' if big else 'dynimg_goal_onepass_kernel<%s, false, 256, 1>') % ('true' if C == 4 else 'false')], names
  assert not ws2[:16 * N].view(N, 16)[:, :2].any()      # the arrival / departure counters are zero again
  assert torch.equal(cur, cur_ref)
  assert torch.equal(buf, buf_ref)
  assert torch.equal(dif, dif_ref)
  sub = slice(0, N if HW <= 136 * 136 else 2)          # (the oracle in fp64 on the big cases: two samples)
  _close(buf[sub][..., :C], O.dynimg(frames64[sub]), 0, 5e-6, 'buffer image vs fp64')
  _close(dif[sub][..., :C], O.dynimg(torch.stack([frames64[sub, K - 1], tgt64[sub]], 1)), 0, 5e-6, 'pair image vs fp64')


@pytest.mark.parametrize('N,H,W,u8', [(25, 256, 256, False), (3, 136, 136, False), (26, 256, 256, True)],
                         ids=['two samples per block', 'small blocks', 'uint8 frames'])
def test_goal_dynimgs_expired_wait_is_loud(dev, N, H, W, u8):
  """The one-pass input stage never continues on stale min / max (csrc/dynimg.hip): when a block's wait for the other blocks
  of its sample expires, that sample's images are NaN, the block counts itself into the workspace's sticky error word, and
  geeco_goal_dynimgs_timeouts / model.check_device_errors report it.  The wait cannot be made to expire through the data
  (blocks of a launch start in index order on this hardware, which is exactly why the bound is never reached), so the test
  sets the bound to ZERO polls: every block that is not the last of its sample to arrive reports at once.  With the default
  bound the same calls report nothing and produce finite images; a healthy workspace stays at zero across calls."""
  from geeco_amd import ops
  from geeco_amd._native import load as lib
  r = np.random.default_rng(5)
  K, HW = 3, H * W
  tgt = torch.tensor(r.random([N, H, W, 3]).astype(np.float32), device=dev)
  rgb = torch.tensor(r.random([N, K, H, W, 3]).astype(np.float32), device=dev)
  cur, buf, dif = (torch.zeros(N, H, W, 4, device=dev) for _ in range(3))
  if u8:
    rgb8 = (rgb * 255).to(torch.uint8).contiguous()
    tgt8 = (tgt * 255).to(torch.uint8).contiguous()
    wp = torch.tensor([rgb8.data_ptr() + n * K * HW * 3 for n in range(N)], dtype=torch.int64, device=dev)
    tp = torch.tensor([tgt8.data_ptr() + n * HW * 3 for n in range(N)], dtype=torch.int64, device=dev)
    run = lambda ws: ops.goal_dynimgs_u8_into(cur, buf, dif, wp, tp, K, N, HW, ws)
  else:
    run = lambda ws: ops.goal_dynimgs_into(cur, buf, dif, rgb, tgt, K, N, HW, ws, K * HW * 3, HW * 3)
  good = ops.goal_dynimgs_ws(N, HW, dev)
  for _ in range(3):
    run(good)
  assert ops.goal_dynimgs_timeouts(good, N) == 0
  ops.check_input_stage(good, N)
  assert torch.isfinite(buf).all() and torch.isfinite(dif).all()
  bad = ops.goal_dynimgs_ws(N, HW, dev)
  old = lib().geeco_goal_dynimgs_set_wait_polls(0)
  try:
    assert old == 1 << 22
    run(bad)
    n_bad = ops.goal_dynimgs_timeouts(bad, N)
  finally:
    assert lib().geeco_goal_dynimgs_set_wait_polls(old) == 0
  # at least one block of some sample arrived before that sample's last block (several blocks per sample in every case here)
  assert n_bad >= 1, n_bad
  per_sample = bad[:16 * N].view(N, 16)[:, 2].cpu().numpy()
  assert per_sample.sum() == n_bad
  for n in range(N):      # a sample with a reporting block has NaN pixels in BOTH images; a sample without one is finite throughout
    has_nan = bool(torch.isnan(buf[n]).any()) and bool(torch.isnan(dif[n]).any())
    assert has_nan == bool(per_sample[n] > 0), (n, per_sample[n])
  assert torch.isfinite(cur).all()      # (the current frame's copy does not depend on the hand-off)
  with pytest.raises(RuntimeError, match='gave up waiting'):
    ops.check_input_stage(bad, N)
  run(bad)                              # sticky: a later call with the default bound does not clear the word
  assert ops.goal_dynimgs_timeouts(bad, N) >= n_bad
  assert ops.goal_dynimgs_timeouts(good, N) == 0


def test_model_reports_an_expired_input_stage_wait(dev):
  """GoalE2EVMC.check_device_errors (what Estimator.train / evaluate, bench.py and endpoints() call where they synchronise):
  silent on a healthy model, raises once the input stage of THIS model's workspace reported.  The NaN images do NOT reach the
  loss -- conv1's ReLU is max(x, 0), which returns 0 for a NaN -- so the error word and this check are what makes the event loud."""
  from geeco_amd import graph, ops
  from geeco_amd._native import load as lib
  from geeco_amd.params import create_e2evmc_config
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=3))
  model = graph.GoalE2EVMC(cfg, 3, dev, training=True)
  model.store.initialize(seed=1)
  feats, labels = O.synthetic_batch(O.make_config(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136,
                                                  batch_size=3), True, 3, seed=8, H=136, W=136)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  model.train_step()
  model.check_device_errors()
  assert np.isfinite(float(model.loss)) and 'dynbuff' in model.endpoints()
  old = lib().geeco_goal_dynimgs_set_wait_polls(0)
  try:
    model.forward(backward_too=False)
    torch.cuda.synchronize()
  finally:
    lib().geeco_goal_dynimgs_set_wait_polls(old)
  assert torch.isnan(model.enc.x_in[1]).any() and torch.isnan(model.enc.x_in[2]).any() and np.isfinite(float(model.loss))
  with pytest.raises(RuntimeError, match='one-pass input stage'):
    model.check_device_errors()
  with pytest.raises(RuntimeError, match='one-pass input stage'):
    model.endpoints()


@pytest.mark.parametrize('N,K,H,W,C', [(2, 4, 16, 24, 3), (3, 16, 136, 136, 3), (2, 3, 16, 24, 4), (1, 1, 8, 8, 3), (4, 2, 40, 36, 4),
                                       (25, 2, 256, 256, 3), (24, 2, 256, 256, 4)])
def test_goal_dynimgs_from_resident_u8_frames(dev, N, K, H, W, C):
  """The input stage fed from the episodes' resident uint8 frames through per-sample window addresses
  (geeco_goal_dynimgs_u8_fwd): bitwise the three images of geeco_gather_windows (/ 255, geeco_gym.py:312) followed by
  geeco_goal_dynimgs_fwd, for overlapping windows of two 'episodes' of different length, every byte value present."""
  from geeco_amd import ops
  r = np.random.default_rng(61)
  HW = H * W
  fe = HW * 3
  eps = []
  for T in (K + 5, K + 2):
    f = r.integers(0, 256, size=(T, fe), dtype=np.uint8)
    f.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    f.reshape(-1)[-256:] = np.arange(255, -1, -1).astype(np.uint8)
    eps.append(torch.tensor(f, device=dev))
  tgts = [torch.tensor(r.integers(0, 256, size=(1, fe), dtype=np.uint8), device=dev) for _ in eps]
  pick = [(n % 2, [0, 2, 5, 1][n % 4] % (eps[n % 2].shape[0] - K + 1)) for n in range(N)]
  pick[-1] = (0, eps[0].shape[0] - K)                      # the last window an episode has
  # dense path: gather (u8 -> float / 255) then the fp32 input stage
  rgb = torch.empty(N, K, H, W, 3, device=dev)
  tgt = torch.empty(N, H, W, 3, device=dev)
  for n, (e, st) in enumerate(pick):
    ops.gather_windows_into(rgb[n:n + 1], eps[e], torch.tensor([st], dtype=torch.int32, device=dev), 1, K, fe, 255.0)
    ops.gather_windows_into(tgt[n:n + 1].view(1, 1, H, W, 3), tgts[e], torch.zeros(1, dtype=torch.int32, device=dev), 1, 1, fe, 255.0)
  win = torch.tensor([eps[e].data_ptr() + st * fe for e, st in pick], dtype=torch.int64, device=dev)
  tpt = torch.tensor([tgts[e].data_ptr() for e, _ in pick], dtype=torch.int64, device=dev)
  ws2 = ops.goal_dynimgs_ws(N, HW, dev)
  ref = [torch.empty(N, H, W, 4, device=dev) for _ in range(3)]
  got = [torch.full((N, H, W, 4), float('nan'), device=dev) for _ in range(3)]
  kw = {}
  if C == 4:
    dep = torch.tensor((0.5 + 2.5 * r.random([N, K, H, W, 1])).astype(np.float32), device=dev)
    tdep = torch.tensor((0.5 + 2.5 * r.random([N, H, W, 1])).astype(np.float32), device=dev)
    kw = dict(depth=dep, tgt_depth=tdep, dsample_stride=K * HW, dframe_stride=HW)
  ops.goal_dynimgs_into(ref[0], ref[1], ref[2], rgb, tgt, K, N, HW, ws2, K * fe, fe, **kw)
  ops.goal_dynimgs_u8_into(got[0], got[1], got[2], win, tpt, K, N, HW, ws2, **kw)
  torch.cuda.synchronize()
  # the dense windows themselves are the reference's values: float32(u8) / float32(255)
  e, st = pick[0]
  np.testing.assert_array_equal(rgb[0].cpu().numpy().reshape(K, fe),
                                eps[e][st:st + K].cpu().numpy().astype(np.float32) / np.float32(255.0))
  for name, a, b in zip(('current frame', 'buffer image', 'pair image'), got, ref):
    assert torch.equal(a, b), name
  with pytest.raises(ValueError):
    ops.goal_dynimgs_u8_into(got[0], got[1], got[2], win.int(), tpt, K, N, HW, ws2, **kw)


@pytest.mark.parametrize('G,N,H,W', [(1, 2, 32, 64), (3, 2, 64, 64)])
def test_conv3_relu_fields_and_conv4_dgrad_fields(dev, G, N, H, W):
  """conv3's forward with the byte sign fields of its output (byte (T >> 1) * 4 + q, bit 4 (T & 1) + j <-> channel
  16 T + 4 q + j) and the LDS-staged input gradient of the next layer (64 -> 128) masked by them: y3 equals the plain
  forward, the fields equal the packed signs, dx equals (bitwise) the gradient masked by y3 itself."""
  from geeco_amd import ops
  r = np.random.default_rng(59)
  x = torch.tensor(r.standard_normal([G, N, H, W, 48]).astype(np.float32), device=dev)
  w3 = torch.tensor((r.standard_normal([G, 3, 3, 48, 64]) / 20).astype(np.float32), device=dev)
  b3 = torch.tensor((0.1 * r.standard_normal([G, 64])).astype(np.float32), device=dev)
  H3, W3 = H // 2, W // 2
  y_plain = torch.empty(G, N, H3, W3, 64, device=dev)
  ws = torch.empty(ops.conv3x3_fwd_ws_bytes(G, N, H, W, 48, 64, 2) // 4 + 4, device=dev)
  ops.conv3x3_fwd_into(y_plain, x, w3, b3, G, x[0].numel(), w3[0].numel(), 64, y_plain[0].numel(), N, H, W, 48, 64, 2,
                       relu=True, ws=ws)
  y = torch.empty_like(y_plain)
  fields = torch.zeros(G, N, H3, W3, 8, dtype=torch.uint8, device=dev)
  names = ops.kernel_trace(lambda: ops.conv3_fwd_relu_fields_into(y, fields, x, w3, b3, G, x[0].numel(), w3[0].numel(), 64,
                                                                  y[0].numel(), fields[0].numel(), N, H, W))
  torch.cuda.synchronize()
  assert names[0].startswith('conv_s2_halo_fwd_chunked_kernel'), names
  assert torch.equal(y, y_plain)
  pos = (y.cpu().numpy() > 0).reshape(G, N, H3, W3, 4, 4, 4)          # [.., T, q, j]
  want = np.zeros([G, N, H3, W3, 8], np.uint8)
  for T in range(4):
    for q in range(4):
      for j in range(4):
        want[..., (T >> 1) * 4 + q] |= (pos[..., T, q, j].astype(np.uint8) << (4 * (T & 1) + j))
  assert np.array_equal(fields.cpu().numpy(), want)
  # the next layer's input gradient (64 -> 128, stride 2) on the y3 grid
  w4 = torch.tensor((r.standard_normal([G, 3, 3, 64, 128]) / 24).astype(np.float32), device=dev)
  dz4 = torch.tensor(r.standard_normal([G, N, H3 // 2, W3 // 2, 128]).astype(np.float32), device=dev)
  wt = torch.empty(G, 3, 3, 128, 64, device=dev)
  ops.transpose_hwio_into(wt, w4, G, w4[0].numel(), wt[0].numel(), 64, 128)
  dx_ref = torch.full((G, N, H3, W3, 64), float('nan'), device=dev)
  dws = torch.empty(ops.conv3x3_dgrad_ws_bytes(G, N, H3, W3, 64, 128, 2) // 4 + 4, device=dev)
  n0 = ops.kernel_trace(lambda: ops.conv3x3_dgrad_into(dx_ref, dz4, wt, y, G, dz4[0].numel(), wt[0].numel(), dx_ref[0].numel(), N,
                                                       H3, W3, 64, 128, 2, ws=dws, w=w4, gs_w=w4[0].numel()))
  dx = torch.full_like(dx_ref, float('nan'))
  n1 = ops.kernel_trace(lambda: ops.conv3x3_dgrad_relu_fields_into(dx, dz4, w4, fields, G, dz4[0].numel(), w4[0].numel(),
                                                                   fields[0].numel(), dx[0].numel(), N, H3, W3, 64, 128, 2))
  torch.cuda.synchronize()
  assert n0 == n1 and n1[0].startswith('conv_s2_dgrad_lds_kernel'), (n0, n1)
  assert not torch.isnan(dx).any() and torch.equal(dx, dx_ref)


def test_slab_reduce_batch_bitwise(dev):
  """Deferred slab sums (geeco_conv3x3_wgrad_partial x 4 layers of different kernels + the fused bottom, then ONE
  geeco_slab_reduce_batch) give bitwise the gradients of the plain calls."""
  from geeco_amd import ops
  G, Nf = 3, 4
  r = np.random.default_rng(41)
  layers = [(32, 48, 64, 64, 2), (48, 64, 32, 32, 2), (64, 128, 16, 16, 2), (128, 192, 8, 8, 2), (256, 256, 4, 4, 2)]
  jobs = []
  for Cin, Cout, H, W, stride in layers:
    x = torch.tensor(r.standard_normal([G, Nf, H, W, Cin]).astype(np.float32), device=dev)
    dz = torch.tensor(r.standard_normal([G, Nf, H // stride, W // stride, Cout]).astype(np.float32), device=dev)
    ws = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, Nf, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
    jobs.append((Cin, Cout, H, W, stride, x, dz, ws))

  def run(pending):
    outs = []
    for Cin, Cout, H, W, stride, x, dz, ws in jobs:
      dw = torch.full((G, 9 * Cin * Cout), float('nan'), device=dev)
      db = torch.full((G, Cout), float('nan'), device=dev)
      ops.conv3x3_wgrad_into(dw, db, x, dz, G, x[0].numel(), dz[0].numel(), dw[0].numel(), Cout, Nf, H, W, Cin, Cout,
                             stride, ws, pending=pending)
      outs += [dw, db]
    # fused encoder bottom
    H, W = 16, 64
    rr = np.random.default_rng(43)
    x4 = torch.tensor(rr.standard_normal([G, Nf, H, W, 4]).astype(np.float32), device=dev)
    x4[..., 3] = 0
    y1 = torch.tensor(rr.standard_normal([G, Nf, H, W, 32]).astype(np.float32), device=dev)
    w2 = torch.tensor((rr.standard_normal([G, 3, 3, 32, 48]) / 17).astype(np.float32), device=dev)
    dz2 = torch.tensor(rr.standard_normal([G, Nf, H // 2, W // 2, 48]).astype(np.float32), device=dev)
    dw1 = torch.full((G, 9, 3, 32), float('nan'), device=dev)
    db1 = torch.full((G, 32), float('nan'), device=dev)
    wsf = torch.empty(ops.conv2_dgrad_conv1_wgrad_ws_bytes(G) // 4 + 4, device=dev)
    ops.conv2_dgrad_conv1_wgrad_into(dw1, db1, dz2, w2, y1, x4, G, dz2[0].numel(), w2[0].numel(), y1[0].numel(),
                                     x4[0].numel(), dw1[0].numel(), 32, Nf, H, W, wsf, real_channels=3, pending=pending)
    outs += [dw1, db1]
    return outs, wsf

  plain, _ = run(None)
  pending = []
  deferred, keep = run(pending)
  assert 3 <= len(pending) <= 6, len(pending)       # the small top layer may write its gradient directly
  names = ops.kernel_trace(lambda: ops.slab_reduce_batch(pending))
  torch.cuda.synchronize()
  assert names == ['wgrad_reduce_batch_kernel'] and not pending, names
  for a, b in zip(plain, deferred):
    assert not torch.isnan(b).any()
    assert torch.equal(a, b)


@pytest.mark.parametrize('Cin,Cout,H,W', [(48, 64, 40, 72), (64, 128, 24, 40), (128, 192, 16, 16)])
def test_conv3x3_wgrad_grouped_lds(dev, Cin, Cout, H, W):
  """Grouped launch (G encoders at padded arena strides) of the LDS-staged filter gradient: every encoder's dw / db
  against the fp64 oracle; also checks the dispatch really took the new kernel."""
  from geeco_amd import ops
  G, Nf, stride = 3, 5, 2
  Ho, Wo = H // 2, W // 2
  r = np.random.default_rng(33)
  x = r.standard_normal([G, Nf, H, W, Cin]).astype(np.float32)
  dz = r.standard_normal([G, Nf, Ho, Wo, Cout]).astype(np.float32)
  xd, dzd = torch.tensor(x, device=dev), torch.tensor(dz, device=dev)
  gs_w, gs_b = 9 * Cin * Cout + 32, Cout + 16
  dw = torch.full((G, gs_w), float('nan'), device=dev)
  db = torch.full((G, gs_b), float('nan'), device=dev)
  ws = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, Nf, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
  names = ops.kernel_trace(lambda: ops.conv3x3_wgrad_into(dw, db, xd, dzd, G, xd[0].numel(), dzd[0].numel(), gs_w, gs_b, Nf,
                                                          H, W, Cin, Cout, stride, ws))
  torch.cuda.synchronize()
  assert names and names[0].startswith('conv_s2_wgrad_lds_kernel'), names
  scale = np.sqrt(Nf * Ho * Wo)
  for g in range(G):
    wt = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = O.conv2d_same(torch.tensor(x[g], dtype=torch.float64), wt, bt, stride, relu=False)
    y.backward(torch.tensor(dz[g], dtype=torch.float64))
    _close(dw[g, :9 * Cin * Cout].reshape(3, 3, Cin, Cout), wt.grad, 2e-5, 2e-5 * scale, 'grouped wgrad, encoder %d' % g)
    _close(db[g, :Cout], bt.grad, 2e-5, 2e-5 * scale, 'grouped bias grad, encoder %d' % g)
  assert torch.isnan(dw[:, 9 * Cin * Cout:]).all() and torch.isnan(db[:, Cout:]).all()     # pads untouched


@pytest.mark.parametrize('Cin,Cout,H,W,Nd', [(64, 128, 32, 32, 5), (128, 192, 16, 32, 4), (192, 256, 16, 16, 7)])
def test_conv3x3_dgrad_grouped_lds(dev, Cin, Cout, H, W, Nd):
  """Grouped launch (G encoders, padded arena strides) of the LDS-staged input gradient with the fused ReluGrad mask and
  WITHOUT a mask, against the fp64 oracle; also checks that the dispatch took the new kernel."""
  from geeco_amd import ops
  G, stride = 3, 2
  Ho, Wo = H // 2, W // 2
  r = np.random.default_rng(51)
  gs_w = 9 * Cin * Cout + 48
  warena = torch.zeros(G, gs_w, device=dev)
  w = (r.standard_normal([G, 3, 3, Cin, Cout]) / np.sqrt(9 * Cout)).astype(np.float32)
  warena[:, :9 * Cin * Cout] = torch.tensor(w.reshape(G, -1), device=dev)
  dz = r.standard_normal([G, Nd, Ho, Wo, Cout]).astype(np.float32)
  mask = r.standard_normal([G, Nd, H, W, Cin]).astype(np.float32)
  dzd, md = torch.tensor(dz, device=dev), torch.tensor(mask, device=dev)
  wt = torch.empty(G, 3, 3, Cout, Cin, device=dev)
  ops.transpose_hwio_into(wt, warena, G, gs_w, wt[0].numel(), Cin, Cout)
  dws = torch.empty(ops.conv3x3_dgrad_ws_bytes(G, Nd, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
  for use_mask in (True, False):
    dx = torch.full((G, Nd, H, W, Cin), float('nan'), device=dev)
    names = ops.kernel_trace(lambda: ops.conv3x3_dgrad_into(dx, dzd, wt, md if use_mask else None, G, dzd[0].numel(), wt[0].numel(),
                                                            dx[0].numel(), Nd, H, W, Cin, Cout, stride, ws=dws, w=warena, gs_w=gs_w))
    torch.cuda.synchronize()
    assert names and names[0].startswith('conv_s2_dgrad_lds_kernel'), names
    for g in range(G):
      xg = torch.zeros(Nd, H, W, Cin, dtype=torch.float64, requires_grad=True)
      yy = O.conv2d_same(xg, torch.tensor(w[g], dtype=torch.float64), torch.zeros(Cout, dtype=torch.float64), stride, relu=False)
      yy.backward(torch.tensor(dz[g], dtype=torch.float64))
      ref = xg.grad * (torch.tensor(mask[g]) > 0) if use_mask else xg.grad
      _close(dx[g], ref, 2e-5, 2e-5, 'grouped dgrad (mask %s), encoder %d' % (use_mask, g))


@pytest.mark.parametrize('N,H,W,Cin,Cout,stride', [(2, 4, 4, 192, 256, 2), (4, 2, 2, 256, 256, 2), (1, 10, 10, 16, 64, 1),
                                                  (2, 9, 7, 16, 16, 2), (3, 8, 8, 64, 96, 2),
                                                  (64, 33, 33, 128, 192, 2)])   # the last one takes 128 x 128 tiles
def test_conv3x3_dgrad_hwio_vs_transposed_copy(dev, N, H, W, Cin, Cout, stride):
  """The gather-GEMM input gradient reading the HWIO kernel itself (tile transposed on its way into LDS; the default when
  the caller hands over `w`) against the same kernel fed the per-tap transposed copy `wt` (w = NULL): same K order, so the
  results are bitwise equal; wt may be NULL in the first form."""
  from geeco_amd import ops
  r = np.random.default_rng(91)
  Ho, Wo = -(-H // stride), -(-W // stride)
  dz = torch.tensor(r.standard_normal([N, Ho, Wo, Cout]).astype(np.float32), device=dev)
  w = torch.tensor((r.standard_normal([3, 3, Cin, Cout]) / np.sqrt(9 * Cout)).astype(np.float32), device=dev)
  mask = torch.tensor(r.standard_normal([N, H, W, Cin]).astype(np.float32), device=dev)
  wt = torch.empty(3, 3, Cout, Cin, device=dev)
  ops.transpose_hwio_into(wt, w, 1, 0, 0, Cin, Cout)
  ws = ops._ws(ops.conv3x3_dgrad_ws_bytes(1, N, H, W, Cin, Cout, stride), dev)
  assert not ops.conv3x3_dgrad_needs_wt(H, W, Cin, Cout, stride)
  a = torch.full((N, H, W, Cin), float('nan'), device=dev)
  b = torch.full((N, H, W, Cin), float('nan'), device=dev)
  names = ops.kernel_trace(lambda: ops.conv3x3_dgrad_into(a, dz, None, mask, 1, 0, 0, 0, N, H, W, Cin, Cout, stride, ws, w=w))
  ops.conv3x3_dgrad_into(b, dz, wt, mask, 1, 0, 0, 0, N, H, W, Cin, Cout, stride, ws, w=None)
  torch.cuda.synchronize()
  if names[0].startswith('conv_gemm_kernel'):
    assert torch.equal(a, b)
    if N == 64:
      assert names[0].startswith('conv_gemm_kernel<128, 128'), names
  else:                     # an LDS-staged kernel took the shape (it reads HWIO anyway)
    _close(a, b.double(), 2e-5, 2e-5, 'dgrad')


@pytest.mark.parametrize('G,N,H,W', [(1, 2, 16, 64), (3, 2, 20, 44)])
def test_conv1_rgb_kernel_variable(dev, G, N, H, W):
  """conv1's forward reading the RGB model's kernel variable [3][3][3][32] itself == reading its channel-padded copy:
  y1 and the sign words bitwise."""
  from geeco_amd import ops
  r = np.random.default_rng(97)
  x = torch.tensor(r.standard_normal([G, N, H, W, 4]).astype(np.float32), device=dev)
  x[..., 3] = 0
  w3 = torch.tensor((r.standard_normal([G, 3, 3, 3, 32]) / 5).astype(np.float32), device=dev)
  w4 = torch.zeros(G, 3, 3, 4, 32, device=dev)
  w4[:, :, :, :3] = w3
  b = torch.tensor((0.1 * r.standard_normal([G, 32])).astype(np.float32), device=dev)
  Wp, Hp = ops.relu_bits_pitch(W), ops.relu_bits_rows(H)
  ys, bs = [], []
  for rgb in (False, True):
    y = torch.full((G, N, H, W, 32), float('nan'), device=dev)
    bits = torch.zeros(G, N, Hp, Wp, dtype=torch.int32, device=dev)
    if rgb:
      ops.conv1_fwd_relu_bits_rgb_into(y, bits, x, w3, b, G, x[0].numel(), w3[0].numel(), 32, y[0].numel(), bits[0].numel(), N, H, W)
    else:
      ops.conv1_fwd_relu_bits_into(y, bits, x, w4, b, G, x[0].numel(), w4[0].numel(), 32, y[0].numel(), bits[0].numel(), N, H, W)
    ys.append(y); bs.append(bits)
  torch.cuda.synchronize()
  assert torch.equal(ys[0], ys[1]) and torch.equal(bs[0], bs[1])


@pytest.mark.parametrize('N,chs,J', [(32, (256, 256, 256), 7), (5, (256, 128, 64), 7), (2, (256,), 7), (33, (64, 64), 3), (160, (256, 64), 7)])
def test_lstm_step_bwd_one_launch(dev, N, chs, J):
  """geeco_lstm_step_bwd (dWx tiles, dX split-K tiles and the bias sums as the blocks of one grid, then dX's slab sum with
  the state-concat backward in its epilogue) against the five separate launches it replaces (gemm ta, colsum, gemm tb with
  split-K + reduce, state_concat_bwd): bitwise (same tile code, same K and slab order), and against fp64; the
  joint-state columns of dX are not scattered anywhere.  N = 160: geeco_gemm_f32 splits dWx's batch dimension from N = 128
  on and the one-launch form does not (include/geeco_hip.h), so dWx is compared to fp64 / to the split form at fp32
  tolerance there, everything else stays bitwise."""
  from geeco_amd import ops
  r = np.random.default_rng(71)
  H4, cells = 512, 4
  nf = len(chs)
  Ctot = sum(chs) + J
  D = cells * Ctot
  x = torch.tensor(r.standard_normal([N, D]).astype(np.float32), device=dev)
  dz = torch.tensor(r.standard_normal([N, H4]).astype(np.float32), device=dev)
  w = torch.tensor((r.standard_normal([D + 128, H4]) / 30).astype(np.float32), device=dev)
  wx = w[:D]
  feats = [torch.tensor(r.standard_normal([N, cells, c]).astype(np.float32), device=dev) for c in chs]
  jnt_pos = min(2, nf)
  # separate launches
  dw_ref = torch.zeros(D + 128, H4, device=dev)
  db_ref = torch.empty(H4, device=dev)
  dx_ref = torch.empty(N, D, device=dev)
  ws = torch.empty(max(ops.gemm_ws_bytes(D, H4, N), ops.gemm_ws_bytes(N, D, H4)) // 4 + 4, device=dev)
  ops.gemm_into(dw_ref[:D], x, dz, D, H4, N, D, H4, H4, ta=True, ws=ws)
  ops.colsum_into(db_ref, dz, H4, N, H4)
  ops.gemm_into(dx_ref, dz, wx, N, D, H4, H4, H4, D, tb=True, ws=ws)
  df_ref = [torch.full_like(f, float('nan')) for f in feats]
  ops.state_concat_bwd_into(df_ref, dx_ref, D, feats, list(chs), jnt_pos, J, N, cells)
  # one launch
  dw = torch.zeros(D + 128, H4, device=dev)
  db = torch.full((H4,), float('nan'), device=dev)
  dx = torch.full((N, D), float('nan'), device=dev)
  df = [torch.full_like(f, float('nan')) for f in feats]
  ws2 = torch.empty(ops.lstm_step_bwd_ws_bytes(N, D, H4) // 4 + 4, device=dev)
  ops.lstm_step_bwd_into(dw[:D], db, dx, x, dz, wx, N, D, H4, H4, ws2, feats_fwd=feats, dfeats=df, feat_ch=list(chs),
                         jnt_pos=jnt_pos, J=J, cells=cells)
  torch.cuda.synchronize()
  if N < 128:
    assert torch.equal(dw, dw_ref)
  else:
    _close(dw[:D], x.double().cpu().T @ dz.double().cpu(), 1e-5, 2e-5 * np.sqrt(N), 'dWx vs fp64 (unsplit K)')
    _close(dw[:D], dw_ref[:D], 1e-5, 2e-5 * np.sqrt(N), 'dWx vs the split-K form')
  assert torch.equal(db, db_ref)
  assert float(dw[D:].abs().max()) == 0.0                      # the recurrent rows are not touched
  ref64 = dz.double().cpu() @ wx.double().cpu().T
  _close(dx, ref64, 1e-5, 2e-5 * np.sqrt(H4), 'dX vs fp64')
  assert torch.equal(dx, dx_ref)
  for a, b in zip(df, df_ref):
    assert not torch.isnan(a).any() and torch.equal(a, b)
  # without a concat description only the three products are written
  dx2 = torch.full((N, D), float('nan'), device=dev)
  ops.lstm_step_bwd_into(dw[:D], db, dx2, x, dz, wx, N, D, H4, H4, ws2)
  torch.cuda.synchronize()
  assert torch.equal(dx2, dx)


@pytest.mark.parametrize('N,H,D', [(32, 128, 3100), (5, 128, 1052), (1, 64, 40), (33, 96, 3100), (64, 128, 3100)])
def test_lstm_input_step_fwd_two_launches(dev, N, H, D):
  """geeco_lstm_input_step_fwd (the cell's first step: gate GEMM, then the gate math with the split-K slab sum inside) against
  geeco_gemm_f32 + geeco_lstm_gates_fwd(c_prev = NULL): bitwise z, c, h, gates (same tile code, K split and slab order), split
  (D = 3100: 39 slabs at N = 32) and unsplit (D = 40) products, row counts that do not fill a tile; z against fp64."""
  from geeco_amd import ops
  r = np.random.default_rng(83)
  x = torch.tensor(r.standard_normal([N, D]).astype(np.float32), device=dev)
  w = torch.tensor((r.standard_normal([D + H, 4 * H]) / np.sqrt(D)).astype(np.float32), device=dev)
  bias = torch.tensor(r.standard_normal([4 * H]).astype(np.float32), device=dev)
  ws = torch.empty(ops.gemm_ws_bytes(N, 4 * H, D) // 4 + 4, device=dev)
  ref = [torch.empty(N, 4 * H, device=dev), torch.empty(N, H, device=dev), torch.empty(N, H, device=dev), torch.empty(N, 4 * H, device=dev)]
  got = [torch.full_like(t, float('nan')) for t in ref]
  ops.gemm_into(ref[0], x, w[:D], N, 4 * H, D, D, 4 * H, 4 * H, ws=ws)
  ops.lstm_gates_fwd_into(ref[1], ref[2], ref[3], ref[0], bias, None, N, H)
  ops.lstm_input_step_fwd_into(got[0], got[1], got[2], got[3], x, w[:D], bias, N, H, D, D, 4 * H, ws)
  torch.cuda.synchronize()
  for name, a, b in zip(('z', 'c', 'h', 'gates'), got, ref):
    assert not torch.isnan(a).any() and torch.equal(a, b), name
  _close(got[0], x.double().cpu() @ w[:D].double().cpu(), 1e-5, 2e-5, 'z vs fp64')


@pytest.mark.parametrize('N,D,mode,deferred', [(32, 3100, 'cartesian', True), (5, 3100, 'cartesian', False), (33, 1052, 'velocity', True),
                                               (64, 40, 'cartesian', True), (1, 3100, 'cartesian', True)])
def test_lstm_step_heads_one_launch_per_sample(dev, N, D, mode, deferred):
  """geeco_lstm_step_heads_fwd_bwd (one-step decoder from the zero state: gate GEMM, then ONE workgroup per sample for the slab
  sum, the gate math, fc1, the heads, the loss terms and the way back to the gate gradients dz) against the entry points it
  stands for: geeco_lstm_input_step_fwd + geeco_heads_loss_fwd_bwd + geeco_lstm_gates_bwd.  z, c, h, gates, the predictions,
  every loss and every gradient BITWISE (same per-sample code, same slab / summation orders); with ``deferred`` the batch sums
  ride in geeco_lstm_step_bwd's first grid (pending), whose own outputs stay bitwise those of the call without them; split
  (39 slabs) and unsplit (D = 40) gate products; forward only (evaluation): predictions and losses."""
  from geeco_amd import _native, ops
  r = np.random.default_rng(131 + N)
  H = F = 128
  heads = [(3, 0, 1.0), (3, 1, 1.0), (3, 0, 0.5), (3, 0, 0.5)] if mode == 'cartesian' else \
          [(7, 0, 1.0), (3, 0, 1.0), (2, 0, 1.0), (3, 0, 1.0), (3, 0, 1.0)]
  t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
  x = t(r.standard_normal([N, D]))
  w = t(r.standard_normal([D + H, 4 * H]) / np.sqrt(D))
  bias = t(r.standard_normal([4 * H]))
  w1, b1 = t(0.1 * r.standard_normal([H, F])), t(0.1 * r.standard_normal([F]))
  hw = [t(0.1 * r.standard_normal([F, sz])) for sz, _, _ in heads]
  hb = [t(0.1 * r.standard_normal([sz])) for sz, _, _ in heads]
  tg = [t(r.standard_normal([N, sz])) if kind == 0 else t(r.integers(-1, 2, [N, 1])) for sz, kind, _ in heads]
  meta = ([sz for sz, _, _ in heads], [k for _, k, _ in heads], [wt for _, _, wt in heads], tg, [a.shape[1] for a in tg])
  OT = sum(meta[0])
  gws = torch.empty(max(ops.gemm_ws_bytes(N, 4 * H, D), ops.lstm_step_bwd_ws_bytes(N, D, 4 * H)) // 4 + 4, device=dev)
  hws = torch.empty(ops.heads_ws_bytes(N, H, F) // 4 + 4, device=dev)

  def outputs():
    o = dict(z=torch.full((N, 4 * H), float('nan'), device=dev), c=torch.full((N, H), float('nan'), device=dev),
             h=torch.full((N, H), float('nan'), device=dev), gates=torch.full((N, 4 * H), float('nan'), device=dev),
             preds=torch.full((N, OT), float('nan'), device=dev), losses=torch.zeros(8, device=dev),
             dz=torch.full((N, 4 * H), float('nan'), device=dev), dw1=torch.full((H, F), float('nan'), device=dev),
             db1=torch.full((F,), float('nan'), device=dev), dhw=[torch.full_like(a, float('nan')) for a in hw],
             dhb=[torch.full_like(a, float('nan')) for a in hb], dwx=torch.zeros(D, 4 * H, device=dev),
             dbx=torch.full((4 * H,), float('nan'), device=dev), dx=torch.full((N, D), float('nan'), device=dev))
    return o
  # ---- the separate entry points ----
  a = outputs()
  dh = torch.empty(N, H, device=dev)
  ops.lstm_input_step_fwd_into(a['z'], a['c'], a['h'], a['gates'], x, w[:D], bias, N, H, D, D, 4 * H, gws)
  ops.heads_loss_into(a['preds'], a['losses'], a['h'], w1, b1, hw, hb, *meta, 1.0, N, H, F, hws, dh=dh, d_fc1_w=a['dw1'],
                      d_fc1_b=a['db1'], d_heads_w=a['dhw'], d_heads_b=a['dhb'])
  ops.lstm_gates_bwd_into(a['dz'], None, a['gates'], None, a['c'], dh, None, N, H)
  ops.lstm_step_bwd_into(a['dwx'], a['dbx'], a['dx'], x, a['dz'], w[:D], N, D, 4 * H, 4 * H, gws)
  torch.cuda.synchronize()
  # ---- the fused form ----
  b = outputs()
  hws2 = torch.empty_like(hws)
  pend = _native.HeadsFinish() if deferred else None
  names = ops.kernel_trace(lambda: ops.lstm_step_heads_into(
      b['z'], b['c'], b['h'], b['gates'], x, w[:D], bias, N, H, D, D, 4 * H, gws, b['preds'], b['losses'], w1, b1, hw, hb, *meta, 1.0, F,
      hws2, dz=b['dz'], d_fc1_w=b['dw1'], d_fc1_b=b['db1'], d_heads_w=b['dhw'], d_heads_b=b['dhb'], pending=pend))
  assert names == ['gemm_f32_kernel', 'heads_sample_kernel<true>'] + ([] if deferred else ['heads_finish_kernel']), names
  names = ops.kernel_trace(lambda: ops.lstm_step_bwd_into(b['dwx'], b['dbx'], b['dx'], x, b['dz'], w[:D], N, D, 4 * H, 4 * H, gws,
                                                         pending=pend))
  assert ('lstm_step_bwd_heads_kernel' in names) == deferred, names
  torch.cuda.synchronize()
  for k in a:
    for u, v in zip(a[k] if isinstance(a[k], list) else [a[k]], b[k] if isinstance(b[k], list) else [b[k]]):
      assert not torch.isnan(v).any() and torch.equal(u, v), k
  # ---- forward only ----
  c = outputs()
  assert ops.lstm_step_heads_into(c['z'], c['c'], c['h'], c['gates'], x, w[:D], bias, N, H, D, D, 4 * H, gws, c['preds'], c['losses'],
                                  w1, b1, hw, hb, *meta, 1.0, F, hws2)
  torch.cuda.synchronize()
  for k in ('z', 'c', 'h', 'gates', 'preds', 'losses'):
    assert torch.equal(a[k], c[k]), k
  # a shape the per-sample kernel does not serve is declined, nothing launched
  assert ops.lstm_step_heads_into(c['z'], c['c'], c['h'], c['gates'], x, w[:D], bias, N, H, D, D, 4 * H, gws, c['preds'], c['losses'],
                                  torch.zeros(H, 96, device=dev), torch.zeros(96, device=dev), [torch.zeros(96, sz, device=dev) for sz in meta[0]],
                                  hb, *meta, 1.0, 96, torch.empty(ops.heads_ws_bytes(N, H, 96) // 4 + 4, device=dev)) is False


@pytest.mark.parametrize('G,N,C,J', [(3, 32, 256, 7), (3, 5, 256, 7), (1, 2, 64, 3), (2, 33, 128, 7)])
def test_top_layer_forward_carries_the_state_concat(dev, G, N, C, J):
  """geeco_conv3x3_fwd_state (conv8 + bias + ReLU of all encoders, the one-step decoder's state concat in the epilogue of the
  split-K sum) against geeco_conv3x3_fwd + geeco_state_concat_fwd: features and state bitwise, no state element left
  unwritten; a shape without a split-K epilogue is declined (False, nothing written)."""
  from geeco_amd import ops
  r = np.random.default_rng(97)
  H = W = 4
  cells = 4
  x = torch.tensor(r.standard_normal([G, N, H, W, 256]).astype(np.float32), device=dev)
  w = torch.tensor((r.standard_normal([G, 3, 3, 256, C]) / 48).astype(np.float32), device=dev)
  b = torch.tensor(r.standard_normal([G, C]).astype(np.float32), device=dev)
  K = 3
  jnts = torch.tensor(r.standard_normal([N, K, J]).astype(np.float32), device=dev)
  jnt = jnts[:, K - 1]
  ch = [C] * G
  jnt_pos = min(2, G)
  Ctot = G * C + J
  off, o = [], 0
  for g in range(G):
    if g == jnt_pos:
      o += J
    off.append(o)
    o += C
  jnt_off = sum(ch[:jnt_pos])
  ws = torch.empty(ops.conv3x3_fwd_ws_bytes(G, N, H, W, 256, C, 2) // 4 + 4, device=dev)
  y_ref = torch.empty(G, N, 2, 2, C, device=dev)
  st_ref = torch.full((N, cells * Ctot), float('nan'), device=dev)
  ops.conv3x3_fwd_into(y_ref, x, w, b, G, x[0].numel(), w[0].numel(), C, y_ref[0].numel(), N, H, W, 256, C, 2, relu=True, ws=ws)
  ops.state_concat_fwd_into(st_ref, [y_ref[g] for g in range(G)], ch, jnt_pos, jnt, K * J, J, N, cells, cells * Ctot)
  y = torch.full_like(y_ref, float('nan'))
  st = torch.full_like(st_ref, float('nan'))
  did = ops.conv3x3_fwd_state_into(y, x, w, b, G, x[0].numel(), w[0].numel(), C, y[0].numel(), N, H, W, 256, C, 2, ws, state=st,
                                   state_stride=cells * Ctot, feat_off=off, Ctot=Ctot, jnt=jnt, jnt_stride=K * J, jnt_off=jnt_off, J=J)
  torch.cuda.synchronize()
  assert did
  assert not torch.isnan(st_ref).any()
  assert torch.equal(y, y_ref)
  assert torch.equal(st, st_ref)
  # no workspace = no split-K epilogue: declined, nothing launched
  st2 = torch.full_like(st_ref, float('nan'))
  assert not ops.conv3x3_fwd_state_into(y, x, w, b, G, x[0].numel(), w[0].numel(), C, y[0].numel(), N, H, W, 256, C, 2, None, state=st2,
                                        state_stride=cells * Ctot, feat_off=off, Ctot=Ctot, jnt=jnt, jnt_stride=K * J, jnt_off=jnt_off, J=J)
  torch.cuda.synchronize()
  assert torch.isnan(st2).all()
  with pytest.raises(RuntimeError):      # overlapping feature / joint columns
    ops.conv3x3_fwd_state_into(y, x, w, b, G, x[0].numel(), w[0].numel(), C, y[0].numel(), N, H, W, 256, C, 2, ws, state=st2,
                               state_stride=cells * Ctot, feat_off=off, Ctot=Ctot, jnt=jnt, jnt_stride=K * J, jnt_off=0, J=J)


def test_reserved_cus_shrink_the_persistent_bottom_kernels(dev):
  """``reserved_cus`` = k (an ARGUMENT of the deferred-slab-sum entry points since ABI 4): conv2's filter gradient and the fused
  bottom launch (256 - k) / groups blocks per encoder (the workspace is sized for k = 0); results equal the k = 0 launches to
  fp32 summation-order differences (other slab partition); nothing outlives the call (the next plain call is bitwise the
  k = 0 result); values outside 0..128 and the plain (no ``pending``) forms with k > 0 are refused."""
  from geeco_amd import ops
  G, N, H, W = 3, 4, 32, 128
  r = np.random.default_rng(83)
  y1 = torch.tensor(r.standard_normal([G, N, H, W, 32]).astype(np.float32), device=dev)
  x4 = torch.tensor(r.standard_normal([G, N, H, W, 4]).astype(np.float32), device=dev)
  x4[..., 3] = 0
  dz2 = torch.tensor(r.standard_normal([G, N, H // 2, W // 2, 48]).astype(np.float32), device=dev)
  w2 = torch.tensor((r.standard_normal([G, 3, 3, 32, 48]) / 17).astype(np.float32), device=dev)
  wsf = torch.empty(ops.conv2_dgrad_conv1_wgrad_ws_bytes(G) // 4 + 4, device=dev)
  wsw = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, N, H, W, 32, 48, 2) // 4 + 4, device=dev)

  def run(k):
    pending = []
    dw1 = torch.full((G, 9, 3, 32), float('nan'), device=dev)
    db1 = torch.full((G, 32), float('nan'), device=dev)
    ops.conv2_dgrad_conv1_wgrad_into(dw1, db1, dz2, w2, y1, x4, G, dz2[0].numel(), w2[0].numel(), y1[0].numel(), x4[0].numel(),
                                     dw1[0].numel(), 32, N, H, W, wsf, real_channels=3, pending=pending, reserved_cus=k)
    dw2 = torch.full((G, 9 * 32 * 48), float('nan'), device=dev)
    db2 = torch.full((G, 48), float('nan'), device=dev)
    ops.conv3x3_wgrad_into(dw2, db2, y1, dz2, G, y1[0].numel(), dz2[0].numel(), dw2[0].numel(), 48, N, H, W, 32, 48, 2, wsw,
                           pending=pending, reserved_cus=k)
    slabs = [it.S for it in pending]
    ops.slab_reduce_batch(pending)
    torch.cuda.synchronize()
    return (dw1, db1, dw2, db2), slabs

  ref, slabs0 = run(0)
  got, slabs16 = run(16)
  assert slabs16 == [(256 - 16) // G] * 2 and all(a >= b for a, b in zip(slabs0, slabs16)), (slabs0, slabs16)
  scale = np.sqrt(N * H * W)
  for a, b, what in zip(got, ref, ('dw1', 'db1', 'dw2', 'db2')):
    assert not torch.isnan(a).any()
    _close(a, b.double(), 2e-5, 2e-5 * scale, what + ' with 16 CUs reserved')
  again, slabs = run(0)
  assert slabs == slabs0 and torch.equal(again[0], ref[0]) and torch.equal(again[2], ref[2])     # nothing outlives the call
  with pytest.raises(Exception):
    run(500)
  with pytest.raises(ValueError):
    ops.conv3x3_wgrad_into(torch.empty(G, 9 * 32 * 48, device=dev), torch.empty(G, 48, device=dev), y1, dz2, G, y1[0].numel(),
                           dz2[0].numel(), 9 * 32 * 48, 48, N, H, W, 32, 48, 2, wsw, reserved_cus=16)


@pytest.mark.parametrize('G,N,dim_out', [(3, 32, 256), (1, 5, 256), (2, 3, 128)])
def test_wgrad_pair_conv7_conv8_bitwise(dev, G, N, dim_out):
  """conv7's and conv8's filter gradients as ONE grid (geeco_conv3x3_wgrad_pair) against the two separate launches: bitwise
  (the same tile code on the same (split, tile, group) decomposition), both against fp64 in their own tests; the separate
  launches really are the generic 64 x 64-tile kernel for these shapes; shapes outside the paired kernel report "launch twice"."""
  from geeco_amd import ops
  r = np.random.default_rng(89)
  shapes = [(8, 256, 256), (4, 256, dim_out)]          # (H = W of the layer's input, Cin, Cout): conv7, conv8
  probs, refs = [], []
  for H, Cin, Cout in shapes:
    x = torch.tensor(r.standard_normal([G, N, H, H, Cin]).astype(np.float32), device=dev)
    dz = torch.tensor(r.standard_normal([G, N, H // 2, H // 2, Cout]).astype(np.float32), device=dev)
    ws = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, N, H, H, Cin, Cout, 2) // 4 + 4, device=dev)
    dw_ref = torch.full((G, 9 * Cin * Cout), float('nan'), device=dev)
    db_ref = torch.full((G, Cout), float('nan'), device=dev)
    names = ops.kernel_trace(lambda: ops.conv3x3_wgrad_into(dw_ref, db_ref, x, dz, G, x[0].numel(), dz[0].numel(), dw_ref[0].numel(),
                                                            Cout, N, H, H, Cin, Cout, 2, ws))
    assert names[0] == 'conv_wgrad_kernel<64, 64, 32>', names
    refs.append((dw_ref, db_ref))
    probs.append(dict(dw=torch.full_like(dw_ref, float('nan')), db=torch.full_like(db_ref, float('nan')), x=x, dz=dz, gs_x=x[0].numel(),
                      gs_dz=dz[0].numel(), gs_dw=dw_ref[0].numel(), gs_db=Cout, N=N, H=H, W=H, Cin=Cin, Cout=Cout,
                      ws=torch.empty_like(ws)))
  pending = []
  names = ops.kernel_trace(lambda: ops.conv3x3_wgrad_pair_into(probs[0], probs[1], G, 2, pending=pending))
  ops.slab_reduce_batch(pending)
  torch.cuda.synchronize()
  assert names[0] == 'conv_wgrad_pair_kernel<64, 64, 32>', names
  for pr, (dw_ref, db_ref) in zip(probs, refs):
    assert not torch.isnan(pr['dw']).any() and not torch.isnan(pr['db']).any()
    assert torch.equal(pr['dw'], dw_ref) and torch.equal(pr['db'], db_ref)
  # a layer the LDS-staged kernels serve (Cin = 192) is refused: nothing launched, the caller falls back
  bad = dict(probs[0], Cin=192)
  assert ops.conv3x3_wgrad_pair_into(bad, probs[1], G, 2) is False


@pytest.mark.parametrize('G,N,dim_out', [(3, 32, 256), (1, 5, 256), (2, 3, 128)])
def test_top_bwd_heterogeneous_grid_bitwise(dev, G, N, dim_out):
  """conv7's input gradient + conv7's / conv8's filter gradients as ONE heterogeneous grid (geeco_conv_top_bwd) against the
  separate launches (geeco_conv3x3_dgrad, 2 x geeco_conv3x3_wgrad): dx, both dw and both db bitwise; the separate input
  gradient really is the gather GEMM with split K for this shape; a shape the LDS-staged kernels serve is refused."""
  from geeco_amd import ops
  r = np.random.default_rng(97)
  # conv7: 8 x 8 x 256 -> 4 x 4 x 256; conv8: 4 x 4 x 256 -> 2 x 2 x dim_out
  x6 = torch.tensor(r.standard_normal([G, N, 8, 8, 256]).astype(np.float32), device=dev)        # conv7's input (ReluGrad mask of dx)
  dz7 = torch.tensor(r.standard_normal([G, N, 4, 4, 256]).astype(np.float32), device=dev)
  x7 = torch.tensor(r.standard_normal([G, N, 4, 4, 256]).astype(np.float32), device=dev)        # conv8's input
  dz8 = torch.tensor(r.standard_normal([G, N, 2, 2, dim_out]).astype(np.float32), device=dev)
  w7 = torch.tensor((r.standard_normal([G, 3, 3, 256, 256]) / 48).astype(np.float32), device=dev)
  dws = torch.empty(ops.conv3x3_dgrad_ws_bytes(G, N, 8, 8, 256, 256, 2) // 4 + 4, device=dev)
  dx_ref = torch.full((G, N, 8, 8, 256), float('nan'), device=dev)
  names = ops.kernel_trace(lambda: ops.conv3x3_dgrad_into(dx_ref, dz7, None, x6, G, dz7[0].numel(), 0, dx_ref[0].numel(), N, 8, 8, 256,
                                                          256, 2, ws=dws, w=w7, gs_w=w7[0].numel()))
  assert names[0].startswith('conv_gemm_kernel<64, 64, 16'), names
  wg = []
  for x, dz, H, Cout in ((x6, dz7, 8, 256), (x7, dz8, 4, dim_out)):
    ws = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, N, H, H, 256, Cout, 2) // 4 + 4, device=dev)
    dw_ref = torch.full((G, 9 * 256 * Cout), float('nan'), device=dev)
    db_ref = torch.full((G, Cout), float('nan'), device=dev)
    ops.conv3x3_wgrad_into(dw_ref, db_ref, x, dz, G, x[0].numel(), dz[0].numel(), dw_ref[0].numel(), Cout, N, H, H, 256, Cout, 2, ws)
    wg.append((dw_ref, db_ref, dict(dw=torch.full_like(dw_ref, float('nan')), db=torch.full_like(db_ref, float('nan')), x=x, dz=dz,
                                    gs_x=x[0].numel(), gs_dz=dz[0].numel(), gs_dw=dw_ref[0].numel(), gs_db=Cout, N=N, H=H, W=H, Cin=256,
                                    Cout=Cout, ws=torch.empty_like(ws))))
  dx = torch.full_like(dx_ref, float('nan'))
  d = dict(dx=dx, dz=dz7, wt=None, ymask=x6, w=w7, gs_dz=dz7[0].numel(), gs_w=w7[0].numel(), gs_wt=0, gs_dx=dx[0].numel(), N=N, H=8,
           W=8, Cin=256, Cout=256, ws=torch.empty_like(dws))
  pending = []
  names = ops.kernel_trace(lambda: ops.conv_top_bwd_into(d, wg[0][2], wg[1][2], G, 2, pending=pending))
  ops.slab_reduce_batch(pending)
  torch.cuda.synchronize()
  assert names[0] == 'conv_top_bwd_kernel<true>', names
  assert not torch.isnan(dx).any() and torch.equal(dx, dx_ref)
  for dw_ref, db_ref, pr in wg:
    assert torch.equal(pr['dw'], dw_ref) and torch.equal(pr['db'], db_ref)
  # conv6's input gradient (192 -> 256 at 16 x 16) belongs to the LDS-staged kernel: refused, nothing launched
  bad = dict(d, Cin=192, H=16, W=16)
  assert ops.conv_top_bwd_into(bad, wg[0][2], wg[1][2], G, 2) is False


def test_conv2_wgrad_remainder_block(dev):
  """conv2's filter gradient, three encoders, 512 tiles each = 85 blocks x 6 + 2: the 256th block walks the two tiles
  every encoder leaves over (three segments, slab 85 of each encoder); every encoder's dw / db against the fp64 oracle,
  and the same with CUs reserved (80 blocks per encoder, no CU over: plain ceil slices)."""
  from geeco_amd import ops
  G, Nf, H, W, Cin, Cout, stride = 3, 8, 64, 256, 32, 48, 2
  Ho, Wo = H // 2, W // 2
  r = np.random.default_rng(35)
  x = r.standard_normal([G, Nf, H, W, Cin]).astype(np.float32)
  dz = r.standard_normal([G, Nf, Ho, Wo, Cout]).astype(np.float32)
  xd, dzd = torch.tensor(x, device=dev), torch.tensor(dz, device=dev)
  gs_w, gs_b = 9 * Cin * Cout + 32, Cout + 16
  ws = torch.empty(ops.conv3x3_wgrad_ws_bytes(G, Nf, H, W, Cin, Cout, stride) // 4 + 4, device=dev)
  refs = []
  for g in range(G):
    wt = torch.zeros(3, 3, Cin, Cout, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = O.conv2d_same(torch.tensor(x[g], dtype=torch.float64), wt, bt, stride, relu=False)
    y.backward(torch.tensor(dz[g], dtype=torch.float64))
    refs.append((wt.grad, bt.grad))
  scale = np.sqrt(Nf * Ho * Wo)
  for reserved in (0, 16):
    dw = torch.full((G, gs_w), float('nan'), device=dev)
    db = torch.full((G, gs_b), float('nan'), device=dev)
    pending = []
    names = ops.kernel_trace(lambda: ops.conv3x3_wgrad_into(dw, db, xd, dzd, G, xd[0].numel(), dzd[0].numel(), gs_w, gs_b,
                                                            Nf, H, W, Cin, Cout, stride, ws, pending=pending,
                                                            reserved_cus=reserved))
    ops.slab_reduce_batch(pending)
    torch.cuda.synchronize()
    assert names and names[0].startswith('conv_s2_halo_wgrad_kernel'), names
    for g in range(G):
      _close(dw[g, :9 * Cin * Cout].reshape(3, 3, Cin, Cout), refs[g][0], 2e-5, 2e-5 * scale, 'wgrad, encoder %d' % g)
      _close(db[g, :Cout], refs[g][1], 2e-5, 2e-5 * scale, 'bias grad, encoder %d' % g)
