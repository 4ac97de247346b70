"""Test of the development-only one-launch encoder bottom forward (see README.md in this directory).  Needs the development
library:  scripts/dev/build_dev_lib.sh && GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so python -m pytest scripts/dev/experiments/conv_bottom_fwd -q
(not part of tests/: the product library does not contain this kernel)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..', '..', '..'))
for p_ in (ROOT, HERE, os.path.join(ROOT, 'tests')):
  if p_ not in sys.path:
    sys.path.insert(0, p_)

from oracle import geeco_oracle as O        # noqa: E402
from test_kernels_gpu import _close         # noqa: E402


@pytest.fixture(scope='module')
def dev():
  if not torch.cuda.is_available():
    pytest.skip('no GPU')
  from geeco_amd import _native
  if not _native.load().geeco_has_dev_kernels():
    pytest.skip('needs the development library (GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so)')
  return torch.device('cuda:0')


@pytest.mark.parametrize('G,N,H,W,C,side', [(1, 1, 8, 32, 3, True), (1, 2, 16, 64, 3, True), (3, 3, 24, 72, 3, True), (2, 3, 40, 136, 4, True),
                                            (3, 2, 256, 256, 3, True), (2, 5, 136, 136, 3, False), (1, 1, 2, 2, 4, True)])
def test_conv1_conv2_fused_forward_bitwise(dev, G, N, H, W, C, side):
  """The one-launch encoder bottom forward (csrc/conv_bottom_fwd.hip: conv1 waves fill an LDS image of y1 that conv2's waves
  consume) against the two separate launches it replaces: y1, conv1's sign words, y2 and conv2's sign fields are BITWISE equal
  (same MFMA sequence per output); and y1 / y2 against the fp64 oracle at the kernel tolerance.  Ragged tiles in both
  directions, tile ranges that cross encoders and frames, the RGB (packed-K, w_cin = 3) and RGB-D kernels, side outputs absent
  (evaluation); buffers are NaN-filled first: every output element must be written, and the padding of the side arrays never."""
  from geeco_amd import ops
  from binding import conv1_conv2_fwd_into
  r = np.random.default_rng(41)
  x3 = r.standard_normal([G, N, H, W, C]).astype(np.float32)
  x4 = np.concatenate([x3, np.zeros([G, N, H, W, 4 - C], np.float32)], -1)
  w1 = (r.standard_normal([G, 3, 3, C, 32]) / np.sqrt(9 * C)).astype(np.float32)
  b1 = (0.1 * r.standard_normal([G, 32])).astype(np.float32)
  w2 = (r.standard_normal([G, 3, 3, 32, 48]) / 17).astype(np.float32)
  b2 = (0.1 * r.standard_normal([G, 48])).astype(np.float32)
  xd, w1d, b1d, w2d, b2d = (torch.tensor(a, device=dev) for a in (x4, w1, b1, w2, b2))
  H2, W2 = H // 2, W // 2
  Wp, Hp = ops.relu_bits_pitch(W), ops.relu_bits_rows(H)
  ne = ops.relu_fields_elems(N, H2, W2)
  # reference: the two launches
  y1_ref = torch.full((G, N, H, W, 32), float('nan'), device=dev)
  bits_ref = torch.zeros(G, N, Hp, Wp, dtype=torch.int32, device=dev)
  y2_ref = torch.full((G, N, H2, W2, 48), float('nan'), device=dev)
  fields_ref = torch.zeros(G, ne, dtype=torch.int16, device=dev)
  if C == 3:
    ops.conv1_fwd_relu_bits_rgb_into(y1_ref, bits_ref, xd, w1d, b1d, G, xd[0].numel(), w1d[0].numel(), 32, y1_ref[0].numel(),
                                     bits_ref[0].numel(), N, H, W)
  else:
    ops.conv1_fwd_relu_bits_into(y1_ref, bits_ref, xd, w1d, b1d, G, xd[0].numel(), w1d[0].numel(), 32, y1_ref[0].numel(),
                                 bits_ref[0].numel(), N, H, W)
  ops.conv2_fwd_relu_fields_into(y2_ref, fields_ref, y1_ref, w2d, b2d, G, y1_ref[0].numel(), w2d[0].numel(), 48, y2_ref[0].numel(),
                                 ne, N, H, W)
  # the fused launch
  y1 = torch.full_like(y1_ref, float('nan'))
  y2 = torch.full_like(y2_ref, float('nan'))
  bits = torch.zeros_like(bits_ref) if side else None
  fields = torch.zeros_like(fields_ref) if side else None
  names = ops.kernel_trace(lambda: conv1_conv2_fwd_into(
      y1, bits, y2, fields, xd, w1d, b1d, w2d, b2d, G, xd[0].numel(), w1d[0].numel(), 32, y1[0].numel(),
      bits[0].numel() if side else 0, w2d[0].numel(), 48, y2[0].numel(), ne if side else 0, N, H, W, C))
  torch.cuda.synchronize()
  assert names == ['conv1_conv2_fwd_kernel<%s>' % ('true' if C == 3 else 'false')], names
  assert not torch.isnan(y1).any() and not torch.isnan(y2).any()
  assert torch.equal(y1, y1_ref)
  assert torch.equal(y2, y2_ref)
  if side:
    assert torch.equal(bits, bits_ref)
    assert torch.equal(fields, fields_ref)
  # ... and against the oracle (the separate kernels have their own oracle tests; this one must not depend on them being right)
  for g in range(G if H * W <= 136 * 136 else 1):
    n_chk = N if H * W <= 136 * 136 else 1
    ref1 = O.conv2d_same(torch.tensor(x3[g, :n_chk], dtype=torch.float64), torch.tensor(w1[g], dtype=torch.float64),
                         torch.tensor(b1[g], dtype=torch.float64), 1)
    ref2 = O.conv2d_same(ref1, torch.tensor(w2[g], dtype=torch.float64), torch.tensor(b2[g], dtype=torch.float64), 2)
    _close(y1[g, :n_chk], ref1.numpy(), 2e-5, 2e-5, 'fused forward y1, encoder %d' % g)
    _close(y2[g, :n_chk], ref2.numpy(), 2e-5, 2e-5, 'fused forward y2, encoder %d' % g)
