"""Elementwise parity of EVERY conv launch of the train step at the launch geometries the bench runs
(BASELINE.json configs[1]: geeco-f rgb, 3 encoders x 32 frames of 256x256 = 96 frame passes; 255-block
persistent kernels, split-K factors, XCD dealing, grouped arena strides), not only at the reduced shapes of
tests/test_kernels_gpu.py.

Every launch goes through the model's own ``ConvEncoderStack.launch_fwd / launch_dgrad / launch_wgrad`` (the
calls the captured step replays, C-ABI underneath) and is compared ELEMENTWISE with the fp64 CPU oracle
(``oracle.conv2d_same`` and its autograd; graph.py:61-117, estimator.py:243-244) on the launch's own inputs:

  forward   conv1..conv8: y = relu(conv(x) + b); frames 0, 1, 31, 32, 47, 63, 64, 95 of the 96 (first / last frame
            of every encoder and the verdict's 0 / 47 / 95); rtol 2e-5 + atol 2e-5
  dgrad     conv8..conv3 (and conv2 inside the fused bottom): dx = conv_dgrad(dz) * (y_below > 0) with the mask the
            step really uses (the activation tensor, conv1's sign words, conv2's / conv3's sign fields -- all written
            by the forward launches above); same frames; rtol 2e-5 + atol 2e-5
  wgrad     conv8..conv2 and the fused bottom's conv1: dw, db over ALL 96 frames (oracle in chunks);
            rtol 2e-5 + atol 2e-5 * sqrt(#summed pixels)

The masks are taken from the DEVICE activations (sign of what the forward launch wrote), so no ReLU decision of
the comparison depends on rounding: every check is mask-independent and a wrong tile edge shows up as a wrong
element.  conv1's forward is also checked at the 1024-frame size of config 4 (exactly 2^31 output floats: frames 0,
511, 512, 1023 sit on either side of the 2^32-byte and at the end of the 2^31-element offsets) and the RGB-D
bottom (conv1 forward / fused bottom with 4 real channels) at the per-GPU shape of config 5.
"""
import time

import numpy as np
import pytest
import torch

from oracle import geeco_oracle as O

pytestmark = pytest.mark.gpu

FRAMES = (0, 1, 31, 32, 47, 63, 64, 95)      # flat index g * 32 + n


def _cmp(got, ref, rtol, atol, what):
  got = np.asarray(got, np.float64)
  ref = np.asarray(ref, np.float64)
  assert got.shape == ref.shape, (what, got.shape, ref.shape)
  assert np.isfinite(got).all(), what
  err = np.abs(got - ref)
  tol = atol + rtol * np.abs(ref)
  bad = err > tol
  assert not bad.any(), '%s: %d of %d elements off, worst err %.3e (tol %.3e) at %s' % (
      what, int(bad.sum()), bad.size, float(err.max()), float(tol.flat[err.argmax()]),
      np.unravel_index(err.argmax(), err.shape))
  return float((err / tol).max())


def _he_params(store, seed):
  """He-scaled kernels (the ReLU chain keeps O(1) activations through all eight layers, so the absolute part of
  the tolerances means the same at every layer) and non-zero biases."""
  r = np.random.default_rng(seed)
  P = {}
  for name, shp in store.shapes.items():
    if name.endswith('/bias'):
      P[name] = (0.1 * r.standard_normal(shp)).astype(np.float32)
    elif len(shp) == 4:
      P[name] = (r.standard_normal(shp) * np.sqrt(2.0 / (9 * shp[2]))).astype(np.float32)
    else:
      P[name] = (r.standard_normal(shp) / np.sqrt(shp[0])).astype(np.float32)
  return P


def _build(dev, channels, K, N=32, goal=True):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  kw = dict(window_size=K, img_channels=channels, batch_size=N)
  if goal:
    kw.update(proc_obs='dynimg', proc_tgt='dyndiff')
  model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(create_e2evmc_config(kw), N, dev, training=True)
  model.store.load_numpy(_he_params(model.store, 5))
  model.enc.refresh_derived()
  return model


def _fill_inputs(enc, seed):
  g = torch.Generator(device=enc.x_in.device)
  g.manual_seed(seed)
  enc.x_in.normal_(generator=g)
  if enc.Cpad != enc.Cin:
    enc.x_in[..., enc.Cin:] = 0.0


def _w64(enc, l, g):
  return enc._w(l, g).detach().cpu().double(), enc._b(l, g).detach().cpu().double()


def _frame(t, f, Nf):
  return t[f // Nf, f % Nf].detach().cpu()


def _check_fwd(enc, l, frames):
  L = enc.layers[l]
  x = enc.x_in if l == 0 else enc.acts[l - 1]
  worst = 0.0
  for f in frames:
    g = f // enc.Nf
    w, b = _w64(enc, l, g)
    xf = _frame(x, f, enc.Nf).double()[None]
    if l == 0:
      xf = xf[..., :enc.Cin]
    ref = O.conv2d_same(xf, w, b, L['stride'], relu=True)[0]
    worst = max(worst, _cmp(_frame(enc.acts[l], f, enc.Nf).numpy(), ref.numpy(), 2e-5, 2e-5, 'conv%d fwd, frame %d' % (l + 1, f)))
  return worst


def _oracle_dgrad(dz, w, in_hw, cin, stride):
  x = torch.zeros((dz.shape[0],) + tuple(in_hw) + (cin,), dtype=torch.float64, requires_grad=True)
  y = O.conv2d_same(x, w, torch.zeros(w.shape[3], dtype=torch.float64), stride, relu=False)
  y.backward(dz)
  return x.grad


def _check_dgrad(enc, l, frames):
  """dz[l-1] (written by launch_dgrad(l)) against the oracle, masked by the sign of the device's acts[l-1]."""
  L = enc.layers[l]
  worst = 0.0
  for f in frames:
    g = f // enc.Nf
    w, _ = _w64(enc, l, g)
    dz = _frame(enc.dz[l], f, enc.Nf).double()[None]
    ref = _oracle_dgrad(dz, w, (L['H'], L['W']), L['Cin'], L['stride'])[0]
    ref = ref * (_frame(enc.acts[l - 1], f, enc.Nf) > 0)
    worst = max(worst, _cmp(_frame(enc.dz[l - 1], f, enc.Nf).numpy(), ref.numpy(), 2e-5, 2e-5, 'conv%d dgrad, frame %d' % (l + 1, f)))
  return worst


def _oracle_wgrad(x, dz, cin, cout, stride, chunk=8):
  """Sum over all frames of d(conv)/d(kernel, bias) in fp64, `chunk` frames at a time.  x / dz: device tensors
  [Nf][H][W][C] / [Nf][Ho][Wo][Cout] (x's pad channels beyond `cin` are dropped)."""
  w = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
  b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
  for i in range(0, x.shape[0], chunk):
    xc = x[i:i + chunk, ..., :cin].cpu().double()
    y = O.conv2d_same(xc, w, b, stride, relu=False)
    y.backward(dz[i:i + chunk].cpu().double())
  return w.grad, b.grad


def _check_wgrad(enc, l):
  L = enc.layers[l]
  x = enc.x_in if l == 0 else enc.acts[l - 1]
  cin = enc.Cin if l == 0 else L['Cin']
  scale = np.sqrt(enc.Nf * L['Ho'] * L['Wo'])
  worst = 0.0
  for g in range(enc.G):
    dw_ref, db_ref = _oracle_wgrad(x[g], enc.dz[l][g], cin, L['Cout'], L['stride'])
    worst = max(worst, _cmp(enc._dw(l, g).cpu().numpy(), dw_ref.numpy(), 2e-5, 2e-5 * scale, 'conv%d wgrad, encoder %d' % (l + 1, g)))
    worst = max(worst, _cmp(enc._db(l, g).cpu().numpy(), db_ref.numpy(), 2e-5, 2e-5 * scale, 'conv%d bias grad, encoder %d' % (l + 1, g)))
  return worst


def _check_fused_bottom(enc, dz1_frames=()):
  """conv1's dw / db out of the fused bottom (conv2 dgrad -> ReluGrad by conv1's sign words -> conv1 wgrad, dz1 on
  chip) over all frames: oracle dz1 = conv2_dgrad(dz2) * (y1 > 0) with the device's y1, then conv1's wgrad."""
  L0, L1 = enc.layers[0], enc.layers[1]
  scale = np.sqrt(enc.Nf * L0['Ho'] * L0['Wo'])
  worst = 0.0
  for g in range(enc.G):
    w2, _ = _w64(enc, 1, g)
    w = torch.zeros(3, 3, enc.Cin, L0['Cout'], dtype=torch.float64, requires_grad=True)
    b = torch.zeros(L0['Cout'], dtype=torch.float64, requires_grad=True)
    for i in range(0, enc.Nf, 4):
      dz2 = enc.dz[1][g, i:i + 4].cpu().double()
      dz1 = _oracle_dgrad(dz2, w2, (L1['H'], L1['W']), L1['Cin'], 2) * (enc.acts[0][g, i:i + 4].cpu() > 0)
      y = O.conv2d_same(enc.x_in[g, i:i + 4, ..., :enc.Cin].cpu().double(), w, b, 1, relu=False)
      y.backward(dz1)
    worst = max(worst, _cmp(enc._dw(0, g).cpu().numpy(), w.grad.numpy(), 2e-5, 2e-5 * scale, 'fused bottom dw1, encoder %d' % g))
    worst = max(worst, _cmp(enc._db(0, g).cpu().numpy(), b.grad.numpy(), 2e-5, 2e-5 * scale, 'fused bottom db1, encoder %d' % g))
  return worst


def _randomize(t, seed):
  g = torch.Generator(device=t.device)
  g.manual_seed(seed)
  t.normal_(generator=g)


def test_config2_every_conv_launch_elementwise(dev):
  """geeco-f rgb N=32 K=16: 3 encoders x 32 frames, the bench workload."""
  from geeco_amd import ops
  t0 = time.time()
  model = _build(dev, 3, 16)
  enc = model.enc
  assert (enc.G, enc.Nf) == (3, 32) and enc.fused_bottom and enc.relu_bits and enc.relu_fields and enc.relu_fields3
  _fill_inputs(enc, 71)
  report = []
  # ---- forward chain: every layer consumes the device output of the layer below ------------------------------
  kernels = {}
  for l in range(8):
    kernels['conv%d fwd' % (l + 1)] = ops.kernel_trace(lambda l=l: enc.launch_fwd(l))
  torch.cuda.synchronize()
  for l in range(8):
    report.append(('conv%d fwd' % (l + 1), _check_fwd(enc, l, FRAMES)))
  # ---- backward: fresh random dz per layer (independent of the chain), masks from the forward above -----------
  for l in range(7, 0, -1):
    _randomize(enc.dz[l], 100 + l)
    pending = []
    kernels['conv%d wgrad' % (l + 1)] = ops.kernel_trace(lambda l=l: enc.launch_wgrad(l, pending))
    kernels['conv%d dgrad' % (l + 1)] = ops.kernel_trace(lambda l=l: enc.launch_dgrad(l, pending))
    if pending:
      ops.slab_reduce_batch(pending)
    torch.cuda.synchronize()
    report.append(('conv%d wgrad' % (l + 1), _check_wgrad(enc, l)))
    if l >= 2:
      report.append(('conv%d dgrad' % (l + 1), _check_dgrad(enc, l, FRAMES)))
    else:
      report.append(('conv2 dgrad + conv1 wgrad (fused bottom)', _check_fused_bottom(enc)))
  # the launches under test are the ones the bench step runs
  assert kernels['conv2 dgrad'][0].startswith('conv2_dgrad_conv1_wgrad_kernel<3, true>'), kernels['conv2 dgrad']
  assert kernels['conv2 fwd'][0].startswith('conv_s2_halo_fwd_ws_kernel'), kernels['conv2 fwd']
  assert kernels['conv3 dgrad'][0].startswith('conv_s2_halo_dgrad_chunked_kernel'), kernels['conv3 dgrad']
  for l in (3, 4, 5, 6):
    assert kernels['conv%d wgrad' % l][0].startswith('conv_s2_wgrad_lds_kernel'), (l, kernels['conv%d wgrad' % l])
  for l in (4, 5, 6):
    assert kernels['conv%d dgrad' % l][0].startswith('conv_s2_dgrad_lds_kernel'), (l, kernels['conv%d dgrad' % l])
  print('config 2 launch shapes, worst err / tol per launch (%.0f s):' % (time.time() - t0))
  for name, w in report:
    print('  %-44s %.3f  %s' % (name, w, kernels.get(name, [''])[0] if name in kernels else ''))


def test_config5_rgbd_bottom_elementwise(dev):
  """geeco-f rgbd N=32 K=32 (per-GPU shape of config 5): only the bottom differs from config 2 (4 real input channels:
  conv1's forward reads the 4-channel kernel, fused bottom <4, true>); conv2.. are the same launches as above."""
  from geeco_amd import ops
  model = _build(dev, 4, 32)
  enc = model.enc
  assert enc.fused_bottom and enc.relu_bits and enc.Cin == 4
  _fill_inputs(enc, 73)
  enc.launch_fwd(0)
  torch.cuda.synchronize()
  w0 = _check_fwd(enc, 0, FRAMES)
  _randomize(enc.dz[1], 9)
  pending = []
  names = ops.kernel_trace(lambda: enc.launch_dgrad(1, pending))
  if pending:
    ops.slab_reduce_batch(pending)
  torch.cuda.synchronize()
  assert names[0].startswith('conv2_dgrad_conv1_wgrad_kernel<4, true>'), names
  w1 = _check_fused_bottom(enc)
  print('config 5 bottom: conv1 fwd %.3f, fused bottom %.3f (worst err / tol)' % (w0, w1))


def test_config4_conv1_forward_2pow31_outputs(dev):
  """e2e_vmc N=64 K=16: 1024 frames through conv1 = exactly 2^31 output floats (8 GiB).  Frames on either side of the
  2^32-byte offset (511 | 512) and the last one (ends at element 2^31) against the oracle, plus their sign words."""
  model = _build(dev, 3, 16, N=64, goal=False)
  enc = model.enc
  assert enc.G * enc.Nf == 1024 and enc.acts[0].numel() == 2 ** 31
  _fill_inputs(enc, 79)
  enc.acts[0].fill_(float('nan'))
  enc.launch_fwd(0)
  torch.cuda.synchronize()
  worst = _check_fwd(enc, 0, (0, 1, 511, 512, 513, 1022, 1023))
  if enc.relu_bits:
    c = np.arange(32)
    weights = (1 << ((c & 3) * 8 + (c >> 2))).astype(np.int64)
    H, W = enc.layers[0]['H'], enc.layers[0]['W']
    for f in (0, 511, 512, 1023):
      y = enc.acts[0][0, f].cpu().numpy()
      want = ((y > 0).astype(np.int64) * weights).sum(-1).astype(np.uint32)
      got = enc.bits1[0, f].cpu().numpy().view(np.uint32)[:H, :W]
      assert np.array_equal(got, want), 'sign words of frame %d' % f
  # nothing left unwritten anywhere in the 8 GiB (a dropped tile would still hold NaN)
  assert not bool(torch.isnan(enc.acts[0].view(-1)[::4099]).any())
  assert not bool(torch.isnan(enc.acts[0][0, 1023]).any()) and not bool(torch.isnan(enc.acts[0][0, 512]).any())
  print('config 4 conv1 forward at 1024 frames: worst err / tol %.3f' % worst)
