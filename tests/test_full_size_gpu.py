"""Parity at the FULL single-GPU shapes of BASELINE.json (not only at the reduced test shapes):

  config 2  geeco-f rgb   256x256  N=32 K=16   (the bench workload)
  config 4  e2e_vmc rgb   256x256  N=64 K=16   (1024 frame passes; conv1's output is exactly 2^31 floats)
  config 5  geeco-f rgbd  256x256  N=32 K=32   (per-GPU shape of the 8-GPU rgbd config)

One forward + backward of the HIP path vs the fp64 CPU oracle evaluated in chunks
(``oracle.loss_and_grads_chunked``: identical mathematics, bounded memory; checked against the plain oracle in
tests/test_oracle_kat.py).  Compared against the PLAIN oracle (its own ReLU decisions): loss and loss parts (1e-4
relative, BASELINE.json north_star), predictions, conv8's features of the FIRST and the LAST frame of every encoder (the
last frame is where a 32-bit offset overflow would corrupt data silently), the dynamic images.

Gradients -- every variable, two norms, NO fitted constant.  Round 2 bounded them by a max-norm cap (3e-3 / 5e-3 of
max |g|) that had been raised after a red run.  Round 3 first tried the yardstick of the small-shape tests (the same oracle
with an fp32 encoder; bound max(2e-4, 2 x its error)): it does not hold and cannot -- at these sizes a gradient's error is
set by a HANDFUL of ReLU decisions on pre-activations of 1e-8 that each fp32 implementation rounds to its own side of zero
(measured, tests/golden/full_size_plain_oracle_r03.json: DynDiffEncoder conv4-6 differ from the fp64 oracle by the SAME
1.26e-4 / 1.36e-4 / 5.80e-4 on the device and in the fp32 CPU run, same flip; ConvEncoder/conv3 4.3e-4 on the device and
1e-6 on the CPU, a flip only the device made), so the ratio of two such errors is noise.  The test therefore removes the
effect instead of bounding it: the oracle's backward runs under the DEVICE's ReLU decisions (``masks_fn``: the sign of
every activation the device wrote) on the device's conv1 inputs, and
  (a) every gradient equals that oracle's to 2e-5 of max |g| AND to 2e-5 in relative L2 (the tolerance of the per-kernel
      tests; a wrong tile edge, a mis-written slice or a systematic error in small entries cannot hide under it), and
  (b) every ReLU decision in which the device differs from the fp64 oracle sits on a pre-activation |z| <= 2e-5 -- the
      device never "decides" anything the forward tolerance does not already allow; the count per layer is printed.
  (c) (round 4, mask-INDEPENDENT backstops) the device's conv8 features of EVERY frame equal the plain oracle's (2e-4 of
      the maximum: a forward error that changes magnitudes but not signs cannot hide in the frames between the first and
      the last); the decisions are copied out of the activation buffers BEFORE the device's backward runs (no buffer the
      backward reuses can alias them); EVERY config (round 6; rounds 4-5: config 2 only, for the suite's wall time -- measured now:
      + 9 s for config 5, + 25 s for config 4): every gradient also within PLAIN_TOL = 5e-3 of max |g| of the oracle under its OWN decisions (loose by
      necessity: a handful of rounding-level flips move a filter gradient by up to 1e-3, the measured worst being 5.8e-4,
      tests/golden/full_size_plain_oracle_r03.json); configs 4 (oracle encoder in fp32 for time) and 5: the first, the two
      middle and the last frame of encoder 0 (config 4: 0, 511, 512, 1023 -- around the 2^31-element boundary) additionally
      through a single-frame fp64 encoder -- features 2e-5, decisions as (b).
The achieved errors of every variable go to gpurun_out/full_size_achieved_<config>.json; the committed copy is
tests/golden/full_size_achieved.json.  Mask-independent elementwise checks of every launch at these shapes:
tests/test_bench_shapes_gpu.py.  The oracle appends a line per chunk to gpurun_out/full_size_progress.log (signs of life).
"""
import json
import os
import time

import numpy as np
import pytest
import torch

from oracle import geeco_oracle as O
import _relu_taps as T
from _relu_taps import GRAD_TOL, Z_TOL

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAIN_TOL = 5e-3         # max-norm, against the fp64 oracle under its OWN decisions (backstop (c); config 2)

# Oracle precision: fp64 throughout for configs 2 and 5.  The 1024-frame config 4 runs the oracle's ENCODER in fp32 (decoder
# and loss in fp64): in fp64 it takes 286 s on the GPU box's 16 host threads (measured; the achieved errors of that run are
# the committed tests/golden/full_size_achieved.json: worst 1.48e-6), too close to the harness's silence limit.  Under the
# device's ReLU decisions an fp32 oracle has no flips either, so the same 2e-5 bound holds (its own rounding is ~5e-7).
#        name               cfg overrides                                                     goal  N   oracle encoder
FULL = [
    ('config2 geeco-f rgb N=32 K=16', dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=16), True, 32, torch.float64),
    ('config4 e2e_vmc rgb N=64 K=16', dict(window_size=16), False, 64, torch.float32),
    ('config5 geeco-f rgbd N=32 K=32', dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=32, img_channels=4), True, 32,
     torch.float64),
]


def _rel_max(a, b):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _rel_l2(a, b):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


@pytest.mark.parametrize('name,cfg_kw,goal,N,enc_dtype', FULL, ids=[c[0].split()[0] for c in FULL])
def test_full_size_forward_backward(dev, name, cfg_kw, goal, N, enc_dtype):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  cfg_kw = dict(cfg_kw, batch_size=N)
  ocfg = O.make_config(**cfg_kw)
  P = O.init_params(O.model_param_shapes(ocfg, goal), seed=21)
  r = np.random.default_rng(22)
  for k in P:
    if k.endswith('/bias'):
      P[k] = (0.05 * r.standard_normal(P[k].shape)).astype(np.float32)
  feats, labels = O.synthetic_batch(ocfg, goal, N, seed=23)

  t0 = time.time()
  model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(create_e2evmc_config(ocfg._asdict()), N, dev, training=True)
  model.store.load_numpy(P)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  model.forward(backward_too=True)
  torch.cuda.synchronize()
  enc, C = model.enc, ocfg.img_channels
  masks = T.snapshot_masks(enc)             # bool copies, taken BEFORE the backward (0.3 / 3.4 GB of HBM at 96 / 1024 frames)
  f8 = enc.features.cpu().numpy()           # [G][Nf][2][2][C]
  model.backward()
  torch.cuda.synchronize()
  t_hip = time.time() - t0

  # ---- the oracle: plain forward (its own ReLU decisions), backward under the device's decisions ----------------------
  enc_inputs = [enc.x_in[g][..., :C].cpu() for g in range(enc.G)]           # what the device fed its conv1 (fp32)
  masks_fn = lambda g, i0, i1: [masks[l][g][i0:i1].cpu() for l in range(8)]
  out_dir = os.path.join(ROOT, 'gpurun_out')
  os.makedirs(out_dir, exist_ok=True)
  t0 = time.time()

  def progress(text):
    with open(os.path.join(out_dir, 'full_size_progress.log'), 'a') as f:
      f.write('%s %6.1f s %s\n' % (name.split()[0], time.time() - t0, text))

  plain_backstop = True       # the second full backward, under the oracle's OWN decisions (round 6: every config; rounds 4-5: config 2 only)
  fp64_frames = not name.startswith('config2')      # configs 4 and 5 additionally: single frames through the fp64 encoder
  tr = O.OracleTrainer(ocfg, goal, P, dtype=torch.float64)
  loss_ref, parts_ref, grads_ref, pred_ref, ep_ref = O.loss_and_grads_chunked(
      tr, feats, labels, chunk=32, enc_dtype=enc_dtype, encoder_inputs=enc_inputs, masks_fn=masks_fn, plain_grads=plain_backstop,
      progress=progress)
  t_ora = time.time() - t0

  # ---- (c) conv8 features of EVERY frame of every encoder against the plain oracle ------------------------------
  for g, scope in enumerate(enc.scopes):
    ref8 = ep_ref['conv8'][scope].numpy()
    assert ref8.shape == f8[g].shape, (scope, ref8.shape, f8[g].shape)
    scale = max(float(np.abs(ref8).max()), 1e-30)
    err = np.abs(f8[g] - ref8).reshape(ref8.shape[0], -1).max(axis=1)
    assert err.max() <= 2e-4 * scale, (scope, 'frame %d' % int(err.argmax()), float(err.max()), scale)
  fp64_subset = None
  if fp64_frames:          # configs 4 and 5: the frames at the ends and around the 2^31-element boundary through the fp64 encoder
    Pe = {k: v for k, v in tr.P.items() if '/conv' in k}
    fp64_subset = {}
    for fr in sorted({0, enc.Nf // 2 - 1, enc.Nf // 2, enc.Nf - 1}):
      st = [[0, 0.0, 0] for _ in range(8)]
      with torch.no_grad():
        out = O.conv_encoder(enc_inputs[0][fr:fr + 1].to(torch.float64), {k: v.to(torch.float64) for k, v in Pe.items() if k.startswith(enc.scopes[0] + '/')},
                             enc.scopes[0], masks=masks_fn(0, fr, fr + 1), stats=st)
      e = float(np.abs(f8[0][fr] - out[0].numpy()).max() / max(float(out.abs().max()), 1e-30))
      assert e <= 2e-5, ('fp64 encoder, frame %d' % fr, e)
      T.check_decisions({'frame %d' % fr: st})
      fp64_subset[fr] = {'conv8_err': float('%.3g' % e), 'decisions_differing': int(sum(s[0] for s in st)),
                         'max_abs_preactivation_where_differing': float('%.3g' % max(s[1] for s in st))}
      progress('fp64 encoder of frame %d' % fr)
  if goal:      # the device's dynamic images (fed to the oracle's encoders above) against the oracle's own fp64 ones
    ep = model.endpoints()
    for k in ('dynbuff', 'dyndiff'):
      assert _rel_max(ep[k].cpu().numpy(), ep_ref[k].numpy()) < 2e-5, k

  # ---- predictions, loss ------------------------------------------------------------------------------
  preds = {k: v.cpu().numpy() for k, v in model.predictions().items()}
  for k, v in pred_ref.items():
    np.testing.assert_allclose(preds[k], v.numpy(), rtol=2e-4, atol=5e-5, err_msg=k)
  parts = {k: float(v) for k, v in model.loss_parts().items()}
  assert abs(parts['loss'] - float(loss_ref)) <= 1e-4 * abs(float(loss_ref)), (parts['loss'], float(loss_ref))
  for k, v in parts_ref.items():
    if k != 'loss_reg':
      assert abs(parts[k] - float(v)) <= 1e-4 * abs(float(v)) + 1e-7, (k, parts[k], float(v))

  # ---- (b) the device's ReLU decisions: different from the fp64 oracle's only at rounding-level pre-activations ----------
  dis = ep_ref['relu_disagreements']
  n_dis, n_tot, worst_z = T.check_decisions(dis)

  # ---- (a) every variable's gradient, max-norm and relative L2; (c) the loose bound against the oracle's own decisions ---------
  grads = model.store.to_numpy('grads')
  achieved, failures = {}, []
  worst = ('', 0.0)
  worst_plain = ('', 0.0)
  for k, g in grads_ref.items():
    g = g.numpy()
    assert np.isfinite(grads[k]).all(), k
    e_max, e_l2 = _rel_max(grads[k], g), _rel_l2(grads[k], g)
    achieved[k] = {'max': float('%.3g' % e_max), 'l2': float('%.3g' % e_l2)}
    if e_max > GRAD_TOL or e_l2 > GRAD_TOL:
      failures.append((k, e_max, e_l2))
    if max(e_max, e_l2) > worst[1]:
      worst = (k, max(e_max, e_l2))
    gp = ep_ref.get('plain_grads', {}).get(k)
    if gp is not None:
      e_p = _rel_max(grads[k], gp.numpy())
      achieved[k]['plain_oracle_max'] = float('%.3g' % e_p)
      if e_p > PLAIN_TOL:
        failures.append((k, 'plain oracle', e_p))
      if e_p > worst_plain[1]:
        worst_plain = (k, e_p)
  with open(os.path.join(out_dir, 'full_size_achieved_%s.json' % name.split()[0]), 'w') as f:
    json.dump({'case': name, 'bound': '%g of max|g| and %g relative L2, against the fp64 oracle under the device\'s ReLU decisions' % (GRAD_TOL, GRAD_TOL),
               'oracle_encoder_dtype': str(enc_dtype).replace('torch.', ''),
               'backstop': 'plain oracle (own decisions; encoder in %s), %g of max|g|: worst %.3g at %s' %
                           (str(enc_dtype).replace('torch.', ''), PLAIN_TOL, worst_plain[1], worst_plain[0]) +
                           ('; fp64 encoder of frames %s' % sorted(fp64_subset) if fp64_subset else ''),
               'fp64_frame_subset': fp64_subset,
               'relu_decisions': {'differing': n_dis, 'total': n_tot, 'max_abs_preactivation_where_differing': float('%.3g' % worst_z),
                                  'per_layer': {s: [[n, float('%.3g' % z), t] for n, z, t in st] for s, st in dis.items()}},
               'variables': achieved}, f, indent=1)
  print('%s: loss %.6f (oracle %.6f); worst gradient error %.2e (max-norm or rel. L2, bound %.0e) at %s; %d of %d ReLU decisions '
        'differ from the fp64 oracle, all at |z| <= %.1e; plain-oracle backstop %.2e at %s; hip %.1f s, oracle %.1f s'
        % (name, parts['loss'], float(loss_ref), worst[1], GRAD_TOL, worst[0], n_dis, n_tot, worst_z, worst_plain[1], worst_plain[0],
           t_hip, t_ora))
  assert not failures, failures
