"""Shared by the whole-graph GPU parity tests: the device's ReLU decisions as an oracle ``ReluTap``.

Gradient standard of every whole-graph test (small shapes and full size alike): the oracle's backward runs under the
DEVICE's ReLU decisions (the sign of every activation the device wrote, snapshotted BEFORE the device's backward so that no
buffer reuse can alias them) on the device's conv1 inputs; every gradient must then match to 2e-5 in max-norm AND in
relative L2, and every decision in which the device differs from the fp64 oracle must sit on a pre-activation
|z| <= 2e-5.  Loss, predictions and features are always compared against the PLAIN oracle (its own decisions).
"""
import numpy as np
import torch

from oracle import geeco_oracle as O

GRAD_TOL = 2e-5          # both norms, against the oracle under the device's ReLU decisions
Z_TOL = 2e-5             # |pre-activation| wherever the device's ReLU decision differs from the fp64 oracle's


def rel_max(a, b):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
  a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
  return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def snapshot_masks(enc):
  """The device's eight ReLU decisions per encoder as bool tensors, COPIED out of the activation buffers (call between
  ``model.forward`` and ``model.backward``).  -> masks[l][g] = bool [Nf, h, w, c]."""
  out = []
  for l in range(8):
    a = enc.acts[l]
    out.append([(a[g] > 0).clone() for g in range(enc.G)])
  torch.cuda.synchronize()
  return out


def call_slots(model, goal):
  """(scope, call) of the oracle's ``conv_encoder`` invocations -> (encoder g, first frame, frames) in the device's
  stacked layout (geeco_amd/graph.py: time-major frame slots of N frames).  Oracle call order: e2e_vmc graph.py:310-313;
  goal/sequence: the target frame FIRST (:354), then one call per time step (:362-381); goal/dynimg: one call per scope."""
  N, K, enc = model.N, model.K, model.enc
  slots = {}
  if not goal:
    for k in range(K):
      slots[(enc.scopes[0], k)] = (0, k * N, N)
  elif model.mode == 'dynimg':
    for g, sc in enumerate(enc.scopes):
      slots[(sc, 0)] = (g, 0, N)
  elif model.mode in ('seq_constant', 'seq_residual'):
    slots[(enc.scopes[0], 0)] = (0, K * N, N)
    for k in range(K):
      slots[(enc.scopes[0], 1 + k)] = (0, k * N, N)
  else:
    for g, sc in enumerate(enc.scopes):
      for k in range(K):
        slots[(sc, k)] = (g, k * N, N)
  return slots


def device_tap(model, goal, masks, C):
  """``ReluTap`` with the device's decisions (``masks`` from ``snapshot_masks``) and the device's conv1 inputs."""
  slots = call_slots(model, goal)
  enc = model.enc

  def masks_fn(scope, call):
    g, f0, n = slots[(scope, call)]
    return [masks[l][g][f0:f0 + n].cpu() for l in range(8)]

  def inputs_fn(scope, call):
    g, f0, n = slots[(scope, call)]
    return enc.x_in[g][f0:f0 + n][..., :C].cpu()

  return O.ReluTap(masks_fn, inputs_fn, force=True), slots


def check_decisions(stats, z_tol=Z_TOL):
  """(b): -> (differing, total, worst |z|); asserts every differing decision sits at |z| <= z_tol."""
  worst_z, n_dis, n_tot = 0.0, 0, 0
  for scope, st in stats.items():
    for l, (n, z, tot) in enumerate(st):
      assert tot > 0, (scope, l)
      assert z <= z_tol, '%s conv%d: ReLU decision differs at |z| = %.3e (%d of %d decisions differ)' % (scope, l + 1, z, n, tot)
      worst_z, n_dis, n_tot = max(worst_z, z), n_dis + n, n_tot + tot
  return n_dis, n_tot, worst_z


def check_gradients(grads, grads_ref, tol=GRAD_TOL):
  """(a): every variable, max-norm and relative L2 -> (achieved dict, worst (name, err)); asserts the bound."""
  achieved, failures, worst = {}, [], ('', 0.0)
  for k, g in grads_ref.items():
    g = g.numpy() if hasattr(g, 'numpy') else np.asarray(g)
    assert np.isfinite(grads[k]).all(), k
    e_max, e_l2 = rel_max(grads[k], g), rel_l2(grads[k], g)
    achieved[k] = {'max': float('%.3g' % e_max), 'l2': float('%.3g' % e_l2)}
    if e_max > tol or e_l2 > tol:
      failures.append((k, e_max, e_l2))
    if max(e_max, e_l2) > worst[1]:
      worst = (k, max(e_max, e_l2))
  assert not failures, failures
  return achieved, worst
