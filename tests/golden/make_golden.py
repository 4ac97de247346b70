"""Generates tests/golden/*.json from the fp64 CPU oracle (oracle/geeco_oracle.py).

These are SELF-golden vectors: the reference has no tests or fixtures and TensorFlow 1.15 cannot
run here, so the expected values come from this repository's restatement of the reference, not
from the reference itself ("parity unpinned").  Inputs and weights are regenerated
from seeds (numpy default_rng); only outputs are stored.

  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import geeco_oracle as O  # noqa: E402

CASES = {
    'geeco_f_rgb_k4_136': dict(goal=True, N=2, H=136, seed=21, cfg=dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=4)),
    'geeco_f_rgbd_k3_144': dict(goal=True, N=2, H=144, seed=22,
                                cfg=dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_channels=4, lambda_aux=0.25)),
    'e2e_vmc_rgb_k3_136': dict(goal=False, N=2, H=136, seed=23, cfg=dict(window_size=3)),
}
PROBES = 6   # elements sampled per variable


def build(case):
  c = CASES[case]
  cfg = O.make_config(img_height=c['H'], img_width=c['H'], batch_size=c['N'], **c['cfg'])
  P = O.init_params(O.model_param_shapes(cfg, c['goal']), seed=c['seed'])
  r = np.random.default_rng(c['seed'] + 1)
  for k in P:
    if k.endswith('/bias'):
      P[k] = (0.05 * r.standard_normal(P[k].shape)).astype(np.float32)
  feats, labels = O.synthetic_batch(cfg, c['goal'], c['N'], seed=c['seed'] + 2, H=c['H'], W=c['H'])
  return cfg, c['goal'], P, feats, labels


def probe_indices(shape, k):
  n = int(np.prod(shape))
  return [int(i) for i in np.linspace(0, n - 1, num=min(PROBES, n)).astype(np.int64)]


def run(case):
  cfg, goal, P, feats, labels = build(case)
  tr = O.OracleTrainer(cfg, goal, P, dtype=torch.float64)
  loss, parts, grads, pred, ep = tr.loss_and_grads(feats, labels)
  out = {'case': case, 'loss': float(loss), 'parts': {k: float(v) for k, v in parts.items()},
         'pred': {k: v.numpy().tolist() for k, v in pred.items()}, 'grads': {}, 'params_after_2_steps': {}}
  for k, g in grads.items():
    g = g.numpy()
    idx = probe_indices(g.shape, k)
    out['grads'][k] = {'sum': float(g.sum()), 'abs_sum': float(np.abs(g).sum()), 'max_abs': float(np.abs(g).max()),
                       'idx': idx, 'val': [float(g.reshape(-1)[i]) for i in idx]}
  if goal:
    out['dynbuff_mean'] = float(ep['dynbuff'].mean())
    out['dyndiff_mean'] = float(ep['dyndiff'].mean())
  losses = []
  for _ in range(2):
    l, _p = tr.train_step(feats, labels)
    losses.append(l)
  out['train_losses'] = losses
  for k, v in tr.P.items():
    v = v.numpy()
    idx = probe_indices(v.shape, k)
    out['params_after_2_steps'][k] = {'idx': idx, 'val': [float(v.reshape(-1)[i]) for i in idx]}
  return out


if __name__ == '__main__':
  here = os.path.dirname(os.path.abspath(__file__))
  for case in CASES:
    res = run(case)
    with open(os.path.join(here, case + '.json'), 'w') as f:
      json.dump(res, f, indent=1)
    print(case, 'loss', res['loss'])
