"""HIP path vs the committed golden vectors (fp64 oracle outputs; see tests/golden/make_golden.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as G  # noqa: E402


@pytest.mark.parametrize('case', list(G.CASES.keys()))
def test_hip_matches_golden(dev, case):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  want = json.load(open(os.path.join(HERE, 'golden', case + '.json')))
  ocfg, goal, P, feats, labels = G.build(case)
  model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(create_e2evmc_config(ocfg._asdict()), feats['rgb'].shape[0], dev, True)
  model.store.load_numpy(P)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  model.forward(backward_too=True)
  model.backward()
  torch.cuda.synchronize()
  assert abs(float(model.loss) - want['loss']) <= 1e-4 * abs(want['loss'])        # fp32 tolerance of north_star
  for k, v in model.predictions().items():
    np.testing.assert_allclose(v.cpu().numpy(), want['pred'][k], rtol=1e-4, atol=2e-5, err_msg=k)
  g = model.store.to_numpy('grads')
  for k, v in want['grads'].items():
    got = g[k].reshape(-1)[v['idx']]
    np.testing.assert_allclose(got, v['val'], rtol=0, atol=3e-4 * v['max_abs'], err_msg=k)
    assert abs(np.abs(g[k]).sum() - v['abs_sum']) <= 2e-3 * v['abs_sum'], k
  losses = []
  for _ in range(2):
    model.train_step()
    torch.cuda.synchronize()
    losses.append(float(model.loss))
  np.testing.assert_allclose(losses, want['train_losses'], rtol=1e-4)
  Pn = model.store.to_numpy('params')
  for k, v in want['params_after_2_steps'].items():
    np.testing.assert_allclose(Pn[k].reshape(-1)[v['idx']], v['val'], rtol=0, atol=2.5 * ocfg.lr, err_msg=k)
