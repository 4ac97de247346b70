"""bench.py's measurement helpers on small models of every input path (the driver only ever runs the default
configuration, so the tables of the other BASELINE shapes are exercised here)."""
import argparse
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('model_name,channels', [('geeco-f', 3), ('geeco-f', 4), ('e2e_vmc', 3)])
def test_bench_tables_small(model_name, channels):
  import bench
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  dev = torch.device('cuda', 0)
  N, K, H = 2, 3, 136
  goal = model_name == 'geeco-f'
  kw = dict(window_size=K, img_channels=channels, batch_size=N, img_height=H, img_width=H)
  if goal:
    kw.update(proc_obs='dynimg', proc_tgt='dyndiff')
  cfg = create_e2evmc_config(kw)
  model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(cfg, N, dev, training=True)
  model.store.initialize(seed=0)
  bench.synthetic_batch(model, 1234)
  runner = TrainStepRunner(model, use_graph=False)
  runner.step()
  torch.cuda.synchronize()
  args = argparse.Namespace(model=model_name, channels=channels, seq_len=K, batch=N)
  hbm = bench.hbm_table(model, args, 1)
  assert hbm and hbm[-1]['piece'].startswith('adam') and all(r['us'] > 0 for r in hbm)
  if goal:
    assert len(hbm) == 4 and hbm[0]['piece'].startswith('dynimg buffer image') and hbm[2]['piece'].startswith('goal inputs as in the step')
  layers = bench.layer_table(model, 1)
  assert len(layers) >= 20 and all(r['us'] > 0 and r['kernel'] for r in layers)
  rl = bench.dominant_roofline(layers)
  assert rl['bound'] == 'mfma' and 0 < rl['frac'] < 1
