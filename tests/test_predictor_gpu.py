"""Predictor (batch-1 inference) vs the oracle's forward on the same frame sequence."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import geeco_oracle as O

pytestmark = pytest.mark.gpu


def _make_model_dir(tmp_path, goal, kw):
  from geeco_amd import estimator as est
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  cfg = create_e2evmc_config(kw)
  json.dump(cfg._asdict(), open(os.path.join(tmp_path, 'e2evmc_config.json'), 'w'))
  st = VariableStore(model_variable_shapes(cfg, goal), 'cpu')
  st.initialize(seed=4)
  st.global_step.fill_(7)
  est.save_checkpoint(st, str(tmp_path), keep_max=2)
  return cfg, st.to_numpy('params')


@pytest.mark.parametrize('goal', [True, False])
def test_predictor_matches_oracle(dev, tmp_path, goal):
  from geeco_amd.predictor import E2EVMCPredictor, GoalE2EVMCPredictor
  kw = dict(window_size=3, img_height=136, img_width=136)
  if goal:
    kw.update(proc_obs='dynimg', proc_tgt='dyndiff')
  cfg, P = _make_model_dir(tmp_path, goal, kw)
  pred = (GoalE2EVMCPredictor if goal else E2EVMCPredictor)(str(tmp_path))
  assert pred.cfg.batch_size == 1
  r = np.random.default_rng(2)
  frames = r.random([5, 136, 136, 3], dtype=np.float32)
  jnts = r.standard_normal([5, 7]).astype(np.float32)
  tgt = r.random([136, 136, 4], dtype=np.float32)        # extra channel must be cut off by set_goal
  if goal:
    pred.set_goal(tgt)
  ocfg = O.make_config(batch_size=1, **kw)
  Pt = {k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()}
  window = []
  for t in range(5):
    out = pred.predict(frames[t], jnts[t])
    window.append((frames[t], jnts[t]))
    while len(window) < 3:
      window.append((frames[t], jnts[t]))                 # first-frame padding after reset
    window = window[-3:]
    feats = {'rgb': torch.tensor(np.stack([w[0] for w in window])[None], dtype=torch.float64),
             'jnt_state': torch.tensor(np.stack([w[1] for w in window])[None], dtype=torch.float64)}
    if goal:
      feats['target_rgb'] = torch.tensor(tgt[None, :, :, :3], dtype=torch.float64)
    ref, ep = O.model_forward(feats, Pt, ocfg, goal)
    # t == 0 with a goal model: the padded window holds ONE frame K times, so the dynamic image is
    # sum(alpha) * frame = rounding noise divided by (noise range + 1e-6): ill-conditioned in any
    # precision (also in the TF reference); only shapes are checked there.
    noise = goal and t == 0
    for k in ('cmd_ee', 'pos_ee', 'pos_obj'):
      np.testing.assert_allclose(out[k], ref[k][0].numpy(), rtol=1e-4, atol=5e-2 if noise else 2e-5,
                                 err_msg='%s @%d' % (k, t))
    assert out['cmd_grp'].shape == (1,)
    if not noise:
      assert out['cmd_grp'][0] == float(int(ref['logits_cmd_grp'][0].argmax()) - 1)
    if goal:
      assert out['dynbuff'].shape == (136, 136, 3) and out['dyndiff'].shape == (136, 136, 3)
      if not noise:
        np.testing.assert_allclose(out['dynbuff'], ep['dynbuff'][0].numpy(), atol=1e-5)
      np.testing.assert_allclose(out['dyndiff'], ep['dyndiff'][0].numpy(), atol=1e-5)
  pred.reset()
  out2 = pred.predict(frames[0], jnts[0])
  out_first = None
  # after reset the buffer is padded with the fed frame again: same result as the very first call
  pred.reset()
  out_first = pred.predict(frames[0], jnts[0])
  for k in ('cmd_ee', 'pos_ee', 'pos_obj'):
    np.testing.assert_array_equal(out2[k], out_first[k])
  with pytest.raises(AssertionError):
    pred.predict(frames[0] * 3.0, jnts[0])                # range check (predictor.py:135-138)
  with pytest.raises(AssertionError):
    pred.predict(frames[0][:100], jnts[0])                # shape check


def test_tf_bundle_end_to_end(dev, tmp_path):
  """f3 closed on the GPU: the Estimator trains two steps and writes its checkpoint ALSO as a TF-1.15 tensor bundle
  (model.ckpt-2.index / .data-00000-of-00001: reference variable names, Adam slots, global_step, lstm_memory); a fresh
  GoalE2EVMCPredictor that finds only the bundle predicts bit-identically to one restored from the native file, and an
  Estimator resumes from the bundle with the same evaluation and optimiser state (predictor.py:85-95,
  train_e2evmc.py:160-181)."""
  import shutil
  from geeco_amd import estimator as est
  from geeco_amd import tf_checkpoint
  from geeco_amd.input_fn import synthetic_batches
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.predictor import GoalE2EVMCPredictor
  kw = dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4, lr=1e-3)
  cfg = create_e2evmc_config(kw)
  params = {'e2evmc_config': cfg, 'log_steps': 100, 'debug': False}
  md = str(tmp_path / 'run')
  e = est.Estimator(est.goal_e2evmc_model_fn, md, est.RunConfig(save_tf_bundle=True), params)
  e.train(input_fn=synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=5))
  json.dump(cfg._asdict(), open(os.path.join(md, 'e2evmc_config.json'), 'w'))
  prefix = est.latest_checkpoint(md)
  assert os.path.basename(prefix) == 'model.ckpt-2'
  assert all(os.path.exists(prefix + sfx) for sfx in ('.pt', '.index', '.data-00000-of-00001'))
  t = tf_checkpoint.read_checkpoint(prefix)
  assert int(t['global_step']) == 2 and t['GoalVMC/LSTMDecoder/lstm_memory'].shape == (4, 256)
  assert 'GoalVMC/DynBuffEncoder/conv5/kernel/Adam_1' in t and 'beta1_power' in t
  ev = e.evaluate(input_fn=synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=6))

  r = np.random.default_rng(3)
  frames, jnts = r.random([4, 136, 136, 3], dtype=np.float32), r.standard_normal([4, 7]).astype(np.float32)
  tgt = r.random([136, 136, 3], dtype=np.float32)

  def run(model_dir):
    p = GoalE2EVMCPredictor(model_dir)
    p.set_goal(tgt)
    return [p.predict(frames[i], jnts[i]) for i in range(4)]
  native = run(md)
  only_bundle = str(tmp_path / 'bundle_only')
  os.makedirs(only_bundle)
  for fn in os.listdir(md):
    if not fn.endswith('.pt'):
      shutil.copy(os.path.join(md, fn), only_bundle)
  assert est.latest_checkpoint(only_bundle) == os.path.join(only_bundle, 'model.ckpt-2')
  bundle = run(only_bundle)
  for a, b in zip(native, bundle):
    assert set(a) == set(b)
    for k in a:
      np.testing.assert_array_equal(a[k], b[k], err_msg=k)          # bit-equal: the bundle holds the same float32 values
  # an Estimator pointed at the bundle-only directory resumes: same eval, same step, and training continues
  e2 = est.Estimator(est.goal_e2evmc_model_fn, only_bundle, est.RunConfig(save_tf_bundle=True), params)
  ev2 = e2.evaluate(input_fn=synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=6))
  assert ev2['global_step'] == 2
  np.testing.assert_allclose(ev2['loss'], ev['loss'], rtol=1e-6)
  np.testing.assert_array_equal(e2._store.adam_v.cpu().numpy(), e._store.adam_v.cpu().numpy())
  e2.train(input_fn=synthetic_batches(4, 3, 1, (136, 136), 3, True, seed=7))
  assert os.path.exists(os.path.join(only_bundle, 'model.ckpt-3.index'))
