"""Boundary test on the GPU: Estimator.train / evaluate / checkpoint-resume through the model_fn
surface, and the train_e2evmc.py counterpart end to end on synthetic windows."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(**kw):
  from geeco_amd.params import create_e2evmc_config
  base = dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4, lr=1e-3)
  base.update(kw)
  return {'e2evmc_config': create_e2evmc_config(base), 'log_steps': 2, 'debug': False}


def test_estimator_train_eval_resume(dev, tmp_path):
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import synthetic_batches
  params = _params()
  train_in = synthetic_batches(4, 3, 6, (136, 136), 3, True, seed=5)
  eval_in = synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=6)
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(save_checkpoints_steps=4, keep_checkpoint_max=2), params)
  r0 = None
  for epoch in range(3):
    e.train(input_fn=train_in)
    r = e.evaluate(input_fn=eval_in)
    assert set(r) == {'loss', 'cmd_ee', 'pos_ee', 'pos_obj', 'cmd_grp', 'global_step'}
    r0 = r0 or r
  assert r['global_step'] == 18
  assert r['loss'] < r0['loss']            # same batches every epoch: the loss must go down
  assert 0.0 <= r['cmd_grp'] <= 1.0
  ck = est.latest_checkpoint(str(tmp_path))
  assert os.path.basename(ck) == 'model.ckpt-18' and os.path.exists(ck + '.pt')
  kept = sorted(fn for fn in os.listdir(tmp_path) if fn.endswith('.pt'))
  assert len(kept) == 2, kept
  assert os.path.exists(tmp_path / 'events.jsonl')
  # a fresh Estimator on the same model_dir resumes from the checkpoint: same eval result
  e2 = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(), params)
  r2 = e2.evaluate(input_fn=eval_in)
  assert r2['global_step'] == 18
  np.testing.assert_allclose(r2['loss'], r['loss'], rtol=1e-6)
  # hipGraph replay and eager launches give the same numbers
  e3 = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(use_hipgraph=False), params)
  r3 = e3.evaluate(input_fn=eval_in)
  np.testing.assert_allclose(r3['loss'], r['loss'], rtol=1e-6)


def test_estimator_errors(dev, tmp_path):
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import synthetic_batches
  from geeco_amd.params import create_e2evmc_config
  bad = {'e2evmc_config': create_e2evmc_config(dict(proc_obs='nope', proc_tgt='dyndiff', window_size=3, img_height=136,
                                                    img_width=136)), 'log_steps': 1, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(), bad)
  with pytest.raises(ValueError):
    e.train(input_fn=synthetic_batches(2, 3, 1, (136, 136), 3, True))
  bad2 = {'e2evmc_config': create_e2evmc_config(dict(img_channels=5)), 'log_steps': 1, 'debug': False}
  with pytest.raises(ValueError):
    est.Estimator(est.e2evmc_model_fn, str(tmp_path), est.RunConfig(), bad2).train(
        input_fn=synthetic_batches(2, 4, 1, (136, 136), 3, False))


def test_estimator_raises_on_an_expired_input_stage_wait(dev, tmp_path):
  """A device-side error of the one-pass input stage reaches the caller of Estimator.train / evaluate (the places where the host reads
  results anyway: loss read-outs, epoch ends) as a RuntimeError, and nothing is saved for that epoch.  Provoked with a wait bound of zero
  polls (tests/test_kernels_gpu.py::test_goal_dynimgs_expired_wait_is_loud); a healthy epoch before it passes."""
  from geeco_amd import estimator as est
  from geeco_amd._native import load as lib
  from geeco_amd.input_fn import synthetic_batches
  params = _params()
  params['log_steps'] = 1000               # (no loss read-out inside the epoch: the check at the epoch's end must catch it)
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(use_hipgraph=False), params)
  e.train(input_fn=synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=5))
  assert os.path.basename(est.latest_checkpoint(str(tmp_path))) == 'model.ckpt-2'
  old = lib().geeco_goal_dynimgs_set_wait_polls(0)
  try:
    with pytest.raises(RuntimeError, match='one-pass input stage'):
      e.train(input_fn=synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=6))
  finally:
    lib().geeco_goal_dynimgs_set_wait_polls(old)
  assert os.path.basename(est.latest_checkpoint(str(tmp_path))) == 'model.ckpt-2'      # the poisoned epoch wrote no checkpoint
  # evaluation through the same (poisoned) workspace reports it too: the word is sticky until the workspace is zero-filled again
  with pytest.raises(RuntimeError, match='one-pass input stage'):
    e.evaluate(input_fn=synthetic_batches(4, 3, 1, (136, 136), 3, True, seed=7))


def test_train_script_synthetic(dev, tmp_path):
  md = str(tmp_path / 'run')
  cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'train_e2evmc.py'), '--dataset_dir', 'synthetic:4:136x136',
         '--model_dir', md, '--goal_condition', 'target', '--proc_obs', 'dynimg', '--proc_tgt', 'dyndiff',
         '--window_size', '3', '--batch_size', '4', '--train_epochs', '2', '--log_steps', '2', '--num_best_ckpt', '1']
  # image size comes from the config defaults (256); override through a pre-seeded config JSON (restart protocol)
  os.makedirs(md)
  from geeco_amd.params import create_e2evmc_config
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136,
                                  batch_size=4))
  with open(os.path.join(md, 'e2evmc_config.json'), 'w') as f:
    json.dump(cfg._asdict(), f)
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  assert os.path.exists(os.path.join(md, 'snapshots', 'snapshot_index.json'))
  idx = json.load(open(os.path.join(md, 'snapshots', 'snapshot_index.json')))
  assert len(idx) == 1
  (name, info), = idx.items()
  assert os.path.exists(os.path.join(info['dir'], name + '.pt'))
  assert os.path.exists(os.path.join(info['dir'], 'checkpoint'))
  assert any(fn.endswith('runcmd.json') for fn in os.listdir(info['dir']))


def test_device_windows_match_host_pipeline(dev, tmp_path):
  """pickplace_input_fn(device='cuda') (episode uploaded once as uint8, windows gathered in HBM by
  geeco_gather_windows) yields bit-identical batches to the host pipeline, and the Estimator trains from it."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_host_logic_cpu import _make_dataset
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import DeviceWindows, pickplace_input_fn
  root = str(tmp_path / 'ds')
  os.makedirs(root)
  _make_dataset(root, n_eps=2, T=9, H=136, W=136)
  kw = dict(window_size=3, fetch_target=True, batch_size=4)
  host = list(pickplace_input_fn(root, 'default', 'eval', **kw))
  devb = list(pickplace_input_fn(root, 'default', 'eval', device='cuda', **kw))
  assert len(host) == len(devb) == 3
  for (fh, lh), (fd, ld) in zip(host, devb):
    for k in fh:
      got = fd[k].numpy() if isinstance(fd[k], DeviceWindows) else fd[k]
      assert isinstance(fd[k], DeviceWindows) == (k in ('rgb', 'depth', 'target_rgb', 'target_depth')), k
      if isinstance(fd[k], DeviceWindows):     # uploaded from the prefetch THREAD to the explicit device of the caller
        assert fd[k].segments[0][0].device == torch.device('cuda', torch.cuda.current_device()), k
      np.testing.assert_array_equal(got, fh[k], err_msg=k)          # incl. the /255.0 of geeco_gym.py:312
    for k in lh:
      np.testing.assert_array_equal(ld[k], lh[k])
  from geeco_amd.params import create_e2evmc_config
  params = {'e2evmc_config': create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136,
                                                       img_width=136, batch_size=4)), 'log_steps': 1, 'debug': False}
  res = []
  for device in (None, 'cuda'):
    md = str(tmp_path / ('m_%s' % device))
    e = est.Estimator(est.goal_e2evmc_model_fn, md, est.RunConfig(), params)
    e.train(input_fn=lambda: pickplace_input_fn(root, 'default', 'train', seed=3, device=device, **kw))
    res.append(e.evaluate(input_fn=lambda: pickplace_input_fn(root, 'default', 'eval', device=device, **kw)))
  assert res[0]['global_step'] == res[1]['global_step'] == 3
  np.testing.assert_allclose(res[0]['loss'], res[1]['loss'], rtol=1e-6)


def test_batch_that_mixes_uint8_and_float_episodes(dev, tmp_path):
  """a dataset may hold episodes whose RGB values are integral (resident as uint8, divisor 255) next to episodes
  whose values are not (resident as float32 / 255, divisor 1).  A batch that straddles two such episodes is gathered segment
  by segment with each segment's own divisor (round 4 raised ValueError at the first such batch): bitwise the host pipeline's
  batches; the mixed batch is not ``is_u8()``, so the Estimator runs it through the dense-window model, and training works."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_host_logic_cpu import _make_dataset
  from geeco_amd import estimator as est
  from geeco_amd import input_fn as I
  root = str(tmp_path / 'ds')
  os.makedirs(root)
  meta, eps = _make_dataset(root, n_eps=2, T=9, H=136, W=136)
  d = eps[1]                      # rewrite episode 1 with non-integral colour values
  I.write_episode(os.path.join(root, 'data', 'ep001.tfrecord.zlib'), meta, d['rgb'].astype(np.float32) * np.float32(0.5) + np.float32(0.25),
                  d['depth'], d['cmd'], d['ctrl'], d['qpos'], d['qvel'], d['mocap'], d['obj'], d['goal'])
  kw = dict(window_size=3, fetch_target=True, batch_size=4)
  host = list(I.pickplace_input_fn(root, 'default', 'eval', **kw))
  devb = list(I.pickplace_input_fn(root, 'default', 'eval', device='cuda', cache=False, **kw))
  assert len(host) == len(devb) == 3
  kinds = []
  for (fh, _), (fd, _) in zip(host, devb):
    kinds.append((fd['rgb'].is_u8(), sorted({dv for _, _, dv in fd['rgb'].segments})))
    for k in ('rgb', 'target_rgb', 'depth'):
      np.testing.assert_array_equal(fd[k].numpy(), fh[k], err_msg=k)
  assert kinds == [(True, [255.0]), (False, [1.0, 255.0]), (False, [1.0])], kinds     # 6 + 6 windows in batches of 4: the middle one straddles
  params = {'e2evmc_config': __import__('geeco_amd.params', fromlist=['x']).create_e2evmc_config(
      dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4)), 'log_steps': 1, 'debug': False}
  res = []
  for device in (None, 'cuda'):
    e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path / ('m_%s' % device)), est.RunConfig(), params)
    e.train(input_fn=lambda: I.pickplace_input_fn(root, 'default', 'train', seed=3, device=device, cache=False, **kw))
    res.append(e.evaluate(input_fn=lambda: I.pickplace_input_fn(root, 'default', 'eval', device=device, cache=False, **kw)))
  assert res[0]['global_step'] == res[1]['global_step'] == 3
  np.testing.assert_allclose(res[0]['loss'], res[1]['loss'], rtol=1e-6)


def test_u8_window_addresses_equal_dense_windows(dev, tmp_path):
  """geeco-f on HBM-resident episodes: the model's input kernel follows window addresses into the uint8 frames
  (input_fn.WindowFeed.pointers(), no fp32 window tensor, no gather launch); training and evaluation are BITWISE what the
  dense path (geeco_gather_windows into a float32 buffer, GEECO_NO_U8_WINDOWS) computes - RGB and RGB-D, batches that span
  two episodes, a ragged last batch, graph replay across repointed tables."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_host_logic_cpu import _make_dataset
  from geeco_amd import estimator as est, _dev
  from geeco_amd.input_fn import EPISODE_CACHE, WindowFeed, pickplace_input_fn
  from geeco_amd.params import create_e2evmc_config
  root = str(tmp_path / 'ds')
  os.makedirs(root)
  _make_dataset(root, n_eps=3, T=9, H=136, W=136)
  kw = dict(window_size=3, fetch_target=True, batch_size=5, device='cuda')
  for channels in (3, 4):
    params = {'e2evmc_config': create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136,
                                                         img_width=136, img_channels=channels, batch_size=5)),
              'log_steps': 1000, 'debug': False}
    res = []
    for dense in (False, True):
      EPISODE_CACHE.clear()
      old = dict(os.environ)
      if dense:
        os.environ.update(GEECO_DEV='1', GEECO_NO_U8_WINDOWS='1')
      try:
        e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(init_seed=5), params)
        for _ in range(2):          # epoch 2 comes from the episode cache
          e.train(input_fn=lambda: pickplace_input_fn(root, 'default', 'train', seed=3, **kw))
        ev = e.evaluate(input_fn=lambda: pickplace_input_fn(root, 'default', 'eval', **kw))
      finally:
        os.environ.clear()
        os.environ.update(old)
      feeds = [f for (spec, fbuf, lbuf) in e._specs.values() for f in fbuf.values() if isinstance(f, WindowFeed)]
      took = {(f.table is not None, f.buffer is not None) for f in feeds if f.frame_shape[-1] == 3}
      assert took == ({(False, True)} if dense else {(True, False)}), took
      assert all(f.buffer is not None for f in feeds if f.frame_shape[-1] == 1)        # depth stays dense float32
      res.append((ev, {n: e.get_variable_value(n) for n in e.get_variable_names()}))
    (ev_a, var_a), (ev_b, var_b) = res
    assert ev_a == ev_b, (ev_a, ev_b)
    assert ev_a['global_step'] > 2
    for n in var_a:
      np.testing.assert_array_equal(var_a[n], var_b[n], err_msg=n)


def test_ragged_final_batch_and_shared_store(dev, tmp_path):
  """dataset.batch() keeps a ragged final batch (geeco_gym.py:471); the Estimator builds a second graph for
  it on the SAME variables.  A 4+4+2 epoch followed by a full batch must equal the same four steps taken through one-off models."""
  from geeco_amd import estimator as est
  from geeco_amd import graph
  from geeco_amd.input_fn import synthetic_batches
  params = _params(window_size=2)
  full = list(synthetic_batches(4, 2, 3, (136, 136), 3, True, seed=8)())
  f3, l3 = full[2]
  ragged = full[:2] + [({k: v[:2] for k, v in f3.items()}, {k: v[:2] for k, v in l3.items()})]
  # ... followed by a full batch again (the next epoch's first step runs on the PRIMARY model, whose derived weight
  # copies must have been refreshed by the ragged model's Adam step)
  ragged = ragged + [full[0]]
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path / 'a'), est.RunConfig(use_hipgraph=True), params)
  e.train(input_fn=lambda: iter(ragged))
  assert int(e._store.global_step.item()) == 4 and len(e._specs) == 2
  got = e._store.to_numpy('params')
  # reference: the same three steps, eager, fresh model objects sharing one store
  cfg = params['e2evmc_config']
  store = None
  for f, l in ragged:
    n = f['rgb'].shape[0]
    m = graph.GoalE2EVMC(cfg, n, dev, training=True, store=store)
    if store is None:
      store = m.store
      store.initialize(seed=0)
    m.load_batch({k: torch.from_numpy(v) for k, v in f.items()}, {k: torch.from_numpy(v) for k, v in l.items()})
    m.train_step()
  torch.cuda.synchronize()
  want = store.to_numpy('params')
  for k in want:
    np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-7, err_msg=k)


def test_prefetch_uploads_during_graph_capture(dev, tmp_path):
  """The input prefetch thread keeps uploading episodes (hipMalloc + synchronous H2D copies) while the training thread
  captures its hipGraphs at step 3 and replays them afterwards: both sides take runtime.CAPTURE_LOCK, so neither the
  capture nor an upload is invalidated.  Eight episodes behind a prefetch depth of one batch keep the producer busy for
  the whole run."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_host_logic_cpu import _make_dataset
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import pickplace_input_fn
  from geeco_amd.params import create_e2evmc_config
  root = str(tmp_path / 'ds')
  os.makedirs(root)
  _make_dataset(root, n_eps=8, T=12, H=136, W=136)
  params = {'e2evmc_config': create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136,
                                                       img_width=136, batch_size=4)), 'log_steps': 100, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path / 'm'), est.RunConfig(use_hipgraph=True), params)
  e.train(input_fn=lambda: pickplace_input_fn(root, 'default', 'train', window_size=3, fetch_target=True, batch_size=4,
                                              prefetch_size=1, seed=1, device='cuda'))
  # 8 episodes x (11 - 3 + 1) windows = 72 windows = 18 steps, all but the first two replayed
  assert int(e._store.global_step.item()) == 18
  r = e.evaluate(input_fn=lambda: pickplace_input_fn(root, 'default', 'eval', window_size=3, fetch_target=True, batch_size=4,
                                                     prefetch_size=1, device='cuda'))
  assert np.isfinite(r['loss']) and r['global_step'] == 18


@pytest.mark.parametrize('mode', ['cartesian', 'velocity'])
def test_eval_metric_values_match_oracle(dev, tmp_path, mode):
  """Estimator.evaluate's numbers, not only its keys: 'loss' = mean of the per-batch losses [TF1.15 Estimator], streaming
  mean_squared_error of cmd_ee / pos_ee / pos_obj (and cmd_vel / cmd_grp in velocity mode) = sum of squared errors over
  ALL elements of all batches / their count, accuracy of cmd_grp = correct / total (estimator.py:246-258), against the
  fp64 oracle evaluated on the same three batches (the last one ragged) with the same weights."""
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import synthetic_batches
  from oracle import geeco_oracle as O
  kw = dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4, control_mode=mode)
  params = _params(**{k: v for k, v in kw.items() if k not in ('img_height', 'img_width', 'batch_size')})
  batches = list(synthetic_batches(4, 3, 3, (136, 136), 3, True, seed=12)())
  f3, l3 = batches[2]
  batches[2] = ({k: v[:3] for k, v in f3.items()}, {k: v[:3] for k, v in l3.items()})
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(init_seed=6), params)
  res = e.evaluate(input_fn=lambda: iter(batches))
  ocfg = O.make_config(**kw)
  P = {k: torch.tensor(e.get_variable_value(k), dtype=torch.float64) for k in e.get_variable_names()}
  losses, sq, cnt, correct, total = [], {}, {}, 0, 0
  for f, l in batches:
    ft = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in f.items() if k != 'step'}
    lt = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in l.items()}
    pred, _ = O.model_forward(ft, P, ocfg, True)
    tgt = O.build_targets(ft, lt, ocfg)
    loss, _ = O.model_loss(pred, tgt, P, ocfg)
    losses.append(float(loss))
    keys = ['cmd_ee', 'pos_ee', 'pos_obj'] + (['cmd_vel', 'cmd_grp'] if mode == 'velocity' else [])
    for k in keys:
      sq[k] = sq.get(k, 0.0) + float(((pred[k] - tgt[k]) ** 2).sum())
      cnt[k] = cnt.get(k, 0) + tgt[k].numel()
    if mode == 'cartesian':
      correct += int((pred['logits_cmd_grp'].argmax(dim=-1) == tgt['cmd_grp'].long()).sum())
      total += int(tgt['cmd_grp'].numel())
  np.testing.assert_allclose(res['loss'], np.mean(losses), rtol=1e-4)
  for k in sq:
    np.testing.assert_allclose(res[k], sq[k] / cnt[k], rtol=2e-4, err_msg=k)
  if mode == 'cartesian':
    assert abs(res['cmd_grp'] - correct / total) < 1e-9
  assert res['global_step'] == 0


def test_estimator_predict_mode(dev, tmp_path):
  """Estimator.predict (ModeKeys.PREDICT, estimator.py:63-70 / 183-197): label-side inputs may be absent, predictions carry
  the reference's keys and equal the oracle's forward on the same weights; an unknown mode raises RuntimeError."""
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import synthetic_batches
  from oracle import geeco_oracle as O
  params = _params()
  e = est.Estimator(est.goal_e2evmc_model_fn, str(tmp_path), est.RunConfig(init_seed=8), params)
  batches = list(synthetic_batches(4, 3, 2, (136, 136), 3, True, seed=21)())
  feats_only = [{k: v for k, v in f.items() if k in ('rgb', 'target_rgb', 'jnt_state', 'step')} for f, _ in batches]
  outs = list(e.predict(input_fn=lambda: iter(feats_only)))
  assert len(outs) == 2 and set(outs[0]) == {'cmd_ee', 'logits_cmd_grp', 'pos_ee', 'pos_obj'}
  ocfg = O.make_config(**params['e2evmc_config']._asdict())
  P = {k: torch.tensor(e.get_variable_value(k), dtype=torch.float64) for k in e.get_variable_names()}
  for f, out in zip(feats_only, outs):
    ft = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in f.items() if k != 'step'}
    pred, _ = O.model_forward(ft, P, ocfg, True)
    for k in out:
      np.testing.assert_allclose(out[k], pred[k].numpy(), rtol=1e-4, atol=2e-5, err_msg=k)
  with pytest.raises(RuntimeError):
    est.goal_e2evmc_model_fn({'rgb': torch.zeros(1, 3, 136, 136, 3, device=dev)}, None, 'train_and_eval', params)
