"""High-level predictor API (batch-1 inference) on the HIP kernels.

Counterpart of the reference's ``src/models/e2evmc/predictor.py``: ``GoalE2EVMCPredictor`` (:43-209)
and ``E2EVMCPredictor`` (:212-379) with the same constructor and methods (``predict``, ``reset``,
``set_goal``, ``cfg``).  Semantics kept: the frame buffer holds the last ``window_size`` frames and is
padded with the first frame after a reset (:192-200); frames must be [H, W, C] in [0, 1] (:127-138);
the gripper logits are re-mapped to {-1, 0, 1} (:183-189); ``dynbuff`` / ``dyndiff`` debug images are
returned when the model computes them (:167-170).  The frame buffer lives in HBM: a new frame is
uploaded once and the window is shifted on the device; the forward pass is a replayed hipGraph.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import estimator as est
from . import graph
from .params import create_e2evmc_config
from .runtime import EvalStepRunner
from .utils import load_model_config

TOL_FRAME_RANGE = 1e-6  # tolerance for value range of fed frames (predictor.py:18)


def _latest_tf_bundle(model_dir):
  """'<model_dir>/model.ckpt-<step>' named by a TF ``checkpoint`` file whose bundle (.index) exists."""
  import re
  index = os.path.join(model_dir, 'checkpoint')
  if not os.path.exists(index):
    return None
  m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(index).read())
  if not m:
    return None
  path = os.path.join(model_dir, os.path.basename(m.group(1)))
  return path if os.path.exists(path + '.index') else None


class _PredictorBase:
  _goal = False

  def __init__(self, model_dir, checkpoint_name=None, memcap=0.8, device=None):
    self._model_dir = model_dir
    cfg = load_model_config(model_dir, 'e2evmc_config')
    cfg['batch_size'] = 1   # one prediction at a time (predictor.py:56)
    self._cfg = create_e2evmc_config(cfg)
    if not torch.cuda.is_available():
      raise RuntimeError('geeco_amd predictor needs an MI355X (no CPU fallback)')
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    if memcap and 0.0 < memcap < 1.0:
      torch.cuda.set_per_process_memory_fraction(float(memcap), dev)
    ctor = graph.GoalE2EVMC if self._goal else graph.E2EVMC
    self._model = ctor(self._cfg, 1, dev, training=False)
    ckpt = os.path.join(model_dir, checkpoint_name) if checkpoint_name else est.latest_checkpoint(model_dir)
    if ckpt is None and not checkpoint_name:
      ckpt = _latest_tf_bundle(model_dir)
    if ckpt is None:
      raise FileNotFoundError('no checkpoint in %s' % model_dir)
    if os.path.exists(ckpt + '.pt'):
      est.load_checkpoint(self._model.store, ckpt)
    else:     # a TensorFlow-1.15 tensor bundle (e.g. the published geeco_models_icra21 weights)
      from . import tf_checkpoint
      tf_checkpoint.import_checkpoint(self._model.store, ckpt, load_optimizer=False)
    print('>>> Restored model parameters from %s' % (ckpt,))
    self._runner = EvalStepRunner(self._model, use_graph=True, warmup=1)
    self._buffer_size = self._cfg.window_size
    self._filled = 0
    self._target_set = False

  @property
  def cfg(self):
    return self._cfg

  # -- frame buffer (device resident) --------------------------------------------------------------
  def _check_frame(self, frame):
    cfg = self._cfg
    expected = (cfg.img_height, cfg.img_width, cfg.img_channels)
    assert tuple(frame.shape) == expected, \
        "Fed frame has wrong dimensions! Expected %s, got %s!" % (expected, tuple(frame.shape))
    lo, hi = float(np.amin(frame[..., :3])), float(np.amax(frame[..., :3]))
    assert -TOL_FRAME_RANGE <= lo and hi <= 1 + TOL_FRAME_RANGE, \
        "Fed frame exceeds range! Expected %s, got %s!" % ((0 - TOL_FRAME_RANGE, 1 + TOL_FRAME_RANGE), (lo, hi))

  def _feed_frame(self, rgb_frame, jnt_state):
    self._check_frame(rgb_frame)
    inp = self._model.inputs
    dev = self._model.device
    f = torch.as_tensor(np.ascontiguousarray(rgb_frame, dtype=np.float32)).to(dev, non_blocking=True)
    j = torch.as_tensor(np.ascontiguousarray(jnt_state, dtype=np.float32).reshape(-1)).to(dev, non_blocking=True)
    K = self._buffer_size
    first = self._filled == 0
    for key, val in (('rgb', f[..., :3]), ('depth', f[..., 3:4] if self._cfg.img_channels == 4 else None),
                     ('jnt_state', j)):
      if val is None:
        continue
      buf = inp[key][0]
      if first:     # pad the whole window with the first frame (predictor.py:197-198)
        buf.copy_(val.unsqueeze(0).expand_as(buf))
      else:         # drop the oldest frame, append the new one (predictor.py:144-146)
        if K > 1:
          buf[:-1].copy_(buf[1:].clone())
        buf[-1].copy_(val)
    self._filled = min(self._filled + 1, K)

  def _fetch(self):
    self._runner.step()
    torch.cuda.synchronize()
    preds = {k: v.detach().cpu().numpy().squeeze(0).copy() for k, v in self._model.predictions().items()}
    if self._cfg.control_mode == 'cartesian':
      out = {'cmd_ee': preds['cmd_ee'], 'pos_ee': preds['pos_ee'], 'pos_obj': preds['pos_obj']}
      # re-map the discrete gripper command: argmax class - 1 (predictor.py:183-189)
      out['cmd_grp'] = np.asarray([np.argmax(preds['logits_cmd_grp']) - 1], dtype=np.float32)
    else:                      # velocity mode fetches (predictor.py:157-164)
      out = {k: preds[k] for k in ('cmd_vel', 'cmd_ee', 'cmd_grp', 'pos_ee', 'pos_obj')}
    return out

  def predict(self, rgb_frame, jnt_state):
    """Feeds the frame (padding the buffer after a reset) and returns the predictions."""
    self._feed_frame(rgb_frame, jnt_state)
    return self._predict_command()

  def reset(self):
    self._filled = 0


class GoalE2EVMCPredictor(_PredictorBase):
  """High-level API to run goal-conditioned E2EVMC (predictor.py:43-209)."""
  _goal = True

  def _predict_command(self):
    if not self._target_set:
      raise RuntimeError('set_goal(tgt_frame) must be called before predict()')
    out = self._fetch()
    C = self._cfg.img_channels
    ep = self._model.endpoints()
    if self._cfg.proc_obs == 'dynimg':
      out['dynbuff'] = ep['dynbuff'][0].detach().cpu().numpy()[..., :C].copy()
    if self._cfg.proc_tgt == 'dyndiff':
      out['dyndiff'] = ep['dyndiff'][0].detach().cpu().numpy()[..., :C].copy()
    return out

  def set_goal(self, tgt_frame):
    """Sets the target frame (predictor.py:206-209)."""
    C = self._cfg.img_channels
    t = np.ascontiguousarray(tgt_frame[:, :, :C], dtype=np.float32)
    inp = self._model.inputs
    inp['target_rgb'][0].copy_(torch.from_numpy(t[..., :3]))
    if C == 4:
      inp['target_depth'][0].copy_(torch.from_numpy(t[..., 3:4]))
    self._target_set = True


class E2EVMCPredictor(_PredictorBase):
  """High-level API to run E2E VMC (predictor.py:212-379)."""
  _goal = False

  def _predict_command(self):
    return self._fetch()
