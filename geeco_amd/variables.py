"""Flat parameter arenas with TF-named views.

The reference keeps 60 ``tf.Variable``s (+ Adam slots) created by ``tf.layers`` under the scopes
listed in SURVEY.md 8b.  Here all trainable variables live in ONE contiguous fp32 buffer in HBM
(creation order = TF's), with three same-shaped siblings (gradient, Adam m, Adam v), so the
optimiser is a single fused pass and data-parallel training all-reduces a single bucket.
Each variable starts on a 16-byte boundary (float4 loads in the conv kernels); the few pad floats
are zero and stay zero under Adam (g = 0).
"""
from __future__ import annotations

import collections
import math

import numpy as np
import torch

ENC_FILTERS = (32, 48, 64, 128, 192, 256, 256)      # conv1..conv7 (graph.py:76-110); conv8 = dim_out
ENC_STRIDES = (1, 2, 2, 2, 2, 2, 2, 2)


def encoder_shapes(scope, cin, dim_out):
  """<scope>/conv{i}/{kernel [3,3,Cin,Cout], bias [Cout]} (graph.py:76-115)."""
  shapes = collections.OrderedDict()
  c = cin
  for i, f in enumerate(list(ENC_FILTERS) + [dim_out]):
    shapes['%s/conv%d/kernel' % (scope, i + 1)] = (3, 3, c, f)
    shapes['%s/conv%d/bias' % (scope, i + 1)] = (f,)
    c = f
  return shapes


def decoder_shapes(scope, dim_in, cfg):
  """LSTMCell kernel/bias, fc1, heads (graph.py:217-259)."""
  H = cfg.dim_h_lstm
  s = collections.OrderedDict()
  s[scope + '/lstm_cell/kernel'] = (dim_in + H, 4 * H)
  s[scope + '/lstm_cell/bias'] = (4 * H,)
  s[scope + '/fc1/kernel'] = (H, cfg.dim_h_fc)
  s[scope + '/fc1/bias'] = (cfg.dim_h_fc,)
  if cfg.control_mode == 'cartesian':
    heads = [('pred_cmd_ee', 3), ('logits_cmd_grp', cfg.num_grp_states)]
  elif cfg.control_mode == 'velocity':
    heads = [('pred_cmd_vel', cfg.dim_jnt_state), ('pred_cmd_ee', 3), ('pred_cmd_grp', cfg.dim_grp_command)]
  else:
    raise ValueError("Unknown control mode '%s'" % (cfg.control_mode,))
  heads += [('pred_aux_ee', 3), ('pred_aux_obj', 3)]
  for name, n in heads:
    s['%s/%s/kernel' % (scope, name)] = (cfg.dim_h_fc, n)
    s['%s/%s/bias' % (scope, name)] = (n,)
  return s


class VariableStore:
  """Named views into flat param / grad / Adam arenas."""

  ALIGN = 4   # floats

  def __init__(self, shapes, device, uniform_scopes=None):
    """``uniform_scopes``: scope prefixes (the encoders of one model) whose variable blocks must start at a COMMON
    stride, so that one grouped launch can address encoder g at base + g * stride.  Blocks of equal size (the default
    dim_s_obs == dim_s_dyn == dim_s_diff) already do; a shorter block (smaller conv8) is followed by zero pad floats."""
    self.shapes = collections.OrderedDict(shapes)
    self.device = torch.device(device)
    self.offsets = collections.OrderedDict()
    al = lambda v: -(-v // self.ALIGN) * self.ALIGN
    stride, first = 0, {}
    if uniform_scopes and len(uniform_scopes) > 1:
      for g, sc in enumerate(uniform_scopes):
        names = [n for n in self.shapes if n.startswith(sc + '/')]
        first[names[0]] = g
        stride = max(stride, sum(al(int(np.prod(self.shapes[n]))) for n in names))
    off, start0 = 0, None
    for name, shp in self.shapes.items():
      off = al(off)
      if name in first:
        if first[name] == 0:
          start0 = off
        else:
          off = max(off, start0 + first[name] * stride)
      self.offsets[name] = off
      off += int(np.prod(shp))
    self.size = -(-off // self.ALIGN) * self.ALIGN
    self.params = torch.zeros(self.size, dtype=torch.float32, device=self.device)
    self.grads = torch.zeros(self.size, dtype=torch.float32, device=self.device)
    self.adam_m = torch.zeros(self.size, dtype=torch.float32, device=self.device)
    self.adam_v = torch.zeros(self.size, dtype=torch.float32, device=self.device)
    self.global_step = torch.zeros(1, dtype=torch.int64, device=self.device)
    self.version = 0      # bumped whenever the parameters are (re)written from the host side

  # -- views ------------------------------------------------------------------------------
  def _view(self, arena, name):
    shp = self.shapes[name]
    o = self.offsets[name]
    return arena[o:o + int(np.prod(shp))].view(*shp)

  def var(self, name):
    return self._view(self.params, name)

  def grad(self, name):
    return self._view(self.grads, name)

  def count_parameters(self) -> int:
    """models/e2evmc/utils.py:10-14."""
    return int(sum(int(np.prod(s)) for s in self.shapes.values()))

  # -- initialisation / (de)serialisation -----------------------------------------------------
  def initialize(self, seed=0):
    """glorot-uniform kernels, zero biases (tf.layers / LSTMCell defaults [TF1.15])."""
    rng = np.random.default_rng(seed)
    for name, shp in self.shapes.items():
      if name.endswith('/bias'):
        val = np.zeros(shp, np.float32)
      else:
        if len(shp) == 4:
          fan_in, fan_out = shp[0] * shp[1] * shp[2], shp[0] * shp[1] * shp[3]
        else:
          fan_in, fan_out = shp
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        val = rng.uniform(-lim, lim, size=shp).astype(np.float32)
      self.var(name).copy_(torch.from_numpy(val))
    self.version += 1

  def load_numpy(self, values: dict):
    for name in self.shapes:
      self.var(name).copy_(torch.from_numpy(np.ascontiguousarray(values[name], dtype=np.float32)))
    self.version += 1

  def to_numpy(self, which='params'):
    arena = {'params': self.params, 'grads': self.grads, 'adam_m': self.adam_m, 'adam_v': self.adam_v}[which]
    host = arena.detach().cpu().numpy()
    out = collections.OrderedDict()
    for name, shp in self.shapes.items():
      o = self.offsets[name]
      out[name] = host[o:o + int(np.prod(shp))].reshape(shp).copy()
    return out

  def state_dict(self):
    return {'params': self.params.detach().cpu(), 'adam_m': self.adam_m.detach().cpu(),
            'adam_v': self.adam_v.detach().cpu(), 'global_step': self.global_step.detach().cpu(),
            'names': list(self.shapes.keys()), 'shapes': [tuple(s) for s in self.shapes.values()],
            'offsets': list(self.offsets.values())}

  def load_state_dict(self, sd):
    if list(sd['names']) != list(self.shapes.keys()) or [tuple(s) for s in sd['shapes']] != \
        [tuple(s) for s in self.shapes.values()]:
      raise ValueError('checkpoint variables do not match the model graph')
    self.params.copy_(sd['params'])
    self.adam_m.copy_(sd['adam_m'])
    self.adam_v.copy_(sd['adam_v'])
    self.global_step.copy_(sd['global_step'])
    self.version += 1
