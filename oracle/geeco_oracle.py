"""CPU oracle for GEECO's e2evmc training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the shipped package ``geeco_amd`` never does and fails loudly when its
HIP library is missing.

PARITY UNPINNED: the reference (ogroth/geeco) has no tests, golden vectors or fixtures, and
its arithmetic lives in TensorFlow 1.15.0 (environment.yml:189-192), which is not installed
here (``import tensorflow`` -> ModuleNotFoundError).  This file therefore restates the
reference's graph from its source text plus the published TF-1.15 op semantics listed in
SURVEY.md par. 8c, and is pinned only by the analytic known-answer tests in
``tests/test_oracle_kat.py`` (alpha tables, SAME/stride-2 alignment probe, zero-weight LSTM,
ln 3 cross-entropy, Adam step 1, parameter counts) and -- round 6 -- by literals of TensorFlow 1.15's OWN
unit tests reproduced from memory and confirmed by exact agreement with this file (rnn_cell_test.py
testBasicLSTMCell; conv_ops_test.py testConv2D2x2FilterStride2Same / testConv2DKernelSmallerThanStrideSame /
testConv2D1x1Filter; losses_test.py MeanSquaredErrorTest.testNonZeroLoss, SoftmaxCrossEntropyLossTest): the
LSTM cell formula, padding='SAME' with strides, the two loss reductions.  Still unpinned by any TF-written
number: Adam's epsilon placement, the LSTM gate order, rint.

Every function cites the reference file:line it follows (paths relative to /root/reference).
The restatement is written with torch CPU tensors (fp32 or fp64 selectable) so gradients come
from autograd of the restated forward; ``conv2d_same_numpy`` is an independent loop/einsum
restatement used to check the torch conv path on small cases.
"""
from __future__ import annotations

import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# hyper-parameters: src/models/e2evmc/params.py:7-28
# --------------------------------------------------------------------------------------
DEFAULT_PARAMS = collections.OrderedDict([
    ('img_height', 256), ('img_width', 256), ('img_channels', 3), ('dim_jnt_state', 7),
    ('dim_grp_command', 2), ('control_mode', 'cartesian'), ('num_grp_states', 3),
    ('dim_action', 4), ('proc_obs', 'sequence'), ('proc_tgt', 'constant'),
    ('dim_s_obs', 256), ('dim_s_dyn', 256), ('dim_s_diff', 256), ('dim_h_lstm', 128),
    ('dim_h_fc', 128), ('window_size', 4), ('l2_regularizer', 0.0), ('lambda_aux', 1.0),
    ('batch_size', 32), ('lr', 1e-4),
])
Config = collections.namedtuple('Config', list(DEFAULT_PARAMS.keys()))


def make_config(**kw):
  """params.py:37-47 -- unknown keys are silently dropped."""
  d = dict(DEFAULT_PARAMS)
  for k, v in kw.items():
    if k in d:
      d[k] = v
  return Config(**d)


ENC_FILTERS = (32, 48, 64, 128, 192, 256, 256)   # conv1..conv7; conv8 -> dim_out  (graph.py:76-115)
ENC_STRIDES = (1, 2, 2, 2, 2, 2, 2, 2)


# --------------------------------------------------------------------------------------
# dynamic image: graph.py:17-55
# --------------------------------------------------------------------------------------
def harmonic(t: int) -> np.float32:
  """graph.py:17-23: H_t = sum_{i=1..t} 1/i in float32 (0 for t == 0)."""
  h = np.float32(0.0)
  for i in range(1, int(t) + 1):
    h = np.float32(h + np.float32(1.0) / np.float32(i))
  return h


def dynimg_alpha(T: int) -> np.ndarray:
  """graph.py:25-28,41-42: alpha_t = 2(T-t+1) - (T+1)(H_T - H_{t-1}), t = 1..T, float32."""
  HT = harmonic(T)
  out = np.zeros([T], np.float32)
  for t in range(1, T + 1):
    out[t - 1] = np.float32(2 * (T - t + 1)) - np.float32(T + 1) * np.float32(HT - harmonic(t - 1))
  return out


def dynimg(frames: torch.Tensor) -> torch.Tensor:
  """graph.py:30-55. frames [N,K,H,W,C] -> normalised dynamic image [N,H,W,C]."""
  N, K = frames.shape[0], frames.shape[1]
  w = torch.from_numpy(dynimg_alpha(K)).to(frames.dtype).reshape(1, K, 1, 1, 1)
  d = (w * frames).sum(dim=1)                                   # :44-45
  mn = d.reshape(N, -1).min(dim=1).values.reshape(N, 1, 1, 1)    # :47
  mx = d.reshape(N, -1).max(dim=1).values.reshape(N, 1, 1, 1)    # :48
  rng = mx - mn + 1e-6                                          # :49
  return (d - mn) / rng                                         # :54


# --------------------------------------------------------------------------------------
# TF 'SAME' padding [TF1.15 semantics, SURVEY 8c(1)]
# --------------------------------------------------------------------------------------
def same_pad(size: int, k: int, s: int):
  out = -(-size // s)
  total = max((out - 1) * s + k - size, 0)
  before = total // 2
  return out, before, total - before


def conv2d_same(x: torch.Tensor, w_hwio: torch.Tensor, b: torch.Tensor, stride: int, relu=True):
  """tf.layers.conv2d(padding='SAME', activation=relu) on NHWC input with an HWIO kernel
  (graph.py:76-115)."""
  N, H, W, C = x.shape
  kh, kw = w_hwio.shape[0], w_hwio.shape[1]
  _, pt, pb = same_pad(H, kh, stride)
  _, pl, pr = same_pad(W, kw, stride)
  xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
  y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), b, stride=stride)
  y = y.permute(0, 2, 3, 1)
  return torch.relu(y) if relu else y


def conv2d_same_numpy(x, w_hwio, b, stride, relu=True):
  """Independent restatement (explicit window gather + einsum) used to check conv2d_same."""
  x = np.asarray(x, np.float64); w = np.asarray(w_hwio, np.float64); b = np.asarray(b, np.float64)
  N, H, W, C = x.shape
  Ho, pt, pb = same_pad(H, 3, stride)
  Wo, pl, pr = same_pad(W, 3, stride)
  xp = np.zeros([N, H + pt + pb, W + pl + pr, C])
  xp[:, pt:pt + H, pl:pl + W] = x
  y = np.zeros([N, Ho, Wo, w.shape[3]])
  for ky in range(3):
    for kx in range(3):
      win = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride]
      y += np.einsum('nhwc,co->nhwo', win, w[ky, kx])
  y += b
  return np.maximum(y, 0.0) if relu else y


# --------------------------------------------------------------------------------------
# parameters (TF variable names / shapes / initialisers, SURVEY 8b + 8c(2))
# --------------------------------------------------------------------------------------
def encoder_param_shapes(scope: str, cin: int, dim_out: int):
  """graph.py:76-115: <scope>/conv{i}/{kernel[3,3,Cin,Cout], bias[Cout]}."""
  shapes = collections.OrderedDict()
  c = cin
  for i, f in enumerate(list(ENC_FILTERS) + [dim_out]):
    shapes['%s/conv%d/kernel' % (scope, i + 1)] = (3, 3, c, f)
    shapes['%s/conv%d/bias' % (scope, i + 1)] = (f,)
    c = f
  return shapes


def decoder_param_shapes(scope: str, dim_in: int, cfg: Config):
  """graph.py:217-259: LSTMCell kernel [in+H, 4H]; fc1; heads."""
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


def model_param_shapes(cfg: Config, goal: bool):
  """Variable creation order of e2e_vmc (graph.py:268-319) / goal_e2evmc (graph.py:321-416)."""
  C = cfg.img_channels
  jn = cfg.dim_jnt_state
  s = collections.OrderedDict()
  if not goal:
    s.update(encoder_param_shapes('VMC/ConvEncoder', C, 256))           # :311 default dim_out
    s.update(decoder_param_shapes('VMC/LSTMDecoder', 4 * (256 + jn), cfg))
    return s
  root = 'GoalVMC'
  if cfg.proc_tgt not in ('constant', 'residual', 'dyndiff'):
    raise ValueError("Unknown processing mode for target image: %s!" % (cfg.proc_tgt,))
  if cfg.proc_obs == 'sequence':
    s.update(encoder_param_shapes(root + '/ConvEncoder', C, cfg.dim_s_obs))
    if cfg.proc_tgt == 'constant':
      din = 4 * (cfg.dim_s_obs + jn + cfg.dim_s_obs)
    elif cfg.proc_tgt == 'residual':
      din = 4 * (cfg.dim_s_obs + jn)
    else:
      s.update(encoder_param_shapes(root + '/DynDiffEncoder', C, cfg.dim_s_diff))
      din = 4 * (cfg.dim_s_obs + jn + cfg.dim_s_diff)
  elif cfg.proc_obs == 'dynimg':
    # NB graph.py:386-407 ignores proc_tgt in this branch (always dyndiff).
    s.update(encoder_param_shapes(root + '/ConvEncoder', C, cfg.dim_s_obs))
    s.update(encoder_param_shapes(root + '/DynBuffEncoder', C, cfg.dim_s_dyn))
    s.update(encoder_param_shapes(root + '/DynDiffEncoder', C, cfg.dim_s_diff))
    din = 4 * (cfg.dim_s_obs + cfg.dim_s_dyn + jn + cfg.dim_s_diff)
  else:
    raise ValueError("Unknown processing mode for frame buffer: %s!" % (cfg.proc_obs,))
  s.update(decoder_param_shapes(root + '/LSTMDecoder', din, cfg))
  return s


def count_parameters(shapes) -> int:
  """utils.py:10-14."""
  return int(sum(int(np.prod(v)) for v in shapes.values()))


def init_params(shapes, seed=0, dtype=np.float32):
  """glorot-uniform kernels / zero biases [TF1.15 defaults]; numpy RNG (TF's stream cannot be
  matched, weights are always exported/imported when comparing)."""
  rng = np.random.default_rng(seed)
  out = collections.OrderedDict()
  for name, shp in shapes.items():
    if name.endswith('/bias'):
      out[name] = np.zeros(shp, dtype)
    else:
      if len(shp) == 4:
        fan_in, fan_out = shp[0] * shp[1] * shp[2], shp[0] * shp[1] * shp[3]
      else:
        fan_in, fan_out = shp
      lim = math.sqrt(6.0 / (fan_in + fan_out))
      out[name] = rng.uniform(-lim, lim, size=shp).astype(dtype)
  return out


# --------------------------------------------------------------------------------------
# graph pieces
# --------------------------------------------------------------------------------------
class ReluTap:
  """Test instrumentation for the whole-graph restatements below (``e2e_vmc`` / ``goal_e2evmc``): SOMEBODY ELSE'S ReLU
  decisions (and, optionally, conv1 inputs) for every ``conv_encoder`` call of one forward pass.

  ``masks_fn(scope, call)`` -> eight bool tensors ``[n, h, w, c]`` (the device's ``y > 0`` of the frames which the
  ``call``-th ``conv_encoder`` invocation under ``scope`` processes; calls are counted per scope in graph order:
  graph.py:310-313, :354, :362-381, :390-402).  ``inputs_fn(scope, call)`` -> the frames the device fed its conv1 for
  that call, or None to keep this restatement's own (only the dynamic images differ, by fp32 rounding).  With
  ``force`` the activations are ``z * mask`` (see ``conv_encoder``); ``stats[scope][layer]`` =
  [disagreements, max |z| there, elements] against this restatement's own ``z > 0``."""

  def __init__(self, masks_fn, inputs_fn=None, force=True):
    self.masks_fn, self.inputs_fn, self.force = masks_fn, inputs_fn, force
    self.calls, self.stats = {}, {}
    self.recorded = {}          # masks_fn None: (scope, call) -> this restatement's own eight decisions

  def enter(self, scope, x):
    call = self.calls.get(scope, 0)
    self.calls[scope] = call + 1
    st = self.stats.setdefault(scope, [[0, 0.0, 0] for _ in range(8)])
    if self.inputs_fn is not None:
      xi = self.inputs_fn(scope, call)
      if xi is not None:
        assert tuple(xi.shape) == tuple(x.shape), (scope, call, tuple(xi.shape), tuple(x.shape))
        x = xi.to(x.dtype)
    if self.masks_fn is None:
      self.recorded[(scope, call)] = rec = []
      return x, None, rec
    return x, self.masks_fn(scope, call), st


def conv_encoder(x, P, scope, collect=None, masks=None, force=False, stats=None, tap=None):
  """graph.py:61-117.

  Test instrumentation (the product of every other caller is unchanged): ``masks`` = per layer a bool tensor with
  SOMEBODY ELSE'S ReLU decisions (the device's ``y > 0``).  ``stats`` ([layer] -> [disagreements, max |z| there,
  elements]) compares them with this restatement's own ``z > 0``; with ``force`` the activation is ``z * mask``:
  the forward changes only where the two disagree (|z| at rounding level, checked through ``stats``) and the
  backward is the gradient under the given decisions -- at z == 0 both are valid subgradients of ReLU, and which side
  of zero a pre-activation of 1e-8 lands on is rounding, not mathematics."""
  if tap is not None:
    x, masks, stats = tap.enter(scope, x)
    force = tap.force
  net = x
  for i in range(8):
    w, b = P['%s/conv%d/kernel' % (scope, i + 1)], P['%s/conv%d/bias' % (scope, i + 1)]
    if masks is None:
      net = conv2d_same(net, w, b, ENC_STRIDES[i], relu=True)
      if tap is not None:
        stats.append((net > 0).detach())
    else:
      z = conv2d_same(net, w, b, ENC_STRIDES[i], relu=False)
      own = z > 0
      if stats is not None:
        dis = own != masks[i]
        n = int(dis.sum())
        stats[i][0] += n
        if n:
          stats[i][1] = max(stats[i][1], float(z.detach().abs()[dis].max()))
        stats[i][2] += dis.numel()
      net = z * (masks[i] if force else own).to(z.dtype)
    if collect is not None:
      collect['%s/conv%d' % (scope, i + 1)] = net
  return net


def state_concatenation(feat, jnt):
  """graph.py:123-144: [feat | jnt] per 2x2 cell, flatten (h,w,c)."""
  N = feat.shape[0]
  st = jnt.reshape(N, 1, 1, -1).expand(N, 2, 2, jnt.shape[-1])
  return torch.cat([feat, st], dim=-1).reshape(N, -1)


def representation_concatenation(obs, tgt, jnt):
  """graph.py:146-167: [obs | jnt | tgt] (jnt in the middle)."""
  N = obs.shape[0]
  st = jnt.reshape(N, 1, 1, -1).expand(N, 2, 2, jnt.shape[-1])
  return torch.cat([obs, st, tgt], dim=-1).reshape(N, -1)


def representation_concatenation_v2(obs, dyn, jnt, tgt):
  """graph.py:169-192: [obs | dyn | jnt | tgt]."""
  N = obs.shape[0]
  st = jnt.reshape(N, 1, 1, -1).expand(N, 2, 2, jnt.shape[-1])
  return torch.cat([obs, dyn, st, tgt], dim=-1).reshape(N, -1)


def lstm_cell(x, c, h, kernel, bias, forget_bias=1.0):
  """tf.nn.rnn_cell.LSTMCell.call [TF1.15]: z = [x|h] W + b; i, j, f, o = split(z, 4);
  c' = sigmoid(f + 1) c + sigmoid(i) tanh(j); h' = sigmoid(o) tanh(c')."""
  z = torch.cat([x, h], dim=1) @ kernel + bias
  i, j, f, o = torch.chunk(z, 4, dim=1)
  c2 = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
  h2 = torch.sigmoid(o) * torch.tanh(c2)
  return c2, h2


def lstm_decoder(feat_list, P, scope, cfg: Config):
  """graph.py:198-260.  The state is always zero: ``lstm_memory`` is never assigned (the
  tf.assign at :226 is not fetched) so both tf.cond branches at :220 yield zeros."""
  N = feat_list[0].shape[0]
  H = cfg.dim_h_lstm
  c = feat_list[0].new_zeros(N, H)
  h = feat_list[0].new_zeros(N, H)
  for feat in feat_list:
    c, h = lstm_cell(feat, c, h, P[scope + '/lstm_cell/kernel'], P[scope + '/lstm_cell/bias'])
  net = torch.relu(h @ P[scope + '/fc1/kernel'] + P[scope + '/fc1/bias'])           # :229-230
  ep = {'fc1': net}
  if cfg.control_mode == 'cartesian':
    names = ['pred_cmd_ee', 'logits_cmd_grp', 'pred_aux_ee', 'pred_aux_obj']
  elif cfg.control_mode == 'velocity':
    names = ['pred_cmd_vel', 'pred_cmd_ee', 'pred_cmd_grp', 'pred_aux_ee', 'pred_aux_obj']
  else:
    raise ValueError("Unknown control mode '%s'" % (cfg.control_mode,))
  for n in names:
    ep[n] = net @ P['%s/%s/kernel' % (scope, n)] + P['%s/%s/bias' % (scope, n)]
  return ep


def e2e_vmc(frames, jnt_states, P, cfg: Config, collect=None, tap=None):
  """graph.py:268-319, scope 'VMC'."""
  K = cfg.window_size
  feats = []
  for k in range(K):
    f = conv_encoder(frames[:, k], P, 'VMC/ConvEncoder', collect if k == K - 1 else None, tap=tap)
    feats.append(state_concatenation(f, jnt_states[:, k]))
  return lstm_decoder(feats, P, 'VMC/LSTMDecoder', cfg)


def goal_e2evmc(frames, jnt_states, tgt_frame, P, cfg: Config, collect=None, tap=None):
  """graph.py:321-416, scope 'GoalVMC'."""
  root = 'GoalVMC'
  K = cfg.window_size
  ep = {}
  if cfg.proc_tgt in ('constant', 'residual'):
    tgt_feat = conv_encoder(tgt_frame, P, root + '/ConvEncoder', tap=tap)          # :354
  elif cfg.proc_tgt == 'dyndiff':
    pass
  else:
    raise ValueError("Unknown processing mode for target image: %s!" % (cfg.proc_tgt,))
  feats = []
  if cfg.proc_obs == 'sequence':
    for k in range(K):
      frame, jnt = frames[:, k], jnt_states[:, k]
      feat = conv_encoder(frame, P, root + '/ConvEncoder', tap=tap)
      if cfg.proc_tgt == 'constant':
        st = representation_concatenation(feat, tgt_feat, jnt)
      elif cfg.proc_tgt == 'residual':
        st = state_concatenation(tgt_feat - feat, jnt)                              # :369-370
      else:
        dd = dynimg(torch.stack([frame, tgt_frame], dim=1))                        # :373-376
        ep['dyndiff'] = dd
        tf_ = conv_encoder(dd, P, root + '/DynDiffEncoder', tap=tap)
        st = representation_concatenation(feat, tf_, jnt)
      feats.append(st)
  elif cfg.proc_obs == 'dynimg':
    frame, jnt = frames[:, -1], jnt_states[:, -1]                                   # :387-388
    feat = conv_encoder(frame, P, root + '/ConvEncoder', collect, tap=tap)          # :390
    db = dynimg(frames)                                                             # :392
    ep['dynbuff'] = db
    dyn_feat = conv_encoder(db, P, root + '/DynBuffEncoder', collect, tap=tap)      # :394
    dd = dynimg(torch.stack([frame, tgt_frame], dim=1))                            # :397-400
    ep['dyndiff'] = dd
    tf_ = conv_encoder(dd, P, root + '/DynDiffEncoder', collect, tap=tap)           # :402
    feats.append(representation_concatenation_v2(feat, dyn_feat, jnt, tf_))         # :405-407
  else:
    raise ValueError("Unknown processing mode for frame buffer: %s!" % (cfg.proc_obs,))
  ep.update(lstm_decoder(feats, P, root + '/LSTMDecoder', cfg))
  return ep


# --------------------------------------------------------------------------------------
# losses: graph.py:430-500 ; model_fn: estimator.py:14-141 / 144-279
# --------------------------------------------------------------------------------------
def mse(pred, tgt):
  """tf.losses.mean_squared_error, SUM_BY_NONZERO_WEIGHTS == mean over all elements."""
  return ((pred - tgt) ** 2).mean()


def softmax_xent(logits, labels_int, depth):
  """tf.losses.softmax_cross_entropy(one_hot(labels)) == mean over N (graph.py:468-476)."""
  lse = torch.logsumexp(logits, dim=1)
  picked = logits.gather(1, labels_int.long().reshape(-1, 1)).reshape(-1)
  return (lse - picked).mean()


def build_targets(features, labels, cfg: Config):
  """estimator.py:206-216 / 230-236."""
  if cfg.control_mode == 'cartesian':
    cmd = labels['cmd']
    grp = torch.round(cmd[:, 3]).to(torch.int32) + 1      # tf.math.rint == round-half-even == torch.round
    return {'cmd_ee': cmd[:, :3], 'cmd_grp': grp,
            'pos_ee': features['ee_state'][:, -1, :3], 'pos_obj': features['obj_state'][:, -1, :3]}
  return {'cmd_vel': labels['vel_target'], 'cmd_ee': labels['ee_target'][:, :3],
          'cmd_grp': labels['grp_target'],
          'pos_ee': features['ee_state'][:, -1, :3], 'pos_obj': features['obj_state'][:, -1, :3]}


def model_forward(features, P, cfg: Config, goal: bool, collect=None, tap=None):
  """Feature decode + graph + predictions dict (estimator.py:28-61 / 159-197)."""
  if cfg.img_channels == 3:
    obs = features['rgb']
    tgt = features.get('target_rgb') if goal else None
  elif cfg.img_channels == 4:
    obs = torch.cat([features['rgb'], features['depth']], dim=-1)
    tgt = torch.cat([features['target_rgb'], features['target_depth']], dim=-1) if goal else None
  else:
    raise ValueError("Unsupported number of channels for input frame: %d!" % cfg.img_channels)
  jnt = features['jnt_state']
  ep = goal_e2evmc(obs, jnt, tgt, P, cfg, collect, tap) if goal else e2e_vmc(obs, jnt, P, cfg, collect, tap)
  if cfg.control_mode == 'cartesian':
    pred = {'cmd_ee': ep['pred_cmd_ee'], 'logits_cmd_grp': ep['logits_cmd_grp'],
            'pos_ee': ep['pred_aux_ee'], 'pos_obj': ep['pred_aux_obj']}
  else:
    pred = {'cmd_vel': ep['pred_cmd_vel'], 'cmd_ee': ep['pred_cmd_ee'], 'cmd_grp': ep['pred_cmd_grp'],
            'pos_ee': ep['pred_aux_ee'], 'pos_obj': ep['pred_aux_obj']}
  return pred, ep


def model_loss(pred, targets, P, cfg: Config):
  """estimator.py:198-240: loss = (L_cmd_ee + L_cmd_grp) + lambda_aux (L_pos_ee + L_pos_obj) + L_reg."""
  parts = {}
  if cfg.control_mode == 'cartesian':
    parts['loss_cmd_ee'] = mse(pred['cmd_ee'], targets['cmd_ee'])
    parts['loss_cmd_grp'] = softmax_xent(pred['logits_cmd_grp'], targets['cmd_grp'], cfg.num_grp_states)
    parts['loss_pos_ee'] = mse(pred['pos_ee'], targets['pos_ee'])
    parts['loss_pos_obj'] = mse(pred['pos_obj'], targets['pos_obj'])
    loss = (parts['loss_cmd_ee'] + parts['loss_cmd_grp']) \
        + cfg.lambda_aux * (parts['loss_pos_ee'] + parts['loss_pos_obj'])
  else:
    loss = 0.0
    for k in ['cmd_vel', 'cmd_ee', 'cmd_grp', 'pos_ee', 'pos_obj']:               # graph.py:430-450
      parts['loss_' + k] = mse(pred[k], targets[k])
      loss = loss + parts['loss_' + k]
  if cfg.l2_regularizer > 0.0:
    # tf.contrib.layers.l2_regularizer(scale): scale * sum(v^2)/2 for every variable of the scope
    reg = sum((v ** 2).sum() for v in P.values()) * (0.5 * cfg.l2_regularizer)
  else:
    reg = torch.zeros((), dtype=loss.dtype)
  parts['loss_reg'] = reg
  return loss + reg, parts


def adam_step_tf(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
  """tf.train.AdamOptimizer [TF1.15]: lr_t = lr sqrt(1-b2^t)/(1-b1^t); theta -= lr_t m/(sqrt(v)+eps).
  ``step`` is the 1-based update count. In-place on numpy/torch arrays; returns lr_t.
  Where epsilon sits is published twice by TF 1.15 itself: the class docstring ("... uses the formulation just before Section 2.1
  of the Kingma and Ba paper rather than the formulation in Algorithm 1, the 'epsilon' referred to here is 'epsilon hat'") and the
  numpy reference its own unit test compares every step against (tensorflow/python/training/adam_test.py, adam_update_numpy:
  ``param - alpha_t * m_t / (np.sqrt(v_t) + epsilon)``); tests/test_oracle_kat.py runs that test's protocol."""
  lr_t = lr * math.sqrt(1.0 - b2 ** step) / (1.0 - b1 ** step)
  m *= b1; m += (1.0 - b1) * g
  v *= b2; v += (1.0 - b2) * g * g
  p -= lr_t * m / (v ** 0.5 + eps)
  return lr_t


class OracleTrainer:
  """Stateful restatement of Estimator.train's hot loop (one ``session.run(train_op)`` per call)."""

  def __init__(self, cfg: Config, goal: bool, params_np, dtype=torch.float32):
    self.cfg, self.goal, self.dtype = cfg, goal, dtype
    self.P = collections.OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype)) for k, v in params_np.items())
    self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
    self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
    self.global_step = 0

  def _cast(self, d):
    return {k: (torch.as_tensor(np.asarray(v)).to(self.dtype) if torch.as_tensor(np.asarray(v)).is_floating_point()
                else torch.as_tensor(np.asarray(v))) for k, v in d.items()}

  def loss_and_grads(self, features, labels, collect=None, tap=None):
    """``tap`` (a ``ReluTap``, test instrumentation): gradients under somebody else's ReLU decisions -- every
    proc_obs x proc_tgt branch and the K-step e2e_vmc; ``tap.stats`` then says where those decisions differ from this
    restatement's own.  Without it this is the plain restatement."""
    features, labels = self._cast(features), self._cast(labels)
    for p in self.P.values():
      p.requires_grad_(True); p.grad = None
    pred, ep = model_forward(features, self.P, self.cfg, self.goal, collect, tap)
    loss, parts = model_loss(pred, build_targets(features, labels, self.cfg), self.P, self.cfg)
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in self.P.items()}
    for p in self.P.values():
      p.requires_grad_(False)
    return loss.detach(), {k: v.detach() for k, v in parts.items()}, grads, {k: v.detach() for k, v in pred.items()}, ep

  def train_step(self, features, labels):
    loss, parts, grads, pred, _ = self.loss_and_grads(features, labels)
    self.global_step += 1
    with torch.no_grad():
      for k in self.P:
        adam_step_tf(self.P[k], grads[k], self.m[k], self.v[k], self.global_step, self.cfg.lr)
    return float(loss), {k: float(v) for k, v in parts.items()}


# --------------------------------------------------------------------------------------
# synthetic batch -- numpy, seeded; shared by tests and bench's cpu leg
# --------------------------------------------------------------------------------------
def synthetic_batch(cfg: Config, goal: bool, N: int, seed=1234, H=None, W=None):
  H = H or cfg.img_height; W = W or cfg.img_width
  K = cfg.window_size
  r = np.random.default_rng(seed)
  f = {
      'rgb': r.random([N, K, H, W, 3], dtype=np.float32),
      'jnt_state': r.standard_normal([N, K, 7]).astype(np.float32),
      'ee_state': (r.random([N, K, 7], dtype=np.float32) * 1.5),
      'obj_state': (r.random([N, K, 7], dtype=np.float32) * 1.5),
      'step': (np.arange(K)[None, :] + r.integers(1, 80, size=[N, 1])).astype(np.int64),
  }
  if cfg.img_channels == 4:
    f['depth'] = (0.5 + 2.5 * r.random([N, K, H, W, 1], dtype=np.float32))
  if goal:
    f['target_rgb'] = r.random([N, H, W, 3], dtype=np.float32)
    if cfg.img_channels == 4:
      f['target_depth'] = (0.5 + 2.5 * r.random([N, H, W, 1], dtype=np.float32))
  cmd = np.concatenate([0.3 * r.standard_normal([N, 3]), r.integers(-1, 2, size=[N, 1]).astype(np.float64)], axis=1)
  l = {'cmd': cmd.astype(np.float32),
       'vel_target': r.standard_normal([N, 7]).astype(np.float32),
       'ee_target': r.random([N, 7], dtype=np.float32),
       'grp_target': r.random([N, 2], dtype=np.float32)}
  return f, l


# --------------------------------------------------------------------------------------
# chunked evaluation for the full BASELINE shapes (same mathematics, bounded memory)
# --------------------------------------------------------------------------------------
def _encoder_jobs(features, cfg: Config, goal: bool, dtype):
  """(scope, frames [F,H,W,C]) per encoder pass of model_forward, in the order whose features
  feed the decoder.  e2e_vmc: time-major frames (graph.py:310-313); goal/dynimg: graph.py:386-402."""
  if cfg.img_channels == 3:
    obs = features['rgb']
    tgt = features.get('target_rgb') if goal else None
  else:
    obs = torch.cat([features['rgb'], features['depth']], dim=-1)
    tgt = torch.cat([features['target_rgb'], features['target_depth']], dim=-1) if goal else None
  N, K = obs.shape[0], obs.shape[1]
  if not goal:
    return [('VMC/ConvEncoder', obs.transpose(0, 1).reshape((K * N,) + tuple(obs.shape[2:])))], {}
  if cfg.proc_obs != 'dynimg':
    raise NotImplementedError('chunked oracle: goal model only for proc_obs=dynimg (the sequence branches are small-shape '
                              'cases: OracleTrainer.loss_and_grads(tap=ReluTap(...)) gives their mask-consistent gradients)')
  cur = obs[:, -1]
  db = dynimg(obs)
  dd = dynimg(torch.stack([cur, tgt], dim=1))
  return ([('GoalVMC/ConvEncoder', cur), ('GoalVMC/DynBuffEncoder', db), ('GoalVMC/DynDiffEncoder', dd)],
          {'dynbuff': db, 'dyndiff': dd})


def loss_and_grads_chunked(trainer: 'OracleTrainer', features, labels, chunk=16, enc_dtype=None, encoder_inputs=None,
                           masks_fn=None, plain_grads=False, progress=None):
  """Same result as ``OracleTrainer.loss_and_grads`` (up to summation order) without holding the
  autograd graph of every frame at once: (1) encoder forward per chunk without graph -> conv8
  features; (2) decoder + loss with autograd on the features; (3) per chunk, encoder forward
  again WITH graph and backward from d(loss)/d(features).  ``enc_dtype`` lets the encoder run
  in float32 while decoder/loss stay in the trainer's dtype.

  Mask-consistent mode (full-size parity tests): ``encoder_inputs`` = per encoder pass the [F,H,W,C] input frames the
  device actually fed its conv1 (its fp32 dynamic images; these are compared with this oracle's separately), and
  ``masks_fn(job, i0, i1)`` -> the device's eight ReLU decisions (bool [i1-i0, h, w, c]) for those frames.  Pass (1)
  stays the plain restatement (own ReLU decisions: loss, predictions, features are the reference's) and only COUNTS
  where the device decided differently and how large |z| was there (``ep['relu_disagreements']``); pass (3) computes
  the gradient under the device's decisions (``conv_encoder(force=True)``), which removes the one effect that makes
  fp32 gradients of this graph incomparable at full size: a pre-activation of 1e-8 rounded to the other side of zero
  moves a filter gradient by 1e-3 of its maximum in ANY fp32 implementation.  ``plain_grads`` (with ``masks_fn``): pass
  (3) also runs the backward under this restatement's OWN decisions -> ``ep['plain_grads']`` (the loose, mask-independent
  backstop of the full-size test).  ``ep['conv8']`` = the plain restatement's features of EVERY frame per encoder.
  ``progress(text)`` is called once per chunk (long runs must show signs of life)."""
  cfg, goal, dt = trainer.cfg, trainer.goal, trainer.dtype
  edt = enc_dtype or dt
  features, labels = trainer._cast(features), trainer._cast(labels)
  P = trainer.P
  jobs, ep = _encoder_jobs(features, cfg, goal, dt)
  if encoder_inputs is not None:
    assert len(encoder_inputs) == len(jobs) and all(tuple(a.shape) == tuple(x.shape) for a, (_, x) in zip(encoder_inputs, jobs))
    jobs = [(scope, a) for a, (scope, _) in zip(encoder_inputs, jobs)]
  Pe = {k: v.detach().to(edt) for k, v in P.items() if '/conv' in k}
  feats = []
  stats = [[[0, 0.0, 0] for _ in range(8)] for _ in jobs] if masks_fn is not None else None
  with torch.no_grad():
    for j, (scope, x) in enumerate(jobs):
      outs = []
      for i in range(0, x.shape[0], chunk):
        i1 = min(i + chunk, x.shape[0])
        kw = dict(masks=masks_fn(j, i, i1), stats=stats[j]) if masks_fn is not None else {}
        outs.append(conv_encoder(x[i:i1].to(edt), Pe, scope, **kw))
        if progress is not None:
          progress('forward %s frames %d-%d of %d' % (scope, i, i1, x.shape[0]))
      feats.append(torch.cat(outs, dim=0).to(dt))
  first_last = {scope: (f[0].clone(), f[-1].clone()) for (scope, _), f in zip(jobs, feats)}
  conv8_all = {scope: f.detach().clone() for (scope, _), f in zip(jobs, feats)}
  for f in feats:
    f.requires_grad_(True)
  dec = {k: v for k, v in P.items() if '/conv' not in k}
  for p in dec.values():
    p.requires_grad_(True); p.grad = None
  jnt = features['jnt_state']
  N, K = jnt.shape[0], jnt.shape[1]
  if goal:
    st = [representation_concatenation_v2(feats[0], feats[1], jnt[:, -1], feats[2])]
    dep = lstm_decoder(st, P, 'GoalVMC/LSTMDecoder', cfg)
  else:
    f = feats[0].reshape((K, N) + tuple(feats[0].shape[1:]))
    st = [state_concatenation(f[k], jnt[:, k]) for k in range(K)]
    dep = lstm_decoder(st, P, 'VMC/LSTMDecoder', cfg)
  if cfg.control_mode == 'cartesian':
    pred = {'cmd_ee': dep['pred_cmd_ee'], 'logits_cmd_grp': dep['logits_cmd_grp'],
            'pos_ee': dep['pred_aux_ee'], 'pos_obj': dep['pred_aux_obj']}
  else:
    pred = {'cmd_vel': dep['pred_cmd_vel'], 'cmd_ee': dep['pred_cmd_ee'], 'cmd_grp': dep['pred_cmd_grp'],
            'pos_ee': dep['pred_aux_ee'], 'pos_obj': dep['pred_aux_obj']}
  if cfg.l2_regularizer > 0.0:
    raise NotImplementedError('chunked oracle: l2_regularizer must be 0')
  loss, parts = model_loss(pred, build_targets(features, labels, cfg), P, cfg)
  loss.backward()
  grads = {k: p.grad.detach().clone() for k, p in dec.items()}
  for p in dec.values():
    p.requires_grad_(False)
  plain = {} if (plain_grads and masks_fn is not None) else None
  for j, ((scope, x), f) in enumerate(zip(jobs, feats)):
    names = [k for k in P if k.startswith(scope + '/')]
    Pg = {k: P[k].detach().to(edt).requires_grad_(True) for k in names}
    Pp = {k: P[k].detach().to(edt).requires_grad_(True) for k in names} if plain is not None else None
    for i in range(0, x.shape[0], chunk):
      i1 = min(i + chunk, x.shape[0])
      kw = dict(masks=masks_fn(j, i, i1), force=True) if masks_fn is not None else {}
      out = conv_encoder(x[i:i1].to(edt), Pg, scope, **kw)
      out.backward(f.grad[i:i1].to(edt))
      if Pp is not None:
        conv_encoder(x[i:i1].to(edt), Pp, scope).backward(f.grad[i:i1].to(edt))
      if progress is not None:
        progress('backward %s frames %d-%d of %d' % (scope, i, i1, x.shape[0]))
    for k in names:
      grads[k] = Pg[k].grad.detach().to(dt)
      if Pp is not None:
        plain[k] = Pp[k].grad.detach().to(dt)
  ep = dict(ep)
  ep['conv8_first_last'] = first_last
  ep['conv8'] = conv8_all
  if plain is not None:
    ep['plain_grads'] = plain
  if stats is not None:
    ep['relu_disagreements'] = {scope: [tuple(s) for s in st] for (scope, _), st in zip(jobs, stats)}
  return (loss.detach(), {k: v.detach() for k, v in parts.items()}, grads, {k: v.detach() for k, v in pred.items()}, ep)
