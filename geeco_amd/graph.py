"""Model graphs of the E2E-VMC controllers on the HIP kernels.

Counterpart of the reference's ``src/models/e2evmc/graph.py``: ``dynimg`` (:30-55), ``conv_encoder``
(:61-117), ``state_concatenation`` / ``representation_concatenation[_v2]`` (:123-192),
``lstm_decoder`` (:198-260), ``e2e_vmc`` (:268-319), ``goal_e2evmc`` (:321-416) and the loss
functions (:430-500).  The reference builds a TF graph once and runs it per batch; here a model
object owns static HBM buffers (inputs, activations, gradients) for a fixed batch size and
``forward()`` / ``backward()`` enqueue the same sequence of HIP kernels every step, so a whole
train step can be captured into a hipGraph (geeco_amd/estimator.py).

Data layout in HBM: NHWC fp32 activations; the G encoders of a model ("ConvEncoder",
"DynBuffEncoder", "DynDiffEncoder") are stacked along a leading group axis and each layer is ONE
launch over all groups; RGB inputs are channel-padded to 4 so conv1 gathers float4 pixels.
"""
from __future__ import annotations

import collections
import os

import torch

from . import ops
from .variables import ENC_FILTERS, ENC_STRIDES, VariableStore, decoder_shapes, encoder_shapes

_CELLS = 4   # the reference hard-codes the 2x2 tiling of the joint state (graph.py:139,163,188)


def model_variable_shapes(cfg, goal: bool):
  """Variable creation order of ``e2e_vmc`` (graph.py:268-319) / ``goal_e2evmc`` (graph.py:321-416)."""
  C, jn = cfg.img_channels, cfg.dim_jnt_state
  if C not in (3, 4):
    raise ValueError("Unsupported number of channels for input frame: %d!" % C)
  s = collections.OrderedDict()
  if not goal:
    s.update(encoder_shapes('VMC/ConvEncoder', C, 256))
    s.update(decoder_shapes('VMC/LSTMDecoder', _CELLS * (256 + jn), cfg))
    return s
  root = 'GoalVMC'
  if cfg.proc_tgt not in ('constant', 'residual', 'dyndiff'):
    raise ValueError("Unknown processing mode for target image: %s!" % (cfg.proc_tgt,))
  if cfg.proc_obs == 'sequence':
    s.update(encoder_shapes(root + '/ConvEncoder', C, cfg.dim_s_obs))
    if cfg.proc_tgt == 'constant':
      din = _CELLS * (cfg.dim_s_obs + jn + cfg.dim_s_obs)
    elif cfg.proc_tgt == 'residual':
      din = _CELLS * (cfg.dim_s_obs + jn)
    else:
      s.update(encoder_shapes(root + '/DynDiffEncoder', C, cfg.dim_s_diff))
      din = _CELLS * (cfg.dim_s_obs + jn + cfg.dim_s_diff)
  elif cfg.proc_obs == 'dynimg':
    s.update(encoder_shapes(root + '/ConvEncoder', C, cfg.dim_s_obs))
    s.update(encoder_shapes(root + '/DynBuffEncoder', C, cfg.dim_s_dyn))
    s.update(encoder_shapes(root + '/DynDiffEncoder', C, cfg.dim_s_diff))
    din = _CELLS * (cfg.dim_s_obs + cfg.dim_s_dyn + jn + cfg.dim_s_diff)
  else:
    raise ValueError("Unknown processing mode for frame buffer: %s!" % (cfg.proc_obs,))
  s.update(decoder_shapes(root + '/LSTMDecoder', din, cfg))
  return s


# ================================================================================================
# conv encoder stack (graph.py:61-117), G instances per launch
# ================================================================================================
class ConvEncoderStack:
  """``conv_encoder`` for G weight sets with identical shapes, Nf frames each."""

  def __init__(self, store: VariableStore, scopes, Nf, H, W, Cin, dim_out, training):
    self.store, self.scopes, self.G, self.Nf = store, list(scopes), len(scopes), Nf
    self.H, self.W, self.Cin = H, W, Cin
    self.two_streams = os.environ.get('GEECO_ONE_STREAM') is None
    self.Cpad = -(-Cin // 4) * 4
    self.training = training
    dev = store.device
    G = self.G
    # group stride inside the parameter arena (all encoders have the same shapes)
    if G > 1:
      gs = store.offsets[self.scopes[1] + '/conv1/kernel'] - store.offsets[self.scopes[0] + '/conv1/kernel']
      for g in range(G):
        for l in range(1, 9):
          for kind in ('kernel', 'bias'):
            a = store.offsets['%s/conv%d/%s' % (self.scopes[g], l, kind)]
            b = store.offsets['%s/conv%d/%s' % (self.scopes[0], l, kind)]
            if a - b != g * gs:
              raise ValueError('encoders are not uniformly strided in the arena')
      self.gs_p = gs
    else:
      self.gs_p = 0
    # layer geometry
    self.layers = []
    h, w, c = H, W, self.Cpad
    for l in range(8):
      cout = (list(ENC_FILTERS) + [dim_out])[l]
      s = ENC_STRIDES[l]
      ho, wo = ops.same_out(h, s), ops.same_out(w, s)
      self.layers.append(dict(H=h, W=w, Cin=c, Cout=cout, stride=s, Ho=ho, Wo=wo))
      h, w, c = ho, wo, cout
    self.out_hw = (h, w)
    f32 = dict(dtype=torch.float32, device=dev)
    self.x_in = torch.zeros(G, Nf, H, W, self.Cpad, **f32)
    self.acts = [torch.empty(G, Nf, L['Ho'], L['Wo'], L['Cout'], **f32) for L in self.layers]
    self.pad1 = self.Cpad != Cin
    if self.pad1:
      self.w1p = torch.zeros(G, 3, 3, self.Cpad, self.layers[0]['Cout'], **f32)
    if training:
      self.dz = [torch.empty_like(a) for a in self.acts]
      self.wt = [None] + [torch.empty(G, 3, 3, L['Cout'], L['Cin'], **f32) for L in self.layers[1:]]
      if self.pad1:
        self.dw1p = torch.zeros(G, 3, 3, self.Cpad, self.layers[0]['Cout'], **f32)
      wsb = max(ops.conv3x3_wgrad_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'])
                for L in self.layers)
      self.ws = torch.empty(wsb // 4 + 4, **f32)
      dsb = max(ops.conv3x3_dgrad_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'])
                for L in self.layers[1:])
      self.dws = torch.empty(dsb // 4 + 4, **f32)
      # wgrad(l) and dgrad(l) only share their input dz[l]: they run on two streams (two branches of
      # the captured hipGraph) so the small top layers overlap instead of leaving CUs idle in their tails
      self.side = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
    fsb = max(ops.conv3x3_fwd_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride']) for L in self.layers)
    self.fws = torch.empty(fsb // 4 + 4, **f32)

  def _w(self, l, g=0):
    return self.store.var('%s/conv%d/kernel' % (self.scopes[g], l + 1))

  def _b(self, l, g=0):
    return self.store.var('%s/conv%d/bias' % (self.scopes[g], l + 1))

  def _dw(self, l, g=0):
    return self.store.grad('%s/conv%d/kernel' % (self.scopes[g], l + 1))

  def _db(self, l, g=0):
    return self.store.grad('%s/conv%d/bias' % (self.scopes[g], l + 1))

  @property
  def features(self):
    """[G][Nf][h][w][dim_out] output of conv8 (endpoints['conv8'], graph.py:116)."""
    return self.acts[7]

  @property
  def dfeatures(self):
    return self.dz[7]

  def forward(self):
    G, Nf = self.G, self.Nf
    for l, L in enumerate(self.layers):
      x = self.x_in if l == 0 else self.acts[l - 1]
      y = self.acts[l]
      if l == 0 and self.pad1:
        for g in range(G):
          ops.pad_mid_into(self.w1p[g], self._w(0, g), 9, self.Cin, self.Cpad, L['Cout'])
        w, gs_w = self.w1p, self.w1p[0].numel()
      else:
        w, gs_w = self._w(l), self.gs_p
      ops.conv3x3_fwd_into(y, x, w, self._b(l), G, x[0].numel(), gs_w, self.gs_p, y[0].numel(), Nf, L['H'], L['W'],
                           L['Cin'], L['Cout'], L['stride'], relu=True, ws=self.fws)

  def backward(self):
    """Expects ``self.dz[7]`` = d(loss)/d(pre-activation of conv8) (ReluGrad already applied)."""
    G, Nf = self.G, self.Nf
    main = torch.cuda.current_stream()
    side = self.side if self.two_streams else None
    for l in range(7, -1, -1):
      L = self.layers[l]
      x = self.x_in if l == 0 else self.acts[l - 1]
      dz = self.dz[l]
      if l == 0 and self.pad1:
        dw, gs_dw = self.dw1p, self.dw1p[0].numel()
      else:
        dw, gs_dw = self._dw(l), self.gs_p
      if side is not None:
        side.wait_stream(main)          # dz[l] is ready
      with torch.cuda.stream(side if side is not None else main):
        ops.conv3x3_wgrad_into(dw, self._db(l), x, dz, G, x[0].numel(), dz[0].numel(), gs_dw, self.gs_p, Nf, L['H'],
                               L['W'], L['Cin'], L['Cout'], L['stride'], self.ws)
        if l == 0 and self.pad1:
          for g in range(G):
            ops.pad_mid_into(self._dw(0, g), self.dw1p[g], 9, self.Cpad, self.Cin, L['Cout'])
      if l == 0:
        break   # conv1's input is data: no dgrad
      wt = self.wt[l]
      ops.transpose_hwio_into(wt, self._w(l), G, self.gs_p, wt[0].numel(), L['Cin'], L['Cout'])
      dx = self.dz[l - 1]
      ops.conv3x3_dgrad_into(dx, dz, wt, x, G, dz[0].numel(), wt[0].numel(), dx[0].numel(), Nf, L['H'], L['W'],
                             L['Cin'], L['Cout'], L['stride'], ws=self.dws, w=self._w(l), gs_w=self.gs_p)
    if side is not None:
      main.wait_stream(side)


# ================================================================================================
# LSTM decoder + heads + losses (graph.py:198-260, 452-500; estimator.py:206-239)
# ================================================================================================
class LSTMDecoder:
  """T LSTM steps over states [T][N][D] from a zero state, fc1 + heads on the last output."""

  def __init__(self, store: VariableStore, scope, cfg, N, T, D, training):
    if cfg.control_mode != 'cartesian':
      raise NotImplementedError("control_mode '%s' is not built yet (cartesian only)" % cfg.control_mode)
    if cfg.num_grp_states != 3:
      raise NotImplementedError('num_grp_states != 3')
    self.store, self.scope, self.cfg, self.N, self.T, self.D = store, scope, cfg, N, T, D
    self.H, self.F = cfg.dim_h_lstm, cfg.dim_h_fc
    self.training = training
    dev = store.device
    f32 = dict(dtype=torch.float32, device=dev)
    H, F = self.H, self.F
    self.states = torch.empty(T, N, D, **f32)
    self.z = torch.empty(T, N, 4 * H, **f32)
    self.gates = torch.empty(T, N, 4 * H, **f32)
    self.c = torch.empty(T, N, H, **f32)
    self.h = torch.empty(T, N, H, **f32)
    self.preds = torch.empty(N, 12, **f32)
    self.losses = torch.zeros(8, **f32)
    self.heads_ws = torch.empty(ops.heads_ws_bytes(N, H, F) // 4 + 4, **f32)
    gemm_shapes = [(T * N, 4 * H, D), (N, 4 * H, H)]
    if training:
      self.dstates = torch.empty(T, N, D, **f32)
      self.dz = torch.empty(T, N, 4 * H, **f32)
      self.dh = torch.empty(N, H, **f32)
      self.dc = torch.empty(N, H, **f32)
      gemm_shapes += [(D, 4 * H, T * N), (H, 4 * H, max((T - 1) * N, 1)), (T * N, D, 4 * H), (N, H, 4 * H)]
    self.gemm_ws = torch.empty(max(ops.gemm_ws_bytes(*s) for s in gemm_shapes) // 4 + 4, **f32)
    self.head_names = ['pred_cmd_ee', 'logits_cmd_grp', 'pred_aux_ee', 'pred_aux_obj']
    # label pointers (bound by the model)
    self.cmd = self.ee_tgt = self.obj_tgt = None
    self.ee_stride = self.obj_stride = 0
    self.loss_scale = 1.0

  def _v(self, n):
    return self.store.var('%s/%s' % (self.scope, n))

  def _g(self, n):
    return self.store.grad('%s/%s' % (self.scope, n))

  def forward(self, backward_too):
    N, T, D, H, F = self.N, self.T, self.D, self.H, self.F
    W = self._v('lstm_cell/kernel')            # [D + H][4H]: rows 0..D-1 multiply x, D.. multiply h
    Wx, Wh = W[:D], W[D:]
    bias = self._v('lstm_cell/bias')
    # hoisted input projection for all steps: Z = X Wx
    ops.gemm_into(self.z, self.states, Wx, T * N, 4 * H, D, D, 4 * H, 4 * H, ws=self.gemm_ws)
    for t in range(T):
      if t > 0:
        ops.gemm_into(self.z[t], self.h[t - 1], Wh, N, 4 * H, H, H, 4 * H, 4 * H, accumulate=True, ws=self.gemm_ws)
      ops.lstm_gates_fwd_into(self.c[t], self.h[t], self.gates[t], self.z[t], bias,
                              self.c[t - 1] if t > 0 else None, N, H)
    hw = [self._v(n + '/kernel') for n in self.head_names]
    hb = [self._v(n + '/bias') for n in self.head_names]
    kw = {}
    if backward_too:
      kw = dict(dh=self.dh, d_fc1_w=self._g('fc1/kernel'), d_fc1_b=self._g('fc1/bias'),
                d_heads_w=[self._g(n + '/kernel') for n in self.head_names],
                d_heads_b=[self._g(n + '/bias') for n in self.head_names])
    ops.heads_loss_into(self.preds, self.losses, self.h[T - 1], self._v('fc1/kernel'), self._v('fc1/bias'), hw, hb,
                        self.cmd, self.ee_tgt, self.ee_stride, self.obj_tgt, self.obj_stride,
                        float(self.cfg.lambda_aux), float(self.loss_scale), N, H, F, self.heads_ws, **kw)

  def backward(self):
    """After forward(backward_too=True): fills d(states) and the LSTM variable gradients."""
    N, T, D, H = self.N, self.T, self.D, self.H
    W = self._v('lstm_cell/kernel')
    Wx, Wh = W[:D], W[D:]
    dW = self._g('lstm_cell/kernel')
    for t in range(T - 1, -1, -1):
      last = t == T - 1
      ops.lstm_gates_bwd_into(self.dz[t], self.dc if t > 0 else None, self.gates[t],
                              self.c[t - 1] if t > 0 else None, self.c[t], self.dh, None if last else self.dc, N, H)
      if t > 0:   # dh_{t-1} = dz_t Wh^T
        ops.gemm_into(self.dh, self.dz[t], Wh, N, H, 4 * H, 4 * H, 4 * H, H, tb=True, ws=self.gemm_ws)
    # dWx = X^T dZ ; dWh = H_prev^T dZ[1:] ; db = colsum(dZ) ; dX = dZ Wx^T
    ops.gemm_into(dW[:D], self.states, self.dz, D, 4 * H, T * N, D, 4 * H, 4 * H, ta=True, ws=self.gemm_ws)
    if T > 1:
      ops.gemm_into(dW[D:], self.h, self.dz[1:], H, 4 * H, (T - 1) * N, H, 4 * H, 4 * H, ta=True, ws=self.gemm_ws)
    ops.colsum_into(self._g('lstm_cell/bias'), self.dz, 4 * H, T * N, 4 * H)
    ops.gemm_into(self.dstates, self.dz, Wx, T * N, D, 4 * H, 4 * H, 4 * H, D, tb=True, ws=self.gemm_ws)


# ================================================================================================
# models
# ================================================================================================
class _ModelBase:
  """Static input buffers shared by both controllers (feed contract: geeco_gym.py:375-398)."""

  def __init__(self, cfg, N, device, goal, training, store=None):
    self.cfg, self.N, self.goal, self.training = cfg, N, goal, training
    self.device = torch.device(device)
    self.K = cfg.window_size
    self.H, self.W, self.C = cfg.img_height, cfg.img_width, cfg.img_channels
    if (ops.same_out(self.H, 128), ops.same_out(self.W, 128)) != (2, 2):
      # seven stride-2 layers must end on the hard-coded 2x2 grid (graph.py:139)
      raise ValueError('the 2x2 state tiling needs 129..256 pixel inputs, got %dx%d' % (self.H, self.W))
    self.store = store or VariableStore(model_variable_shapes(cfg, goal), self.device)
    f32 = dict(dtype=torch.float32, device=self.device)
    N, K, H, W = self.N, self.K, self.H, self.W
    self.inputs = {
        'rgb': torch.zeros(N, K, H, W, 3, **f32),
        'jnt_state': torch.zeros(N, K, cfg.dim_jnt_state, **f32),
        'ee_state': torch.zeros(N, K, 7, **f32),
        'obj_state': torch.zeros(N, K, 7, **f32),
        'cmd': torch.zeros(N, 4, **f32),
    }
    if self.C == 4:
      self.inputs['depth'] = torch.zeros(N, K, H, W, 1, **f32)
    if goal:
      self.inputs['target_rgb'] = torch.zeros(N, H, W, 3, **f32)
      if self.C == 4:
        self.inputs['target_depth'] = torch.zeros(N, H, W, 1, **f32)
    self.scal = torch.zeros(4, **f32)          # [0] = Adam lr_t
    self.world = 1

  def load_batch(self, features, labels=None):
    """Copies one batch into the static input buffers (H2D or D2D; torch is plumbing here)."""
    for k, buf in self.inputs.items():
      src = labels.get(k) if (labels is not None and k == 'cmd') else features.get(k)
      if src is None:
        if k == 'cmd':
          continue
        raise KeyError("missing feature '%s'" % k)
      src = torch.as_tensor(src)
      if tuple(src.shape) != tuple(buf.shape):
        raise ValueError("feature '%s': expected shape %s, got %s" % (k, tuple(buf.shape), tuple(src.shape)))
      buf.copy_(src, non_blocking=True)

  def _bind_labels(self):
    K = self.K
    d = self.decoder
    d.cmd = self.inputs['cmd']
    d.ee_tgt = self.inputs['ee_state'][:, K - 1]      # features['ee_state'][:, -1, :3]  (estimator.py:209)
    d.obj_tgt = self.inputs['obj_state'][:, K - 1]
    d.ee_stride = d.obj_stride = K * 7

  # -- optimiser step (estimator.py:243-244) -------------------------------------------------
  def apply_gradients(self):
    s, cfg = self.store, self.cfg
    ops.adam_prepare(s.global_step, float(cfg.lr), self.scal)
    ops.adam_tf(s.params, s.grads, s.adam_m, s.adam_v, s.size, self.scal, grad_scale=1.0 / self.world,
                l2=float(cfg.l2_regularizer))

  def predictions(self):
    """estimator.py:183-189."""
    p = self.decoder.preds
    return {'cmd_ee': p[:, 0:3], 'logits_cmd_grp': p[:, 3:6], 'pos_ee': p[:, 6:9], 'pos_obj': p[:, 9:12]}

  @property
  def loss(self):
    """Device scalar: total loss of the last forward (local batch mean; without the L2 term)."""
    return self.decoder.losses[0]

  def loss_parts(self):
    l = self.decoder.losses
    return {'loss': l[0], 'loss_cmd_ee': l[1], 'loss_cmd_grp': l[2], 'loss_pos_ee': l[3], 'loss_pos_obj': l[4]}

  def train_step(self):
    self.forward(backward_too=True)
    self.backward()
    self.apply_gradients()


class GoalE2EVMC(_ModelBase):
  """``goal_e2evmc`` (graph.py:321-416), proc_obs='dynimg' branch (geeco-f; :386-407)."""

  def __init__(self, cfg, N, device, training=True, store=None):
    super().__init__(cfg, N, device, goal=True, training=training, store=store)
    if cfg.proc_obs != 'dynimg':
      if cfg.proc_obs == 'sequence':
        raise NotImplementedError("proc_obs='sequence' for goal_e2evmc is not built yet")
      raise ValueError("Unknown processing mode for frame buffer: %s!" % (cfg.proc_obs,))
    if cfg.proc_tgt not in ('constant', 'residual', 'dyndiff'):
      raise ValueError("Unknown processing mode for target image: %s!" % (cfg.proc_tgt,))
    if not (cfg.dim_s_obs == cfg.dim_s_dyn == cfg.dim_s_diff):
      raise NotImplementedError('dim_s_obs, dim_s_dyn and dim_s_diff must be equal (grouped encoders)')
    root = 'GoalVMC'
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    self.enc = ConvEncoderStack(self.store, [root + '/ConvEncoder', root + '/DynBuffEncoder', root + '/DynDiffEncoder'],
                                N, H, W, C, cfg.dim_s_obs, training)
    self.feat_ch = [cfg.dim_s_obs, cfg.dim_s_dyn, cfg.dim_s_diff]
    D = _CELLS * (sum(self.feat_ch) + cfg.dim_jnt_state)
    self.decoder = LSTMDecoder(self.store, root + '/LSTMDecoder', cfg, N, 1, D, training)
    self._bind_labels()
    f32 = dict(dtype=torch.float32, device=self.device)
    self.dyn_ws = ops.dynimg_ws(N, H * W * max(C, 4), self.device)
    if C == 4:
      self.obs4 = torch.empty(N, K, H, W, 4, **f32)      # rgb || depth (estimator.py:169)
      self.tgt4 = torch.empty(N, H, W, 4, **f32)         # target_rgb || target_depth (estimator.py:172)

  def forward(self, backward_too=False):
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    HW = H * W
    x_in = self.enc.x_in
    if C == 3:
      frames, tgt = self.inputs['rgb'], self.inputs['target_rgb']
    else:
      ops.pack_pixels_into(self.obs4, self.inputs['rgb'], HW * 3, N * K, HW, 3, 4, self.inputs['depth'], HW, 1)
      ops.pack_pixels_into(self.tgt4, self.inputs['target_rgb'], HW * 3, N, HW, 3, 4, self.inputs['target_depth'], HW, 1)
      frames, tgt = self.obs4, self.tgt4
    cur = frames[:, K - 1]                                  # rgb_frame_list[-1] (graph.py:387)
    # g0: current frame;  g1: dynimg(buffer) (:392);  g2: dynimg([cur, tgt]) (:397-400)
    ops.pack_pixels_into(x_in[0], cur, K * HW * C, N, HW, C, 4)
    ops.dynimg_into(x_in[1], frames, K, N, HW, C, 4, self.dyn_ws, K * HW * C, HW * C)
    ops.dynimg_into(x_in[2], cur, 2, N, HW, C, 4, self.dyn_ws, K * HW * C, 0, frames2=tgt)
    self.enc.forward()
    feats = self.enc.features                               # [3][N][2][2][256]
    jn = self.cfg.dim_jnt_state
    jnt = self.inputs['jnt_state'][:, K - 1]                # jnt_state_list[-1] (graph.py:388)
    d = self.decoder
    # representation_concatenation_v2: [obs | dyn | jnt | tgt] (graph.py:169-192)
    ops.state_concat_fwd_into(d.states[0], [feats[0], feats[1], feats[2]], self.feat_ch, 2, jnt, K * jn, jn, N, _CELLS,
                              d.D)
    d.forward(backward_too)

  def backward(self):
    d = self.decoder
    d.backward()
    feats, dfe = self.enc.features, self.enc.dfeatures
    ops.state_concat_bwd_into([dfe[0], dfe[1], dfe[2]], d.dstates[0], d.D, [feats[0], feats[1], feats[2]], self.feat_ch,
                              2, self.cfg.dim_jnt_state, self.N, _CELLS)
    self.enc.backward()

  def endpoints(self):
    """dynbuff / dyndiff debug endpoints (graph.py:393,401)."""
    return {'dynbuff': self.enc.x_in[1][..., :self.C], 'dyndiff': self.enc.x_in[2][..., :self.C],
            'conv8': self.enc.features}


class E2EVMC(_ModelBase):
  """``e2e_vmc`` (graph.py:268-319): per-frame encoder, K LSTM steps (scope 'VMC')."""

  def __init__(self, cfg, N, device, training=True, store=None):
    super().__init__(cfg, N, device, goal=False, training=training, store=store)
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    # frames are processed time-major ([K][N]) so that step t's features are one dense block
    self.enc = ConvEncoderStack(self.store, ['VMC/ConvEncoder'], K * N, H, W, C, 256, training)
    D = _CELLS * (256 + cfg.dim_jnt_state)
    self.decoder = LSTMDecoder(self.store, 'VMC/LSTMDecoder', cfg, N, K, D, training)
    self._bind_labels()

  def forward(self, backward_too=False):
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    HW = H * W
    x_in = self.enc.x_in[0].view(K, N, H, W, 4)
    rgb = self.inputs['rgb']
    dep = self.inputs.get('depth')
    for t in range(K):
      if C == 3:
        ops.pack_pixels_into(x_in[t], rgb[:, t], K * HW * 3, N, HW, 3, 4)
      else:
        ops.pack_pixels_into(x_in[t], rgb[:, t], K * HW * 3, N, HW, 3, 4, dep[:, t], K * HW, 1)
    self.enc.forward()
    feats = self.enc.features[0].view(K, N, _CELLS, 256)
    jn = self.cfg.dim_jnt_state
    d = self.decoder
    for t in range(K):   # state_concatenation (graph.py:123-144)
      ops.state_concat_fwd_into(d.states[t], [feats[t]], [256], 1, self.inputs['jnt_state'][:, t], K * jn, jn, N,
                                _CELLS, d.D)
    d.forward(backward_too)

  def backward(self):
    N, K = self.N, self.K
    d = self.decoder
    d.backward()
    feats = self.enc.features[0].view(K, N, _CELLS, 256)
    dfe = self.enc.dfeatures[0].view(K, N, _CELLS, 256)
    for t in range(K):
      ops.state_concat_bwd_into([dfe[t]], d.dstates[t], d.D, [feats[t]], [256], 1, self.cfg.dim_jnt_state, N, _CELLS)
    self.enc.backward()

  def endpoints(self):
    return {'conv8': self.enc.features}
