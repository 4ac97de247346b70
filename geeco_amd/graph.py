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

from . import _dev, _native, ops
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
    """``dim_out``: conv8's output channels, one int or one per encoder (dim_s_obs / dim_s_dyn / dim_s_diff,
    graph.py:390,394,402).  With unequal values conv1..conv7 still run as grouped launches and conv8 (the only layer
    whose shape differs) runs once per encoder (``split_top``)."""
    self.store, self.scopes, self.G, self.Nf = store, list(scopes), len(scopes), Nf
    self.H, self.W, self.Cin = H, W, Cin
    self.late = None       # (staging buffer, per-encoder length): see redirect_late_gradients
    # CUs the two persistent bottom-of-the-backward launches leave to a collective running beside them: a launch argument
    # (runtime.TrainStepRunner sets it around its own part 2, so every runner captures the grids it was built for)
    self.reserved_cus = 0
    self.dim_outs = [int(d) for d in dim_out] if isinstance(dim_out, (list, tuple)) else [int(dim_out)] * len(self.scopes)
    self.split_top = len(set(self.dim_outs)) > 1
    dim_out = max(self.dim_outs)
    # Backward schedule: ONE stream by default.  Round 1 ran the filter-gradient launches of the upper layers on two side
    # streams beside the input-gradient chain (+1-2 % then: the gather wgrad kernel left MFMA slack for its neighbour);
    # with the LDS-staged wgrad kernels every big launch fills the chip by itself and the two schedules measure the same
    # (3.717 vs 3.719 ms), so the simpler one is the default.  GEECO_MULTI_STREAM=1 restores the side streams.
    self.two_streams = _dev.env('GEECO_MULTI_STREAM') is not None
    # conv7's + conv8's filter gradients in one launch (GEECO_NO_WGRAD_PAIR: two)
    self.pair_top = _dev.env('GEECO_NO_WGRAD_PAIR') is None
    # ... and conv7's input gradient in the same grid (GEECO_NO_TOP_BWD: its own launch)
    # (GEECO_TOP_BWD=2: two heterogeneous launches instead, conv8's pair then conv7's)
    self.hetero_top = 0 if _dev.env('GEECO_NO_TOP_BWD') is not None else int(_dev.env('GEECO_TOP_BWD', '1'))
    self.w8_done = False
    # the filter-gradient kernels' slab sums of a backward part go into one launch (GEECO_NO_BATCH_REDUCE: one per layer)
    self.batch_reduce = _dev.env('GEECO_NO_BATCH_REDUCE') is None
    self.derived_version = -1
    # Only the FIRST training stack built on a store may rely on the post-Adam refresh of its derived
    # weight copies; eval / predict stacks and any later training stack (e.g. the model built for a
    # ragged final batch) share the parameters but not the copies, so they re-derive on every forward.
    self.lazy_refresh = (training and getattr(store, 'primary_stack', None) is None and
                         _dev.env('GEECO_EAGER_DERIVED') is None)
    if self.lazy_refresh:
      store.primary_stack = self
    self.Cpad = -(-Cin // 4) * 4
    self.training = training
    dev = store.device
    G = self.G
    # group stride inside the parameter arena (all encoders have the same shapes)
    if G > 1:
      gs = store.offsets[self.scopes[1] + '/conv1/kernel'] - store.offsets[self.scopes[0] + '/conv1/kernel']
      for g in range(G):
        for l in range(1, 8 if self.split_top else 9):
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
    if self.split_top:      # per-encoder conv8 outputs of different widths
      L7 = self.layers[7]
      self.acts[7] = [torch.empty(Nf, L7['Ho'], L7['Wo'], d, **f32) for d in self.dim_outs]
    self.pad1 = self.Cpad != Cin
    self.pad1_copy = self.pad1      # the channel-padded copy of conv1's kernel is kept up to date (see below)
    if self.pad1:
      self.w1p = torch.zeros(G, 3, 3, self.Cpad, self.layers[0]['Cout'], **f32)
    if training:
      # encoder bottom fused backward (conv2 dgrad + conv1 wgrad): the reference encoder's shapes, even sizes
      L0, L1 = self.layers[0], self.layers[1]
      self.fused_bottom = (_dev.env('GEECO_NO_FUSED_BOTTOM') is None and _dev.env('GEECO_NO_HALO') is None
                           and self.Cpad == 4 and self.Cin in (3, 4) and L0['Cout'] == 32 and L0['stride'] == 1
                           and L1['Cout'] == 48 and L1['stride'] == 2 and L1['H'] % 2 == 0 and L1['W'] % 2 == 0)
      # the fused bottom only needs the SIGN of conv1's output (ReluGrad): conv1's forward writes one bit word per pixel
      # next to y1 and the backward reads those 25 MB instead of the 805 MB of y1 (GEECO_NO_RELU_BITS: read y1)
      self.relu_bits = self.fused_bottom and _dev.env('GEECO_NO_RELU_BITS') is None
      # with the fused bottom and the sign bits nothing reads the channel-padded copy of conv1's kernel any more: conv1's
      # forward takes the RGB variable itself (together with the gather GEMM reading HWIO kernels this leaves NO weight
      # copy to re-derive after Adam: one launch less per step)
      if self.relu_bits and self.Cin == 3 and _dev.env('GEECO_PAD1_COPY') is None:
        self.pad1_copy = False
      if self.relu_bits:
        self.bits1 = torch.zeros(G, Nf, ops.relu_bits_rows(L0['H']), ops.relu_bits_pitch(L0['W']), dtype=torch.int32, device=dev)
      # the same one layer up: conv2's forward leaves 16-bit sign fields of y2 for conv3's input-gradient kernel
      L2 = self.layers[2]
      self.relu_fields = (_dev.env('GEECO_NO_RELU_BITS') is None and _dev.env('GEECO_NO_HALO') is None
                          and _dev.env('GEECO_NO_HALO3') is None and _dev.env('GEECO_HALO_WS') is None
                          and (L1['Cin'], L1['Cout'], L1['stride']) == (32, 48, 2)
                          and (L2['Cin'], L2['Cout'], L2['stride']) == (48, 64, 2)
                          and L1['H'] % 2 == 0 and L1['W'] % 2 == 0 and L2['H'] % 2 == 0 and L2['W'] % 2 == 0)
      if self.relu_fields:
        self.fields2 = torch.zeros(G, ops.relu_fields_elems(Nf, L2['H'], L2['W']), dtype=torch.int16, device=dev)
      # ... and conv3's forward leaves byte sign fields of y3 for conv4's LDS-staged input-gradient kernel
      L3 = self.layers[3]
      self.relu_fields3 = (self.relu_fields and _dev.env('GEECO_NO_DGRAD_LDS') is None and _dev.env('GEECO_NO_FIELDS3') is None and L3['Cin'] == 64
                           and L3['stride'] == 2 and ops.conv3x3_dgrad_relu_fields_supported(L3['H'], L3['W'], L3['Cin'], L3['Cout'], 2))
      if self.relu_fields3:
        self.fields3 = torch.zeros(G, Nf, L3['H'], L3['W'], L3['Cin'] // 8, dtype=torch.uint8, device=dev)
      # dz[0] (conv1's pre-activation gradient, the largest tensor of the step) never exists when the bottom is fused
      self.dz = [None if (i == 0 and self.fused_bottom) else
                 ([torch.empty_like(t) for t in a] if isinstance(a, list) else torch.empty_like(a)) for i, a in enumerate(self.acts)]
      # per-tap transposed kernel copies exist ONLY for the layers whose input-gradient kernel reads them (none in the bench
      # shapes: the LDS-staged kernels and the gather GEMM read the HWIO kernel); every other layer passes wt = NULL, so
      # a dispatcher that disagreed with geeco_conv3x3_dgrad_needs_wt would fail its null-pointer check, not read garbage
      self.needs_wt = [False] + [ops.conv3x3_dgrad_needs_wt(L['H'], L['W'], L['Cin'], L['Cout'], L['stride'])
                                 for L in self.layers[1:]]
      self.wt = [None] + [torch.empty(G, 3, 3, L['Cout'], L['Cin'], **f32) if self.needs_wt[l] else None
                          for l, L in enumerate(self.layers) if l >= 1]
      if self.split_top:
        L7 = self.layers[7]
        self.needs_wt7 = [ops.conv3x3_dgrad_needs_wt(L7['H'], L7['W'], L7['Cin'], d, L7['stride']) for d in self.dim_outs]
        self.wt[7] = [torch.empty(3, 3, d, L7['Cin'], **f32) if nw else None for d, nw in zip(self.dim_outs, self.needs_wt7)]
      if self.pad1:
        self.dw1p = torch.zeros(G, 3, 3, self.Cpad, self.layers[0]['Cout'], **f32)
      # wgrads of different layers may run concurrently (different streams): one split-K workspace each
      nside = int(_dev.env('GEECO_WGRAD_STREAMS', '2'))   # measured: 1 -> 2 streams +1.1 %, 3 slower
      self.wgrad1_on_main = _dev.env('GEECO_WGRAD1_SIDE') is None
      # the LDS-halo wgrad kernels of conv1 / conv2 want whole CUs: beside the dgrad chain they only slow it
      # down (measured: layers below 2 serial +0.8 %, below 3 +0.1 %, below 4 -0.5 %)
      self.serial_below = int(_dev.env('GEECO_SERIAL_BELOW', '2'))
      self.ws_l = [torch.empty(ops.conv3x3_wgrad_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride']) // 4 + 4,
                               **f32) for L in self.layers]
      self.ws = self.ws_l[0]
      dsb = max(ops.conv3x3_dgrad_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'])
                for L in self.layers[1:])
      self.dws = torch.empty(dsb // 4 + 4, **f32)
      # wgrad(l) and dgrad(l) only share their input dz[l]: the dgrad chain stays on the main stream and
      # the wgrads alternate between side streams (branches of the captured hipGraph), so the small top
      # layers overlap instead of leaving CUs idle in their tails
      self.sides = [torch.cuda.Stream(device=dev) for _ in range(nside)] if dev.type == 'cuda' else []
      if self.fused_bottom:
        self.fws_fused = torch.empty(ops.conv2_dgrad_conv1_wgrad_ws_bytes(G) // 4 + 4, **f32)
    fsb = max(ops.conv3x3_fwd_ws_bytes(G, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride']) for L in self.layers)
    self.fws = torch.empty(fsb // 4 + 4, **f32)

  def _w(self, l, g=0):
    return self.store.var('%s/conv%d/kernel' % (self.scopes[g], l + 1))

  def _b(self, l, g=0):
    return self.store.var('%s/conv%d/bias' % (self.scopes[g], l + 1))

  def _grad_view(self, l, g, kind):
    name = '%s/conv%d/%s' % (self.scopes[g], l + 1, kind)
    if self.late is None or l >= ConvEncoderStack.SPLIT:
      return self.store.grad(name)
    staging, stride = self.late
    shp = self.store.shapes[name]
    o = self.store.offsets[name] - self.store.offsets[self.scopes[g] + '/conv1/kernel'] + g * stride
    n = 1
    for d in shp:
      n *= int(d)
    return staging[o:o + n].view(*shp)

  def _dw(self, l, g=0):
    return self._grad_view(l, g, 'kernel')

  def _db(self, l, g=0):
    return self._grad_view(l, g, 'bias')

  def _gs_g(self, l):
    """Group stride of layer l's gradient views (the arena's, or the late staging buffer's for conv1 / conv2)."""
    return self.late[1] if (self.late is not None and l < ConvEncoderStack.SPLIT) else self.gs_p

  def redirect_late_gradients(self, staging, late_ranges):
    """Data parallel (runtime.TrainStepRunner): the gradients of conv1 / conv2 -- the LATE bucket, written by the last
    launches of the backward -- go straight into ``staging`` (encoder g's block at g * len, same inner layout as the
    arena) instead of the gradient arena, so that the arena is not written while the early bucket is being reduced.
    Returns False if the late ranges are not the uniformly strided conv1 / conv2 blocks, or when called with
    ``staging=None``, which ends a redirection (the gradients go to the arena again)."""
    self.late = None
    if staging is None:
      return False
    off = self.store.offsets
    lo0 = off[self.scopes[0] + '/conv1/kernel']
    length = off[self.scopes[0] + '/conv%d/kernel' % (ConvEncoderStack.SPLIT + 1)] - lo0
    want = [(lo0 + g * self.gs_p, length) for g in range(self.G)]
    if [tuple(r) for r in late_ranges] != want or staging.numel() != self.G * length or not self.training:
      return False
    self.late = (staging, length)
    return True

  @property
  def features(self):
    """[G][Nf][h][w][dim_out] output of conv8 (endpoints['conv8'], graph.py:116)."""
    return self.acts[7]

  @property
  def dfeatures(self):
    return self.dz[7]

  def refresh_derived(self):
    """Re-derives the weight copies the kernels read (conv1's kernel padded to 4 input channels; the
    per-tap transposed kernels of the dgrad GEMMs).  Training calls it right after Adam (inside the
    Adam hipGraph), so the forward / backward graphs contain no pad or transpose launches."""
    G = self.G
    # only the layers whose input-gradient kernel reads the transposed copy (the LDS-staged ones read the HWIO kernel)
    ls = [l for l in range(1, 7 if self.split_top else 8) if self.needs_wt[l]] if self.training else []
    if self.training and self.split_top:
      L7 = self.layers[7]
      for g in range(G):
        if self.wt[7][g] is not None:
          ops.derive_conv_weights([self._w(7, g)], [self.wt[7][g].unsqueeze(0)], [L7['Cin']], [self.dim_outs[g]], 1, 0)
    pad = dict(pad_src=self._w(0), pad_dst=self.w1p, pad_cin=self.Cin, pad_cin_padded=self.Cpad,
               pad_cout=self.layers[0]['Cout']) if self.pad1_copy else {}
    if ls or pad:
      ops.derive_conv_weights([self._w(l) for l in ls], [self.wt[l] for l in ls], [self.layers[l]['Cin'] for l in ls],
                              [self.layers[l]['Cout'] for l in ls], G, self.gs_p, **pad)
    self.derived_version = self.store.version

  # -- single launches (also timed one by one by bench.py's per-layer table) ---------------------------
  # the state concat of a one-step decoder inside conv8's split-K epilogue (GEECO_NO_CONCAT_IN_TOP: its own launch)
  concat_in_top = _dev.env('GEECO_NO_CONCAT_IN_TOP') is None

  def launch_fwd(self, l):
    G, Nf, L = self.G, self.Nf, self.layers[l]
    if l == 7 and self.split_top:
      for g in range(G):
        ops.conv3x3_fwd_into(self.acts[7][g], self.acts[6][g], self._w(7, g), self._b(7, g), 1, 0, 0, 0, 0, Nf, L['H'], L['W'],
                             L['Cin'], self.dim_outs[g], L['stride'], relu=True, ws=self.fws)
      return
    x = self.x_in if l == 0 else self.acts[l - 1]
    y = self.acts[l]
    if l == 0 and self.pad1:
      w, gs_w = self.w1p, self.w1p[0].numel()
    else:
      w, gs_w = self._w(l), self.gs_p
    if l == 2 and self.training and self.relu_fields3:
      ops.conv3_fwd_relu_fields_into(y, self.fields3, x, w, self._b(2), G, x[0].numel(), gs_w, self.gs_p, y[0].numel(),
                                     self.fields3[0].numel(), Nf, L['H'], L['W'])
      return
    if l == 1 and self.training and self.relu_fields:
      ops.conv2_fwd_relu_fields_into(y, self.fields2, x, w, self._b(1), G, x[0].numel(), gs_w, self.gs_p, y[0].numel(),
                                     self.fields2[0].numel(), Nf, L['H'], L['W'])
      return
    if l == 0 and self.training and self.relu_bits and self.pad1 and not self.pad1_copy:
      ops.conv1_fwd_relu_bits_rgb_into(y, self.bits1, x, self._w(0), self._b(0), G, x[0].numel(), self.gs_p, self.gs_p, y[0].numel(),
                                       self.bits1[0].numel(), Nf, L['H'], L['W'])
      return
    if l == 0 and self.training and self.relu_bits:
      ops.conv1_fwd_relu_bits_into(y, self.bits1, x, w, self._b(0), G, x[0].numel(), gs_w, self.gs_p, y[0].numel(),
                                   self.bits1[0].numel(), Nf, L['H'], L['W'])
      return
    ops.conv3x3_fwd_into(y, x, w, self._b(l), G, x[0].numel(), gs_w, self.gs_p, y[0].numel(), Nf, L['H'], L['W'],
                         L['Cin'], L['Cout'], L['stride'], relu=True, ws=self.fws)

  def launch_wgrad(self, l, pending=None):
    """Filter + bias gradient of layer l (skipped for conv1 when the encoder bottom is fused: launch_dgrad(1) does it).
    ``pending`` (a list): the kernel's final slab sum is deferred to ``ops.slab_reduce_batch(pending)``."""
    G, Nf, L = self.G, self.Nf, self.layers[l]
    if l == 0 and self.fused_bottom:
      return
    if l == 7 and self.split_top:
      for g in range(G):
        ops.conv3x3_wgrad_into(self._dw(7, g), self._db(7, g), self.acts[6][g], self.dz[7][g], 1, 0, 0, 0, 0, Nf, L['H'],
                               L['W'], L['Cin'], self.dim_outs[g], L['stride'], self.ws_l[7])
      return
    x = self.x_in if l == 0 else self.acts[l - 1]
    dz = self.dz[l]
    if l == 0 and self.pad1:
      dw, gs_dw = self.dw1p, self.dw1p[0].numel()
    else:
      dw, gs_dw = self._dw(l), self._gs_g(l)
    if l == 0 and self.pad1:
      pending = None     # the padded gradient is repacked right below
    ops.conv3x3_wgrad_into(dw, self._db(l), x, dz, G, x[0].numel(), dz[0].numel(), gs_dw, self._gs_g(l), Nf, L['H'],
                           L['W'], L['Cin'], L['Cout'], L['stride'], self.ws_l[l], pending=pending,
                           reserved_cus=self.reserved_cus if (l == 1 and pending is not None) else 0)
    if l == 0 and self.pad1:
      for g in range(G):
        ops.pad_mid_into(self._dw(0, g), self.dw1p[g], 9, self.Cpad, self.Cin, L['Cout'])

  def _wgrad_args(self, l):
    L = self.layers[l]
    x, dz = self.acts[l - 1], self.dz[l]
    return dict(dw=self._dw(l), db=self._db(l), x=x, dz=dz, gs_x=x[0].numel(), gs_dz=dz[0].numel(), gs_dw=self._gs_g(l),
                gs_db=self._gs_g(l), N=self.Nf, H=L['H'], W=L['W'], Cin=L['Cin'], Cout=L['Cout'], ws=self.ws_l[l])

  def launch_wgrad_top_pair(self, pending=None):
    """conv7's and conv8's filter gradients as ONE launch (both are ready once conv8's input gradient exists; each alone is
    432 blocks on 256 CUs): False when the shapes are outside the paired kernel (the caller launches them one by one)."""
    return ops.conv3x3_wgrad_pair_into(self._wgrad_args(6), self._wgrad_args(7), self.G, self.layers[6]['stride'], pending=pending)

  def launch_top_bwd(self, l, wgrads, pending=None):
    """Layer l's input gradient AND the filter gradients of layers ``wgrads`` as one heterogeneous launch (independent work
    that needs only dz[l]): l = 6 with (6, 7), or l = 7 with (7,) followed by l = 6 with (6,)."""
    L = self.layers[l]
    wt = self.wt[l]
    d = dict(dx=self.dz[l - 1], dz=self.dz[l], wt=wt, ymask=self.acts[l - 1], w=self._w(l), gs_dz=self.dz[l][0].numel(),
             gs_w=self.gs_p, gs_wt=wt[0].numel() if wt is not None else 0, gs_dx=self.dz[l - 1][0].numel(), N=self.Nf, H=L['H'],
             W=L['W'], Cin=L['Cin'], Cout=L['Cout'], ws=self.dws)
    return ops.conv_top_bwd_into(d, self._wgrad_args(wgrads[0]), self._wgrad_args(wgrads[1]) if len(wgrads) > 1 else None, self.G,
                                 L['stride'], pending=pending)

  def launch_dgrad(self, l, pending=None):
    """Input gradient of layer l >= 1 into dz[l-1] (ReluGrad of the layer below fused).  With the fused encoder
    bottom, l == 1 also produces conv1's filter / bias gradient: dz1 has no other consumer and stays on chip
    (805 MB less written and read again per step, one big launch less)."""
    G, Nf, L = self.G, self.Nf, self.layers[l]
    if l == 7 and self.split_top:
      for g in range(G):
        ops.conv3x3_dgrad_into(self.dz[6][g], self.dz[7][g], self.wt[7][g], self.acts[6][g], 1, 0, 0, 0, Nf, L['H'], L['W'],
                               L['Cin'], self.dim_outs[g], L['stride'], ws=self.dws, w=self._w(7, g), gs_w=0)
      return
    x = self.acts[l - 1]
    dz = self.dz[l]
    if l == 1 and self.fused_bottom:
      # the kernel writes conv1's gradient in the variable's own [3][3][Cin][32] layout (no padded copy to repack)
      if self.relu_bits:
        ops.conv2_dgrad_conv1_wgrad_bits_into(self._dw(0), self._db(0), dz, self._w(1), self.bits1, self.x_in, G,
                                              dz[0].numel(), self.gs_p, self.bits1[0].numel(), self.x_in[0].numel(),
                                              self._gs_g(0), self._gs_g(0), Nf, L['H'], L['W'], self.fws_fused,
                                              real_channels=self.Cin, pending=pending, reserved_cus=self.reserved_cus)
        return
      ops.conv2_dgrad_conv1_wgrad_into(self._dw(0), self._db(0), dz, self._w(1), x, self.x_in, G, dz[0].numel(), self.gs_p,
                                       x[0].numel(), self.x_in[0].numel(), self._gs_g(0), self._gs_g(0), Nf, L['H'], L['W'],
                                       self.fws_fused, real_channels=self.Cin, pending=pending,
                                       reserved_cus=self.reserved_cus if pending is not None else 0)
      return
    wt = self.wt[l]
    dx = self.dz[l - 1]
    if l == 3 and self.relu_fields3:
      ops.conv3x3_dgrad_relu_fields_into(dx, dz, self._w(3), self.fields3, G, dz[0].numel(), self.gs_p, self.fields3[0].numel(),
                                         dx[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'])
      return
    if l == 2 and self.relu_fields:
      ops.conv3_dgrad_relu_fields_into(dx, dz, self._w(2), self.fields2, G, dz[0].numel(), self.gs_p, self.fields2[0].numel(),
                                       dx[0].numel(), Nf, L['H'], L['W'], reserved_cus=self.reserved_cus)
      return
    ops.conv3x3_dgrad_into(dx, dz, wt, x, G, dz[0].numel(), wt[0].numel() if wt is not None else 0, dx[0].numel(), Nf, L['H'],
                           L['W'], L['Cin'], L['Cout'], L['stride'], ws=self.dws, w=self._w(l), gs_w=self.gs_p)

  def forward(self, state=None):
    """``state`` (one-step decoders): dict(state, state_stride, feat_off, Ctot, jnt, jnt_stride, jnt_off, J) of the state
    concat that consumes the features; it then rides in the epilogue of the top layer's split-K sum where that exists.
    Returns True if it did (else the caller launches the concat)."""
    if not self.lazy_refresh or self.derived_version != self.store.version:
      self.refresh_derived()
    top = len(self.layers) - 1
    for l in range(top):
      self.launch_fwd(l)
    if state is not None and self.concat_in_top and not self.split_top:
      G, Nf, L = self.G, self.Nf, self.layers[top]
      x, y = self.acts[top - 1], self.acts[top]
      if ops.conv3x3_fwd_state_into(y, x, self._w(top), self._b(top), G, x[0].numel(), self.gs_p, self.gs_p, y[0].numel(), Nf,
                                    L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], self.fws, **state):
        return True
    self.launch_fwd(top)
    return False

  def backward(self, hi=7, lo=0, prepare=None, defer_dgrad=False, lead_dgrad=None, defer_sums=None, before_bottom=None):
    """Expects ``self.dz[7]`` = d(loss)/d(pre-activation of conv8) (ReluGrad already applied).  Runs layers
    hi..lo (the data-parallel runner splits the chain at conv3 / conv2 to start the gradient exchange early).
    ``prepare`` = (global_step, lr, scal): the optimiser's per-step scalars ride in this part's slab-sum launch.
    ``defer_dgrad``: layer lo's INPUT gradient is left to the next part, which opens with it (``lead_dgrad=lo``): every
    gradient of the early bucket exists once layer lo's filter gradient does, so the bucket leaves a launch earlier.
    ``defer_sums`` (a list): this part's pending slab sums are handed to the caller instead of launched (``prepare`` must be None).
    ``before_bottom``: called right before the LAST launch of the chain, conv2's input gradient (+ conv1's filter gradient when
    the bottom is fused) -- _ModelBase.backward_and_apply releases the optimiser's early piece onto a second stream there."""
    assert defer_sums is None or prepare is None
    main = torch.cuda.current_stream()
    sides = self.sides if self.two_streams else []
    pending = [] if self.batch_reduce else None   # slab sums of all layers of this part: one launch at the end
    # conv8's filter gradient waits for conv7's: one launch for both (independent work batched into one grid)
    pair_top = (hi == 7 and lo <= 6 and not self.split_top and not sides and self.pair_top
                and self.layers[6]['stride'] == self.layers[7]['stride'] == 2)
    if lead_dgrad is not None:
      self.launch_dgrad(lead_dgrad, pending)
    for l in range(hi, lo - 1, -1):
      if pair_top and l == 7:
        self.w8_done = self.hetero_top == 2 and self.launch_top_bwd(7, (7,), pending)
        if not self.w8_done:
          self.launch_dgrad(7, pending)
        continue
      if pair_top and l == 6 and self.w8_done:
        if not self.launch_top_bwd(6, (6,), pending):
          self.launch_wgrad(6, pending)
          self.launch_dgrad(6, pending)
        continue
      if pair_top and l == 6:
        if self.hetero_top and self.launch_top_bwd(6, (6, 7), pending):
          continue
        if not self.launch_wgrad_top_pair(pending):
          self.launch_wgrad(7, pending)
          self.launch_wgrad(6, pending)
        self.launch_dgrad(6, pending)
        continue
      # wgrad(l) of the upper layers is off the critical path (the dgrad chain on `main`): it goes to a side
      # stream; the bottom layers' (LDS-halo kernels, one or two blocks per CU) stay on `main`
      side = None
      if sides and not (l == 0 and self.wgrad1_on_main) and l >= self.serial_below:
        side = sides[l % len(sides)]
      if side is not None:
        side.wait_stream(main)          # dz[l] is ready
      with torch.cuda.stream(side if side is not None else main):
        self.launch_wgrad(l, pending)
      if l == 0 or (l == lo and defer_dgrad):
        break   # conv1's input is data: no dgrad / the next part opens with this layer's
      if l == 1 and before_bottom is not None:
        before_bottom(pending)
      self.launch_dgrad(l, pending)
      if l == 1 and self.fused_bottom:
        break
    for side in sides:
      main.wait_stream(side)
    if defer_sums is not None and pending is not None:
      defer_sums.extend(pending)
      return
    if pending or (prepare is not None and pending is not None):
      ops.slab_reduce_batch(pending, prepare)
    elif prepare is not None:
      ops.adam_prepare(*prepare)

  SPLIT = 2   # backward(part='upper') = layers 7..SPLIT, 'bottom' = SPLIT-1..0
  # ... except layer SPLIT's INPUT gradient, which opens the bottom part (GEECO_DP_DGRAD_IN_UPPER: round 4's cut)
  DEFER_SPLIT_DGRAD = _dev.env('GEECO_DP_DGRAD_IN_UPPER') is None


# ================================================================================================
# LSTM decoder + heads + losses (graph.py:198-260, 430-500; estimator.py:206-239)
# ================================================================================================
def head_table(cfg):
  """(variable name, prediction key, size, kind, loss weight) per head, in variable creation order.
  kind 0 = mean_squared_error, 1 = softmax cross-entropy (graph.py:233-259, 430-500; estimator.py:224-237)."""
  if cfg.control_mode == 'cartesian':
    lam = float(cfg.lambda_aux)
    return [('pred_cmd_ee', 'cmd_ee', 3, 0, 1.0), ('logits_cmd_grp', 'logits_cmd_grp', cfg.num_grp_states, 1, 1.0),
            ('pred_aux_ee', 'pos_ee', 3, 0, lam), ('pred_aux_obj', 'pos_obj', 3, 0, lam)]
  if cfg.control_mode == 'velocity':   # mse_loss sums all five terms unweighted (graph.py:446-449)
    return [('pred_cmd_vel', 'cmd_vel', cfg.dim_jnt_state, 0, 1.0), ('pred_cmd_ee', 'cmd_ee', 3, 0, 1.0),
            ('pred_cmd_grp', 'cmd_grp', cfg.dim_grp_command, 0, 1.0), ('pred_aux_ee', 'pos_ee', 3, 0, 1.0),
            ('pred_aux_obj', 'pos_obj', 3, 0, 1.0)]
  raise ValueError("Unknown control mode '%s'" % (cfg.control_mode,))


class LSTMDecoder:
  """T LSTM steps over states [T][N][D] from a zero state, fc1 + heads on the last output."""

  def __init__(self, store: VariableStore, scope, cfg, N, T, D, training):
    self.store, self.scope, self.cfg, self.N, self.T, self.D = store, scope, cfg, N, T, D
    self.H, self.F = cfg.dim_h_lstm, cfg.dim_h_fc
    self.training = training
    self.heads = head_table(cfg)
    self.OT = sum(h[2] for h in self.heads)
    dev = store.device
    f32 = dict(dtype=torch.float32, device=dev)
    H, F = self.H, self.F
    self.states = torch.empty(T, N, D, **f32)
    self.z = torch.empty(T, N, 4 * H, **f32)
    self.gates = torch.empty(T, N, 4 * H, **f32)
    self.c = torch.empty(T, N, H, **f32)
    self.h = torch.empty(T, N, H, **f32)
    self.preds = torch.empty(N, self.OT, **f32)
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
    self.targets, self.target_strides = None, None     # bound by the model
    self.loss_scale = 1.0
    # one-step decoders: weight / bias / input gradients (+ the state-concat backward) as TWO launches instead of five
    # dependent ones (GEECO_DEV=1 GEECO_NO_LSTM_BATCH: the separate launches)
    self.one_launch_bwd = _dev.env('GEECO_NO_LSTM_BATCH') is None
    self.one_launch_fwd = _dev.env('GEECO_NO_LSTM_FWD_FUSE') is None   # ... and the forward's slab sum inside the gate kernel
    # round 5: gate math + fc1 + heads + losses (+ their backward down to the gate gradients) per sample in ONE launch
    self.fused_step = _dev.env('GEECO_NO_STEP_HEADS') is None
    self.heads_pending, self.dz_from_heads = None, False

  def _v(self, n):
    return self.store.var('%s/%s' % (self.scope, n))

  def _g(self, n):
    return self.store.grad('%s/%s' % (self.scope, n))

  def forward(self, backward_too):
    N, T, D, H, F = self.N, self.T, self.D, self.H, self.F
    W = self._v('lstm_cell/kernel')            # [D + H][4H]: rows 0..D-1 multiply x, D.. multiply h
    Wx, Wh = W[:D], W[D:]
    bias = self._v('lstm_cell/bias')
    names = [h[0] for h in self.heads]
    hkw = {}
    if backward_too:
      hkw = dict(d_fc1_w=self._g('fc1/kernel'), d_fc1_b=self._g('fc1/bias'),
                 d_heads_w=[self._g(n + '/kernel') for n in names], d_heads_b=[self._g(n + '/bias') for n in names])
    self.heads_pending = None
    if T == 1 and self.one_launch_fwd and self.one_launch_bwd and self.fused_step:
      # one step from a zero state: gate GEMM + ONE per-sample launch for the slab sum, the gate math, fc1, the heads, the loss
      # terms and (training) everything back to the gate gradients dz; the batch sums (weight / bias gradients, loss means)
      # ride in the first grid of backward()'s launch pair -- losses / those gradients are final after backward()
      pend = _native.HeadsFinish() if backward_too else None
      if ops.lstm_step_heads_into(self.z[0], self.c[0], self.h[0], self.gates[0], self.states[0], Wx, bias, N, H, D, D, 4 * H,
                                  self.gemm_ws, self.preds, self.losses, self._v('fc1/kernel'), self._v('fc1/bias'),
                                  [self._v(n + '/kernel') for n in names], [self._v(n + '/bias') for n in names],
                                  [h[2] for h in self.heads], [h[3] for h in self.heads], [h[4] for h in self.heads],
                                  self.targets, self.target_strides, float(self.loss_scale), F, self.heads_ws,
                                  dz=self.dz[0] if backward_too else None, pending=pend, **hkw):
        self.heads_pending = pend if backward_too else None
        self.dz_from_heads = backward_too
        return
    self.dz_from_heads = False
    if T == 1 and self.one_launch_fwd:
      # one step from a zero state: the slab sum of the gate GEMM rides in the gate kernel (bitwise the same)
      ops.lstm_input_step_fwd_into(self.z[0], self.c[0], self.h[0], self.gates[0], self.states[0], Wx, bias, N, H, D, D, 4 * H,
                                   self.gemm_ws)
      T = 0
    else:
      # hoisted input projection for all steps: Z = X Wx
      ops.gemm_into(self.z, self.states, Wx, T * N, 4 * H, D, D, 4 * H, 4 * H, ws=self.gemm_ws)
    for t in range(T):
      if t > 0:
        ops.gemm_into(self.z[t], self.h[t - 1], Wh, N, 4 * H, H, H, 4 * H, 4 * H, accumulate=True, ws=self.gemm_ws)
      ops.lstm_gates_fwd_into(self.c[t], self.h[t], self.gates[t], self.z[t], bias,
                              self.c[t - 1] if t > 0 else None, N, H)
    kw = dict(dh=self.dh, **hkw) if backward_too else {}
    ops.heads_loss_into(self.preds, self.losses, self.h[self.T - 1], self._v('fc1/kernel'), self._v('fc1/bias'),
                        [self._v(n + '/kernel') for n in names], [self._v(n + '/bias') for n in names],
                        [h[2] for h in self.heads], [h[3] for h in self.heads], [h[4] for h in self.heads],
                        self.targets, self.target_strides, float(self.loss_scale), N, H, F, self.heads_ws, **kw)

  def backward(self, concat=None):
    """After forward(backward_too=True): fills d(states) and the LSTM variable gradients.  ``concat`` (one-step decoders
    only): dict(feats, dfeats, feat_ch, jnt_pos, J, cells) of the state concat that produced ``states``; its backward
    (feature gradients + ReluGrad of conv8) then rides in the same launch as the weight / input gradients and the method
    returns True (else the caller scatters ``dstates`` itself)."""
    N, T, D, H = self.N, self.T, self.D, self.H
    W = self._v('lstm_cell/kernel')
    Wx, Wh = W[:D], W[D:]
    dW = self._g('lstm_cell/kernel')
    if T == 1 and self.one_launch_bwd:
      # one step from a zero state: dWh = h_prev^T dz = 0 (the arena's rows stay zero); everything else in ONE launch
      if not self.dz_from_heads:       # (the fused forward left dz itself)
        ops.lstm_gates_bwd_into(self.dz[0], None, self.gates[0], None, self.c[0], self.dh, None, N, H)
      kw = dict(feats_fwd=concat['feats'], dfeats=concat['dfeats'], feat_ch=concat['feat_ch'], jnt_pos=concat['jnt_pos'],
                J=concat['J'], cells=concat['cells']) if concat else {}
      ops.lstm_step_bwd_into(dW[:D], self._g('lstm_cell/bias'), self.dstates[0], self.states[0], self.dz[0], Wx, N, D, 4 * H,
                             4 * H, self.gemm_ws, pending=self.heads_pending, **kw)
      self.heads_pending = None
      return concat is not None
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
    return False


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
    shapes = model_variable_shapes(cfg, goal)
    encoder_scopes = sorted({n.split('/conv')[0] for n in shapes if '/conv' in n}, key=lambda sc: list(shapes).index(sc + '/conv1/kernel'))
    self.store = store or VariableStore(shapes, self.device, uniform_scopes=encoder_scopes)
    f32 = dict(dtype=torch.float32, device=self.device)
    N, K, H, W = self.N, self.K, self.H, self.W
    self.inputs = {
        'rgb': torch.zeros(N, K, H, W, 3, **f32),
        'jnt_state': torch.zeros(N, K, cfg.dim_jnt_state, **f32),
        'ee_state': torch.zeros(N, K, 7, **f32),
        'obj_state': torch.zeros(N, K, 7, **f32),
    }
    if cfg.control_mode == 'cartesian':        # labels consumed by the losses (estimator.py:206-216, 230-236)
      self.label_keys = ['cmd']
      self.inputs['cmd'] = torch.zeros(N, 4, **f32)
    elif cfg.control_mode == 'velocity':
      self.label_keys = ['vel_target', 'ee_target', 'grp_target']
      self.inputs['vel_target'] = torch.zeros(N, cfg.dim_jnt_state, **f32)
      self.inputs['ee_target'] = torch.zeros(N, 7, **f32)
      self.inputs['grp_target'] = torch.zeros(N, cfg.dim_grp_command, **f32)
    else:
      raise ValueError("Unknown control mode '%s'" % (cfg.control_mode,))
    if self.C == 4:
      self.inputs['depth'] = torch.zeros(N, K, H, W, 1, **f32)
    if goal:
      self.inputs['target_rgb'] = torch.zeros(N, H, W, 3, **f32)
      if self.C == 4:
        self.inputs['target_depth'] = torch.zeros(N, H, W, 1, **f32)
    self.scal = torch.zeros(4, **f32)          # [0] = Adam lr_t, [1] = sum of squares of the arena
    self.world = 1
    # RGB-D: rgb || depth (estimator.py:36,169,172).  The dynimg branch of the goal model forms the concat inside its
    # input kernels (no packed copy of all N * K frames: 1.07 GB read + 1.43 GB written per step at K = 32); the other
    # graphs pack once per step.
    self.last_from_dynimg = _dev.env('GEECO_PACK_CURRENT') is None   # current frame's padded copy out of the buffer-image kernel
    self.split_rgbd = (self.C == 4 and goal and cfg.proc_obs == 'dynimg' and (H * W) % 4 == 0 and
                       _dev.env('GEECO_PACK_RGBD') is None)
    if self.C == 4 and not self.split_rgbd:
      self.obs4 = torch.empty(N, K, H, W, 4, **f32)
      if goal:
        self.tgt4 = torch.empty(N, H, W, 4, **f32)

  # image inputs this model can read as uint8 frames behind window addresses (input_fn.WindowFeed.pointers()) instead of dense
  # float32 windows; () = none
  u8_window_keys = ()

  def load_batch(self, features, labels=None):
    """Copies one batch into the static input buffers (H2D or D2D; torch is plumbing here)."""
    for k, buf in self.inputs.items():
      if hasattr(buf, 'pointers'):
        raise RuntimeError("load_batch: input '%s' is fed through window addresses (input_fn.WindowFeed.feed)" % k)
      src = labels.get(k) if (labels is not None and k in self.label_keys) else features.get(k)
      if src is None:
        if k in self.label_keys:
          continue
        raise KeyError("missing feature '%s'" % k)
      src = torch.as_tensor(src)
      if tuple(src.shape) != tuple(buf.shape):
        raise ValueError("feature '%s': expected shape %s, got %s" % (k, tuple(buf.shape), tuple(src.shape)))
      buf.copy_(src, non_blocking=True)

  def _frames(self):
    """obs_frames / tgt_frame of the model_fn (estimator.py:30-39, 161-175): rgb, or rgb||depth packed to 4 channels."""
    N, K, HW = self.N, self.K, self.H * self.W
    if self.C == 3:
      return self.inputs['rgb'], self.inputs.get('target_rgb')
    ops.pack_pixels_into(self.obs4, self.inputs['rgb'], HW * 3, N * K, HW, 3, 4, self.inputs['depth'], HW, 1)
    if self.goal:
      ops.pack_pixels_into(self.tgt4, self.inputs['target_rgb'], HW * 3, N, HW, 3, 4, self.inputs['target_depth'], HW, 1)
      return self.obs4, self.tgt4
    return self.obs4, None

  def _bind_labels(self):
    K, inp = self.K, self.inputs
    ee_last, obj_last = inp['ee_state'][:, K - 1], inp['obj_state'][:, K - 1]   # features[...][:, -1, :3]
    if self.cfg.control_mode == 'cartesian':
      cmd = inp['cmd']
      tg = [(cmd, 4), (cmd[:, 3:], 4), (ee_last, K * 7), (obj_last, K * 7)]
    else:
      tg = [(inp['vel_target'], inp['vel_target'].shape[1]), (inp['ee_target'], 7),
            (inp['grp_target'], inp['grp_target'].shape[1]), (ee_last, K * 7), (obj_last, K * 7)]
    self.decoder.targets = [t for t, _ in tg]
    self.decoder.target_strides = [s for _, s in tg]

  def redirect_late_gradients(self, staging, late_ranges):
    return self.enc.redirect_late_gradients(staging, late_ranges)

  # -- optimiser step (estimator.py:243-244) -------------------------------------------------
  def _begin_step(self, backward_too):
    """A training forward opens a new optimiser step: a ``_prepared`` left over from a step whose apply_gradients never came
    (an exception between the parts, a one-graph capture that failed after recording part 2, a caller that ran the backward
    alone to look at gradients) must not make THIS step's apply_gradients skip its adam_prepare."""
    if backward_too:
      self._prepared = False

  def _prepare_args(self, adam_prepare):
    """backward(adam_prepare=True): the step counter / lr_t update rides in the backward's last slab-sum launch (one dependent
    launch less); apply_gradients() then skips its own adam_prepare.  Only callers that DO apply the gradients next pass it
    (train_step, runtime.TrainStepRunner): the counter must advance exactly once per optimiser step."""
    if not adam_prepare:
      return None
    self._prepared = True
    return (self.store.global_step, float(self.cfg.lr), self.scal)

  def apply_gradients_of(self, segments, g_out=None, last=True):
    """The optimiser step of SOME pieces of the arena (``ops.adam_tf_segments``; data parallel: everything that came with the early
    bucket first, the late bucket's variables when it has arrived).  ``last``: the pieces complete the step (weight copies are
    re-derived behind it)."""
    s, cfg = self.store, self.cfg
    if not getattr(self, '_prepared', False):
      ops.adam_prepare(s.global_step, float(cfg.lr), self.scal)
      self._prepared = not last
    if last:
      self._prepared = False
    ops.adam_tf_segments(s.params, s.adam_m, s.adam_v, segments, self.scal, g_out=g_out, grad_scale=1.0 / self.world,
                         l2=float(cfg.l2_regularizer))
    if last:
      self._refresh_after_update()

  def apply_gradients(self):
    s, cfg = self.store, self.cfg
    if not getattr(self, '_prepared', False):
      ops.adam_prepare(s.global_step, float(cfg.lr), self.scal)
    self._prepared = False
    ops.adam_tf(s.params, s.grads, s.adam_m, s.adam_v, s.size, self.scal, grad_scale=1.0 / self.world,
                l2=float(cfg.l2_regularizer))
    self._refresh_after_update()

  def can_apply_beside_bottom(self):
    """backward_and_apply needs the slab sums batched per part (the default) and a CUDA device."""
    return bool(self.enc.training and self.enc.batch_reduce and not self.enc.two_streams and self.store.params.is_cuda)

  def optimizer_stream(self):
    """The second stream of backward_and_apply: one of the encoder's filter-gradient side streams, idle from conv2's filter gradient
    on (a process gets 4 hardware queues by default; a stream more would share one with another stream -- RCCL's, perhaps)."""
    side = getattr(self, '_opt_stream', None)
    if side is None:
      sides = getattr(self.enc, 'sides', None)
      side = self._opt_stream = sides[0] if sides else torch.cuda.Stream(device=self.store.params.device)
    return side

  def backward_and_apply(self, early, late):
    """Backward + optimiser step of a single-GPU training step with the optimiser's HBM-streaming work hidden beside the fused
    encoder-bottom backward (round 6, profiles/HARDWARE_FINDINGS.md 38).  ``early`` / ``late`` = runtime.gradient_buckets(store):
    late = conv1 / conv2 of the encoders, whose gradients the last launch of the backward produces; early = everything else
    (99.4 % of the arena), complete once conv3's filter gradient exists.

      main:  ... conv3 wgrad | conv3 dgrad | conv2 wgrad | * | conv2 dgrad + conv1 wgrad (454 us, MFMA-bound) | conv1's slab sum | join | Adam(late)
      side:                                                * -> slab sums of conv2..conv8 (+ lr_t) -> Adam(early)

    The fused bottom holds two 209-VGPR waves per SIMD and 151 KB of LDS: 80 registers per lane and 9 KB of LDS stay free on every
    CU, room for one block of the slab sums (55 VGPRs, 4 KB) or of Adam (51 VGPRs) at a time, and the bottom moves 1.1 TB/s of the
    8 the HBM has.  Beside it the 35 + 33 us of streaming work take 190 + 155 us and end long before its 455 us are over, which
    stay 455.  Order matters twice: released any earlier the side work could not co-reside (conv2's filter gradient and conv3's
    input gradient fill the register file) but would start first and delay them -- and `*` is behind conv3's input gradient, the
    last reader of a variable of the early piece (conv3's kernel); and the side launches are issued BEHIND the bottom's (see
    below).  Element by element the arithmetic of backward(adam_prepare=True) + apply_gradients(): bitwise the same parameters,
    slots and gradients (tests/test_model_gpu.py)."""
    side = self.optimizer_stream()
    main = torch.cuda.current_stream()
    what = _dev.env('GEECO_BESIDE', 'both')      # development A/B: which of the two pieces goes beside the bottom
    sums = []
    self.backward(part='upper', defer_sums=sums)
    prepare = self._prepare_args(True)
    if what == 'adam':
      ops.slab_reduce_batch(sums, prepare)
      prepare = None
    g = self.store.grads
    ev = torch.cuda.Event()
    marked = []

    def mark(pending=None):      # conv2's filter gradient has been launched: everything before it (conv3's input gradient too) precedes `ev`
      ev.record(main)
      if pending:                # ... and its slab sum joins the ones that run beside the bottom: only conv1's stays behind it
        sums.extend(pending)
        del pending[:]
      marked.append(True)
    self.backward(part='bottom', before_bottom=mark)
    if not marked:       # (an encoder whose chain has no conv2 input gradient: nothing to hide behind)
      mark()
    # The side work is launched BEHIND the fused bottom (and waits for `ev`, recorded in front of it): the bottom's packet is the
    # older one, its 256 persistent blocks take their CUs first, and the streaming blocks then fill what those leave, one per CU
    # at a time.  Issued in front of it (hipGraph replays nodes in creation order) the streaming blocks take the wave slots first,
    # the bottom's block cannot become resident on a CU until they have drained, and the bottom ends 35-45 us late (measured).
    side.wait_event(ev)
    with torch.cuda.stream(side):
      if sums or prepare is not None:
        ops.slab_reduce_batch(sums, prepare)
      if what != 'reduce':
        self.apply_gradients_of([(g[off:off + n], off, n) for off, n in early], last=False)
    main.wait_stream(side)
    if what == 'reduce':
      self.apply_gradients_of([(g[off:off + n], off, n) for off, n in early], last=False)
    self.apply_gradients_of([(g[off:off + n], off, n) for off, n in late], last=True)

  def _refresh_after_update(self):
    s = self.store
    # weights changed: re-derive the padded / transposed copies now (the version stamp is unchanged, so
    # the next forward, eager or replayed, launches no pad / transpose kernels).  A model that shares the store
    # with the primary training stack (the one built for a ragged final batch) must refresh THAT stack's copies
    # too: the primary relies on its own post-Adam refresh and would otherwise run one step on stale copies.
    self.enc.refresh_derived()
    primary = getattr(s, 'primary_stack', None)
    if primary is not None and primary is not self.enc:
      primary.refresh_derived()

  def predictions(self):
    """estimator.py:48-61 / 183-197."""
    p, out, off = self.decoder.preds, {}, 0
    for _, key, size, _, _ in self.decoder.heads:
      out[key] = p[:, off:off + size]
      off += size
    return out

  def check_device_errors(self):
    """Raises if a kernel of this model reported an error on the device (today: a block of the one-pass input stage that gave up
    waiting, csrc/dynimg.hip).  SYNCHRONISES the stream: called where the host reads results anyway (Estimator's loss
    read-outs and epoch ends, bench.py's loss check, ``endpoints``), never inside the step."""
    ws = getattr(self, 'dyn_ws2', None)
    if ws is not None and getattr(self, 'mode', None) == 'dynimg':
      ops.check_input_stage(ws, self.N)

  def _finish_forward(self):
    if self.cfg.l2_regularizer > 0.0:    # loss_reg = l2 * sum(v^2)/2 over every variable (graph.py:13-15, estimator.py:66,202)
      ops.sumsq_into(self.scal[1:2], self.store.params, self.store.size)

  @property
  def loss(self):
    """Device scalar: total loss of the last forward (local batch mean + L2 term, estimator.py:101,239)."""
    l = self.decoder.losses[0]
    if self.cfg.l2_regularizer > 0.0:
      l = l + (0.5 * float(self.cfg.l2_regularizer)) * self.scal[1]
    return l

  def loss_parts(self):
    l = self.decoder.losses
    out = {'loss': self.loss}
    for i, (_, key, _, _, _) in enumerate(self.decoder.heads):
      out['loss_' + key.replace('logits_', '')] = l[1 + i]
    if self.cfg.l2_regularizer > 0.0:
      out['loss_reg'] = (0.5 * float(self.cfg.l2_regularizer)) * self.scal[1]
    return out

  def train_step(self):
    self.forward(backward_too=True)
    self.backward(adam_prepare=True)
    self.apply_gradients()


class GoalE2EVMC(_ModelBase):
  """``goal_e2evmc`` (graph.py:321-416), every proc_obs x proc_tgt branch (scope 'GoalVMC')."""

  def __init__(self, cfg, N, device, training=True, store=None):
    super().__init__(cfg, N, device, goal=True, training=training, store=store)
    if cfg.proc_tgt not in ('constant', 'residual', 'dyndiff'):
      raise ValueError("Unknown processing mode for target image: %s!" % (cfg.proc_tgt,))
    if cfg.proc_obs not in ('sequence', 'dynimg'):
      raise ValueError("Unknown processing mode for frame buffer: %s!" % (cfg.proc_obs,))
    root = 'GoalVMC'
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    jn = cfg.dim_jnt_state
    self.mode = cfg.proc_obs if cfg.proc_obs == 'dynimg' else 'seq_' + cfg.proc_tgt
    if self.mode == 'dynimg':             # geeco-f (:386-407); proc_tgt is ignored by this branch
      scopes, Nf, T = [root + '/ConvEncoder', root + '/DynBuffEncoder', root + '/DynDiffEncoder'], N, 1
      self.feat_ch = [cfg.dim_s_obs, cfg.dim_s_dyn, cfg.dim_s_diff]
      dims = self.feat_ch
    elif self.mode in ('seq_constant', 'seq_residual'):   # target goes through the SAME ConvEncoder (:354, 364)
      scopes, Nf, T = [root + '/ConvEncoder'], (K + 1) * N, K
      self.feat_ch = [cfg.dim_s_obs, cfg.dim_s_obs] if self.mode == 'seq_constant' else [cfg.dim_s_obs]
      dims = [cfg.dim_s_obs]
    else:                                  # seq_dyndiff (:371-381)
      scopes, Nf, T = [root + '/ConvEncoder', root + '/DynDiffEncoder'], K * N, K
      self.feat_ch = [cfg.dim_s_obs, cfg.dim_s_diff]
      dims = self.feat_ch
    self.enc = ConvEncoderStack(self.store, scopes, Nf, H, W, C, dims, training)
    D = _CELLS * (sum(self.feat_ch) + jn)
    self.decoder = LSTMDecoder(self.store, root + '/LSTMDecoder', cfg, N, T, D, training)
    self._bind_labels()
    self.dyn_ws = ops.dynimg_ws(N, H * W * 4, self.device)
    self.dyn_ws2 = ops.goal_dynimgs_ws(N, H * W, self.device)      # control block of the one-pass input stage (zero-filled once)
    # geeco-f reads its K-frame window ONCE, in the input kernel: that kernel can take the episodes' resident uint8 frames
    # directly (RGB, or RGB of RGB-D with depth dense), see ops.goal_dynimgs_u8_into
    if (self.mode == 'dynimg' and self.last_from_dynimg and (H * W) % 4 == 0 and (C == 3 or self.split_rgbd) and
        _dev.env('GEECO_NO_U8_WINDOWS') is None):
      self.u8_window_keys = ('rgb', 'target_rgb')

  def _encode_dynimg_state(self):
    """The three encoders and representation_concatenation_v2: [obs | dyn | jnt | tgt] (graph.py:169-192) of the current
    step's joint state, jnt_state_list[-1] (:388); the concat rides in conv8's split-K epilogue where that exists."""
    N, K, jn, d = self.N, self.K, self.cfg.dim_jnt_state, self.decoder
    jnt = self.inputs['jnt_state'][:, K - 1]
    ch = self.feat_ch
    Ctot = sum(ch) + jn
    state = dict(state=d.states[0], state_stride=d.D, feat_off=[0, ch[0], ch[0] + ch[1] + jn], Ctot=Ctot, jnt=jnt,
                 jnt_stride=K * jn, jnt_off=ch[0] + ch[1], J=jn)
    if not self.enc.forward(state if len(set(ch)) == 1 else None):
      feats = self.enc.features                               # [3][N][2][2][256]
      ops.state_concat_fwd_into(d.states[0], [feats[0], feats[1], feats[2]], ch, 2, jnt, K * jn, jn, N, _CELLS, d.D)

  def forward(self, backward_too=False):
    self._begin_step(backward_too)
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    HW = H * W
    x_in = self.enc.x_in
    jn = self.cfg.dim_jnt_state
    jnts = self.inputs['jnt_state']
    d = self.decoder
    u8 = hasattr(self.inputs['rgb'], 'pointers')          # estimator: the Estimator bound window addresses (uint8 frames)
    if u8 and not hasattr(self.inputs['target_rgb'], 'pointers'):
      raise RuntimeError('GoalE2EVMC: rgb comes as window addresses but target_rgb as a dense tensor')
    if self.mode == 'dynimg' and self.split_rgbd:
      inp = self.inputs
      rgb, dep = inp['rgb'], inp['depth']
      if u8:
        ops.goal_dynimgs_u8_into(x_in[0], x_in[1], x_in[2], rgb.table, inp['target_rgb'].table, K, N, HW, self.dyn_ws2,
                                 depth=dep, tgt_depth=inp['target_depth'], dsample_stride=K * HW, dframe_stride=HW)
      elif self.last_from_dynimg:
        ops.goal_dynimgs_into(x_in[0], x_in[1], x_in[2], rgb, inp['target_rgb'], K, N, HW, self.dyn_ws2, K * HW * 3, HW * 3,
                              depth=dep, tgt_depth=inp['target_depth'], dsample_stride=K * HW, dframe_stride=HW)
      else:
        cur_rgb, cur_dep = rgb[:, K - 1], dep[:, K - 1]
        ops.pack_pixels_into(x_in[0], cur_rgb, K * HW * 3, N, HW, 3, 4, cur_dep, K * HW, 1)
        ops.dynimg_rgbd_into(x_in[1], rgb, dep, K, N, HW, self.dyn_ws, K * HW * 3, HW * 3, K * HW, HW)
        ops.dynimg_rgbd_into(x_in[2], cur_rgb, cur_dep, 2, N, HW, self.dyn_ws, K * HW * 3, 0, K * HW, 0,
                             rgb2=inp['target_rgb'], depth2=inp['target_depth'])
      self._encode_dynimg_state()
      d.forward(backward_too)
      self._finish_forward()
      return
    frames, tgt = (None, None) if u8 else self._frames()
    if self.mode == 'dynimg':
      cur = None if u8 else frames[:, K - 1]                  # rgb_frame_list[-1] (graph.py:387)
      # g0: current frame;  g1: dynimg(buffer) (:392);  g2: dynimg([cur, tgt]) (:397-400)
      if u8:
        ops.goal_dynimgs_u8_into(x_in[0], x_in[1], x_in[2], self.inputs['rgb'].table, self.inputs['target_rgb'].table, K, N,
                                 HW, self.dyn_ws2)
      elif C == 3 and HW % 4 == 0 and self.last_from_dynimg:
        # ONE launch: the pass over the window has the current frame in registers (its channel-padded copy and the pair image come
        # from there) and keeps both images in registers across their per-sample min / max
        ops.goal_dynimgs_into(x_in[0], x_in[1], x_in[2], frames, tgt, K, N, HW, self.dyn_ws2, K * HW * C, HW * C)
      else:
        ops.pack_pixels_into(x_in[0], cur, K * HW * C, N, HW, C, 4)
        ops.dynimg_into(x_in[1], frames, K, N, HW, C, 4, self.dyn_ws, K * HW * C, HW * C)
        ops.dynimg_into(x_in[2], cur, 2, N, HW, C, 4, self.dyn_ws, K * HW * C, 0, frames2=tgt)
      self._encode_dynimg_state()
    elif self.mode in ('seq_constant', 'seq_residual'):
      xs = x_in[0].view(K + 1, N, H, W, 4)                    # time-major; slot K = target frame
      for t in range(K):
        ops.pack_pixels_into(xs[t], frames[:, t], K * HW * C, N, HW, C, 4)
      ops.pack_pixels_into(xs[K], tgt, HW * C, N, HW, C, 4)
      self.enc.forward()
      feats = self.enc.features[0].view(K + 1, N, _CELLS, self.feat_ch[0])
      for t in range(K):
        if self.mode == 'seq_constant':   # representation_concatenation: [obs | jnt | tgt] (:146-167, 367)
          ops.state_concat_fwd_into(d.states[t], [feats[t], feats[K]], self.feat_ch, 1, jnts[:, t], K * jn, jn, N,
                                    _CELLS, d.D)
        else:                             # state_concatenation(tgt_feat - feat, jnt) (:369-370)
          ops.state_concat_fwd_into(d.states[t], [feats[t]], self.feat_ch, 1, jnts[:, t], K * jn, jn, N, _CELLS, d.D,
                                    sub_from=feats[K])
    else:                                 # seq_dyndiff
      xs = x_in.view(2, K, N, H, W, 4)
      for t in range(K):
        ops.pack_pixels_into(xs[0][t], frames[:, t], K * HW * C, N, HW, C, 4)
        ops.dynimg_into(xs[1][t], frames[:, t], 2, N, HW, C, 4, self.dyn_ws, K * HW * C, 0, frames2=tgt)   # :373-376
      self.enc.forward()
      f0 = self.enc.features[0].view(K, N, _CELLS, self.feat_ch[0])
      f1 = self.enc.features[1].view(K, N, _CELLS, self.feat_ch[1])
      for t in range(K):                  # representation_concatenation(feat, tgt_feat, jnt) (:381)
        ops.state_concat_fwd_into(d.states[t], [f0[t], f1[t]], self.feat_ch, 1, jnts[:, t], K * jn, jn, N, _CELLS, d.D)
    d.forward(backward_too)
    self._finish_forward()

  def backward(self, part=None, adam_prepare=False, defer_sums=None, before_bottom=None):
    """part None = whole backward; 'upper' / 'bottom' = the two halves the data-parallel runner captures
    separately (runtime.py): everything down to conv3, then the encoder bottom (conv2 / conv1).  ``adam_prepare``: see
    _prepare_args (the optimiser's scalars ride in the slab-sum launch of the part it is passed to -- once per step).
    ``defer_sums`` (part 'upper') / ``before_bottom`` (part 'bottom'): ConvEncoderStack.backward, used by backward_and_apply."""
    if part == 'bottom':
      self.enc.backward(hi=ConvEncoderStack.SPLIT - 1, lo=0, prepare=self._prepare_args(adam_prepare), lead_dgrad=ConvEncoderStack.SPLIT if ConvEncoderStack.DEFER_SPLIT_DGRAD else None,
                        before_bottom=before_bottom)
      return
    N, K, jn = self.N, self.K, self.cfg.dim_jnt_state
    d = self.decoder
    if self.mode == 'dynimg':
      feats, dfe = self.enc.features, self.enc.dfeatures
      cc = dict(feats=[feats[0], feats[1], feats[2]], dfeats=[dfe[0], dfe[1], dfe[2]], feat_ch=self.feat_ch, jnt_pos=2, J=jn,
                cells=_CELLS)
      if not d.backward(concat=cc):
        ops.state_concat_bwd_into(cc['dfeats'], d.dstates[0], d.D, cc['feats'], self.feat_ch, 2, jn, N, _CELLS)
    elif self.mode in ('seq_constant', 'seq_residual'):
      d.backward()
      ch = self.feat_ch[0]
      feats = self.enc.features[0].view(K + 1, N, _CELLS, ch)
      dfe = self.enc.dfeatures[0].view(K + 1, N, _CELLS, ch)
      for t in range(K):
        if self.mode == 'seq_constant':
          ops.state_concat_bwd_into([dfe[t], None], d.dstates[t], d.D, [feats[t], feats[K]], self.feat_ch, 1, jn, N, _CELLS)
          ops.state_concat_bwd_into([None, dfe[K]], d.dstates[t], d.D, [feats[t], feats[K]], self.feat_ch, 1, jn, N,
                                    _CELLS, accumulate=t > 0)
        else:   # d(tgt - feat): -1 into feat_t, +1 (summed over the window) into the target features
          ops.state_concat_bwd_into([dfe[t]], d.dstates[t], d.D, [feats[t]], self.feat_ch, 1, jn, N, _CELLS, scale=-1.0)
          ops.state_concat_bwd_into([dfe[K]], d.dstates[t], d.D, [feats[K]], self.feat_ch, 1, jn, N, _CELLS,
                                    accumulate=t > 0, scale=1.0)
    else:
      d.backward()
      f = [self.enc.features[g].view(K, N, _CELLS, self.feat_ch[g]) for g in range(2)]
      df = [self.enc.dfeatures[g].view(K, N, _CELLS, self.feat_ch[g]) for g in range(2)]
      for t in range(K):
        ops.state_concat_bwd_into([df[0][t], df[1][t]], d.dstates[t], d.D, [f[0][t], f[1][t]], self.feat_ch, 1, jn, N, _CELLS)
    self.enc.backward(hi=7, lo=ConvEncoderStack.SPLIT if part == 'upper' else 0,
                      prepare=self._prepare_args(adam_prepare and defer_sums is None), defer_dgrad=part == 'upper' and ConvEncoderStack.DEFER_SPLIT_DGRAD,
                      defer_sums=defer_sums if part == 'upper' else None)

  def endpoints(self):
    """dynbuff / dyndiff debug endpoints (graph.py:377,393,401): the LAST computed images."""
    C, K, N = self.C, self.K, self.N
    self.check_device_errors()
    ep = {'conv8': self.enc.features}
    if self.mode == 'dynimg':
      ep['dynbuff'] = self.enc.x_in[1][..., :C]
      ep['dyndiff'] = self.enc.x_in[2][..., :C]
    elif self.mode == 'seq_dyndiff':
      ep['dyndiff'] = self.enc.x_in.view(2, K, N, self.H, self.W, 4)[1][K - 1][..., :C]
    return ep


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
    self._begin_step(backward_too)
    N, K, H, W, C = self.N, self.K, self.H, self.W, self.C
    HW = H * W
    x_in = self.enc.x_in[0].view(K, N, H, W, 4)
    frames, _ = self._frames()
    for t in range(K):
      ops.pack_pixels_into(x_in[t], frames[:, t], K * HW * C, N, HW, C, 4)
    self.enc.forward()
    feats = self.enc.features[0].view(K, N, _CELLS, 256)
    jn = self.cfg.dim_jnt_state
    d = self.decoder
    for t in range(K):   # state_concatenation (graph.py:123-144)
      ops.state_concat_fwd_into(d.states[t], [feats[t]], [256], 1, self.inputs['jnt_state'][:, t], K * jn, jn, N,
                                _CELLS, d.D)
    d.forward(backward_too)
    self._finish_forward()

  def backward(self, part=None, adam_prepare=False, defer_sums=None, before_bottom=None):
    if part == 'bottom':
      self.enc.backward(hi=ConvEncoderStack.SPLIT - 1, lo=0, prepare=self._prepare_args(adam_prepare), lead_dgrad=ConvEncoderStack.SPLIT if ConvEncoderStack.DEFER_SPLIT_DGRAD else None,
                        before_bottom=before_bottom)
      return
    N, K = self.N, self.K
    d = self.decoder
    d.backward()
    feats = self.enc.features[0].view(K, N, _CELLS, 256)
    dfe = self.enc.dfeatures[0].view(K, N, _CELLS, 256)
    for t in range(K):
      ops.state_concat_bwd_into([dfe[t]], d.dstates[t], d.D, [feats[t]], [256], 1, self.cfg.dim_jnt_state, N, _CELLS)
    self.enc.backward(hi=7, lo=ConvEncoderStack.SPLIT if part == 'upper' else 0,
                      prepare=self._prepare_args(adam_prepare and defer_sums is None), defer_dgrad=part == 'upper' and ConvEncoderStack.DEFER_SPLIT_DGRAD,
                      defer_sums=defer_sums if part == 'upper' else None)

  def endpoints(self):
    return {'conv8': self.enc.features}
