"""Estimator-style training harness and the two ``model_fn``s.

Mirrors the call surface that the reference's ``scripts/train_e2evmc.py:210-291`` touches:
``Estimator(model_fn=, model_dir=, config=, params=)``, ``.train(input_fn=)``,
``.evaluate(input_fn=) -> {'loss', 'cmd_ee', 'pos_ee', 'pos_obj', 'cmd_grp', 'global_step'}``,
``RunConfig``, ``GPUOptions`` / ``ConfigProto``, ``ModeKeys``, ``EstimatorSpec``,
``latest_checkpoint`` -- and ``e2evmc_model_fn`` / ``goal_e2evmc_model_fn`` with the signature
``model_fn(features, labels, mode, params)`` of ``src/models/e2evmc/estimator.py:14,144``.

Graph-mode semantics are kept: ``model_fn`` is called ONCE per (mode, batch size) with the static
device buffers that will hold every batch; it builds the model on the HIP kernels and returns an
``EstimatorSpec`` whose ``train_op`` replays the captured step.  The Estimator's loop then only
copies each batch into those buffers and calls ``train_op()``  (= ``session.run(train_op)``).
State lives in ``model_dir`` as ``model.ckpt-<step>.pt`` + a TF-style ``checkpoint`` index file.
"""
from __future__ import annotations

import collections
import json
import os
import re
import time

import numpy as np
import torch

from . import dist as gdist
from . import graph
from .runtime import EvalStepRunner, TrainStepRunner, dp_form_kwargs


class ModeKeys:
  TRAIN = 'train'
  EVAL = 'eval'
  PREDICT = 'infer'


class GPUOptions:
  """tf.GPUOptions(allow_growth, per_process_gpu_memory_fraction) (train_e2evmc.py:217-219)."""

  def __init__(self, allow_growth=True, per_process_gpu_memory_fraction=None):
    self.allow_growth = allow_growth
    self.per_process_gpu_memory_fraction = per_process_gpu_memory_fraction


class ConfigProto:

  def __init__(self, gpu_options=None):
    self.gpu_options = gpu_options or GPUOptions()


class RunConfig:
  """tf.estimator.RunConfig(session_config, save_checkpoints_steps, keep_checkpoint_max) (train_e2evmc.py:221-224)."""

  def __init__(self, session_config=None, save_checkpoints_steps=None, keep_checkpoint_max=5, device=None,
               use_hipgraph=True, init_seed=0, save_tf_bundle=False, dp_form=None):
    self.session_config = session_config or ConfigProto()
    self.save_checkpoints_steps = save_checkpoints_steps
    self.keep_checkpoint_max = keep_checkpoint_max
    self.device = device
    self.use_hipgraph = use_hipgraph
    self.init_seed = init_seed
    # also write every checkpoint as a TF-1.15 tensor bundle (model.ckpt-<step>.index / .data-00000-of-00001 with the
    # reference's variable names, Adam slots and global_step): what tf.train.Saver consumers such as the reference's
    # predictor (predictor.py:85-95) and _export_snapshot (train_e2evmc.py:160-181) expect to find
    self.save_tf_bundle = save_tf_bundle
    # data parallel only: the form of the step (runtime.DP_FORMS).  None = 'three_graphs_reserve16' -- every
    # collective an ordinary RCCL launch between three replayed graphs (safe with ragged epochs by construction), part 2's persistent
    # kernels leaving 16 CUs to RCCL's workgroups, which cannot share a CU with them (profiles/HARDWARE_FINDINGS.md 37).  'overlap'
    # (the whole step incl. both all-reduces as ONE hipGraph), 'overlap_reserve16/32', 'serial', 'three_graphs_serial' are
    # opt-in: take the one bench.py reports as fastest on the node at hand (comm.step_ms / comm.timed_form of an N > 1 run)
    self.dp_form = dp_form


EstimatorSpec = collections.namedtuple(
    'EstimatorSpec', ['mode', 'loss', 'train_op', 'eval_metric_ops', 'predictions', 'training_hooks',
                      'evaluation_hooks', 'model'])
EstimatorSpec.__new__.__defaults__ = (None,) * 7


# ================================================================================================
# checkpoints (TF-style names so that _export_snapshot-like tooling works unchanged)
# ================================================================================================
_BUNDLE_SUFFIXES = ('.index', '.data-00000-of-00001')


def latest_checkpoint(model_dir):
  """tf.train.latest_checkpoint: '<model_dir>/model.ckpt-<step>' or None (train_e2evmc.py:160).  The prefix counts
  when its native file (.pt) or its TF tensor bundle (.index) exists."""
  index = os.path.join(model_dir, 'checkpoint')
  if not os.path.exists(index):
    return None
  with open(index) as f:
    m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
  if not m:
    return None
  path = os.path.join(model_dir, os.path.basename(m.group(1)))
  return path if (os.path.exists(path + '.pt') or os.path.exists(path + '.index')) else None


def save_checkpoint(store, model_dir, keep_max, tf_bundle=False, lstm_batch=None):
  step = int(store.global_step.item())
  name = 'model.ckpt-%d' % step
  tmp = os.path.join(model_dir, name + '.pt.tmp')
  torch.save(store.state_dict(), tmp)
  os.replace(tmp, os.path.join(model_dir, name + '.pt'))
  if tf_bundle:
    from . import tf_checkpoint
    mem = [n.rsplit('/', 2)[0] + '/lstm_memory' for n in store.shapes if n.endswith('/lstm_cell/kernel')]
    tf_checkpoint.export_checkpoint(store, os.path.join(model_dir, name), lstm_memory_name=mem[0] if mem else None,
                                    batch_size=lstm_batch)
  existing = sorted((int(re.match(r'model\.ckpt-(\d+)\.pt$', fn).group(1)) for fn in os.listdir(model_dir)
                     if re.match(r'model\.ckpt-(\d+)\.pt$', fn)))
  if keep_max and len(existing) > keep_max:
    for s in existing[:-keep_max]:
      for suffix in ('.pt',) + _BUNDLE_SUFFIXES:
        fn = os.path.join(model_dir, 'model.ckpt-%d%s' % (s, suffix))
        if os.path.exists(fn):
          os.remove(fn)
    existing = existing[-keep_max:]
  with open(os.path.join(model_dir, 'checkpoint'), 'w') as f:
    f.write('model_checkpoint_path: "%s"\n' % name)
    for s in existing:
      f.write('all_model_checkpoint_paths: "model.ckpt-%d"\n' % s)
  return os.path.join(model_dir, name)


def load_checkpoint(store, prefix):
  """Native .pt file when present, else the TF-1.15 tensor bundle of the same prefix (parameters, Adam slots, step)."""
  if os.path.exists(prefix + '.pt'):
    store.load_state_dict(torch.load(prefix + '.pt', map_location='cpu'))
  else:
    from . import tf_checkpoint
    tf_checkpoint.import_checkpoint(store, prefix, load_optimizer=True)


# ================================================================================================
# model_fn (estimator.py:14-141 and 144-279)
# ================================================================================================
class _SummarySaverHook:
  """Counterpart of the per-loss-term SummarySaverHooks (estimator.py:305-313): every ``log_steps`` steps the loss
  terms go to a TensorBoard event file (<model_dir>/events.out.tfevents.*, tags ``loss`` and ``loss_<term>`` like the
  reference's tf.summary.scalar names) and as one JSON line to <model_dir>/events.jsonl."""

  _writers = {}     # one event file per model_dir and process

  def __init__(self, model, every):
    self.model, self.every = model, max(int(every), 1)

  @classmethod
  def _writer(cls, model_dir):
    if model_dir not in cls._writers:
      from .summary import EventFileWriter
      cls._writers[model_dir] = EventFileWriter(model_dir)
    return cls._writers[model_dir]

  def after_run(self, step, model_dir):
    if step % self.every or gdist.rank() != 0:
      return
    self.model.check_device_errors()        # (the loss read-out below waits for the device anyway)
    parts = {k: float(v) for k, v in self.model.loss_parts().items()}
    parts['global_step'] = step
    parts['wall_time'] = time.time()
    if model_dir:
      with open(os.path.join(model_dir, 'events.jsonl'), 'a') as f:
        f.write(json.dumps(parts) + '\n')
      self._writer(model_dir).add_scalars({k: v for k, v in parts.items() if k.startswith('loss')}, step, parts['wall_time'])
    print('INFO: loss = %.6f, step = %d' % (parts['loss'], step), flush=True)


def _model_fn(features, labels, mode, params, goal):
  cfg = params['e2evmc_config']
  if cfg.img_channels not in (3, 4):
    raise ValueError("Unsupported number of channels for input frame: %d!" % cfg.img_channels)
  if mode not in (ModeKeys.TRAIN, ModeKeys.EVAL, ModeKeys.PREDICT):
    raise RuntimeError("Unknown estimator mode: %s" % (mode,))
  rgb = features['rgb']
  if rgb.device.type != 'cuda':
    raise RuntimeError('geeco_amd needs a GPU: model_fn got features on %s (no CPU fallback)' % rgb.device)
  N = int(rgb.shape[0])
  training = mode == ModeKeys.TRAIN
  ctor = graph.GoalE2EVMC if goal else graph.E2EVMC
  model = ctor(cfg, N, rgb.device, training=training, store=params.get('_variable_store'))
  # adopt the caller's static buffers as the model inputs (placeholders)
  source = lambda k: labels.get(k) if (labels is not None and k in model.label_keys) else features.get(k)
  # all of the model's image inputs that CAN come as window addresses must, or none does
  take_u8 = bool(model.u8_window_keys) and all(getattr(source(k), 'u8', False) for k in model.u8_window_keys)
  for k in list(model.inputs.keys()):
    src = source(k)
    if src is None:
      if mode == ModeKeys.PREDICT and k in model.label_keys + ['ee_state', 'obj_state']:
        continue   # label-side inputs are not needed for predictions
      raise KeyError("model_fn: missing input '%s'" % k)
    if hasattr(src, 'pointers'):
      # input_fn.WindowFeed (windows of HBM-resident episodes): a model whose input kernel follows window addresses takes the
      # address table and the fp32 windows are never written; any other model gets the dense buffer the gather fills
      if tuple(src.shape) != tuple(model.inputs[k].shape):
        raise ValueError("model_fn: input '%s' must have shape %s, got %s" % (k, tuple(model.inputs[k].shape), tuple(src.shape)))
      if take_u8 and k in model.u8_window_keys:
        src.pointers()
        model.inputs[k] = src
      else:
        model.inputs[k] = src.dense()
    elif (src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and
        tuple(src.shape) == tuple(model.inputs[k].shape)):
      model.inputs[k] = src
    else:
      raise ValueError("model_fn: input '%s' must be a contiguous float32 device tensor of shape %s" %
                       (k, tuple(model.inputs[k].shape)))
  model._bind_labels()
  # the reference computes `reset` from features['step'] (estimator.py:41-42); it is numerically
  # inert (both tf.cond branches are zeros, graph.py:218-220), so it is validated and dropped.
  if 'step' in features and features['step'].dtype != torch.int64:
    raise ValueError("features['step'] must be int64")
  print('>>> Graph Summary (%d trainable parameters):' % (model.store.count_parameters(),))
  if mode == ModeKeys.TRAIN:
    runner = TrainStepRunner(model, use_graph=params.get('use_hipgraph', True), **dp_form_kwargs(params.get('dp_form')))
    hooks = [_SummarySaverHook(model, params.get('log_steps', 1000))]
    return EstimatorSpec(mode=mode, loss=model.loss, train_op=runner.step, training_hooks=hooks, model=model)
  runner = EvalStepRunner(model, use_graph=params.get('use_hipgraph', True))
  if mode == ModeKeys.EVAL:
    return EstimatorSpec(mode=mode, loss=model.loss, train_op=runner.step, eval_metric_ops=_eval_metric_fn(model),
                         evaluation_hooks=[], model=model)
  return EstimatorSpec(mode=mode, predictions=model.predictions(), train_op=runner.step, model=model)


def e2evmc_model_fn(features, labels, mode, params):
  """Unconditional reflex (scope 'VMC'); reference estimator.py:14-141."""
  return _model_fn(features, labels, mode, params, goal=False)


def goal_e2evmc_model_fn(features, labels, mode, params):
  """Goal-conditioned controller (scope 'GoalVMC'); reference estimator.py:144-279."""
  return _model_fn(features, labels, mode, params, goal=True)


def _eval_metric_fn(model):
  """Per-batch sufficient statistics for tf.metrics.mean_squared_error / accuracy
  (estimator.py:246-254): returns a callable giving {key: (sum, count)} device tensors."""
  K = model.K

  def batch_stats():
    p = model.predictions()
    inp = model.inputs
    tgt = {'pos_ee': inp['ee_state'][:, K - 1, :3], 'pos_obj': inp['obj_state'][:, K - 1, :3]}
    if model.cfg.control_mode == 'cartesian':
      tgt['cmd_ee'] = inp['cmd'][:, :3]
    else:                                                     # estimator.py:117-120 / 255-258
      tgt.update({'cmd_vel': inp['vel_target'], 'cmd_ee': inp['ee_target'][:, :3], 'cmd_grp': inp['grp_target']})
    out = {}
    for k, t in tgt.items():
      out[k] = (((p[k] - t) ** 2).sum(), float(t.numel()))
    if model.cfg.control_mode == 'cartesian':
      label = torch.round(inp['cmd'][:, 3]).to(torch.int64) + 1
      out['cmd_grp'] = ((p['logits_cmd_grp'].argmax(dim=-1) == label).float().sum(), float(label.numel()))
    return out
  return batch_stats


class _FeedDict(dict):
  """name -> static buffer (a view of the spec's FeedArena, or an input_fn.WindowFeed) of the features or the labels."""

  def __init__(self, arena, tag, items):
    super().__init__(items)
    self.arena, self.tag = arena, tag


# ================================================================================================
# Estimator
# ================================================================================================
class Estimator:
  """tf.estimator.Estimator counterpart (train_e2evmc.py:260-264, 284-291)."""

  def __init__(self, model_fn, model_dir=None, config=None, params=None):
    self._model_fn = model_fn
    self.model_dir = model_dir
    self.config = config or RunConfig()
    self.params = dict(params or {})
    self._specs = {}          # (mode, N) -> (spec, feature buffers, label buffers)
    self.last_train_stats = None
    self._store = None
    self._restored = False
    if model_dir and gdist.rank() == 0:
      os.makedirs(model_dir, exist_ok=True)
    frac = getattr(self.config.session_config.gpu_options, 'per_process_gpu_memory_fraction', None)
    if frac and torch.cuda.is_available() and 0.0 < frac < 1.0:
      torch.cuda.set_per_process_memory_fraction(float(frac))   # --memcap (train_e2evmc.py:102-104)

  # -- device / batches ------------------------------------------------------------------------
  def _device(self):
    if self.config.device is not None:
      return torch.device(self.config.device)
    if not torch.cuda.is_available():
      raise RuntimeError('geeco_amd.Estimator needs an MI355X (torch.cuda.is_available() is False); '
                         'there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())

  @staticmethod
  def _shard(batch, world, rank):
    """Slice of a GLOBAL batch for this rank (synthetic inputs / unsharded input_fns): contiguous, sizes differing by
    at most one, so a ragged global batch is kept (geeco_gym.py:471 has no drop_remainder).  Returns
    (features, labels, n_local, n_global); n_local may be 0."""
    feats, labels = batch
    n = int(feats['step'].shape[0]) if 'step' in feats else int(next(iter(feats.values())).shape[0])
    if world == 1:
      return feats, labels, n, n
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    if hi == lo:
      return None, None, 0, n
    sl = lambda d: {k: v[lo:hi] for k, v in d.items()} if d is not None else None
    return sl(feats), sl(labels), hi - lo, n

  def _local_batches(self, it, world, rank):
    """Yields (features, labels, n_local, n_global) per optimiser step.  An input_fn that already sharded the
    episodes over the ranks (pickplace_input_fn(shard=...)) carries ``dp_schedule`` = every rank's window count per
    step; otherwise each rank slices the same global batch."""
    sched = getattr(it, 'dp_schedule', None) if world > 1 else None
    if sched is None:
      for batch in it:
        yield self._shard(batch, world, rank)
      return
    src = iter(it)
    for counts in sched:
      if counts[rank] > 0:
        feats, labels = next(src)
        n = int(feats['step'].shape[0])
        if n != counts[rank]:
          raise RuntimeError('data-parallel schedule expected %d windows on rank %d, the input pipeline delivered %d'
                             % (counts[rank], rank, n))
        yield feats, labels, n, sum(counts)
      else:
        yield None, None, 0, sum(counts)

  def _get_spec(self, mode, feats, labels, n, loss_scale=1.0):
    """One model (+ captured graphs) per (mode, local batch size, loss scale).  ``loss_scale`` = n_local * world /
    n_global weights this rank's batch-mean loss so that the SUM all-reduce followed by 1/world is the mean over the
    GLOBAL batch also when the ranks hold different numbers of windows (ragged final batch); 1 for equal shards."""
    key = (mode, n) if loss_scale == 1.0 else (mode, n, round(float(loss_scale), 9))
    # uint8 and float32 resident frames are read by different input kernels (a float32 episode = one whose recorded values
    # were not integral): one model per frame type, sharing the variables
    key += tuple(k for k, v in sorted(feats.items()) if hasattr(v, 'is_u8') and not v.is_u8())
    if key in self._specs:
      return self._specs[key]
    dev = self._device()
    # one arena for everything the host writes per batch (states, labels, window address tables): one H2D copy per step
    from .input_fn import FeedArena, WindowFeed
    arena = FeedArena(dev)
    feeds = {}
    for tag, d in (('features', feats), ('labels', labels)):
      for k, v in (d or {}).items():
        if hasattr(v, 'materialize_into'):      # input_fn.DeviceWindows: windows of HBM-resident episodes
          feeds[tag, k] = WindowFeed(v, arena, (tag, k))
        elif isinstance(v, np.ndarray) and v.nbytes <= self._ARENA_MAX_BYTES:
          arena.reserve((tag, k), v.shape, v.dtype)
        else:                                   # device tensors (synthetic inputs), dense host windows: a buffer and a copy of their own
          feeds[tag, k] = torch.as_tensor(v).to(dev).contiguous()
    arena.seal()
    def to_dev(tag, d):
      if d is None:
        return None
      return _FeedDict(arena, tag, {k: feeds[tag, k] if (tag, k) in feeds else arena.view((tag, k)) for k in d})
    fbuf, lbuf = to_dev('features', feats), to_dev('labels', labels)
    params = dict(self.params)
    params['_variable_store'] = self._store
    params.setdefault('use_hipgraph', self.config.use_hipgraph)
    params.setdefault('dp_form', getattr(self.config, 'dp_form', None))
    spec = self._model_fn(fbuf, lbuf, mode, params)
    spec.model.decoder.loss_scale = float(loss_scale)
    if self._store is None:
      self._store = spec.model.store
      self._store.initialize(seed=self.config.init_seed)
    self._restore_once()
    # only the buffers the model adopted are fed per batch
    used = {id(v) for v in spec.model.inputs.values()}
    fbuf = _FeedDict(arena, 'features', {k: v for k, v in fbuf.items() if id(v) in used or
                                         (getattr(v, 'buffer', None) is not None and id(v.buffer) in used)})
    lbuf = _FeedDict(arena, 'labels', {k: v for k, v in (lbuf or {}).items() if id(v) in used})
    self._specs[key] = (spec, fbuf, lbuf)
    return self._specs[key]

  def _restore_once(self):
    if self._restored:
      return
    self._restored = True
    ckpt = latest_checkpoint(self.model_dir) if self.model_dir else None
    if ckpt:
      load_checkpoint(self._store, ckpt)
      print('INFO: restored parameters from %s' % ckpt)
    gdist.broadcast_variables(self._store)

  _ARENA_MAX_BYTES = 1 << 20      # host arrays up to this size share the arena's one copy; larger ones (dense windows) go alone

  @staticmethod
  def _feed(bufs, batch):
    """One dict of a batch into its static buffers; on its own, or inside a ``_feed_step`` that shares the arena's copy."""
    if batch is None:
      return
    alone = not bufs.arena.is_open
    if alone:
      bufs.arena.begin()
    for k, buf in bufs.items():
      src = batch[k]
      if hasattr(buf, 'feed'):                 # input_fn.WindowFeed: repoint the address table or gather into the dense buffer
        buf.feed(src)
      elif bufs.arena.has((bufs.tag, k)):
        bufs.arena.write((bufs.tag, k), src.detach().cpu().numpy() if torch.is_tensor(src) else src)
      elif hasattr(src, 'materialize_into'):
        src.materialize_into(buf.view((len(src), src.K) + src.frame_shape))
      else:
        buf.copy_(torch.as_tensor(src), non_blocking=True)
    if alone:
      bufs.arena.flush()

  @classmethod
  def _feed_step(cls, fbuf, lbuf, feats, labels):
    """Features and labels of one step: every host array through the arena's ONE staging block and copy."""
    fbuf.arena.begin()
    try:
      cls._feed(fbuf, feats)
      if lbuf is not None and labels is not None:
        cls._feed(lbuf, labels)
    finally:
      fbuf.arena.flush()

  # -- public API --------------------------------------------------------------------------------
  def latest_checkpoint(self):
    return latest_checkpoint(self.model_dir)

  def get_variable_value(self, name):
    return self._store.var(name).detach().cpu().numpy()

  def get_variable_names(self):
    return list(self._store.shapes.keys())

  def train(self, input_fn, steps=None, max_steps=None):
    world, rank = gdist.world_size(), gdist.rank()
    t0, nsteps, step = time.time(), 0, None
    last_runner = None
    source = input_fn()
    for feats, labels, n, n_global in self._local_batches(source, world, rank):
      if n == 0:
        # this rank has no window in this step (ragged end of the epoch): it still takes part in the gradient
        # exchange, with zeros, and applies the same update as the others
        if last_runner is None:
          raise RuntimeError('rank %d has no data in its first training step' % rank)
        last_runner.null_step()
        spec = None
      else:
        spec, fbuf, lbuf = self._get_spec(ModeKeys.TRAIN, feats, labels, n, loss_scale=n * world / float(n_global))
        self._feed_step(fbuf, lbuf, feats, labels)
        spec.train_op()
        last_runner = spec.train_op.__self__
      nsteps += 1
      if (spec is not None and spec.training_hooks) or self.config.save_checkpoints_steps:
        step = int(self._store.global_step.item()) if (nsteps == 1 or step is None) else step + 1
        for h in (spec.training_hooks if spec is not None else None) or []:
          h.after_run(step, self.model_dir)
        if (self.config.save_checkpoints_steps and step % self.config.save_checkpoints_steps == 0 and rank == 0
            and self.model_dir):
          save_checkpoint(self._store, self.model_dir, self.config.keep_checkpoint_max, self.config.save_tf_bundle,
                          self.params['e2evmc_config'].batch_size if 'e2evmc_config' in self.params else None)
      if steps is not None and nsteps >= steps:
        break
      if max_steps is not None and step is not None and step >= max_steps:
        break
    if hasattr(source, 'close'):
      source.close()      # an epoch left early must not leave reader threads filling a queue nobody drains
    if nsteps and torch.cuda.is_available():
      torch.cuda.synchronize()
      for sp, _, _ in self._specs.values():      # a device-side error of any model that ran this epoch (input-stage timeout)
        sp.model.check_device_errors()
    if nsteps and world > 1 and not gdist.replicas_identical(self._store):
      # data parallel: summed gradients + a deterministic optimiser keep every replica bitwise equal; if they are not, the exchange
      # lost or raced something and rank 0's checkpoint would be one replica's opinion
      raise RuntimeError('geeco_amd: the replicas differ after %d data-parallel steps (rank %d): the gradient exchange is broken '
                         '(dp_form=%s); nothing was saved' % (nsteps, rank, getattr(self.config, 'dp_form', None) or 'three_graphs_reserve16'))
    # wall time of the input + step loop alone (the checkpoint written below is not part of the data path)
    self.last_train_stats = {'steps': nsteps, 'loop_seconds': time.time() - t0}
    if nsteps and rank == 0 and self.model_dir:
      path = save_checkpoint(self._store, self.model_dir, self.config.keep_checkpoint_max, self.config.save_tf_bundle,
                          self.params['e2evmc_config'].batch_size if 'e2evmc_config' in self.params else None)
      print('INFO: saved %s after %d steps (%.1f steps/s)' % (path, nsteps, nsteps / max(time.time() - t0, 1e-9)))
    return self

  def evaluate(self, input_fn, steps=None):
    """Streams the eval metrics of estimator.py:246-254; 'loss' = mean of per-batch losses [TF1.15]."""
    world, rank = gdist.world_size(), gdist.rank()
    sums, nb, loss_sum = {}, 0, None
    source = input_fn()
    for feats, labels, n, _ in self._local_batches(source, world, rank):
      if n == 0:
        continue
      spec, fbuf, lbuf = self._get_spec(ModeKeys.EVAL, feats, labels, n)
      self._feed_step(fbuf, lbuf, feats, labels)
      spec.train_op()
      loss_sum = spec.loss.clone() if loss_sum is None else loss_sum + spec.loss
      for k, (s, c) in spec.eval_metric_ops().items():
        if k in sums:
          sums[k][0] += s
          sums[k][1] += c
        else:
          sums[k] = [s.clone(), c]
      nb += 1
      if steps is not None and nb >= steps:
        break
    if hasattr(source, 'close'):
      source.close()
    # metric keys are fixed by the control mode (estimator.py:246-258): a rank whose shard of the eval split is empty still
    # takes part in the reduction, with zeros, instead of leaving the others waiting
    cfg = self.params.get('e2evmc_config')
    keys = sorted(sums) if sums else sorted(['cmd_ee', 'pos_ee', 'pos_obj'] + (
        ['cmd_grp'] if (cfg is None or cfg.control_mode == 'cartesian') else ['cmd_vel', 'cmd_grp']))
    if nb == 0 and world == 1:
      raise RuntimeError('evaluate(): input_fn produced no batches')
    dev = loss_sum.device if loss_sum is not None else self._device()
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    vec = torch.stack([loss_sum if loss_sum is not None else zero, torch.tensor(float(nb), device=dev)] +
                      [x for k in keys for x in ((sums[k][0], torch.tensor(float(sums[k][1]), device=dev)) if k in sums
                                                 else (zero, zero))])
    if world > 1:
      torch.distributed.all_reduce(vec)
    vec = vec.double().cpu()
    for sp, _, _ in self._specs.values():
      sp.model.check_device_errors()
    if float(vec[1]) == 0.0:
      raise RuntimeError('evaluate(): input_fn produced no batches on any rank')
    out = {'loss': float(vec[0] / vec[1])}
    for i, k in enumerate(keys):
      out[k] = float(vec[2 + 2 * i] / vec[3 + 2 * i])
    out['global_step'] = int(self._store.global_step.item())
    return out

  def predict(self, input_fn):
    for batch in input_fn():
      feats = batch[0] if isinstance(batch, tuple) else batch
      n = int(next(iter(feats.values())).shape[0])
      spec, fbuf, _ = self._get_spec(ModeKeys.PREDICT, feats, None, n)
      self._feed_step(fbuf, None, feats, None)
      spec.train_op()
      torch.cuda.synchronize()
      yield {k: v.detach().cpu().numpy().copy() for k, v in spec.predictions.items()}
