"""Input pipelines: the GEECO pick&place dataset reader (encoding v4) and a synthetic generator.

Counterpart of the reference's ``src/data/geeco_gym.py`` live path: ``pickplace_input_fn`` (:234-279)
-> ``pickplace_input_fn_v4`` (:401-474) with ``_get_meta_v4`` (:283), ``_parse_v4`` (:291),
``_preprocess_states_v4`` (:317), ``_preprocess_targets_v3`` (:598), ``_window_v3`` (:615),
``_prepare_v4`` (:373) and ``_collect_tfrecords_v2`` (:780).  Same dataset directory layout, same
feature / label dictionaries (shapes and key names), same ordering semantics: record-level shuffle
in 'train' mode only, NO sample-level shuffle (it is commented out in the reference, :447-448), so
a batch holds consecutive windows of one episode; the final batch may be ragged (no drop_remainder).

Unlike the reference (which materialises every K-frame window on the host: 84 windows x 12.6 MB
for K = 16), an episode's frames are kept once and every batch is sliced from them; windows are
built per batch, not per episode.
"""
from __future__ import annotations

import collections
import json
import os
import queue
import threading

import numpy as np

from . import tfrecord

PickAndPlaceMetaV4 = collections.namedtuple('PickAndPlaceMetaV4', [
    'episode_length', 'img_height', 'img_width', 'monitored_joints', 'actuated_joints', 'monitored_mocaps',
    'monitored_objects', 'dim_cmd', 'dim_ctrl'])

_ARM_JOINTS = ['shoulder_pan_joint', 'shoulder_lift_joint', 'upperarm_roll_joint', 'elbow_flex_joint',
               'forearm_roll_joint', 'wrist_flex_joint', 'wrist_roll_joint']          # geeco_gym.py:336-344
_FINGER_JOINTS = ['l_gripper_finger_joint', 'r_gripper_finger_joint']                # geeco_gym.py:364-367


def get_meta_v4(dataset_dir):
  """geeco_gym.py:283-289."""
  with open(os.path.join(dataset_dir, 'meta', 'meta_info.json'), 'r') as fp:
    return PickAndPlaceMetaV4(**json.load(fp))


def collect_tfrecords(dataset_dir, split_name, mode):
  """geeco_gym.py:780-793: record file names listed in splits/<split>/<mode>.txt (or all of data/)."""
  record_dir = os.path.join(dataset_dir, 'data')
  if split_name is None and mode is None:
    names = [fn for fn in os.listdir(record_dir) if fn.endswith('.tfrecord.zlib')]
  else:
    with open(os.path.join(dataset_dir, 'splits', split_name, '%s.txt' % (mode,))) as fp:
      names = fp.read().split('\n')
  return [os.path.join(record_dir, fn) for fn in names if fn.endswith('.tfrecord.zlib')]


# ---- target frames of an episode for the controller loop (the predictor's set_goal) ------------------------------------
def _episode_stem(tfrecord_name):
  return os.path.basename(tfrecord_name).split('.')[0]


def _read_rgb_png(path):
  from PIL import Image      # only the controller-side loaders need an image decoder
  with Image.open(path) as im:
    return np.array(im, dtype=np.float32) / 255.0


def load_target_frame(dataset_dir, tfrecord_name, load_depth=True):
  """geeco_gym.py:179-192: the episode's goal image ``images/targets/rgb/<stem>.png`` as float32 [H, W, 3] in [0, 1];
  with ``load_depth`` the raw depth map ``images/targets/depth/<stem>.npy`` becomes a 4th channel ([H, W, 4], the layout
  ``GoalE2EVMCPredictor.set_goal`` takes).  ``tfrecord_name`` may be a path; the stem is the name up to the first dot."""
  stem = _episode_stem(tfrecord_name)
  frame = _read_rgb_png(os.path.join(dataset_dir, 'images', 'targets', 'rgb', stem + '.png'))
  if load_depth:
    depth = np.load(os.path.join(dataset_dir, 'images', 'targets', 'depth', stem + '.npy'))
    frame = np.concatenate([frame, np.expand_dims(depth, axis=-1)], axis=-1)
  return frame


def load_keyframes(dataset_dir, tfrecord_name):
  """geeco_gym.py:194-211: every key frame of the episode (``images/keyframes/{rgb,depth}/<stem>*`` in sorted order,
  rgb and depth files paired by position) as RGB-D float32 [H, W, 4] arrays."""
  stem = _episode_stem(tfrecord_name)
  rgb_dir = os.path.join(dataset_dir, 'images', 'keyframes', 'rgb')
  depth_dir = os.path.join(dataset_dir, 'images', 'keyframes', 'depth')
  rgb_files = sorted(f for f in os.listdir(rgb_dir) if f.startswith(stem))
  depth_files = sorted(f for f in os.listdir(depth_dir) if f.startswith(stem))
  frames = []
  for rf, df in zip(rgb_files, depth_files):
    depth = np.load(os.path.join(depth_dir, df))
    frames.append(np.concatenate([_read_rgb_png(os.path.join(rgb_dir, rf)), np.expand_dims(depth, axis=-1)], axis=-1))
  return frames


def load_target_frames(dataset_dir, tfrecord_name, load_depth=True):
  """geeco_gym.py:165-177: the key frames when ``data/key_frames_<id>.json`` exists for the record (id = the first run of
  digits in the name), else the single goal image, as a list."""
  import re
  record_id = re.search(r'\d+', tfrecord_name).group(0)
  if os.path.exists(os.path.join(dataset_dir, 'data', 'key_frames_%s.json' % (record_id,))):
    return load_keyframes(dataset_dir, tfrecord_name)
  return [load_target_frame(dataset_dir, tfrecord_name, load_depth)]


def load_episode(path, meta, fetch_target, raw_rgb=False):
  """One episode -> dict of per-frame arrays after _parse_v4 + _preprocess_states_v4 +
  _preprocess_targets_v3 (i.e. the last frame already dropped: T = episode_length - 1).
  ``raw_rgb``: keep 'rgb' / 'target_rgb' as the recorded 0..255 values (the device path divides by
  255 on the GPU)."""
  H, W = meta.img_height, meta.img_width
  payload = next(iter(tfrecord.read_records(path, 'zlib')))
  _, fl = tfrecord.parse_sequence_example(payload)

  def stack(key, shape=None, dtype=np.float32):
    if key not in fl:
      raise KeyError("%s: feature list '%s' missing" % (path, key))
    arr = np.stack([np.asarray(f, dtype=dtype) for f in fl[key]], axis=0)
    return arr.reshape((arr.shape[0],) + tuple(shape)) if shape is not None else arr

  ex = {
      'step': stack('step', (), np.int64),
      'ts': stack('ts', ()),
      'rgb': stack('rgb', (H, W, 3)) / np.float32(1.0 if raw_rgb else 255.0),   # :312 RGB recorded as uint8 0..255
      'depth': stack('depth', (H, W, 1)),
      'cmd': stack('cmd', (meta.dim_cmd,)),
      'ctrl': stack('ctrl', (meta.dim_ctrl,)),
      'ee_state': stack('mocap_qpos-robot0:mocap', (7,)),
      'goal_state': stack('goal_qpos', (7,)),
      'obj_state': stack('obj_qpos', (7,)),
  }
  ex['jnt_state'] = np.stack([stack('joint_qpos-robot0:%s' % j, ()) for j in _ARM_JOINTS], axis=1)
  ex['vel_state'] = np.stack([stack('joint_qvel-robot0:%s' % j, ()) for j in _ARM_JOINTS], axis=1)
  ex['grp_state'] = np.stack([stack('joint_qpos-robot0:%s' % j, ()) for j in _FINGER_JOINTS], axis=1)
  target = None
  if fetch_target:      # :313-315: target = LAST frame of the full episode
    target = {'target_rgb': ex['rgb'][-1].copy(), 'target_depth': ex['depth'][-1].copy()}
  # _preprocess_targets_v3 (:598-613): next-frame states as targets, then drop the last frame
  ex['vel_target'] = np.roll(ex['vel_state'], -1, axis=0)
  ex['ee_target'] = np.roll(ex['ee_state'], -1, axis=0)
  ex['grp_target'] = np.roll(ex['grp_state'], -1, axis=0)
  ex = {k: v[:-1] for k, v in ex.items()}
  if target:
    ex.update(target)
  return ex


_FEATURE_KEYS = ['step', 'ts', 'rgb', 'depth', 'jnt_state', 'vel_state', 'ee_state', 'grp_state', 'goal_state',
                 'obj_state', 'cmd', 'ctrl']
_LABEL_KEYS = ['cmd', 'ctrl', 'vel_target', 'ee_target', 'grp_target']


class DeviceWindows:
  """A batch of K-frame windows that lives in HBM as (episode frames, start indices) segments.

  Stands in for a dense [n, K, *frame_shape] float32 array in the features dict: the Estimator
  materialises it straight into the model's static input buffer with geeco_gather_windows (frames
  were uploaded once per episode, RGB as uint8), so no window ever crosses PCIe."""

  def __init__(self, K, frame_shape, divisor, squeeze_k=False):
    self.K, self.frame_shape, self.divisor, self.squeeze_k = K, tuple(frame_shape), float(divisor), squeeze_k
    self.segments = []      # (device tensor [T, frame_elems], np.int32 starts)
    self.n = 0

  def add(self, frames_dev, starts):
    starts = np.asarray(starts, np.int32)
    self.segments.append((frames_dev, starts))
    self.n += len(starts)

  @property
  def shape(self):
    return (self.n,) + (() if self.squeeze_k else (self.K,)) + self.frame_shape

  def __len__(self):
    return self.n

  @staticmethod
  def concat(a, b):
    out = DeviceWindows(a.K, a.frame_shape, a.divisor, a.squeeze_k)
    out.segments = a.segments + b.segments
    out.n = a.n + b.n
    return out

  def materialize_into(self, out):
    import torch
    from . import ops
    fe = int(np.prod(self.frame_shape))
    off = 0
    for frames_dev, starts in self.segments:
      n = len(starts)
      if frames_dev.device != out.device:
        raise RuntimeError('DeviceWindows: episode frames live on %s but the batch buffer on %s (each rank must '
                           'upload to its own GPU)' % (frames_dev.device, out.device))
      st = torch.as_tensor(starts, device=out.device)
      ops.gather_windows_into(out[off:off + n], frames_dev, st, n, self.K, fe, self.divisor)
      off += n

  def numpy(self):
    """Dense host copy (tests / debugging)."""
    import torch
    dev = self.segments[0][0].device
    out = torch.empty((self.n, self.K) + self.frame_shape, dtype=torch.float32, device=dev)
    self.materialize_into(out)
    torch.cuda.synchronize()
    arr = out.cpu().numpy()
    return arr[:, 0] if self.squeeze_k else arr


def _concat_feature(a, b):
  if isinstance(a, DeviceWindows):
    return DeviceWindows.concat(a, b)
  return np.concatenate([a, b], axis=0)


def episode_to_device(ex, device):
  """Uploads the image streams of one episode: RGB as uint8 when the recorded values are integral
  (they are: the recorder stores uint8 frames as float lists), depth as float32."""
  import torch
  from .runtime import CAPTURE_LOCK
  device = resolve_device(device)
  T = ex['rgb'].shape[0]
  dev = {}
  rgb = ex['rgb'].reshape(T, -1)
  as_u8 = bool(np.all(rgb == np.rint(rgb)) and rgb.min() >= 0 and rgb.max() <= 255)
  host = {'rgb': torch.as_tensor(rgb.astype(np.uint8) if as_u8 else rgb / np.float32(255.0)),
          'depth': torch.as_tensor(np.ascontiguousarray(ex['depth'].reshape(T, -1)))}
  dev['rgb_div'] = 255.0 if as_u8 else 1.0
  if 'target_rgb' in ex:
    t = ex['target_rgb'].reshape(1, -1)
    t_u8 = bool(np.all(t == np.rint(t)) and t.min() >= 0 and t.max() <= 255)
    host['target_rgb'] = torch.as_tensor(t.astype(np.uint8) if t_u8 else t / np.float32(255.0))
    dev['target_rgb_div'] = 255.0 if t_u8 else 1.0
    host['target_depth'] = torch.as_tensor(np.ascontiguousarray(ex['target_depth'].reshape(1, -1)))
  # allocations / synchronous copies from this (prefetch) thread must not fall into a hipGraph capture window of
  # the training thread (runtime.CAPTURE_LOCK)
  with CAPTURE_LOCK:
    for k, v in host.items():
      dev[k] = v.to(device)
  return dev


def resolve_device(device):
  """An explicit (type, index) device.  A bare 'cuda' means the CALLING thread's current device: resolve it
  before handing work to another thread (the current HIP device is thread-local and starts at 0 there)."""
  import torch
  d = torch.device(device)
  if d.type == 'cuda' and d.index is None:
    d = torch.device('cuda', torch.cuda.current_device())
  return d


def episode_windows(ex, window_size, starts, dev=None):
  """(features, labels) for the windows beginning at ``starts`` (_window_v3 :615-631, _prepare_v4 :373-399).
  With ``dev`` (episode_to_device) the image features are DeviceWindows instead of host arrays."""
  K = window_size
  idx = np.asarray(starts)[:, None] + np.arange(K)[None, :]
  if dev is not None:
    feats = {k: ex[k][idx] for k in _FEATURE_KEYS if k not in ('rgb', 'depth')}
    H, W = ex['rgb'].shape[1:3]
    for key, shp, div in (('rgb', (H, W, 3), dev['rgb_div']), ('depth', (H, W, 1), 1.0)):
      dw = DeviceWindows(K, shp, div)
      dw.add(dev[key], starts)
      feats[key] = dw
    if 'target_rgb' in dev:
      for key, shp, div in (('target_rgb', (H, W, 3), dev['target_rgb_div']), ('target_depth', (H, W, 1), 1.0)):
        dw = DeviceWindows(1, shp, div, squeeze_k=True)
        dw.add(dev[key], np.zeros(len(starts), np.int32))
        feats[key] = dw
    last = np.asarray(starts) + K - 1
    return feats, {k: ex[k][last] for k in _LABEL_KEYS}
  feats = {k: ex[k][idx] for k in _FEATURE_KEYS}
  if 'target_rgb' in ex:
    n = len(starts)
    feats['target_rgb'] = np.broadcast_to(ex['target_rgb'], (n,) + ex['target_rgb'].shape).copy()
    feats['target_depth'] = np.broadcast_to(ex['target_depth'], (n,) + ex['target_depth'].shape).copy()
  last = np.asarray(starts) + K - 1
  labels = {k: ex[k][last] for k in _LABEL_KEYS}
  return feats, labels


class _Prefetcher:
  """Runs an iterator factory in background threads (tf.data's num_parallel_calls / prefetch)."""

  def __init__(self, make_iter, depth, device=None):
    self._q = queue.Queue(maxsize=max(int(depth), 1))
    self._device = device
    self._t = threading.Thread(target=self._run, args=(make_iter,), daemon=True)
    self._t.start()

  def _run(self, make_iter):
    try:
      if self._device is not None and self._device.type == 'cuda':
        import torch
        torch.cuda.set_device(self._device)     # thread-local: a new thread starts on device 0 whatever LOCAL_RANK is
      for item in make_iter():
        self._q.put(('item', item))
      self._q.put(('end', None))
    except BaseException as e:   # surfaced in the consumer
      self._q.put(('error', e))

  def __iter__(self):
    while True:
      kind, val = self._q.get()
      if kind == 'item':
        yield val
      elif kind == 'end':
        return
      else:
        raise val


def pickplace_input_fn(dataset_dir, split_name, mode, encoding='v4', window_size=4, fetch_target=False,
                       shuffle_buffer=128, batch_size=1, num_epochs=1, num_threads=4, prefetch_size=4, seed=None,
                       shard=None, device=None):
  """Same signature as the reference's pickplace_input_fn (geeco_gym.py:234-279).  Returns an iterable of
  (features, labels) numpy batches.  ``shard = (rank, world)`` makes each data-parallel rank read a
  disjoint, rank-strided subset of the episodes.  ``device`` (e.g. 'cuda'): upload every episode's
  frames once and hand out image features as DeviceWindows (windows are gathered in HBM)."""
  if encoding != 'v4':
    # v1-v3 are dead code in the reference (undefined PickAndPlaceEncodingV1/2/3 -> NameError)
    raise KeyError(encoding)
  if dataset_dir.startswith('synthetic:'):
    return synthetic_from_spec(dataset_dir, mode, window_size, fetch_target, batch_size, seed)
  meta = get_meta_v4(dataset_dir)
  paths = collect_tfrecords(dataset_dir, split_name, mode)
  if mode == 'train':   # record-level shuffle only (:436-437)
    if shard is not None and seed is None:
      raise ValueError('sharded training input needs one seed shared by all ranks (they must agree on the episode order)')
    np.random.default_rng(seed).shuffle(paths)
  K = window_size
  dp_schedule = None
  if shard is not None:
    # every rank reads its own rank-strided subset of the episodes.  Episodes have the fixed length
    # of the meta file (pickplace.py:157), so each rank can work out how many windows EVERY rank contributes to
    # each step without talking to the others: dp_schedule[s][r] = windows of rank r in step s.
    rank, world = shard
    nwin = (meta.episode_length - 1) - K + 1
    per_rank = [len(paths[r::world]) * nwin * num_epochs for r in range(world)]
    steps = max(-(-w // batch_size) for w in per_rank) if per_rank else 0
    dp_schedule = [tuple(max(0, min(batch_size, w - s * batch_size)) for w in per_rank) for s in range(steps)]
    paths = paths[rank::world]
  if device is not None:
    device = resolve_device(device)
  print('[pickplace_input_fn_v4] #tfrecords: %d' % len(paths))

  def batches():
    carry_f, carry_l = None, None   # windows left over from the previous episode (batch() spans episodes)
    for _ in range(num_epochs):
      for path in paths:
        ex = load_episode(path, meta, fetch_target, raw_rgb=device is not None)
        dev = episode_to_device(ex, device) if device is not None else None
        T = ex['step'].shape[0]
        if shard is not None and T != meta.episode_length - 1:
          raise ValueError('%s holds %d frames, meta_info.json says %d: the data-parallel batch schedule assumes '
                           'fixed-length episodes' % (path, T + 1, meta.episode_length))
        nwin = T - K + 1
        pos = 0
        while pos < nwin:
          need = batch_size - (0 if carry_f is None else len(carry_f['step']))
          take = min(need, nwin - pos)
          f, l = episode_windows(ex, K, np.arange(pos, pos + take), dev)
          pos += take
          if carry_f is not None:
            f = {k: _concat_feature(carry_f[k], f[k]) for k in f}
            l = {k: np.concatenate([carry_l[k], l[k]], axis=0) for k in l}
            carry_f = carry_l = None
          if len(f['step']) == batch_size:
            yield f, l
          else:
            carry_f, carry_l = f, l
    if carry_f is not None:   # ragged final batch (dataset.batch without drop_remainder, :471)
      yield carry_f, carry_l

  it = _Prefetcher(batches, prefetch_size, device)
  it.dp_schedule = dp_schedule
  return it


# ------------------------------------------------------------------------------------------------
# synthetic data
# ------------------------------------------------------------------------------------------------
def synthetic_batches(batch_size, window_size, num_batches, img_hw=(256, 256), channels=3, fetch_target=True, seed=1234):
  """Seeded random batches with the feature / label dictionaries of _prepare_v4."""
  H, W = img_hw
  K = window_size

  def gen():
    r = np.random.default_rng(seed)
    for b in range(num_batches):
      N = batch_size
      f = {
          'step': (np.arange(K)[None, :] + r.integers(1, 80, size=[N, 1])).astype(np.int64),
          'ts': r.random([N, K], dtype=np.float32),
          'rgb': r.integers(0, 256, size=[N, K, H, W, 3]).astype(np.float32) / np.float32(255.0),
          'depth': (0.5 + 2.5 * r.random([N, K, H, W, 1], dtype=np.float32)),
          'jnt_state': r.standard_normal([N, K, 7]).astype(np.float32),
          'vel_state': r.standard_normal([N, K, 7]).astype(np.float32),
          'ee_state': 1.5 * r.random([N, K, 7], dtype=np.float32),
          'grp_state': 0.05 * r.random([N, K, 2], dtype=np.float32),
          'goal_state': 1.5 * r.random([N, K, 7], dtype=np.float32),
          'obj_state': 1.5 * r.random([N, K, 7], dtype=np.float32),
          'ctrl': r.standard_normal([N, K, 2]).astype(np.float32),
      }
      cmd = np.concatenate([0.3 * r.standard_normal([N, K, 3]), r.integers(-1, 2, size=[N, K, 1])], axis=2)
      f['cmd'] = cmd.astype(np.float32)
      if fetch_target:
        f['target_rgb'] = r.integers(0, 256, size=[N, H, W, 3]).astype(np.float32) / np.float32(255.0)
        f['target_depth'] = (0.5 + 2.5 * r.random([N, H, W, 1], dtype=np.float32))
      l = {'cmd': f['cmd'][:, -1], 'ctrl': f['ctrl'][:, -1], 'vel_target': r.standard_normal([N, 7]).astype(np.float32),
           'ee_target': r.random([N, 7], dtype=np.float32), 'grp_target': r.random([N, 2], dtype=np.float32)}
      yield f, l
  return gen


def synthetic_from_spec(spec, mode, window_size, fetch_target, batch_size, seed):
  """``--dataset_dir synthetic:<num_batches>[:<H>x<W>]`` (no dataset on disk; used by the benches and tests)."""
  parts = spec.split(':')
  nb = int(parts[1]) if len(parts) > 1 and parts[1] else 8
  hw = tuple(int(x) for x in parts[2].split('x')) if len(parts) > 2 else (256, 256)
  if mode != 'train':
    nb = max(1, nb // 4)
  base = 1234 if seed is None else seed
  return synthetic_batches(batch_size, window_size, nb, hw, 3, fetch_target, seed=base + (0 if mode == 'train' else 1))()


def write_episode(path, meta, frames_rgb_u8, depth, cmd, ctrl, joints_qpos, joints_qvel, mocap_qpos, obj_qpos,
                  goal_qpos, ts=None, task_goal='goal', task_object='object'):
  """Writes one episode in the reference's on-disk format (PickAndPlaceEncodingV4: geeco_gym.py:117-176,
  data_recorder.py:37-59,134-156).  Used to build fixtures and synthetic datasets, not by training."""
  T = frames_rgb_u8.shape[0]
  ctx = collections.OrderedDict([
      ('episode_length', np.array([meta.episode_length], np.int64)), ('img_height', np.array([meta.img_height], np.int64)),
      ('img_width', np.array([meta.img_width], np.int64)), ('monitored_joints', list(meta.monitored_joints)),
      ('actuated_joints', list(meta.actuated_joints)), ('monitored_mocaps', list(meta.monitored_mocaps)),
      ('monitored_objects', list(meta.monitored_objects)), ('dim_cmd', np.array([meta.dim_cmd], np.int64)),
      ('dim_ctrl', np.array([meta.dim_ctrl], np.int64)), ('task_goal', task_goal), ('task_object', task_object)])
  frames = []
  for t in range(T):
    fr = collections.OrderedDict()
    fr['step'] = np.array([t], np.int64)
    fr['ts'] = np.array([0.04 * t if ts is None else ts[t]], np.float32)
    fr['rgb'] = frames_rgb_u8[t]          # uint8 -> float list (tfrecord.py:73-74)
    fr['depth'] = depth[t].astype(np.float32)
    fr['cmd'] = cmd[t].astype(np.float32)
    fr['ctrl'] = ctrl[t].astype(np.float32)
    fr['goal_qpos'] = goal_qpos[t].astype(np.float32)
    fr['obj_qpos'] = obj_qpos[t].astype(np.float32)
    for j, name in enumerate(meta.monitored_joints):
      fr['joint_qpos-%s' % name] = np.array([joints_qpos[t, j]], np.float32)
      fr['joint_qvel-%s' % name] = np.array([joints_qvel[t, j]], np.float32)
    for name in meta.monitored_mocaps:
      fr['mocap_qpos-%s' % name] = mocap_qpos[t].astype(np.float32)
    for name in meta.monitored_objects:
      fr['object_qpos-%s' % name] = obj_qpos[t].astype(np.float32)
    frames.append(fr)
  tfrecord.write_records(path, [tfrecord.encode_sequence_example(ctx, frames)], 'zlib')
