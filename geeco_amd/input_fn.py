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


def _finish_episode(ex, fetch_target):
  """_parse_v4's target (:313-315: LAST frame of the full episode) + _preprocess_targets_v3 (:598-613: next-frame
  states as targets, then the last frame is dropped) on per-frame arrays of the whole episode."""
  target = None
  if fetch_target:
    target = {'target_rgb': ex['rgb'][-1].copy(), 'target_depth': ex['depth'][-1].copy()}
  ex['vel_target'] = np.roll(ex['vel_state'], -1, axis=0)
  ex['ee_target'] = np.roll(ex['ee_state'], -1, axis=0)
  ex['grp_target'] = np.roll(ex['grp_state'], -1, axis=0)
  ex = {k: v[:-1] for k, v in ex.items()}
  if target:
    ex.update(target)
  return ex


def load_episode_py(path, meta, fetch_target, raw_rgb=False):
  """``load_episode`` through the pure-Python TFRecord / protobuf reader of tfrecord.py (one thread, holds the GIL):
  the independent restatement the native reader is tested against; not used by the pipeline."""
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
  return _finish_episode(ex, fetch_target)


def load_episode(path, meta, fetch_target, raw_rgb=False, image_keys=('rgb', 'depth'), alloc=None):
  """One episode -> dict of per-frame arrays after _parse_v4 + _preprocess_states_v4 + _preprocess_targets_v3 (i.e.
  the last frame already dropped: T = episode_length - 1), read by the native reader (inflate, CRC, SequenceExample
  scan and the float -> array copies run without the GIL: ``num_threads`` of these calls proceed side by side).

  ``raw_rgb``: 'rgb' / 'target_rgb' keep the recorded 0..255 values — as a uint8 array when every value is integral
  (the recorder stores uint8 frames as float lists, tfrecord.py:73-74; the test runs inside the conversion pass), else
  as float32; the device path divides by 255 on the GPU.  Otherwise float32 / 255 (:312).
  ``image_keys``: which of the image streams to decode (an RGB-only model never reads 'depth': 26 MB per episode).
  ``alloc(nbytes) -> uint8 array``: where the image arrays go (pinned staging memory of the device path)."""
  H, W = meta.img_height, meta.img_width
  with tfrecord.EpisodeReader(path, 'zlib') as rd:
    T = rd.frames('step')
    ex = {
        'step': rd.i64('step', 1).reshape(T),
        'ts': rd.f32('ts', 1).reshape(T),
        'cmd': rd.f32('cmd', meta.dim_cmd),
        'ctrl': rd.f32('ctrl', meta.dim_ctrl),
        'ee_state': rd.f32('mocap_qpos-robot0:mocap', 7),
        'goal_state': rd.f32('goal_qpos', 7),
        'obj_state': rd.f32('obj_qpos', 7),
        'jnt_state': np.concatenate([rd.f32('joint_qpos-robot0:%s' % j, 1) for j in _ARM_JOINTS], axis=1),
        'vel_state': np.concatenate([rd.f32('joint_qvel-robot0:%s' % j, 1) for j in _ARM_JOINTS], axis=1),
        'grp_state': np.concatenate([rd.f32('joint_qpos-robot0:%s' % j, 1) for j in _FINGER_JOINTS], axis=1),
    }
    new = (lambda n, dt: np.empty(n, dt)) if alloc is None else (lambda n, dt: alloc(n * np.dtype(dt).itemsize).view(dt))
    if 'rgb' in image_keys:
      n = H * W * 3
      rgb, exact = rd.u8('rgb', n, out=new(T * n, np.uint8))
      if not exact:
        rgb = rd.f32('rgb', n, out=new(T * n, np.float32))
      if not raw_rgb:
        rgb = rgb.astype(np.float32) / np.float32(255.0)
      ex['rgb'] = rgb.reshape(T, H, W, 3)
    else:
      rd.frames('rgb')
      ex['rgb'] = _Omitted((T, H, W, 3), 'rgb')
    if 'depth' in image_keys:
      ex['depth'] = rd.f32('depth', H * W, out=new(T * H * W, np.float32)).reshape(T, H, W, 1)
    else:
      rd.frames('depth')
      ex['depth'] = _Omitted((T, H, W, 1), 'depth')
  return _finish_episode(ex, fetch_target)


class _Omitted:
  """Stands in for an image stream the caller chose not to decode (``image_keys``): has the shape, indexes like the
  array would (frames, windows), refuses to produce values."""

  def __init__(self, shape, key):
    self.shape, self.key, self.dtype = tuple(shape), key, np.dtype(np.float32)

  def __len__(self):
    return self.shape[0]

  def __getitem__(self, idx):
    lead = np.empty(self.shape[:1], np.bool_)[idx].shape
    return _Omitted(lead + self.shape[1:], self.key)

  def copy(self):
    return self

  def __array__(self, *a, **k):
    raise RuntimeError("feature '%s' was not decoded (image_keys / device_keys excluded it)" % self.key)


_FEATURE_KEYS = ['step', 'ts', 'rgb', 'depth', 'jnt_state', 'vel_state', 'ee_state', 'grp_state', 'goal_state',
                 'obj_state', 'cmd', 'ctrl']
_LABEL_KEYS = ['cmd', 'ctrl', 'vel_target', 'ee_target', 'grp_target']
_IMAGE_KEYS = ('rgb', 'depth')


class DeviceWindows:
  """A batch of K-frame windows that lives in HBM as (episode frames, start indices) segments.

  Stands in for a dense [n, K, *frame_shape] float32 array in the features dict: the Estimator
  materialises it straight into the model's static input buffer with geeco_gather_windows (frames
  were uploaded once per episode, RGB as uint8), so no window ever crosses PCIe."""

  def __init__(self, K, frame_shape, divisor, squeeze_k=False):
    self.K, self.frame_shape, self.divisor, self.squeeze_k = K, tuple(frame_shape), float(divisor), squeeze_k
    self.segments = []      # (device tensor [T, frame_elems], np.int32 starts, divisor of THIS segment's frames)
    self.n = 0

  def add(self, frames_dev, starts, divisor=None):
    """``divisor``: what the gather divides this segment's frames by (default: the constructor's).  One batch can hold
    episodes stored as uint8 (255) next to episodes kept as float32 (1): a batch that straddles two such episodes is
    gathered segment by segment with each one's own divisor (and is then not ``is_u8()``: dense path)."""
    starts = np.asarray(starts, np.int32)
    self.segments.append((frames_dev, starts, self.divisor if divisor is None else float(divisor)))
    self.n += len(starts)

  @property
  def shape(self):
    return (self.n,) + (() if self.squeeze_k else (self.K,)) + self.frame_shape

  def __len__(self):
    return self.n

  @staticmethod
  def concat(a, b):
    if (a.K, a.frame_shape, a.squeeze_k) != (b.K, b.frame_shape, b.squeeze_k):
      raise ValueError('DeviceWindows.concat: windows of different shapes (%s, %s)' % (a.shape[1:], b.shape[1:]))
    out = DeviceWindows(a.K, a.frame_shape, a.divisor, a.squeeze_k)
    out.segments = a.segments + b.segments
    out.n = a.n + b.n
    return out

  def is_u8(self):
    """Every segment is resident uint8 frames (the recorder's values; the consumer divides by 255)."""
    import torch
    return bool(self.segments) and all(
        f is not None and d == 255.0 and f.dtype == torch.uint8 and f.is_contiguous() for f, _, d in self.segments)

  def addresses(self, device):
    """int64 address of each window's first frame (WindowFeed.pointers(): the input kernel follows them)."""
    fe = int(np.prod(self.frame_shape))
    out = np.empty(self.n, np.int64)
    off = 0
    device = resolve_device(device)
    for frames_dev, starts, _ in self.segments:
      if frames_dev.device != device:
        raise RuntimeError('DeviceWindows: episode frames live on %s but the model on %s (each rank must upload to its '
                           'own GPU)' % (frames_dev.device, device))
      T = frames_dev.shape[0]
      if len(starts) and (int(starts.min()) < 0 or int(starts.max()) + self.K > T):
        raise IndexError('DeviceWindows: window [%d, %d) outside the %d resident frames' %
                         (int(starts.min()), int(starts.max()) + self.K, T))
      out[off:off + len(starts)] = frames_dev.data_ptr() + starts.astype(np.int64) * fe
      off += len(starts)
    return out

  def materialize_into(self, out):
    import torch
    from . import ops
    fe = int(np.prod(self.frame_shape))
    off = 0
    for frames_dev, starts, divisor in self.segments:
      n = len(starts)
      if frames_dev is None:
        raise RuntimeError('DeviceWindows: this image stream was not uploaded (device_keys excluded it)')
      if frames_dev.device != out.device:
        raise RuntimeError('DeviceWindows: episode frames live on %s but the batch buffer on %s (each rank must '
                           'upload to its own GPU)' % (frames_dev.device, out.device))
      st = torch.as_tensor(starts).to(out.device, non_blocking=True)
      ops.gather_windows_into(out[off:off + n], frames_dev, st, n, self.K, fe, divisor)
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


class FeedArena:
  """Every per-batch host array of a model's feed (states, labels, window address tables) in ONE device block, written
  through ONE pinned staging block and ONE H2D copy per step: half a dozen small copies queued between two graph replays
  cost the host ~70 us per step, one ~30.  ``reserve`` while building, then ``seal``; per batch ``begin`` / ``write``... /
  ``flush``.  A ring of staging blocks lets the host run ahead: a block is rewritten only after its upload has finished.
  (Measured and not kept: uploading on a side stream into device-side landing blocks and moving them into place with a
  device-to-device copy - no gain; one small command between two replays of the step costs 6-10 us whatever it is,
  scripts/dev/between_graphs.py.)"""

  SLOTS = 4
  ALIGN = 256

  def __init__(self, device):
    self.device = resolve_device(device)      # indexed (a bare 'cuda' never compares equal to a tensor's cuda:N)
    self._layout = {}       # key -> (offset, nbytes, np dtype, shape)
    self._size = 0
    self.block = None
    self._open = False
    self._turn = 0

  def reserve(self, key, shape, dtype):
    if self.block is not None:
      raise RuntimeError('FeedArena.reserve after seal')
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    self._layout[key] = (self._size, nbytes, dt, tuple(shape))
    self._size += -(-max(nbytes, 1) // self.ALIGN) * self.ALIGN

  def seal(self):
    import torch
    self.block = torch.zeros(max(self._size, self.ALIGN), dtype=torch.uint8, device=self.device)
    self._stage = [torch.zeros(self.block.numel(), dtype=torch.uint8, pin_memory=True) for _ in range(self.SLOTS if self._layout else 0)]
    self._events = [None] * self.SLOTS                       # upload of slot i finished (host may rewrite its staging block)
    self._host = [{k: st.numpy()[off:off + nb].view(dt).reshape(shape) for k, (off, nb, dt, shape) in self._layout.items()}
                  for st in self._stage]
    return self

  def view(self, key):
    """The device tensor of one entry (a view of the block: static address, graph-safe)."""
    import torch
    off, nb, dt, shape = self._layout[key]
    tdt = torch.from_numpy(np.empty(0, dt)).dtype
    return self.block[off:off + nb].view(tdt).view(shape)

  def has(self, key):
    return key in self._layout

  @property
  def is_open(self):
    return self._open

  def begin(self):
    i = self._turn % self.SLOTS
    if self._layout and self._events[i] is not None:
      self._events[i].synchronize()
    self._open = True

  def write(self, key, values):
    if not self._open:
      raise RuntimeError('FeedArena.write outside begin() / flush()')
    dst = self._host[self._turn % self.SLOTS][key]
    values = np.asarray(values)
    if values.shape != dst.shape:
      raise ValueError("feed '%s': expected shape %s, got %s" % (key[-1], dst.shape, values.shape))
    np.copyto(dst, values, casting='same_kind')

  def flush(self):
    import torch
    self._open = False
    if not self._layout:          # nothing is fed through the arena (e.g. every input is a device tensor): no copy to queue
      return
    i = self._turn % self.SLOTS
    self.block.copy_(self._stage[i], non_blocking=True)
    if self._events[i] is None:
      self._events[i] = torch.cuda.Event()
    self._events[i].record()
    self._turn += 1
    self._open = False


class WindowFeed:
  """The static feed slot of one DeviceWindows feature (what the Estimator hands the model_fn in place of a dense tensor).

  The model picks ONE of two forms before its graph is captured:
    * ``pointers()``: an int64 device table of per-sample window addresses (an entry of the step's FeedArena); the model's
      input kernel reads the resident uint8 frames itself (ops.goal_dynimgs_u8_into), the fp32 windows are never written.
      Only offered when ``u8`` (every segment of the first batch is uint8 frames with divisor 255).
    * ``dense()``: a float32 [n, K, *frame_shape] buffer filled by geeco_gather_windows per batch.
  ``feed(windows)`` then repoints / refills per batch; both are stream-ordered in front of the replay."""

  def __init__(self, windows, arena, key):
    self.n, self.K, self.frame_shape, self.squeeze_k = windows.n, windows.K, windows.frame_shape, windows.squeeze_k
    self.arena, self.key, self.device = arena, key, arena.device
    self.u8 = windows.is_u8()
    self.shape = tuple(windows.shape)
    self.table = self.buffer = None
    self._want_table = False
    # The batches whose frames a QUEUED replay may still read through the address table: the host runs up to
    # FeedArena.SLOTS feeds ahead of the device, so that many (+ the one being written) stay referenced here.  The uploads
    # of the prefetch thread and the replays share the default stream today (the caching allocator then orders any reuse
    # behind the replays anyway); this bound does not rely on that.
    self._live = collections.deque(maxlen=FeedArena.SLOTS + 1)
    if self.u8:
      arena.reserve(key, (self.n,), np.int64)

  def pointers(self):
    if not self.u8:
      raise RuntimeError('WindowFeed.pointers(): the windows are not uint8 frames')
    self._want_table = True
    if self.arena.block is not None:
      self.table = self.arena.view(self.key)
    return self

  def dense(self):
    import torch
    if self.buffer is None:
      self.buffer = torch.empty(self.shape, dtype=torch.float32, device=self.device)
    return self.buffer

  def feed(self, windows):
    if (windows.n, windows.K, windows.frame_shape) != (self.n, self.K, self.frame_shape):
      raise ValueError('WindowFeed: batch of %s windows does not fit the slot %s' % (tuple(windows.shape), self.shape))
    if self.buffer is not None:
      windows.materialize_into(self.buffer.view((self.n, self.K) + self.frame_shape))
    if self._want_table:
      if not windows.is_u8():
        raise RuntimeError('WindowFeed: float32 frames in a slot whose model reads uint8 frames (the Estimator keys its '
                           'models by the frame type)')
      self.arena.write(self.key, windows.addresses(self.device))
      self._live.append(windows)


def _concat_feature(a, b):
  if isinstance(a, DeviceWindows):
    return DeviceWindows.concat(a, b)
  if isinstance(a, _Omitted):
    return _Omitted((a.shape[0] + b.shape[0],) + a.shape[1:], a.key)
  return np.concatenate([a, b], axis=0)


def resolve_device(device):
  """An explicit (type, index) device.  A bare 'cuda' means the CALLING thread's current device: resolve it
  before handing work to another thread (the current HIP device is thread-local and starts at 0 there)."""
  import torch
  d = torch.device(device)
  if d.type == 'cuda' and d.index is None:
    d = torch.device('cuda', torch.cuda.current_device())
  return d


# ------------------------------------------------------------------------------------------------
# staging memory and the HBM-resident episode cache of the device path
# ------------------------------------------------------------------------------------------------
class _PinnedPool:
  """Page-locked staging arrays for the reader threads (the native reader writes the uint8 frames straight into
  them; the upload is then one DMA).  Blocks are recycled: pinning 20 MB costs more than reading it."""

  def __init__(self):
    self._free = collections.defaultdict(list)
    self._lock = threading.Lock()

  def take(self, nbytes):
    import torch
    from .runtime import CAPTURE_LOCK
    with self._lock:
      if self._free[nbytes]:
        return self._free[nbytes].pop()
    with CAPTURE_LOCK:       # hipHostMalloc must not fall into the training thread's capture window
      return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)

  def give(self, t):
    with self._lock:
      if len(self._free[t.numel()]) < 32:
        self._free[t.numel()].append(t)


_PINNED = _PinnedPool()


class EpisodeCache:
  """Episodes that stay in HBM across epochs: uint8 RGB frames (19.7 MB per 100-frame 256 x 256 episode; float32 depth
  26 MB more when the model reads it) + the few KB of per-frame states on the host.  Keyed by (file identity, device,
  streams held); filled first-come until ``budget_bytes`` of device memory are in use — no eviction: an epoch scans
  the dataset cyclically, where evicting the least recently used entry would always evict the next one needed.
  Epochs >= 2 then touch neither the disk nor PCIe for cached episodes (MI355X: 288 GB holds ~10 k RGB episodes)."""

  def __init__(self, budget_bytes=None):
    self.budget_bytes = budget_bytes      # None: 60 % of the device's memory, decided on first use
    self._entries = {}
    self._bytes = 0
    self._lock = threading.Lock()
    self.hits = self.misses = 0

  @staticmethod
  def key(path, device, fetch_target, image_keys):
    st = os.stat(path)
    return (os.path.realpath(path), st.st_size, st.st_mtime_ns, str(device), bool(fetch_target), tuple(sorted(image_keys)))

  def get(self, key):
    with self._lock:
      e = self._entries.get(key)
      if e is None:
        self.misses += 1
      else:
        self.hits += 1
      return e

  def put(self, key, ex, dev, device):
    import torch
    nbytes = sum(v.numel() * v.element_size() for v in dev.values() if hasattr(v, 'numel'))
    with self._lock:
      if self.budget_bytes is None:
        self.budget_bytes = int(0.6 * torch.cuda.get_device_properties(device).total_memory)
      if key in self._entries or self._bytes + nbytes > self.budget_bytes:
        return False
      self._entries[key] = (ex, dev)
      self._bytes += nbytes
      return True

  def clear(self):
    with self._lock:
      self._entries.clear()
      self._bytes = 0
      self.hits = self.misses = 0

  @property
  def bytes_in_use(self):
    return self._bytes

  def __len__(self):
    return len(self._entries)


EPISODE_CACHE = EpisodeCache()


def episode_to_device(ex, device, image_keys=_IMAGE_KEYS):
  """Uploads the image streams of one episode (``load_episode(raw_rgb=True)``): RGB as uint8 when the recorded
  values were integral, depth as float32.  Returns (states, dev): ``states`` = ``ex`` without the image arrays (what
  the cache keeps on the host), ``dev`` = device tensors + the divisors the window gather applies."""
  import torch
  from .runtime import CAPTURE_LOCK
  device = resolve_device(device)
  T = ex['step'].shape[0]
  dev, host = {}, {}

  def stage(arr, rows):
    if arr.dtype == np.uint8:
      return torch.from_numpy(arr.reshape(rows, -1)), 255.0
    return torch.from_numpy(np.ascontiguousarray(arr.reshape(rows, -1) / np.float32(255.0))), 1.0

  if 'rgb' in image_keys:
    host['rgb'], dev['rgb_div'] = stage(ex['rgb'], T)
    if 'target_rgb' in ex:
      host['target_rgb'], dev['target_rgb_div'] = stage(ex['target_rgb'], 1)
  if 'depth' in image_keys:
    host['depth'] = torch.from_numpy(np.ascontiguousarray(ex['depth'].reshape(T, -1)))
    if 'target_depth' in ex:
      host['target_depth'] = torch.from_numpy(np.ascontiguousarray(ex['target_depth'].reshape(1, -1)))
  # allocations / synchronous copies from this (prefetch) thread must not fall into a hipGraph capture window of
  # the training thread (runtime.CAPTURE_LOCK)
  with CAPTURE_LOCK:
    for k, v in host.items():
      dev[k] = v.to(device)
  states = {k: v for k, v in ex.items() if k not in ('rgb', 'depth', 'target_rgb', 'target_depth')}
  states['_hw'] = ex['rgb'].shape[1:3]
  return states, dev


def episode_windows(ex, window_size, starts, dev=None):
  """(features, labels) for the windows beginning at ``starts`` (_window_v3 :615-631, _prepare_v4 :373-399).
  With ``dev`` (episode_to_device) the image features are DeviceWindows instead of host arrays and ``ex`` needs only
  the per-frame states."""
  K = window_size
  starts = np.asarray(starts)
  idx = starts[:, None] + np.arange(K)[None, :]
  last = starts + K - 1
  if dev is not None:
    feats = {k: ex[k][idx] for k in _FEATURE_KEYS if k not in _IMAGE_KEYS}
    H, W = ex['_hw'] if '_hw' in ex else ex['rgb'].shape[1:3]
    for key, shp in (('rgb', (H, W, 3)), ('depth', (H, W, 1))):
      dw = DeviceWindows(K, shp, dev.get(key + '_div', 1.0))
      dw.add(dev.get(key), starts)
      feats[key] = dw
    if 'target_rgb' in dev or 'target_depth' in dev:
      for key, shp in (('target_rgb', (H, W, 3)), ('target_depth', (H, W, 1))):
        dw = DeviceWindows(1, shp, dev.get(key + '_div', 1.0), squeeze_k=True)
        dw.add(dev.get(key), np.zeros(len(starts), np.int32))
        feats[key] = dw
    return feats, {k: ex[k][last] for k in _LABEL_KEYS}
  feats = {k: ex[k][idx] for k in _FEATURE_KEYS}
  if 'target_rgb' in ex:
    n = len(starts)
    for k in ('target_rgb', 'target_depth'):
      t = ex[k]
      feats[k] = _Omitted((n,) + t.shape, t.key) if isinstance(t, _Omitted) else np.broadcast_to(t, (n,) + t.shape).copy()
  labels = {k: ex[k][last] for k in _LABEL_KEYS}
  return feats, labels


class _Prefetcher:
  """Runs an iterator factory in a background thread (tf.data's prefetch); the episode readers it draws from are a
  thread pool of their own (``_EpisodeSource``)."""

  def __init__(self, make_iter, depth, device=None):
    self._q = queue.Queue(maxsize=max(int(depth), 1))
    self._device = device
    self._stop = threading.Event()
    self._t = threading.Thread(target=self._run, args=(make_iter,), daemon=True)
    self._t.start()

  def _put(self, item):
    while not self._stop.is_set():
      try:
        self._q.put(item, timeout=0.2)
        return True
      except queue.Full:
        continue
    return False

  def _run(self, make_iter):
    try:
      if self._device is not None and self._device.type == 'cuda':
        import torch
        torch.cuda.set_device(self._device)     # thread-local: a new thread starts on device 0 whatever LOCAL_RANK is
      for item in make_iter():
        if not self._put(('item', item)):
          return
      self._put(('end', None))
    except BaseException as e:   # surfaced in the consumer
      self._put(('error', e))

  def close(self):
    """Stops the producer (a consumer that leaves the epoch early, e.g. Estimator.train(steps=...))."""
    self._stop.set()

  def __del__(self):
    self._stop.set()

  def __iter__(self):
    while True:
      kind, val = self._q.get()
      if kind == 'item':
        yield val
      elif kind == 'end':
        return
      else:
        raise val


class _EpisodeSource:
  """Episodes of ``paths`` in order, read ``num_threads`` at a time (tf.data's num_parallel_reads /
  num_parallel_calls, geeco_gym.py:442-473: parallel and order-preserving): a window of reads runs ahead of the
  consumer in a thread pool; the native reader holds no Python lock, so the threads really overlap.  On the device
  path a cached episode (EPISODE_CACHE) is never read again."""

  def __init__(self, paths, meta, fetch_target, num_threads, device, image_keys, cache):
    from concurrent.futures import ThreadPoolExecutor
    self.paths, self.meta, self.fetch_target = list(paths), meta, fetch_target
    self.device, self.image_keys, self.cache = device, tuple(image_keys), cache
    self.num_threads = max(int(num_threads or 1), 1)
    self._pool = ThreadPoolExecutor(max_workers=self.num_threads, thread_name_prefix='geeco-reader')
    self._reads = 0

  def _read(self, path):
    staged = []
    def alloc(nbytes):
      t = _PINNED.take(nbytes)
      staged.append(t)
      return t.numpy()
    pinned = self.device is not None and self.device.type == 'cuda'
    ex = load_episode(path, self.meta, self.fetch_target, raw_rgb=self.device is not None, image_keys=self.image_keys,
                      alloc=alloc if pinned else None)
    return ex, staged

  def __iter__(self):
    """Yields (states / ex, dev or None) per episode, in the order of ``paths``."""
    ahead = self.num_threads + 1
    pending = collections.deque()      # (path, key, cached entry or future)
    it = iter(self.paths)

    def submit():
      for path in it:
        key = entry = None
        if self.device is not None and self.cache is not None:
          key = self.cache.key(path, self.device, self.fetch_target, self.image_keys)
          entry = self.cache.get(key)
        if entry is None:
          if not self._reads:     # the reader keeps one mapped inflate buffer (~105 MB) per thread + 1 between episodes
            _buffer_limit(self, self.num_threads + 1)
          self._reads += 1
        pending.append((path, key, entry if entry is not None else self._pool.submit(self._read, path)))
        return

    try:
      for _ in range(ahead):
        submit()
      while pending:
        path, key, item = pending.popleft()
        submit()
        if isinstance(item, tuple):
          yield item
          continue
        ex, staged = item.result()
        if self.device is None:
          yield ex, None
          continue
        states, dev = episode_to_device(ex, self.device, self.image_keys)
        for t in staged:          # the copies above were synchronous: the staging blocks are free again
          _PINNED.give(t)
        if self.cache is not None:
          self.cache.put(key, states, dev, self.device)
        yield states, dev
    finally:
      for _, _, item in pending:
        if not isinstance(item, tuple) and not item.cancel():
          item.add_done_callback(_return_staging)      # a read already running: nobody will take its result
      # reads already running finish (they cannot be interrupted inside the native reader).  An epoch that read NOTHING (every
      # episode came out of the HBM cache) hands the reader's spare inflate buffers back to the OS: no reader will run again, and
      # under data parallelism every rank would otherwise hold its own pool (up to num_threads + 1 buffers of ~105 MB) for the
      # rest of training.  An epoch that did read keeps them for the next one (sixteen threads faulting fresh 105 MB mappings in
      # at once cost epoch 1 of the bench 0.13 s when the pool was emptied after every epoch).
      # (not waited for: an epoch left early -- Estimator.train(steps=...), an exception in the step -- returns at once)
      self._pool.shutdown(wait=False)
      _buffer_limit(self, None)
      if not self._reads:
        # off this thread: unmapping up to num_threads + 1 touched 105 MB buffers takes ~0.15 s (measured: the first fully cached
        # epoch of the bench ran 0.68 s instead of 0.53 s with the release inline); nothing waits for it
        threading.Thread(target=tfrecord._host().geeco_host_release_buffers, name='geeco-release', daemon=True).start()


def _return_staging(fut):
  """Done-callback of a read whose result nobody takes (the epoch was left early): its pinned staging blocks go back."""
  if fut.cancelled() or fut.exception() is not None:
    return
  for t in fut.result()[1]:
    _PINNED.give(t)


_LIMITS, _LIMITS_LOCK = {}, threading.Lock()


def _buffer_limit(source, n):
  """The native reader's spare-buffer limit is ONE number per process (geeco_host_set_buffer_limit): with several sources alive
  (a train and an eval pipeline with different thread counts) it is the MAX of what they asked for; ``n`` None = the source is done
  (the limit stays where it is while no source reads, so the next epoch finds the buffers of this one)."""
  with _LIMITS_LOCK:
    if n is None:
      _LIMITS.pop(id(source), None)
    else:
      _LIMITS[id(source)] = int(n)
    if _LIMITS:
      tfrecord._host().geeco_host_set_buffer_limit(max(_LIMITS.values()))


def usable_host_cores():
  """Host threads this process may use: the affinity mask capped by the cgroup CPU quota."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return max(1, n)


def default_reader_threads(world=1):
  """``num_threads=None``: this rank's share of the host, usable cores // ranks ON THIS NODE (LOCAL_WORLD_SIZE as torchrun exports
  it; ``world`` -- the global size -- only where that is absent, i.e. one node), within 4 (the reference's default,
  train_e2evmc.py:67) ... 32.  Reading an episode is 97 % inflate and scales with threads up to the cores a rank owns
  (profiles/r05/reader_scaling.json): one rank's GPU consumes ~119 episodes/s, one reader thread delivers ~9, so epoch 1 (before
  the HBM episode cache serves everything) is reader-bound below ~13 cores per rank."""
  try:
    local = int(os.environ.get('LOCAL_WORLD_SIZE', '') or world)
  except ValueError:
    local = world
  return max(4, min(32, usable_host_cores() // max(min(int(local), max(int(world), 1)), 1)))


def pickplace_input_fn(dataset_dir, split_name, mode, encoding='v4', window_size=4, fetch_target=False,
                       shuffle_buffer=128, batch_size=1, num_epochs=1, num_threads=4, prefetch_size=4, seed=None,
                       shard=None, device=None, device_keys=None, cache=True):
  """Same signature as the reference's pickplace_input_fn (geeco_gym.py:234-279).  Returns an iterable of
  (features, labels) numpy batches.  ``num_threads`` episodes are read in parallel, in order (num_parallel_reads /
  num_parallel_calls of :442-473; None = ``default_reader_threads``: this rank's share of the host cores); ``prefetch_size`` batches are prepared ahead of the consumer (:473).
  Extensions: ``shard = (rank, world)`` makes each data-parallel rank read a disjoint, rank-strided subset of the
  episodes.  ``device`` (e.g. 'cuda'): upload every episode's frames once and hand out image features as DeviceWindows
  (windows are gathered in HBM); ``device_keys``: which image streams the model reads (default both; ('rgb',) for an
  RGB-only model skips decoding, uploading and caching 26 MB of depth per episode — the 'depth' features then refuse
  to materialise); ``cache``: keep uploaded episodes in HBM across epochs (EPISODE_CACHE; True, False or an
  EpisodeCache)."""
  if encoding != 'v4':
    # v1-v3 are dead code in the reference (undefined PickAndPlaceEncodingV1/2/3 -> NameError)
    raise KeyError(encoding)
  if num_threads is None:
    num_threads = default_reader_threads(shard[1] if shard is not None else 1)
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
  image_keys = _IMAGE_KEYS if device_keys is None else tuple(device_keys)
  if not set(image_keys) <= set(_IMAGE_KEYS):
    raise ValueError('device_keys must be a subset of %s' % (_IMAGE_KEYS,))
  ep_cache = None
  if device is not None and device.type == 'cuda' and cache:
    ep_cache = cache if isinstance(cache, EpisodeCache) else EPISODE_CACHE
  print('[pickplace_input_fn_v4] #tfrecords: %d' % len(paths))

  def batches():
    carry_f, carry_l = None, None   # windows left over from the previous episode (batch() spans episodes)
    source = _EpisodeSource(paths * num_epochs, meta, fetch_target, num_threads, device, image_keys, ep_cache)
    for ex, dev in source:
      T = ex['step'].shape[0]
      if shard is not None and T != meta.episode_length - 1:
        raise ValueError('an episode holds %d frames, meta_info.json says %d: the data-parallel batch schedule assumes '
                         'fixed-length episodes' % (T + 1, meta.episode_length))
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


def synthetic_scene_frames(T, H, W, seed):
  """uint8 RGB frames [T, H, W, 3] + float32 depth [T, H, W, 1] of a toy table-top scene (shaded background, a textured
  table, a few boxes sliding between frames): flat and smooth regions with a little sensor-like noise, which is what
  makes a rendered frame compress — uniform noise (test fixtures) would not.  Generator of on-disk datasets for the
  input-pipeline benchmark; not part of training."""
  r = np.random.default_rng(seed)
  yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
  base = np.stack([90 + 60 * yy / H, 110 + 40 * xx / W, 140 - 50 * yy / H], axis=-1)            # wall gradient
  table = yy > 0.55 * H
  tex = r.integers(-6, 7, size=[H, W, 1]).astype(np.float32) * table[..., None]
  base = np.where(table[..., None], np.float32([150, 120, 90]) + tex, base)
  depth0 = (2.5 - 1.5 * yy / H + 0.02 * np.sin(xx / 9.0)).astype(np.float32)
  nbox = 4
  pos0, vel = r.random([nbox, 2]) * [0.4 * H, 0.8 * W] + [0.5 * H, 0.0], r.standard_normal([nbox, 2]) * 0.6
  size = r.integers(H // 16, H // 6, size=[nbox, 2])
  col = r.integers(20, 236, size=[nbox, 3]).astype(np.float32)
  rgb = np.empty([T, H, W, 3], np.uint8)
  depth = np.empty([T, H, W, 1], np.float32)
  for t in range(T):
    img, dep = base.copy(), depth0.copy()
    for b in range(nbox):
      y0, x0 = (pos0[b] + t * vel[b]).astype(int) % [H, W]
      y1, x1 = min(H, y0 + size[b, 0]), min(W, x0 + size[b, 1])
      shade = np.linspace(1.0, 0.8, max(x1 - x0, 1), dtype=np.float32)[None, :, None]
      img[y0:y1, x0:x1] = col[b] * shade
      dep[y0:y1, x0:x1] = 0.8 + 0.1 * b
    noise = r.integers(-1, 2, size=[H, W, 3]) * (r.random([H, W, 1]) < 0.15)                   # sparse +-1 sensor noise
    rgb[t] = np.clip(np.rint(img + noise), 0, 255).astype(np.uint8)
    depth[t, :, :, 0] = dep + (1e-3 * r.standard_normal([H, W])).astype(np.float32)
  return rgb, depth


def write_synthetic_dataset(root, num_episodes, episode_length=100, img_hw=(256, 256), seed=0, eval_episodes=None):
  """A dataset directory in the reference's layout (geeco_gym.py:249-264: meta/meta_info.json, data/*.tfrecord.zlib,
  splits/default/{train,eval}.txt) filled with ``synthetic_scene_frames`` episodes.  ``eval_episodes``: how many of
  the episodes the eval split lists (default: all).  Returns the meta tuple."""
  H, W = img_hw
  joints = ['robot0:%s' % j for j in _ARM_JOINTS + _FINGER_JOINTS]
  meta = PickAndPlaceMetaV4(episode_length=episode_length, img_height=H, img_width=W, monitored_joints=joints,
                            actuated_joints=joints[:2], monitored_mocaps=['robot0:mocap'],
                            monitored_objects=['object0:joint'], dim_cmd=4, dim_ctrl=2)
  for sub in ('meta', 'data', os.path.join('splits', 'default')):
    os.makedirs(os.path.join(root, sub), exist_ok=True)
  with open(os.path.join(root, 'meta', 'meta_info.json'), 'w') as fp:
    json.dump(meta._asdict(), fp)
  names = []
  for e in range(num_episodes):
    r = np.random.default_rng([seed, e])
    T = episode_length
    rgb, depth = synthetic_scene_frames(T, H, W, seed=[seed, e, 1])
    cmd = np.concatenate([0.3 * r.standard_normal([T, 3]), r.integers(-1, 2, [T, 1])], 1).astype(np.float32)
    name = 'ep%05d.tfrecord.zlib' % e
    write_episode(os.path.join(root, 'data', name), meta, rgb, depth, cmd, r.standard_normal([T, 2]).astype(np.float32),
                  r.standard_normal([T, 9]).astype(np.float32), r.standard_normal([T, 9]).astype(np.float32),
                  (1.5 * r.random([T, 7])).astype(np.float32), (1.5 * r.random([T, 7])).astype(np.float32),
                  (1.5 * r.random([T, 7])).astype(np.float32))
    names.append(name)
  n_eval = num_episodes if eval_episodes is None else eval_episodes
  for mode, sel in (('train', names), ('eval', names[:n_eval])):
    with open(os.path.join(root, 'splits', 'default', mode + '.txt'), 'w') as fp:
      fp.write('\n'.join(sel) + '\n')
  return meta
