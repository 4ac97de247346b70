"""The native episode reader (libgeeco_host.so: inflate, TFRecord framing + CRC-32C, SequenceExample scan, float -> array
copies; include/geeco_host.h) against the pure-Python reader of geeco_amd/tfrecord.py, and the parallel, order-preserving
episode source of the input pipeline (reference: src/data/geeco_gym.py:291-315, 436-473)."""
import gzip
import os
import struct
import sys
import threading
import time
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_host_logic_cpu import _crc32c_bitwise, _make_dataset    # noqa: E402


def _paths(root):
  return sorted(os.path.join(root, 'data', f) for f in os.listdir(os.path.join(root, 'data')))


def test_native_reader_equals_python_reader(tmp_path):
  from geeco_amd import input_fn as I
  meta, eps = _make_dataset(str(tmp_path), n_eps=2, T=7, H=12, W=10)
  for path, d in zip(_paths(str(tmp_path)), eps):
    for fetch_target in (False, True):
      for raw in (False, True):
        a = I.load_episode(path, meta, fetch_target, raw_rgb=raw)
        b = I.load_episode_py(path, meta, fetch_target, raw_rgb=raw)
        assert sorted(a) == sorted(b)
        for k in b:
          if raw and k in ('rgb', 'target_rgb'):
            assert a[k].dtype == np.uint8          # integral recorded values: kept as bytes
          else:
            assert a[k].dtype == b[k].dtype, k
          np.testing.assert_array_equal(np.asarray(a[k], b[k].dtype), b[k], err_msg=k)
    np.testing.assert_array_equal(I.load_episode(path, meta, True, raw_rgb=True)['target_rgb'], d['rgb'][-1])


def test_native_reader_list_index_and_kinds(tmp_path):
  from geeco_amd import tfrecord as T
  meta, eps = _make_dataset(str(tmp_path), n_eps=1, T=5, H=8, W=8)
  path = _paths(str(tmp_path))[0]
  _, fl = T.parse_sequence_example(next(iter(T.read_records(path))))
  with T.EpisodeReader(path) as rd:
    assert rd.num_records == 1 and rd.inflated_bytes == len(zlib.decompress(open(path, 'rb').read()))
    assert rd.names() == list(fl.keys())                       # same lists, file order
    for name, frames in fl.items():
      assert rd.frames(name) == len(frames) == 5
      kind, vals = rd.kind(name)
      assert vals == len(frames[0]) and kind == (3 if frames[0].dtype == np.int64 else 2), name
      got = rd.i64(name, vals) if kind == 3 else rd.f32(name, vals)
      np.testing.assert_array_equal(got, np.stack(frames), err_msg=name)
    with pytest.raises(KeyError):
      rd.frames('no_such_list')
    with pytest.raises(IOError, match='value count'):
      rd.f32('cmd', 3)
    with pytest.raises(IOError, match='not a float list'):
      rd.f32('step', 1)
    with pytest.raises(IOError, match='not an int64 list'):
      rd.i64('cmd', 4)
    with pytest.raises(ValueError):
      rd.f32('cmd', 4, out=np.empty((5, 4), np.float64))


def test_native_reader_non_integral_rgb_falls_back_to_float(tmp_path):
  """uint8 upload only when EVERY recorded value is an integer in [0, 255] (tfrecord.py:73-74 writes uint8 images as
  floats; nothing stops a dataset from holding real floats): 0.5, -1, 256, NaN each break exactness."""
  from geeco_amd import input_fn as I
  from geeco_amd import tfrecord as T
  meta, eps = _make_dataset(str(tmp_path), n_eps=1, T=4, H=8, W=8)
  d = eps[0]
  for bad in (0.5, -1.0, 256.0, float('nan'), 255.00002, float('inf'), 1e10, -1e10, 65536.0, 4294967296.0):
    rgb = d['rgb'].astype(np.float32)
    rgb[2, 3, 4, 1] = bad
    p = str(tmp_path / 'bad.tfrecord.zlib')
    frames = [{'step': np.array([t], np.int64), 'rgb': rgb[t]} for t in range(4)]
    T.write_records(p, [T.encode_sequence_example({}, frames)])
    with T.EpisodeReader(p) as rd:
      _, exact = rd.u8('rgb', 8 * 8 * 3)
      assert not exact, bad
      np.testing.assert_array_equal(rd.f32('rgb', 8 * 8 * 3).reshape(rgb.shape), rgb)
  # -0.0 and 255 are exact
  rgb = d['rgb'].astype(np.float32)
  rgb[0, 0, 0, 0], rgb[0, 0, 0, 1] = -0.0, 255.0
  T.write_records(p, [T.encode_sequence_example({}, [{'step': np.array([0], np.int64), 'rgb': rgb[0]}])])
  with T.EpisodeReader(p) as rd:
    got, exact = rd.u8('rgb', 8 * 8 * 3)
    assert exact and got[0, 0] == 0 and got[0, 1] == 255
  # and load_episode(raw_rgb=True) hands float32 0..255 values to the device path, which divides on the host then
  rgb[0, 0, 0, 0] = 0.25
  joints = {('joint_qpos-%s' % j): np.zeros(1, np.float32) for j in meta.monitored_joints}
  joints.update({('joint_qvel-%s' % j): np.zeros(1, np.float32) for j in meta.monitored_joints})
  frames = [dict(step=np.array([t], np.int64), ts=np.zeros(1, np.float32), rgb=rgb[t], depth=d['depth'][t], cmd=d['cmd'][t],
                 ctrl=d['ctrl'][t], goal_qpos=d['goal'][t], obj_qpos=d['obj'][t], **joints,
                 **{'mocap_qpos-robot0:mocap': d['mocap'][t]}) for t in range(4)]
  T.write_records(p, [T.encode_sequence_example({}, frames)])
  ex = I.load_episode(p, meta, False, raw_rgb=True)
  assert ex['rgb'].dtype == np.float32 and ex['rgb'][0, 0, 0, 0] == 0.25
  np.testing.assert_array_equal(I.load_episode(p, meta, False)['rgb'], rgb[:-1] / np.float32(255.0))


def test_native_reader_refuses_damaged_files(tmp_path):
  from geeco_amd import tfrecord as T
  meta, _ = _make_dataset(str(tmp_path), n_eps=1, T=4, H=8, W=8)
  path = _paths(str(tmp_path))[0]
  raw = bytearray(zlib.decompress(open(path, 'rb').read()))
  bad = str(tmp_path / 'bad.tfrecord.zlib')

  def opens(data, compressed=True, **kw):
    open(bad, 'wb').write(zlib.compress(bytes(data)) if compressed else bytes(data))
    return T.EpisodeReader(bad, **kw)

  flipped = bytearray(raw); flipped[len(raw) // 2] ^= 0x10
  with pytest.raises(IOError, match='corrupted record payload'):
    opens(flipped)
  assert opens(flipped, verify=False).num_records == 1          # as read_records(verify=False)
  hdr = bytearray(raw); hdr[9] ^= 0x01
  with pytest.raises(IOError, match='corrupted record length'):
    opens(hdr)
  with pytest.raises(IOError, match='truncated record'):
    opens(raw[:-7])
  with pytest.raises(IOError, match='truncated record header'):
    opens(raw + b'\x01\x02\x03')
  with pytest.raises(IOError, match='no record'):
    opens(b'')
  open(bad, 'wb').write(open(path, 'rb').read()[:-20])            # cut inside the zlib stream
  with pytest.raises(IOError, match='inflate'):
    T.EpisodeReader(bad)
  open(bad, 'wb').write(b'not a zlib stream at all' * 10)
  with pytest.raises(IOError, match='inflate'):
    T.EpisodeReader(bad)
  with pytest.raises(IOError):
    T.EpisodeReader(str(tmp_path / 'missing.tfrecord.zlib'))
  # a structurally broken SequenceExample inside a well-formed record
  body = b'\x12\x05\x0a\x03\x0a\xff\xff'                          # feature_lists { entry { len runs past the end
  rec = struct.pack('<Q', len(body)); rec += struct.pack('<I', T.masked_crc32c(rec)) + body + struct.pack('<I', T.masked_crc32c(body))
  with pytest.raises(IOError, match='malformed SequenceExample'):
    opens(rec)
  # two records: both framings are checked, the first is parsed (data_recorder.py:134-156 writes one per file)
  two = bytes(raw) + bytes(raw)
  assert opens(two).num_records == 2
  # uncompressed and gzip files
  assert opens(raw, compressed=False, compression=None).num_records == 1
  open(bad, 'wb').write(gzip.compress(bytes(raw)))
  assert T.EpisodeReader(bad, compression='gzip').num_records == 1
  with pytest.raises(ValueError):
    T.EpisodeReader(bad, compression='lz4')


def test_inflate_matches_zlib_and_crc32c_matches_bitwise():
  from geeco_amd import tfrecord as T
  lib = T._host()
  r = np.random.default_rng(5)
  cases = [b'', b'a', bytes(100000), r.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
           np.repeat(r.integers(0, 7, 4000, dtype=np.uint8), r.integers(1, 300, 4000)).tobytes(),
           r.integers(0, 256, 3000, dtype=np.uint8).astype(np.float32).tobytes() * 9]
  for data in cases:
    for level in (0, 1, 6, 9):
      for fmt, comp in ((1, zlib.compress(data, level)), (2, gzip.compress(data, compresslevel=level))):
        dst = np.empty(len(data) + 8, np.uint8)
        n = lib.geeco_inflate(comp, len(comp), dst.ctypes.data, len(data), fmt)        # exactly fitting destination
        assert n == len(data) and dst[:n].tobytes() == data
        if len(data) > 1:
          assert lib.geeco_inflate(comp, len(comp), dst.ctypes.data, len(data) - 1, fmt) == -2      # too small
        assert lib.geeco_inflate(comp[:-3], len(comp) - 3, dst.ctypes.data, len(data) + 8, fmt) == -1   # truncated
    if len(data) > 100:
      comp = bytearray(zlib.compress(data, 6)); comp[-2] ^= 0x40                       # Adler-32 mismatch
      assert lib.geeco_inflate(bytes(comp), len(comp), dst.ctypes.data, len(data) + 8, 1) == -1
  for n in (0, 1, 7, 8, 9, 31, 32, 33, 1000, 4099):
    for off in (0, 1, 3):
      data = r.integers(0, 256, n + off, dtype=np.uint8).tobytes()[off:]
      assert lib.geeco_crc32c(data, len(data), 0) == _crc32c_bitwise(data), (n, off)
  a, b = b'hello, ', b'world'
  assert lib.geeco_crc32c(b, len(b), lib.geeco_crc32c(a, len(a), 0)) == _crc32c_bitwise(a + b)      # running value


def test_table_driven_inflate_equals_zlib(tmp_path):
  """csrc/host_inflate.cpp (the decoder geeco_episode_open tries first): every stream it accepts is bit-identical to zlib's
  output - stored / fixed / dynamic blocks, overlapping copies at every short distance, long codes with subtables, the
  recorder's uint8-as-float lists, 24 MB outputs that outgrow the first buffer; damaged streams are declined (and zlib, which
  the reader then asks, reports them); an episode decodes to the same arrays with the decoder on and off."""
  from geeco_amd import tfrecord as T, input_fn as I
  lib = T._host()
  r = np.random.default_rng(11)
  big = r.integers(0, 256, 6_000_000, dtype=np.uint8).astype(np.float32).tobytes()          # 24 MB
  cases = [b'', b'a', b'ab' * 5, bytes(100000), r.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
           np.repeat(r.integers(0, 7, 4000, dtype=np.uint8), r.integers(1, 300, 4000)).tobytes(),
           b'abc' * 30000, b'abcde' * 20000, b'abcdef' * 20000, b'abcdefg' * 20000, b'abcdefghijk' * 9000,
           bytes((i * 7) & 255 if r.random() < 0.97 else int(r.integers(0, 256)) for i in range(120000)), big]
  for data in cases:
    for level in ((0, 1, 6, 9) if len(data) < 1_000_000 else (6,)):
      comp = zlib.compress(data, level)
      dst = np.empty(len(data) + 8, np.uint8)
      n = lib.geeco_inflate_fast(comp, len(comp), dst.ctypes.data, len(data))
      assert n == len(data) and dst[:n].tobytes() == data, (len(data), level, n)
      if len(data) > 1:
        assert lib.geeco_inflate_fast(comp, len(comp), dst.ctypes.data, len(data) - 1) == -2          # destination too small
      for cut in (1, 3, 5, len(comp) // 2):
        if cut < len(comp):
          assert lib.geeco_inflate_fast(comp[:-cut], len(comp) - cut, dst.ctypes.data, len(data) + 8) == -3   # truncated: declined
          assert lib.geeco_inflate(comp[:-cut], len(comp) - cut, dst.ctypes.data, len(data) + 8, 1) == -1     # ... and zlib says why
    if len(data) > 100:
      comp = bytearray(zlib.compress(data, 6)); comp[-2] ^= 0x40                                       # Adler-32 mismatch
      assert lib.geeco_inflate_fast(bytes(comp), len(comp), dst.ctypes.data, len(data) + 8) == -3
  # not zlib streams at all: gzip, raw deflate, a preset dictionary header, garbage
  co = zlib.compressobj(6, zlib.DEFLATED, -15)
  raw_deflate = co.compress(b'hello' * 100) + co.flush()
  cd = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, b'hello')
  with_dict = cd.compress(b'hello' * 100) + cd.flush()
  dst = np.empty(4096, np.uint8)
  for bad in (gzip.compress(b'hello' * 100), raw_deflate, with_dict, b'not a zlib stream at all' * 10, b'\x78'):
    assert lib.geeco_inflate_fast(bad, len(bad), dst.ctypes.data, 4096) == -3
  # the reader: same episode with the decoder on and off (and the spare-buffer pool in between)
  meta, _ = _make_dataset(str(tmp_path), n_eps=2, T=5, H=16, W=16)
  got = []
  for on in (1, 0, 1):
    lib.geeco_host_set_fast_inflate(on)
    got.append([I.load_episode(p, meta, True, raw_rgb=True) for p in _paths(str(tmp_path))])
  lib.geeco_host_set_fast_inflate(1)
  lib.geeco_host_release_buffers()
  for a, b in zip(got[0] + got[0], got[1] + got[2]):
    for k in a:
      np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=k)


def test_reader_returns_its_spare_buffers_after_a_cached_epoch(tmp_path):
  """the native reader keeps mapped inflate buffers between episodes -- one per reader thread + 1, not a fixed
  32 / 6 GiB -- and an epoch that read nothing (every episode served by the HBM cache: no reader will run again) hands them
  back to the OS; an epoch that did read keeps them for the next."""
  from geeco_amd import input_fn as I
  from geeco_amd import tfrecord as T
  lib = T._host()
  meta, _ = _make_dataset(str(tmp_path), n_eps=5, T=6, H=16, W=16)
  lib.geeco_host_release_buffers()
  assert lib.geeco_host_spare_buffers() == 0
  lib.geeco_host_set_buffer_limit(8)
  eps = [I.load_episode(p, meta, True, raw_rgb=True) for p in _paths(str(tmp_path))]        # direct reads: buffers are kept ...
  assert len(eps) == 5 and 1 <= lib.geeco_host_spare_buffers() <= 8
  lib.geeco_host_set_buffer_limit(1)                                                         # ... a lower limit trims them
  assert lib.geeco_host_spare_buffers() <= 1
  n = sum(1 for _ in I.pickplace_input_fn(str(tmp_path), 'default', 'train', window_size=3, batch_size=2, num_threads=2, seed=0))
  assert n > 0 and 1 <= lib.geeco_host_spare_buffers() <= 3                                  # an epoch that read: kept, at most threads + 1

  class AllCached:                       # stands in for input_fn.EPISODE_CACHE with every episode resident
    def key(self, path, *a):
      return path
    def get(self, key):
      return ({'step': np.zeros(5, np.int64)}, {})
  import torch
  src = I._EpisodeSource(_paths(str(tmp_path)), meta, True, 3, torch.device('cpu'), ('rgb', 'depth'), AllCached())
  assert sum(1 for _ in src) == 5 and src._reads == 0
  for _ in range(200):                                                                       # (released by a background thread)
    if lib.geeco_host_spare_buffers() == 0:
      break
    time.sleep(0.01)
  assert lib.geeco_host_spare_buffers() == 0                                                 # nothing was read: all returned


def test_inflate_under_address_sanitizer(tmp_path):
  """tests/native/fuzz_inflate.cpp: the decoder built with -fsanitize=address,undefined against 40 valid and ~3 900 damaged
  streams (bit flips, truncations, overwritten stretches, damaged block headers): no access outside the buffers, nothing
  accepted that zlib rejects or decodes differently.  (Sanitizers run on the CPU build only.)"""
  import shutil, subprocess
  if shutil.which('g++') is None:
    pytest.skip('no g++')
  here = os.path.dirname(os.path.abspath(__file__))
  csrc = os.path.join(here, '..', 'geeco_amd', 'csrc')
  exe = str(tmp_path / 'fuzz_inflate')
  subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-I' + csrc,
                  os.path.join(here, 'native', 'fuzz_inflate.cpp'), os.path.join(csrc, 'host_inflate.cpp'), '-lz', '-o', exe],
                 check=True, timeout=300)
  res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0, res.stdout + res.stderr
  assert res.stdout.startswith('ok:'), res.stdout


def test_episode_source_is_parallel_and_order_preserving(tmp_path, monkeypatch):
  """num_threads readers at once (geeco_gym.py:442-473 num_parallel_reads / num_parallel_calls), episodes delivered in
  list order whatever order the reads finish in, batches identical for every thread count."""
  from geeco_amd import input_fn as I
  meta, eps = _make_dataset(str(tmp_path), n_eps=7, T=9, H=8, W=8)
  kw = dict(window_size=3, fetch_target=True, batch_size=5)
  ref = list(I.pickplace_input_fn(str(tmp_path), 'default', 'eval', num_threads=1, **kw))
  real = I.load_episode
  state = {'now': 0, 'peak': 0}
  lock = threading.Lock()

  def slow(path, *a, **k):
    with lock:
      state['now'] += 1
      state['peak'] = max(state['peak'], state['now'])
    time.sleep(0.05 if path.endswith('ep000.tfrecord.zlib') or path.endswith('ep003.tfrecord.zlib') else 0.005)
    try:
      return real(path, *a, **k)
    finally:
      with lock:
        state['now'] -= 1

  monkeypatch.setattr(I, 'load_episode', slow)
  for nt in (1, 3, 8):
    state['peak'] = 0
    got = list(I.pickplace_input_fn(str(tmp_path), 'default', 'eval', num_threads=nt, **kw))
    assert state['peak'] == min(nt, 7) or (nt > 1 and 1 < state['peak'] <= nt), (nt, state['peak'])
    assert len(got) == len(ref)
    for (fa, la), (fb, lb) in zip(got, ref):
      for k in fb:
        np.testing.assert_array_equal(fa[k], fb[k], err_msg=k)
      for k in lb:
        np.testing.assert_array_equal(la[k], lb[k], err_msg=k)
  assert state['peak'] > 1
  # a reader error surfaces in the consumer, not in a worker thread
  monkeypatch.setattr(I, 'load_episode', lambda path, *a, **k: (_ for _ in ()).throw(IOError('boom %s' % path)))
  with pytest.raises(IOError, match='boom'):
    list(I.pickplace_input_fn(str(tmp_path), 'default', 'eval', num_threads=2, **kw))


def test_undecoded_image_stream_keeps_shapes_and_refuses_values(tmp_path):
  from geeco_amd import input_fn as I
  meta, _ = _make_dataset(str(tmp_path), n_eps=1, T=9, H=8, W=8)
  path = _paths(str(tmp_path))[0]
  ex = I.load_episode(path, meta, True, image_keys=('rgb',))
  assert ex['depth'].shape == (8, 8, 8, 1) and ex['target_depth'].shape == (8, 8, 1) and ex['rgb'].dtype == np.float32
  f, _ = I.episode_windows(ex, 3, np.arange(4))
  assert f['depth'].shape == (4, 3, 8, 8, 1) and f['target_depth'].shape == (4, 8, 8, 1) and f['rgb'].shape == (4, 3, 8, 8, 3)
  with pytest.raises(RuntimeError, match='not decoded'):
    np.asarray(f['depth'])
  with pytest.raises(ValueError):
    I.pickplace_input_fn(str(tmp_path), 'default', 'eval', device_keys=('rgb', 'flow'))


def test_synthetic_dataset_writer_round_trip(tmp_path):
  from geeco_amd import input_fn as I
  meta = I.write_synthetic_dataset(str(tmp_path), 3, episode_length=6, img_hw=(16, 24), seed=4, eval_episodes=1)
  assert meta == I.get_meta_v4(str(tmp_path)) and meta.img_width == 24
  assert len(I.collect_tfrecords(str(tmp_path), 'default', 'train')) == 3
  assert len(I.collect_tfrecords(str(tmp_path), 'default', 'eval')) == 1
  rgb, depth = I.synthetic_scene_frames(6, 16, 24, seed=[4, 1, 1])
  ex = I.load_episode(I.collect_tfrecords(str(tmp_path), 'default', 'train')[1], meta, True, raw_rgb=True)
  np.testing.assert_array_equal(ex['rgb'], rgb[:-1])
  np.testing.assert_array_equal(ex['target_depth'], depth[-1])
  batches = list(I.pickplace_input_fn(str(tmp_path), 'default', 'train', window_size=2, batch_size=4, seed=0, num_threads=2))
  assert sum(len(f['step']) for f, _ in batches) == 3 * 4


# ----------------------------------------------------------------------------------------------------
# Interoperability with an INDEPENDENT encoder / decoder of the wire format: the protobuf runtime (google.protobuf 7.x, the
# same wire format TensorFlow's tf.train.SequenceExample uses; the message types of tensorflow/core/example/{example,
# feature}.proto are declared here field by field).  Until round 5 the reader had only ever parsed files its sibling writer
# (input_fn.write_episode) or a test had assembled.  TensorFlow itself is not installable here; this is the closest third
# party that speaks the format.
# ----------------------------------------------------------------------------------------------------
def _tf_example_classes():
  from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
  T = descriptor_pb2.FieldDescriptorProto
  fd = descriptor_pb2.FileDescriptorProto(name='tf_example_subset.proto', package='tensorflow', syntax='proto3')

  def msg(name, parent=None):
    m = (parent.nested_type if parent is not None else fd.message_type).add()
    m.name = name
    return m

  def field(m, name, num, typ, label=1, type_name=None, packed=None, oneof=None):
    f = m.field.add(name=name, number=num, type=typ, label=label)
    if type_name:
      f.type_name = type_name
    if packed is not None:
      f.options.packed = packed
    if oneof is not None:
      f.oneof_index = oneof
  field(msg('BytesList'), 'value', 1, T.TYPE_BYTES, 3)
  field(msg('FloatList'), 'value', 1, T.TYPE_FLOAT, 3, packed=True)
  field(msg('Int64List'), 'value', 1, T.TYPE_INT64, 3, packed=True)
  m = msg('Feature')
  m.oneof_decl.add(name='kind')
  for n, num in (('bytes_list', 1), ('float_list', 2), ('int64_list', 3)):
    field(m, n, num, T.TYPE_MESSAGE, 1, '.tensorflow.' + n.title().replace('_', ''), oneof=0)
  for outer, entry, key, val in (('Features', 'FeatureEntry', 'feature', 'Feature'),
                                 ('FeatureLists', 'FeatureListEntry', 'feature_list', 'FeatureList')):
    if outer == 'FeatureLists':
      field(msg('FeatureList'), 'feature', 1, T.TYPE_MESSAGE, 3, '.tensorflow.Feature')
    m = msg(outer)
    e = msg(entry, m)
    e.options.map_entry = True
    field(e, 'key', 1, T.TYPE_STRING)
    field(e, 'value', 2, T.TYPE_MESSAGE, 1, '.tensorflow.' + val)
    field(m, key, 1, T.TYPE_MESSAGE, 3, '.tensorflow.%s.%s' % (outer, entry))
  m = msg('SequenceExample')
  field(m, 'context', 1, T.TYPE_MESSAGE, 1, '.tensorflow.Features')
  field(m, 'feature_lists', 2, T.TYPE_MESSAGE, 1, '.tensorflow.FeatureLists')
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.SequenceExample'))


def test_reader_and_writer_interoperate_with_the_protobuf_runtime(tmp_path):
  """(a) An episode ENCODED BY THE PROTOBUF RUNTIME (map entries in the runtime's own order, its varint / packed encodings),
  framed as a TFRecord with this file's bit-wise CRC-32C and compressed by Python's zlib, is read by the native reader and by the
  pure-Python reader: every array equals its source.  (b) An episode written by input_fn.write_episode is un-framed here by hand
  and DECODED BY THE PROTOBUF RUNTIME: every feature list equals what was written."""
  import struct
  pytest.importorskip('google.protobuf')
  from test_host_logic_cpu import _masked
  from geeco_amd import input_fn as I
  SE = _tf_example_classes()
  meta, eps = _make_dataset(str(tmp_path / 'ours'), n_eps=1, T=6, H=10, W=12)
  d = eps[0]
  T_, joints = 6, list(meta.monitored_joints)
  # ---- (a) protobuf-encoded episode -> our readers ----
  se = SE()
  ctx = se.context.feature
  for k, v in (('episode_length', meta.episode_length), ('img_height', meta.img_height), ('img_width', meta.img_width),
               ('dim_cmd', meta.dim_cmd), ('dim_ctrl', meta.dim_ctrl)):
    ctx[k].int64_list.value.append(int(v))
  for k, names in (('monitored_joints', joints), ('actuated_joints', list(meta.actuated_joints)),
                   ('monitored_mocaps', list(meta.monitored_mocaps)), ('monitored_objects', list(meta.monitored_objects))):
    ctx[k].bytes_list.value.extend(n.encode() for n in names)
  ctx['task_goal'].bytes_list.value.append(b'goal')
  ctx['task_object'].bytes_list.value.append(b'object')
  fl = se.feature_lists.feature_list
  for t in range(T_):
    fl['step'].feature.add().int64_list.value.append(t)
    fl['ts'].feature.add().float_list.value.append(0.04 * t)
    fl['rgb'].feature.add().float_list.value.extend(d['rgb'][t].reshape(-1).astype(np.float32).tolist())
    fl['depth'].feature.add().float_list.value.extend(d['depth'][t].reshape(-1).tolist())
    for key, arr in (('cmd', d['cmd']), ('ctrl', d['ctrl']), ('goal_qpos', d['goal']), ('obj_qpos', d['obj'])):
      fl[key].feature.add().float_list.value.extend(arr[t].tolist())
    for j, name in enumerate(joints):
      fl['joint_qpos-%s' % name].feature.add().float_list.value.append(float(d['qpos'][t, j]))
      fl['joint_qvel-%s' % name].feature.add().float_list.value.append(float(d['qvel'][t, j]))
    for name in meta.monitored_mocaps:
      fl['mocap_qpos-%s' % name].feature.add().float_list.value.extend(d['mocap'][t].tolist())
    for name in meta.monitored_objects:
      fl['object_qpos-%s' % name].feature.add().float_list.value.extend(d['obj'][t].tolist())
  payload = se.SerializeToString()
  ln = struct.pack('<Q', len(payload))
  record = ln + struct.pack('<I', _masked(ln)) + payload + struct.pack('<I', _masked(payload))
  path = str(tmp_path / 'pb.tfrecord.zlib')
  open(path, 'wb').write(zlib.compress(record, 6))
  ours = _paths(str(tmp_path / 'ours'))[0]
  for reader in (I.load_episode, I.load_episode_py):
    a, b = reader(path, meta, True, raw_rgb=True), reader(ours, meta, True, raw_rgb=True)
    assert set(a) == set(b)
    for k in a:
      np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg='%s %s' % (reader.__name__, k))
    np.testing.assert_array_equal(np.asarray(a['rgb']).astype(np.float32), d['rgb'][:-1].astype(np.float32))
    np.testing.assert_array_equal(a['jnt_state'], d['qpos'][:-1, :7])
  # ---- (b) our writer -> protobuf runtime ----
  raw = zlib.decompress(open(ours, 'rb').read())
  n, = struct.unpack('<Q', raw[:8])
  assert struct.unpack('<I', raw[8:12])[0] == _masked(raw[:8]) and struct.unpack('<I', raw[12 + n:16 + n])[0] == _masked(raw[12:12 + n])
  got = SE()
  got.ParseFromString(raw[12:12 + n])
  assert got.context.feature['episode_length'].int64_list.value[0] == meta.episode_length
  assert [v.decode() for v in got.context.feature['monitored_joints'].bytes_list.value] == joints
  gl = got.feature_lists.feature_list
  assert len(gl['rgb'].feature) == T_ and [f.int64_list.value[0] for f in gl['step'].feature] == list(range(T_))
  for t in range(T_):
    np.testing.assert_array_equal(np.asarray(gl['rgb'].feature[t].float_list.value, np.float32), d['rgb'][t].reshape(-1).astype(np.float32))
    np.testing.assert_array_equal(np.asarray(gl['depth'].feature[t].float_list.value, np.float32), d['depth'][t].reshape(-1))
    np.testing.assert_array_equal(np.asarray(gl['cmd'].feature[t].float_list.value, np.float32), d['cmd'][t])
    np.testing.assert_array_equal(np.float32(gl['joint_qpos-%s' % joints[3]].feature[t].float_list.value[0]), d['qpos'][t, 3])
