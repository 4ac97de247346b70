"""Host-side logic that needs no GPU: config mirror, variable arena, TFRecord codec, the dataset
reader's windowing against a literal restatement of the reference's pipeline, checkpoints."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import geeco_oracle as O


def test_params_mirror():
  from geeco_amd import params as P
  assert list(P.E2EVMCConfig._fields) == list(O.DEFAULT_PARAMS.keys())           # same names, same order
  assert P.E2E_VMC_DEFAULT_CONFIG._asdict() == dict(O.DEFAULT_PARAMS)
  c = P.create_e2evmc_config({'window_size': 16, 'unknown': 1})
  assert c.window_size == 16 and not hasattr(c, 'unknown')


def test_variable_store_matches_tf_names_and_counts():
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  for goal, kw in ((True, dict(proc_obs='dynimg', proc_tgt='dyndiff')), (False, {}), (True, dict(img_channels=4, proc_obs='dynimg'))):
    cfg = create_e2evmc_config(kw)
    shapes = model_variable_shapes(cfg, goal)
    ref = O.model_param_shapes(O.make_config(**kw), goal)
    assert list(shapes.items()) == list(ref.items())
  st = VariableStore(model_variable_shapes(create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff')), True), 'cpu')
  assert st.count_parameters() == 7552796
  assert all(o % 4 == 0 for o in st.offsets.values())                           # 16-byte aligned variables
  st.initialize(seed=3)
  k = st.var('GoalVMC/ConvEncoder/conv1/kernel')
  lim = np.sqrt(6.0 / (27 + 9 * 32))
  assert float(k.abs().max()) <= lim and float(k.abs().max()) > 0.9 * lim       # glorot-uniform limits
  assert float(st.var('GoalVMC/ConvEncoder/conv1/bias').abs().max()) == 0.0
  sd = st.state_dict()
  st2 = VariableStore(st.shapes, 'cpu')
  st2.load_state_dict(sd)
  assert torch.equal(st.params, st2.params)
  with pytest.raises(ValueError):
    VariableStore(model_variable_shapes(create_e2evmc_config({}), False), 'cpu').load_state_dict(sd)
  with pytest.raises(ValueError):
    model_variable_shapes(create_e2evmc_config(dict(proc_obs='bad')), True)
  with pytest.raises(ValueError):
    model_variable_shapes(create_e2evmc_config(dict(img_channels=2)), True)


def test_tfrecord_roundtrip_and_corruption(tmp_path):
  from geeco_amd import tfrecord as T
  assert T._host().geeco_crc32c(b'123456789', 9, 0) == 0xe3069283                # CRC-32C check value
  assert T._host().geeco_crc32c(b'\x00' * 32, 32, 0) == 0x8a9136aa                # RFC 3720 B.4
  ctx = {'episode_length': np.array([2], np.int64), 'task_goal': 'pad', 'names': ['a', 'bc']}
  frames = [{'step': np.array([i], np.int64), 'neg': np.array([-3 - i], np.int64),
             'rgb': (np.arange(12) + i).astype(np.uint8), 'x': np.array([0.25 * i, -1.5], np.float32)} for i in range(2)]
  fn = str(tmp_path / 'a.tfrecord.zlib')
  T.write_records(fn, [T.encode_sequence_example(ctx, frames), b'second'])
  recs = [bytes(r) for r in T.read_records(fn)]
  assert recs[1] == b'second'
  c, fl = T.parse_sequence_example(recs[0])
  assert c['task_goal'] == [b'pad'] and c['names'] == [b'a', b'bc'] and c['episode_length'].tolist() == [2]
  assert [f.tolist() for f in fl['neg']] == [[-3], [-4]]
  assert fl['rgb'][1].dtype == np.float32 and fl['rgb'][1].tolist() == list(map(float, range(1, 13)))
  assert fl['x'][1].tolist() == [0.25, -1.5]
  # flip one payload byte inside the zlib stream's plaintext -> CRC mismatch
  import zlib
  raw = bytearray(zlib.decompress(open(fn, 'rb').read()))
  raw[20] ^= 0x01
  open(fn, 'wb').write(zlib.compress(bytes(raw)))
  with pytest.raises(IOError):
    list(T.read_records(fn))
  assert len(list(T.read_records(fn, verify=False))) == 2


def _make_dataset(root, n_eps=2, T=9, H=8, W=8):
  from geeco_amd import input_fn as I
  joints = ['robot0:%s' % j for j in I._ARM_JOINTS + I._FINGER_JOINTS]
  meta = I.PickAndPlaceMetaV4(episode_length=T, img_height=H, img_width=W, monitored_joints=joints,
                              actuated_joints=joints[:2], monitored_mocaps=['robot0:mocap'],
                              monitored_objects=['object0:joint'], dim_cmd=4, dim_ctrl=2)
  os.makedirs(os.path.join(root, 'meta')); os.makedirs(os.path.join(root, 'data')); os.makedirs(os.path.join(root, 'splits', 'default'))
  json.dump(meta._asdict(), open(os.path.join(root, 'meta', 'meta_info.json'), 'w'))
  r = np.random.default_rng(0)
  eps = []
  for e in range(n_eps):
    d = dict(rgb=r.integers(0, 256, [T, H, W, 3]).astype(np.uint8), depth=r.random([T, H, W, 1]).astype(np.float32),
             cmd=np.concatenate([r.standard_normal([T, 3]), r.integers(-1, 2, [T, 1])], 1).astype(np.float32),
             ctrl=r.standard_normal([T, 2]).astype(np.float32), qpos=r.standard_normal([T, 9]).astype(np.float32),
             qvel=r.standard_normal([T, 9]).astype(np.float32), mocap=r.standard_normal([T, 7]).astype(np.float32),
             obj=r.standard_normal([T, 7]).astype(np.float32), goal=r.standard_normal([T, 7]).astype(np.float32))
    I.write_episode(os.path.join(root, 'data', 'ep%03d.tfrecord.zlib' % e), meta, d['rgb'], d['depth'], d['cmd'], d['ctrl'],
                    d['qpos'], d['qvel'], d['mocap'], d['obj'], d['goal'])
    eps.append(d)
  names = '\n'.join('ep%03d.tfrecord.zlib' % e for e in range(n_eps))
  for mode in ('train', 'eval'):
    open(os.path.join(root, 'splits', 'default', mode + '.txt'), 'w').write(names + '\n')
  return meta, eps


def test_pickplace_input_fn_windows(tmp_path):
  """Windows, labels and batching vs a literal restatement of geeco_gym.py:291-399, 598-631."""
  from geeco_amd.input_fn import pickplace_input_fn
  T, K, B = 9, 3, 4
  meta, eps = _make_dataset(str(tmp_path), n_eps=2, T=T)
  got = list(pickplace_input_fn(str(tmp_path), 'default', 'eval', window_size=K, fetch_target=True, batch_size=B))
  # expected: per episode, drop the last frame (T-1 = 8 frames), 8-K+1 = 6 windows; batches of 4 span episodes: 4,4,4
  assert [len(f['step']) for f, _ in got] == [4, 4, 4]
  exp_f, exp_l = [], []
  for d in eps:
    rgb = d['rgb'].astype(np.float32) / np.float32(255.0)
    ee_t, vel_t, grp_t = np.roll(d['mocap'], -1, 0), np.roll(d['qvel'][:, :7], -1, 0), np.roll(d['qpos'][:, 7:9], -1, 0)
    for i in range((T - 1) - K + 1):
      sl = slice(i, i + K)
      exp_f.append(dict(rgb=rgb[:-1][sl], depth=d['depth'][:-1][sl], jnt_state=d['qpos'][:-1, :7][sl],
                        vel_state=d['qvel'][:-1, :7][sl], grp_state=d['qpos'][:-1, 7:9][sl], ee_state=d['mocap'][:-1][sl],
                        obj_state=d['obj'][:-1][sl], goal_state=d['goal'][:-1][sl], cmd=d['cmd'][:-1][sl],
                        step=np.arange(T - 1)[sl], target_rgb=rgb[-1], target_depth=d['depth'][-1]))
      exp_l.append(dict(cmd=d['cmd'][i + K - 1], ctrl=d['ctrl'][i + K - 1], vel_target=vel_t[i + K - 1],
                        ee_target=ee_t[i + K - 1], grp_target=grp_t[i + K - 1]))
  flat_f = [{k: f[k][j] for k in f} for f, _ in got for j in range(len(f['step']))]
  flat_l = [{k: l[k][j] for k in l} for _, l in got for j in range(len(l['cmd']))]
  assert len(flat_f) == len(exp_f) == 12
  for a, b in zip(flat_f, exp_f):
    for k, v in b.items():
      np.testing.assert_array_equal(a[k], v, err_msg=k)
  for a, b in zip(flat_l, exp_l):
    for k, v in b.items():
      np.testing.assert_array_equal(a[k], v, err_msg=k)
  f0 = got[0][0]
  assert f0['rgb'].shape == (4, K, 8, 8, 3) and f0['rgb'].dtype == np.float32 and f0['step'].dtype == np.int64
  assert f0['target_rgb'].shape == (4, 8, 8, 3) and f0['depth'].shape == (4, K, 8, 8, 1)
  # ragged final batch (no drop_remainder) and rank-strided sharding
  got5 = list(pickplace_input_fn(str(tmp_path), 'default', 'eval', window_size=K, batch_size=5))
  assert [len(f['step']) for f, _ in got5] == [5, 5, 2]
  sh = list(pickplace_input_fn(str(tmp_path), 'default', 'eval', window_size=K, batch_size=6, shard=(1, 2)))
  assert len(sh) == 1 and 'target_rgb' not in sh[0][0]
  np.testing.assert_array_equal(sh[0][0]['cmd'][0], eps[1]['cmd'][:K])
  with pytest.raises(KeyError):
    pickplace_input_fn(str(tmp_path), 'default', 'eval', encoding='v2')


def test_checkpoint_files_and_latest(tmp_path):
  from geeco_amd import estimator as est
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  st = VariableStore(model_variable_shapes(create_e2evmc_config({}), False), 'cpu')
  st.initialize(1)
  assert est.latest_checkpoint(str(tmp_path)) is None
  for step in (3, 7, 12):
    st.global_step.fill_(step)
    est.save_checkpoint(st, str(tmp_path), keep_max=2)
  assert sorted(f for f in os.listdir(tmp_path) if f.endswith('.pt')) == ['model.ckpt-12.pt', 'model.ckpt-7.pt']
  ck = est.latest_checkpoint(str(tmp_path))
  assert os.path.basename(ck) == 'model.ckpt-12'
  assert 'model_checkpoint_path: "model.ckpt-12"' in open(tmp_path / 'checkpoint').read()
  st2 = VariableStore(st.shapes, 'cpu')
  est.load_checkpoint(st2, ck)
  assert torch.equal(st2.params, st.params) and int(st2.global_step) == 12


def test_estimator_refuses_cpu(tmp_path):
  """No silent CPU fallback: the product path fails loudly without a GPU."""
  from geeco_amd import estimator as est
  from geeco_amd.input_fn import synthetic_batches
  from geeco_amd.params import create_e2evmc_config
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  e = est.Estimator(est.e2evmc_model_fn, str(tmp_path), est.RunConfig(),
                    {'e2evmc_config': create_e2evmc_config(dict(img_height=136, img_width=136)), 'log_steps': 1})
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    e.train(input_fn=synthetic_batches(2, 4, 1, (136, 136), 3, False))


def test_tf_checkpoint_bundle_roundtrip(tmp_path):
  """TF-1.15 tensor-bundle writer -> reader (multi-block index, checksums, dtype/shape/offset fields) and
  the mapping to / from the variable store.  (Not verified against a TF-written file: none exists here.)"""
  import struct
  from geeco_amd import tf_checkpoint as C
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  st = VariableStore(model_variable_shapes(create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff')), True), 'cpu')
  st.initialize(2)
  st.adam_m.normal_(); st.adam_v.uniform_(); st.global_step.fill_(4321)
  prefix = str(tmp_path / 'model.ckpt-4321')
  C.export_checkpoint(st, prefix, lstm_memory_name='GoalVMC/LSTMDecoder/lstm_memory', batch_size=32)
  raw = open(prefix + '.index', 'rb').read()
  assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) > 4096 + 1024   # more than one data block
  t = C.read_checkpoint(prefix)
  assert len(t) == 3 * 60 + 4                                          # vars + 2 slots each, global_step, beta powers, lstm_memory
  assert t['global_step'].dtype == np.int64 and int(t['global_step']) == 4321
  assert t['GoalVMC/LSTMDecoder/lstm_memory'].shape == (32, 256)
  np.testing.assert_array_equal(t['GoalVMC/DynDiffEncoder/conv4/kernel'], st.to_numpy()['GoalVMC/DynDiffEncoder/conv4/kernel'])
  st2 = VariableStore(st.shapes, 'cpu')
  extra = C.import_checkpoint(st2, prefix)
  assert 'GoalVMC/LSTMDecoder/lstm_memory' in extra and 'beta1_power' in extra
  for which in ('params', 'adam_m', 'adam_v'):      # named regions (the arena's alignment pads are not variables)
    a, b = st.to_numpy(which), st2.to_numpy(which)
    assert all(np.array_equal(a[k], b[k]) for k in a), which
  assert int(st2.global_step) == 4321
  # corruption is detected (flip a data byte; flip an index byte)
  d = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read()); d[100] ^= 1
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(d))
  with pytest.raises(IOError):
    C.read_checkpoint(prefix)
  i = bytearray(raw); i[50] ^= 1
  open(prefix + '.index', 'wb').write(bytes(i))
  with pytest.raises(IOError):
    C.read_checkpoint(prefix)
  with pytest.raises(KeyError):
    C.write_checkpoint(str(tmp_path / 'small'), {'a': np.zeros([2, 2], np.float32)})
    C.import_checkpoint(st2, str(tmp_path / 'small'))


def test_dp_sharded_input_schedule(tmp_path):
  """pickplace_input_fn(shard=(rank, world)): disjoint rank-strided episodes, one shuffle order for all ranks, and a
  dp_schedule that tells every rank how many windows EVERY rank holds per step (ragged ends included)."""
  from geeco_amd.estimator import Estimator
  from geeco_amd.input_fn import pickplace_input_fn
  T, K, B, world = 9, 3, 4, 2
  _make_dataset(str(tmp_path), n_eps=3, T=T)         # 3 episodes x 6 windows: rank 0 gets 2 episodes, rank 1 gets 1
  its = [pickplace_input_fn(str(tmp_path), 'default', 'train', window_size=K, batch_size=B, shard=(r, world), seed=7)
         for r in range(world)]
  assert its[0].dp_schedule == its[1].dp_schedule == [(4, 4), (4, 2), (4, 0)]
  got = [list(it) for it in its]
  assert [len(f['step']) for f, _ in got[0]] == [4, 4, 4] and [len(f['step']) for f, _ in got[1]] == [4, 2]
  # the two ranks together see every window of the split exactly once
  full = list(pickplace_input_fn(str(tmp_path), 'default', 'train', window_size=K, batch_size=100, seed=7))[0][0]
  key = lambda f: sorted(map(tuple, np.concatenate([f['cmd'].reshape(len(f['cmd']), -1)], 0).tolist()))
  seen = sorted(sum([key(f) for r in range(world) for f, _ in got[r]], []))
  assert seen == key(full)
  with pytest.raises(ValueError):    # every rank must use the same shuffle seed
    pickplace_input_fn(str(tmp_path), 'default', 'train', window_size=K, batch_size=B, shard=(0, 2), seed=None)
  # slicing of an unsharded (global) batch: contiguous, sizes differ by at most one, nothing dropped
  batch = ({'step': np.arange(7)[:, None], 'x': np.arange(7)}, {'cmd': np.arange(7)})
  parts = [Estimator._shard(batch, 3, r) for r in range(3)]
  assert [p[2] for p in parts] == [3, 2, 2] and all(p[3] == 7 for p in parts)
  assert np.concatenate([p[0]['x'] for p in parts]).tolist() == list(range(7))
  f, l, n, ng = Estimator._shard(({'step': np.arange(1)[:, None]}, None), 2, 1)
  assert f is None and n == 0 and ng == 1


def test_variable_store_uniform_encoder_stride():
  """Unequal dim_s_*: the encoders' variable blocks still start at one common stride (grouped launches address
  encoder g at base + g * stride); equal dims keep the packed layout of the default model."""
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  scopes = ['GoalVMC/ConvEncoder', 'GoalVMC/DynBuffEncoder', 'GoalVMC/DynDiffEncoder']
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', dim_s_obs=64, dim_s_dyn=256, dim_s_diff=128))
  shapes = model_variable_shapes(cfg, True)
  st = VariableStore(shapes, 'cpu', uniform_scopes=scopes)
  for l in range(1, 8):
    for kind in ('kernel', 'bias'):
      o = [st.offsets['%s/conv%d/%s' % (sc, l, kind)] for sc in scopes]
      assert o[1] - o[0] == o[2] - o[1] > 0
  assert st.count_parameters() == O.count_parameters(O.model_param_shapes(O.make_config(**cfg._asdict()), True))
  ends = sorted((o, o + int(np.prod(shapes[n]))) for n, o in st.offsets.items())
  assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:]))                     # no overlap
  dflt = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff'))
  a = VariableStore(model_variable_shapes(dflt, True), 'cpu', uniform_scopes=scopes)
  b = VariableStore(model_variable_shapes(dflt, True), 'cpu')
  assert a.offsets == b.offsets and a.size == b.size


def _crc32c_bitwise(data):
  """Independent CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bit by bit: not the product's table code."""
  crc = 0xFFFFFFFF
  for b in data:
    crc ^= b
    for _ in range(8):
      crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
  return crc ^ 0xFFFFFFFF


def _masked(data):
  c = _crc32c_bitwise(data)
  return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_tf_bundle_hand_assembled_golden(tmp_path):
  """A tensor bundle assembled BY HAND in this test from the documented layouts (LevelDB table format: prefix-compressed
  entries, restart array, 5-byte block trailer with masked CRC-32C, metaindex + index blocks, 48-byte footer with the
  magic; tensor_bundle.proto field numbers), with its own bit-wise CRC-32C, so that reader and writer are not only
  checked against each other.  The index keys are SHORTENED separators as TF's table builder emits them
  (FindShortestSeparator / FindShortSuccessor), which the product's writer does not produce."""
  import struct
  from geeco_amd import tf_checkpoint as T
  assert _crc32c_bitwise(b'123456789') == 0xe3069283
  a = np.arange(6, dtype='<f4').reshape(2, 3) * 0.5
  g = np.asarray(77, dtype='<i8')
  data = a.tobytes() + g.tobytes()
  open(str(tmp_path / 'm.data-00000-of-00001'), 'wb').write(data)

  def varint(v):
    out = bytearray()
    while True:
      b = v & 0x7f
      v >>= 7
      out.append(b | (0x80 if v else 0))
      if not v:
        return bytes(out)
  def shape_proto(dims):        # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
    return b''.join(b'\x12' + varint(len(b'\x08' + varint(d))) + b'\x08' + varint(d) for d in dims)
  def entry(dtype, dims, offset, size, payload):      # BundleEntryProto: dtype=1 shape=2 shard_id=3 offset=4 size=5 crc32c=6 (fixed32)
    sp = shape_proto(dims)
    e = b'\x08' + varint(dtype) + b'\x12' + varint(len(sp)) + sp
    if offset:
      e += b'\x20' + varint(offset)
    return e + b'\x28' + varint(size) + b'\x35' + struct.pack('<I', _masked(payload))
  header = b'\x08\x01' + b'\x10\x00' + b'\x1a\x02\x08\x01'     # BundleHeaderProto: num_shards=1, endianness=LITTLE, version{producer=1}
  kv = [(b'', header), (b'enc/a', entry(1, [2, 3], 0, 24, a.tobytes())), (b'enc/global_step', entry(9, [], 24, 8, g.tobytes()))]

  def block(items, restart_every=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
      shared = 0
      if i % restart_every == 0:
        restarts.append(len(out))
      else:
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
          shared += 1
      out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
      prev = k
    for r in restarts or [0]:
      out += struct.pack('<I', r)
    return bytes(out + struct.pack('<I', len(restarts or [0])))
  def with_trailer(b):
    return b + b'\x00' + struct.pack('<I', _masked(b + b'\x00'))
  # two data blocks (split after the second key) so that the index has two entries with shortened separator keys
  b0, b1 = block(kv[:2]), block(kv[2:])
  f = bytearray()
  h0 = (len(f), len(b0)); f += with_trailer(b0)
  h1 = (len(f), len(b1)); f += with_trailer(b1)
  meta = block([])
  hm = (len(f), len(meta)); f += with_trailer(meta)
  idx = block([(b'enc/b', varint(h0[0]) + varint(h0[1])),       # 'enc/a' < 'enc/b' <= 'enc/global_step': shortest separator
               (b'f', varint(h1[0]) + varint(h1[1]))])          # short successor of 'enc/global_step'
  hi = (len(f), len(idx)); f += with_trailer(idx)
  footer = varint(hm[0]) + varint(hm[1]) + varint(hi[0]) + varint(hi[1])
  f += footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
  open(str(tmp_path / 'm.index'), 'wb').write(bytes(f))

  t = T.read_checkpoint(str(tmp_path / 'm'))
  assert set(t) == {'enc/a', 'enc/global_step'}
  np.testing.assert_array_equal(t['enc/a'], a)
  assert t['enc/global_step'].dtype == np.int64 and int(t['enc/global_step']) == 77
  # the product's writer, given the same tensors, produces the same data file, per-tensor entries and header bytes
  T.write_checkpoint(str(tmp_path / 'w'), {'enc/a': a, 'enc/global_step': g})
  assert open(str(tmp_path / 'w.data-00000-of-00001'), 'rb').read() == data
  got = dict(T._read_table(str(tmp_path / 'w.index')))
  assert {k: bytes(v) for k, v in got.items()} == dict(kv)
  # corrupting one byte of a data block is caught by the trailer CRC
  bad = bytearray(f); bad[3] ^= 0x40
  open(str(tmp_path / 'bad.index'), 'wb').write(bytes(bad))
  open(str(tmp_path / 'bad.data-00000-of-00001'), 'wb').write(data)
  with pytest.raises(IOError):
    T.read_checkpoint(str(tmp_path / 'bad'))


def test_tensorboard_event_file(tmp_path):
  """summary.EventFileWriter: TFRecord-framed Event protos (file_version first, then step + scalar values), parsed back
  here field by field; CRCs are verified by the record reader."""
  import struct
  from geeco_amd import tfrecord as T
  from geeco_amd.summary import EventFileWriter
  w = EventFileWriter(str(tmp_path))
  w.add_scalars({'loss': 1.5, 'loss_cmd_ee': 0.25}, step=7, wall_time=123.5)
  w.add_scalars({'loss': 1.25}, step=300)
  w.close()
  assert os.path.basename(w.path).startswith('events.out.tfevents.')
  recs = [bytes(r) for r in T.read_records(w.path, compression=None)]
  assert len(recs) == 3

  def parse(rec):
    ev = {}
    for fnum, wt, val in T._fields(memoryview(rec)):
      if fnum == 1:
        ev['wall_time'] = struct.unpack('<d', bytes(val))[0]
      elif fnum == 2:
        ev['step'] = val
      elif fnum == 3:
        ev['file_version'] = bytes(val).decode()
      elif fnum == 5:
        ev['scalars'] = {}
        for f2, _, v2 in T._fields(val):
          tag, sv = None, None
          for f3, _, v3 in T._fields(v2):
            if f3 == 1:
              tag = bytes(v3).decode()
            elif f3 == 2:
              sv = struct.unpack('<f', bytes(v3))[0]
          ev['scalars'][tag] = sv
    return ev
  e0, e1, e2 = map(parse, recs)
  assert e0['file_version'] == 'brain.Event:2'
  assert e1 == {'wall_time': 123.5, 'step': 7, 'scalars': {'loss': 1.5, 'loss_cmd_ee': 0.25}}
  assert e2['step'] == 300 and e2['scalars'] == {'loss': 1.25}


def test_bench_refuses_missing_gpus():
  """`python bench.py --gpus N` without a launcher must start N ranks itself or fail loudly: with fewer visible GPUs than
  ranks it exits non-zero BEFORE touching a GPU and prints no JSON line (a driver must never record a 1-GPU number as
  the N-GPU result)."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
  out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=env, timeout=300)
  assert out.returncode == 2, (out.returncode, out.stderr[-500:])
  assert '"metric"' not in out.stdout
  assert 'only' in out.stderr and 'GPU' in out.stderr


def test_bench_reads_what_rccl_logged(tmp_path, monkeypatch):
  """bench.py sends RCCL's INIT / GRAPH lines to a private file and reports version, collective channels, rings and transports
  in ``comm.rccl`` (the channel count decides how many CUs the early bucket wants beside part 2).  The parser on lines of the
  shape RCCL 2.26 writes; a caller who directs the log itself is left alone; a plain NCCL_DEBUG=VERSION is raised to INFO."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, root)
  try:
    import bench
  finally:
    sys.path.remove(root)
  log = tmp_path / 'rccl.log'
  log.write_text('\n'.join([
      'host:123:123 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6',
      'host:123:140 [0] NCCL INFO Pattern 4, crossNic 0, nChannels 28, bw 48.000000/48.000000, type XGMI/PIX, sameChannels 1',
      'host:123:140 [0] NCCL INFO Channel 00/28 : 0 1 2 3 4 5 6 7',
      'host:123:140 [0] NCCL INFO Channel 01/28 : 0 2 4 6 1 3 5 7',
      'host:123:141 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC',
      'host:123:141 [0] NCCL WARN something worth repeating',
      'host:123:140 [0] NCCL INFO 28 coll channels, 28 collnet channels, 0 nvls channels, 32 p2p channels, 4 p2p channels per peer',
  ]))
  info = bench.rccl_info(str(log))
  assert info['status'] == 'ok' and info['coll_channels'] == 28 and info['version'].startswith('RCCL version : 2.26.6')
  assert info['ring_channels_seen'] == 2 and info['transports'] == ['P2P/IPC'] and not log.exists()
  assert any('coll channels' in l for l in info['lines'])
  assert bench.rccl_info(None)['status'].startswith('the caller directs')
  assert bench.rccl_info(str(tmp_path / 'missing.log'))['status'].startswith('no log')
  for k in ('NCCL_DEBUG', 'NCCL_DEBUG_FILE', 'NCCL_DEBUG_SUBSYS'):
    monkeypatch.delenv(k, raising=False)
  monkeypatch.setenv('NCCL_DEBUG_FILE', '/somewhere/else')
  assert bench.rccl_debug_on() is None and 'NCCL_DEBUG' not in os.environ
  monkeypatch.delenv('NCCL_DEBUG_FILE')
  monkeypatch.setenv('NCCL_DEBUG', 'VERSION')
  path = bench.rccl_debug_on()
  assert path and os.environ['NCCL_DEBUG'] == 'INFO' and os.environ['NCCL_DEBUG_FILE'] == path and os.environ['NCCL_DEBUG_SUBSYS'] == 'INIT,GRAPH'


def test_bench_watchdog_leaves_with_the_provisional_line(tmp_path):
  """bench.Watchdog (N > 1): when what follows the safe form's measurement does not finish in time, the provisional line is
  written and the process exits 0 although its main thread is stuck; a disarmed watchdog does nothing.  And the names of
  runtime.DP_FORMS resolve (default = the three-graph form, nothing captured)."""
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  prog = (
      "import sys, json, time, threading\n"
      "sys.path.insert(0, %r)\n"
      "import bench\n"
      "def emit(p, why):\n"
      "  p['comm'] = {'status': 'WATCHDOG ' + why}\n"
      "  print(json.dumps(p), flush=True)\n"
      "quiet = bench.Watchdog(0.2, emit); quiet.provisional = {'value': 0}; quiet.arm(); quiet.disarm()\n"
      "time.sleep(0.5)\n"
      "dog = bench.Watchdog(0.5, emit); dog.provisional = {'metric': 'm', 'value': 1.5}; dog.phase = 'trial'; dog.arm()\n"
      "threading.Event().wait(60)\n"             # the main thread 'hangs'
      "print('not reached')\n") % root
  prog2 = prog.replace("dog = bench.Watchdog(0.5, emit)", "flag = []; threading.Timer(1.5, lambda: flag.append(1)).start(); "
                       "dog = bench.Watchdog(45.0, emit, aborted=lambda: bool(flag))")
  t0 = __import__('time').time()
  out = subprocess.run([sys.executable, '-c', prog], capture_output=True, text=True, timeout=50)
  assert out.returncode == 0 and __import__('time').time() - t0 < 30, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1 and 'not reached' not in out.stdout
  d = json.loads(lines[0])
  assert d['value'] == 1.5 and d['comm']['status'].startswith('WATCHDOG what follows the safe form did not finish') and 'phase: trial' in d['comm']['status']
  assert 'WATCHDOG' in out.stderr and 'phase: trial' in out.stderr
  # a peer's abort flag ends the wait long before the deadline
  t0 = __import__('time').time()
  out = subprocess.run([sys.executable, '-c', prog2], capture_output=True, text=True, timeout=50)
  assert out.returncode == 0 and __import__('time').time() - t0 < 30, out.stderr[-2000:]
  d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
  assert 'another rank gave up' in d['comm']['status'] and 'not reached' not in out.stdout
  from geeco_amd.runtime import DP_FORMS, DP_FORM_DEFAULT, dp_form_kwargs
  assert DP_FORM_DEFAULT == 'three_graphs_reserve16' and dp_form_kwargs() == dict(overlap=True, capture_exchange=False, reserved_cus=16)
  assert dp_form_kwargs('three_graphs') == dict(overlap=True, capture_exchange=False)
  assert dp_form_kwargs('overlap_reserve16') == dict(overlap=True, capture_exchange=True, reserved_cus=16)
  assert all(set(kw) <= {'overlap', 'capture_exchange', 'reserved_cus', 'eager_adam'} for kw in DP_FORMS.values())
  assert dp_form_kwargs('two_graphs') == dict(overlap=True, capture_exchange=False, eager_adam=True)
  from geeco_amd.runtime import DP_CANDIDATES_CAPTURED, DP_CANDIDATES_SAFE
  assert all(not kw['capture_exchange'] for _, kw in DP_CANDIDATES_SAFE) and all(kw['capture_exchange'] for _, kw in DP_CANDIDATES_CAPTURED)
  with pytest.raises(ValueError):
    dp_form_kwargs('fastest')


def test_tfrecord_sequence_example_hand_assembled_golden(tmp_path):
  """A zlib TFRecord file with one tf.train.SequenceExample assembled BY HAND in this test from the published layouts
  (TFRecord framing: u64 length, masked CRC-32C of the length, payload, masked CRC-32C of the payload; example.proto /
  feature.proto field numbers; proto3 wire format) with its own bit-wise CRC-32C, including encodings the repo's writer
  never produces: UNPACKED repeated floats / int64s (one tag per element, as older writers emit), a negative int64 as a
  10-byte varint, map entries with value before key.  Reader (geeco_amd.tfrecord) against a third party."""
  import struct, zlib
  from geeco_amd import tfrecord as T

  def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
      b = v & 0x7f
      v >>= 7
      out.append(b | (0x80 if v else 0))
      if not v:
        return bytes(out)
  def ld(fnum, payload):                    # length-delimited field
    return varint((fnum << 3) | 2) + varint(len(payload)) + payload
  float_packed = lambda vals: ld(2, ld(1, b''.join(struct.pack('<f', v) for v in vals)))            # Feature.float_list = 2, FloatList.value = 1 (packed)
  float_unpacked = lambda vals: ld(2, b''.join(varint((1 << 3) | 5) + struct.pack('<f', v) for v in vals))
  int_packed = lambda vals: ld(3, ld(1, b''.join(varint(v) for v in vals)))                          # Feature.int64_list = 3
  int_unpacked = lambda vals: ld(3, b''.join(varint((1 << 3) | 0) + varint(v) for v in vals))
  bytes_list = lambda vals: ld(1, b''.join(ld(1, v) for v in vals))                                  # Feature.bytes_list = 1
  entry = lambda k, feat: ld(1, ld(1, k.encode()) + ld(2, feat))                                     # map<string, X>: key = 1, value = 2
  entry_rev = lambda k, feat: ld(1, ld(2, feat) + ld(1, k.encode()))
  feature_list = lambda feats: b''.join(ld(1, f) for f in feats)                                     # FeatureList.feature = 1
  context = entry('episode_length', int_packed([3])) + entry_rev('task_goal', bytes_list([b'pad2']))
  lists = (entry('step', feature_list([int_packed([0]), int_unpacked([1]), int_packed([-7])])) +
           entry_rev('rgb', feature_list([float_packed([0.0, 255.0, 17.0]), float_unpacked([1.0, 2.0, 3.0]), float_packed([4.5, 5.5, 6.5])])))
  example = ld(1, context) + ld(2, lists)            # SequenceExample.context = 1, .feature_lists = 2 (each a map field = 1)
  hdr = struct.pack('<Q', len(example))
  rec = hdr + struct.pack('<I', _masked(hdr)) + example + struct.pack('<I', _masked(example))
  fn = str(tmp_path / 'hand.tfrecord.zlib')
  open(fn, 'wb').write(zlib.compress(rec + rec))     # two records in one stream
  recs = [bytes(r) for r in T.read_records(fn)]
  assert len(recs) == 2 and recs[0] == example
  ctx, fl = T.parse_sequence_example(recs[1])
  assert ctx['episode_length'].tolist() == [3] and ctx['task_goal'] == [b'pad2']
  assert [f.tolist() for f in fl['step']] == [[0], [1], [-7]]
  assert [f.tolist() for f in fl['rgb']] == [[0.0, 255.0, 17.0], [1.0, 2.0, 3.0], [4.5, 5.5, 6.5]]
  assert fl['rgb'][0].dtype == np.float32 and fl['step'][2].dtype == np.int64
  # the repo's writer produces the same record bytes for the packed forms it emits
  ours = T.encode_sequence_example({'episode_length': np.array([3], np.int64), 'task_goal': 'pad2'},
                                   [{'step': np.array([i], np.int64)} for i in (0, 1, -7)])
  c2, f2 = T.parse_sequence_example(ours)
  assert c2['episode_length'].tolist() == [3] and [f.tolist() for f in f2['step']] == [[0], [1], [-7]]
  # a flipped payload bit is caught by the record checksum
  bad = bytearray(rec); bad[20] ^= 0x04
  open(str(tmp_path / 'bad.tfrecord.zlib'), 'wb').write(zlib.compress(bytes(bad)))
  with pytest.raises(IOError):
    list(T.read_records(str(tmp_path / 'bad.tfrecord.zlib')))


def test_target_frame_loaders_round_trip(tmp_path):
  """load_target_frame / load_keyframes / load_target_frames (geeco_gym.py:165-211) on a dataset directory written here:
  PNG goal images come back as float32 / 255, depth maps as the 4th channel, key frames in sorted order paired by
  position, and the key-frame branch is taken exactly when data/key_frames_<id>.json exists."""
  from PIL import Image
  from geeco_amd import input_fn as I
  root = str(tmp_path)
  for d in ('data', 'images/targets/rgb', 'images/targets/depth', 'images/keyframes/rgb', 'images/keyframes/depth'):
    os.makedirs(os.path.join(root, d))
  r = np.random.default_rng(3)
  H, W = 12, 20
  rgb = r.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
  depth = (0.5 + 2.5 * r.random((H, W))).astype(np.float32)
  Image.fromarray(rgb).save(os.path.join(root, 'images', 'targets', 'rgb', 'episode_0007.png'))
  np.save(os.path.join(root, 'images', 'targets', 'depth', 'episode_0007.npy'), depth)
  name = os.path.join(root, 'data', 'episode_0007.tfrecord.zlib')
  f4 = I.load_target_frame(root, name)
  assert f4.shape == (H, W, 4) and f4.dtype == np.float32
  assert np.array_equal(f4[..., :3], rgb.astype(np.float32) / 255.0) and np.array_equal(f4[..., 3], depth)
  f3 = I.load_target_frame(root, 'episode_0007.tfrecord.zlib', load_depth=False)
  assert f3.shape == (H, W, 3) and np.array_equal(f3, f4[..., :3])
  assert 0.0 <= f3.min() and f3.max() <= 1.0                     # what the predictor's range check expects
  # no key-frame index for record 0007: the single goal image
  got = I.load_target_frames(root, 'episode_0007.tfrecord.zlib')
  assert len(got) == 1 and np.array_equal(got[0], f4)
  # key frames: written out of order, another episode's files in the same directories
  kf = []
  for i in (2, 0, 1):
    a = r.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    d = r.random((H, W)).astype(np.float32)
    Image.fromarray(a).save(os.path.join(root, 'images', 'keyframes', 'rgb', 'episode_0007_key%d.png' % i))
    np.save(os.path.join(root, 'images', 'keyframes', 'depth', 'episode_0007_key%d.npy' % i), d)
    kf.append((i, a, d))
  Image.fromarray(rgb).save(os.path.join(root, 'images', 'keyframes', 'rgb', 'episode_0008_key0.png'))
  np.save(os.path.join(root, 'images', 'keyframes', 'depth', 'episode_0008_key0.npy'), depth)
  frames = I.load_keyframes(root, name)
  assert len(frames) == 3
  for (i, a, d), fr in zip(sorted(kf), frames):
    assert fr.shape == (H, W, 4) and np.array_equal(fr[..., :3], a.astype(np.float32) / 255.0) and np.array_equal(fr[..., 3], d)
  with open(os.path.join(root, 'data', 'key_frames_0007.json'), 'w') as fp:
    fp.write('[]')
  got = I.load_target_frames(root, 'episode_0007.tfrecord.zlib')
  assert len(got) == 3 and all(np.array_equal(x, y) for x, y in zip(got, frames))
  with pytest.raises(FileNotFoundError):
    I.load_target_frame(root, 'episode_0099.tfrecord.zlib')
