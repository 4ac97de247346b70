"""TF-1.15 checkpoint ("tensor bundle") reader / writer, without TensorFlow.

The reference stores and restores its weights with ``tf.train.Saver`` (Estimator checkpoints
``model.ckpt-<step>.{index,data-00000-of-00001}``; restore in ``src/models/e2evmc/predictor.py:85-95``,
snapshot export in ``scripts/train_e2evmc.py:143-205``); the published ``geeco_models_icra21`` weights
are such bundles.  This module maps them to / from the flat arenas of ``geeco_amd.variables``.

Format [TF 1.15: core/util/tensor_bundle, core/lib/io/table (a LevelDB table)]:
  <prefix>.index   table of  ""  -> BundleHeaderProto{num_shards=1, endianness=2, version=3}
                             name -> BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}
                   data blocks = prefix-compressed entries (varint shared, non_shared, value_len, key tail,
                   value) + fixed32 restart offsets + fixed32 count, each followed by a 5-byte trailer
                   (compression type 0, masked crc32c); then metaindex block, index block (last key ->
                   BlockHandle varint(offset), varint(size)) and a 48-byte footer ending in the magic
                   0xdb4775248b80fb57.
  <prefix>.data-00000-of-00001   raw little-endian tensor bytes at (offset, size).
Variable names: the model variables of SURVEY.md 8b, their Adam slots ``<var>/Adam`` (m) and
``<var>/Adam_1`` (v), ``beta1_power`` / ``beta2_power``, ``global_step`` (int64) and the unused
``.../LSTMDecoder/lstm_memory`` (skipped on import exactly as predictor.py:87 does).

NOT VERIFIED AGAINST A TF-WRITTEN FILE: no TensorFlow and no checkpoint exist in this environment.  The tests cover
writer -> reader round trips, a bundle assembled by hand from the documented table / proto layouts with an independent
CRC-32C (tests/test_host_logic_cpu.py::test_tf_bundle_hand_assembled_golden: reader and writer against a third party)
and the GPU end-to-end path Estimator -> bundle -> predictor (tests/test_predictor_gpu.py::test_tf_bundle_end_to_end).
"""
from __future__ import annotations

import struct

import numpy as np

from .tfrecord import _enc_varint, _fields, _varint, masked_crc32c

_MAGIC = 0xdb4775248b80fb57
_BLOCK_SIZE = 4096
_RESTART_INTERVAL = 16
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8')}   # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64
_DTYPE_ENUM = {np.dtype('<f4'): 1, np.dtype('<f8'): 2, np.dtype('<i4'): 3, np.dtype('<i8'): 9}


# ------------------------------------------------------------------------------------------------
# table reader
# ------------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
  data = bytes(buf[offset:offset + size])
  ctype = buf[offset + size]
  (crc,) = struct.unpack_from('<I', buf, offset + size + 1)
  if verify and masked_crc32c(data + bytes([ctype])) != crc:
    raise IOError('tensor bundle index: block checksum mismatch at %d' % offset)
  if ctype != 0:
    raise NotImplementedError('compressed index blocks (type %d) are not supported' % ctype)
  (nrestarts,) = struct.unpack_from('<I', data, len(data) - 4)
  end = len(data) - 4 - 4 * nrestarts
  out, pos, key = [], 0, b''
  while pos < end:
    shared, pos = _varint(data, pos)
    non_shared, pos = _varint(data, pos)
    vlen, pos = _varint(data, pos)
    key = key[:shared] + data[pos:pos + non_shared]
    pos += non_shared
    out.append((key, data[pos:pos + vlen]))
    pos += vlen
  return out


def _read_table(path, verify=True):
  with open(path, 'rb') as f:
    buf = f.read()
  if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != _MAGIC:
    raise IOError('%s is not a TF tensor-bundle index (bad magic)' % path)
  footer = buf[len(buf) - 48:]
  pos = 0
  _, pos = _varint(footer, pos)      # metaindex handle
  _, pos = _varint(footer, pos)
  ioff, pos = _varint(footer, pos)
  isize, pos = _varint(footer, pos)
  entries = []
  for _, handle in _read_block(buf, ioff, isize, verify):
    boff, p2 = _varint(handle, 0)
    bsize, _ = _varint(handle, p2)
    entries.extend(_read_block(buf, boff, bsize, verify))
  return entries


def _parse_entry(value):
  e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None}
  for fnum, wt, val in _fields(memoryview(value)):
    if fnum == 1:
      e['dtype'] = val
    elif fnum == 2:
      for f2, w2, dim in _fields(val):
        if f2 == 2:
          size = 0
          for f3, w3, v3 in _fields(dim):
            if f3 == 1:
              size = v3
          e['shape'].append(size)
    elif fnum == 3:
      e['shard_id'] = val
    elif fnum == 4:
      e['offset'] = val
    elif fnum == 5:
      e['size'] = val
    elif fnum == 6:
      e['crc32c'] = struct.unpack('<I', bytes(val))[0]
  return e


def read_checkpoint(prefix, verify=True):
  """-> {variable name: numpy array} for every tensor of the bundle ``<prefix>.index`` / ``.data-*``."""
  entries = _read_table(prefix + '.index', verify)
  num_shards = 1
  tensors = {}
  shards = {}
  for key, value in entries:
    if key == b'':
      for fnum, wt, val in _fields(memoryview(value)):
        if fnum == 1:
          num_shards = val
        elif fnum == 2 and val != 0:
          raise NotImplementedError('big-endian bundles are not supported')
      continue
    e = _parse_entry(value)
    if e['dtype'] not in _DTYPES:
      continue      # strings etc. are not model state
    sid = e['shard_id']
    if sid not in shards:
      with open('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), 'rb') as f:
        shards[sid] = f.read()
    raw = shards[sid][e['offset']:e['offset'] + e['size']]
    if verify and e['crc32c'] is not None and masked_crc32c(raw) != e['crc32c']:
      raise IOError('tensor %s: data checksum mismatch' % key.decode())
    tensors[key.decode()] = np.frombuffer(raw, dtype=_DTYPES[e['dtype']]).reshape(e['shape']).copy()
  return tensors


# ------------------------------------------------------------------------------------------------
# table writer
# ------------------------------------------------------------------------------------------------
def _build_block(items):
  out, restarts, prev = bytearray(), [], b''
  for i, (key, value) in enumerate(items):
    shared = 0
    if i % _RESTART_INTERVAL == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
        shared += 1
    out += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value)) + key[shared:] + value
    prev = key
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _emit_block(fout, block):
  off = fout.tell()
  fout.write(block)
  fout.write(b'\x00' + struct.pack('<I', masked_crc32c(block + b'\x00')))
  return off, len(block)


def _len_field(fnum, payload):
  return _enc_varint((fnum << 3) | 2) + _enc_varint(len(payload)) + payload


def _var_field(fnum, value):
  return _enc_varint((fnum << 3) | 0) + _enc_varint(value)


def write_checkpoint(prefix, tensors: dict):
  """Writes ``tensors`` ({name: numpy array of float32/float64/int32/int64}) as a one-shard bundle."""
  names = sorted(tensors.keys(), key=lambda s: s.encode())
  header = _var_field(1, 1) + _var_field(2, 0) + _len_field(3, _var_field(1, 1))     # num_shards, LITTLE, version.producer
  items = [(b'', header)]
  offset = 0
  with open(prefix + '.data-00000-of-00001', 'wb') as fdata:
    for name in names:
      arr = np.asarray(tensors[name])          # NB np.ascontiguousarray would turn a scalar (global_step) into shape [1]
      dt = arr.dtype.newbyteorder('<') if arr.dtype.byteorder == '>' else arr.dtype
      if np.dtype(dt) not in _DTYPE_ENUM:
        raise TypeError('%s: unsupported dtype %s' % (name, arr.dtype))
      raw = arr.astype(dt, order='C', copy=False).tobytes()
      shape = b''.join(_len_field(2, _var_field(1, int(d))) for d in arr.shape)
      entry = _var_field(1, _DTYPE_ENUM[np.dtype(dt)]) + _len_field(2, shape)
      if offset:
        entry += _var_field(4, offset)
      entry += _var_field(5, len(raw)) + _enc_varint((6 << 3) | 5) + struct.pack('<I', masked_crc32c(raw))
      items.append((name.encode(), entry))
      fdata.write(raw)
      offset += len(raw)
  with open(prefix + '.index', 'wb') as f:
    index_items, cur, cur_size = [], [], 0
    for key, value in items:
      cur.append((key, value))
      cur_size += len(key) + len(value) + 3
      if cur_size >= _BLOCK_SIZE:
        off, size = _emit_block(f, _build_block(cur))
        index_items.append((cur[-1][0], _enc_varint(off) + _enc_varint(size)))
        cur, cur_size = [], 0
    if cur:
      off, size = _emit_block(f, _build_block(cur))
      index_items.append((cur[-1][0], _enc_varint(off) + _enc_varint(size)))
    moff, msize = _emit_block(f, _build_block([]))
    ioff, isize = _emit_block(f, _build_block(index_items))
    footer = _enc_varint(moff) + _enc_varint(msize) + _enc_varint(ioff) + _enc_varint(isize)
    f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))


# ------------------------------------------------------------------------------------------------
# mapping to / from the variable store
# ------------------------------------------------------------------------------------------------
def import_checkpoint(store, prefix, load_optimizer=True):
  """Fills ``store`` (parameters, and Adam slots / global_step when present) from a TF checkpoint."""
  import torch
  t = read_checkpoint(prefix)
  missing = [n for n in store.shapes if n not in t]
  if missing:
    raise KeyError('checkpoint %s lacks variables: %s' % (prefix, missing[:5]))
  for name, shp in store.shapes.items():
    if tuple(t[name].shape) != tuple(shp):
      raise ValueError('%s: checkpoint shape %s != model shape %s' % (name, t[name].shape, tuple(shp)))
  store.load_numpy({n: t[n] for n in store.shapes})
  if load_optimizer:
    for arena, suffix in ((store.adam_m, '/Adam'), (store.adam_v, '/Adam_1')):
      for name, shp in store.shapes.items():
        if name + suffix in t:
          o = store.offsets[name]
          arena[o:o + int(np.prod(shp))].copy_(torch.from_numpy(t[name + suffix].astype(np.float32).reshape(-1)))
    if 'global_step' in t:
      store.global_step.fill_(int(np.asarray(t['global_step']).reshape(-1)[0]))
  return sorted(set(t) - set(store.shapes))


def export_checkpoint(store, prefix, lstm_memory_name=None, batch_size=None, beta1=0.9, beta2=0.999):
  """Writes the store as a TF-1.15 bundle with the reference's variable names (incl. Adam slots)."""
  tensors = dict(store.to_numpy('params'))
  m, v = store.to_numpy('adam_m'), store.to_numpy('adam_v')
  for name in store.shapes:
    tensors[name + '/Adam'] = m[name]
    tensors[name + '/Adam_1'] = v[name]
  step = int(store.global_step.item())
  tensors['global_step'] = np.asarray(step, np.int64)
  tensors['beta1_power'] = np.asarray(beta1 ** (step + 1), np.float32)     # AdamOptimizer keeps beta^(t+1) after t updates
  tensors['beta2_power'] = np.asarray(beta2 ** (step + 1), np.float32)
  if lstm_memory_name and batch_size:
    H4 = store.shapes[[n for n in store.shapes if n.endswith('lstm_cell/bias')][0]][0]
    tensors[lstm_memory_name] = np.zeros([batch_size, H4 // 2], np.float32)   # never assigned (graph.py:219-226)
  write_checkpoint(prefix, tensors)
