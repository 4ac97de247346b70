"""TensorBoard event files without TensorFlow.

The reference logs every term of ``GraphKeys.LOSSES`` with ``tf.summary.scalar`` + ``SummarySaverHook``
(``src/models/e2evmc/estimator.py:305-313``) and the Estimator adds loss / global_step; TensorBoard then reads
``<model_dir>/events.out.tfevents.*``.  This writer produces such a file: TFRecord framing (u64 length, masked
CRC-32C of the length, payload, masked CRC-32C of the payload; uncompressed) around ``Event`` protos
  Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
  Summary { repeated Value value = 1; }      Value { string tag = 1; float simple_value = 2; }
with the mandatory first record ``file_version = "brain.Event:2"``.
"""
from __future__ import annotations

import os
import socket
import struct
import time

from .tfrecord import _enc_len, _enc_varint, masked_crc32c


def _event(wall_time, step=None, file_version=None, scalars=None):
  out = _enc_varint((1 << 3) | 1) + struct.pack('<d', float(wall_time))
  if step is not None:
    out += _enc_varint((2 << 3) | 0) + _enc_varint(int(step))
  if file_version is not None:
    out += _enc_len(3, file_version.encode())
  if scalars:
    summary = b''
    for tag, value in scalars.items():
      v = _enc_len(1, tag.encode()) + _enc_varint((2 << 3) | 5) + struct.pack('<f', float(value))
      summary += _enc_len(1, v)
    out += _enc_len(5, summary)
  return out


class EventFileWriter:
  """Appends scalar summaries to ``<logdir>/events.out.tfevents.<time>.<host>``."""

  def __init__(self, logdir):
    os.makedirs(logdir, exist_ok=True)
    self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
    self._f = open(self.path, 'ab')
    self._write(_event(time.time(), file_version='brain.Event:2'))

  def _write(self, payload):
    hdr = struct.pack('<Q', len(payload))
    self._f.write(hdr + struct.pack('<I', masked_crc32c(hdr)) + payload + struct.pack('<I', masked_crc32c(payload)))
    self._f.flush()

  def add_scalars(self, scalars: dict, step: int, wall_time=None):
    self._write(_event(time.time() if wall_time is None else wall_time, step=step, scalars=scalars))

  def close(self):
    self._f.close()
