"""ctypes binding of the development-only entry point geeco_conv1_conv2_fwd (conv_bottom_fwd.hip in this directory; declared in
conv_bottom_fwd.h, NOT in include/geeco_hip.h: the product library does not export it)."""
import ctypes

from geeco_amd import _native
from geeco_amd.ops import _p, _stream, check

_I, _L, _P = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
_bound = False


def _lib():
  global _bound
  lib = _native.load()
  if not _bound:
    if not hasattr(lib, 'geeco_conv1_conv2_fwd'):
      raise RuntimeError('geeco_conv1_conv2_fwd needs the development library: scripts/dev/build_dev_lib.sh, then '
                         'GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so')
    lib.geeco_conv1_conv2_fwd.restype = _I
    lib.geeco_conv1_conv2_fwd.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _P]
    _bound = True
  return lib


def conv1_conv2_fwd_into(y1, bits, y2, fields, x, w1, b1, w2, b2, G, gs_x, gs_w1, gs_b1, gs_y1, gs_bits, gs_w2, gs_b2, gs_y2,
                         gs_fields, N, H, W, w_cin):
  """Encoder bottom forward in one launch (conv1 -> conv2, y1 halo through LDS); bits / fields may be None."""
  check(_lib().geeco_conv1_conv2_fwd(_p(x), _p(w1), _p(b1), _p(y1), _p(bits) if bits is not None else None, _p(w2), _p(b2), _p(y2),
                                     _p(fields) if fields is not None else None, G, gs_x, gs_w1, gs_b1, gs_y1, gs_bits, gs_w2,
                                     gs_b2, gs_y2, gs_fields, N, H, W, w_cin, _stream()), 'geeco_conv1_conv2_fwd')


def launch_fwd_bottom(enc):
  """conv1 -> conv2 of a graph.ConvEncoderStack in one launch: y1, y2 and, in training, conv1's sign words / conv2's sign fields
  (what graph.ConvEncoderStack.forward did for layers 0 and 1 under GEECO_FUSED_FWD=1 in round 4)."""
  G, Nf, L0 = enc.G, enc.Nf, enc.layers[0]
  x, y1, y2 = enc.x_in, enc.acts[0], enc.acts[1]
  if enc.pad1 and enc.pad1_copy:
    w1, gs_w1, w_cin = enc.w1p, enc.w1p[0].numel(), 4
  else:
    w1, gs_w1, w_cin = enc._w(0), enc.gs_p, enc.Cin
  bits = enc.bits1 if (enc.training and enc.relu_bits) else None
  fields = enc.fields2 if (enc.training and enc.relu_fields) else None
  conv1_conv2_fwd_into(y1, bits, y2, fields, x, w1, enc._b(0), enc._w(1), enc._b(1), G, x[0].numel(), gs_w1, enc.gs_p,
                       y1[0].numel(), bits[0].numel() if bits is not None else 0, enc.gs_p, enc.gs_p, y2[0].numel(),
                       fields[0].numel() if fields is not None else 0, Nf, L0['H'], L0['W'], w_cin)
