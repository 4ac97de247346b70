"""ctypes binding of the C-ABI library ``libgeeco_hip.so`` (declared in include/geeco_hip.h).

The library is the product's only compute path.  There is deliberately no CPU/PyTorch
fallback: if the shared object is missing or a symbol does not resolve, importing the ops
raises ``GeecoNativeError`` (the build recipe is ``geeco_amd/csrc/build.sh``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_void_p, POINTER, Structure

from . import _dev

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, _dev.env('GEECO_LIB', 'libgeeco_hip.so'))   # GEECO_DEV=1 GEECO_LIB=...: A/B builds side by side


GEECO_EINVAL, GEECO_ENOSUP = -1, -2      # include/geeco_hip.h
ABI_VERSION = 6        # GEECO_ABI_VERSION of include/geeco_hip.h this binding was written against


class GeecoNativeError(RuntimeError):
  pass


_P = c_void_p          # device (or host) pointer
_PP = POINTER(c_void_p)
_I, _L, _F = c_int, c_int64, c_float


class HeadsFinish(Structure):
  """geeco_heads_finish (include/geeco_hip.h): the pending batch sums of the heads' backward, opaque to the caller."""
  _fields_ = [('opaque', ctypes.c_ubyte * 1024)]


class SlabReduce(Structure):
  """geeco_slab_reduce (include/geeco_hip.h): one pending slab sum of a filter-gradient kernel."""
  _fields_ = [('part', c_void_p), ('dw', c_void_p), ('db', c_void_p), ('gs_dw', c_int64), ('gs_db', c_int64),
              ('KC', c_int64), ('S', c_int32), ('Cout', c_int32), ('groups', c_int32), ('reserved', c_int32)]


class AdamSegment(Structure):
  """geeco_adam_segment (include/geeco_hip.h): one piece of the arena with its own gradient source."""
  _fields_ = [('g', c_void_p), ('p_off', c_int64), ('count', c_int64)]


ADAM_SEGMENTS_MAX = 8

# symbol -> (restype, argtypes); mirrors include/geeco_hip.h one to one
SIGNATURES = {
    'geeco_abi_version': (_I, []),
    'geeco_last_error': (c_char_p, []),
    'geeco_has_dev_kernels': (_I, []),
    'geeco_debug_kernel_trace_begin': (None, []),
    'geeco_debug_kernel_trace_end': (c_char_p, []),
    'geeco_dynimg_alpha': (None, [_I, _P]),
    'geeco_dynimg_ws_bytes': (_L, [_I, _L]),
    'geeco_dynimg_fwd': (_I, [_P, _P, _L, _L, _P, _I, _I, _L, _I, _I, _P, _P, _P]),
    'geeco_goal_dynimgs_ws_bytes': (_L, [_I, _L]),
    'geeco_goal_dynimgs_fwd': (_I, [_P, _L, _L, _P, _P, _L, _L, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P]),
    'geeco_goal_dynimgs_u8_fwd': (_I, [_P, _P, _P, _L, _L, _P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _P]),
    'geeco_goal_dynimgs_timeouts': (_I, [_P, _I, _P, _P]),
    'geeco_goal_dynimgs_set_wait_polls': (ctypes.c_uint, [ctypes.c_uint]),
    'geeco_dynimg_rgbd_fwd': (_I, [_P, _P, _L, _L, _P, _P, _L, _L, _P, _I, _I, _L, _P, _P, _P]),
    'geeco_pack_pixels': (_I, [_P, _L, _P, _L, _I, _L, _I, _I, _I, _P, _P]),
    'geeco_gather_windows': (_I, [_P, _I, _P, _I, _I, _L, _F, _P, _P]),
    'geeco_conv3x3_fwd': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'geeco_conv3x3_fwd_state': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _L, _I, _I, _P, _L, _P]),
    'geeco_conv3x3_fwd_ws_bytes': (_L, [_I, _I, _I, _I, _I, _I, _I]),
    'geeco_conv3x3_dgrad': (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    'geeco_conv3x3_dgrad_ws_bytes': (_L, [_I, _I, _I, _I, _I, _I, _I]),
    'geeco_conv3x3_dgrad_needs_wt': (_I, [_I, _I, _I, _I, _I]),
    'geeco_conv3x3_dgrad_relu_fields_supported': (_I, [_I, _I, _I, _I, _I]),
    'geeco_conv3x3_wgrad_ws_bytes': (_L, [_I, _I, _I, _I, _I, _I, _I]),
    'geeco_conv3x3_wgrad': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    'geeco_conv2_dgrad_conv1_wgrad_ws_bytes': (_L, [_I]),
    'geeco_conv2_dgrad_conv1_wgrad': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P]),
    'geeco_conv3x3_wgrad_partial': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P,
                                         POINTER(SlabReduce), _I]),
    'geeco_conv2_dgrad_conv1_wgrad_partial': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I,
                                                   _P, _P, POINTER(SlabReduce), _I]),
    'geeco_slab_reduce_batch': (_I, [POINTER(SlabReduce), _I, _P]),
    'geeco_slab_reduce_batch_prepare': (_I, [POINTER(SlabReduce), _I, _P, _F, _F, _F, _P, _P]),
    'geeco_conv_top_bwd': (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P,
                                _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P,
                                _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, POINTER(SlabReduce)]),
    'geeco_conv3x3_wgrad_pair': (_I, [_P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P,
                                      _P, _P, _P, _P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, POINTER(SlabReduce)]),
    'geeco_relu_bits_pitch': (_L, [_I]),
    'geeco_conv3_fwd_relu_fields': (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _P]),
    'geeco_conv3x3_dgrad_relu_fields': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P]),
    'geeco_relu_fields_elems': (_L, [_I, _I, _I]),
    'geeco_conv2_fwd_relu_fields': (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _P]),
    'geeco_conv3_dgrad_relu_fields': (_I, [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I, _P, _I]),
    'geeco_relu_bits_rows': (_L, [_I]),
    'geeco_conv1_fwd_relu_bits': (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _P]),
    'geeco_conv1_fwd_relu_bits_rgb': (_I, [_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _P]),
    'geeco_conv2_dgrad_conv1_wgrad_bits': (_I, [_P, _P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _P, _P,
                                                POINTER(SlabReduce), _I]),
    'geeco_transpose_hwio': (_I, [_P, _P, _I, _L, _L, _I, _I, _P]),
    'geeco_derive_conv_weights': (_I, [_I, _PP, _PP, POINTER(_I), POINTER(_I), POINTER(_L), _I, _L, _P, _P, _I, _I, _I, _L,
                                       _P]),
    'geeco_pad_mid': (_I, [_P, _P, _L, _I, _I, _I, _P]),
    'geeco_state_concat_fwd': (_I, [_PP, POINTER(_I), _I, _I, _P, _L, _I, _P, _I, _I, _P, _L, _P]),
    'geeco_state_concat_bwd': (_I, [_P, _L, _PP, _PP, POINTER(_I), _I, _I, _I, _I, _I, _I, _F, _P]),
    'geeco_gemm_ws_bytes': (_L, [_I, _I, _I]),
    'geeco_gemm_f32': (_I, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P, _P]),
    'geeco_lstm_gates_fwd': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'geeco_lstm_input_step_fwd': (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    'geeco_lstm_gates_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'geeco_lstm_step_bwd_ws_bytes': (_L, [_I, _I, _I]),
    'geeco_lstm_step_bwd': (_I, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _PP, _PP, POINTER(_I), _I, _I, _I, _I, _P, _P, _P]),
    'geeco_colsum': (_I, [_P, _L, _I, _I, _P, _I, _P]),
    'geeco_heads_ws_bytes': (_L, [_I, _I, _I]),
    'geeco_heads_loss_fwd_bwd': (_I, [_P, _P, _P, _I, _PP, _PP, POINTER(_I), POINTER(_I), POINTER(_F), _PP,
                                      POINTER(_L), _F, _I, _I, _I, _P, _P, _I, _P, _P, _P, _PP, _PP, _P, _P]),
    'geeco_lstm_step_heads_fwd_bwd': (_I, [_P, _L, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _PP, _PP, POINTER(_I),
                                           POINTER(_I), POINTER(_F), _PP, POINTER(_L), _F, _I, _P, _P, _I, _P, _P, _P, _PP, _PP,
                                           _P, _P, _P]),
    'geeco_adam_prepare': (_I, [_P, _F, _F, _F, _P, _P]),
    'geeco_adam_tf': (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P]),
    'geeco_adam_tf_segments': (_I, [_P, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _F, _P]),
    'geeco_sumsq': (_I, [_P, _L, _P, _P]),
}

_lib = None


def load():
  """Loads the shared library once and types every declared symbol. Raises if anything is missing."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise GeecoNativeError(
        "HIP library not built: %s is missing. Run geeco_amd/csrc/build.sh (hipcc, gfx950). "
        "geeco_amd has no CPU fallback." % LIB_PATH)
  try:
    lib = ctypes.CDLL(LIB_PATH)
  except OSError as e:
    raise GeecoNativeError("cannot load %s: %s" % (LIB_PATH, e))
  # the version first: a stale build (or another one picked through GEECO_LIB) must be named as such, not fail later on a
  # symbol or - worse - load with an entry point whose calling convention has changed
  try:
    lib.geeco_abi_version.restype = c_int
    have = int(lib.geeco_abi_version())
  except AttributeError:
    raise GeecoNativeError("%s is not a geeco_hip library (no geeco_abi_version)" % LIB_PATH)
  if have != ABI_VERSION:
    raise GeecoNativeError("ABI version mismatch: %s is version %d, this binding needs %d (rebuild: geeco_amd/csrc/build.sh)"
                           % (LIB_PATH, have, ABI_VERSION))
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError:
      raise GeecoNativeError("%s does not export %s (stale build?)" % (LIB_PATH, name))
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


def check(rc: int, what: str):
  if rc != 0:
    msg = load().geeco_last_error()
    raise GeecoNativeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ''))
