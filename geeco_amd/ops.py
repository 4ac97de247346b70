"""Thin tensor-level wrappers over the C-ABI (geeco_amd/_native.py).

PyTorch is plumbing only: tensors own device memory, ``torch.cuda.current_stream()`` provides the
HIP stream.  Every function enqueues HIP kernels from libgeeco_hip.so on that stream.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _native
from ._native import check


def _lib():
  return _native.load()


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
  if t is None:
    return None
  assert t.is_cuda and t.dtype in (torch.float32, torch.int64, torch.int32, torch.int16, torch.uint8), (t.device, t.dtype)
  return ctypes.c_void_p(t.data_ptr())


def _parr(ts):
  arr = (ctypes.c_void_p * len(ts))()
  for i, t in enumerate(ts):
    arr[i] = t.data_ptr() if t is not None else None
  return arr


def _iarr(vals):
  arr = (ctypes.c_int * len(vals))()
  for i, v in enumerate(vals):
    arr[i] = int(v)
  return arr


def same_out(size: int, stride: int) -> int:
  return -(-size // stride)


# --------------------------------------------------------------------------------------------
# dynamic image
# --------------------------------------------------------------------------------------------
def dynimg_alpha(K: int):
  buf = (ctypes.c_float * K)()
  _lib().geeco_dynimg_alpha(K, ctypes.cast(buf, ctypes.c_void_p))
  return [float(v) for v in buf]


_ALPHA_CACHE = {}


def _alpha_buf(K):
  if K not in _ALPHA_CACHE:
    buf = (ctypes.c_float * K)()
    _lib().geeco_dynimg_alpha(K, ctypes.cast(buf, ctypes.c_void_p))
    _ALPHA_CACHE[K] = buf
  return _ALPHA_CACHE[K]


def dynimg_ws(N, hwc, device):
  nbytes = _lib().geeco_dynimg_ws_bytes(N, hwc)
  return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def dynimg_into(out, frames, K, N, HW, C, Cpad, ws, sample_stride, frame_stride, frames2=None):
  """out [N][HW][Cpad] <- normalised dynamic image of K frames (graph.py:30-55)."""
  check(_lib().geeco_dynimg_fwd(_p(frames), _p(frames2), sample_stride, frame_stride,
                                ctypes.cast(_alpha_buf(K), ctypes.c_void_p), N, K, HW, C, Cpad, _p(out), _p(ws),
                                _stream()), 'geeco_dynimg_fwd')


def dynimg_rgbd_into(out, rgb, depth, K, N, HW, ws, sample_stride, frame_stride, dsample_stride, dframe_stride, rgb2=None,
                     depth2=None):
  """out [N][HW][4] <- normalised dynamic image of K RGB-D frames whose rgb / depth live in separate tensors."""
  check(_lib().geeco_dynimg_rgbd_fwd(_p(rgb), _p(rgb2), sample_stride, frame_stride, _p(depth), _p(depth2), dsample_stride,
                                     dframe_stride, ctypes.cast(_alpha_buf(K), ctypes.c_void_p), N, K, HW, _p(out), _p(ws),
                                     _stream()), 'geeco_dynimg_rgbd_fwd')


def goal_dynimgs_ws(N, HW, device):
  """The workspace of goal_dynimgs_into / goal_dynimgs_u8_into (per-sample arrival counters + per-block min / max slots):
  zero-filled here ONCE; every call finds and leaves the counters zero."""
  return torch.zeros(max(int(_lib().geeco_goal_dynimgs_ws_bytes(N, HW)) // 4, 1), dtype=torch.int32, device=device)


def goal_dynimgs_into(cur_out, buf_out, diff_out, rgb, tgt_rgb, K, N, HW, ws, sample_stride, frame_stride, depth=None,
                      tgt_depth=None, dsample_stride=0, dframe_stride=0):
  """The goal model's three conv1 inputs (current frame padded, buffer image, diff image) in ONE launch and one pass over the
  window (both images normalised in registers before their only store).  ``ws`` = goal_dynimgs_ws(N, HW, device)."""
  check(_lib().geeco_goal_dynimgs_fwd(_p(rgb), sample_stride, frame_stride, _p(tgt_rgb), _p(depth), dsample_stride,
                                      dframe_stride, _p(tgt_depth), ctypes.cast(_alpha_buf(K), ctypes.c_void_p),
                                      ctypes.cast(_alpha_buf(2), ctypes.c_void_p), N, K, HW, _p(cur_out), _p(buf_out),
                                      _p(diff_out), _p(ws), _stream()), 'geeco_goal_dynimgs_fwd')


def goal_dynimgs_u8_into(cur_out, buf_out, diff_out, win_ptrs, tgt_ptrs, K, N, HW, ws, depth=None, tgt_depth=None,
                         dsample_stride=0, dframe_stride=0):
  """goal_dynimgs_into fed from resident uint8 frames: win_ptrs / tgt_ptrs are int64 DEVICE tensors of N addresses (window n =
  K consecutive [HW][3] uint8 frames; its target frame).  Bitwise the images of gather_windows_into(divisor 255) +
  goal_dynimgs_into, without the fp32 window tensor in between."""
  for t in (win_ptrs, tgt_ptrs):
    if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and t.numel() == N and t.device == buf_out.device):
      raise ValueError('goal_dynimgs_u8: address tables must be contiguous int64 tensors of N=%d entries on %s' % (N, buf_out.device))
  check(_lib().geeco_goal_dynimgs_u8_fwd(ctypes.c_void_p(win_ptrs.data_ptr()), ctypes.c_void_p(tgt_ptrs.data_ptr()), _p(depth),
                                         dsample_stride, dframe_stride, _p(tgt_depth), ctypes.cast(_alpha_buf(K), ctypes.c_void_p),
                                         ctypes.cast(_alpha_buf(2), ctypes.c_void_p), N, K, HW, _p(cur_out), _p(buf_out),
                                         _p(diff_out), _p(ws), _stream()), 'geeco_goal_dynimgs_u8_fwd')


def goal_dynimgs_timeouts(ws, N) -> int:
  """Blocks of the one-pass input stage whose wait for their sample's blocks EVER expired on this workspace (their images are
  NaN).  Synchronises the current stream: for the places where the host reads the loss anyway."""
  n = ctypes.c_int64(-1)
  check(_lib().geeco_goal_dynimgs_timeouts(_p(ws), N, _stream(), ctypes.cast(ctypes.byref(n), ctypes.c_void_p)), 'geeco_goal_dynimgs_timeouts')
  return int(n.value)


def check_input_stage(ws, N):
  """Raises when the one-pass input stage reported a timeout on ``ws`` (include/geeco_hip.h: geeco_goal_dynimgs_timeouts)."""
  n = goal_dynimgs_timeouts(ws, N)
  if n:
    raise RuntimeError('geeco_amd: %d block(s) of the one-pass input stage gave up waiting for the other blocks of their sample '
                       '(the device did not start the blocks of a launch in index order: CU masking, a partitioned device, a '
                       'co-resident persistent kernel?).  The images of those samples were written as NaN (conv1\'s ReLU turns '
                       'them into zeros: the losses of those steps are finite and WRONG); zero-fill the workspace (ops.goal_dynimgs_ws) before using it again.' % n)


def dynimg(frames: torch.Tensor, Cpad=None) -> torch.Tensor:
  """frames [N,K,H,W,C] contiguous -> [N,H,W,Cpad]."""
  N, K, H, W, C = frames.shape
  Cpad = Cpad or C
  out = torch.empty(N, H, W, Cpad, dtype=torch.float32, device=frames.device)
  ws = dynimg_ws(N, H * W * C, frames.device)
  dynimg_into(out, frames.contiguous(), K, N, H * W, C, Cpad, ws, K * H * W * C, H * W * C)
  return out


def pack_pixels_into(dst, src, src_sample_stride, N, HW, C1, Cpad, src2=None, src2_sample_stride=0, C2=0):
  check(_lib().geeco_pack_pixels(_p(src), src_sample_stride, _p(src2), src2_sample_stride, N, HW, C1, C2, Cpad,
                                 _p(dst), _stream()), 'geeco_pack_pixels')


def gather_windows_into(out, src, starts_dev, N, K, frame_elems, divisor=1.0):
  """out[n][k] <- src[starts[n] + k] / divisor for an episode resident in HBM (uint8 or float32 frames)."""
  assert src.is_cuda and src.dtype in (torch.uint8, torch.float32) and starts_dev.dtype == torch.int32
  if not (src.device == out.device == starts_dev.device):
    raise RuntimeError('gather_windows: source %s, starts %s and output %s must be on one device' %
                       (src.device, starts_dev.device, out.device))
  check(_lib().geeco_gather_windows(ctypes.c_void_p(src.data_ptr()), 1 if src.dtype == torch.uint8 else 0,
                                    ctypes.c_void_p(starts_dev.data_ptr()), N, K, frame_elems, float(divisor), _p(out),
                                    _stream()), 'geeco_gather_windows')


# --------------------------------------------------------------------------------------------
# conv encoder
# --------------------------------------------------------------------------------------------
def conv3x3_fwd_into(y, x, w, b, G, gs_x, gs_w, gs_b, gs_y, N, H, W, Cin, Cout, stride, relu=True, ws=None):
  check(_lib().geeco_conv3x3_fwd(_p(x), _p(w), _p(b), _p(y), G, gs_x, gs_w, gs_b, gs_y, N, H, W, Cin, Cout,
                                 stride, 1 if relu else 0, _p(ws), _stream()), 'geeco_conv3x3_fwd')


def conv3x3_fwd_state_into(y, x, w, b, G, gs_x, gs_w, gs_b, gs_y, N, H, W, Cin, Cout, stride, ws, state, state_stride, feat_off,
                           Ctot, jnt, jnt_stride, jnt_off, J):
  """The top layer's forward with the one-step decoder's state concat in the split-K epilogue (``geeco_conv3x3_fwd_state``).
  Returns False (nothing launched) when the shape has no such epilogue: run conv3x3_fwd_into + state_concat_fwd_into."""
  rc = _lib().geeco_conv3x3_fwd_state(_p(x), _p(w), _p(b), _p(y), G, gs_x, gs_w, gs_b, gs_y, N, H, W, Cin, Cout, stride, _p(ws),
                                      _iarr(feat_off), Ctot, _p(jnt), jnt_stride, jnt_off, J, _p(state), state_stride, _stream())
  if rc == _native.GEECO_ENOSUP:
    return False
  check(rc, 'geeco_conv3x3_fwd_state')
  return True


def conv3x3_fwd_ws_bytes(G, N, H, W, Cin, Cout, stride):
  return int(_lib().geeco_conv3x3_fwd_ws_bytes(G, N, H, W, Cin, Cout, stride))


def conv3x3_dgrad_into(dx, dz, wt, ymask, G, gs_dz, gs_wt, gs_dx, N, H, W, Cin, Cout, stride, ws=None, w=None, gs_w=0):
  check(_lib().geeco_conv3x3_dgrad(_p(dz), _p(w), _p(wt), _p(ymask), _p(dx), G, gs_dz, gs_w, gs_wt, gs_dx, N, H, W,
                                   Cin, Cout, stride, _p(ws), _stream()), 'geeco_conv3x3_dgrad')


def conv3x3_dgrad_ws_bytes(G, N, H, W, Cin, Cout, stride):
  return int(_lib().geeco_conv3x3_dgrad_ws_bytes(G, N, H, W, Cin, Cout, stride))


def _ws(nbytes, device):
  return torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=device) if nbytes > 0 else None


def conv3x3_dgrad_needs_wt(H, W, Cin, Cout, stride):
  return bool(_lib().geeco_conv3x3_dgrad_needs_wt(H, W, Cin, Cout, stride))


def conv3x3_dgrad_relu_fields_supported(H, W, Cin, Cout, stride):
  return bool(_lib().geeco_conv3x3_dgrad_relu_fields_supported(H, W, Cin, Cout, stride))


def conv3x3_wgrad_ws_bytes(G, N, H, W, Cin, Cout, stride):
  return int(_lib().geeco_conv3x3_wgrad_ws_bytes(G, N, H, W, Cin, Cout, stride))


def conv3x3_wgrad_into(dw, db, x, dz, G, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, stride, ws, pending=None,
                       reserved_cus=0):
  """``pending`` (a list): the kernel's final slab sum is not launched but appended to it (``slab_reduce_batch``
  finishes all of them in one launch; ``ws`` must stay untouched until then).  ``reserved_cus`` (data parallel; needs
  ``pending``): CUs conv2's persistent filter-gradient kernel leaves to a collective running beside it."""
  if pending is None:
    if reserved_cus:
      raise ValueError('reserved_cus is an argument of the deferred-slab-sum form: pass pending=[...]')
    check(_lib().geeco_conv3x3_wgrad(_p(x), _p(dz), _p(dw), _p(db), G, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin,
                                     Cout, stride, _p(ws), _stream()), 'geeco_conv3x3_wgrad')
    return
  item = _native.SlabReduce()
  check(_lib().geeco_conv3x3_wgrad_partial(_p(x), _p(dz), _p(dw), _p(db), G, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin,
                                           Cout, stride, _p(ws), _stream(), ctypes.byref(item), int(reserved_cus)),
        'geeco_conv3x3_wgrad_partial')
  if item.S > 0:
    pending.append(item)


def conv3x3_wgrad_pair_into(a, b, G, stride, pending=None):
  """Two independent filter gradients in ONE launch (``geeco_conv3x3_wgrad_pair``).  ``a`` / ``b``: dicts with the arguments of
  ``conv3x3_wgrad_into`` (dw, db, x, dz, gs_x, gs_dz, gs_dw, gs_db, N, H, W, Cin, Cout, ws); ``a`` = the longer problem.
  Returns False (nothing launched) when the shapes are outside the paired kernel: the caller launches them one by one."""
  lib = _lib()
  items = (_native.SlabReduce * 2)() if pending is not None else None
  rc = lib.geeco_conv3x3_wgrad_pair(
      _p(a['x']), _p(a['dz']), _p(a['dw']), _p(a['db']), a['gs_x'], a['gs_dz'], a['gs_dw'], a['gs_db'], a['N'], a['H'], a['W'],
      a['Cin'], a['Cout'], _p(a['ws']),
      _p(b['x']), _p(b['dz']), _p(b['dw']), _p(b['db']), b['gs_x'], b['gs_dz'], b['gs_dw'], b['gs_db'], b['N'], b['H'], b['W'],
      b['Cin'], b['Cout'], _p(b['ws']), G, stride, _stream(), items)
  if rc == _native.GEECO_ENOSUP:
    return False
  check(rc, 'geeco_conv3x3_wgrad_pair')
  if items is not None:
    for it in items:
      if it.S > 0:
        c = _native.SlabReduce()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(it), ctypes.sizeof(c))
        pending.append(c)
  return True


def conv_top_bwd_into(d, a, b, G, stride, pending=None):
  """conv7's input gradient + conv7's / conv8's filter gradients as ONE heterogeneous grid (``geeco_conv_top_bwd``).  ``d``: dict
  with the arguments of ``conv3x3_dgrad_into`` (dx, dz, wt, ymask, w, gs_dz, gs_w, gs_wt, gs_dx, N, H, W, Cin, Cout, ws);
  ``a`` / ``b`` as for ``conv3x3_wgrad_pair_into``.  Returns False (nothing launched) outside the combined kernels."""
  lib = _lib()
  items = (_native.SlabReduce * 2)() if pending is not None else None
  none = (None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None)      # b is None: ONE filter gradient beside the input gradient
  wa = lambda q: none if q is None else (_p(q['x']), _p(q['dz']), _p(q['dw']), _p(q['db']), q['gs_x'], q['gs_dz'], q['gs_dw'],
                                         q['gs_db'], q['N'], q['H'], q['W'], q['Cin'], q['Cout'], _p(q['ws']))
  rc = lib.geeco_conv_top_bwd(_p(d['dz']), _p(d['w']), _p(d['wt']), _p(d['ymask']), _p(d['dx']), d['gs_dz'], d['gs_w'], d['gs_wt'],
                              d['gs_dx'], d['N'], d['H'], d['W'], d['Cin'], d['Cout'], _p(d['ws']), *wa(a), *wa(b), G, stride,
                              _stream(), items)
  if rc == _native.GEECO_ENOSUP:
    return False
  check(rc, 'geeco_conv_top_bwd')
  if items is not None:
    for it in items:
      if it.S > 0:
        c = _native.SlabReduce()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(it), ctypes.sizeof(c))
        pending.append(c)
  return True


def slab_reduce_batch(pending, prepare=None, beta1=0.9, beta2=0.999):
  """Finishes the deferred slab sums of ``pending`` (<= 8 per launch) and empties the list.  ``prepare`` = (global_step, lr,
  scal): the last launch also does adam_prepare's work (one block more instead of a dependent launch)."""
  MAX = 8
  chunks = [pending[i:i + MAX] for i in range(0, len(pending), MAX)] or ([[]] if prepare is not None else [])
  for j, chunk in enumerate(chunks):
    arr = (_native.SlabReduce * max(len(chunk), 1))(*chunk)
    if prepare is not None and j == len(chunks) - 1:
      step, lr, scal = prepare
      check(_lib().geeco_slab_reduce_batch_prepare(arr, len(chunk), _p(step), float(lr), beta1, beta2, _p(scal), _stream()),
            'geeco_slab_reduce_batch_prepare')
    else:
      check(_lib().geeco_slab_reduce_batch(arr, len(chunk), _stream()), 'geeco_slab_reduce_batch')
  del pending[:]


def conv2_dgrad_conv1_wgrad_ws_bytes(G):
  return int(_lib().geeco_conv2_dgrad_conv1_wgrad_ws_bytes(G))


def conv2_dgrad_conv1_wgrad_into(dw1, db1, dz2, w2, y1, x, G, gs_dz2, gs_w2, gs_y1, gs_x, gs_dw1, gs_db1, N, H, W, ws,
                                 dz1=None, real_channels=3, pending=None, reserved_cus=0):
  """Fused encoder bottom backward: conv2's input gradient + conv1's filter/bias gradient (dz1 stays on chip).
  dw1 [G][3][3][real_channels][32]; ``pending`` / ``reserved_cus`` as for ``conv3x3_wgrad_into``."""
  if pending is None:
    if reserved_cus:
      raise ValueError('reserved_cus is an argument of the deferred-slab-sum form: pass pending=[...]')
    check(_lib().geeco_conv2_dgrad_conv1_wgrad(_p(dz2), _p(w2), _p(y1), _p(x), _p(dw1), _p(db1), _p(dz1), G, gs_dz2,
                                               gs_w2, gs_y1, gs_x, gs_dw1, gs_db1, N, H, W, real_channels, _p(ws),
                                               _stream()), 'geeco_conv2_dgrad_conv1_wgrad')
    return
  item = _native.SlabReduce()
  check(_lib().geeco_conv2_dgrad_conv1_wgrad_partial(_p(dz2), _p(w2), _p(y1), _p(x), _p(dw1), _p(db1), _p(dz1), G,
                                                     gs_dz2, gs_w2, gs_y1, gs_x, gs_dw1, gs_db1, N, H, W, real_channels,
                                                     _p(ws), _stream(), ctypes.byref(item), int(reserved_cus)),
        'geeco_conv2_dgrad_conv1_wgrad_partial')
  if item.S > 0:
    pending.append(item)


def relu_bits_pitch(W):
  return int(_lib().geeco_relu_bits_pitch(W))


def relu_bits_rows(H):
  return int(_lib().geeco_relu_bits_rows(H))


def conv1_fwd_relu_bits_into(y, bits, x, w, b, G, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H, W):
  """conv1 forward (4 -> 32, stride 1, bias, ReLU) that also writes y's sign bits (int32 [G][N][Hp][Wp], zero-filled once by the caller)."""
  check(_lib().geeco_conv1_fwd_relu_bits(_p(x), _p(w), _p(b), _p(y), _p(bits), G, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H,
                                         W, _stream()), 'geeco_conv1_fwd_relu_bits')


def conv1_fwd_relu_bits_rgb_into(y, bits, x, w3, b, G, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H, W):
  """conv1_fwd_relu_bits_into reading the RGB kernel variable [G][3][3][3][32] itself (no channel-padded copy)."""
  check(_lib().geeco_conv1_fwd_relu_bits_rgb(_p(x), _p(w3), _p(b), _p(y), _p(bits), G, gs_x, gs_w, gs_b, gs_y, gs_bits, N, H,
                                             W, _stream()), 'geeco_conv1_fwd_relu_bits_rgb')


def conv2_dgrad_conv1_wgrad_bits_into(dw1, db1, dz2, w2, y1_bits, x, G, gs_dz2, gs_w2, gs_bits, gs_x, gs_dw1, gs_db1, N, H,
                                      W, ws, real_channels=3, pending=None, reserved_cus=0):
  """Fused encoder bottom backward with the ReluGrad mask given as conv1's sign bits."""
  item = _native.SlabReduce() if pending is not None else None
  check(_lib().geeco_conv2_dgrad_conv1_wgrad_bits(_p(dz2), _p(w2), _p(y1_bits), _p(x), _p(dw1), _p(db1), G, gs_dz2, gs_w2,
                                                  gs_bits, gs_x, gs_dw1, gs_db1, N, H, W, real_channels, _p(ws), _stream(),
                                                  ctypes.byref(item) if item is not None else None, int(reserved_cus)),
        'geeco_conv2_dgrad_conv1_wgrad_bits')
  if item is not None and item.S > 0:
    pending.append(item)


def relu_fields_elems(N, H, W):
  return int(_lib().geeco_relu_fields_elems(N, H, W))


def conv2_fwd_relu_fields_into(y, fields, x, w, b, G, gs_x, gs_w, gs_b, gs_y, gs_fields, N, H, W):
  """conv2 forward (32 -> 48, stride 2, bias, ReLU) that also writes y's sign fields (int16, relu_fields_elems per encoder)."""
  check(_lib().geeco_conv2_fwd_relu_fields(_p(x), _p(w), _p(b), _p(y), _p(fields), G, gs_x, gs_w, gs_b, gs_y, gs_fields, N,
                                           H, W, _stream()), 'geeco_conv2_fwd_relu_fields')


def conv3_dgrad_relu_fields_into(dx, dz, w, fields, G, gs_dz, gs_w, gs_fields, gs_dx, N, H, W, reserved_cus=0):
  """conv3 input gradient (48 -> 64, stride 2) masked by the sign fields of conv2's output; H, W = dims of dx.
  ``reserved_cus`` (data parallel): CUs this persistent kernel leaves to a collective running beside it."""
  check(_lib().geeco_conv3_dgrad_relu_fields(_p(dz), _p(w), _p(fields), _p(dx), G, gs_dz, gs_w, gs_fields, gs_dx, N, H, W,
                                             _stream(), int(reserved_cus)), 'geeco_conv3_dgrad_relu_fields')


def conv3_fwd_relu_fields_into(y, fields, x, w, b, G, gs_x, gs_w, gs_b, gs_y, gs_fields, N, H, W):
  """conv3 forward (48 -> 64, stride 2, bias, ReLU) that also writes y's byte sign fields (uint8 [G][N][H/2][W/2][8])."""
  check(_lib().geeco_conv3_fwd_relu_fields(_p(x), _p(w), _p(b), _p(y), _p(fields), G, gs_x, gs_w, gs_b, gs_y, gs_fields, N,
                                           H, W, _stream()), 'geeco_conv3_fwd_relu_fields')


def conv3x3_dgrad_relu_fields_into(dx, dz, w, fields, G, gs_dz, gs_w, gs_fields, gs_dx, N, H, W, Cin, Cout, stride):
  """LDS-staged input gradient masked by the byte sign fields of the layer below ([G][N][H][W][Cin / 8])."""
  check(_lib().geeco_conv3x3_dgrad_relu_fields(_p(dz), _p(w), _p(fields), _p(dx), G, gs_dz, gs_w, gs_fields, gs_dx, N, H, W,
                                               Cin, Cout, stride, _stream()), 'geeco_conv3x3_dgrad_relu_fields')


def transpose_hwio_into(wt, w, G, gs_w, gs_wt, Cin, Cout):
  check(_lib().geeco_transpose_hwio(_p(w), _p(wt), G, gs_w, gs_wt, Cin, Cout, _stream()), 'geeco_transpose_hwio')


def derive_conv_weights(ws, wts, cins, couts, G, gs_w, pad_src=None, pad_dst=None, pad_cin=0, pad_cin_padded=0,
                        pad_cout=0):
  """One launch: wts[i] = per-tap transpose of ws[i] (G encoders at stride gs_w) and the zero-padded conv1 kernel."""
  n = len(ws)
  larr = (ctypes.c_int64 * max(n, 1))(*[int(t[0].numel()) for t in wts])
  check(_lib().geeco_derive_conv_weights(
      n, _parr(ws) if n else None, _parr(wts) if n else None, _iarr(cins) if n else None, _iarr(couts) if n else None,
      larr, G, gs_w, _p(pad_src), _p(pad_dst), pad_cin, pad_cin_padded, pad_cout,
      int(pad_dst[0].numel()) if pad_dst is not None else 0, _stream()), 'geeco_derive_conv_weights')


def pad_mid_into(dst, src, A, B, Bd, C):
  check(_lib().geeco_pad_mid(_p(src), _p(dst), A, B, Bd, C, _stream()), 'geeco_pad_mid')


# convenience (allocating) forms used by the parity tests -------------------------------------
def conv3x3(x, w, b, stride, relu=True):
  """x [N,H,W,Cin] (Cin % 4 == 0), w [3,3,Cin,Cout], b [Cout] -> y [N,Ho,Wo,Cout]."""
  N, H, W, Cin = x.shape
  Cout = w.shape[3]
  y = torch.empty(N, same_out(H, stride), same_out(W, stride), Cout, dtype=torch.float32, device=x.device)
  ws = _ws(conv3x3_fwd_ws_bytes(1, N, H, W, Cin, Cout, stride), x.device)
  conv3x3_fwd_into(y, x.contiguous(), w.contiguous(), b.contiguous(), 1, 0, 0, 0, 0, N, H, W, Cin, Cout, stride, relu,
                   ws)
  return y


def conv3x3_dgrad(dz, w, ymask, in_hw, stride):
  """dz [N,Ho,Wo,Cout], w [3,3,Cin,Cout], ymask [N,H,W,Cin] or None -> dx [N,H,W,Cin]."""
  N, Ho, Wo, Cout = dz.shape
  Cin = w.shape[2]
  H, W = in_hw
  wt = torch.empty(3, 3, Cout, Cin, dtype=torch.float32, device=dz.device)
  transpose_hwio_into(wt, w.contiguous(), 1, 0, 0, Cin, Cout)
  dx = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dz.device)
  ws = _ws(conv3x3_dgrad_ws_bytes(1, N, H, W, Cin, Cout, stride), dz.device)
  conv3x3_dgrad_into(dx, dz.contiguous(), wt, ymask, 1, 0, 0, 0, N, H, W, Cin, Cout, stride, ws, w=w.contiguous())
  return dx


def conv3x3_wgrad(x, dz, stride):
  """x [N,H,W,Cin], dz [N,Ho,Wo,Cout] -> dw [3,3,Cin,Cout], db [Cout]."""
  N, H, W, Cin = x.shape
  Cout = dz.shape[3]
  dw = torch.empty(3, 3, Cin, Cout, dtype=torch.float32, device=x.device)
  db = torch.empty(Cout, dtype=torch.float32, device=x.device)
  ws = torch.empty(conv3x3_wgrad_ws_bytes(1, N, H, W, Cin, Cout, stride) // 4 + 4, dtype=torch.float32, device=x.device)
  conv3x3_wgrad_into(dw, db, x.contiguous(), dz.contiguous(), 1, 0, 0, 0, 0, N, H, W, Cin, Cout, stride, ws)
  return dw, db


# --------------------------------------------------------------------------------------------
# decoder pieces
# --------------------------------------------------------------------------------------------
def state_concat_fwd_into(state, feats, feat_ch, jnt_pos, jnt, jnt_stride, J, N, cells, state_stride, sub_from=None):
  check(_lib().geeco_state_concat_fwd(_parr(feats), _iarr(feat_ch), len(feats), jnt_pos, _p(jnt), jnt_stride, J,
                                      _p(sub_from), N, cells, _p(state), state_stride, _stream()),
        'geeco_state_concat_fwd')


def state_concat_bwd_into(dfeats, dstate, dstate_stride, feats_fwd, feat_ch, jnt_pos, J, N, cells, accumulate=False,
                          scale=1.0):
  check(_lib().geeco_state_concat_bwd(_p(dstate), dstate_stride, _parr(feats_fwd), _parr(dfeats), _iarr(feat_ch),
                                      len(feats_fwd), jnt_pos, J, N, cells, 1 if accumulate else 0, float(scale),
                                      _stream()), 'geeco_state_concat_bwd')


def gemm_ws_bytes(M, N, K):
  return int(_lib().geeco_gemm_ws_bytes(M, N, K))


def gemm_into(C, A, B, M, N, K, lda, ldb, ldc, ta=False, tb=False, accumulate=False, ws=None):
  check(_lib().geeco_gemm_f32(_p(A), lda, 1 if ta else 0, _p(B), ldb, 1 if tb else 0, _p(C), ldc, M, N, K,
                              1 if accumulate else 0, _p(ws), _stream()), 'geeco_gemm_f32')


def gemm(A, B, ta=False, tb=False):
  """Allocating form for tests: op(A) @ op(B) with row-major 2-D tensors."""
  M, K = (A.shape[1], A.shape[0]) if ta else A.shape
  N = B.shape[0] if tb else B.shape[1]
  C = torch.empty(M, N, dtype=torch.float32, device=A.device)
  ws = torch.empty(gemm_ws_bytes(M, N, K) // 4 + 4, dtype=torch.float32, device=A.device)
  gemm_into(C, A.contiguous(), B.contiguous(), M, N, K, A.shape[1], B.shape[1], N, ta, tb, False, ws)
  return C


def lstm_gates_fwd_into(c, h, gates, z, bias, c_prev, N, H):
  check(_lib().geeco_lstm_gates_fwd(_p(z), _p(bias), _p(c_prev), _p(c), _p(h), _p(gates), N, H, _stream()),
        'geeco_lstm_gates_fwd')


def lstm_input_step_fwd_into(z, c, h, gates, x, wx, bias, N, H, D, ldx, ldw, ws):
  """The cell's first step (zero state): gate GEMM + gate math, the split-K slab sum inside the gate kernel (two launches)."""
  check(_lib().geeco_lstm_input_step_fwd(_p(x), ldx, _p(wx), ldw, _p(bias), _p(z), _p(c), _p(h), _p(gates), N, H, D, _p(ws),
                                         _stream()), 'geeco_lstm_input_step_fwd')


def lstm_gates_bwd_into(dz, dc_prev, gates, c_prev, c, dh, dc, N, H):
  check(_lib().geeco_lstm_gates_bwd(_p(gates), _p(c_prev), _p(c), _p(dh), _p(dc), _p(dz), _p(dc_prev), N, H,
                                    _stream()), 'geeco_lstm_gates_bwd')


def lstm_step_bwd_ws_bytes(N, D, H4):
  return int(_lib().geeco_lstm_step_bwd_ws_bytes(N, D, H4))


def lstm_step_bwd_into(dwx, db, dx, x, dz, wx, N, D, H4, ldw, ws, feats_fwd=None, dfeats=None, feat_ch=None, jnt_pos=0, J=0,
                       cells=0, pending=None):
  """Two launches: dwx = x^T dz, db = colsum(dz), split-K partials of dx = dz wx^T side by side; then dx's slab sum with
  the state-concat backward (``dfeats``) in its epilogue -- and, if ``pending`` (what lstm_step_heads_into left behind), the
  heads' batch sums as extra blocks of the first grid."""
  nf = len(feats_fwd) if feats_fwd else 0
  check(_lib().geeco_lstm_step_bwd(_p(x), D, _p(dz), H4, _p(wx), ldw, _p(dwx), ldw, _p(db), _p(dx), D, N, D, H4,
                                   _parr(feats_fwd) if nf else None, _parr(dfeats) if nf else None,
                                   _iarr(feat_ch) if nf else None, nf, jnt_pos, J, cells, _p(ws),
                                   ctypes.byref(pending) if pending is not None else None, _stream()),
        'geeco_lstm_step_bwd')


def colsum_into(out, a, lda, M, N, accumulate=False):
  check(_lib().geeco_colsum(_p(a), lda, M, N, _p(out), 1 if accumulate else 0, _stream()), 'geeco_colsum')


def heads_ws_bytes(N, H, Hfc):
  return int(_lib().geeco_heads_ws_bytes(N, H, Hfc))


def heads_loss_into(preds, losses, h, fc1_w, fc1_b, heads_w, heads_b, head_size, head_kind, head_weight, targets,
                    target_stride, loss_scale, N, H, Hfc, ws, dh=None, d_fc1_w=None, d_fc1_b=None, d_heads_w=None,
                    d_heads_b=None):
  """fc1 + heads + losses (+ their gradients when ``dh`` is given); see include/geeco_hip.h."""
  backward = dh is not None
  nh = len(heads_w)
  farr = (ctypes.c_float * nh)(*[float(v) for v in head_weight])
  larr = (ctypes.c_int64 * nh)(*[int(v) for v in target_stride])
  check(_lib().geeco_heads_loss_fwd_bwd(
      _p(h), _p(fc1_w), _p(fc1_b), nh, _parr(heads_w), _parr(heads_b), _iarr(head_size), _iarr(head_kind), farr,
      _parr(targets), larr, loss_scale, N, H, Hfc, _p(preds), _p(losses), 1 if backward else 0, _p(dh),
      _p(d_fc1_w), _p(d_fc1_b), _parr(d_heads_w) if backward else None, _parr(d_heads_b) if backward else None,
      _p(ws), _stream()), 'geeco_heads_loss_fwd_bwd')


def lstm_step_heads_into(z, c, h, gates, x, wx, bias, N, H, D, ldx, ldw, gemm_ws, preds, losses, fc1_w, fc1_b, heads_w, heads_b,
                         head_size, head_kind, head_weight, targets, target_stride, loss_scale, Hfc, heads_ws, dz=None,
                         d_fc1_w=None, d_fc1_b=None, d_heads_w=None, d_heads_b=None, pending=None):
  """One-step decoder from a zero state: the gate GEMM + ONE per-sample launch (slab sum, gate math, fc1, heads, loss terms and,
  when ``dz`` is given, everything back to the gate gradients).  ``pending`` (an _native.HeadsFinish): the batch sums are left for
  lstm_step_bwd_into(pending=...); None: they run here.  Returns False (nothing launched) for shapes the per-sample kernel does
  not serve: the caller then runs lstm_input_step_fwd_into + heads_loss_into (+ lstm_gates_bwd_into)."""
  backward = dz is not None
  nh = len(heads_w)
  farr = (ctypes.c_float * nh)(*[float(v) for v in head_weight])
  larr = (ctypes.c_int64 * nh)(*[int(v) for v in target_stride])
  rc = _lib().geeco_lstm_step_heads_fwd_bwd(
      _p(x), ldx, _p(wx), ldw, _p(bias), _p(z), _p(c), _p(h), _p(gates), N, H, D, _p(gemm_ws), _p(fc1_w), _p(fc1_b), nh,
      _parr(heads_w), _parr(heads_b), _iarr(head_size), _iarr(head_kind), farr, _parr(targets), larr, loss_scale, Hfc, _p(preds),
      _p(losses), 1 if backward else 0, _p(dz), _p(d_fc1_w), _p(d_fc1_b), _parr(d_heads_w) if backward else None,
      _parr(d_heads_b) if backward else None, _p(heads_ws), ctypes.byref(pending) if pending is not None else None, _stream())
  if rc == _native.GEECO_ENOSUP:
    return False
  check(rc, 'geeco_lstm_step_heads_fwd_bwd')
  return True


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
def adam_prepare(global_step, lr, scal, beta1=0.9, beta2=0.999):
  check(_lib().geeco_adam_prepare(_p(global_step), lr, beta1, beta2, _p(scal), _stream()), 'geeco_adam_prepare')


def adam_tf(p, g, m, v, n, scal, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0):
  check(_lib().geeco_adam_tf(_p(p), _p(g), _p(m), _p(v), n, _p(scal), beta1, beta2, eps, grad_scale, l2, _stream()),
        'geeco_adam_tf')


def adam_tf_segments(p, m, v, segments, scal, g_out=None, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, l2=0.0):
  """``geeco_adam_tf_segments``: the Adam update of up to 8 pieces of the arena, ``segments`` = [(gradient tensor of the piece, offset
  in the arena, count)], offsets / counts in floats and multiples of 4; ``g_out``: the gradient arena that also receives the pieces'
  gradients (a piece whose source IS its place in that arena needs none)."""
  if not 1 <= len(segments) <= _native.ADAM_SEGMENTS_MAX:
    raise ValueError('adam_tf_segments: 1..%d pieces, got %d' % (_native.ADAM_SEGMENTS_MAX, len(segments)))
  arr = (_native.AdamSegment * len(segments))()
  for a, (g, off, n) in zip(arr, segments):
    if g.numel() < n:
      raise ValueError('adam_tf_segments: a piece of %d floats with %d gradients' % (n, g.numel()))
    a.g, a.p_off, a.count = _p(g), int(off), int(n)
  check(_lib().geeco_adam_tf_segments(_p(p), _p(g_out), _p(m), _p(v), arr, len(segments), _p(scal), beta1, beta2, eps, grad_scale, l2,
                                      _stream()), 'geeco_adam_tf_segments')


def sumsq_into(out, p, n):
  check(_lib().geeco_sumsq(_p(p), n, _p(out), _stream()), 'geeco_sumsq')


# --------------------------------------------------------------------------------------------
# diagnostics
# --------------------------------------------------------------------------------------------
def kernel_trace(fn):
  """Runs ``fn()`` and returns the names of the conv kernels its calls dispatched (in launch order)."""
  lib = _lib()
  lib.geeco_debug_kernel_trace_begin()
  try:
    fn()
  finally:
    names = lib.geeco_debug_kernel_trace_end()
  return [n for n in (names.decode() if names else '').split(';') if n]
