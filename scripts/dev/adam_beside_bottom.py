#!/usr/bin/env python
"""Probe: can the optimiser's HBM-streaming pass hide beside the fused encoder-bottom backward?

The fused bottom (conv2 dgrad + conv1 wgrad, 454 us, MFMA-bound at ~1.1 TB/s of HBM traffic) holds two 209-VGPR waves per SIMD:
80 registers per lane stay free, enough for ONE narrow Adam wave per SIMD (scripts/dev/ub/narrow_adam.hip: 43 / 60 VGPRs).
Measured here on the bench model: part 2 of the backward (conv3 dgrad, conv2 wgrad, fused bottom, slab sums) followed by Adam,
against the same with Adam on a second stream released right before the fused bottom's launch.

  hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o scripts/dev/ub/libnarrow_adam.so scripts/dev/ub/narrow_adam.hip
  python scripts/dev/adam_beside_bottom.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402
from geeco_amd import ops         # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  lib = ctypes.CDLL(os.path.join(ROOT, 'scripts', 'dev', 'ub', 'libnarrow_adam.so'))
  lib.narrow_adam.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
  cfg, model = bench.build_model('geeco-f', 3, 16, 32, dev)
  model.store.initialize(seed=0)
  bench.synthetic_batch(model, 1234)
  model.forward(backward_too=True)
  model.backward(part='upper')
  torch.cuda.synchronize()
  enc = model.enc
  n = model.store.size
  p, g, m = (torch.randn(n, device=dev) * 0.01 for _ in range(3))
  v = torch.rand(n, device=dev) * 1e-4
  scal = torch.full((4,), 1e-4, device=dev)
  main_s = torch.cuda.current_stream()
  side = torch.cuda.Stream()

  def adam(blocks, unroll, stream):
    rc = lib.narrow_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, scal.data_ptr(), blocks, unroll, stream.cuda_stream)
    assert rc == 0, rc

  def part2(hook=None):
    pending = []
    enc.launch_dgrad(2, pending)
    enc.launch_wgrad(1, pending)
    if hook:
      hook()
    enc.launch_dgrad(1, pending)
    ops.slab_reduce_batch(pending, None)

  def timed(fn, reps=20):
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      fn()
      e1.record()
      torch.cuda.synchronize()
      ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

  t_part2 = timed(part2)
  print('part 2 alone (conv3 dgrad, conv2 wgrad, fused bottom, slab sums): %.1f us' % t_part2, flush=True)
  for blocks, unroll in ((2048, 1), (1024, 1), (512, 1), (256, 1), (1024, 2), (512, 2), (256, 2)):
    t_adam = timed(lambda: adam(blocks, unroll, main_s))

    def seq():
      part2()
      adam(blocks, unroll, main_s)

    def beside(where):
      def fn():
        ev = torch.cuda.Event()

        def hook():
          ev.record(main_s)
          side.wait_event(ev)
          adam(blocks, unroll, side)
        if where == 'bottom':
          part2(hook)
        else:      # released at the start of part 2
          hook()
          part2()
        main_s.wait_stream(side)
      return fn
    t_seq, t_b, t_s = timed(seq), timed(beside('bottom')), timed(beside('start'))
    print('adam %4d blocks x 256, unroll %d: alone %.1f us (%.2f TB/s); part 2 then adam %.1f; adam released before the fused bottom %.1f '
          '(%+.1f vs part 2 alone); released at the start of part 2 %.1f' % (blocks, unroll, t_adam, 28.0 * n / t_adam / 1e6, t_seq, t_b, t_b - t_part2, t_s),
          flush=True)


if __name__ == '__main__':
  main()
