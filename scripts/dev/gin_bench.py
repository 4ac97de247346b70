"""Time the fused goal-inputs launch vs the separate launches (N=32, K=16, 256x256 RGB)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from geeco_amd import ops
N, K, H, W, C = 32, 16, 256, 256, 3
HW = H * W
fr = torch.rand(N, K, H, W, C, device='cuda'); tg = torch.rand(N, H, W, C, device='cuda')
o = [torch.empty(N, H, W, 4, device='cuda') for _ in range(3)]
ws = ops.goal_inputs_ws(N, 'cuda'); dws = ops.dynimg_ws(N, HW * 4, 'cuda')
def fused(): ops.goal_inputs_into(o[0], o[1], o[2], fr, tg, K, N, HW, C, ws, K * HW * C, HW * C)
def sep():
  cur = fr[:, K - 1]
  ops.pack_pixels_into(o[0], cur, K * HW * C, N, HW, C, 4)
  ops.dynimg_into(o[1], fr, K, N, HW, C, 4, dws, K * HW * C, HW * C)
  ops.dynimg_into(o[2], cur, 2, N, HW, C, 4, dws, K * HW * C, 0, frames2=tg)
for name, fn in (('fused', fused), ('separate', sep)):
  for _ in range(3): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): fn()
  e1.record(); e1.synchronize()
  print('%s %s: %.1f us' % (os.environ.get('GEECO_GIN_NOWAIT', ''), name, e0.elapsed_time(e1) / 20 * 1e3))
