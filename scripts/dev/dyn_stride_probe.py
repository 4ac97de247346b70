"""Does the frame stride of the window tensor matter to the one-pass input stage?  A thread's 16 frame streams sit 768 KiB apart in the
dense [N][K][H][W][3] layout (a multiple of 256 KiB); the entry point takes sample / frame strides, so padded layouts can be timed without
touching the kernel.  usage: python scripts/dev/dyn_stride_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from geeco_amd import ops
N, K, H, W = 32, 16, 256, 256
HW = H * W
tgt = torch.rand(N, H, W, 3, device='cuda')
cur, buf, dif = (torch.empty(N, H, W, 4, device='cuda') for _ in range(3))
ws = ops.goal_dynimgs_ws(N, HW, 'cuda')
for pad_f, pad_s in ((0, 0), (1024, 0), (4096, 0), (16384, 0), (65536, 0), (0, 4096), (0, 65536), (1024, 1024), (3 * 4096, 5 * 1024)):
  fs = HW * 3 + pad_f
  ss = K * fs + pad_s
  fr = torch.rand(N * ss, device='cuda')
  def f(): ops.goal_dynimgs_into(cur, buf, dif, fr, tgt, K, N, HW, ws, ss, fs)
  for _ in range(5): f()
  ts = []
  for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
  ts.sort()
  print('frame stride %d floats (+%d), sample stride +%d floats: median %.1f us  min %.1f' % (fs, pad_f, pad_s, ts[2], ts[0]), flush=True)
  del fr
