"""Dev: the input stage alone from resident uint8 frames (geeco_goal_dynimgs_u8_fwd) against the fp32 form, N=32 K=16 256x256."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from geeco_amd import ops
dev = torch.device('cuda', 0)
N, K, H, W = 32, 16, 256, 256
HW, fe = H * W, H * W * 3
r = np.random.default_rng(0)
eps = [torch.tensor(r.integers(0, 256, size=(100, fe), dtype=np.uint8), device=dev) for _ in range(N)]
tg = [torch.tensor(r.integers(0, 256, size=(1, fe), dtype=np.uint8), device=dev) for _ in range(N)]
win = torch.tensor([e.data_ptr() + int(r.integers(0, 84)) * fe for e in eps], dtype=torch.int64, device=dev)
tpt = torch.tensor([t.data_ptr() for t in tg], dtype=torch.int64, device=dev)
rgb = torch.rand(N, K, H, W, 3, device=dev)
tgt = torch.rand(N, H, W, 3, device=dev)

ws2 = ops.goal_dynimgs_ws(N, HW, dev)
out = [torch.empty(N, H, W, 4, device=dev) for _ in range(3)]
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)      # flushes the caches between samples
def timeit(fn, reps=30):
  ts = []
  for _ in range(reps):
    big.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
  ts.sort()
  return ts[len(ts) // 2], ts[2], ts[-3]
u8 = lambda: ops.goal_dynimgs_u8_into(out[0], out[1], out[2], win, tpt, K, N, HW, ws2)
f32 = lambda: ops.goal_dynimgs_into(out[0], out[1], out[2], rgb, tgt, K, N, HW, ws2, K * fe, fe)
for _ in range(3):
  u8(); f32()
print('%s: uint8 by address %.1f us (p10 %.1f p90 %.1f) | fp32 windows %.1f us (one launch each)' %
      ((os.environ.get('GEECO_LIB', 'default'),) + timeit(u8) + timeit(f32)[:1]))
