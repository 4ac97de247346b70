import os, sys, torch
sys.path.insert(0, '/root/repo')
from geeco_amd import ops
N, K, H, W, C = 32, 16, 256, 256, 3
HW = H * W
fr = torch.rand(N, K, H, W, C, device='cuda')
o = torch.empty(N, H, W, 4, device='cuda'); dws = ops.dynimg_ws(N, HW * 4, 'cuda')
def f(): ops.dynimg_into(o, fr, K, N, HW, C, 4, dws, K * HW * C, HW * C)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); e1.synchronize()
print(os.environ.get('GEECO_DYN_NT', '-'), 'dynimg K=16: %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
