"""Dev: the decoder tail alone (geeco_heads_loss_fwd_bwd at N = 32, H = Hfc = 128), for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from geeco_amd import ops
dev = torch.device('cuda', 0)
N, H, F = int(os.environ.get('HB_N', '32')), 128, 128
heads = [(3, 0, 1.0), (3, 1, 1.0), (3, 0, 0.5), (3, 0, 0.5)]
g = torch.Generator().manual_seed(1)
d = lambda t: t.to(dev).contiguous()
h, w1, b1 = d(torch.randn(N, H, generator=g)), d(torch.randn(H, F, generator=g) * 0.1), d(torch.randn(F, generator=g) * 0.1)
hw = [d(torch.randn(F, sz, generator=g) * 0.1) for sz, _, _ in heads]
hb = [d(torch.randn(sz, generator=g) * 0.1) for sz, _, _ in heads]
tg = [d(torch.randn(N, sz, generator=g)) if k == 0 else d(torch.randint(-1, 2, (N, 1), generator=g).float()) for sz, k, _ in heads]
OT = sum(sz for sz, _, _ in heads)
preds, losses = torch.empty(N, OT, device=dev), torch.zeros(8, device=dev)
ws = torch.empty(ops.heads_ws_bytes(N, H, F) // 4 + 4, device=dev)
dh, dw1, db1 = torch.empty(N, H, device=dev), torch.empty(H, F, device=dev), torch.empty(F, device=dev)
dhw, dhb = [torch.empty_like(t) for t in hw], [torch.empty_like(t) for t in hb]
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for i in range(20):
  big.zero_()      # something else runs in between, as in the step
  ops.heads_loss_into(preds, losses, h, w1, b1, hw, hb, [sz for sz, _, _ in heads], [k for _, k, _ in heads], [w for _, _, w in heads], tg,
                      [t.shape[1] for t in tg], 1.0, N, H, F, ws, dh=dh, d_fc1_w=dw1, d_fc1_b=db1, d_heads_w=dhw, d_heads_b=dhb)
torch.cuda.synchronize()
print('loss', float(losses[0]))
