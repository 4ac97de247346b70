"""Phase timeline (s_memtime) of the LDS heads/loss kernel; needs the -DGEECO_STAMPS build."""
import os, sys
os.environ.setdefault('GEECO_DEV', '1'); os.environ.setdefault('GEECO_LIB', 'libgeeco_hip_stamps.so')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from geeco_amd import graph
from geeco_amd.params import create_e2evmc_config
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
for _ in range(3): m.train_step()
torch.cuda.synchronize()
dec = m.decoder
N, F = 32, cfg.dim_h_fc
from geeco_amd import ops
def show(tag):
  torch.cuda.synchronize()
  st = dec.heads_ws[2 * N * F:2 * N * F + 24].view(torch.int64).cpu().numpy()
  names = ['copy-in (all inputs)', '-', 'gemm fc1', 'relu pass + gemm preds', 'losses + gemm d(a1)', 'mask pass + gemm d(heads)', 'scatter + gemm d(h)', 'gemm d(fc1 kernel)']
  print(tag)
  for i, n in enumerate(names): print('  %-28s %8d cycles' % (n, st[i + 1] - st[i]))
  print('  total %d cycles' % (st[10] - st[0]))
show('cold (inside a training step)')
def heads():
  names = [h[0] for h in dec.heads]
  T, H = dec.T, dec.H
  kw = dict(dh=dec.dh, d_fc1_w=dec._g('fc1/kernel'), d_fc1_b=dec._g('fc1/bias'),
            d_heads_w=[dec._g(n + '/kernel') for n in names], d_heads_b=[dec._g(n + '/bias') for n in names])
  ops.heads_loss_into(dec.preds, dec.losses, dec.h[T - 1], dec._v('fc1/kernel'), dec._v('fc1/bias'),
                      [dec._v(n + '/kernel') for n in names], [dec._v(n + '/bias') for n in names],
                      [h[2] for h in dec.heads], [h[3] for h in dec.heads], [h[4] for h in dec.heads],
                      dec.targets, dec.target_strides, float(dec.loss_scale), N, H, F, dec.heads_ws, **kw)
for _ in range(3): heads()
show('warm (third back-to-back launch)')
sys.exit(0)
names = ['copy-in (all inputs)', '-', 'gemm fc1', 'relu pass + gemm preds', 'losses + gemm d(a1)', 'mask pass + gemm d(heads)', 'scatter + gemm d(h)', 'gemm d(fc1 kernel)', '-', 'tail']
for i, n in enumerate(names): print('%-28s %8d cycles' % (n, st[i + 1] - st[i]))
print('total %d cycles' % (st[10] - st[0]))
