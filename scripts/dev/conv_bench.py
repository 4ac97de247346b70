"""Time one conv layer's forward / dgrad / wgrad launches (G=3, N=32 like the bench). usage: conv_bench.py L [iters]"""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from geeco_amd import graph, ops
from geeco_amd.params import create_e2evmc_config
l = int(sys.argv[1]) - 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
m.train_step(); torch.cuda.synchronize()
ZERO = os.environ.get('BENCH_ZERO')   # 1: zero activations, 2: zero weights too (clock/power sensitivity)
enc = m.enc; L = enc.layers[l]; G, Nf = enc.G, enc.Nf
x = enc.x_in if l == 0 else enc.acts[l - 1]; y = enc.acts[l]
if enc.dz[0] is None: enc.dz[0] = torch.randn_like(enc.acts[0])      # fused bottom: not allocated by the model
dz = enc.dz[l]
w, gs_w = (enc.w1p, enc.w1p[0].numel()) if (l == 0 and enc.pad1) else (enc._w(l), enc.gs_p)
def fwd(): ops.conv3x3_fwd_into(y, x, w, enc._b(l), G, x[0].numel(), gs_w, enc.gs_p, y[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], relu=True, ws=enc.fws)
def wgrad():
  dw, gs_dw = (enc.dw1p, enc.dw1p[0].numel()) if (l == 0 and enc.pad1) else (enc._dw(l), enc.gs_p)
  ops.conv3x3_wgrad_into(dw, enc._db(l), x, dz, G, x[0].numel(), dz[0].numel(), gs_dw, enc.gs_p, Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], enc.ws_l[l])
def dgrad():
  wt = enc.wt[l]; dx = enc.dz[l - 1]
  ops.conv3x3_dgrad_into(dx, dz, wt, None if os.environ.get('BENCH_NOMASK') else x, G, dz[0].numel(), wt[0].numel(), dx[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], ws=enc.dws, w=enc._w(l), gs_w=enc.gs_p)
if ZERO:
  x.zero_(); dz.zero_()
  if ZERO == '2': m.store.params.zero_(); enc.refresh_derived()
def fused():
  ops.conv2_dgrad_conv1_wgrad_into(enc.dw1p, enc._db(0), enc.dz[1], enc._w(1), enc.acts[0], enc.x_in, G, enc.dz[1][0].numel(), enc.gs_p, enc.acts[0][0].numel(), enc.x_in[0].numel(), enc.dw1p[0].numel(), enc.gs_p, Nf, L['H'], L['W'], enc.fws_fused)
flop = 2.0 * G * Nf * L['Ho'] * L['Wo'] * L['Cout'] * 9 * L['Cin']
for name, fn in (('fwd', fwd), ('dgrad', dgrad if l > 0 else None), ('wgrad', wgrad), ('fused', fused if (l == 1 and getattr(enc, 'fused_bottom', False)) else None)):
  if fn is None: continue
  for _ in range(3): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); e1.synchronize()
  ms = e0.elapsed_time(e1) / iters
  print('conv%d %-5s %8.1f us  %6.1f TFLOP/s' % (l + 1, name, ms * 1e3, flop / ms / 1e9))
