"""Dev: would two independent launches of the backward gain from sharing the chip?  Times layer l's filter gradient and input
gradient (both read dz_l only) back to back on one stream and concurrently on two streams (different kernels co-reside on a CU
when their LDS / registers allow) at the bench shape.  usage: pair_concurrency.py [layer=5]"""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from geeco_amd import graph
from geeco_amd.params import create_e2evmc_config
l = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
m.train_step(); torch.cuda.synchronize()
enc = m.enc
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=20):
  out = []
  for _ in range(reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record(); e1.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3)
  return statistics.median(out)
def seq():
  for _ in range(4):
    enc.launch_wgrad(l); enc.launch_dgrad(l)
def conc():
  cur = torch.cuda.current_stream()
  for _ in range(4):
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): enc.launch_wgrad(l)
    with torch.cuda.stream(s2): enc.launch_dgrad(l)
    cur.wait_stream(s1); cur.wait_stream(s2)
def only(fn):
  return lambda: [fn(l) for _ in range(4)]
for _ in range(3): seq(); conc()
print('layer conv%d: wgrad alone %.1f us, dgrad alone %.1f us, back to back %.1f us, on two streams %.1f us (per pair, 4 pairs per sample)' %
      (l + 1, timed(only(enc.launch_wgrad)) / 4, timed(only(enc.launch_dgrad)) / 4, timed(seq) / 4, timed(conc) / 4))
