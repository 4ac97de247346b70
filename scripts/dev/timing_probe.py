"""How should bench.py time ONE launch?  Compares, for a few launches of the bench step: one event pair around 30
back-to-back launches (round 2), one pair per launch (host ahead of the GPU behind a busy-wait kernel / not), one pair per
R launches.  The reference is the kernel duration rocprofv3 reports for the same launches inside the step."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from geeco_amd import ops

dev = torch.device('cuda', 0)
cfg, model = bench.build_model('geeco-f', 3, 16, 32, dev)
model.store.initialize(seed=0)
bench.synthetic_batch(model, 1234)
model.train_step(); torch.cuda.synchronize()
enc = model.enc


def pairs(fn, samples, per_pair, blocker):
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(samples)]
  torch.cuda.synchronize()
  if blocker:
    torch.cuda._sleep(int(8e6))
  for a, b in evs:
    a.record()
    for _ in range(per_pair):
      fn()
    b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) / per_pair for a, b in evs)
  return ts[len(ts) // 2] * 1e3, ts[len(ts) // 10] * 1e3, ts[(9 * len(ts)) // 10] * 1e3


def empty():
  pass


print('empty pair (blocker): %.1f us' % pairs(empty, 30, 1, True)[0])
for name, fn in (('conv1 fwd', lambda: enc.launch_fwd(0)), ('conv2 fwd', lambda: enc.launch_fwd(1)), ('fused bottom', lambda: enc.launch_dgrad(1)),
                 ('conv4 fwd', lambda: enc.launch_fwd(3)), ('conv7 fwd', lambda: enc.launch_fwd(6)), ('conv8 fwd', lambda: enc.launch_fwd(7)),
                 ('adam', lambda: ops.adam_tf(model.store.params, model.store.grads, model.store.adam_m, model.store.adam_v, model.store.size, model.scal))):
  for _ in range(3):
    fn()
  old = bench.time_region(fn, 30) * 1e3
  row = ['%-13s one pair / 30: %7.1f' % (name, old)]
  for per_pair, blocker in ((1, True), (1, False), (2, True), (4, True), (8, True)):
    m, lo, hi = pairs(fn, 30, per_pair, blocker)
    row.append('| %d/pair%s %7.1f (%.1f..%.1f)' % (per_pair, '' if blocker else ' nb', m, lo, hi))
  print(' '.join(row))
