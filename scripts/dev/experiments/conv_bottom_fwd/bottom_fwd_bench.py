"""Encoder bottom forward at the bench shape (3 x 32 frames of 256 x 256): the one-launch kernel (conv_bottom_fwd.hip, this
directory) against conv1 forward + conv2 forward.  Median of 20 event pairs around 5 launches each.  Needs the development
library: GEECO_DEV=1 GEECO_LIB=libgeeco_hip_dev.so (scripts/dev/build_dev_lib.sh [suffix] [-DBF_PRIO=1 ...])."""
import os, sys, statistics, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, '..', '..', '..', '..'), HERE]
from binding import launch_fwd_bottom
from geeco_amd import graph
from geeco_amd.params import create_e2evmc_config
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
m.train_step(); torch.cuda.synchronize()
enc = m.enc
QUICK = os.environ.get('BF_QUICK')      # under rocprofv3 --pmc: a handful of launches
def timed(fn, reps=20, inner=5):
  if QUICK: reps, inner = 2, 2
  for _ in range(3): fn()
  out = []
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner): fn()
    e1.record(); e1.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / inner)
  return statistics.median(out)
def separate():
  enc.launch_fwd(0); enc.launch_fwd(1)
res = {}
for rep in range(1 if QUICK else 2):
  res.setdefault('conv1 fwd', []).append(timed(lambda: enc.launch_fwd(0)))
  res.setdefault('conv2 fwd', []).append(timed(lambda: enc.launch_fwd(1)))
  res.setdefault('conv1 + conv2 (two launches)', []).append(timed(separate))
  res.setdefault('one launch', []).append(timed(lambda: launch_fwd_bottom(enc)))
flop = 2.0 * 96 * (56623104 + 226492416)
print('[%s] ' % os.environ.get('GEECO_LIB', 'default') + '; '.join('%s %s us' % (k, '/'.join('%.1f' % v for v in vs)) for k, vs in res.items())
      + '; one launch = %.1f TFLOP/s' % (flop / min(res.get('one launch', [1e9])) / 1e6))
