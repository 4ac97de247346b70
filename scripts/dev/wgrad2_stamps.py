"""Per-tile s_memtime timeline of the conv2 filter-gradient kernel (conv_s2_halo_wgrad_kernel), waves 0 and 4
of every block (needs the -DGEECO_STAMPS build: scripts/dev/build_stamps.sh)."""
import os, sys, ctypes
os.environ.setdefault('GEECO_DEV', '1'); os.environ.setdefault('GEECO_LIB', 'libgeeco_hip_stamps.so')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from geeco_amd import graph, ops, _native
from geeco_amd.params import create_e2evmc_config
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
m.train_step(); torch.cuda.synchronize()
enc = m.enc
for _ in range(5): enc.launch_wgrad(1)
torch.cuda.synchronize()
lib = _native.load()
lib.geeco_debug_dump_halo_stamps.argtypes = [ctypes.c_char_p]
assert lib.geeco_debug_dump_halo_stamps(b'/tmp/hstamps.bin') == 0
s = np.fromfile('/tmp/hstamps.bin', dtype=np.uint64).reshape(256, 2, 64).astype(np.int64)
names = ['issue next tile DMA', 'bias sums (LDS)', 'MFMA loop (108)', 'wait vmcnt + barrier']
for kh in (0, 1):
  print('wave %d:' % (4 * kh))
  t = s[:84, kh, :60].reshape(84, 10, 6)[:, 2:9]           # encoder 0's blocks
  for i, nme in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i]).reshape(-1)
    print('  %-26s mean %7.0f  p10 %7.0f  p90 %7.0f' % (nme, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
  per = (t[:, 1:, 0] - t[:, :-1, 0]).reshape(-1)
  print('  %-26s mean %7.0f  p10 %7.0f  p90 %7.0f   (s_memtime ticks: 100 MHz -> x clock/100 for cycles)' % ('tile period', per.mean(), np.percentile(per, 10), np.percentile(per, 90)))
