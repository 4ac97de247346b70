"""Per-tile s_memtime timeline of the conv2 forward LDS-halo kernel, one wave of each K half per block
(needs the -DGEECO_STAMPS build: scripts/dev/build_stamps.sh)."""
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
enc = m.enc; l = 1; L = enc.layers[l]; G, Nf = enc.G, enc.Nf
x = enc.acts[0]; y = enc.acts[1]
def fwd(): ops.conv3x3_fwd_into(y, x, enc._w(l), enc._b(l), G, x[0].numel(), enc.gs_p, enc.gs_p, y[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], relu=True, ws=enc.fws)
for _ in range(5): fwd()
torch.cuda.synchronize()
lib = _native.load()
lib.geeco_debug_dump_halo_stamps.argtypes = [ctypes.c_char_p]
assert lib.geeco_debug_dump_halo_stamps(b'/tmp/hstamps.bin') == 0
s = np.fromfile('/tmp/hstamps.bin', dtype=np.uint64).reshape(256, 2, 64).astype(np.int64)
names = ['issue next DMA / advance', 'MFMA loop (9 steps)', 'partial sums -> LDS', 'wait vmcnt + barrier', 'epilogue']
for kh in (0, 1):
  print('K half %d (wave %d):' % (kh, 4 * kh))
  t = s[:, kh, :60].reshape(256, 10, 6)[:, 2:9]           # tiles 2..8 of every block (steady state)
  for i, nme in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i]).reshape(-1)
    print('  %-24s mean %7.0f  p10 %7.0f  p90 %7.0f' % (nme, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
  per = (t[:, 1:, 0] - t[:, :-1, 0]).reshape(-1)
  print('  %-24s mean %7.0f  p10 %7.0f  p90 %7.0f' % ('tile period', per.mean(), np.percentile(per, 10), np.percentile(per, 90)))
