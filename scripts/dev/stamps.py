"""Per-block s_memtime timeline of the gather-GEMM conv kernel (needs the -DGEECO_STAMPS build:
scripts/dev/build_stamps.sh).  usage: stamps.py L [fwd|dgrad]"""
import os, sys, ctypes
os.environ.setdefault('GEECO_DEV', '1'); os.environ.setdefault('GEECO_LIB', 'libgeeco_hip_stamps.so')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from geeco_amd import graph, ops, _native
from geeco_amd.params import create_e2evmc_config
l = int(sys.argv[1]) - 1
which = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2, batch_size=32))
m = graph.GoalE2EVMC(cfg, 32, 'cuda', training=True)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
m.train_step(); torch.cuda.synchronize()
enc = m.enc; L = enc.layers[l]; G, Nf = enc.G, enc.Nf
x = enc.acts[l - 1]; y = enc.acts[l]; dz = enc.dz[l]
def fwd(): ops.conv3x3_fwd_into(y, x, enc._w(l), enc._b(l), G, x[0].numel(), enc.gs_p, enc.gs_p, y[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], relu=True, ws=enc.fws)
def dgrad():
  wt = enc.wt[l]; dx = enc.dz[l - 1]
  ops.conv3x3_dgrad_into(dx, dz, wt, x, G, dz[0].numel(), wt[0].numel(), dx[0].numel(), Nf, L['H'], L['W'], L['Cin'], L['Cout'], L['stride'], ws=enc.dws, w=enc._w(l), gs_w=enc.gs_p)
fn = fwd if which == 'fwd' else dgrad
for _ in range(5): fn()
torch.cuda.synchronize()
lib = _native.load()
lib.geeco_debug_dump_stamps.argtypes = [ctypes.c_char_p]
path = '/tmp/stamps.bin'
assert lib.geeco_debug_dump_stamps(path.encode()) == 0
s = np.fromfile(path, dtype=np.uint64).reshape(-1, 64)
s = s[s[:, 0] != 0].astype(np.int64)
print('blocks', len(s))
def st(name, v): print('%-34s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f' % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
st('block lifetime', s[:, 61] - s[:, 0])
st('tap table + barrier', s[:, 1] - s[:, 0])
st('row split (divisions)', s[:, 2] - s[:, 1])
st('pointer set-up', s[:, 3] - s[:, 2])
st('first loads issued', s[:, 4] - s[:, 3])
st('first loads landed + LDS + barrier', s[:, 5] - s[:, 4])
nks = min(18, int(((s[0, 6:60] != 0).sum()) // 3))
iss = np.stack([s[:, 7 + 3 * k] - s[:, 6 + 3 * k] for k in range(nks)], 1)
mm = np.stack([s[:, 8 + 3 * k] - s[:, 7 + 3 * k] for k in range(nks)], 1)
sb = np.stack([s[:, 6 + 3 * (k + 1)] - s[:, 8 + 3 * k] for k in range(nks - 1)], 1)
st('per K-step: issue next loads', iss.mean(1))
st('per K-step: LDS reads + MFMA', mm.mean(1))
st('per K-step: LDS stores + barrier', sb.mean(1))
st('K loop total', s[:, 60] - s[:, 5])
st('epilogue (stores landed)', s[:, 61] - s[:, 60])
hwid = s[:, 63]
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
print('(CU id bits only identify a CU inside its XCD)')
