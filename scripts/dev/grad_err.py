import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import geeco_oracle as O
import test_model_gpu as T
name, cfg_kw, goal, N, H = T.CASES[int(sys.argv[1])]
dev = torch.device('cuda:0')
ocfg, P, feats, labels = T._mk(cfg_kw, goal, N, H)
model = T._build(ocfg, goal, P, feats, labels, dev)
o64 = O.OracleTrainer(ocfg, goal, P, dtype=torch.float64)
o32 = O.OracleTrainer(ocfg, goal, P, dtype=torch.float32)
l64, _, g64, _, _ = o64.loss_and_grads(feats, labels)
l32, _, g32, _, _ = o32.loss_and_grads(feats, labels)
model.forward(backward_too=True); model.backward(); torch.cuda.synchronize()
g = model.store.to_numpy('grads')
print('loss hip %.8f o32 %.8f o64 %.8f' % (float(model.loss), float(l32), float(l64)))
for k in g64:
    e_hip = T._rel_max(g[k], g64[k].numpy()); e_32 = T._rel_max(g32[k].numpy(), g64[k].numpy())
    flag = ' <<<' if e_hip > 5e-5 else ''
    print('%-45s hip %.2e  cpu-fp32 %.2e  max|g| %.3e%s' % (k, e_hip, e_32, np.abs(g64[k].numpy()).max(), flag))
