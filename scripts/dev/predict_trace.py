"""Dev: one batch-1 forward of the goal model (what GoalE2EVMCPredictor replays), eager, for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from geeco_amd import graph
from geeco_amd.params import create_e2evmc_config
goal = os.environ.get('PT_MODEL', 'goal') == 'goal'
cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=16, batch_size=1) if goal else dict(window_size=16, batch_size=1))
m = (graph.GoalE2EVMC if goal else graph.E2EVMC)(cfg, 1, 'cuda', training=False)
m.store.initialize(0)
for k in m.inputs: m.inputs[k].normal_()
for _ in range(5):
  m.forward(backward_too=False)
torch.cuda.synchronize()
