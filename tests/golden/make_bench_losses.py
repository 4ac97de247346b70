#!/usr/bin/env python
"""Generates tests/golden/bench_losses.json (run on the GPU box: ``python tests/golden/make_bench_losses.py OUT.json``).

For each bench workload: the synthetic batch bench.py generates on the device (seed 1234) is copied to the host and
the loss of the FIRST optimiser step (seed-0 glorot weights) is computed by the fp64 CPU oracle
(``oracle.loss_and_grads_chunked``) -> ``first_step_loss_oracle_fp64``.  bench.py asserts its own first-step loss
against that value (1e-4 relative).  ``final_loss_hip`` records the HIP path's loss after the total number of steps
of the usual bench invocations (regression guard only, it is the product's own number).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench                                   # noqa: E402
from geeco_amd import graph                    # noqa: E402
from geeco_amd.params import create_e2evmc_config   # noqa: E402
from geeco_amd.runtime import TrainStepRunner  # noqa: E402
from oracle import geeco_oracle as O           # noqa: E402

# (model, channels, batch, seq_len): bench.py's default = BASELINE configs[1]; the per-GPU shapes of configs[3] and configs[4]
WORKLOADS = [('geeco-f', 3, 32, 16), ('e2e_vmc', 3, 64, 16), ('geeco-f', 4, 32, 32)]
STEP_COUNTS = [3 + 5 + 20, 3 + 20 + 100]        # driver invocation (--steps 20 --warmup 5) and the default flags


def main(out_path):
  dev = torch.device('cuda', 0)
  res = {}
  for name, C, N, K in WORKLOADS:
    goal = name == 'geeco-f'
    kw = dict(window_size=K, img_channels=C, batch_size=N)
    if goal:
      kw.update(proc_obs='dynimg', proc_tgt='dyndiff')
    cfg = create_e2evmc_config(kw)
    model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(cfg, N, dev, training=True)
    model.store.initialize(seed=0)
    bench.synthetic_batch(model, 1234)
    torch.cuda.synchronize()
    P = model.store.to_numpy('params')
    feats = {k: v.cpu().numpy() for k, v in model.inputs.items() if k not in model.label_keys}
    labels = {k: model.inputs[k].cpu().numpy() for k in model.label_keys}
    tr = O.OracleTrainer(O.make_config(**kw), goal, P, dtype=torch.float64)
    print('oracle (fp64, chunked) for %s c%d b%d k%d ...' % (name, C, N, K), flush=True)
    loss_ref = float(O.loss_and_grads_chunked(tr, feats, labels, chunk=16)[0])
    runner = TrainStepRunner(model, use_graph=True, warmup=2)
    runner.step()
    torch.cuda.synchronize()
    first = float(model.loss)
    finals = {}
    for total in STEP_COUNTS:
      while runner._calls < total:
        runner.step()
      torch.cuda.synchronize()
      finals[str(total)] = round(float(model.loss), 6)
    key = '%s c%d b%d k%d' % (name, C, N, K)
    res[key] = {'first_step_loss_oracle_fp64': round(loss_ref, 8), 'first_step_loss_hip': round(first, 8),
                'final_loss_hip': finals}
    print(key, res[key], flush=True)
    assert abs(first - loss_ref) <= 1e-4 * abs(loss_ref), (first, loss_ref)
  with open(out_path, 'w') as f:
    json.dump(res, f, indent=1, sort_keys=True)


if __name__ == '__main__':
  main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'bench_losses.json'))
