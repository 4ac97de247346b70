#!/usr/bin/env python
"""The single-GPU step with the optimiser beside the encoder bottom (graph._ModelBase.backward_and_apply) against the plain step:
bitwise the same parameters / Adam slots / gradients / step counter after a few steps, eager and replayed; and its step time."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from geeco_amd.runtime import TrainStepRunner  # noqa: E402


def run(beside, use_graph, steps, batch, seq):
  dev = torch.device('cuda', 0)
  cfg, model = bench.build_model('geeco-f', 3, seq, batch, dev)
  model.store.initialize(seed=0)
  bench.synthetic_batch(model, 1234)
  r = TrainStepRunner(model, use_graph=use_graph, warmup=2)
  if not beside:
    r.beside_bottom = False
  assert r.beside_bottom == beside
  losses = []
  for _ in range(steps):
    r.step()
    torch.cuda.synchronize()
    losses.append(float(model.loss))
  more = int(os.environ.get('CHECK_STEPS', '50'))
  t0 = time.perf_counter()
  for _ in range(more):
    r.step()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / more * 1e3
  s = model.store
  return [t.detach().clone() for t in (s.params, s.adam_m, s.adam_v, s.grads)], int(s.global_step.item()), losses, ms


def main():
  batch, seq = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 16
  for use_graph in ((True,) if 'graph' in sys.argv[2:] else (False, True)):
    a, sa, la, ma = run(True, use_graph, 6, batch, seq)
    b, sb, lb, mb = run(False, use_graph, 6, batch, seq)
    same = [bool(torch.equal(x, y)) for x, y in zip(a, b)]
    print('graph=%s: params/m/v/grads bitwise equal %s, step %d/%d (bitwise comparison AFTER that many steps), losses equal %s; ms/step beside %.4f plain %.4f (%+.1f us)'
          % (use_graph, same, sa, sb, la == lb, ma, mb, (ma - mb) * 1e3), flush=True)
    assert all(same) and sa == sb and la == lb


if __name__ == '__main__':
  main()
