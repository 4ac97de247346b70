"""Step runner: replays one training step as captured hipGraphs, with the data-parallel exchange.

The reference's hot loop is ``session.run(train_op)`` once per batch.  All shapes
are static, so the ~150 kernel launches of a step are captured once (HIP stream capture through
``torch.cuda.CUDAGraph``, our kernels are launched on torch's current stream) and replayed:
  graph A = forward + backward  ->  [RCCL all-reduce of the gradient arena, world > 1]  ->
  graph B = Adam.
The Adam step counter and lr_t live in device memory, so replays advance them correctly.
"""
from __future__ import annotations

import torch

from . import dist as gdist


class TrainStepRunner:

  def __init__(self, model, use_graph=True, warmup=2):
    self.model = model
    self.world = gdist.world_size()
    model.world = self.world
    self.use_graph = bool(use_graph) and torch.cuda.is_available()
    self._ga = self._gb = None
    self._warm = warmup
    self._calls = 0

  def _fwd_bwd(self):
    self.model.forward(backward_too=True)
    self.model.backward()

  def _capture(self):
    # capture on a side stream; the warm-up steps before this call already ran eagerly
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
      self._fwd_bwd()
    with torch.cuda.graph(gb):
      self.model.apply_gradients()
    self._ga, self._gb = ga, gb

  def prepare(self):
    """Untimed set-up for benchmarks: run the eager warm-up steps and capture the graphs now, so that
    no later call pays for the capture."""
    while self.use_graph and self._ga is None:
      self.step()

  def step(self):
    """One optimiser step on the batch currently in ``model.inputs``."""
    if self.use_graph and self._ga is None and self._calls >= self._warm:
      # NB capture itself does not execute the step; fall through to replay
      self._capture()
    self._calls += 1
    if self._ga is not None:
      self._ga.replay()
      gdist.allreduce_gradients(self.model.store.grads)
      self._gb.replay()
    else:
      self._fwd_bwd()
      gdist.allreduce_gradients(self.model.store.grads)
      self.model.apply_gradients()


class EvalStepRunner:
  """Forward-only replay (Estimator.evaluate / predict)."""

  def __init__(self, model, use_graph=True, warmup=1):
    self.model = model
    self.use_graph = bool(use_graph) and torch.cuda.is_available()
    self._g = None
    self._warm = warmup
    self._calls = 0

  def step(self):
    if self.use_graph and self._g is None and self._calls >= self._warm:
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        self.model.forward(backward_too=False)
      self._g = g
    self._calls += 1
    if self._g is not None:
      self._g.replay()
    else:
      self.model.forward(backward_too=False)
