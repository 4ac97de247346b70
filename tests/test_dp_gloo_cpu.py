"""Data-parallel path with world_size = 2, 4 and 8 on CPU (gloo): the dist helpers shard the batch, SUM
all-reduce the flat gradient arena and scale by 1/world; with equal shards this must equal the
single-process gradient of the global batch (all losses are batch means).  The compute engine in
this test is the CPU oracle (no GPU here); the exchange code is the product's geeco_amd.dist.
Also: the step runner's bucketed exchange at world 2 / 4 / 8 and the ragged end of an epoch at world 4 (ranks without a
window take part through ``null_step``; unequal window counts are weighted by n_local * world / n_global)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, q):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.set_num_threads(2 if world <= 2 else 1)
  from geeco_amd import dist as gdist
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  from oracle import geeco_oracle as O
  assert gdist.init_from_env('gloo') == world and gdist.rank() == rank
  kw = dict(window_size=2, img_height=136, img_width=136)
  ocfg = O.make_config(**kw)
  store = VariableStore(model_variable_shapes(create_e2evmc_config(kw), False), 'cpu')
  if rank == 0:
    store.initialize(seed=5)              # other ranks start from zeros and must receive rank 0's weights
  gdist.broadcast_variables(store)
  P = store.to_numpy('params')
  NB = max(4, world)
  feats, labels = O.synthetic_batch(ocfg, False, NB, seed=9, H=136, W=136)
  lo, hi = gdist.shard_bounds(NB)
  assert hi - lo == NB // world
  sl = lambda d: {k: v[lo:hi] for k, v in d.items()}
  tr = O.OracleTrainer(ocfg, False, P, dtype=torch.float64)
  loss, _, grads, _, _ = tr.loss_and_grads(sl(feats), sl(labels))
  g64 = torch.zeros(store.size, dtype=torch.float64)
  for k, g in grads.items():
    o = store.offsets[k]
    g64[o:o + g.numel()] = g.reshape(-1)
  gdist.allreduce_gradients(g64)
  g64 /= gdist.world_size()
  lmax = gdist.max_over_ranks(float(loss), 'cpu')
  assert gdist.gather_floats(10.0 + rank, 'cpu') == [10.0 + r for r in range(world)]            # bench.py: per-rank ms/step
  assert gdist.gather_strings('host gpu-%d' % rank, 'cpu') == ['host gpu-%d' % r for r in range(world)]   # ... and device identities
  q.put((rank, g64.numpy(), float(loss), lmax, float(np.abs(P['VMC/ConvEncoder/conv3/kernel']).sum())))
  dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_dp_gradients_equal_global_batch(tmp_path, world):
  sys.path.insert(0, ROOT)
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  from oracle import geeco_oracle as O
  port = 20000 + (os.getpid() % 2000) + world
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  # single-process reference on the global batch
  kw = dict(window_size=2, img_height=136, img_width=136)
  ocfg = O.make_config(**kw)
  store = VariableStore(model_variable_shapes(create_e2evmc_config(kw), False), 'cpu')
  store.initialize(seed=5)
  P = store.to_numpy('params')
  assert abs(res[1][4] - float(np.abs(P['VMC/ConvEncoder/conv3/kernel']).sum())) < 1e-3     # broadcast reached rank 1
  feats, labels = O.synthetic_batch(ocfg, False, max(4, world), seed=9, H=136, W=136)
  loss, _, grads, _, _ = O.OracleTrainer(ocfg, False, P, dtype=torch.float64).loss_and_grads(feats, labels)
  ref = np.zeros(store.size)
  for k, g in grads.items():
    o = store.offsets[k]
    ref[o:o + g.numel()] = g.numpy().reshape(-1)
  for r in range(1, world):
    np.testing.assert_allclose(res[0][1], res[r][1], rtol=0, atol=0)                # identical on every rank
  np.testing.assert_allclose(res[0][1], ref, rtol=1e-9, atol=1e-12)                 # == global-batch gradient
  assert abs(sum(r[2] for r in res) / world - float(loss)) < 1e-12                  # mean of shard means == global mean
  assert all(r[3] == max(q[2] for q in res) for r in res)


def test_shard_bounds_errors():
  sys.path.insert(0, ROOT)
  from geeco_amd import dist as gdist
  assert gdist.shard_bounds(8, 1, 4) == (2, 4)
  with pytest.raises(ValueError):
    gdist.shard_bounds(7, 0, 2)


# ----------------------------------------------------------------------------------------------------
# bucketed exchange of the step runner (early ranges beside the bottom of the backward, late staging)
# ----------------------------------------------------------------------------------------------------
class _FakeModel:
  """Stands in for a graph.* model on the CPU: 'backward' writes rank-dependent gradients, the upper part into
  the early ranges only, the bottom part into the late (conv1 / conv2) ranges only."""

  def __init__(self, store, rank, early, late, can_redirect=False, split=False):
    self.store, self.rank, self.early, self.late = store, rank, early, late
    self.world = 1
    self.applied = None
    self.staging = None
    self.pieces = []
    if can_redirect:      # like graph.ConvEncoderStack.redirect_late_gradients: the bottom part writes the staging buffer
      self.redirect_late_gradients = self._redirect
    if split:             # like graph._ModelBase.apply_gradients_of: the optimiser step in pieces (runtime: two, around the late bucket)
      self.apply_gradients_of = self._apply_of

  def _apply_of(self, segments, g_out=None, last=True):
    if self.applied is None or not self.pieces:
      self.applied = torch.full_like(self.store.grads, float('nan'))
    for g, off, n in segments:
      assert g.numel() >= n
      self.applied[off:off + n] = g[:n] / self.world
      if g_out is not None:
        g_out[off:off + n] = g[:n]
    self.pieces.append((len(segments), g_out is not None, last))

  def _redirect(self, staging, late_ranges):
    if staging is None:            # the runner ends its redirection right behind its own part 2
      self.staging = None
      return False
    assert [tuple(r) for r in late_ranges] == [tuple(r) for r in self.late]
    self.staging = staging
    return True

  def forward(self, backward_too=False):
    self.store.grads.fill_(float('nan'))            # every element must be rewritten by the two backward parts

  def backward(self, part=None, adam_prepare=False):
    g = self.store.grads
    idx = torch.arange(g.numel(), dtype=torch.float32)
    val = (self.rank + 1) * (1.0 + 0.001 * (idx % 97))
    if part in (None, 'upper'):
      for off, n in self.early:
        g[off:off + n] = val[off:off + n]
    if part in (None, 'bottom'):
      pos = 0
      for off, n in self.late:
        if self.staging is not None:       # the arena's late slots stay NaN until the runner unpacks the staging buffer
          self.staging[pos:pos + n] = val[off:off + n]
        else:
          g[off:off + n] = val[off:off + n]
        pos += n

  def apply_gradients(self):
    self.applied = self.store.grads.clone() / self.world


def _bucket_worker(rank, world, port, q, can_redirect=False, overlap=True, split=False):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.set_num_threads(1)
  from geeco_amd import dist as gdist
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner, gradient_buckets
  from geeco_amd.variables import VariableStore
  gdist.init_from_env('gloo')
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2))
  store = VariableStore(model_variable_shapes(cfg, True), 'cpu')
  early, late = gradient_buckets(store)
  model = _FakeModel(store, rank, early, late, can_redirect, split)
  runner = TrainStepRunner(model, use_graph=False, overlap=overlap)
  assert runner.world == world and model.world == world and runner.split_adam == split
  runner.step()
  if split:      # two pieces: the early bucket's variables from the arena, then the late ones from the staging buffer (the arena gets them too)
    assert model.pieces == [(len(early), False, False), (len(late), True, True)], model.pieces
    assert torch.equal(model.applied, store.grads / world)
  q.put((rank, model.applied.numpy(), early, late, runner.bucket_info()))
  dist.destroy_process_group()


@pytest.mark.parametrize('world,can_redirect,overlap,split', [(2, False, True, False), (2, True, True, False), (2, True, False, False),
                                                              (4, True, True, False), (4, False, True, False), (8, True, True, False),
                                                              (2, True, True, True), (2, False, False, True), (4, True, False, True)],
                         ids=['packed-late', 'late-in-place', 'late-in-place-serial', 'world4-late-in-place', 'world4-packed-late',
                              'world8-late-in-place', 'adam-in-two-pieces', 'adam-in-two-pieces-packed-serial', 'world4-adam-in-two-pieces-serial'])
def test_bucketed_exchange_covers_the_arena(world, can_redirect, overlap, split):
  sys.path.insert(0, ROOT)
  port = 22100 + (os.getpid() % 2000) + 3 * int(can_redirect) + int(overlap) + 10 * world + 100 * int(split)
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q, can_redirect, overlap, split)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  g0, early, late, info = res[0][1], res[0][2], res[0][3], res[0][4]
  # the ranges partition the arena
  cover = np.zeros(g0.size, np.int32)
  for off, n in early + late:
    cover[off:off + n] += 1
  assert (cover == 1).all()
  # geeco-f: conv1 + conv2 of three encoders = 3 x (3*3*3*32 + 32 + 3*3*32*48 + 48) floats
  assert info['late_bytes'] == 4 * 3 * (864 + 32 + 13824 + 48) and info['late_ranges'] == 3
  assert info['mode'] == ('overlap' if overlap else 'serial')
  if can_redirect:   # the bottom part writes the staging buffer: ONE early all-reduce over the span (late slots ride along)
    assert info['early_allreduce_calls'] == 1 and info['late_written_in_place'] and info['early_ranges'] == 3
    assert info['early_bytes_on_the_wire'] == info['early_bytes'] + info['late_bytes'] - 4 * (864 + 32 + 13824 + 48)
  else:              # a model without redirect_late_gradients: one call per early range, late gradients packed
    assert info['early_allreduce_calls'] == info['early_ranges'] == 3 and not info['late_written_in_place']
  assert info['early_bytes'] + info['late_bytes'] == 4 * g0.size
  # every element = mean over ranks of (rank + 1) * pattern = (world + 1) / 2 * pattern, identical on every rank
  idx = np.arange(g0.size, dtype=np.float32)
  want = 0.5 * (world + 1) * (1.0 + 0.001 * (idx % 97))
  np.testing.assert_allclose(g0, want, rtol=1e-6)
  for r in range(1, world):
    np.testing.assert_array_equal(res[0][1], res[r][1])


# ----------------------------------------------------------------------------------------------------
# ragged end of an epoch at world 4: dp_schedule -> loss scaling, null steps (geeco_amd/estimator.py train loop)
# ----------------------------------------------------------------------------------------------------
_SCHED = [(4, 4, 4, 4), (4, 4, 2, 0), (4, 0, 0, 0), (3, 1, 0, 0)]


class _ScheduleSource:
  """What ``pickplace_input_fn(shard=(rank, world))`` hands the Estimator: this rank's batches + every rank's counts."""

  def __init__(self, rank):
    self.dp_schedule = _SCHED
    self.rank = rank

  def __iter__(self):
    for s, counts in enumerate(_SCHED):
      if counts[self.rank]:
        first = sum(counts[:self.rank])
        yield {'step': torch.arange(first, first + counts[self.rank]) + 100 * s}, None


def _ragged_worker(rank, world, port, q):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  torch.set_num_threads(1)
  from geeco_amd import dist as gdist
  from geeco_amd.estimator import Estimator
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner, gradient_buckets
  from geeco_amd.variables import VariableStore
  gdist.init_from_env('gloo')
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=2))
  store = VariableStore(model_variable_shapes(cfg, True), 'cpu')
  early, late = gradient_buckets(store)
  model = _FakeModel(store, rank, early, late, can_redirect=True)
  runner = TrainStepRunner(model, use_graph=False, overlap=True)
  idx = torch.arange(store.size, dtype=torch.float32)
  pattern = 1.0 + 0.001 * (idx % 97)
  applied = []
  # the Estimator's own schedule walk (estimator.py: _local_batches), driven like Estimator.train drives it
  for feats, labels, n, n_global in Estimator._local_batches(Estimator, _ScheduleSource(rank), world, rank):
    if n == 0:
      runner.null_step()
    else:
      # this rank's batch-mean gradient of per-window gradients (window id) * pattern, weighted like decoder.loss_scale
      scale = n * world / float(n_global)
      model.rank = float(feats['step'].double().mean()) * scale - 1.0        # _FakeModel writes (rank + 1) * pattern
      runner.step()
    applied.append(model.applied.clone().numpy())
  q.put((rank, applied))
  dist.destroy_process_group()


def test_ragged_schedule_null_steps_world4():
  """Four ranks walk a schedule whose last steps leave ranks with fewer or NO windows: every step's update must be the mean
  over the GLOBAL batch on every rank, and the mixed ``step`` / ``null_step`` calls must pair up in the exchange (a
  mismatch hangs: the queue read below times out)."""
  sys.path.insert(0, ROOT)
  world, port = 4, 24300 + (os.getpid() % 2000)
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  n = res[0][1][0].size
  pattern = 1.0 + 0.001 * (np.arange(n, dtype=np.float32) % 97)
  for s, counts in enumerate(_SCHED):
    ids = np.arange(sum(counts)) + 100 * s
    want = ids.mean() * pattern                                 # mean over the global batch of (window id) * pattern
    for r in range(world):
      np.testing.assert_allclose(res[r][1][s], want, rtol=2e-6, err_msg='step %d rank %d' % (s, r))
      np.testing.assert_array_equal(res[r][1][s], res[0][1][s])


# ----------------------------------------------------------------------------------------------------
# bench.py's exit protocol at N > 1: ranks 1..N-1 wait on the HOST (a counter in the rendezvous store, no collective
# enqueued) until rank 0 has finished its post-processing; nobody leaves before the last rank has arrived.
# ----------------------------------------------------------------------------------------------------
def _rendezvous_worker(rank, world, port, q):
  import time
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  assert gdist.init_from_env('gloo') == world
  if rank == 0:
    time.sleep(1.5)                      # rank 0's tables
  t_arrive = time.time()
  gdist.host_rendezvous('bench_done')
  t_leave = time.time()
  gdist.host_rendezvous('second_tag')    # tags are independent counters
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, t_arrive, t_leave))


@pytest.mark.parametrize('world', [2, 4])
def test_host_rendezvous_all_ranks_leave_together(world):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 26400 + os.getpid() % 40 + world
  procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  last_arrival = max(a for _, a, _ in res)
  assert all(l >= last_arrival for _, _, l in res)                 # nobody left before the slowest rank arrived
  assert min(a for _, a, _ in res) < last_arrival - 1.0            # (the others really did wait)


# ----------------------------------------------------------------------------------------------------
# bench.py's guard for the first N > 1 run: after any number of data-parallel steps the replicas are BITWISE identical
# (dist.replicas_identical: a checksum per arena, gathered); one differing word on one rank must show.
# ----------------------------------------------------------------------------------------------------
def _replica_worker(rank, world, port, q):
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  from geeco_amd.variables import VariableStore
  assert gdist.init_from_env('gloo') == world
  st = VariableStore({'a/kernel': (3, 3, 4, 16), 'a/bias': (16,), 'b/kernel': (40, 8)}, 'cpu')
  st.initialize(seed=3)
  out = [gdist.replicas_identical(st)]
  if rank == world - 1:
    st.adam_v[17] = 1e-30                 # one word of one slot on one rank
  out.append(gdist.replicas_identical(st))
  if rank == world - 1:
    st.adam_v[17] = 0.0
    st.global_step += 1                   # a rank that took a step the others did not
  out.append(gdist.replicas_identical(st))
  if rank == world - 1:
    st.global_step -= 1
  out.append(gdist.replicas_identical(st))
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, out))


def test_replicas_identical_sees_one_differing_word():
  world = 3
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 26500 + os.getpid() % 40
  procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted(q.get(timeout=120) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(out == [True, False, False, True] for _, out in res), res      # and every rank reaches the same verdict


def test_host_rendezvous_without_a_group_returns():
  sys.path.insert(0, ROOT)
  from geeco_amd import dist as gdist
  gdist.host_rendezvous('nobody')
