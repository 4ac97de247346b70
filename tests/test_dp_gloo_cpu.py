"""Data-parallel path with world_size = 2 on CPU (gloo): the dist helpers shard the batch, SUM
all-reduce the flat gradient arena and scale by 1/world; with equal shards this must equal the
single-process gradient of the global batch (all losses are batch means).  The compute engine in
this test is the CPU oracle (no GPU here); the exchange code is the product's geeco_amd.dist."""
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
  torch.set_num_threads(2)
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
  feats, labels = O.synthetic_batch(ocfg, False, 4, seed=9, H=136, W=136)
  lo, hi = gdist.shard_bounds(4)
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
  q.put((rank, g64.numpy(), float(loss), lmax, float(np.abs(P['VMC/ConvEncoder/conv3/kernel']).sum())))
  dist.destroy_process_group()


def test_dp_gradients_equal_global_batch(tmp_path):
  sys.path.insert(0, ROOT)
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.variables import VariableStore
  from oracle import geeco_oracle as O
  world, port = 2, 29500 + (os.getpid() % 2000)
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
  feats, labels = O.synthetic_batch(ocfg, False, 4, seed=9, H=136, W=136)
  loss, _, grads, _, _ = O.OracleTrainer(ocfg, False, P, dtype=torch.float64).loss_and_grads(feats, labels)
  ref = np.zeros(store.size)
  for k, g in grads.items():
    o = store.offsets[k]
    ref[o:o + g.numel()] = g.numpy().reshape(-1)
  np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=0)                  # identical on both ranks
  np.testing.assert_allclose(res[0][1], ref, rtol=1e-9, atol=1e-12)                 # == global-batch gradient
  assert abs(0.5 * (res[0][2] + res[1][2]) - float(loss)) < 1e-12                   # mean of shard means == global mean
  assert res[0][3] == res[1][3] == max(res[0][2], res[1][2])


def test_shard_bounds_errors():
  sys.path.insert(0, ROOT)
  from geeco_amd import dist as gdist
  assert gdist.shard_bounds(8, 1, 4) == (2, 4)
  with pytest.raises(ValueError):
    gdist.shard_bounds(7, 0, 2)
