"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through ``torch.distributed``.

The reference has no distributed code; the batch-of-windows shards naturally
(samples are independent, all losses are batch means), so DP is: same weights on every rank,
each rank runs the step on its own shard, ONE all-reduce(sum) of the flat gradient arena
(30.2 MB for geeco-f), Adam with grad_scale = 1/world.  backend 'nccl' is RCCL on ROCm;
'gloo' is used by the CPU tests.
"""
from __future__ import annotations

import os

# The host driver of this pool only supports dmabuf IPC: without this RCCL / cross-process device memory fails with
# `hipIpcGetMemHandle: invalid argument`.  Already exported on the GPU boxes; kept here for any other launcher.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist


def world_size() -> int:
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
  """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1."""
  ws = int(os.environ.get('WORLD_SIZE', '1'))
  force = os.environ.get('GEECO_FORCE_DIST') and 'RANK' in os.environ      # exercise the RCCL path with one rank
  if (ws <= 1 and not force) or dist.is_initialized():
    return world_size()
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29500')
  # Rehearsal on a one-GPU box: GEECO_SHARE_GPU=1 puts every rank on cuda:0 and GEECO_DIST_BACKEND=gloo replaces RCCL
  # (which refuses two ranks on one device).  Production: one rank per GPU, backend 'nccl' (= RCCL over xGMI).
  if os.environ.get('GEECO_SHARE_GPU'):
    os.environ['LOCAL_RANK'] = '0'
  backend = os.environ.get('GEECO_DIST_BACKEND') or backend
  if backend is None:
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  if torch.cuda.is_available():
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
  dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
  return ws


def broadcast_variables(store, src=0):
  """Every replica starts from rank 0's weights and optimiser state."""
  if world_size() == 1:
    return
  for t in (store.params, store.adam_m, store.adam_v, store.global_step):
    dist.broadcast(t, src=src)
  store.version += 1


def allreduce_gradients(grads: torch.Tensor):
  """SUM over ranks of the flat gradient arena, in place, on the current stream (the 1/world
  factor is folded into the Adam kernel's grad_scale)."""
  if world_size() == 1:
    return
  dist.all_reduce(grads, op=dist.ReduceOp.SUM)


class _Done:
  def wait(self):
    return True


def allreduce_async(t: torch.Tensor):
  """SUM over ranks of one gradient bucket, in place.  With RCCL the collective runs on the communicator's own
  stream, ordered behind the work already enqueued on the current stream; ``.wait()`` on the returned handle makes
  the current stream wait for it (the host does not block), so kernels enqueued in between overlap it."""
  if world_size() == 1:
    return _Done()
  return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def shard_bounds(n_items: int, r=None, w=None):
  """Contiguous equal shards; requires n_items % world == 0 so that mean-of-means == global mean."""
  r = rank() if r is None else r
  w = world_size() if w is None else w
  if n_items % w:
    raise ValueError('global batch %d is not divisible by world size %d' % (n_items, w))
  per = n_items // w
  return r * per, (r + 1) * per


def broadcast_int(value: int, device='cpu', src=0) -> int:
  """Rank `src`'s integer on every rank (e.g. the epoch's shuffle seed: all ranks must agree on the episode order)."""
  if world_size() == 1:
    return int(value)
  t = torch.tensor([int(value)], dtype=torch.int64, device=device)
  dist.broadcast(t, src=src)
  return int(t.item())


def max_over_ranks(value: float, device) -> float:
  if world_size() == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())
