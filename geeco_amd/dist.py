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
# Collectives captured into a hipGraph (runtime.DP_FORMS 'overlap' / 'serial' / 'overlap_reserve*'): no user-buffer registration at
# capture time.  With registration on, a rank that REPLAYS its captured all-reduce and a rank that issues the same all-reduce
# eagerly (ragged end of an epoch: a new-batch-size runner in its warm-up steps, null_step) would hand the communicator
# differently registered buffers for one collective.  Read by RCCL when the communicator is created; the caller's own value wins.
os.environ.setdefault('NCCL_GRAPH_REGISTER', '0')
# RCCL's collective kernels on gfx950 (ncclDevKernel_Generic_*: read out of librccl's code object, ROCm 7.2) take 248-256 VGPRs per
# wave and 37.7 KB of LDS per 256-thread workgroup, one workgroup per channel.  The kernels the early gradient bucket runs beside
# (conv3's input gradient 244 VGPRs, conv2's filter gradient 256, the fused bottom 209: two waves per SIMD, one block per CU) leave
# no SIMD with 256 free registers, so an RCCL workgroup can NEVER share a CU with one of their blocks: every channel RCCL opens for
# that all-reduce takes a whole CU away from them (or waits for one).  30 MB in the ~1 ms of part 2 needs nowhere near RCCL's default
# channel count; 16 channels are 16 CUs -- what the `*_reserve16` forms of runtime.DP_FORMS leave free (6.5 % of part 2) -- and at
# >= 5 GB/s per channel still finish inside part 2.  Reasoned, NOT measured (no multi-GPU box was available to this build): read by
# RCCL when the communicator is created; the caller's own value wins; bench.py reports the value in force (comm.rccl.max_nchannels).
os.environ.setdefault('NCCL_MAX_NCHANNELS', '16')

import torch
import torch.distributed as dist


def world_size() -> int:
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None, device_index=None, single_rank_group=False):
  """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1 (one process per GPU,
  backend 'nccl' = RCCL over xGMI on a GPU box, 'gloo' without a GPU).  Returns the world size.

  ``device_index``: the GPU of this rank (default LOCAL_RANK).  ``single_rank_group``: form the group even for
  WORLD_SIZE == 1, so that the collectives really go through the backend with one rank (tests).  Both exist for
  callers that know better than the environment (the tests' launcher puts two ranks on the one test GPU over gloo);
  nothing here reads rehearsal switches from the environment."""
  ws = int(os.environ.get('WORLD_SIZE', '1'))
  if dist.is_initialized():
    return world_size()
  if ws <= 1 and not (single_rank_group and 'RANK' in os.environ):
    return 1
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29500')
  if backend is None:
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  if torch.cuda.is_available():
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) if device_index is None else int(device_index))
  dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
  return ws


def backend() -> str:
  return dist.get_backend() if dist.is_available() and dist.is_initialized() else ''


def group_active() -> bool:
  """True when collectives go through a backend (also for a one-rank group)."""
  return dist.is_available() and dist.is_initialized()


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
  if not group_active():
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


def _coll_device(device):
  """Small host-side exchanges: RCCL needs device tensors, gloo takes host tensors."""
  return device if dist.get_backend() == 'nccl' else 'cpu'


def gather_floats(value: float, device):
  """Every rank's value, in rank order, on every rank."""
  if world_size() == 1:
    return [float(value)]
  t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
  out = [torch.zeros_like(t) for _ in range(world_size())]
  dist.all_gather(out, t)
  return [float(o.item()) for o in out]


def replicas_identical(store) -> bool:
  """True when parameters, Adam slots and step counter are BITWISE the same on every rank -- what data parallelism with a
  summed gradient and a deterministic optimiser guarantees after any number of steps, and what a broken exchange (a bucket that
  raced its producer, a rank that skipped a collective) destroys.  A 64-bit checksum per arena (sum of the raw words; order
  independent, computed on the device), gathered over the ranks; one small collective."""
  if world_size() == 1:
    return True
  sums = torch.stack([t.detach().view(torch.int32).sum(dtype=torch.int64) for t in (store.params, store.adam_m, store.adam_v)] +
                     [store.global_step.detach().reshape(-1)[0].to(torch.int64)])
  sums = sums.to(_coll_device(store.params.device))
  out = [torch.zeros_like(sums) for _ in range(world_size())]
  dist.all_gather(out, sums)
  return all(torch.equal(o, out[0]) for o in out)


def gather_strings(value: str, device, width=96):
  """Every rank's short string (a device identity), in rank order, on every rank."""
  if world_size() == 1:
    return [value]
  raw = value.encode()[:width].ljust(width, b'\0')
  t = torch.tensor(list(raw), dtype=torch.uint8, device=_coll_device(device))
  out = [torch.zeros_like(t) for _ in range(world_size())]
  dist.all_gather(out, t)
  return [bytes(o.cpu().tolist()).rstrip(b'\0').decode(errors='replace') for o in out]


def host_rendezvous(tag: str, timeout_s: float = 3600.0, poll_s: float = 0.05):
  """Every rank blocks ON THE HOST until all ranks have arrived at `tag` -- a counter in the rendezvous store the group was
  formed through (no collective is enqueued, so a rank that waits here for tens of seconds while another finishes its
  GPU-side post-processing keeps its GPU idle and RCCL's watchdog has nothing pending to time out on).  One-rank groups
  and processes without a group return at once.  Falls back to a backend barrier if the store is not reachable."""
  if world_size() == 1:
    return
  import time
  try:
    store = dist.distributed_c10d._get_default_store()
    store.add(tag, 1)
    t0 = time.monotonic()
    while int(store.add(tag, 0)) < world_size():
      if time.monotonic() - t0 > timeout_s:
        raise TimeoutError('host_rendezvous(%r): %d of %d ranks after %.0f s' % (tag, int(store.add(tag, 0)), world_size(), timeout_s))
      time.sleep(poll_s)
  except (AttributeError, RuntimeError):
    dist.barrier()


def max_over_ranks(value: float, device) -> float:
  if world_size() == 1:
    return value
  t = torch.tensor([value], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())
