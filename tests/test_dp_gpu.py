"""Data-parallel training step on the REAL HIP path with two ranks sharing the one GPU of the test box
(gloo transport: RCCL refuses two ranks on one device; the exchange code path -- broadcast, shard, SUM
all-reduce of the gradient arena between the two graph replays, 1/world in the Adam kernel -- is the
product's).  Two ranks x N/2 must reproduce the single-process step on the global batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4, lr=1e-3)


def _batch(n):
  sys.path.insert(0, ROOT)
  from oracle import geeco_oracle as O
  return O.synthetic_batch(O.make_config(**KW), True, n, seed=31, H=136, W=136)


def _child_watchdog(seconds=240):
  """In a spawned worker: a child that is stuck (a collective that never returns, a graph replay that never ends) writes every
  thread's Python stack to stderr and exits non-zero instead of sitting there until the parent's limit."""
  import faulthandler
  faulthandler.dump_traceback_later(seconds, exit=True)


def _results(procs, q, n, limit=300):
  """n results from the workers' queue; a worker that died (or the limit) fails the test AT ONCE with the exit codes, not after
  a silent q.get(timeout=...) -- ten silent minutes look like a hung GPU box from outside."""
  import queue
  import time
  out, t0 = [], time.time()
  while len(out) < n:
    try:
      out.append(q.get(timeout=2))
      continue
    except queue.Empty:
      pass
    dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
    if dead or time.time() - t0 > limit:
      for p in procs:
        if p.is_alive():
          p.kill()
      raise AssertionError('worker exit codes %s after %.0f s, %d of %d results (worker stderr above)' % ([p.exitcode for p in procs], time.time() - t0, len(out), n))
  return out


def _worker(rank, world, port, q):
  _child_watchdog()
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  torch.cuda.set_device(0)
  gdist.init_from_env('gloo')
  feats, labels = _batch(4)
  lo, hi = gdist.shard_bounds(4)
  model = graph.GoalE2EVMC(create_e2evmc_config(KW), hi - lo, 'cuda:0', training=True)
  if rank == 0:
    model.store.initialize(seed=9)
  gdist.broadcast_variables(model.store)
  model.load_batch({k: torch.from_numpy(v[lo:hi]) for k, v in feats.items()}, {k: torch.from_numpy(v[lo:hi]) for k, v in labels.items()})
  runner = TrainStepRunner(model, use_graph=True, warmup=1)     # step 1 eager, steps 2-3 replayed
  losses = []
  for _ in range(3):
    runner.step()
    torch.cuda.synchronize()
    losses.append(float(model.loss))
  q.put((rank, model.store.params.detach().cpu().numpy(), losses))
  torch.distributed.destroy_process_group()


def test_two_ranks_equal_single_process(dev):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 26600 + os.getpid() % 1000
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted(_results(procs, q, 2), key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  feats, labels = _batch(4)
  model = graph.GoalE2EVMC(create_e2evmc_config(KW), 4, dev, training=True)
  model.store.initialize(seed=9)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  ref_losses = []
  for _ in range(3):
    model.train_step()
    torch.cuda.synchronize()
    ref_losses.append(float(model.loss))
  ref = model.store.params.detach().cpu().numpy()
  np.testing.assert_array_equal(res[0][1], res[1][1])                               # replicas stay identical
  # lr = 1e-3, 3 Adam steps: summation-order differences between N=4 and 2 x N=2 move weights by << lr
  np.testing.assert_allclose(res[0][1], ref, rtol=0, atol=3e-4)
  frac_close = np.mean(np.abs(res[0][1] - ref) < 2e-5)
  assert frac_close > 0.99, frac_close
  # global loss = mean of the two shard losses
  for s in range(3):
    assert abs(0.5 * (res[0][2][s] + res[1][2][s]) - ref_losses[s]) < 2e-4 * abs(ref_losses[s]) + 1e-5


# ----------------------------------------------------------------------------------------------------
# capture_exchange=True on a backend whose collectives cannot be captured.  (a) The runner decides BEFORE it captures anything
# (gloo's all-reduce of a device tensor goes through the host): told once, three graphs, and the steps equal, bitwise, those of
# a runner that was never asked.  (b) Why it must decide before: a capture that FAILS cannot be undone on ROCm 7.2
# (scripts/dev/ub/capture_abort.py, profiles/r06/ub_capture_abort.txt) -- the first build of this round let gloo's all-reduce into the
# capture and tried to re-capture afterwards: `capturing stream has unjoined work`, stream stuck in capture mode.  The runner
# therefore raises CaptureFailed (with the current stream put back), which bench.py answers with the line of the safe form it
# measured first; provoked here by making the runner believe gloo is RCCL, in a process that is thrown away afterwards.
# ----------------------------------------------------------------------------------------------------
def _fallback_worker(rank, world, port, q):
  _child_watchdog()
  import warnings
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  torch.cuda.set_device(0)
  gdist.init_from_env('gloo')
  feats, labels = _batch(4)
  lo, hi = gdist.shard_bounds(4)
  out = {}
  for name, kw in (('forced', dict(capture_exchange=True)), ('plain', {})):
    model = graph.GoalE2EVMC(create_e2evmc_config(KW), hi - lo, 'cuda:0', training=True)
    model.store.initialize(seed=9)
    gdist.broadcast_variables(model.store)
    model.load_batch({k: torch.from_numpy(v[lo:hi]) for k, v in feats.items()}, {k: torch.from_numpy(v[lo:hi]) for k, v in labels.items()})
    runner = TrainStepRunner(model, use_graph=True, warmup=1, **kw)
    assert runner.capture_exchange == (name == 'forced') and runner.split_adam
    losses, caught = [], []
    for i in range(4):
      with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        runner.step()                     # step 1 eager (forced: the one-pass form); step 2 captures three graphs and replays
      caught += [str(x.message) for x in w if 'cannot be captured' in str(x.message)]
      torch.cuda.synchronize()
      losses.append(float(model.loss))
      assert not torch.cuda.is_current_stream_capturing()
    out[name] = dict(params=model.store.params.detach().cpu().numpy(), m=model.store.adam_m.detach().cpu().numpy(), losses=losses,
                     step=int(model.store.global_step.item()), graphs=len(runner._graphs), capture_exchange=runner.capture_exchange,
                     warned=len(caught), prepared=bool(getattr(model, '_prepared', False)))
  q.put((rank, out))
  torch.distributed.destroy_process_group()


def test_one_graph_form_on_a_backend_that_cannot_capture(dev):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 27600 + os.getpid() % 1000
  procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted(_results(procs, q, 2), key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for rank, out in res:
    f, pl = out['forced'], out['plain']
    assert f['warned'] == 1 and pl['warned'] == 0, (rank, f['warned'], pl['warned'])       # said so once
    assert f['graphs'] == 3 and not f['capture_exchange'] and pl['graphs'] == 3
    assert f['step'] == pl['step'] == 4 and not f['prepared']
    assert f['losses'] == pl['losses'], (rank, f['losses'], pl['losses'])
    np.testing.assert_array_equal(f['params'], pl['params'])
    np.testing.assert_array_equal(f['m'], pl['m'])
  np.testing.assert_array_equal(res[0][1]['forced']['params'], res[1][1]['forced']['params'])   # replicas identical


def _capture_failed_worker(port, q):
  _child_watchdog()
  sys.path.insert(0, ROOT)
  os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  from geeco_amd import graph
  from geeco_amd import runtime
  from geeco_amd.params import create_e2evmc_config
  torch.cuda.set_device(0)
  assert gdist.init_from_env('gloo', single_rank_group=True) == 1 and gdist.group_active()
  runtime.gdist.backend = lambda: 'nccl'          # the runner now believes the group's collectives can be captured
  feats, labels = _batch(4)
  model = graph.GoalE2EVMC(create_e2evmc_config(KW), 4, 'cuda:0', training=True)
  model.store.initialize(seed=9)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  runner = runtime.TrainStepRunner(model, use_graph=True, warmup=1, dp=True, capture_exchange=True)
  runner.step()                                   # eager
  torch.cuda.synchronize()
  step_before = int(model.store.global_step.item())
  orig = torch.cuda.current_stream()
  res = {'raised': None}
  try:
    runner.step()                                 # captures: gloo's all-reduce inside the capture fails
  except runtime.CaptureFailed as e:
    res.update(raised=str(e), stream_back=torch.cuda.current_stream() == orig, graphs=runner._graphs is None,
               prepared=bool(getattr(model, '_prepared', False)))
  q.put(res)
  q.close()
  q.join_thread()
  os._exit(0)                                     # the streams of the failed capture are unusable: leave without teardown


def test_failed_one_graph_capture_raises_capture_failed(dev):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  p = ctx.Process(target=_capture_failed_worker, args=(28600 + os.getpid() % 1000, q))
  p.start()
  res = _results([p], q, 1)[0]
  p.join(timeout=60)
  assert p.exitcode == 0
  assert res['raised'] and 'one hipGraph failed' in res['raised'] and 'three_graphs' in res['raised'], res
  assert res['stream_back'] and res['graphs'] and not res['prepared'], res


# ----------------------------------------------------------------------------------------------------
# Estimator under data parallelism with ragged global batches: 4 -> 2+2, 3 -> 2+1 (loss scales 4/3 and 2/3),
# 1 -> 1+0 (rank 1 takes a null step: zeros into the exchange, same Adam update)
# ----------------------------------------------------------------------------------------------------
def _global_batches():
  sys.path.insert(0, ROOT)
  from geeco_amd.input_fn import synthetic_batches
  full = list(synthetic_batches(4, 3, 3, (136, 136), 3, True, seed=17)())
  cut = lambda b, n: ({k: v[:n] for k, v in b[0].items()}, {k: v[:n] for k, v in b[1].items()})
  return [full[0], cut(full[1], 3), cut(full[2], 1)]


def _est_worker(rank, world, port, q, dp_form=None):
  _child_watchdog()
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from geeco_amd import dist as gdist
  from geeco_amd import estimator as est
  from geeco_amd.params import create_e2evmc_config
  torch.cuda.set_device(0)
  gdist.init_from_env('gloo')
  params = {'e2evmc_config': create_e2evmc_config(KW), 'log_steps': 100, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(init_seed=4, dp_form=dp_form), params)
  e.train(input_fn=lambda: iter(_global_batches()))
  torch.cuda.synchronize()
  q.put((rank, e._store.params.detach().cpu().numpy(), int(e._store.global_step.item())))
  torch.distributed.destroy_process_group()


@pytest.mark.parametrize('dp_form', [None, 'two_graphs'], ids=['default form', 'two_graphs'])
def test_estimator_ragged_batches_two_ranks(dev, dp_form):
  from geeco_amd import estimator as est
  from geeco_amd.params import create_e2evmc_config
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29600 + os.getpid() % 1000 + (7 if dp_form else 0)
  procs = [ctx.Process(target=_est_worker, args=(r, 2, port, q, dp_form)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted(_results(procs, q, 2), key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  params = {'e2evmc_config': create_e2evmc_config(KW), 'log_steps': 100, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(init_seed=4), params)
  e.train(input_fn=lambda: iter(_global_batches()))
  torch.cuda.synchronize()
  ref = e._store.params.detach().cpu().numpy()
  assert res[0][2] == res[1][2] == 3                                               # the null step advanced rank 1 too
  np.testing.assert_array_equal(res[0][1], res[1][1])                               # replicas stay identical
  np.testing.assert_allclose(res[0][1], ref, rtol=0, atol=3e-4)                     # == single process on the global batches
  assert np.mean(np.abs(res[0][1] - ref) < 2e-5) > 0.99


# ----------------------------------------------------------------------------------------------------
# The same ragged schedule over RCCL on TWO REAL GPUs, in every form of the step -- the test ADVICE r05 asked for (a rank that
# replays captured collectives beside a rank that issues them eagerly: warm-up steps of a new batch size, null_step).  Skipped on
# the one-GPU test box; the first multi-GPU box that runs the suite runs it.
# ----------------------------------------------------------------------------------------------------
def _est_worker_rccl(rank, world, port, dp_form, q):
  _child_watchdog()
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  from geeco_amd import dist as gdist
  from geeco_amd import estimator as est
  from geeco_amd.params import create_e2evmc_config
  gdist.init_from_env('nccl')
  params = {'e2evmc_config': create_e2evmc_config(KW), 'log_steps': 100, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(init_seed=4, dp_form=dp_form), params)
  for _ in range(2):                      # two epochs: the second one replays every captured graph, ragged steps included
    e.train(input_fn=lambda: iter(_global_batches()))
  torch.cuda.synchronize()
  q.put((rank, e._store.params.detach().cpu().numpy(), int(e._store.global_step.item())))
  torch.distributed.destroy_process_group()


# (the one-graph forms have never run with two ranks: they are tried only when asked for, so that an unattended suite on a
# multi-GPU box cannot hang in a captured collective; GEECO_TEST_ONE_GRAPH_DP=1 adds them)
_RCCL_FORMS = ['three_graphs_reserve16', 'three_graphs'] + (['overlap', 'serial', 'overlap_reserve16'] if os.environ.get('GEECO_TEST_ONE_GRAPH_DP') == '1' else [])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL refuses two ranks on one device)')
@pytest.mark.parametrize('dp_form', _RCCL_FORMS)
def test_estimator_ragged_batches_two_gpus_over_rccl(dev, dp_form):
  from geeco_amd import estimator as est
  from geeco_amd.params import create_e2evmc_config
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 30610 + os.getpid() % 1000
  procs = [ctx.Process(target=_est_worker_rccl, args=(r, 2, port, dp_form, q)) for r in range(2)]
  for p in procs:
    p.start()
  try:
    res = sorted(_results(procs, q, 2), key=lambda t: t[0])
    for p in procs:
      p.join(timeout=120)
      assert p.exitcode == 0
  finally:
    for p in procs:                       # never leave a rank behind that waits in a collective
      if p.is_alive():
        p.kill()
  params = {'e2evmc_config': create_e2evmc_config(KW), 'log_steps': 100, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(init_seed=4), params)
  for _ in range(2):
    e.train(input_fn=lambda: iter(_global_batches()))
  torch.cuda.synchronize()
  ref = e._store.params.detach().cpu().numpy()
  assert res[0][2] == res[1][2] == 6
  np.testing.assert_array_equal(res[0][1], res[1][1])                               # replicas stay identical (Estimator.train checks it too)
  np.testing.assert_allclose(res[0][1], ref, rtol=0, atol=6e-4)                     # == single process on the global batches
  assert np.mean(np.abs(res[0][1] - ref) < 4e-5) > 0.99


def test_train_script_two_ranks_on_disk_dataset(dev, tmp_path):
  """scripts/train_e2evmc.py under torch.distributed.run with TWO ranks (sharing the one test GPU over gloo through the
  tests' launcher helper tests/_dp_launch.py) on an on-disk dataset of THREE episodes: rank 0 reads two episodes, rank 1 one,
  so the epoch ends with steps in which rank 1 holds fewer or no windows (dp_schedule: loss scaling, null steps); one
  shuffle seed is broadcast; rank 0 alone writes run command, config, checkpoints (+ TF bundles) and snapshots."""
  import json
  import subprocess
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from test_host_logic_cpu import _make_dataset
  from geeco_amd.params import create_e2evmc_config
  root = str(tmp_path / 'ds')
  os.makedirs(root)
  _make_dataset(root, n_eps=3, T=9, H=136, W=136)        # 6 windows per episode (K = 3)
  md = str(tmp_path / 'run')
  os.makedirs(md)
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=3, img_height=136, img_width=136, batch_size=4))
  json.dump(cfg._asdict(), open(os.path.join(md, 'e2evmc_config.json'), 'w'))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
    env.pop(k, None)
  port = 31620 + os.getpid() % 1000
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(port), os.path.join(ROOT, 'tests', '_dp_launch.py'), os.path.join(ROOT, 'scripts', 'train_e2evmc.py'),
         '--dataset_dir', root, '--model_dir', md,
         '--goal_condition', 'target', '--proc_obs', 'dynimg', '--proc_tgt', 'dyndiff', '--window_size', '3', '--batch_size', '4',
         '--train_epochs', '2', '--log_steps', '1', '--num_best_ckpt', '2']
  out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
  assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
  # global batch 4 = 2 windows per rank per step: rank 0 has 12 windows (6 steps), rank 1 has 6 (3 steps) -> 6 steps per epoch
  from geeco_amd import estimator as est
  assert os.path.basename(est.latest_checkpoint(md)) == 'model.ckpt-12'
  assert os.path.exists(os.path.join(md, 'model.ckpt-12.index'))
  idx = json.load(open(os.path.join(md, 'snapshots', 'snapshot_index.json')))
  assert len(idx) == 2
  ev = [json.loads(l) for l in open(os.path.join(md, 'events.jsonl'))]
  assert [e['global_step'] for e in ev] == list(range(1, 13)) and all(np.isfinite(e['loss']) for e in ev)
  assert sum(1 for fn in os.listdir(md) if fn.endswith('runcmd.json')) == 1            # rank 0 only


# ----------------------------------------------------------------------------------------------------
# The data-parallel step through the REAL backend string: "nccl" (= RCCL) with ONE rank.  ONE captured graph that holds the
# exchange too (the default with RCCL) and three captured graphs with the exchange between them
# (thread_local capture mode beside RCCL's watchdog thread), the early bucket as ONE all-reduce over the arena span, the
# late bucket written in place into its staging buffer, both orders of the exchange.  RCCL refuses two ranks on one
# device, so one rank is all a one-GPU box can give; the sum over one rank is the identity, hence BITWISE equality
# with the plain one-graph step.
# ----------------------------------------------------------------------------------------------------
def _rccl_worker(port, overlap, one_graph, q):
  _child_watchdog()
  sys.path.insert(0, ROOT)
  os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  from geeco_amd import dist as gdist
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  assert gdist.init_from_env('nccl', single_rank_group=True) == 1 and gdist.group_active()
  assert torch.distributed.get_backend() == 'nccl'
  feats, labels = _batch(4)
  model = graph.GoalE2EVMC(create_e2evmc_config(KW), 4, 'cuda:0', training=True)
  model.store.initialize(seed=9)
  gdist.broadcast_variables(model.store)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  # True = the exchange captured into the step graph (opt-in: dp_form 'overlap' / 'serial'); default = three graphs
  two_graphs = one_graph == 'two'          # parts 1 and 2 as graphs, the optimiser's two pieces launched eagerly (runtime.DP_FORMS['two_graphs'])
  one_graph = one_graph is True
  runner = TrainStepRunner(model, use_graph=True, warmup=1, dp=True, overlap=overlap,
                           **({'capture_exchange': True} if one_graph else {'eager_adam': True} if two_graphs else {}))
  assert runner.capture_exchange == one_graph
  # one graph + early bucket beside part 2 over RCCL: Adam's early piece runs on a second stream beside the fused encoder bottom
  # (runtime.TrainStepRunner._dp_step_beside); the bitwise comparison with the single-GPU step below covers it
  assert runner.dp_beside_bottom == (one_graph and overlap) and not runner.beside_bottom
  info = runner.bucket_info()
  losses = []
  for i in range(5):
    runner.step()
    torch.cuda.synchronize()
    losses.append(float(model.loss))
    # the redirection of the late gradients and the reserved-CU argument are the runner's, scoped to its own part 2: no
    # model-wide state is left behind (a second runner on the same model neither inherits nor undoes anything)
    assert model.enc.late is None and model.enc.reserved_cus == 0, i
  if overlap:
    # a second runner with CUs reserved for the collective on the SAME model (bench.py's probe): its own graphs, same results
    r2 = TrainStepRunner(model, use_graph=True, warmup=1, dp=True, overlap=True, reserved_cus=16)
    before = model.store.params.clone()
    for i in range(3):
      r2.step()
    torch.cuda.synchronize()
    assert model.enc.late is None and model.enc.reserved_cus == 0 and np.isfinite(float(model.loss))
    assert not torch.equal(before, model.store.params)
    model.store.params.copy_(before)      # (the comparison below is about the first runner's five steps)
  q.put((model.store.params.detach().cpu().numpy(), losses, info, runner._graphs is not None
         and len(runner._graphs) == (1 if one_graph else 2 if two_graphs else 3) and runner.capture_exchange == one_graph))
  torch.distributed.destroy_process_group()


@pytest.mark.parametrize('overlap,one_graph', [(True, True), (False, True), (True, False), (False, False), (True, 'two')],
                         ids=['one graph overlap', 'one graph serial', 'three graphs overlap', 'three graphs serial', 'two graphs'])
def test_dp_step_over_rccl_one_rank(dev, overlap, one_graph):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  p = ctx.Process(target=_rccl_worker, args=(18900 + os.getpid() % 1000 + 2 * int(one_graph is True) + int(overlap) + 4 * int(one_graph == 'two'),
                                             overlap, one_graph, q))
  p.start()
  params, losses, info, three = _results([p], q, 1)[0]
  p.join(timeout=120)
  assert p.exitcode == 0
  assert three and info['graphs_per_step'] == (1 if one_graph is True else 2 if one_graph == 'two' else 3)
  assert info['early_allreduce_calls'] == 1 and info['late_written_in_place'] and info['late_ranges'] == 3
  assert info['mode'] == ('overlap' if overlap else 'serial')
  feats, labels = _batch(4)
  model = graph.GoalE2EVMC(create_e2evmc_config(KW), 4, dev, training=True)
  model.store.initialize(seed=9)
  model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
  runner = TrainStepRunner(model, use_graph=True, warmup=1)
  ref_losses = []
  for _ in range(5):
    runner.step()
    torch.cuda.synchronize()
    ref_losses.append(float(model.loss))
  assert runner._graphs is not None and len(runner._graphs) == 1
  assert losses == ref_losses
  np.testing.assert_array_equal(params, model.store.params.detach().cpu().numpy())


def test_bottom_backward_writes_only_the_late_bucket(dev):
  """The overlapped exchange reduces the early arena ranges in place while the bottom of the backward runs: that is
  race free only if backward(part='bottom') -- its deferred slab sums included -- writes nothing outside conv1 / conv2.
  (a) plain: the early ranges are bit-identical before and after the bottom part, and early + late ranges cover the arena
  exactly once; (b) with the late gradients redirected into the staging buffer (what the runner does) the WHOLE arena is
  untouched by the bottom part and the staging buffer holds bitwise the gradients (a) left in the arena."""
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import gradient_buckets
  feats, labels = _batch(4)
  outs = []
  for redirect in (False, True):
    model = graph.GoalE2EVMC(create_e2evmc_config(KW), 4, dev, training=True)
    model.store.initialize(seed=9)
    model.load_batch({k: torch.from_numpy(v) for k, v in feats.items()}, {k: torch.from_numpy(v) for k, v in labels.items()})
    early, late = gradient_buckets(model.store)
    cover = np.zeros(model.store.size, np.int32)
    for off, n in early + late:
      cover[off:off + n] += 1
    assert (cover == 1).all()
    staging = torch.full((sum(n for _, n in late),), float('nan'), device=dev)
    if redirect:
      assert model.redirect_late_gradients(staging, late)
    model.store.grads.fill_(float('nan'))
    model.forward(backward_too=True)
    model.backward(part='upper')
    torch.cuda.synchronize()
    before = model.store.grads.clone()
    model.backward(part='bottom')
    torch.cuda.synchronize()
    after = model.store.grads
    for off, n in early:
      assert torch.equal(before[off:off + n].view(torch.int32), after[off:off + n].view(torch.int32))
    # every VARIABLE's gradient was written by one of the two parts (never written: the alignment pads between variables,
    # and -- one LSTM step from a zero state -- the recurrent rows of the LSTM kernel, whose gradient h_prev^T dz is
    # identically zero; the arena is zero-initialised and Adam keeps those rows where they are)
    D = model.decoder.D
    for name, g in model.store.to_numpy('grads').items():
      if redirect and ('/conv1/' in name or '/conv2/' in name):
        continue
      if name.endswith('lstm_cell/kernel') and model.decoder.T == 1:
        g = g[:D]
      assert not np.isnan(g).any(), name
    if redirect:
      assert torch.equal(before.view(torch.int32), after.view(torch.int32))        # the arena was not written at all
      outs.append(staging.clone())
      # ending the redirection (what a later single-process runner on the same model does): the arena gets them again
      assert model.redirect_late_gradients(None, None) is False
      model.backward(part='bottom')
      torch.cuda.synchronize()
      assert torch.equal(torch.cat([model.store.grads[off:off + n] for off, n in late]), outs[-1])
    else:
      outs.append(torch.cat([after[off:off + n] for off, n in late]))
  assert not torch.isnan(outs[0]).any() and torch.equal(outs[0], outs[1])
