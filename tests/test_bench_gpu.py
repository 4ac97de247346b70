"""bench.py's measurement helpers on small models of every input path (the driver only ever runs the default
configuration, so the tables of the other BASELINE shapes are exercised here)."""
import argparse
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('model_name,channels', [('geeco-f', 3), ('geeco-f', 4), ('e2e_vmc', 3)])
def test_bench_tables_small(model_name, channels):
  import bench
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner
  dev = torch.device('cuda', 0)
  N, K, H = 2, 3, 136
  goal = model_name == 'geeco-f'
  kw = dict(window_size=K, img_channels=channels, batch_size=N, img_height=H, img_width=H)
  if goal:
    kw.update(proc_obs='dynimg', proc_tgt='dyndiff')
  cfg = create_e2evmc_config(kw)
  model = (graph.GoalE2EVMC if goal else graph.E2EVMC)(cfg, N, dev, training=True)
  model.store.initialize(seed=0)
  bench.synthetic_batch(model, 1234)
  runner = TrainStepRunner(model, use_graph=False)
  runner.step()
  torch.cuda.synchronize()
  args = argparse.Namespace(model=model_name, channels=channels, seq_len=K, batch=N)
  hbm = bench.hbm_table(model, args, 5)
  assert hbm and hbm[-1]['piece'].startswith('adam') and all(r['us'] > 0 for r in hbm)
  if goal:
    assert len(hbm) == 4 and hbm[0]['piece'].startswith('dynimg buffer image') and hbm[2]['piece'].startswith('goal inputs as in the step')
  layers = bench.layer_table(model, 5)
  assert len(layers) >= 20 and all(r['us'] > 0 and r['kernel'] and r['us_p10'] <= r['us'] <= r['us_p90'] for r in layers)
  rl = bench.dominant_roofline(layers)
  assert rl['bound'] == 'mfma' and 0 < rl['frac'] < 1
  assert 'traffic_source' in rl and (rl['traffic'] is None) == (rl['traffic_source'] is None)
  fam = rl['family']
  assert 0 < fam['frac'] < 1 and fam['us_per_step'] >= rl['avg_launch_ms'] * 1e3 * 0.999 and fam['kernel'] and '<' not in fam['kernel']


def test_bench_two_ranks_rehearsal_and_shared_gpu_refusal(tmp_path):
  """bench.py as two ranks on the one test GPU over gloo (tests/_dp_launch.py): (a) without --allow-shared-gpu the run
  refuses to report a number (the ranks' device identities are not distinct: rc 3); (b) with it, the launcher protocol, the
  three-graph step, the one-call early bucket, the per-rank times and the comm report (overlap / serial / no exchange)
  run end to end.  The numbers of such a run mean nothing and are not looked at."""
  import json
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
    env.pop(k, None)
  base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1']
  tail = [os.path.join(root, 'tests', '_dp_launch.py'), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1',
          '--batch', '2', '--seq-len', '4', '--skip-cpu']
  out = subprocess.run(base + ['--master-port', str(18000 + os.getpid() % 500)] + tail, capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode != 0 and 'distinct GPU' in out.stderr, out.stderr[-2000:]
  out = subprocess.run(base + ['--master-port', str(18500 + os.getpid() % 500)] + tail + ['--allow-shared-gpu'],
                       capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
  assert len(line) == 1
  d = json.loads(line[0])
  assert d['n_gpus'] == 2 and 'REHEARSAL' in d['data'] and len(d['ranks']['ms_per_step']) == 2 and d['ranks']['distinct_devices'] == 1
  c = d['comm']
  assert c['mode'] in ('overlap', 'serial') and set(c['step_ms']) == {'overlap', 'serial', 'three_graphs_reserve16', 'three_graphs_overlap', 'three_graphs_serial', 'two_graphs', 'two_graphs_reserve16', 'two_graphs_serial',
                                                                      'no_exchange', 'overlap_reserve16', 'overlap_reserve32'}
  # order of an N > 1 run: the safe form in full (its figure on stderr BEFORE anything captures a collective), the short trial
  # of the one-graph forms, the fastest of those in full if the trial beat the safe form, the comm report
  assert 'safe form (three_graphs' in out.stderr and out.stderr.index('safe form (three_graphs') < out.stderr.index('dp form overlap')
  trial = c['trial_ms']
  assert set(trial) == {'three_graphs', 'two_graphs', 'two_graphs_reserve16', 'two_graphs_reserve32', 'two_graphs_serial', 'three_graphs_serial', 'overlap', 'overlap_reserve16',
                        'overlap_reserve32', 'serial'} and all(v > 0 for v in trial.values())
  full = c['forms_timed_in_full_ms']
  assert 'three_graphs_reserve16' in full and set(full) <= set(trial) | {'three_graphs_reserve16'} and 1 <= len(full) <= 3 and all(v > 0 for v in full.values())
  assert d['config']['dp_form'] in full and full[d['config']['dp_form']] == min(full.values())
  assert abs(d['ms_per_step'] - full[d['config']['dp_form']]) < 2e-3
  # gloo's collectives cannot be captured: every one-graph form of this rehearsal is the three-graph form, decided before
  # anything is captured (TrainStepRunner._capture; RCCL one-graph forms: tests/test_dp_gpu.py)
  assert 'cannot be captured into a hipGraph' in out.stderr
  assert c['replicas_bit_identical_after'] and all(c['replicas_bit_identical_after'].values()), c['replicas_bit_identical_after']
  assert 'rccl' in c            # (gloo rehearsal: whatever RCCL logged, or the reason there is no log)
  assert set(c['graphs_per_step'].values()) == {2, 3} and c['graphs_per_step']['two_graphs'] == c['graphs_per_step']['two_graphs_serial'] == 2 and d['config']['graphs_per_step'] in (2, 3)
  assert all(v > 0 for v in c['step_ms'].values()) and 'reserve_gain_ms' in c
  assert c['buckets']['early_allreduce_calls'] == 1 and c['buckets']['late_written_in_place']
  assert d['roofline']['kernel'] and d['roofline']['frac'] > 0 and len(d['layers']) >= 20      # rank 0's table, at any N


def test_bench_one_gpu_line_and_dp_one_rank_leg():
  """``python bench.py`` at N = 1 on a small batch: stdout is EXACTLY one line (RCCL's banner and everything else goes to
  stderr), and the line carries the ``dp_one_rank`` object: the data-parallel step over a one-rank RCCL group formed inside
  the bench process, one graph with the exchange captured next to round 4's three graphs."""
  import json
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '2', '--batch', '2', '--seq-len', '4',
                        '--skip-cpu', '--skip-other-configs', '--skip-input-pipeline', '--skip-inference'],
                       capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  lines = out.stdout.splitlines()
  assert len(lines) == 1, lines[:5]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 1 and d['config']['graphs_per_step'] == 1 and 'roofline' in d
  o = d['dp_one_rank']
  assert o['status'] == 'ok' and o['backend'] == 'nccl', o
  assert o['graphs_per_step']['overlap'] == 1 and o['graphs_per_step']['three_graphs_overlap'] == 3
  assert set(o['ms_per_step']) == set(o['delta_vs_single_graph_us']) == {n for n, _, _ in __import__('bench').DP_MODES}
  assert all(v > 0 for v in o['ms_per_step'].values()) and len(o['allreduce_us_one_rank']) == 2
  # what RCCL said when the communicator was created (INIT lines into a private file): the channel count is what decides how
  # many CUs the early bucket needs beside part 2
  assert o['rccl']['status'] == 'ok' and o['rccl']['log_lines'] > 0, o['rccl']


def test_bench_one_gpu_line_survives_a_stalled_extra_leg():
  """The legs after the headline (dp_one_rank first: graphs with RCCL's launches captured) run under bench.Watchdog: with a
  limit they cannot meet, stdout is still EXACTLY one line -- metric, value, roofline as they stood -- marked ``extras``,
  and the exit code is 0."""
  import json
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '2', '--batch', '2', '--seq-len', '4',
                        '--skip-cpu', '--extras-watchdog-s', '0.05'], capture_output=True, text=True, timeout=600, env=env)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
  lines = out.stdout.splitlines()
  assert len(lines) == 1, lines[:5]
  d = json.loads(lines[0])
  assert d['value'] > 0 and d['roofline']['frac'] > 0 and 'dp_one_rank' not in d
  assert d['extras'].startswith('WATCHDOG') and 'dp_one_rank' in d['extras'] and 'WATCHDOG' in out.stderr
