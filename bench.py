#!/usr/bin/env python
"""bench.py -- train-step frames/s of geeco-f (goal_e2evmc, rgb/dynimg/dyndiff) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched with
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU, RCCL).  Rank 0 prints
ONE JSON line.  A "step" = forward + backward + gradient all-reduce + Adam on one batch of
synthetic 256x256 RGB x 16-frame windows (BASELINE.json configs[1]: batch 32 per GPU), inputs
resident in HBM before the timed region.  ``value`` = global_batch * seq_len * steps / time.

Extra objects: ``roofline`` (dominant forward kernel: the conv2 launch of the three encoders,
fp32 MFMA bound) and ``cpu_baseline`` (the CPU restatement in oracle/, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip table
ENC_FWD_FLOP_PER_FRAME = {3: 1137180672, 4: 1174929408}
CONV2_MACS_PER_FRAME = 226492416                           # 128*128*48*288


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=32, help='windows per GPU (weak scaling)')
  ap.add_argument('--seq-len', type=int, default=16)
  ap.add_argument('--channels', type=int, default=3)
  ap.add_argument('--model', default='geeco-f', choices=['geeco-f', 'e2e_vmc'])
  ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying hipGraphs')
  ap.add_argument('--skip-cpu', action='store_true', help='skip the cpu_baseline leg')
  ap.add_argument('--cpu-steps', type=int, default=6)
  ap.add_argument('--cpu-batch', type=int, default=4)
  return ap.parse_args()


def log(msg):
  print('[bench] ' + msg, file=sys.stderr, flush=True)


def host_cores():
  """Threads the CPU leg may use: the affinity mask, capped by the cgroup CPU quota when one is set."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return max(1, min(n, int(os.environ.get('GEECO_CPU_THREADS', '16'))))


def synthetic_batch(model, seed):
  """seeded inputs generated on the device."""
  g = torch.Generator(device=model.device)
  g.manual_seed(seed)
  for k, buf in model.inputs.items():
    if k in ('rgb', 'target_rgb'):
      buf.copy_(torch.rand(buf.shape, generator=g, device=model.device))
    elif k in ('depth', 'target_depth'):
      buf.copy_(0.5 + 2.5 * torch.rand(buf.shape, generator=g, device=model.device))
    elif k == 'jnt_state':
      buf.copy_(torch.randn(buf.shape, generator=g, device=model.device))
    elif k in ('ee_state', 'obj_state'):
      buf.copy_(1.5 * torch.rand(buf.shape, generator=g, device=model.device))
    elif k == 'cmd':
      buf[:, :3].copy_(0.3 * torch.randn(buf.shape[0], 3, generator=g, device=model.device))
      buf[:, 3].copy_(torch.randint(-1, 2, (buf.shape[0],), generator=g, device=model.device).float())


def time_region(fn, iters):
  """Average milliseconds per call, HIP events on the current (launch) stream."""
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / iters


def roofline_conv2(model, iters):
  """Dominant forward kernel: conv2 (3x3, stride 2, 32 -> 48, 256^2 -> 128^2) over all encoders'
  frames in one launch.  achieved = algorithmic FLOP (2 * MACs, bias/ReLU excluded) / avg duration."""
  from geeco_amd import ops
  enc = model.enc
  L = enc.layers[1]
  x, y = enc.acts[0], enc.acts[1]
  w, b = enc._w(1), enc._b(1)

  def launch():
    ops.conv3x3_fwd_into(y, x, w, b, enc.G, x[0].numel(), enc.gs_p, enc.gs_p, y[0].numel(), enc.Nf, L['H'], L['W'],
                         L['Cin'], L['Cout'], L['stride'], relu=True, ws=enc.fws)
  for _ in range(3):
    launch()
  ms = time_region(launch, iters)
  macs = enc.G * enc.Nf * L['Ho'] * L['Wo'] * L['Cout'] * 9 * L['Cin']
  achieved = 2.0 * macs / (ms * 1e-3) / 1e12
  halo = os.environ.get('GEECO_NO_HALO') is None and L['H'] % 2 == 0 and L['W'] % 2 == 0
  ws = os.environ.get('GEECO_HALO_WS', '4') != '0'       # rocprof name: conv_s2_halo_fwd_ws_kernel<32, 48, 4>
  kname = ('conv_s2_halo_fwd_ws_kernel<32,48,4>' if ws else 'conv_s2_halo_fwd_kernel<32,48,true>') if halo else 'conv_gemm_kernel<128,48,16,4,1,true>'
  return {'bound': 'mfma', 'kernel': '%s (conv2 forward, %d frames)' % (kname, enc.G * enc.Nf),
          'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
          'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': recorded_traffic(kname),
          'avg_launch_ms': round(ms, 4), 'flop_per_launch': 2 * macs}


def recorded_traffic(kname):
  """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
  (profiles/r01/pmc_roofline_kernel.json: FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for wide reads on
  gfx950, + WRITE_SIZE; separate --pmc passes).  PMC counters cannot be read from inside this process,
  so this is the recorded measurement of the same kernel and shape, or null when there is none."""
  path = os.path.join(ROOT, 'profiles', 'r01', 'pmc_roofline_kernel.json')
  try:
    with open(path) as f:
      rec = json.load(f)
    return rec['traffic_bytes'] if rec.get('kernel') == kname else None
  except (OSError, ValueError, KeyError):
    return None


def encoder_forward_tflops(model, iters, channels):
  enc = model.enc
  for _ in range(2):
    enc.forward()
  ms = time_region(enc.forward, iters)
  flop = ENC_FWD_FLOP_PER_FRAME[channels] * enc.G * enc.Nf
  return flop / (ms * 1e-3) / 1e12, ms


def cpu_baseline(args):
  """CPU restatement (oracle/) of the same step on a bounded sample: BASELINE.json configs[0]
  (geeco-f rgb, batch 4, seq_len 16, 256x256), fwd + bwd + Adam, all host cores."""
  from oracle import geeco_oracle as O
  cores = host_cores()
  torch.set_num_threads(cores)
  log('cpu_baseline: %d threads' % cores)
  if args.model == 'geeco-f':
    cfg = O.make_config(proc_obs='dynimg', proc_tgt='dyndiff', window_size=args.seq_len, img_channels=args.channels,
                        batch_size=args.cpu_batch)
    goal = True
  else:
    cfg = O.make_config(window_size=args.seq_len, img_channels=args.channels, batch_size=args.cpu_batch)
    goal = False
  P = O.init_params(O.model_param_shapes(cfg, goal), seed=0)
  tr = O.OracleTrainer(cfg, goal, P, dtype=torch.float32)
  feats, labels = O.synthetic_batch(cfg, goal, args.cpu_batch, seed=1234)
  t0 = time.perf_counter()
  tr.train_step(feats, labels)   # warm-up
  warm = time.perf_counter() - t0
  log('cpu_baseline: warm-up step %.2f s' % warm)
  steps = max(1, min(args.cpu_steps, int(25.0 / max(warm, 1e-3))))   # keep the leg to ~25 s of CPU work
  t0 = time.perf_counter()
  for i in range(steps):
    loss, _ = tr.train_step(feats, labels)
    log('cpu_baseline: step %d/%d' % (i + 1, steps))
  dt = (time.perf_counter() - t0) / steps
  args.cpu_steps = steps
  return {'value': round(args.cpu_batch * args.seq_len / dt, 2), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
          'kind': 'port',
          'sample': '%d timed train steps (fwd+bwd+Adam, torch-CPU fp32 restatement in oracle/) of %s rgb%s batch=%d '
                    'seq_len=%d 256x256; %.3f s/step; NOT TF1.15 (not installable here)' %
                    (args.cpu_steps, args.model, 'd' if args.channels == 4 else '', args.cpu_batch, args.seq_len, dt)}


def main():
  args = parse_args()
  from geeco_amd import dist as gdist
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.runtime import TrainStepRunner

  world = gdist.init_from_env('nccl')
  rank = gdist.rank()
  if world != args.gpus and rank == 0:
    print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)

  if args.model == 'geeco-f':
    cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=args.seq_len,
                                    img_channels=args.channels, batch_size=args.batch))
    model = graph.GoalE2EVMC(cfg, args.batch, dev, training=True)
  else:
    cfg = create_e2evmc_config(dict(window_size=args.seq_len, img_channels=args.channels, batch_size=args.batch))
    model = graph.E2EVMC(cfg, args.batch, dev, training=True)
  model.store.initialize(seed=0)
  gdist.broadcast_variables(model.store)
  synthetic_batch(model, 1234 + rank)

  log('model built: %d parameters, batch %d/GPU, world %d' % (model.store.count_parameters(), args.batch, world))
  runner = TrainStepRunner(model, use_graph=not args.no_graph, warmup=2)
  runner.prepare()     # 2 eager steps + hipGraph capture, outside warm-up and timed region
  torch.cuda.synchronize()
  log('graphs captured' if not args.no_graph else 'eager mode')
  for i in range(args.warmup):
    runner.step()
    if i < 3:
      torch.cuda.synchronize()
      log('warm-up step %d done' % (i + 1))
  torch.cuda.synchronize()
  if world > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    runner.step()
  torch.cuda.synchronize()
  if world > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  dt = gdist.max_over_ranks(dt, dev)
  loss = float(model.loss)
  log('timed region: %d steps in %.3f s' % (args.steps, dt))

  if rank == 0:
    ms_step = dt / args.steps * 1e3
    frames = world * args.batch * args.seq_len
    out = {
        'metric': 'train-step frames/sec (256x256 RGB, seq_len=%d)' % args.seq_len,
        'value': round(frames * args.steps / dt, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s %s %dx%d seq_len=%d batch=%d/GPU (global %d), fwd+bwd+allreduce+Adam' %
                               (args.model, 'rgb' if args.channels == 3 else 'rgbd', cfg.img_height, cfg.img_width,
                                args.seq_len, args.batch, world * args.batch),
                   'parallelism': 'dp%d' % world, 'hipgraph': not args.no_graph, 'params': model.store.count_parameters()},
        'final_loss': round(loss, 6),
    }
    if world == 1:
      iters = max(10, min(args.steps, 50))
      if args.model == 'geeco-f':
        out['roofline'] = roofline_conv2(model, iters)
      tf_, ms_enc = encoder_forward_tflops(model, iters, args.channels)
      out['encoder_forward'] = {'tflops': round(tf_, 2), 'frac_of_f32_mfma_peak': round(tf_ / PEAK_F32_MFMA_TFLOPS, 4),
                                'ms': round(ms_enc, 3), 'frames': model.enc.G * model.enc.Nf}
      if not args.skip_cpu:
        out['cpu_baseline'] = cpu_baseline(args)
    print(json.dumps(out), flush=True)
  if torch.distributed.is_available() and torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
