#!/usr/bin/env python
"""bench.py -- train-step frames/s of geeco-f (goal_e2evmc, rgb/dynimg/dyndiff) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  With N > 1 and no WORLD_SIZE in
the environment this process NEVER touches the GPU: it starts ``python -m torch.distributed.run
--nproc-per-node N bench.py ...`` as a child (one rank per GPU, RCCL over xGMI), relays its output and
exits with its code; launched by the driver under torch.distributed.run it is one of the N ranks.
Rank 0 prints ONE JSON line.  A "step" = forward + backward + gradient all-reduce + Adam on one batch
of synthetic 256x256 RGB x 16-frame windows (BASELINE.json configs[1]: batch 32 per GPU), inputs
resident in HBM before the timed region.  ``value`` = global_batch * seq_len * steps / wall time of the
timed region (barrier + synchronize on both sides, max over ranks).

Extra objects (N = 1): ``layers`` (every conv launch of the step timed alone: median of >= 30 HIP-event
samples of 5 launches each: FLOP, us, TFLOP/s, fraction of the fp32 MFMA peak), ``roofline`` (the kernel with the largest
share of the step, taken from that table, plus the same by kernel family), ``hbm`` (dynimg calls and Adam against the
HBM peak), ``encoder_forward``, ``other_configs`` (config 4 and the per-GPU shape of config 5, a few steps each, with
their own oracle-pinned loss check) and ``cpu_baseline`` (the CPU restatement in oracle/, timed on the host cores).
N > 1 adds ``ranks`` (per-rank ms/step, device identities: the run refuses ranks that share a GPU) and ``comm``.  Order of an
N > 1 run: the ALWAYS-SAFE form of the data-parallel step first, in full (three replayed graphs, both all-reduces
ordinary RCCL launches between them): its figure goes to stderr at once and its JSON line is kept behind a watchdog; then a short
trial of the other safe forms (early bucket beside part 2 with / without CUs left to RCCL, or behind it), the fastest in full if it beats
the form on record; then the same for the one-graph forms (RCCL captured into the step graph); ``value`` =
the fastest full measurement (``config.dp_form``), with the replicas checked bitwise identical after each; then the comm report
(every form once more, the exchange alone).  A capture that fails, a form that hangs or replicas that differ cannot take the
number away: rank 0 prints the safe form's line and every rank leaves.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip table
PEAK_HBM_TBS = 8.0                # spec; 6.29 TB/s is what a float4 copy achieves (same guide)
ENC_FWD_FLOP_PER_FRAME = {3: 1137180672, 4: 1174929408}


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--batch', type=int, default=32, help='windows per GPU (weak scaling)')
  ap.add_argument('--seq-len', type=int, default=16)
  ap.add_argument('--channels', type=int, default=3)
  ap.add_argument('--model', default='geeco-f', choices=['geeco-f', 'e2e_vmc'])
  ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying hipGraphs')
  ap.add_argument('--skip-cpu', action='store_true', help='skip the cpu_baseline leg')
  ap.add_argument('--skip-layers', action='store_true', help='skip the per-layer / roofline / hbm legs (profiling runs)')
  ap.add_argument('--cpu-steps', type=int, default=60, help='upper bound; the leg stops after ~12 s of CPU work')
  ap.add_argument('--cpu-batch', type=int, default=4)
  ap.add_argument('--dp-serial', action='store_true',
                  help='N > 1: the safe form runs the gradient exchange AFTER the backward instead of beside its bottom part, and '
                       'no other form is timed in full (comm.step_ms still reports every form)')
  ap.add_argument('--dp-fixed', '--dp-three-graphs', dest='dp_fixed', action='store_true',
                  help='N > 1: report the safe form (exchange launched between three captured graphs); the ONE-graph forms '
                       '(RCCL captured into the step graph) are neither tried nor timed in full')
  ap.add_argument('--dp-watchdog-s', type=float, default=240.0,
                  help='N > 1: seconds the captured forms (trial, second timed region, comm report) may take after the safe form '
                       'has been measured; beyond that rank 0 prints the safe form\'s line and every rank exits (0 = no watchdog)')
  ap.add_argument('--extras-watchdog-s', type=float, default=300.0,
                  help='N = 1: seconds the legs after the headline, roofline and CPU baseline (dp_one_rank, other configs, input pipeline, '
                       'inference) may take together; beyond that the line as it stood is printed and the process exits (0 = no watchdog)')
  ap.add_argument('--skip-comm-report', action='store_true', help='N > 1: skip the comm report (every form of the step, exchange alone)')
  ap.add_argument('--skip-other-configs', action='store_true', help='skip the other_configs leg (config 4 / config 5 shapes)')
  ap.add_argument('--skip-input-pipeline', action='store_true', help='skip the input_pipeline leg (on-disk dataset -> Estimator.train)')
  ap.add_argument('--skip-inference', action='store_true', help='skip the inference leg (predictor latency, Estimator.evaluate)')
  ap.add_argument('--pipeline-episodes', type=int, default=64, help='episode files of the generated on-disk dataset')
  ap.add_argument('--skip-dp-one-rank', action='store_true', help='skip the dp_one_rank leg (N = 1: three-graph step over a one-rank RCCL group)')
  ap.add_argument('--allow-shared-gpu', action='store_true',
                  help='REHEARSAL on a one-GPU box (tests/_dp_launch.py): do not refuse ranks that share a device')
  return ap.parse_args()


def log(msg):
  print('[bench] ' + msg, file=sys.stderr, flush=True)


# ======================================================================================================
# N > 1 without a launcher: start the ranks ourselves (before any GPU call in this process)
# ======================================================================================================
def spawn_ranks(args):
  import torch
  have = torch.cuda.device_count()        # counting devices does not initialise the GPU
  if have < args.gpus:
    log('--gpus %d requested but only %d GPU(s) are visible' % (args.gpus, have))
    return 2
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  log('launching %d ranks: %s' % (args.gpus, ' '.join(cmd)))
  proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
  line = None
  for ln in proc.stdout.splitlines():
    if ln.startswith('{') and '"metric"' in ln:
      line = ln
    else:
      print(ln, file=sys.stderr)
  if proc.returncode != 0 or line is None:
    log('child launcher failed (rc %d)' % proc.returncode)
    return proc.returncode or 3
  if json.loads(line).get('n_gpus') != args.gpus:
    log('the process group formed with %s ranks, not %d' % (json.loads(line).get('n_gpus'), args.gpus))
    return 3
  print(line, flush=True)
  return 0


# ======================================================================================================
# helpers
# ======================================================================================================
def host_cores():
  """(threads the CPU leg uses, cores of the box): affinity mask capped by the cgroup CPU quota, then by GEECO_CPU_THREADS."""
  box = os.cpu_count() or 1
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else box
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return max(1, min(n, int(os.environ.get('GEECO_CPU_THREADS', '16')))), box, n


def synthetic_batch(model, seed):
  """seeded inputs generated on the device."""
  import torch
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
  """Average milliseconds per call, HIP events on the current stream (the stream our kernels are launched on)."""
  import torch
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / iters


LAUNCHES_PER_SAMPLE = 5
DP_RESERVE_PROBE = 16          # CUs left to RCCL in the extra comm.step_ms measurement of an N > 1 run


def time_launches(fn, samples=30, per_sample=LAUNCHES_PER_SAMPLE):
  """(median, p10, p90) milliseconds per call of ``fn`` over `samples` samples; one sample = one HIP event pair (on the
  current stream, the one our kernels are launched on) around `per_sample` back-to-back calls.  The median makes one
  hiccup move one sample, not the reported number (round 2 had ONE pair around 20 launches: a 27 us kernel once read
  78 us).  Why not one pair per single launch: measured (scripts/dev/timing_probe.py, profiles/r03/timing_probe.txt) a
  pair around ONE launch reads 3-9 % above the kernel duration rocprofv3 reports inside the step on the big streaming
  kernels (conv2 forward 411.6 vs 377.5 us) and, without a kernel ahead of it in the queue, measures the host's enqueue
  rate on the small ones (conv8 forward 118 us for a 17 us kernel); pairs around 4-8 launches agree with rocprofv3 to
  1-2 % (380.1 / 377.6 us)."""
  import torch
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(samples)]
  torch.cuda.synchronize()
  for a, b in evs:
    a.record()
    for _ in range(per_sample):
      fn()
    b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) / per_sample for a, b in evs)
  return percentile(ts, 0.5), percentile(ts, 0.1), percentile(ts, 0.9)


def percentile(sorted_vals, q):
  if not sorted_vals:
    return None
  i = min(len(sorted_vals) - 1, max(0, int(round(q * (len(sorted_vals) - 1)))))
  return sorted_vals[i]


# ======================================================================================================
# per-layer table, dominant-kernel roofline, HBM-bound pieces
# ======================================================================================================
def layer_table(model, samples):
  """Every conv launch of one step, timed ALONE: forward, input gradient and filter gradient of conv1..conv8 over all
  encoder frames; `samples` (>= 30) event-pair samples of 5 back-to-back launches each, MEDIAN reported (p10 / p90 beside
  it; see time_launches).
  FLOP = 2 * MACs of the layer (bias / ReLU / mask excluded).  A call that also runs a small reduce /
  epilogue kernel is timed as a whole (its names are listed)."""
  from geeco_amd import ops
  enc = model.enc
  rows = []
  frames = enc.G * enc.Nf
  for l, L in enumerate(enc.layers):
    cin_real = enc.Cin if l == 0 else L['Cin']
    flop = 2.0 * frames * L['Ho'] * L['Wo'] * L['Cout'] * 9 * cin_real
    todo = [('fwd', lambda l=l: enc.launch_fwd(l), flop)]
    if l >= 1:
      if l == 1 and enc.fused_bottom:
        L0 = enc.layers[0]
        flop0 = 2.0 * frames * L0['Ho'] * L0['Wo'] * L0['Cout'] * 9 * enc.Cin
        todo.append(('dgrad+conv1_wgrad', lambda l=l: enc.launch_dgrad(l), flop + flop0))
      else:
        todo.append(('dgrad', lambda l=l: enc.launch_dgrad(l), flop))
    if not (l == 0 and enc.fused_bottom):
      todo.append(('wgrad', lambda l=l: enc.launch_wgrad(l), flop))
    for what, fn, fl in todo:
      names = ops.kernel_trace(fn)
      fn()
      ms, p10, p90 = time_launches(fn, samples)
      tf = fl / (ms * 1e-3) / 1e12
      rows.append({'layer': 'conv%d' % (l + 1), 'op': what, 'kernel': names[0] if names else '?', 'kernels': names,
                   'flop': int(fl), 'us': round(ms * 1e3, 1), 'us_p10': round(p10 * 1e3, 1), 'us_p90': round(p90 * 1e3, 1),
                   'tflops': round(tf, 2), 'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4)})
  # the step runs conv7's input gradient and conv7's / conv8's filter gradients as ONE heterogeneous launch: its own row; the
  # three single launches above stay in the table for reference, marked as not part of the step
  if getattr(enc, 'hetero_top', 0) == 1 and getattr(enc, 'pair_top', False) and not enc.split_top and len(enc.layers) == 8:
    fn = lambda: enc.launch_top_bwd(6, (6, 7))
    if fn():
      single = {('conv7', 'dgrad'), ('conv7', 'wgrad'), ('conv8', 'wgrad')}
      fl = sum(r['flop'] for r in rows if (r['layer'], r['op']) in single)
      names = ops.kernel_trace(fn)
      ms, p10, p90 = time_launches(fn, samples)
      tf = fl / (ms * 1e-3) / 1e12
      for r in rows:
        if (r['layer'], r['op']) in single:
          r['in_step'] = False
      rows.append({'layer': 'conv7+conv8', 'op': 'dgrad7+wgrad7+wgrad8', 'kernel': names[0] if names else '?', 'kernels': names,
                   'flop': int(fl), 'us': round(ms * 1e3, 1), 'us_p10': round(p10 * 1e3, 1), 'us_p90': round(p90 * 1e3, 1),
                   'tflops': round(tf, 2), 'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4)})
  return rows


def dominant_roofline(rows):
  """The kernel (by name) whose launches take the largest share of a step; achieved = sum of the algorithmic FLOP
  of those launches / sum of their durations (= FLOP per launch / average launch duration)."""
  rows = [r for r in rows if r.get('in_step', True)]       # launches the step really runs
  by = {}
  for r in rows:
    d = by.setdefault(r['kernel'], {'us': 0.0, 'flop': 0.0, 'launches': []})
    d['us'] += r['us']
    d['flop'] += r['flop']
    d['launches'].append('%s %s' % (r['layer'], r['op']))
  name, d = max(by.items(), key=lambda kv: kv[1]['us'])
  achieved = d['flop'] / (d['us'] * 1e-6) / 1e12
  n = len(d['launches'])
  total = sum(r['us'] for r in rows)
  traffic, source = recorded_traffic(name)
  # the same ranking by kernel FAMILY (template name without its arguments): several instantiations of one kernel
  # (e.g. the LDS-staged filter gradient of conv3..conv6) can together outweigh the largest single instantiation
  fam = {}
  for r in rows:
    f = fam.setdefault(r['kernel'].split('<')[0], {'us': 0.0, 'flop': 0.0, 'launches': []})
    f['us'] += r['us']
    f['flop'] += r['flop']
    f['launches'].append('%s %s' % (r['layer'], r['op']))
  fname, f = max(fam.items(), key=lambda kv: kv[1]['us'])
  fach = f['flop'] / (f['us'] * 1e-6) / 1e12
  return {'bound': 'mfma', 'kernel': name, 'launches_per_step': d['launches'], 'achieved': round(achieved, 2),
          'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
          'traffic': traffic, 'traffic_source': source, 'avg_launch_ms': round(d['us'] / n * 1e-3, 4),
          'flop_per_launch': int(d['flop'] / n), 'share_of_conv_time': round(d['us'] / total, 4),
          'timer': 'HIP event pairs on the launch stream, each around %d back-to-back launches; median of >= 30 samples' % LAUNCHES_PER_SAMPLE,
          'family': {'kernel': fname, 'launches_per_step': f['launches'], 'achieved': round(fach, 2),
                     'frac': round(fach / PEAK_F32_MFMA_TFLOPS, 4), 'us_per_step': round(f['us'], 1),
                     'share_of_conv_time': round(f['us'] / total, 4)}}


def csrc_sha16():
  """First 16 hex digits of the SHA-256 over the kernel sources (geeco_amd/csrc/*.hip, *.h, *.cpp, sorted by name): what a recorded
  PMC figure is tied to (scripts/dev/pmc_roofline.py writes it beside the traffic)."""
  import hashlib
  h = hashlib.sha256()
  d = os.path.join(ROOT, 'geeco_amd', 'csrc')
  for fn in sorted(os.listdir(d)):
    if fn.endswith(('.hip', '.h', '.cpp')):
      h.update(fn.encode() + b'\0')
      with open(os.path.join(d, fn), 'rb') as f:
        h.update(f.read())
  return h.hexdigest()[:16]


def recorded_traffic(kname):
  """(HBM bytes per launch, where the figure comes from).  PMC counters cannot be read from inside this process, so
  this is the RECORDED measurement of the same kernel and shapes from the committed rocprofv3 PMC passes
  (profiles/rNN/pmc_roofline_kernel.json: FETCH_SIZE x 2 as MI355X_MICROARCH.md prescribes for wide reads on gfx950,
  + WRITE_SIZE; separate --pmc passes), newest round first, or (None, None) when there is none.  A record carries the hash of
  the kernel sources its passes ran (``csrc_sha16``); when that is not the hash of THIS tree's sources the figure is
  labelled STALE (it describes an earlier build of the kernels), as is a record from before the hash existed."""
  rounds = sorted((d for d in os.listdir(os.path.join(ROOT, 'profiles')) if re.fullmatch(r'r\d+', d)), reverse=True)
  for rnd in rounds:
    rel = os.path.join('profiles', rnd, 'pmc_roofline_kernel.json')
    try:
      with open(os.path.join(ROOT, rel)) as f:
        rec = json.load(f)
      if rec.get('kernel', '').replace(' ', '') == kname.replace(' ', ''):
        have = rec.get('csrc_sha16')
        fresh = have is not None and have == csrc_sha16()
        return rec['traffic_bytes'], '%s (%s rocprofv3 --pmc passes of this kernel at these shapes, not measured in this run)' % (
            rel, 'recorded' if fresh else 'STALE: recorded for %s, this run is %s;' %
            ('kernel sources ' + have if have else 'a build from before the source hash was recorded', csrc_sha16()))
    except (OSError, ValueError, KeyError):
      pass
  return None, None


def hbm_table(model, args, iters):
  """HBM-bound pieces against the 8.0 TB/s peak, algorithmic bytes per SURVEY.md 8(d).  ``in_step`` False: the single-image entry
  point (geeco_dynimg_fwd: a sum launch + a normalisation launch), timed for reference -- the goal model's step runs both images
  through the one-launch row below."""
  from geeco_amd import ops
  rows = []
  N, K, C = model.N, model.K, model.C
  HW = model.H * model.W

  def add(name, nbytes, fn, in_step=True):
    fn()
    ms, p10, p90 = time_launches(fn, iters)
    tbs = nbytes / (ms * 1e-3) / 1e12
    rows.append({'piece': name, 'bytes': int(nbytes), 'us': round(ms * 1e3, 1), 'us_p10': round(p10 * 1e3, 1),
                 'us_p90': round(p90 * 1e3, 1), 'TB/s': round(tbs, 3), 'frac': round(tbs / PEAK_HBM_TBS, 4), 'in_step': in_step})
  if args.model == 'geeco-f' and getattr(model, 'split_rgbd', False):
    # RGB-D: the dynimg kernels read rgb and depth from their own tensors (no packed copy of the frames)
    inp, x_in = model.inputs, model.enc.x_in
    rgb, dep = inp['rgb'], inp['depth']
    add('dynimg buffer image (K=%d), rgb + depth unpacked' % K, 4.0 * N * HW * C * (K + 1),
        lambda: ops.dynimg_rgbd_into(x_in[1], rgb, dep, K, N, HW, model.dyn_ws, K * HW * 3, HW * 3, K * HW, HW), in_step=False)
    add('dynimg diff image (K=2), rgb + depth unpacked', 4.0 * N * HW * C * 3,
        lambda: ops.dynimg_rgbd_into(x_in[2], rgb[:, K - 1], dep[:, K - 1], 2, N, HW, model.dyn_ws, K * HW * 3, 0, K * HW, 0,
                                     rgb2=inp['target_rgb'], depth2=inp['target_depth']), in_step=False)
  elif args.model == 'geeco-f':
    frames, tgt = model._frames()
    x_in = model.enc.x_in
    cur = frames[:, K - 1]
    add('dynimg buffer image (K=%d)' % K, 4.0 * N * HW * C * (K + 1),
        lambda: ops.dynimg_into(x_in[1], frames, K, N, HW, C, 4, model.dyn_ws, K * HW * C, HW * C), in_step=False)
    add('dynimg diff image (K=2)', 4.0 * N * HW * C * 3,
        lambda: ops.dynimg_into(x_in[2], cur, 2, N, HW, C, 4, model.dyn_ws, K * HW * C, 0, frames2=tgt), in_step=False)
  if args.model == 'geeco-f' and getattr(model, 'last_from_dynimg', False) \
      and HW % 4 == 0 and (C == 3 or getattr(model, 'split_rgbd', False)):
    # what the step really runs: both images + the current frame's padded copy in ONE launch, one pass over the window (both
    # images normalised in registers); algorithmic bytes = the two dynimg figures of SURVEY 8(d)
    inp, x_in = model.inputs, model.enc.x_in
    kw = dict(depth=inp['depth'], tgt_depth=inp['target_depth'], dsample_stride=K * HW, dframe_stride=HW) if C == 4 else {}
    add('goal inputs as in the step: buffer image (K=%d) + diff image (K=2) + current frame, 1 launch' % K,
        4.0 * N * HW * C * (K + 1 + 3),
        lambda: ops.goal_dynimgs_into(x_in[0], x_in[1], x_in[2], inp['rgb'], inp['target_rgb'], K, N, HW, model.dyn_ws2,
                                      K * HW * 3, HW * 3, **kw))
  s = model.store
  P = s.count_parameters()
  add('adam (7 x 4 B x params)', 28.0 * P,
      lambda: ops.adam_tf(s.params, s.grads, s.adam_m, s.adam_v, s.size, model.scal, grad_scale=1.0, l2=0.0))
  return rows


def encoder_forward_tflops(model, iters, channels):
  enc = model.enc
  for _ in range(2):
    enc.forward()
  ms, _, _ = time_launches(enc.forward, iters)
  flop = ENC_FWD_FLOP_PER_FRAME[channels] * enc.G * enc.Nf
  return flop / (ms * 1e-3) / 1e12, ms


def cpu_baseline(args):
  """CPU restatement (oracle/) of the same step on a bounded sample: BASELINE.json configs[0]
  (geeco-f rgb, batch 4, seq_len 16, 256x256), fwd + bwd + Adam, all host cores."""
  import torch
  from oracle import geeco_oracle as O
  threads, box, usable = host_cores()
  torch.set_num_threads(threads)
  log('cpu_baseline: %d threads (box has %d cores, %d usable by this process)' % (threads, box, usable))
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
  steps = max(3, min(args.cpu_steps, int(12.0 / max(warm, 1e-3))))   # a bounded sample: ~12 s of CPU work
  t0 = time.perf_counter()
  for i in range(steps):
    tr.train_step(feats, labels)
    if (i + 1) % 10 == 0 or i + 1 == steps:
      log('cpu_baseline: step %d/%d' % (i + 1, steps))
  dt = (time.perf_counter() - t0) / steps
  return {'value': round(args.cpu_batch * args.seq_len / dt, 2), 'unit': 'frames/s', 'cores': torch.get_num_threads(),
          'box_cores': box, 'usable_cores': usable, 'kind': 'port',
          'sample': '%d timed train steps (fwd+bwd+Adam, torch-CPU fp32 restatement in oracle/) of %s rgb%s batch=%d '
                    'seq_len=%d 256x256; %.3f s/step; NOT TF1.15 (not installable here)' %
                    (steps, args.model, 'd' if args.channels == 4 else '', args.cpu_batch, args.seq_len, dt)}


def check_losses(args, first_loss, final_loss, total_steps):
  """Parity guard of the bench itself: the loss of the FIRST optimiser step (seed-0 weights, batch seed 1234) must equal
  the committed value the fp64 oracle gives on the same inputs (tests/golden/bench_losses.json, written by
  tests/golden/make_bench_losses.py) to 1e-4 relative, or the run exits non-zero.  The loss after the run's last step is
  reported next to the value an earlier build recorded for the same step count, for information only: 100+ Adam steps on
  one repeated batch are a chaotic trajectory (ReLU masks flip) and any change of summation order moves it by percents."""
  key = '%s c%d b%d k%d' % (args.model, args.channels, args.batch, args.seq_len)
  try:
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_losses.json')) as f:
      ref = json.load(f).get(key)
  except (OSError, ValueError):
    ref = None
  out = {'first_step_loss': round(first_loss, 6), 'config_key': key}
  if not ref:
    out['status'] = 'no committed value for this config'
    return out, True
  rel = abs(first_loss - ref['first_step_loss_oracle_fp64']) / abs(ref['first_step_loss_oracle_fp64'])
  out.update({'first_step_loss_oracle': ref['first_step_loss_oracle_fp64'], 'first_step_rel_err': float('%.3g' % rel)})
  ok = rel <= 1e-4
  fin = (ref.get('final_loss_hip') or {}).get(str(total_steps))
  if fin is not None:
    out.update({'final_loss_recorded_earlier_build': fin, 'final_rel_diff_informational': float('%.3g' % (abs(final_loss - fin) / abs(fin)))})
  out['status'] = 'ok' if ok else 'MISMATCH'
  return out, ok


# ======================================================================================================
def device_identity(local):
  """(identity, detail): `identity` is equal for two ranks exactly when they sit on the same physical GPU -- host name + PCI
  address (domain:bus:device), which two devices cannot share.  Where this torch build does not expose the PCI address the
  device UUID stands in; with neither, identity is None and the caller SKIPS the shared-GPU refusal with a warning (a
  per-process local index is never compared: launchers that isolate ranks through HIP_VISIBLE_DEVICES give every rank
  index 0, and an honest N-GPU run must not refuse itself).  `detail` adds the UUID for the report."""
  import torch
  pr = torch.cuda.get_device_properties(local)
  host = socket.gethostname()
  u = getattr(pr, 'uuid', None)
  if hasattr(pr, 'pci_bus_id'):
    ident = '%s pci=%04x:%02x:%02x' % (host, getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, getattr(pr, 'pci_device_id', 0))
  elif u is not None:
    ident = '%s uuid=%s' % (host, u)
  else:
    return None, '%s local index %d (no PCI address / UUID exposed by this torch build)' % (host, local)
  return ident, ident + (' uuid=%s' % u if u is not None else '')


def build_model(model_name, channels, seq_len, batch, dev):
  from geeco_amd import graph
  from geeco_amd.params import create_e2evmc_config
  if model_name == 'geeco-f':
    cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=seq_len, img_channels=channels,
                                    batch_size=batch))
    return cfg, graph.GoalE2EVMC(cfg, batch, dev, training=True)
  cfg = create_e2evmc_config(dict(window_size=seq_len, img_channels=channels, batch_size=batch))
  return cfg, graph.E2EVMC(cfg, batch, dev, training=True)


def timed_region(model, runner, steps, warmup, world, dev, verbose=True):
  """`warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides (max over ranks)."""
  import torch
  from geeco_amd import dist as gdist
  runner.prepare()     # (second eager step +) hipGraph capture, outside warm-up and timed region
  torch.cuda.synchronize()
  if verbose:
    log('graphs captured (%d per step)' % runner.bucket_info()['graphs_per_step'] if runner.use_graph else 'eager mode')
  for i in range(warmup):
    runner.step()
    if i < 3 and verbose:
      torch.cuda.synchronize()
      log('warm-up step %d done' % (i + 1))
  torch.cuda.synchronize()
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
  if world > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  evs[0].record()
  for i in range(steps):
    runner.step()
    evs[i + 1].record()
  torch.cuda.synchronize()
  if world > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  dt_local = time.perf_counter() - t0
  return {'runner': runner, 'final_loss': float(model.loss), 'dt_local': dt_local,
          'dt': gdist.max_over_ranks(dt_local, dev), 'total_steps': runner._calls,
          'rank_ms': gdist.gather_floats(dt_local / steps * 1e3, dev),
          'per_step': sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))}


def timed_steps(model, steps, warmup, use_graph, world, dev, verbose=True, form=None):
  """First optimiser step eager (its loss is the one checked against the oracle), second eager step + hipGraph capture,
  `warmup` untimed replays, then the timed region.  ``form`` (N > 1): a name of runtime.DP_FORMS (None = the safe default,
  three graphs with the exchange launched between them)."""
  import torch
  from geeco_amd.runtime import TrainStepRunner, dp_form_kwargs
  runner = TrainStepRunner(model, use_graph=use_graph, warmup=2, **(dp_form_kwargs(form) if world > 1 else {}))
  runner.step()
  torch.cuda.synchronize()
  first_loss = float(model.loss)
  r = timed_region(model, runner, steps, warmup, world, dev, verbose)
  model.check_device_errors()      # an input-stage block that gave up waiting wrote NaN images: never report a number over that
  r['first_loss'] = first_loss
  return r


class Watchdog:
  """N > 1: the forms of the data-parallel step that capture RCCL's launches into a hipGraph have never run with more than one
  rank on hardware (no multi-GPU box was available to any round of this build).  A collective that HANGS inside a replayed
  graph cannot be recovered in-process, a capture that FAILS leaves its streams in capture mode (runtime.TrainStepRunner._capture)
  and a process that touched the GPU is never restarted.  So: the always-safe form is measured FIRST, in full (warm-up + the K
  timed steps), its line is kept as `provisional`, and this timer is armed around everything that follows.  If that does not
  finish in time -- or any rank raises the abort flag (``abort``: a counter in the rendezvous store, polled once a second, so
  the peers of a rank whose capture failed do not sit in a collective until the deadline) -- rank 0 writes the provisional
  line, a complete and valid measurement of the safe form marked as such, and every rank leaves with os._exit (no atexit
  handlers, no destructors that would wait for a stuck stream).  The run cannot end without a number.

  N = 1: the same timer around the legs that FOLLOW the headline, its roofline and the CPU baseline (dp_one_rank -- which replays
  graphs with RCCL's one-rank launches captured --, the other configurations, input pipeline, inference): one GPU-suite run of
  round 6 sat in exactly such a replay for seven minutes (profiles/r06/asked_and_answered.md), so the line the driver needs is
  complete BEFORE those legs start and is what the run leaves with if one of them stalls (``extras``: which, and why)."""

  def __init__(self, seconds, emit, aborted=None, keeps='the measurement of the safe form', after='what follows the safe form'):
    import threading
    self.seconds, self.emit, self.aborted, self.keeps, self.after = float(seconds), emit, aborted, keeps, after
    self._stop = threading.Event()
    self._once = threading.Lock()      # fire() may be reached by the timer thread and by the main thread's exception path at once
    self._t = threading.Thread(target=self._run, name='bench-watchdog', daemon=True)
    self.provisional = None
    self.phase = 'start'

  def arm(self):
    self._t.start()

  def disarm(self):
    self._stop.set()

  def fire(self, why):
    """Leaves NOW with the provisional line (also called by the main thread when its own captured form raised)."""
    import faulthandler
    if not self._once.acquire(blocking=False):
      time.sleep(3600)                 # the other caller is writing the line and will end the process
    log('WATCHDOG: %s (phase: %s); leaving with %s' % (why, self.phase, self.keeps))
    try:
      faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
    except Exception:
      pass
    try:
      self.emit(self.provisional, '%s; phase: %s' % (why, self.phase))
    finally:
      sys.stderr.flush()
      os._exit(0)

  def _run(self):
    t0 = time.monotonic()
    while not self._stop.wait(1.0 if self.aborted else max(self.seconds, 0.01)):
      if self.seconds > 0 and time.monotonic() - t0 >= self.seconds:
        self.fire('%s did not finish within %.0f s' % (self.after, self.seconds))
      try:
        if self.aborted and self.aborted():
          self.fire('another rank gave up on the captured forms')
      except Exception:
        pass


def step_flop(channels, frames):
  """fwd + bwd FLOP of the encoders (backward = 2 x forward - conv1's input gradient)."""
  return (3 * ENC_FWD_FLOP_PER_FRAME[channels] - 2 * 56623104 * (channels / 3.0)) * frames


def other_configs(args, dev):
  """The other single-GPU shapes of BASELINE.json, a few steps each, with the same oracle-pinned first-step loss check:
  config 4 (e2e_vmc N=64 K=16: 1024 frame passes) and the per-GPU shape of config 5 (geeco-f rgbd N=32 K=32).  Not bench
  lines (the headline metric is quoted on configs[1]); here so that their numbers are driver-run, not only DESIGN text."""
  import argparse
  import torch
  out = []
  for label, mname, ch, k, b in (('config4: e2e_vmc rgb seq_len=16 batch=64', 'e2e_vmc', 3, 16, 64),
                                 ('config5 per-GPU shape: geeco-f rgbd seq_len=32 batch=32', 'geeco-f', 4, 32, 32)):
    if (mname, ch, k, b) == (args.model, args.channels, args.seq_len, args.batch):
      continue
    cfg, model = build_model(mname, ch, k, b, dev)
    model.store.initialize(seed=0)
    synthetic_batch(model, 1234)
    steps = max(5, min(args.steps, 10))
    r = timed_steps(model, steps, 3, not args.no_graph, 1, dev, verbose=False)
    ms = r['dt'] / steps * 1e3
    a2 = argparse.Namespace(model=mname, channels=ch, batch=b, seq_len=k)
    chk, ok = check_losses(a2, r['first_loss'], r['final_loss'], r['total_steps'])
    frames = model.enc.G * model.enc.Nf
    out.append({'workload': label, 'value': round(b * k * steps / r['dt'], 1), 'unit': 'frames/s', 'steps': steps,
                'ms_per_step': round(ms, 3), 'encoder_frame_passes': frames,
                'step_frac_of_f32_mfma_peak': round(step_flop(ch, frames) / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                'loss_check': chk})
    log('other config "%s": %.3f ms/step, loss check %s' % (label, ms, chk['status']))
    del r, model
    torch.cuda.empty_cache()
    if not ok:
      return out, False
  return out, True


# ======================================================================================================
# real-data legs: on-disk dataset -> reader -> Estimator.train; predictor latency; Estimator.evaluate
# ======================================================================================================
DISTINCT_SCENES = 8        # episodes generated (in parallel threads); the rest of the files are byte copies under other names


def make_dataset(root, episodes, threads):
  """``episodes`` episode files of 100 x 256 x 256 frames in the reference's on-disk layout (input_fn.write_synthetic_dataset:
  toy table-top scenes, zlib level 6 as TFRecordWriter's ZLIB option): DISTINCT_SCENES are generated, the others are byte
  copies under their own names — every file is inflated, checked, parsed and uploaded on its own (the cache keys on the
  path), only generating them is shortened."""
  import shutil
  from concurrent.futures import ThreadPoolExecutor
  from geeco_amd import input_fn as I
  n_gen = min(DISTINCT_SCENES, episodes)
  parts = os.path.join(root, 'parts')
  with ThreadPoolExecutor(max_workers=min(threads, n_gen)) as ex:
    list(ex.map(lambda e: I.write_synthetic_dataset(os.path.join(parts, str(e)), 1, seed=100 + e), range(n_gen)))
  meta = I.write_synthetic_dataset(root, 0)
  names = []
  for e in range(episodes):
    name = 'ep%05d.tfrecord.zlib' % e
    src = os.path.join(parts, str(e % n_gen), 'data', 'ep00000.tfrecord.zlib')
    shutil.copy(src, os.path.join(root, 'data', name))
    names.append(name)
  shutil.rmtree(parts)
  for split, mode, sel in (('default', 'train', names), ('default', 'eval', names), ('warmup', 'train', names[:1])):
    os.makedirs(os.path.join(root, 'splits', split), exist_ok=True)
    with open(os.path.join(root, 'splits', split, mode + '.txt'), 'w') as fp:
      fp.write('\n'.join(sel) + '\n')
  return meta, [os.path.join(root, 'data', n) for n in names]


def input_pipeline_report(args, dev, synthetic_ms_per_step, workdir):
  """Training from an ON-DISK dataset through the reference's call surface (pickplace_input_fn -> Estimator.train,
  scripts/train_e2evmc.py:266-291; reader = geeco_gym.py:436-473): reader episodes/s (pure-Python reader, native reader
  with 1 and ``threads`` threads) and end-to-end frames/s of epoch 1 (every episode inflated, parsed, uploaded) and of
  epochs 2-3 (episodes resident in HBM: EPISODE_CACHE).  Not the headline: that one keeps its inputs in HBM by definition."""
  import torch
  from concurrent.futures import ThreadPoolExecutor
  from geeco_amd import estimator as est
  from geeco_amd import input_fn as I
  from geeco_amd.params import create_e2evmc_config
  threads, box, usable = host_cores()
  root = os.path.join(workdir, 'dataset')
  t0 = time.perf_counter()
  meta, paths = make_dataset(root, args.pipeline_episodes, threads)
  gen_s = time.perf_counter() - t0
  K, B = args.seq_len, args.batch
  windows = (meta.episode_length - 1) - K + 1
  frames_per_episode = windows * K
  log('input_pipeline: %d episode files (%.1f MB each) written in %.1f s' % (len(paths), os.path.getsize(paths[0]) / 1e6, gen_s))
  out = {'dataset': '%d files x %d frames x %dx%d rgb(uint8 as float list)+depth(float32), zlib; %d distinct scenes, the rest '
                    'byte copies; %.1f MB per file on disk, %.1f MB inflated' %
                    (len(paths), meta.episode_length, meta.img_height, meta.img_width, min(DISTINCT_SCENES, len(paths)),
                     os.path.getsize(paths[0]) / 1e6, 0.0),
         'frames_per_episode': frames_per_episode, 'host_threads': threads}

  def rate(fn, sel, nthreads):
    t = time.perf_counter()
    if nthreads == 1:
      for p_ in sel:
        fn(p_)
    else:
      with ThreadPoolExecutor(max_workers=nthreads) as ex:
        list(ex.map(fn, sel))
    return len(sel) / (time.perf_counter() - t)
  from geeco_amd import tfrecord as T
  with T.EpisodeReader(paths[0]) as rd:
    out['dataset'] = out['dataset'].replace('0.0 MB inflated', '%.1f MB inflated' % (rd.inflated_bytes / 1e6))
  native = lambda p_: I.load_episode(p_, meta, True, raw_rgb=True, image_keys=('rgb',))
  native(paths[0])
  host = T._host()
  host.geeco_host_set_fast_inflate(0)
  zlib_1 = round(rate(native, paths[:4], 1), 2)
  host.geeco_host_set_fast_inflate(1)
  out['reader_episodes_per_s'] = {
      'python_reader_1_thread (round 3: tfrecord.py + numpy, holds the GIL)': round(rate(lambda p_: I.load_episode_py(p_, meta, True, raw_rgb=True), paths[:2], 1), 2),
      'native_1_thread_system_zlib': zlib_1,
      'native_1_thread': round(rate(native, paths[:4], 1), 2),
      'native_%d_threads' % threads: round(rate(native, paths, threads), 2)}
  out['reader_frames_per_s_%d_threads' % threads] = round(out['reader_episodes_per_s']['native_%d_threads' % threads] * frames_per_episode, 1)
  log('input_pipeline: reader %s' % out['reader_episodes_per_s'])

  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=K, img_channels=3, batch_size=B))
  params = {'e2evmc_config': cfg, 'log_steps': 1000, 'debug': False}
  e = est.Estimator(est.goal_e2evmc_model_fn, os.path.join(workdir, 'model'), est.RunConfig(), params)
  I.EPISODE_CACHE.clear()
  kw = dict(window_size=K, fetch_target=True, batch_size=B, num_threads=threads, prefetch_size=4, device=dev, device_keys=('rgb',))
  # untimed: model build, eager warm-up steps, hipGraph capture (one episode, not cached)
  e.train(input_fn=lambda: I.pickplace_input_fn(root, 'warmup', 'train', seed=0, cache=False, **kw))
  epochs = []
  import gc
  for ep in range(3):
    hits0 = I.EPISODE_CACHE.hits
    gc.collect()        # a full collection now rather than in the middle of a 0.5 s epoch (the timed loop itself is untouched)
    e.train(input_fn=lambda: I.pickplace_input_fn(root, 'default', 'train', seed=ep, **kw))
    st = e.last_train_stats
    fps = st['steps'] * B * K / st['loop_seconds']
    epochs.append({'epoch': ep + 1, 'steps': st['steps'], 'seconds': round(st['loop_seconds'], 4), 'frames_per_s': round(fps, 1),
                   'ms_per_step': round(st['loop_seconds'] / st['steps'] * 1e3, 3),
                   'episodes_from_hbm_cache': I.EPISODE_CACHE.hits - hits0})
    log('input_pipeline: epoch %d: %d steps in %.3f s = %.0f frames/s' % (ep + 1, st['steps'], st['loop_seconds'], fps))
  feeds = [f for (_, fbuf, _) in e._specs.values() for f in fbuf.values() if hasattr(f, 'pointers')]
  by_address = bool(feeds) and all(f.table is not None and f.buffer is None for f in feeds)
  out['estimator_train'] = {
      'workload': 'geeco-f rgb 256x256 seq_len=%d batch=%d, Estimator.train(pickplace_input_fn(device=cuda, device_keys=(rgb,), '
                  'num_threads=%d)); wall time of the input + step loop (checkpoint write excluded)' % (K, B, threads),
      'window_source': ('uint8 frames of the resident episodes, read by the input kernel through per-sample window addresses '
                        '(geeco_goal_dynimgs_u8_fwd): no fp32 window tensor, no gather launch') if by_address else
                       'float32 windows gathered per batch (geeco_gather_windows)',
      'epochs': epochs, 'synthetic_ms_per_step': round(synthetic_ms_per_step, 3),
      'cached_epoch_vs_synthetic': round(synthetic_ms_per_step / epochs[-1]['ms_per_step'], 4),
      'hbm_cache': {'episodes': len(I.EPISODE_CACHE), 'MB': round(I.EPISODE_CACHE.bytes_in_use / 1e6, 1)}}
  return out, e, root, kw


def inference_report(args, dev, estimator, dataset_root, input_kw, workdir):
  """Predictor latency (reference predictor.py:148-200: one frame in, one command out, batch 1) and Estimator.evaluate
  throughput (train_e2evmc.py:290) on the dataset of the input_pipeline leg (episodes resident in HBM)."""
  import numpy as np
  import torch
  from geeco_amd import input_fn as I
  from geeco_amd import estimator as est
  from geeco_amd.graph import model_variable_shapes
  from geeco_amd.params import create_e2evmc_config
  from geeco_amd.predictor import E2EVMCPredictor, GoalE2EVMCPredictor
  from geeco_amd.variables import VariableStore
  out = {}
  K, B = args.seq_len, args.batch
  if estimator is not None:
    estimator.evaluate(input_fn=lambda: I.pickplace_input_fn(dataset_root, 'warmup', 'train', seed=0, **input_kw))      # build + capture
    t = time.perf_counter()
    res = estimator.evaluate(input_fn=lambda: I.pickplace_input_fn(dataset_root, 'default', 'eval', **input_kw))
    dt = time.perf_counter() - t
    windows = args.pipeline_episodes * (100 - 1 - K + 1)
    out['estimator_evaluate'] = {'workload': 'geeco-f rgb seq_len=%d batch=%d over %d windows (episodes resident in HBM)' % (K, B, windows),
                                 'seconds': round(dt, 4), 'frames_per_s': round(windows * K / dt, 1),
                                 'ms_per_batch': round(dt / -(-windows // B) * 1e3, 3), 'loss': round(res['loss'], 6)}
    log('inference: Estimator.evaluate %.0f frames/s' % out['estimator_evaluate']['frames_per_s'])
  r = np.random.default_rng(0)
  for label, goal, kw, cls in (('GoalE2EVMCPredictor (geeco-f: dynimg + dyndiff)', True, dict(proc_obs='dynimg', proc_tgt='dyndiff'), GoalE2EVMCPredictor),
                               ('E2EVMCPredictor (e2e_vmc: K encoder passes + K LSTM steps)', False, {}, E2EVMCPredictor)):
    md = os.path.join(workdir, 'pred_%d' % goal)
    os.makedirs(md, exist_ok=True)
    cfg = create_e2evmc_config(dict(window_size=K, **kw))
    with open(os.path.join(md, 'e2evmc_config.json'), 'w') as fp:
      json.dump(cfg._asdict(), fp)
    st = VariableStore(model_variable_shapes(cfg, goal), 'cpu')
    st.initialize(seed=0)
    est.save_checkpoint(st, md, keep_max=1)
    pred = cls(md, memcap=None, device=dev)
    frames = r.integers(0, 256, [8, 256, 256, 3]).astype(np.float32) / np.float32(255.0)
    jnt = r.standard_normal([8, 7]).astype(np.float32)
    if goal:
      pred.set_goal(frames[7])
    for i in range(20):
      pred.predict(frames[i % 8], jnt[i % 8])
    lat = []
    for i in range(300):
      t = time.perf_counter()
      pred.predict(frames[i % 8], jnt[i % 8])
      lat.append((time.perf_counter() - t) * 1e3)
    lat.sort()
    out[label] = {'calls': len(lat), 'window_size': K, 'p50_ms': round(percentile(lat, 0.5), 3), 'p90_ms': round(percentile(lat, 0.9), 3),
                  'p99_ms': round(percentile(lat, 0.99), 3), 'min_ms': round(lat[0], 3),
                  'includes': 'range check of the frame, 786 KB frame upload, window shift in HBM, forward hipGraph, '
                              'predictions (+ dynbuff / dyndiff images for the goal model) copied back'}
    log('inference: %s p50 %.3f ms p99 %.3f ms' % (label, out[label]['p50_ms'], out[label]['p99_ms']))
    del pred
    torch.cuda.empty_cache()
  return out


def rccl_debug_on():
  """RCCL's INIT / GRAPH lines into a private file (channel count, rings, transports: what decides whether the early bucket
  really runs beside part 2).  A caller who directs RCCL's log to a file of its own, or asked for more than INFO, is left
  alone; a plain NCCL_DEBUG=WARN / VERSION from the environment is raised to INFO (rccl_info repeats the warnings on stderr)."""
  if 'NCCL_DEBUG_FILE' in os.environ or os.environ.get('NCCL_DEBUG', '').upper() in ('TRACE', 'ABORT'):
    return None
  import tempfile
  path = os.path.join(tempfile.gettempdir(), 'geeco_rccl_%d.log' % os.getpid())
  os.environ['NCCL_DEBUG'] = 'INFO'
  os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH')
  os.environ['NCCL_DEBUG_FILE'] = path
  return path


def rccl_info(path):
  """What RCCL said when the communicator was created: version, collective channels, ring / tree lines (first few), transports."""
  import re
  if not path:
    return {'status': 'the caller directs RCCL\'s log itself: not captured'}
  try:
    with open(path, errors='replace') as f:
      lines = f.read().splitlines()
  except OSError as e:
    return {'status': 'no log: %s' % e}
  finally:
    try:
      os.unlink(path)
    except OSError:
      pass
  out = {'status': 'ok', 'log_lines': len(lines), 'max_nchannels': os.environ.get('NCCL_MAX_NCHANNELS'),
         'graph_register': os.environ.get('NCCL_GRAPH_REGISTER')}
  for l in [l for l in lines if ' WARN ' in l][:10]:
    log('RCCL: ' + l[:300])
  text = '\n'.join(lines)
  m = re.search(r'(\d+) coll channels', text)
  if m:
    out['coll_channels'] = int(m.group(1))
  m = re.search(r'(?:RCCL|NCCL) version[^\n]*', text)
  if m:
    out['version'] = m.group(0)[:120]
  chans = sorted({int(c) for c in re.findall(r'Channel (\d+)(?:/\d+)? *:', text)})
  if chans:
    out['ring_channels_seen'] = len(chans)
  via = sorted(set(re.findall(r'via ([A-Za-z0-9/_-]+)', text)))
  if via:
    out['transports'] = via[:8]
  keep = [l.split('] ', 1)[-1][:160] for l in lines if re.search(r'coll channels|Trees|Ring 0|nChannels|Pattern|minCompCap|Setting affinity', l)]
  out['lines'] = keep[:8]
  return out


DP_MODES = (   # name, TrainStepRunner arguments, skip the exchange.  The three-graph forms FIRST (nothing captures RCCL), then the
    # one-graph forms, then the probe that leaves the replicas diverged (comm_report puts them back in step behind it)
    ('three_graphs_reserve16', dict(overlap=True, capture_exchange=False, reserved_cus=16), False),   # = runtime.DP_FORM_DEFAULT
    ('three_graphs_overlap', dict(overlap=True, capture_exchange=False), False),   # = runtime.DP_FORMS['three_graphs']
    ('three_graphs_serial', dict(overlap=False, capture_exchange=False), False),
    ('two_graphs', dict(overlap=True, capture_exchange=False, eager_adam=True), False),      # = runtime.DP_FORMS['two_graphs']
    ('two_graphs_reserve16', dict(overlap=True, capture_exchange=False, eager_adam=True, reserved_cus=16), False),
    ('two_graphs_serial', dict(overlap=False, capture_exchange=False, eager_adam=True), False),
    ('overlap', dict(overlap=True, capture_exchange=True), False),                 # ONE graph, exchange captured
    ('serial', dict(overlap=False, capture_exchange=True), False),
    ('overlap_reserve%d' % DP_RESERVE_PROBE, dict(overlap=True, capture_exchange=True, reserved_cus=DP_RESERVE_PROBE), False),
    ('overlap_reserve%d' % (2 * DP_RESERVE_PROBE), dict(overlap=True, capture_exchange=True, reserved_cus=2 * DP_RESERVE_PROBE), False),
    ('no_exchange', dict(overlap=True, capture_exchange=False), True),        # three graphs, no all-reduce: the graph gaps alone
)


def dp_step_modes(args, model, timer):
  """ms per step of the data-parallel step in every form the runner has (each one its own runner and its own captured
  graphs on the same model and batch).  ``timer(fn)`` -> (ms, p10, p90).  The no_exchange form runs LAST: at N > 1 it
  applies Adam to un-reduced gradients and the caller must put the replicas back in step afterwards."""
  import torch
  from geeco_amd.runtime import TrainStepRunner
  modes, forms, info = {}, {}, None
  for name, kw, skip in DP_MODES:
    r = TrainStepRunner(model, use_graph=not args.no_graph, warmup=2, dp=True, **kw)
    r.skip_allreduce = skip
    r.prepare()
    for _ in range(3):
      r.step()
    torch.cuda.synchronize()
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
      torch.distributed.barrier()
    modes[name] = timer(r.step)
    forms[name] = r.bucket_info()['graphs_per_step']
    if name == 'overlap':
      info = r.bucket_info()
    log('dp form %-22s %.4f ms/step (%d graph(s) per step)' % (name, modes[name][0], forms[name]))
    del r
  return modes, forms, info


def comm_report(args, model, runner, dev, world, step_ms):
  """N > 1 (every rank takes part): the exchange alone, and the step in every form of the exchange (one graph with the
  exchange captured / three graphs with the exchange between them, each overlapped and serial), with CUs reserved for RCCL
  and without any exchange, so that one multi-GPU call shows what each choice is worth on that node."""
  import torch
  from geeco_amd import dist as gdist
  iters = max(5, min(args.steps, 20))

  def exchange():
    for w in runner._exchange_early() + runner._exchange_late():
      w.wait()
  exchange()
  torch.distributed.barrier()
  ar_ms = gdist.max_over_ranks(time_region(exchange, iters), dev)

  def early_only():
    for w in runner._exchange_early():
      w.wait()

  def late_only():
    for w in runner._exchange_late():
      w.wait()
  ar_each = {'early': round(gdist.max_over_ranks(time_region(early_only, iters), dev), 4),
             'late': round(gdist.max_over_ranks(time_region(late_only, iters), dev), 4)}
  modes, forms, _ = dp_step_modes(args, model, lambda fn: (gdist.max_over_ranks(time_region(fn, iters), dev), None, None))
  modes = {k: round(v[0], 4) for k, v in modes.items()}
  # the no_exchange probe applied Adam to UN-reduced per-rank gradients (the batches differ per rank): the replicas have diverged.
  # Put them back in step before anything else runs on this model: parameters and optimiser state from rank 0.
  torch.cuda.synchronize()
  st = model.store
  for buf in (st.params, st.adam_m, st.adam_v):
    torch.distributed.broadcast(buf, src=0)
  torch.distributed.broadcast(st.global_step, src=0)
  torch.cuda.synchronize()
  if getattr(model, 'enc', None) is not None:
    model.enc.refresh_derived()
  wire = 4 * sum(n for _, n in runner.early_calls) + (runner.staging.numel() * 4 if runner.staging is not None else 0)
  timed = ('overlap' if runner.overlap else 'serial') if runner.capture_exchange else \
          ('three_graphs_overlap' if runner.overlap else 'three_graphs_serial')
  if runner.reserved_cus:
    timed += '_reserve%d' % runner.reserved_cus
  rk = 'overlap_reserve%d' % DP_RESERVE_PROBE
  return {'mode': 'overlap' if runner.overlap else 'serial', 'timed_form': timed, 'graphs_per_step': forms,
          'allreduce_ms': round(ar_ms, 4), 'allreduce_ms_each_bucket_alone': ar_each, 'allreduce_bytes': int(model.store.grads.numel() * 4),
          'allreduce_bytes_on_the_wire': int(wire),
          'bus_GB/s': round(2.0 * (world - 1) / world * wire / (ar_ms * 1e-3) / 1e9, 1),
          'step_ms': modes, 'step_ms_without_allreduce': modes['no_exchange'],
          'allreduce_exposed_ms': round(max(step_ms - modes['no_exchange'], 0.0), 4),
          'overlap_gain_ms': round(modes['serial'] - modes['overlap'], 4),
          'one_graph_gain_ms': round(modes['three_graphs_overlap'] - modes['overlap'], 4),
          'reserve_gain_ms': round(modes['overlap'] - modes[rk], 4), 'buckets': runner.bucket_info()}


def dp_one_rank_report(args, model, runner, dev):
  """world == 1: the FIXED cost of the data-parallel form of the step, on this GPU, in this process.  A ONE-rank RCCL group
  is formed in-process (the sum over one rank is the identity), and the SAME batch goes through the step exactly as every
  rank of an N-GPU run executes it (runtime.TrainStepRunner(dp=True)): the 30.2 MB early bucket and the 177 KB late bucket as
  real RCCL all-reduce launches, in every form the runner has -- ONE graph with the exchange captured (what an N > 1 run
  times; exchange overlapped with part 2 / serial), three graphs with the exchange launched between them (round 4's form),
  and three graphs with no exchange (what the two inter-graph gaps alone cost).  ``delta_vs_single_graph_us`` is everything
  an N-GPU step pays on top of the single-GPU step EXCEPT wire time, so single / (dp step + exposed wire time) bounds the
  scaling efficiency.  Every number is the median of `samples` HIP-event pairs, each around 5 back-to-back steps."""
  import torch
  from geeco_amd import dist as gdist
  if gdist.group_active():
    return {'status': 'skipped: a process group is already active'}
  samples = max(10, min(args.steps, 30))
  with socket.socket() as s_:
    s_.bind(('127.0.0.1', 0))
    port = s_.getsockname()[1]
  saved = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK=str(dev.index or 0), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  out = {}
  try:
    for _ in range(3):
      runner.step()
    single = time_launches(runner.step, samples)
    rccl_log = rccl_debug_on()
    assert gdist.init_from_env('nccl', device_index=dev.index or 0, single_rank_group=True) == 1 and gdist.group_active()
    out['backend'] = torch.distributed.get_backend()
    modes, forms, info = dp_step_modes(args, model, lambda fn: time_launches(fn, samples))
    single2 = time_launches(runner.step, samples)           # the single graph again, now with RCCL's threads alive
    from geeco_amd.runtime import TrainStepRunner
    r_dp = TrainStepRunner(model, use_graph=False, dp=True)

    def early():
      for w in r_dp._exchange_early():
        w.wait()

    def late():
      for w in r_dp._exchange_late():
        w.wait()
    early(), late()
    ar_early, ar_late = time_launches(early, samples), time_launches(late, samples)
    us = lambda t: round(t[0] * 1e3, 1)
    one = min(single[0], single2[0])
    out.update({
        'status': 'ok', 'graphs_per_step': forms, 'buckets': info,
        'single_graph_ms': round(single[0], 4), 'single_graph_ms_with_rccl_threads': round(single2[0], 4),
        'ms_per_step': {k: round(v[0], 4) for k, v in modes.items()},
        'ms_per_step_p10_p90': {k: [round(v[1], 4), round(v[2], 4)] for k, v in modes.items()},
        'delta_vs_single_graph_us': {k: round((v[0] - one) * 1e3, 1) for k, v in modes.items()},
        'allreduce_us_one_rank': {'early_%d_bytes' % (4 * sum(n for _, n in r_dp.early_calls)): us(ar_early),
                                  'late_%d_bytes' % (4 * r_dp.staging.numel() if r_dp.staging is not None else 0): us(ar_late)},
        'efficiency_bound_wire_hidden': round(one / modes['overlap'][0], 4),
        'rccl': rccl_info(rccl_log),
        'note': 'one rank: RCCL launches both all-reduces for real (launch floor) but moves no bytes over xGMI; at N > 1 add the '
                'exposed wire time of the late bucket (and of the early one if it outlasts part 2, ~0.85 ms of kernels)'})
    del r_dp
  except Exception as e:        # this leg must never take the headline line with it
    out.update({'status': 'failed: %s: %s' % (type(e).__name__, str(e)[:300])})
  finally:
    try:
      torch.cuda.synchronize()
      if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    except Exception as e:
      out['destroy'] = 'failed: %s' % str(e)[:200]
    for k, v in saved.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v
  return out


def main():
  args = parse_args()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(spawn_ranks(args))

  import torch
  from geeco_amd import dist as gdist

  # stdout carries exactly ONE line, the JSON: libraries that print to fd 1 on their own (RCCL writes its version banner
  # there when a communicator is created) go to stderr for the rest of the run
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)

  rccl_log = rccl_debug_on() if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None
  world = gdist.init_from_env('nccl')      # a launcher that formed the group already (tests/_dp_launch.py) is respected
  rank = gdist.rank()
  if world != args.gpus:
    log('--gpus %d but the process group has %d rank(s): refusing to report a number for the wrong N' % (args.gpus, world))
    sys.exit(3)
  local = torch.cuda.current_device() if gdist.group_active() else int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)
  # N ranks must sit on N distinct GPUs: two ranks on one device would report a number for the wrong machine
  ident, detail = device_identity(local)
  idents = gdist.gather_strings(ident if ident is not None else '?', dev)
  details = gdist.gather_strings(detail, dev, width=160)
  if '?' in idents:
    log('WARNING: no PCI address / UUID available for at least one rank: cannot verify that the %d ranks sit on distinct GPUs' % world)
  shared = '?' not in idents and len(set(idents)) < world
  if shared and not args.allow_shared_gpu:
    log('%d ranks but only %d distinct GPU(s): %s' % (world, len(set(idents)), idents))
    sys.exit(3)

  cfg, model = build_model(args.model, args.channels, args.seq_len, args.batch, dev)
  model.store.initialize(seed=0)
  gdist.broadcast_variables(model.store)
  synthetic_batch(model, 1234 + rank)
  log('model built: %d parameters, batch %d/GPU, world %d' % (model.store.count_parameters(), args.batch, world))

  def headline(r, form):
    """The JSON line of one full measurement (every value in it is already on this rank: nothing here talks to another rank)."""
    ms_step = r['dt'] / args.steps * 1e3
    frames = world * args.batch * args.seq_len
    info = r['runner'].bucket_info()
    out = {
        'metric': 'train-step frames/sec (256x256 %s, seq_len=%d)' % ('RGB' if args.channels == 3 else 'RGB-D', args.seq_len),
        'value': round(frames * args.steps / r['dt'], 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic' if not shared else 'synthetic (REHEARSAL: ranks share one GPU)',
        'config': {'workload': '%s %s %dx%d seq_len=%d batch=%d/GPU (global %d), fwd+bwd+allreduce+Adam' %
                               (args.model, 'rgb' if args.channels == 3 else 'rgbd', cfg.img_height, cfg.img_width,
                                args.seq_len, args.batch, world * args.batch),
                   'parallelism': 'dp%d' % world, 'hipgraph': not args.no_graph, 'graphs_per_step': info['graphs_per_step'],
                   'params': model.store.count_parameters()},
        'step_ms': {'median': round(percentile(r['per_step'], 0.5), 4), 'p10': round(percentile(r['per_step'], 0.1), 4),
                    'p90': round(percentile(r['per_step'], 0.9), 4), 'timer': 'HIP events per step, rank 0'},
        'final_loss': round(r['final_loss'], 6),
    }
    if world > 1:
      out['config']['dp_form'] = form
      out['ranks'] = {'ms_per_step': [round(v, 4) for v in r['rank_ms']], 'devices': details, 'distinct_devices': len(set(idents))}
    chk, ok = check_losses(args, first_loss, r['final_loss'], r['total_steps'])
    out['loss_check'] = chk
    return out, ok

  def emit(out):
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + '\n').encode())

  # ---- the timed region -----------------------------------------------------------------------------------------------------
  # N > 1: the ALWAYS-SAFE form first, in full -- three replayed graphs with both all-reduces as ordinary RCCL launches between
  # them -- and its figure on stderr at once; only then anything that captures RCCL into a graph (see Watchdog).
  safe_form = 'three_graphs_serial' if args.dp_serial else 'three_graphs_reserve16'      # = runtime.DP_FORM_DEFAULT, what Estimator.train runs
  r = timed_steps(model, args.steps, args.warmup, not args.no_graph, world, dev, form=safe_form)
  first_loss = r['first_loss']
  form, full, trial, dog, comm, identical = safe_form, {}, {}, None, None, {}
  log('timed region: %d steps in %.3f s' % (args.steps, r['dt']))
  if world > 1:
    from geeco_amd.runtime import DP_CANDIDATES_CAPTURED, DP_CANDIDATES_SAFE, pick_dp_runner
    ms = r['dt'] / args.steps * 1e3
    full[safe_form] = round(ms, 4)
    # the replicas must be bitwise identical after any number of steps: the cheapest check there is that the exchange did its job
    identical = {safe_form: gdist.replicas_identical(model.store)}
    if not identical[safe_form]:
      log('ERROR: the replicas differ after the safe form\'s steps: the gradient exchange is broken; the line below reports it')
    log('N = %d, safe form (%s: exchange launched between three graphs): %.4f ms/step = %.1f frames/s (max over ranks, %d timed steps)'
        % (world, safe_form, ms, world * args.batch * args.seq_len * args.steps / r['dt'], args.steps))

    def on_watchdog(provisional, why):
      if rank == 0 and provisional is not None:
        provisional['comm'] = {'status': 'WATCHDOG: %s; this line is the full measurement of config.dp_form' % why,
                               'forms_timed_in_full_ms': full, 'trial_ms': trial}
        emit(provisional)

    def abort_flag(raise_it=False):
      store = torch.distributed.distributed_c10d._get_default_store()
      return int(store.add('bench_abort', 1 if raise_it else 0)) > 0
    dog = Watchdog(args.dp_watchdog_s, on_watchdog, aborted=abort_flag)
    dog.provisional = headline(r, form)[0] if rank == 0 else None
    if args.dp_watchdog_s > 0:
      dog.arm()
    try:
      if not (args.dp_fixed or args.dp_serial or args.no_graph):
        # two trials, the second only behind the first: (1) the other forms in which every collective is an ordinary launch (early
        # bucket beside part 2 with and without CUs left to RCCL, or behind it; the optimiser's pieces eager): the best of them in full,
        # so that the best SAFE number is on record before (2) anything captures RCCL into a graph
        for label, cands in (('safe forms', DP_CANDIDATES_SAFE), ('one-graph forms', DP_CANDIDATES_CAPTURED)):
          dog.phase = 'trial of the %s' % label
          cand, t_ = pick_dp_runner(model, use_graph=True, candidates=cands, log=log)
          trial.update(t_)
          best = min(t_, key=t_.get)
          cur_ms = r['dt'] / args.steps * 1e3
          log('%s, short trial: %s; fastest %s (the form on record: %s, %.4f ms)' %
              (label, json.dumps({k: round(v, 4) for k, v in t_.items()}), best, form, cur_ms))
          if t_[best] < cur_ms:
            dog.phase = 'timed region of %s' % best
            r2 = timed_region(model, cand, args.steps, args.warmup, world, dev)
            r2['first_loss'] = first_loss
            # (a form on a backend that cannot capture -- gloo rehearsal -- is the three-graph form under another name: bucket_info says so)
            full[best] = round(r2['dt'] / args.steps * 1e3, 4)
            log('N = %d, form %s: %.4f ms/step in full' % (world, best, full[best]))
            identical[best] = gdist.replicas_identical(model.store)
            if not identical[best]:
              log('ERROR: the replicas differ after the steps of form %s: its measurement is discarded' % best)
            if r2['dt'] < r['dt'] and identical[best]:
              r, form = r2, best
              if rank == 0:
                dog.provisional = headline(r, form)[0]
          del cand
      if not args.skip_comm_report:
        dog.phase = 'comm report'
        comm = comm_report(args, model, r['runner'], dev, world, r['dt'] / args.steps * 1e3)
    except Exception as e:      # a capture that failed (runtime.CaptureFailed) or anything else behind the safe form's measurement
      log('%s: %s' % (type(e).__name__, str(e)[:500]))
      try:
        abort_flag(raise_it=True)      # the peers are (or will be) waiting in a collective this rank will not join
      except Exception:
        pass
      dog.fire('%s on rank %d' % (type(e).__name__, rank))
    dog.disarm()
  runner, dt, per_step = r['runner'], r['dt'], r['per_step']
  loss, total_steps = r['final_loss'], r['total_steps']

  rc = 0
  if rank == 0:
    ms_step = dt / args.steps * 1e3
    out, ok = headline(r, form)
    if not ok:
      rc = 4
    if world > 1 and not all(identical.values()):
      rc = 5
    if world > 1:
      comm = comm or {'status': 'skipped'}
      comm['timed_form'] = form
      comm['replicas_bit_identical_after'] = identical
      comm['forms_timed_in_full_ms'] = full
      comm['trial_ms'] = {k: round(v, 4) for k, v in trial.items()} or None
      comm['order'] = ('safe form (%s) in full -> short trial of the other safe forms, the fastest in full if it beats the form on record -> the '
                       'same for the one-graph forms -> value = the fastest full measurement -> comm report' % safe_form)
      comm['rccl'] = rccl_info(rccl_log)
      out['comm'] = comm
    if not args.skip_layers:      # rank 0's GPU alone, after the timed region (any N: the per-GPU work is the same)
      samples = max(30, min(args.steps, 50))
      rows = layer_table(model, samples)
      out['roofline'] = dominant_roofline(rows)
      for row in rows:
        del row['kernels']
      out['layers'] = rows
      out['hbm'] = hbm_table(model, args, samples)
      tf_, ms_enc = encoder_forward_tflops(model, samples, args.channels)
      out['encoder_forward'] = {'tflops': round(tf_, 2), 'frac_of_f32_mfma_peak': round(tf_ / PEAK_F32_MFMA_TFLOPS, 4),
                                'ms': round(ms_enc, 3), 'frames': model.enc.G * model.enc.Nf}
      out['step_frac_of_f32_mfma_peak'] = round(step_flop(args.channels, model.enc.G * model.enc.Nf) / (ms_step * 1e-3) / 1e12
                                                / PEAK_F32_MFMA_TFLOPS, 4)
    xdog = None
    if world == 1:
      # everything the contract asks of the line is in `out` from here on; the legs below only add to it
      if not args.skip_cpu:
        out['cpu_baseline'] = cpu_baseline(args)

      def on_extras(provisional, why):
        provisional['extras'] = 'WATCHDOG: %s; the legs after it are missing from this line' % why
        emit(provisional)
      xdog = Watchdog(args.extras_watchdog_s, on_extras, keeps='the line as it stood before that leg', after='the legs after the headline')
      xdog.provisional = dict(out)
      if args.extras_watchdog_s > 0:
        xdog.arm()
    if world == 1 and not args.skip_dp_one_rank:
      xdog.phase = 'dp_one_rank'
      out['dp_one_rank'] = dp_one_rank_report(args, model, runner, dev)
      xdog.provisional = dict(out)
      log('dp_one_rank: %s' % json.dumps({k: out['dp_one_rank'].get(k) for k in ('status', 'single_graph_ms', 'ms_per_step', 'delta_vs_single_graph_us', 'allreduce_us_one_rank')}))
    if world == 1 and not args.skip_other_configs:
      del runner, r
      xdog.phase = 'other_configs'
      out['other_configs'], ok2 = other_configs(args, dev)
      if not ok2:
        rc = 4
      xdog.provisional = dict(out)
    if world == 1 and not (args.skip_input_pipeline and args.skip_inference) and args.model == 'geeco-f' and args.channels == 3:
      import shutil
      import tempfile
      model = None
      torch.cuda.empty_cache()
      workdir = tempfile.mkdtemp(prefix='geeco_bench_')
      try:
        e = root = kw = None
        if not args.skip_input_pipeline:
          xdog.phase = 'input_pipeline'
          out['input_pipeline'], e, root, kw = input_pipeline_report(args, dev, ms_step, workdir)
          xdog.provisional = dict(out)
        if not args.skip_inference:
          xdog.phase = 'inference'
          out['inference'] = inference_report(args, dev, e, root, kw, workdir)
      finally:
        shutil.rmtree(workdir, ignore_errors=True)
    if xdog is not None:
      xdog.disarm()
    emit(out)
  # N > 1: ranks 1..N-1 have nothing to do after the timed region and the comm report; they wait here ON THE HOST (no
  # collective pending, GPUs idle) until rank 0 has finished its tables and printed the line, then all ranks leave together
  gdist.host_rendezvous('bench_done')
  if rccl_log and rank != 0:
    try:
      os.unlink(rccl_log)
    except OSError:
      pass
  if torch.distributed.is_available() and torch.distributed.is_initialized():
    torch.cuda.synchronize()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
  sys.exit(rc)


if __name__ == '__main__':
  main()
