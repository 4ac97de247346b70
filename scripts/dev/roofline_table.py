#!/usr/bin/env python
"""Joins a bench.py JSON line (``layers``) with a per-dispatch PMC report (scripts/dev/pmc_report.py) of ONE step into
the per-launch roofline table kept under profiles/rNN/.

  python scripts/dev/roofline_table.py profiles/r02/d_bench.json profiles/r02/d_pmc.txt d > profiles/r02/d_roofline_table.md

The conv dispatches of a step come in a fixed order (forward conv1..conv8, then per layer from the top: wgrad, dgrad;
conv2's dgrad is the fused conv2-dgrad + conv1-wgrad launch), which is how the two files are matched.
"""
import json
import os
import sys

PEAK = 157.3
SKIP = ('conv_splitk_epilogue', 'wgrad_reduce')


def step_order(layers):
  order = [('conv%d' % l, 'fwd') for l in range(1, 9)]
  top = ('conv7+conv8', 'dgrad7+wgrad7+wgrad8')
  if top in layers:     # round 4: conv7's input gradient + conv7's / conv8's filter gradients are ONE heterogeneous launch
    order += [('conv8', 'dgrad'), top]
    first = 6
  else:
    first = 8
  for l in range(first, 2, -1):
    order += [('conv%d' % l, 'wgrad'), ('conv%d' % l, 'dgrad')]
  return order + [('conv2', 'wgrad'), ('conv2', 'dgrad+conv1_wgrad')]


def pmc_rows(path):
  rows = []
  for line in open(path):
    if not line.startswith('conv') or line.startswith(SKIP):
      continue
    # the name column is cut at 34 characters and may contain blanks; the 10 numeric columns are the tail
    cols = line.split()
    nums = [float(c) for c in cols[-10:]]
    rows.append(dict(us=nums[0], mfma=nums[1], ldsbc=nums[5], rd=nums[6], wr=nums[7]))
  return rows


def envelope():
  """(traffic TB/s, TFLOP/s) points of the newest committed MFMA x HBM envelope (scripts/dev/ub/mfma_envelope.hip), or None."""
  root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'profiles')
  for rnd in sorted(os.listdir(root), reverse=True):
    path = os.path.join(root, rnd, 'ub_mfma_envelope.json')
    if os.path.exists(path):
      return json.load(open(path))['points'], 'profiles/%s/ub_mfma_envelope.json' % rnd
  return None, None


def attainable(points, tbs):
  """Piecewise-linear in the traffic; flat beyond the last measured point (3.9 TB/s: nothing in the step moves more beside MFMAs)."""
  if tbs <= points[0][0]:
    return points[0][1]
  for (x0, y0), (x1, y1) in zip(points, points[1:]):
    if tbs <= x1:
      return y0 + (y1 - y0) * (tbs - x0) / (x1 - x0)
  return points[-1][1]


def main(bench_path, pmc_path, tag):
  d = json.loads(open(bench_path).read().strip().splitlines()[-1])
  layers = {(r['layer'], r['op']): r for r in d['layers']}
  order, pmc = step_order(layers), pmc_rows(pmc_path)
  assert len(pmc) == len(order), (len(pmc), len(order))
  cfg = d['config']
  print('# Per-launch roofline table of one training step (set `%s_*`)\n' % tag)
  print('Workload: %s (%d encoder frames per launch).  `us alone`, TFLOP/s and %% of the %.1f TFLOP/s fp32 MFMA peak: '
        '`%s_bench.json` (`layers`: each launch timed alone with HIP events).  `us PMC`, MFMA-busy %%, LDS bank-conflict '
        'share, HBM read / written MB per launch: `%s_pmc.txt` (rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per '
        'MI355X_MICROARCH.md; PMC passes run 3-8 %% slower than unprofiled ones).\n'
        % (cfg['workload'], d['encoder_forward']['frames'], PEAK, tag, tag))
  env, env_src = envelope()
  if env:
    print('`attainable`: what a loop of the product kernels\' SHAPE sustains beside this launch\'s HBM traffic ((read + written) MB / us '
          'alone), interpolated from `%s` -- persistent block per CU, LDS-DMA loaders, resident B operands, one barrier per tile, no '
          'index arithmetic; `of att.` = TFLOP/s / attainable.  It is the bound for the big persistent kernels (conv1-5); the launches '
          'from conv6 up are latency-bound (too few tiles per CU for any steady state) and their `of att.` says how far.\n' % env_src)
  print('| launch | kernel | GFLOP | us alone | TFLOP/s | % of peak | attainable TFLOP/s | of att. % | us PMC | MFMA busy % | LDS conflict % '
        '| HBM read MB | HBM written MB |')
  print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
  total = 0.0
  for key, p in zip(order, pmc):
    r = layers[key]
    total += r['us']
    att = attainable(env, (p['rd'] + p['wr']) / r['us']) if env else None      # MB / us = TB/s
    print('| %s %s | `%s` | %.2f | %.1f | %.1f | %.1f | %s | %s | %.1f | %.1f | %.1f | %.1f | %.1f |'
          % (key[0], key[1], r['kernel'], r['flop'] / 1e9, r['us'], r['tflops'], 100 * r['frac'],
             '%.1f' % att if att else '-', '%.1f' % (100 * r['tflops'] / att) if att else '-', p['us'], p['mfma'],
             p['ldsbc'], p['rd'], p['wr']))
  print('\nSum of the conv launches alone: %.0f us of a %.0f us step (%.1f k frames/s); the rest: input stage, decoder '
        'chain, slab reduces / split-K epilogues, Adam (`%s_step_trace.txt`).'
        % (total, 1e3 * d['ms_per_step'], d['value'] / 1e3, tag))


if __name__ == '__main__':
  main(*sys.argv[1:4])
