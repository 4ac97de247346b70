#!/usr/bin/env python
"""Training script for the end-to-end visuomotor controllers on MI355X.

Drop-in counterpart of the reference's ``scripts/train_e2evmc.py``: identical command line
(:22-124), the same ``model_dir`` protocol (``<ts>-runcmd.json``, ``e2evmc_config.json`` that wins
over the CLI on restart :229-232, ``model.ckpt-<step>*`` + ``checkpoint``, best-k ``snapshots/`` with
``snapshot_index.json`` :143-205) and the same loop: per epoch train -> evaluate -> export snapshot
(:288-291).  Only the imports differ: ``geeco_amd.estimator`` instead of ``tf.estimator``.

Data parallel: launch with ``python -m torch.distributed.run --nproc-per-node N scripts/train_e2evmc.py ...``;
``--batch_size`` stays the GLOBAL batch: each rank reads its own rank-strided share of the episodes in batches of
batch_size / world windows (geeco_amd.input_fn.pickplace_input_fn(shard=...)).
``--dataset_dir synthetic:<num_batches>`` trains on seeded synthetic windows (no dataset on disk).
"""
import argparse
import json
import os
import pprint
import re
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from geeco_amd import _dev                                               # noqa: E402
from geeco_amd import dist as gdist                                      # noqa: E402
from geeco_amd import estimator as est                                   # noqa: E402
from geeco_amd.estimator import e2evmc_model_fn, goal_e2evmc_model_fn    # noqa: E402
from geeco_amd.input_fn import pickplace_input_fn                        # noqa: E402
from geeco_amd.params import create_e2evmc_config                        # noqa: E402
from geeco_amd.utils import load_model_config, save_model_config, save_run_command  # noqa: E402

# ---------- command line arguments (train_e2evmc.py:22-124) ----------

ARGPARSER = argparse.ArgumentParser(description='Train E2E VMC.')
_ARGS = [
    # directories
    ('--dataset_dir', str, '../data/gym-pick-pad2-cube2-v4', 'The path to the dataset (needs to conform with gym_provider).'),
    ('--split_name', str, 'default', 'The name of the data split to be used.'),
    ('--model_dir', str, '../tmp/models/geeco-f', 'The directory where the model will be stored.'),
    # model
    ('--observation_format', str, 'rgb', 'Observation data to be used (sets img_channels): rgb | rgbd.'),
    ('--control_mode', str, 'cartesian', 'Control mode of the robot: cartesian | velocity.'),
    ('--goal_condition', str, 'none', 'Conditioning mode of the reflex: none | target.'),
    ('--window_size', int, 4, 'The number of frames to process before making prediction.'),
    ('--dim_h_lstm', int, 128, 'Hidden state dimension of the LSTM.'),
    ('--dim_h_fc', int, 128, 'Output dimension of the LSTM (before decoding heads).'),
    ('--dim_s_obs', int, 256, 'Output dimension of the observation encoding.'),
    ('--dim_s_dyn', int, 256, 'Output dimension of the dynamics encoding.'),
    ('--dim_s_diff', int, 256, 'Output dimension of the target difference encoding.'),
    ('--proc_obs', str, 'sequence', 'The processing type of the frame buffer: sequence | dynimg'),
    ('--proc_tgt', str, 'constant', 'The processing type of the target frame: constant | residual | dyndiff'),
    ('--l2_regularizer', float, 0.0, 'The weight of the L2 weight regularizer. Zero disables weight regularization.'),
    ('--lambda_aux', float, 1.0, 'The weight of the auxiliary pose prediction losses. Zero disables them.'),
    # data
    ('--data_encoding', str, 'v4', 'Version of the data encoding. Available: v1 | v2 | v3 | v4'),
    # training
    ('--lr', float, 1e-4, 'The learning rate of the ADAM solver.'),
    ('--train_epochs', int, 10, 'The number of epochs to train.'),
    # snapshots
    ('--ckpt_steps', int, 10000, 'Number of steps between checkpoint saves.'),
    ('--num_last_ckpt', int, 2, 'Number of last snapshots to keep.'),
    ('--num_best_ckpt', int, 5, 'Number of best performing snapshots to keep.'),
    # memory / input threads
    ('--batch_size', int, 32, 'The number of data points (windows) per batch.'),
    ('--memcap', float, 0.8, 'Maximum fraction of memory to allocate per GPU.'),
    ('--num_threads', int, None, 'How many parallel threads to run for data fetching (reference default: 4; here, when the '
                                 'flag is absent: this rank\'s share of the host cores, input_fn.default_reader_threads).'),
    ('--prefetch_size', int, 4, 'How many batches to prefetch.'),
    ('--shuffle_buffer', int, 64, 'Number of shuffled examples to draw minibatch from.'),
    # logging
    ('--log_steps', int, 1000, 'Global steps between log output.'),
]
for _flag, _type, _default, _help in _ARGS:
  ARGPARSER.add_argument(_flag, type=_type, default=_default, help=_help)
ARGPARSER.add_argument('--debug', default=False, action='store_true', help='Enables debugging mode.')
ARGPARSER.add_argument('--initial_eval', default=False, action='store_true',
                       help='Runs an evaluation before the first training iteration.')
# the one flag the reference does not have (it has no data parallelism, train_e2evmc.py:221-224, 260-264)
ARGPARSER.add_argument('--dp_form', type=str, default=None,
                       help='Data parallel only: form of the step, one of three_graphs_reserve16 (default: exchange launched between three '
                            'replayed hipGraphs, 16 CUs left to RCCL) | three_graphs | three_graphs_serial | two_graphs (the optimiser launched eagerly behind two graphs; also two_graphs_reserve16, two_graphs_serial) | overlap (whole step '
                            'incl. both all-reduces as ONE hipGraph) '
                            '| overlap_reserve16 | overlap_reserve32 | serial.  bench.py --gpus N reports which is fastest on a node.')

_OBSERVATION_FORMAT_TO_CHANNELS = {'rgb': 3, 'rgbd': 4}                      # train_e2evmc.py:129-132
_GOAL_CONDITION_TO_MODEL = {'none': (e2evmc_model_fn, 'VMC'),               # train_e2evmc.py:134-137
                            'target': (goal_e2evmc_model_fn, 'GoalVMC')}


def _latest_by_ctime(model_dir, suffix):
  files = [os.path.join(model_dir, fn) for fn in os.listdir(model_dir) if fn.endswith(suffix)]
  return max(files, key=lambda fn: os.stat(fn).st_ctime)


def _export_snapshot(model_dir, eval_results, num_best_ckpt):
  """Keeps the ``num_best_ckpt`` best checkpoints by eval loss under <model_dir>/snapshots/
  (train_e2evmc.py:143-205)."""
  snapshots_dir = os.path.join(model_dir, 'snapshots')
  os.makedirs(snapshots_dir, exist_ok=True)
  index_file = os.path.join(snapshots_dir, 'snapshot_index.json')
  index = {}
  if os.path.exists(index_file):
    with open(index_file, 'r') as fp:
      index = json.load(fp)
  print('>>> Current snapshot index contains %d entries.' % len(index))
  ckpt_name = os.path.basename(est.latest_checkpoint(model_dir))
  step = int(re.search(r'\d+', ckpt_name).group(0))
  loss = float(eval_results['loss'])
  ckpt_dir = os.path.join(snapshots_dir, ckpt_name)
  os.makedirs(ckpt_dir, exist_ok=True)
  for cfg in (_latest_by_ctime(model_dir, 'runcmd.json'), _latest_by_ctime(model_dir, 'config.json')):
    shutil.copy(src=cfg, dst=ckpt_dir)
  for fn in os.listdir(model_dir):
    if fn.startswith(ckpt_name) and os.path.isfile(os.path.join(model_dir, fn)):
      shutil.copy(src=os.path.join(model_dir, fn), dst=ckpt_dir)
  with open(os.path.join(ckpt_dir, 'checkpoint'), 'w') as fp:
    fp.write('model_checkpoint_path: "%s"\n' % ckpt_name)
  print('>>> Exported current checkpoint (step=%d; loss=%.06f) to %s.' % (step, loss, ckpt_dir))
  index[ckpt_name] = {'step': step, 'loss': loss, 'dir': ckpt_dir}
  if len(index) > num_best_ckpt:
    worst = max(index.items(), key=lambda kv: kv[1]['loss'])[0]
    shutil.rmtree(index[worst]['dir'])
    info = index.pop(worst)
    print('>>> Removed worst snapshot (step=%d; loss=%.06f): %s' % (info['step'], info['loss'], info['dir']))
  with open(index_file, 'w') as fp:
    json.dump(index, fp, indent=2, sort_keys=True)
  print('>>> Saved snapshot index: %s' % index_file)
  return ckpt_dir


def main(args):
  gdist.init_from_env()
  rank, world = gdist.rank(), gdist.world_size()
  os.makedirs(name=args.model_dir, exist_ok=True)
  if rank == 0:
    save_run_command(argparser=ARGPARSER, run_dir=args.model_dir)
  gpu_options = est.GPUOptions(allow_growth=True, per_process_gpu_memory_fraction=args.memcap)
  run_config = est.RunConfig(session_config=est.ConfigProto(gpu_options=gpu_options),
                             save_checkpoints_steps=args.ckpt_steps, keep_checkpoint_max=args.num_last_ckpt,
                             # checkpoints are also written as TF-1.15 tensor bundles (model.ckpt-<step>.index / .data-*),
                             # the files the reference's predictor and snapshot tooling read
                             save_tf_bundle=_dev.env('GEECO_NO_TF_BUNDLE') is None, dp_form=args.dp_form)
  config_name = 'e2evmc_config'
  config_path = os.path.join(args.model_dir, '%s.json' % config_name)
  if os.path.exists(config_path):    # a previous run's config wins over the CLI (train_e2evmc.py:229-232)
    e2evmc_config = create_e2evmc_config(load_model_config(args.model_dir, config_name))
    print('>>> Loaded existing model config from %s' % (config_path,))
  else:
    e2evmc_config = create_e2evmc_config({
        'img_channels': _OBSERVATION_FORMAT_TO_CHANNELS[args.observation_format],
        'control_mode': args.control_mode, 'window_size': args.window_size, 'dim_h_lstm': args.dim_h_lstm,
        'dim_h_fc': args.dim_h_fc, 'dim_s_obs': args.dim_s_obs, 'dim_s_dyn': args.dim_s_dyn,
        'dim_s_diff': args.dim_s_diff, 'proc_obs': args.proc_obs, 'proc_tgt': args.proc_tgt,
        'l2_regularizer': args.l2_regularizer, 'lambda_aux': args.lambda_aux, 'batch_size': args.batch_size,
        'lr': args.lr})
    if rank == 0:
      save_model_config(e2evmc_config._asdict(), args.model_dir, config_name)
      print('>>> Saved model config to %s' % (config_path,))
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
  estimator_params = {'e2evmc_config': e2evmc_config, 'log_steps': args.log_steps, 'debug': args.debug}
  model_fn, _scope = _GOAL_CONDITION_TO_MODEL[args.goal_condition]
  estimator = est.Estimator(model_fn=model_fn, model_dir=args.model_dir, config=run_config, params=estimator_params)

  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  synthetic = args.dataset_dir.startswith('synthetic:')
  if world > 1 and not synthetic and args.batch_size % world:
    raise ValueError('--batch_size %d (the GLOBAL batch) must be divisible by the %d ranks' % (args.batch_size, world))

  # --num_threads as given; when the flag is absent, this rank's share of the host cores instead of the reference's fixed 4
  # (input_fn.default_reader_threads: epoch 1 of real-data training is reader-bound below ~13 cores per rank)
  reader_threads = args.num_threads

  def input_fn(estimator_mode):
    import torch
    dev = torch.device('cuda', local_rank)
    kw = {}
    if world > 1 and not synthetic:
      # every rank reads its own rank-strided subset of the episodes (one shuffle order agreed through rank 0's
      # seed) in batches of batch_size / world windows; ragged ends are handled by the Estimator (dp_schedule)
      seed = gdist.broadcast_int(int.from_bytes(os.urandom(4), 'little'), dev) if estimator_mode == 'train' else None
      kw = dict(shard=(rank, world), batch_size=args.batch_size // world, seed=seed)
    else:
      kw = dict(batch_size=args.batch_size, seed=None)
    return pickplace_input_fn(
        dataset_dir=args.dataset_dir, split_name=args.split_name, mode=estimator_mode, encoding=args.data_encoding,
        window_size=e2evmc_config.window_size, fetch_target=(args.goal_condition == 'target'),
        shuffle_buffer=args.shuffle_buffer, num_epochs=1, num_threads=reader_threads,
        prefetch_size=args.prefetch_size,
        # episodes are uploaded once (to THIS rank's GPU), stay there across epochs (input_fn.EPISODE_CACHE) and windows are
        # gathered in HBM unless GEECO_HOST_WINDOWS is set; an RGB model never reads the depth stream
        device=None if _dev.env('GEECO_HOST_WINDOWS') else dev,
        device_keys=('rgb',) if e2evmc_config.img_channels == 3 else ('rgb', 'depth'), **kw)
  train_input = lambda: input_fn(estimator_mode='train')
  eval_input = lambda: input_fn(estimator_mode='eval')

  if args.initial_eval:
    eval_results = estimator.evaluate(input_fn=eval_input)
    print('>>> initial eval: %s' % (eval_results,))
  for _epoch in range(args.train_epochs):
    estimator.train(input_fn=train_input)
    eval_results = estimator.evaluate(input_fn=eval_input)
    print('>>> eval: %s' % (eval_results,))
    if rank == 0:
      _export_snapshot(args.model_dir, eval_results, args.num_best_ckpt)


if __name__ == '__main__':
  print('>>> Training E2E VMC.')
  PARSED_ARGS, UNPARSED_ARGS = ARGPARSER.parse_known_args()
  print('>>> PARSED ARGV:')
  pprint.pprint(PARSED_ARGS)
  print('>>> UNPARSED ARGV:')
  pprint.pprint(UNPARSED_ARGS)
  main(PARSED_ARGS)
