"""Dev: step time of the captured geeco-f train step when its input stage follows window addresses into resident uint8 frames,
against the same step fed from dense float32 windows, plus the host cost of one Estimator._feed.  GPU box only."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from geeco_amd import estimator as est, input_fn as I
from geeco_amd.params import create_e2evmc_config

K, B = 16, 32
dev = torch.device('cuda', 0)
work = tempfile.mkdtemp()
root = os.path.join(work, 'dataset')
meta, paths = bench.make_dataset(root, 8, 8)
for dense in (False, True):
  if dense:
    os.environ.update(GEECO_DEV='1', GEECO_NO_U8_WINDOWS='1')
  I.EPISODE_CACHE.clear()
  cfg = create_e2evmc_config(dict(proc_obs='dynimg', proc_tgt='dyndiff', window_size=K, img_channels=3, batch_size=B))
  e = est.Estimator(est.goal_e2evmc_model_fn, None, est.RunConfig(), {'e2evmc_config': cfg, 'log_steps': 100000, 'debug': False})
  kw = dict(window_size=K, fetch_target=True, batch_size=B, num_threads=8, prefetch_size=4, device=dev, device_keys=('rgb',))
  for ep in range(2):
    e.train(input_fn=lambda: I.pickplace_input_fn(root, 'default', 'train', seed=ep, **kw))
  (spec, fbuf, lbuf), = [v for k, v in e._specs.items() if k[1] == B]
  batches = list(I.pickplace_input_fn(root, 'default', 'train', seed=0, **kw))
  batches = [b for b in batches if len(b[0]['step']) == B]
  torch.cuda.synchronize()
  # replay only
  for _ in range(10):
    spec.train_op()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(200):
    spec.train_op()
  torch.cuda.synchronize()
  replay = (time.perf_counter() - t) / 200 * 1e3
  # feed + replay
  t = time.perf_counter()
  for i in range(200):
    f, l = batches[i % len(batches)]
    e._feed_step(fbuf, lbuf, f, l)
    spec.train_op()
  torch.cuda.synchronize()
  fed = (time.perf_counter() - t) / 200 * 1e3
  # host cost of feed alone (GPU idle)
  t = time.perf_counter()
  for i in range(200):
    f, l = batches[i % len(batches)]
    e._feed_step(fbuf, lbuf, f, l)
  host = (time.perf_counter() - t) / 200 * 1e3
  torch.cuda.synchronize()
  print('%s: replay only %.3f ms/step, feed + replay %.3f ms/step, feed host time %.3f ms' %
        ('dense windows' if dense else 'u8 addresses', replay, fed, host), flush=True)
