"""Host side of real-data training: episodes/s of the native reader at 1 / 2 / 4 / 8 / 16 / 32 threads on this box's
host cores, and from it the host cores per rank at which epoch 1 stops being reader-bound.  CPU only (no GPU call).
  python scripts/dev/reader_scaling.py [episodes=32]  ->  JSON on stdout"""
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402  (make_dataset, host_cores)
from geeco_amd import input_fn as I            # noqa: E402

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
threads, box, usable = bench.host_cores()
work = tempfile.mkdtemp(prefix='geeco_reader_')
meta, paths = bench.make_dataset(os.path.join(work, 'dataset'), episodes, max(threads, 8))
read = lambda p: I.load_episode(p, meta, True, raw_rgb=True, image_keys=('rgb',))
read(paths[0])
out = {'box_cores': box, 'usable_cores': usable, 'episodes': episodes, 'frames_per_episode_K16': 84 * 16, 'episodes_per_s': {}}
for n in (1, 2, 4, 8, 16, 32):
  sel = paths[:max(4, min(episodes, 2 * n))] if n < 8 else paths
  best = 0.0
  for _ in range(2):
    t = time.perf_counter()
    if n == 1:
      for p in sel:
        read(p)
    else:
      with ThreadPoolExecutor(max_workers=n) as ex:
        list(ex.map(read, sel))
    best = max(best, len(sel) / (time.perf_counter() - t))
  out['episodes_per_s'][str(n)] = round(best, 2)
  print('%2d threads: %.1f episodes/s' % (n, best), file=sys.stderr, flush=True)
step_ms = 3.2                                   # the GPU step (bench.py)
need = 1e3 / step_ms * 32 * 16 / (84 * 16)      # episodes/s one rank consumes: 512 frames per step, 1344 frames per episode
per_thread = out['episodes_per_s']['1']
out['episodes_per_s_one_rank_consumes'] = round(need, 1)
out['threads_per_rank_to_keep_up'] = round(need / per_thread, 1)
import shutil
shutil.rmtree(work, ignore_errors=True)
print(json.dumps(out))
