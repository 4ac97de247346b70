"""Dev: what a command queued between two replays of the captured train step costs (geeco-f rgb K=16 N=32, synthetic inputs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
cfg, model = bench.build_model('geeco-f', 3, 16, 32, dev)
model.store.initialize(seed=0)
bench.synthetic_batch(model, 1234)
from geeco_amd.runtime import TrainStepRunner
runner = TrainStepRunner(model, use_graph=True)
for _ in range(10):
  runner.step()
torch.cuda.synchronize()
small_h = torch.zeros(4096, dtype=torch.float32, pin_memory=True)
small_d = torch.zeros(4096, dtype=torch.float32, device=dev)
small_d2 = torch.zeros(4096, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(dev)
ev = torch.cuda.Event()

def none(): pass
def record(): ev.record()
def h2d(): small_d.copy_(small_h, non_blocking=True)
def d2d(): small_d2.copy_(small_d, non_blocking=True)
def kernel(): small_d2.add_(1.0)
def side_h2d():
  with torch.cuda.stream(side):
    small_d.copy_(small_h, non_blocking=True)
def side_h2d_wait():
  with torch.cuda.stream(side):
    small_d.copy_(small_h, non_blocking=True)
    ev.record(side)
  torch.cuda.current_stream().wait_event(ev)

for name, fn in (('nothing', none), ('event record', record), ('16 KB H2D', h2d), ('16 KB D2D', d2d), ('tiny kernel', kernel),
                 ('H2D on a side stream, no wait', side_h2d), ('H2D on a side stream + wait_event', side_h2d_wait), ('nothing', none)):
  for _ in range(5):
    fn(); runner.step()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(200):
    fn()
    runner.step()
  torch.cuda.synchronize()
  print('%-40s %.4f ms/step' % (name, (time.perf_counter() - t) / 200 * 1e3), flush=True)
