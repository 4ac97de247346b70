"""The data-parallel step over a ONE-rank RCCL group, graph-replayed, for a kernel trace:
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o r1 -- python3 $R/scripts/dev/dp_trace.py ; then dp_trace_print.py"""
import os, socket, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from geeco_amd import dist as gdist
from geeco_amd.runtime import TrainStepRunner

with socket.socket() as s_:
  s_.bind(('127.0.0.1', 0))
  port = s_.getsockname()[1]
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
assert gdist.init_from_env('nccl', device_index=0, single_rank_group=True) == 1
cfg, model = bench.build_model('geeco-f', 3, 16, 32, dev)
model.store.initialize(seed=0)
bench.synthetic_batch(model, 1234)
r = TrainStepRunner(model, use_graph=True, dp=True, overlap=os.environ.get('DP_SERIAL') is None)
r.prepare()
for _ in range(12):
  r.step()
torch.cuda.synchronize()
print('graphs per step', r.bucket_info()['graphs_per_step'], 'split adam', getattr(r, 'split_adam', None))
torch.distributed.destroy_process_group()
