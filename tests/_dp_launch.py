"""Test-only launcher: rehearses the N > 1 path on a ONE-GPU box.

  python -m torch.distributed.run --nproc-per-node 2 ... tests/_dp_launch.py <script.py> [script args]

RCCL refuses two ranks on one device, so this helper forms the process group over gloo with every rank on cuda:0
(explicit ``backend`` / ``device_index`` arguments of ``geeco_amd.dist.init_from_env`` -- the product reads no
rehearsal switches from the environment) and then runs ``<script.py>`` as ``__main__``: its own
``init_from_env()`` finds the group formed and keeps it.  Exercises the launcher protocol, input sharding, the
three-graph step and the bucketed exchange; throughput numbers of such a run mean nothing.
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

if __name__ == '__main__':
  os.environ['LOCAL_RANK'] = '0'          # every rank computes on the one GPU
  from geeco_amd import dist as gdist
  gdist.init_from_env('gloo', device_index=0)
  script = sys.argv[1]
  sys.argv = sys.argv[1:]
  runpy.run_path(script, run_name='__main__')
