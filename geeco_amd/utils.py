"""Config persistence and run-command dump (reference: src/models/e2evmc/utils.py:16-27,
src/utils/runscript.py:13-30).  File names and JSON layout are kept so that either
implementation can resume the other's ``model_dir``."""
import datetime
import json
import os
import time


def save_model_config(config: dict, run_dir, name):
  with open(os.path.join(run_dir, '%s.json' % (name,)), 'w') as f:
    json.dump(config, f, indent=2, sort_keys=True)


def load_model_config(run_dir, name):
  with open(os.path.join(run_dir, '%s.json' % (name,)), 'r') as f:
    return json.load(f)


def save_run_command(argparser, run_dir):
  """Dumps parsed/unparsed argv to <run_dir>/<timestamp>-runcmd.json and returns the path."""
  stamp = datetime.datetime.fromtimestamp(time.time()).strftime('%Y%m%d_%H%M%S%f')[:-3]
  parsed, unparsed = argparser.parse_known_args()
  path = os.path.join(run_dir, '%s-runcmd.json' % (stamp,))
  with open(path, 'w') as f:
    json.dump({'parsed_args': vars(parsed), 'unparsed_args': unparsed}, f, indent=2, sort_keys=True)
  return path
