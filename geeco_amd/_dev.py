"""Development switches.

The kernels and the graph have A/B switches (kernel variants, stream schedules, forced tile shapes: DESIGN.md lists
them with what each measured).  They are read from the environment ONLY when ``GEECO_DEV=1`` is set; a production
process ignores every other ``GEECO_*`` variable and always runs the measured-best path.  The C library applies the same
gate (``geeco_dev_getenv`` in csrc/errors.cpp).
"""
import os


def enabled() -> bool:
  v = os.environ.get('GEECO_DEV')
  return bool(v) and v != '0'


def env(name, default=None):
  """os.environ.get(name, default) under GEECO_DEV=1, else ``default``."""
  return os.environ.get(name, default) if enabled() else default
