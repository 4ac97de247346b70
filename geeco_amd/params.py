"""Hyper-parameters of the E2E-VMC controllers.

Mirror of the reference's ``src/models/e2evmc/params.py:7-47``: same field names, defaults and
override rule (``create_e2evmc_config`` silently ignores unknown keys), so an
``e2evmc_config.json`` written by either implementation loads in the other.
"""
import collections

_FIELDS_AND_DEFAULTS = (
    ('img_height', 256),
    ('img_width', 256),
    ('img_channels', 3),            # 3 = rgb, 4 = rgbd (train_e2evmc.py:129-132)
    ('dim_jnt_state', 7),
    ('dim_grp_command', 2),
    ('control_mode', 'cartesian'),  # cartesian | velocity
    ('num_grp_states', 3),
    ('dim_action', 4),
    ('proc_obs', 'sequence'),       # sequence | dynimg
    ('proc_tgt', 'constant'),       # constant | residual | dyndiff
    ('dim_s_obs', 256),
    ('dim_s_dyn', 256),
    ('dim_s_diff', 256),
    ('dim_h_lstm', 128),
    ('dim_h_fc', 128),
    ('window_size', 4),
    ('l2_regularizer', 0.0),
    ('lambda_aux', 1.0),
    ('batch_size', 32),
    ('lr', 1e-4),
)

E2E_VMC_DEFAULT_PARAM_DICT = dict(_FIELDS_AND_DEFAULTS)

E2EVMCConfig = collections.namedtuple('E2EVMCConfig', [k for k, _ in _FIELDS_AND_DEFAULTS])

E2E_VMC_DEFAULT_CONFIG = E2EVMCConfig(**E2E_VMC_DEFAULT_PARAM_DICT)


def create_e2evmc_config(custom_params: dict) -> E2EVMCConfig:
  """Defaults overridden by the known keys of ``custom_params`` (params.py:37-47)."""
  merged = dict(E2E_VMC_DEFAULT_PARAM_DICT)
  merged.update({k: v for k, v in custom_params.items() if k in merged})
  return E2EVMCConfig(**merged)
