"""geeco_amd -- MI355X-native implementation of GEECO's e2evmc training hot path.

The compute path is the hand-written HIP library ``libgeeco_hip.so`` (geeco_amd/csrc, C ABI in
include/geeco_hip.h); the Python modules mirror the reference's call surface
(``params``, ``graph``, ``estimator``, ``input_fn``) on top of it.
"""
__version__ = '0.1.0'
