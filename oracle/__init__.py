"""Test infrastructure only: CPU restatement of the reference's e2evmc train step.

Nothing under ``geeco_amd/`` may import this package (see oracle/geeco_oracle.py header).
"""
