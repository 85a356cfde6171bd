"""CPU oracle for the consensus-network forward pass.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this package (see oracle/gru_oracle.c header).  The product path (`medaka_amd/`) never does.
"""
