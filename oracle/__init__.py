"""CPU oracle for the BP hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  PARITY UNPINNED (see bp_oracle.c header and DESIGN.md).
"""
from .oracle import BPOracle, build as build_oracle, csc_from_dense, osd_oracle_postprocess  # noqa: F401
from .bpots import BPOTSOracle  # noqa: F401,E402
