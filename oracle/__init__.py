"""CPU oracle for the RRDB + CEM hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product package
(explorable-super-resolution_amd/) never imports it and fails loudly when its HIP library is
missing.  See oracle/README.md for what is pinned against the reference and how.
"""
