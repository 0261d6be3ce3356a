"""
ORACLE -- CPU restatement of the reference's algorithm for the hot path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, call, link or
execute anything in this directory, and only as the checker / the CPU baseline.  The product
(ground-plane-polling_amd/) never imports it and fails loudly when the HIP library is missing.
"""
