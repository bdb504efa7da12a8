"""CPU oracle for the dpf-nets hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package; dpf_nets_amd never does.
"""
