"""oracle/ -- CPU checkers for the hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
pointnerf_amd/ never does (tests/test_boundary.py greps for it).
"""
