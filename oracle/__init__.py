"""CPU oracle for the tinyimgcodec hot path - TEST INFRASTRUCTURE ONLY (see oracle/tic_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
