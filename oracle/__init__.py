"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/ac_oracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
