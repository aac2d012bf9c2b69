"""Parity oracle (TEST INFRASTRUCTURE ONLY -- never imported by onebit_amd).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  See ``oracle/oracle.py``.
"""
