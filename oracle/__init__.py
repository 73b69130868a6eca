"""oracle -- CPU checker for the gfx950 q8 hot path. TEST INFRASTRUCTURE ONLY.

Nothing under ``qnnpack_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

* ``oracle.o1``  -- the scalar restatement (oracle_q8.c) through ctypes.
* ``oracle.ref`` -- the compiled reference library O2 (oracle/_ref, built from
  /root/reference by oracle/Makefile), through the same ctypes binding class the
  product uses.
"""
