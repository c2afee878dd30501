"""CPU oracle for the VEON lift hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``veon_amd/`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

Two restatements live here:

* ``oracle.c_oracle``  -- ctypes binding of ``oracle/lss_oracle.c`` (plain C,
  serial, canonical summation order; the bit-exact checker).
* ``oracle.lss_torch`` -- pure-PyTorch restatement (``index_add_`` pool), the
  "pure-PyTorch scatter_add CPU path" BASELINE.json asks to time beside the
  HIP numbers.

Parity pinning: pool fwd/bwd by the reference's known-answer test
(mmdet3d/ops/bev_pool_v2/bev_pool.py:145-176); geometry / prepare / two-hot
depth / max-pool by golden vectors generated from the reference's own Python
(``oracle/tools/gen_golden.py`` -> ``tests/golden/``).
"""
