"""CPU oracle for the EK-FAC hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``kronfluence_amd``) never imports it and has no CPU fallback.
"""
