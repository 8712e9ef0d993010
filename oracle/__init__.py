"""oracle/ - TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the hot path, used as the
checker for the HIP kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; nothing under ``simple_pose_amd/`` does, and the
product path raises if the HIP library is missing instead of falling back to anything here.

Pinning status (see DESIGN.md "Oracle pinning"): the reference ships no tests or golden vectors for
this path, so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``oracle/ref_import.py`` and frozen by ``oracle/gen_golden.py``
into ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the oracle against those
fixtures everywhere (no reference needed).
"""
