"""CPU oracle for the fk / rotation-ops hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``pymotion_amd`` may import this package.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
