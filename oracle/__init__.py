"""TEST INFRASTRUCTURE - CPU restatements used only as the checker.

Nothing in the product package (``visual_foresight_amd/``) imports this directory.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.
"""
