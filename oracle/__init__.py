"""CPU oracle for the text-to-audio-grounding hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``texttoaudiogrounding_amd`` (the product)
may import this package.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg, and there only as the checker.
"""
