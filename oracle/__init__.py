"""CPU oracle for the bi-temporal change-detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``stcd_amd``) never imports this package and fails loudly
when its HIP library is missing.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the reference's own leaf modules (``tests/golden/*.npz``, produced
by ``tests/golden/make_golden.py`` in the authoring container, where
``/root/reference`` is importable).  The reference has no tests of its own
(SURVEY.md section 4), so those captured vectors are the only pin.

Exception: ``pseudo_ref.py`` (pseudo-change pair synthesis) is PARITY UNPINNED --
the reference assembles those pairs from files and holds no generator arithmetic
(see the module header for what is and is not taken from the reference).
"""
