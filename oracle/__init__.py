"""CPU oracle for the trimodal gesture GAN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this
directory; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Parity pin: the oracle is checked against golden vectors produced by the real
reference (``tests/golden/make_golden.py`` imports ``/root/reference`` in the
build container and commits inputs/outputs as ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.  The reference itself ships no tests or
golden vectors, so those generated vectors are the pin.
"""
