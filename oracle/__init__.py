"""CPU oracle for the 3D-segmentation hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic the reference performs for the
path named in BASELINE.json (U-Net-family forward/backward, BCE/Dice/CE losses and
the Dice/Jaccard metric).  The reference's arithmetic for that path is PyTorch/ATen
on CPU (it owns no native code), so the restatement is a set of plain
``torch.nn`` modules / functions with the same constructor signatures and
``state_dict`` keys as the reference, plus a NumPy restatement of the integer
metric.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline.
Nothing under the product package imports it, and the product path never falls
back to it: if the HIP library is missing the product raises.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference modules from /root/reference in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this
restatement against them, and ``tests/test_oracle_vs_reference.py`` checks it
against the live reference modules whenever /root/reference is present.
"""
