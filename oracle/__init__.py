"""CPU oracle for the facial-landmark hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product package
(``face-landmark-detector_amd``) never imports this directory and fails loudly
when its HIP extension is missing.

Parity pin status (details in DESIGN.md):

* ``decode_ref``  -- PINNED.  Checked in-container against the reference's own
  ``keypoints_detector/utils/metrics.py`` (the only hot-path module of the
  reference that imports here) and frozen in ``tests/golden/decode_golden.npz``.
* ``fcn_ref``     -- PARITY UNPINNED.  The reference delegates the arithmetic to
  TensorFlow/Keras, which is not installed and cannot be; the reference holds no
  tests, fixtures or golden outputs.  The restatement follows the cited source
  lines plus documented Keras layer defaults and is cross-checked against an
  independent naive-loop restatement (``tests/test_oracle_fcn.py``).
* ``warp_ref``    -- PARITY UNPINNED against the reference (it has no alignment
  code at all); cross-checked against scikit-image 0.18.3, the library the
  reference's only affine warp calls (``data/generator.py:192-200``).
"""
