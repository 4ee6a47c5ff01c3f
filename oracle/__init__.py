"""CPU oracle for the SingleHDR hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference's algorithm for the hot
path named in BASELINE.json (the four sub-networks, the soft-histogram layer,
the VGG16 perceptual-loss forward and the caller arithmetic around them).
Every function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it; the product path (`singlehdr-tf2_amd/`) never does.

PARITY PIN STATUS -- "parity unpinned by the reference's own execution":
the reference is pure TensorFlow-2 Python; TensorFlow is not installed in the
build container (plain ModuleNotFoundError) and the
reference ships no tests, golden vectors or fixtures for this path.  The
oracle is therefore pinned by
  (i)   the only known-answer example the reference holds: the B=5 worked
        soft-histogram example of figure/lin2.png (tests/test_oracle.py),
  (ii)  the invemor.txt check-sums recorded in SURVEY.md section 8c,
  (iii) hand-computable index tests for the TF SAME-padding / pooling /
        REFLECT-sobel / SYMMETRIC-pad conventions,
  (iv)  cross-checks against torch-CPU ops where torch semantics provably
        coincide with TF's (explicit-pad conv2d, interpolate(align_corners=
        False), max_pool2d after explicit pad).
TensorFlow itself was never executed.
"""
from . import ops, nets  # noqa: F401
