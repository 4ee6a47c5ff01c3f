#!/usr/bin/env python3
"""Convert the EMoR inverse-CRF PCA table to a build-owned binary fixture.

Input : the text data table `invemor.txt` shipped with the reference
        (/root/reference/invemor.txt; tags `g0 =` and `hinv(1)=`..`hinv(11)=`,
        each followed by 256 rows x 4 values -- the layout the reference reads at
        linearization_net.py:217-227 / :255-268).
Output: singlehdr-tf2_amd/data/invemor_g0_hinv11.npy  float32 [1024, 12]
        column 0 = g0, columns 1..11 = hinv(1)..hinv(11).

This is a *data* conversion (numbers only); no reference code is copied.
Run once in the build container:  python tools/convert_invemor.py
"""
import os
import sys
import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/invemor.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "singlehdr-tf2_amd", "data", "invemor_g0_hinv11.npy")


def read_block(lines, tag):
    idx = next(i for i, ln in enumerate(lines) if ln == tag)
    vals = []
    for ln in lines[idx + 1: idx + 1 + 256]:
        vals.extend(ln.split())
    assert len(vals) == 1024, (tag, len(vals))
    return np.asarray(vals, dtype=np.float32)


def main():
    with open(SRC) as f:
        lines = [ln.strip() for ln in f]
    cols = [read_block(lines, "g0 =")]
    cols += [read_block(lines, "hinv(%d)=" % (i + 1)) for i in range(11)]
    tab = np.stack(cols, axis=-1).astype(np.float32)
    assert tab.shape == (1024, 12)
    np.save(DST, tab)
    print("wrote", os.path.normpath(DST), tab.shape,
          "sum g0 = %.6f" % tab[:, 0].astype(np.float64).sum(),
          "sum hinv = %.6f" % tab[:, 1:].astype(np.float64).sum())


if __name__ == "__main__":
    main()
