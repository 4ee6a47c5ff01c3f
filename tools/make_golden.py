#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the float64 oracle.

The reference (TensorFlow) cannot be executed in the build container
(SURVEY.md section 8c), so these vectors are ORACLE-generated: they pin the oracle
against regressions and give the GPU tests fixed expected outputs.  Weights
are not stored: they are re-drawn from the stated seeds with
oracle.nets.init_params (NumPy default_rng is stable across platforms).

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets, ops  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEEDS = dict(deq=11, lin=12, hal=13, ref=14)


def main():
    os.makedirs(OUT, exist_ok=True)
    table = np.load(os.path.join(ROOT, "singlehdr-tf2_amd", "data", "invemor_g0_hinv11.npy"))
    rng = np.random.default_rng(0)
    ldr = np.round(rng.random((2, 64, 64, 3)) * 255.0) / 255.0
    ldr[0, :8, :8, :] = 1.0      # saturated patch -> alpha mask fires
    params = {k: nets.init_params(getattr(nets, k + "_spec")(), s) for k, s in SEEDS.items()}
    out = nets.inference(params, ldr, table)
    save = {k: v.astype(np.float32) for k, v in out.items()}
    save["ldr"] = ldr.astype(np.float32)
    for k, s in SEEDS.items():
        save["seed_" + k] = np.int64(s)
    np.savez_compressed(os.path.join(OUT, "inference_64.npz"), **save)

    # soft-histogram / front-end vectors in float32 (bit-exact targets)
    x = (np.round(np.random.default_rng(1).random((1, 16, 16, 3)) * 255.0) / 255.0).astype(np.float32)
    fe = {"x": x, "frontend93": ops.lin_frontend(x)}
    for b in (4, 5, 8, 16, 32):
        fe["hist%d" % b] = ops.histogram_layer(x, b)
    np.savez_compressed(os.path.join(OUT, "frontend_16.npz"), **fe)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
