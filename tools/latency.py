#!/usr/bin/env python3
"""Single-image latency of the full deq+lin+hal+ref inference (the reference tool runs one image at a time)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("singlehdr-tf2_amd")
torch.manual_seed(0)
nets = [pkg.dequantization_net.model(), pkg.linearization_net.model(), pkg.hallucination_net.model(), pkg.refinement_net.model()]
eager = pkg.pipeline.Inference(*nets)
graphed = pkg.pipeline.GraphedInference(*nets)
for h, w in ((512, 512), (1024, 1024), (1088, 1600)):
    x = torch.rand(1, h, w, 3, device="cuda")
    for name, run in (("eager", eager), ("hip-graph", graphed)):
        for _ in range(3):
            run(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run(x)
        torch.cuda.synchronize()
        print("%-9s 1 x %4d x %4d: %.2f ms" % (name, h, w, (time.perf_counter() - t0) / 10 * 1e3))
