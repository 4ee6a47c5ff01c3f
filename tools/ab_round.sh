#!/bin/bash
# GPU tests, then the four networks' forward on _base/ and on this tree, then the bench line.
# _base/ = a checkout of the previous round's last commit with its library built and tools/net_one.py, tools/img_layers.py copied in
# (git worktree add _base <commit> && python _base/singlehdr-tf2_amd/build.py); git-ignored, travels to the GPU box with the snapshot
set -e
python -m pytest tests -m gpu -q -x > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -2 gpurun_out/ab_tests.log
(python _base/tools/img_layers.py; python tools/img_layers.py; for n in deq lin hal ref; do python _base/tools/net_one.py $n; python tools/net_one.py $n; done) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_nets.log
python bench.py --steps 20 --warmup 5 > gpurun_out/ab_bench.log 2> gpurun_out/ab_bench.err; cut -c1-300 gpurun_out/ab_bench.log
