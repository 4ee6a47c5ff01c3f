#!/bin/bash
# Regenerates the judged profiles of a round on the MI355X box:  bash tools/profile_round.sh r01_m
# (kernel-trace stats of the inference bench and of the training leg, two separate PMC passes for HBM traffic).
set -e
TAG=${1:-r03_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
INF="--streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-refinement --train-steps 0 --finetune-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/inf -- python3 $R/bench.py $INF > $O/bench_line.json 2> $O/inf.err
echo "inference profile done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-refinement --train-steps 4 --finetune-steps 0 > $O/train_line.json 2> $O/train.err
echo "training profile done"
PM="--streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-refinement --train-steps 0 --finetune-steps 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PM > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PM > /dev/null 2> $O/pmc_w.err
echo "pmc passes done"
cp $(ls $O/inf/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
cp $(ls $O/train/*/*kernel_stats.csv | head -1) $O/train_kernel_stats.csv
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic.json
rm -rf $O/inf $O/train
