#!/bin/bash
# Per-kernel profile of the fine-tuning leg (BASELINE configs[4], 4 x 1024^2) on the MI355X box, one precision per run:
#   bash tools/profile_finetune.sh r02_a          -> gpurun_out/r02_a/finetune_{fp32,fp16}_kernel_stats.csv
set -e
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in ${2:-fp32 fp16}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft_$P -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-refinement --train-steps 0 --finetune-steps 3 --finetune-prec $P > $O/finetune_${P}_line.json 2> $O/ft_$P.err
  cp $(ls $O/ft_$P/*/*kernel_stats.csv | head -1) $O/finetune_${P}_kernel_stats.csv
  rm -rf $O/ft_$P
  echo "finetune $P profile done"
done
