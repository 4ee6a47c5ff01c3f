#!/bin/bash
# kernel-trace stats of the joint training step only (one warm-up step + four timed ones):  bash tools/prof_train.sh TAG
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG="${1:-train}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-refinement --train-steps 4 --finetune-steps 0 > "$O/train_line.json" 2> "$O/train.err"
cp "$(ls "$O"/train/*/*kernel_stats.csv | head -1)" "$O/train_kernel_stats.csv"
rm -rf "$O/train"
python3 - "$O/train_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.1f ms" % (tot / 1e6))
for r in rows[:45]:
    print("%-92s %5s calls %8.3f ms avg %9.3f ms %5.1f%%" % (r["Name"][:92], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
