#!/bin/bash
# kernel-trace stats of the default inference step only (one stream, no roofline / exact / refinement legs):  bash tools/prof_inf.sh TAG
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG="${1:-inf}"
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/inf" -- python3 "$R/bench.py" --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-refinement --train-steps 0 --finetune-steps 0 > "$O/line.json" 2> "$O/inf.err"
cp "$(ls "$O"/inf/*/*kernel_stats.csv | head -1)" "$O/kernel_stats.csv"
rm -rf "$O/inf"
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-92s %5s calls %8.3f ms avg %9.3f ms %5.1f%%" % (r["Name"][:92], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
