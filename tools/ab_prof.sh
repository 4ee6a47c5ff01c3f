#!/bin/bash
# per-kernel A/B of one network between _base/ (the previous round's tree) and this tree, on one box:  bash tools/ab_prof.sh deq
set -euo pipefail
n=${1:-deq}
R="$GRAFT_REPO_ROOT"
out=$R/gpurun_out/abprof_$n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for t in base new; do
  s=$R/tools/net_one.py; [ $t = base ] && s=$R/_base/tools/net_one.py
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t -- python3 $s $n > $out/$t.log 2>&1
  cp "$(ls $out/$t/*/*kernel_stats.csv | head -1)" $out/$t.csv; rm -rf $out/$t
done
python3 - $out/base.csv $out/new.csv <<'PY'
import csv, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("==", f, "total %.3f ms" % (tot / 1e6))
    for r in rows[:22]:
        print("%-100s %5s calls %8.3f ms avg %9.3f ms %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
