#!/bin/bash
# PMC counters of the split-operand kernel on one layer (separate passes, kernel-trace only):  bash tools/x3_pmc.sh [HW CIN COUT [TAG]]
# The per-launch summary goes to gpurun_out/x3_pmc/summary_<TAG>.txt (copy the ones worth keeping into profiles/).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
HW="${1:-64}"; CIN="${2:-512}"; COUT="${3:-512}"
TAG="${4:-${HW}_${CIN}_${COUT}}"
KS="${5:-3}"; C2="${6:-0}"                                   # kernel size (3 or 1) and channels of a second source
O="$R/gpurun_out/x3_pmc/$TAG"
mkdir -p "$O"
i=0
DEFAULT_SETS="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS;SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT;SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE;SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
IFS=';' read -ra SETS <<< "${X3_PMC_SETS:-$DEFAULT_SETS}"       # X3_PMC_SETS="A B;C D": other counter passes
for SET in "${SETS[@]}"; do
  i=$((i+1))
  # shellcheck disable=SC2086
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/p$i" -- python3 "$R/tools/x3_one.py" "$HW" "$CIN" "$COUT" 16 "$KS" "$C2" > "$O/p$i.log" 2>&1 || echo "pass $i failed"
done
cd "$R" && python3 - "$O" "$TAG" <<'PY'
import csv, glob, collections, sys
o, tag = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for f in glob.glob(o + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_x3_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob(o + "/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_x3_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["layer %s (tools/x3_one.py), per launch" % tag]
if dur:
    lines.append("%-32s %16.1f us (pass 1, under the profiler)" % ("duration", sum(dur) / len(dur)))
for k in sorted(tot):
    lines.append("%-32s %16.0f" % (k, tot[k] / max(n[k], 1)))
open(o + "/../summary_%s.txt" % tag, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
