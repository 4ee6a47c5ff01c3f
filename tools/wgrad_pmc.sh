# PMC counters of the fp16 weight-gradient kernels on one layer (separate passes, kernel-trace only):  bash tools/wgrad_pmc.sh HW CIN COUT [K]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="${1:-1024} ${2:-64} ${3:-64} ${4:-3}"
O=$R/gpurun_out/wgrad_pmc
mkdir -p $O
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/wgrad_h_one.py $A > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/wgrad_pmc/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "wgrad_f16" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-32s %16.0f per launch" % (k, tot[k] / max(n[k], 1)))
PY
