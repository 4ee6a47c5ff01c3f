# PMC counters of the narrow split-operand kernel on the 3x3 4 -> 64 image layer (separate passes, kernel-trace only):  bash tools/x3n_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x3n_pmc
mkdir -p $O
i=0
for SET in "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/dbg/i64_store.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/x3n_pmc/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_x3n_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-32s %16.0f per launch" % (k, tot[k] / max(n[k], 1)))
PY
