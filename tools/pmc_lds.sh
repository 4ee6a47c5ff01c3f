# LDS bank conflicts and vector instruction mix per kernel of the fp16 fine-tuning step:  bash tools/pmc_lds.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lds
mkdir -p $O
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-refinement --train-steps 0 --finetune-steps 1 --finetune-prec fp16"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python3 $R/bench.py $ARGS > $O/p1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p2 -- python3 $R/bench.py $ARGS > $O/p2.log 2>&1 || echo "pass 2 failed"
cd $R && python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_lds/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0))[:16]
for k, c in rows:
    busy = c.get("SQ_BUSY_CU_CYCLES", 1)
    print("%-60s busy %7.1fM  mfma %.2f  lds_conflict/active %.2f  lds_active/busy %.2f  valu/mfma-instr %s" % (
        k, busy / 1e6, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * busy), c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        c.get("SQ_LDS_IDX_ACTIVE", 0) / busy, "%.1fM" % (c.get("SQ_INSTS_VALU", 0) / 1e6)))
PY
