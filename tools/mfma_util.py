#!/usr/bin/env python3
"""Post-process `tools/clock_pmc.sh` (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES + kernel trace):
shader clock under load and the fraction of clocked cycles the matrix pipes are busy, per conv kernel.

    python tools/mfma_util.py gpurun_out/pmc_clk_wf gpurun_out/pmc_clk_dc profiles/r01_mfma_util.json
"""
import collections
import csv
import glob
import json
import re
import sys

N_XCD, N_CU, SIMD_PER_CU = 8, 256, 4


def one(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, c in cnt.items():
        m = re.search(r"(conv_\w+|winograd_fused_kernel)(<[^>]*>)?", k)
        if not m or "filter" in k or k not in dur:
            continue
        avg = {n: sum(v) / len(v) for n, v in c.items()}
        ns = sum(dur[k]) / len(dur[k])
        busy_cu = avg["SQ_BUSY_CU_CYCLES"] / N_CU                       # cycles one CU was busy
        out[m.group(1) + (m.group(2) or "")] = {
            "launches": len(dur[k]), "avg_duration_ms": round(ns / 1e6, 4),
            "shader_clock_ghz": round(avg["GRBM_GUI_ACTIVE"] / N_XCD / ns, 3),
            "mfma_busy_fraction_of_clocked_cycles": round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["SQ_BUSY_CU_CYCLES"] * SIMD_PER_CU), 4),
            "mfma_busy_cycles_per_simd": round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_CU * SIMD_PER_CU)),
            "busy_cycles_per_cu": round(busy_cu)}
    return out


def main():
    res = {"recipe": "tools/clock_pmc.sh; GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_* over all CUs / SIMDs; nominal clock 2.4 GHz "
                     "(the 157.3 TFLOP/s fp32-MFMA peak)", "kernels": {}}
    for d in sys.argv[1:-1]:
        res["kernels"].update(one(d))
    json.dump(res, open(sys.argv[-1], "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, v)


if __name__ == "__main__":
    main()
