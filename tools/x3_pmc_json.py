#!/usr/bin/env python3
"""gpurun_out/x3_pmc/summary_*.txt (tools/x3_pmc.sh) -> one JSON with the derived figures the round report quotes:
    python tools/x3_pmc_json.py profiles/r03_x3_pmc.json gpurun_out/x3_pmc/summary_64_512_512.txt ..."""
import json
import sys


def parse(path):
    d = {}
    for ln in open(path):
        p = ln.split()
        if len(p) >= 2 and p[0].isupper() and p[1].replace(".", "").isdigit():
            d[p[0]] = float(p[1])
        elif p and p[0] == "duration":
            d["duration_us"] = float(p[1])
    return d


def derive(d):
    out = {"duration_us_under_profiler": d.get("duration_us")}
    cu = d["SQ_BUSY_CU_CYCLES"]
    out["mfma_busy_frac_of_simd_cycles"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cu * 4), 4)
    out["mfma_instructions"] = d["SQ_INSTS_MFMA"]
    out["valu_non_mfma_per_mfma"] = round((d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_INSTS_MFMA"], 3)
    out["lds_instructions_per_mfma"] = round(d["SQ_INSTS_LDS"] / d["SQ_INSTS_MFMA"], 3)
    out["lds_bank_conflict_cycles"] = d["SQ_LDS_BANK_CONFLICT"]
    out["lds_active_frac_of_cu_cycles"] = round(d["SQ_LDS_IDX_ACTIVE"] / cu, 4)
    out["valu_mfma_coexec_frac_of_mfma_busy"] = round(d["SQ_VALU_MFMA_COEXEC_CYCLES"] / d["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
    w = d["SQ_WAVE_CYCLES"]
    out["wave_time_split"] = {"parked_in_waitcnt_or_barrier": round(d["SQ_WAIT_ANY"] / w, 4),
                              "waiting_to_issue": round(d["SQ_WAIT_INST_ANY"] / w, 4),
                              "issuing": round(d["SQ_ACTIVE_INST_ANY"] / w, 4)}
    if d.get("duration_us"):
        out["shader_clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / 8 / d["duration_us"] / 1e3, 3)
    return out


def main():
    res = {"source": "rocprofv3 --pmc (separate passes, kernel-trace only), tools/x3_pmc.sh; per launch of conv_x3_kernel<false,3,3>, batch 16",
           "layers": {}}
    for p in sys.argv[2:]:
        d = parse(p)
        tag = p.split("summary_")[-1].replace(".txt", "")
        res["layers"][tag] = {"derived": derive(d), "counters": {k: v for k, v in d.items() if k.isupper()}}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in res["layers"].items()}, indent=1))


if __name__ == "__main__":
    main()
