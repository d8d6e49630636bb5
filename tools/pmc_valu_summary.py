#!/usr/bin/env python3
"""Summarises the vector-unit counter passes of tools/profile_round.sh (pmc_valu_*.csv, pmc_f64_*.csv, pmc_lds_*.csv: mean per dispatch and
kernel) into pmc_valu_summary.json — what bench.py's roofline_lm prices the recompute passes with.
  valu_busy_frac   = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)      (quad-cycles of waves executing VALU instructions)
  f64_flops        = 64 lanes x (ADD + MUL + TRANS + 2 x FMA) wave instructions              (SQ_INSTS_VALU_*_F64; full exec mask assumed: an upper bound)
  mfma_f64_flops   = 512 x SQ_INSTS_VALU_MFMA_MOPS_F64                                      (the counter's unit is 512 operations)
usage: python tools/pmc_valu_summary.py gpurun_out/r06"""
import csv
import json
import os
import sys

out = sys.argv[1]
summary = {"note": "per kernel, mean per dispatch; valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); f64_flops = 64 x (ADD + MUL + TRANS + 2 FMA) "
                   "wave instructions of SQ_INSTS_VALU_*_F64 (exec mask not known to the counter: an upper bound); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"}
for cfg in ("C4", "C5"):
    def rows(kind):
        path = f"{out}/pmc_{kind}_{cfg}.csv"
        return {r["kernel"]: r for r in csv.DictReader(open(path))} if os.path.exists(path) else {}
    valu, f64, lds = rows("valu"), rows("f64"), rows("lds")
    d = {}
    for k, r in valu.items():
        gui = float(r.get("GRBM_GUI_ACTIVE", 0) or 0) / 8.0
        if gui <= 0 or not k.startswith("rsba::"):
            continue
        e = {"dispatches": int(r["dispatches"]), "cycles_per_xcd": gui, "valu_insts": float(r["SQ_INSTS_VALU"]), "salu_insts": float(r["SQ_INSTS_SALU"]),
             "valu_busy_frac": 4.0 * float(r["SQ_ACTIVE_INST_VALU"]) / (gui * 1024.0), "waves": float(r.get("SQ_WAVES", 0) or 0),
             "waves_per_simd": 4.0 * float(r["SQ_WAVE_CYCLES"]) / (gui * 1024.0)}
        if k in f64:
            q = f64[k]
            g = lambda n: float(q.get(n, 0) or 0)
            e["f64_wave_insts"] = {"add": g("SQ_INSTS_VALU_ADD_F64"), "mul": g("SQ_INSTS_VALU_MUL_F64"), "fma": g("SQ_INSTS_VALU_FMA_F64"), "trans": g("SQ_INSTS_VALU_TRANS_F64")}
            e["f64_flops"] = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64"))
            e["mfma_f64_flops"] = 512.0 * g("SQ_INSTS_VALU_MFMA_MOPS_F64")
        if k in lds and float(lds[k].get("SQ_LDS_IDX_ACTIVE", 0) or 0) > 0:
            e["lds_conflict_frac"] = float(lds[k]["SQ_LDS_BANK_CONFLICT"]) / float(lds[k]["SQ_LDS_IDX_ACTIVE"])
            e["lds_insts"] = float(lds[k].get("SQ_INSTS_LDS", 0) or 0)
        d[k] = e
    summary[cfg] = d
json.dump(summary, open(f"{out}/pmc_valu_summary.json", "w"), indent=1)
for cfg in ("C4", "C5"):
    for k, e in summary[cfg].items():
        if e["valu_busy_frac"] > 0.05:
            print(cfg, k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items() if a != "f64_wave_insts"})
