"""counter_collection.csv files of tools/pmc_r06.sh -> profiles/r06/pmc_roofline.json (the file bench.py's roofline blocks cite).
    python tools/pmc_r06.py gpurun_out/pmc_r06 [G]"""
import csv, sys, glob, collections, statistics, json, os, re

root = sys.argv[1]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N_SIMD, N_XCD, N_SE = 1024, 8, 32
cells = G * 1024 * 1024


def med(tag, pat):
    agg = collections.defaultdict(list); names = set()
    for path in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    return {c: statistics.median(v) for c, v in agg.items()}, sorted(names)


def us(tag):
    try:
        m = re.findall(r"([0-9.]+) us/launch", open(os.path.join(root, tag + ".log")).read())
        return float(m[-1]) if m else None
    except OSError:
        return None


def valu_busy(bz):
    cyc = bz["GRBM_GUI_ACTIVE"] / N_XCD if "GRBM_GUI_ACTIVE" in bz else bz.get("SQ_BUSY_CYCLES", 0.0) / N_SE
    return bz.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / max(N_SIMD * cyc, 1.0)


out = {"workload_cells": cells, "workload": "%d x 1024^2 glaciers of bench.py (make_glacier), one MI355X" % G,
       "source": "tools/pmc_r06.sh (rocprofv3 --pmc, one counter group per pass, kernel-trace only), tools/pmc_r06.py",
       "fetch_correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); sizes in KiB",
       "valu_busy_definition": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
# 54 x 54 output tiles of the 8-row strip kernels, 54 x 46 of the 7-row ones: ceil(1024 / 54) = 19, ceil(1024 / 46) = 23 per glacier
t8, t7 = G * 19 * 19, G * 19 * 23
for tag, key, pat, tiles, bpc in (("fused_step_nnA", "fused_step_nn_gridded", "k_rk_fused_strip", t8, 32.0), ("fused_step_const", "fused_step_constA", "k_rk_fused_strip", t8, 24.0),
                                  ("fused_step_nnY_tab", "fused_step_Y_table", "k_rk_fused_strip", t8, 24.0),
                                  ("adj_fused_step_const", "adj_fused_step_constA", "k_adj_fused_strip", t7, 40.0),
                                  ("adj_fused_step_nnY_tab", "adj_fused_step_Y_table", "k_adj_fused_strip", t7, 40.0)):
    f, names = med(tag + "_f64", pat)
    bz, _ = med(tag + "_busy", pat)
    fe, _ = med(tag + "_fetch", pat)
    wr, _ = med(tag + "_write", pat)
    if not f:
        continue
    add, mul, fma, tr = (f.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    flops = 64.0 * (add + mul + 2.0 * fma + tr)
    executed = tiles * 64 * 64 * 5
    e = {"kernel": names, "us_per_launch_back_to_back_under_rocprof": us(tag + "_f64"),
         "wave_insts_add_f64": add, "wave_insts_mul_f64": mul, "wave_insts_fma_f64": fma, "wave_insts_trans_f64": tr,
         "wave_insts_valu": f.get("SQ_INSTS_VALU"), "waves": f.get("SQ_WAVES"),
         "valu_insts_per_wave": f.get("SQ_INSTS_VALU", 0.0) / max(f.get("SQ_WAVES", 1.0), 1.0),
         "flops_executed_per_launch": flops, "executed_cell_stages_per_launch": executed, "tiles": tiles,
         "flop_per_executed_cell_stage": flops / executed, "useful_cell_stages_per_launch": 5 * cells,
         "halo_redundancy": executed / (5.0 * cells), "flop_per_useful_cell_stage": flops / (5.0 * cells),
         "algorithmic_bytes_per_cell": bpc}
    if bz:
        e["valu_busy_frac"] = valu_busy(bz)
        e["wait_frac_of_wave_cycles"] = bz.get("SQ_WAIT_INST_ANY", 0.0) / max(bz.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        e["sq_raw"] = bz
    if fe and wr:
        e["hbm_bytes_per_launch"] = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_per_cell"] = e["hbm_bytes_per_launch"] / cells
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_cell"] / bpc
        e["ratio"] = e["traffic_over_algorithmic"]
    e["valu_insts_per_useful_cell_stage"] = 64.0 * f.get("SQ_INSTS_VALU", 0.0) / (5.0 * cells)
    e["fma_share_of_f64_insts"] = fma / max(add + mul + fma + tr, 1.0)
    out[key] = e
    if key == "adj_fused_step_constA":
        out["adj_fused_step_64"] = e   # (the name bench.py's roofline_adjoint block reads)
for tag, key, pat, bpc in (("dhdt_nnA", "dhdt_nn_gridded_64", "k_dhdt", 32.0), ("rk_stage2_const", "rk_stage2_64", "k_rk_stage", 56.0)):
    fe, names = med(tag + "_fetch", pat)
    wr, _ = med(tag + "_write", pat)
    if fe and wr:
        by = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        out[key] = {"kernel": names, "cells": cells, "us_per_launch_back_to_back_under_rocprof": us(tag + "_fetch"), "hbm_bytes_per_launch": by,
                    "hbm_bytes_per_cell": by / cells, "algorithmic_bytes_per_cell": bpc, "ratio": by / (bpc * cells)}
print(json.dumps(out, indent=1))
