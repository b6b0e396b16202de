"""Turn the counter_collection.csv files of tools/pmc_roofline.sh into profiles/r03/pmc_roofline.json."""
import csv, sys, glob, collections, statistics, json, os

root = sys.argv[1]


def med(tag, pat):
    """median per counter over the dispatches of the kernels whose name contains `pat`"""
    agg = collections.defaultdict(list)
    names = set()
    for path in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    return {c: statistics.median(v) for c, v in agg.items()}, sorted(names)


N_SIMD, N_XCD, N_SE = 1024, 8, 32


def valu_busy(bz):
    """Fraction of the SIMD issue cycles of the launch that carried a VALU instruction: SQ_ACTIVE_INST_VALU counts quad-cycles
    summed over all SIMDs; the launch lasts GRBM_GUI_ACTIVE / 8 cycles (the counter is summed over the 8 XCDs; where a pass
    lacks it, SQ_BUSY_CYCLES / 32 -- summed over the 32 shader engines -- is the same clock to within 10 %)."""
    if "GRBM_GUI_ACTIVE" in bz:
        cyc = bz["GRBM_GUI_ACTIVE"] / N_XCD
    else:
        cyc = bz.get("SQ_BUSY_CYCLES", 0.0) / N_SE
    return bz.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / max(N_SIMD * cyc, 1.0)


cells8, cells32 = 8 * 1024 * 1024, 32 * 1024 * 1024
tiles8 = 8 * 19 * 19  # 54 x 54 output tiles of the 8-row strip kernel on 8 x 1024^2
out = {"workload_cells": cells8, "source": "tools/pmc_roofline.sh (rocprofv3 --pmc, one counter group per pass, kernel-trace only)",
       "fetch_correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); sizes in KiB"}
for law, key in (("nnA", "fused_step_nn_gridded"), ("const", "fused_step_constA")):
    f, names = med(f"fused_{law}_f64", "k_rk_fused_strip")
    bz, _ = med(f"fused_{law}_busy", "k_rk_fused_strip")
    fe, _ = med(f"fused_{law}_fetch", "k_rk_fused_strip")
    wr, _ = med(f"fused_{law}_write", "k_rk_fused_strip")
    if not f:
        continue
    add, mul, fma, tr = (f.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    flops = 64.0 * (add + mul + 2.0 * fma + tr)
    executed = tiles8 * 64 * 64 * 5
    e = {"kernel": names, "wave_insts_add_f64": add, "wave_insts_mul_f64": mul, "wave_insts_fma_f64": fma, "wave_insts_trans_f64": tr,
         "wave_insts_valu": f.get("SQ_INSTS_VALU"), "waves": f.get("SQ_WAVES"),
         "flops_executed_per_launch": flops, "executed_cell_stages_per_launch": executed,
         "flop_per_executed_cell_stage": flops / executed, "useful_cell_stages_per_launch": 5 * cells8,
         "halo_redundancy": executed / (5.0 * cells8)}
    if bz:
        e["valu_busy_frac"] = valu_busy(bz)
        e["valu_busy_definition"] = "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"
        e["sq_raw"] = bz
    if fe and wr:
        e["hbm_bytes_per_launch"] = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        e["hbm_bytes_per_cell"] = e["hbm_bytes_per_launch"] / cells8
    out[key] = e
for tag, key, pat, bpc in (("dhdt32_nnA", "dhdt_nn_gridded_32", "k_dhdt", 32.0), ("stage32", "rk_stage2_32", "k_rk_stage", 56.0),
                           ("vjpH32", "vjp_H_32", "k_vjp_H", 32.0), ("vjpth32", "vjp_theta_32", "k_vjp_theta", 24.0)):
    fe, names = med(tag + "_fetch", pat)
    wr, _ = med(tag + "_write", pat)
    if fe and wr:
        by = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
        out[key] = {"kernel": names, "cells": cells32, "hbm_bytes_per_launch": by, "hbm_bytes_per_cell": by / cells32,
                    "algorithmic_bytes_per_cell": bpc, "ratio": by / (bpc * cells32)}
fe, names = med("adjf8_fetch", "k_adj_fused_strip")
wr, _ = med("adjf8_write", "k_adj_fused_strip")
bz, _ = med("adjf8_busy", "k_adj_fused_strip")
if fe and wr:
    by = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
    e = {"kernel": names, "cells": cells8, "hbm_bytes_per_launch": by, "hbm_bytes_per_cell": by / cells8,
         "algorithmic_bytes_per_cell": 40.0, "ratio": by / (40.0 * cells8),
         "note": "dense launch on 8 x 1024^2 (320 MiB of lambda, H_j, H_j+1, B, lambda': partly inside the 256 MiB Infinity Cache)"}
    if bz:
        e["valu_busy_frac"] = valu_busy(bz)
        e["valu_insts_per_wave"] = bz.get("SQ_INSTS_VALU", 0) / max(bz.get("SQ_WAVES", 1), 1)
        e["sq_raw"] = bz
    out["adj_fused_step_8"] = e
print(json.dumps(out, indent=1))
