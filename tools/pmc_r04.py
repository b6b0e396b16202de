"""Turn the counter_collection.csv files of tools/pmc_r04.sh into profiles/r04/pmc_roofline.json."""
import csv, sys, glob, collections, statistics, json, os, re

root = sys.argv[1]
N_SIMD, N_XCD, N_SE = 1024, 8, 32


def med(tag, pat):
    """median per counter over the dispatches of the kernels whose name contains `pat`"""
    agg, names = collections.defaultdict(list), set()
    for path in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    return {c: statistics.median(v) for c, v in agg.items()}, sorted(names)


def us(tag):
    try:
        return float(re.search(r"([0-9.]+) us/launch", open(os.path.join(root, tag + ".log")).read()).group(1))
    except Exception:
        return None


def valu_busy(bz):
    cyc = bz["GRBM_GUI_ACTIVE"] / N_XCD if "GRBM_GUI_ACTIVE" in bz else bz.get("SQ_BUSY_CYCLES", 0.0) / N_SE
    return bz.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / max(N_SIMD * cyc, 1.0)


def f64_block(tag_f, tag_b, pat, cells, stages, executed=None):
    f, names = med(tag_f, pat)
    if not f:
        return None
    bz, _ = med(tag_b, pat)
    add, mul, fma, tr = (f.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    flops = 64.0 * (add + mul + 2.0 * fma + tr)
    e = {"kernel": names, "wave_insts_add_f64": add, "wave_insts_mul_f64": mul, "wave_insts_fma_f64": fma, "wave_insts_trans_f64": tr,
         "wave_insts_valu": f.get("SQ_INSTS_VALU"), "waves": f.get("SQ_WAVES"), "flops_per_launch": flops,
         "flop_per_useful_cell_stage": flops / (stages * cells), "valu_insts_per_useful_cell_stage": 64.0 * f.get("SQ_INSTS_VALU", 0.0) / (stages * cells),
         "fma_share_of_f64_insts": fma / max(add + mul + fma + tr, 1.0), "us_per_launch_under_pmc": us(tag_f)}
    if executed:
        e["executed_cell_stages_per_launch"] = executed
        e["flop_per_executed_cell_stage"] = flops / executed
        e["halo_redundancy"] = executed / float(stages * cells)
    t = us(tag_f)
    if t:
        e["achieved_TFLOPs"] = flops / (t * 1e-6) / 1e12
        e["frac_of_fp64_peak_78.6"] = e["achieved_TFLOPs"] / 78.6
    if bz:
        e["valu_busy_frac"] = valu_busy(bz)
        e["sq_raw"] = bz
    return e


def traffic(tag, pat, cells, bpc):
    fe, names = med(tag + "_fetch", pat)
    wr, _ = med(tag + "_write", pat)
    if not (fe and wr):
        return None
    by = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0
    return {"kernel": names, "cells": cells, "hbm_bytes_per_launch": by, "hbm_bytes_per_cell": by / cells,
            "algorithmic_bytes_per_cell": bpc, "ratio": by / (bpc * cells)}


G, n = 64, 1024
cells = G * n * n
tiles8 = G * 19 * 19
out = {"workload_cells": cells, "source": "tools/pmc_r04.sh (rocprofv3 --pmc, one counter group per pass, kernel-trace only)",
       "fetch_correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); sizes in KiB",
       "valu_busy_definition": "4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
for law, key, bpc in (("nnA", "fused_step_nn_gridded", 32.0), ("const", "fused_step_constA", 24.0)):
    e = f64_block(f"fused_{law}_f64", f"fused_{law}_busy", "k_rk_fused_strip", cells, 5, tiles8 * 64 * 64 * 5)
    if e:
        t = traffic(f"fused_{law}", "k_rk_fused_strip", cells, bpc)
        if t:
            e.update({k: t[k] for k in ("hbm_bytes_per_launch", "hbm_bytes_per_cell")})
        e["useful_cell_stages_per_launch"] = 5 * cells
        out[key] = e
for tag, key, pat, bpc in (("dhdt_nnA", "dhdt_nn_gridded_64", "k_dhdt", 32.0), ("stage", "rk_stage2_64", "k_rk_stage", 56.0)):
    t = traffic(tag, pat, cells, bpc)
    if t:
        out[key] = t
e = f64_block("adjf_f64", "adjf_busy", "k_adj_fused_strip", cells, 5, G * 19 * 23 * 64 * 56 * 5)
if e:
    t = traffic("adjf", "k_adj_fused_strip", cells, 40.0)
    if t:
        e.update(t)
    e["valu_insts_per_wave"] = (e.get("wave_insts_valu") or 0) / max(e.get("waves") or 1, 1)
    out["adj_fused_step_64"] = e
c512 = 8 * 512 * 512
for law in ("nnY", "nnU"):
    for tag, key, pat in (("adjs", "adj_stage2", "k_adj_stage"), ("fwds", "rk_stage2", "k_rk_stage")):
        e = f64_block(f"{tag}_{law}_f64", f"{tag}_{law}_busy", pat, c512, 1)
        if e:
            e["workload"] = "8 x 512^2, default 2-3-10-3-1 network inlined per dual node"
            out[f"{key}_{law}_8x512"] = e
e = f64_block("fwds_nnY16_f64", "fwds_nnY16_busy", "k_rk_stage", 8 * 1024 * 1024, 1)
if e:
    e["workload"] = "8 x 1024^2, 2-16-16-1 network inlined per dual node (BASELINE configs[2](ii))"
    out["rk_stage2_nnY16_8x1024"] = e
print(json.dumps(out, indent=1))
