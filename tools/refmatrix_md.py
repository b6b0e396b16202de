"""tests/REFERENCE_MATRIX.md from gpurun_out/reference_matrix.jsonl (written by tests/test_gpu_reference_matrix.py on the GPU box):
python tools/refmatrix_md.py > tests/REFERENCE_MATRIX.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for l in open(os.path.join(ROOT, "gpurun_out", "reference_matrix.jsonl")):
    r = json.loads(l)
    rows[r["row"]] = r
NAMES = {
    "core2_discreteVJP": ("Core2", "88-91", "VJP (discrete) of SIA2D vs finite differences", "test_adjoint_SIA2D_row"),
    "core2_discreteVJP_sliding": ("Core2", "92-94", "VJP (discrete) of SIA2D with C>0", "test_adjoint_SIA2D_row"),
    "core2_continuousVJP": ("Core2", "95-96", "VJP (continuous) of SIA2D", "test_adjoint_SIA2D_row"),
    "core2_continuousVJP_sliding": ("Core2", "97-99", "VJP (continuous) of SIA2D with C>0", "test_adjoint_SIA2D_row"),
    "core2_discreteVJP_classical_scalar": ("Core2", "100-102", "VJP (discrete), classical scalar inversion", "test_adjoint_SIA2D_row"),
    "core2_discreteVJP_classical_gridded": ("Core2", "103-106", "VJP (discrete), classical gridded inversion (H part; theta: see note 5)", "test_adjoint_SIA2D_row"),
    "core3_discrete_discrete": ("Core3", "115-116", "Discrete adjoint with discrete VJP", "test_grad_finite_diff_row"),
    "core3_discrete_discrete_classical_scalar": ("Core3", "117-119", "... for scalar classical inversions", "test_grad_finite_diff_row"),
    "core3_discrete_discrete_IC": ("Core3", "120-122", "... (initial condition)", "test_grad_finite_diff_row"),
    "core3_discrete_continuousVJP": ("Core3", "123-124", "Discrete adjoint with continuous VJP", "test_grad_finite_diff_row"),
    "core3_continuous_discrete": ("Core3", "125-126", "Continuous adjoint with discrete VJP", "test_grad_finite_diff_row"),
    "core3_continuous_discrete_IC": ("Core3", "127-129", "... (initial condition)", "test_grad_finite_diff_row"),
    "core3_continuous_discrete_MB": ("Core3", "137-139", "... w/ discrete MB VJP, use_MB", "test_grad_finite_diff_row"),
    "core3_continuous_continuousVJP": ("Core3", "140-141", "Continuous adjoint with continuous VJP", "test_grad_finite_diff_row"),
    "core4_discrete_lossV": ("Core4", "158-160", "Discrete adjoint, LossV", "test_grad_finite_diff_row"),
    "core4_continuous_lossV_L2": ("Core4", "162-164", "Continuous adjoint, LossV (L2)", "test_grad_finite_diff_row"),
    "core4_continuous_lossV_log_abs": ("Core4", "165-167", "Continuous adjoint, LossV(LogSum, :abs)", "test_grad_finite_diff_row"),
    "core5_Dhybrid_continuous_discrete": ("Core5", "175-177", "target :D_hybrid, continuous adjoint, discrete VJP", "test_grad_finite_diff_row"),
    "core5_Dhybrid_continuous_continuousVJP": ("Core5", "178-180", "target :D_hybrid, continuous VJP", "test_grad_finite_diff_row"),
    "core6_D_continuous_discrete": ("Core6", "186-188", "target :D, continuous adjoint, discrete VJP", "test_grad_finite_diff_row"),
    "core6_D_continuous_continuousVJP": ("Core6", "189-191", "target :D, continuous VJP", "test_grad_finite_diff_row"),
    "core6_D_continuous_discrete_lossV": ("Core6", "192-194", "target :D, LossV", "test_grad_finite_diff_row"),
    "core7_D_customNN_lossV": ("Core7", "202-204", "target :D, custom NN, LossV (`:Linear`, see note 4)", "test_grad_finite_diff_row"),
    "core7_D_customNN_lossV:REFM_D_INTERP=None": ("Core7", "202-204", "  the same row with interpolation = :None", "test_grad_finite_diff_row"),
    "core8_multiloss_H": ("Core8", "210-212", "MultiLoss((LossH,), (0.4,))", "test_grad_finite_diff_row"),
    "core8_just_velocity_regularization": ("Core8", "213-215", "MultiLoss((VelocityRegularization,), (1e2,))", "test_grad_finite_diff_row"),
    "core8_H_and_velocity_regularization": ("Core8", "216-220", "MultiLoss((LossH, VelocityRegularization), (1e-2, 2e-1))", "test_grad_finite_diff_row"),
    "core8_rheology_regularization": ("Core8", "221-223", "RheologyRegularization, gridded classical inversion", "test_grad_finite_diff_row"),
    "core8_dhdt_discrete": ("Core8", "224-226", "LossDhdt, discrete adjoint, use_MB", "test_grad_finite_diff_row"),
    "core8_dhdt_continuous": ("Core8", "227-229", "LossDhdt, continuous adjoint, use_MB", "test_grad_finite_diff_row"),
    "core8_avgV_continuous": ("Core8", "233-237", "LossAvgV, continuous adjoint", "test_grad_finite_diff_row"),
    "core10_multiglacier": ("Core10", "257-259", "multiglacier, continuous adjoint", "test_grad_finite_diff_row"),
    "core10_multiglacier_IC": ("Core10", "260-262", "multiglacier (initial condition)", "test_grad_finite_diff_row"),
}
def fmt(x): return "%.1e" % abs(x)
print(open(os.path.join(ROOT, "tools", "refmatrix_head.md")).read())
print("| group | runtests.jl | reference test | test id (`tests/test_gpu_reference_matrix.py`) | reference `[ratio, angle, relerr]` | achieved on the device |")
print("|---|---|---|---|---|---|")
for key, (grp, lines, what, fn) in NAMES.items():
    base = key.split(":")[0]
    if key.startswith("core2"):
        got = []
        for part in ("H", "theta"):
            r = rows.get(key + ":" + part)
            if r: got.append("%s: %s, %s, %s" % (part, fmt(r["ratio"]), fmt(r["angle"]), fmt(r["relerr"])))
        thres = rows[key + ":H"]["thres"]
        print("| %s | :%s | %s | `%s[%s]` | %s | %s |" % (grp, lines, what, fn, key, thres, "; ".join(got)))
    else:
        r = rows[key]
        print("| %s | :%s | %s | `%s[%s]` | %s | %s, %s, %s |" % (grp, lines, what, fn, base, r["thres"], fmt(r["ratio"]), fmt(r["angle"]), fmt(r["relerr"])))
print(open(os.path.join(ROOT, "tools", "refmatrix_tail.md")).read())
