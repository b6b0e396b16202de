"""Debug aid for test_gpu_fuzz.py (velocity draw): `python tests/fuzz_debug.py SEED...` on a GPU box prints, per glacier of a
seed, loss / gradient of the glacier solved alone against the oracle; oracle(), device() are the harness for closer looks."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import _odinn_import
gpu = _odinn_import.load()
import test_gpu_fuzz as F
from oracle import sia2d_oracle as O
from conftest import rel_l2

ADJ_TOL = 1e-8


def oracle(c, g, mode, nq=8):
    v = c["vel"]
    vspec = O.LossVSpec(component=v["component"], scale_loss=v["scale"], log_eps=v["log_eps"])
    cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=c["mbs"][g], mb_times=c["mbt"] if c["mbs"][g] is not None else (),
                      fixed_dt=c["dts"] if mode == "discrete_fixed" else None, h_log_eps=c["log_eps"])
    if mode == "continuous":
        return O.loss_and_grad_continuous(c["gls"][g], c["laws"][g], cfg, c["refs"][g], c["own"][g], O.ContinuousAdjointCfg(n_quadrature=nq),
                                          V_ref=v["Vref"][g], tV_ref=v["tV"][g], vspec=vspec, loss_kind=v["kind"], scaling=v["scaling"])
    return O.loss_and_grad_HV(c["gls"][g], c["laws"][g], cfg, c["refs"][g], c["own"][g], v["Vref"][g], v["tV"][g], vspec,
                              loss_kind=v["kind"], scaling=v["scaling"])

def device(c, idx, mode, sched=None, nq=8, kind=None):
    v = c["vel"]; ph = c["ph"]; G = len(idx)
    b = gpu.GlacierBatch([c["shapes"][i] for i in idx], [c["dxs"][i] for i in idx], [c["dys"][i] for i in idx],
                         phys=[gpu.PhysicalParameters(**c["phs"][i].__dict__) for i in idx], A=[c["As"][i] for i in idx], T=[c["Ts"][i] for i in idx])
    for k, i in enumerate(idx):
        b.set_fields(k, c["gls"][i].H0, c["gls"][i].B)
        b.set_reference(k, c["own"][i], c["refs"][i], 3)
        b.set_velocity_reference(k, v["tV"][i], [m[0] for m in v["Vref"][i]], [m[1] for m in v["Vref"][i]], [m[2] for m in v["Vref"][i]])
        if c["mbs"][i] is not None:
            m = c["mbs"][i]
            b.set_mass_balance(k, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
    if c["kind"] != O.LAW_CONST_A:
        b.set_law(c["kind"], c["gm"], c["th"])
        if c["kind"] == O.LAW_NN_A_GRIDDED:
            for k, i in enumerate(idx):
                b.set_T_field(k, c["laws"][i].T)
        if c["kind"] == O.LAW_NN_Y:
            b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if c["interp"][0] == "linear" else gpu._lib.GRAD_INTERP_NONE, c["interp"][1])
    b.set_surface_velocity_factor(c["fV"])
    lk = kind or v["kind"]
    b.set_loss({"V": gpu._lib.LOSS_V, "HV": gpu._lib.LOSS_HV, "H": gpu._lib.LOSS_H}[lk], v["component"], v["scale"], v["scaling"])
    b.set_velocity_loss_function(v["log_eps"])
    if c["log_eps"] is not None:
        b.set_thickness_loss_function(c["log_eps"])
    if sched:
        b.set_schedule(**sched)
    if mode == "continuous":
        out = b.loss_grad_continuous(c["common"], theta=c["th"], mb_times=c["mbt"], reltol=1e-8, n_quadrature=nq,
                                     adj_reltol=ADJ_TOL, adj_abstol=ADJ_TOL)
    elif mode == "discrete_fixed":
        out = b.loss_grad(c["common"], theta=c["th"], mb_times=c["mbt"], fixed_dt=c["dts"])
    else:
        out = b.loss_grad(c["common"], theta=c["th"], mb_times=c["mbt"], reltol=1e-8)
    lam0 = [b.lambda0(k) for k in range(G)]
    rev = getattr(b, "last_stats_rev", None) if mode == "continuous" else None
    b.close()
    return out + (lam0, rev)

if __name__ == "__main__":
    for seed in map(int, sys.argv[1:]):
        c = F._draw(gpu, seed, velocity=True)
        print("seed", seed, c["mode"], "law", c["kind"], c["vel"]["kind"], c["vel"]["component"], "sched", c["sched"], "mbt", c["mbt"],
              "dxs", c["dxs"], "dys", c["dys"], flush=True)
        for g in range(c["G"]):
            o = oracle(c, g, c["mode"])
            for sched in ({}, dict(c["sched"])):
                d = device(c, [g], c["mode"], sched)
                print("  g", g, c["shapes"][g], "sched", sched, "L rel", abs(d[0] - o[0]) / max(abs(o[0]), 1e-300), "g rel",
                      rel_l2(d[1], np.atleast_1d(o[1])) if np.linalg.norm(o[1]) > 0 else None, "ntV", len(c["vel"]["tV"][g]), flush=True)
