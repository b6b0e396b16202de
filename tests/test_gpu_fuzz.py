"""GPU: randomised parity of the whole gradient path.  Every seed draws a batch -- 1-3 glaciers of random (ragged) shapes,
square or rectangular cells, one of the five laws, with or without sliding, mass balance, per-glacier data times, L2Sum or
LogSum, a random non-default kernel schedule -- and an adjoint (DiscreteAdjoint on a fixed or adaptive step sequence,
ContinuousAdjoint; DiscreteVJP or ContinuousVJP), and compares loss and d loss / d theta of the HIP path through the C ABI with
the oracle's restatement of SIA2D_grad_batch! (src/inverse/SIA2D/gradient.jl:45-275, 276-539) glacier by glacier.

A second test draws the same batches with LossV / LossHV (velocity maps at some of the stops, :xy / :abs / LogSum, scaled or
not, the U law with a surface-velocity factor).

The fixed seeds below run with the suite; ODINN_FUZZ_SEEDS=a:b runs the seeds a..b-1 instead (exploration: 2 x 4000 seeds
take five minutes on eight workers, `-n 8`; ODINN_FUZZ_BIG=1 draws grids of up to 150 x 120 cells; add `--timeout 120`: one draw in a few thousand makes the numpy oracle crawl).  Three kinds of draws are skipped, each with its reason (`-rs`), because the
REFERENCE ALGORITHM's result is not a well-defined function of the inputs there -- no two correct implementations agree:
 * the positivity pattern of a snapshot differs between the device and the checker in cells below 1e-20 (an advancing margin
   leaves subnormal thicknesses behind; the mass-balance mask and dVelocity/dtheta of target :D test H > 0): ~3.5 % of the draws;
 * an adaptive solve whose gradient moves by more than the tolerance when the parameters move by 1e-12 (accept / reject
   sequences of the PID controller on a stability-limited problem);
 * a mismatch below 5e-3 on a draw whose forward solve carries subnormal thicknesses (see _subnormal_margin).
What the 8000 exploration seeds found in the library: NaN gradients of the Y law's `:Linear` branch when a quantile knot of
create_interpolation landed on a subnormal thickness (1 / (x1 - x0) overflowed, k_interp.hip; seeds 1746, 2450, 3206, 3351
are kept), and the second RHS of the reverse solve's initial-step heuristic reading the wrong snapshot interval when
tau_0 + dt_0 lies below the last interior snapshot (k_adj_itp)."""
import os

import numpy as np
import pytest

from conftest import rel_l2
from oracle import sia2d_oracle as O

pytestmark = pytest.mark.gpu

T0 = 2010.0
SCHEDULES = [dict(step_sc=0), dict(step_sc=1), dict(fused_tiles=1), dict(fused_tiles=2), dict(fused_tiles=3), dict(fused_tiles=4),
             dict(dhdt_strip=0), dict(vjph_strip=0), dict(vjph_strip=1), dict(vjpth_strip=0), dict(vjpth_strip=1),
             dict(snap_on_load=0), dict(adj_fused=0), dict(adj_skip=0), dict(adj_segs=0), dict(adj_rows=4), dict(adj_rows=7),
             dict(adj_theta_fused=0), dict(adj_rows=2), dict(adj_sc=0), dict(adj_sc=1), dict(adj_sc=1, adj_rows=2), dict(adj_sc=1, adj_rows=7),
             dict(interp_async=0), dict(interp_async=2)]


NAMED_SEEDS = [1746, 2450, 3206, 3351, 19220, 19681, 22159, 22782, 23649, 23721, 24379]


def _seeds():
    e = os.environ.get("ODINN_FUZZ_SEEDS")
    if e:
        a, b = e.split(":")
        return list(range(int(a), int(b)))
    # the fixed slice the driver's `pytest -m gpu` sees: 236 consecutive draws per test plus every seed that ever found a defect on
    # either side (tests/README.md lists what each one found): subnormal thickness at the margin / `:Linear` knots (1746, 2450, 3206,
    # 3351), a step count that is not a property of the algorithm (19220), the three stuck reverse solves (19681, 22159, 22782), the
    # stuck ill-conditioning probe (23649), a solve at the resolution of tau (23721), the miscompiled self-controlled Y-table step (24379)
    return list(range(236)) + NAMED_SEEDS



def _skip(test, seed, reason, tag, Lg=None, Lo=None, gg=None, go=None, extra=None):
    """pytest.skip with an audit trail: with ODINN_FUZZ_AUDIT=<file> every skipped draw appends one JSON line with the
    device-vs-checker error of the loss and of the gradient, so that the skip rate (~4 % of the exploration draws) can be
    inspected as a histogram instead of being trusted (tools/fuzz_audit.py; profiles/r04/fuzz_skips.*)."""
    path = os.environ.get("ODINN_FUZZ_AUDIT")
    if path:
        import json
        rec = {"test": test, "seed": int(seed), "reason": reason[:48], "law": tag.get("law", tag.get("kind")), "mode": tag.get("mode"),
               "loss_relerr": None if Lg is None else float(abs(Lg - Lo) / max(abs(Lo), 1e-300)),
               "grad_relerr": None if gg is None or np.linalg.norm(go) == 0 else float(rel_l2(gg, go))}
        if extra:
            rec.update(extra)
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    pytest.skip(reason)


def _audit():
    return bool(os.environ.get("ODINN_FUZZ_AUDIT"))

def _maybe_runtime_arch(seed, widths, acts):
    """One draw in five of the per-node-network laws takes a RUN-TIME architecture (law modes 2 / 6: the rolled evaluator over
    padded weight rows) -- the reference's gelu net 2-5-10-5-1 (test/test_grad_loss.jl:182-190), a tanh / softplus mix, or a net
    with a layer wider than 16.  Drawn from a stream of its own, so the other draws are what they were."""
    r = np.random.default_rng(68000 + seed)
    if r.random() >= 0.2:
        return widths, acts
    return [([2, 5, 10, 5, 1], [3, 3, 3, 2]), ([2, 4, 6, 1], [4, 1, 2]), ([2, 6, 20, 4, 1], [1, 3, 1, 2])][int(r.integers(0, 3))]


def _draw(gpu, seed, velocity=False):
    rng = np.random.default_rng((5000000 if velocity else 1000) + seed)
    G = int(rng.integers(1, 4))
    sliding = rng.random() < 0.2
    ph = O.Phys(n=3.2, C=7e-8, q=1.0) if sliding else O.Phys()
    # (a generator of its own, so that the draws of the seeds recorded below stay what they were) one glacier in four batches
    # has physical parameters of its own: exponent, sliding, eta0 -- the kernels' law mode is chosen per batch
    rng_ph = np.random.default_rng(91000 + seed)
    phs = [ph] * G
    if G > 1 and rng_ph.random() < 0.25:
        phs = [[ph, O.Phys(n=3.2, C=7e-8, q=1.0), O.Phys(), O.Phys(eta0=0.7), O.Phys(n=3.0, C=3e-8, q=1.0)][int(rng_ph.integers(0, 5))] for _ in range(G)]
    kind = [O.LAW_CONST_A, O.LAW_NN_A_SCALAR, O.LAW_NN_A_GRIDDED, O.LAW_NN_Y, O.LAW_NN_U][int(rng.integers(0, 5))]
    mode = ["discrete_fixed", "discrete_adaptive", "continuous"][int(rng.integers(0, 3))]
    vjp = "continuous" if rng.random() < 0.2 and not velocity else "discrete"
    log_eps = 0.1 if rng.random() < 0.25 else None
    k = int(rng.integers(3, 6))
    ragged = G > 1 and rng.random() < 0.4 and not velocity
    with_mb = rng.random() < 0.4
    # law
    om = gm = th = None
    interp = None
    if kind in (O.LAW_NN_A_SCALAR, O.LAW_NN_A_GRIDDED):
        widths, acts = [([1, 3, 10, 3, 1], [1, 1, 1, 2]), ([1, 16, 16, 1], [1, 1, 2]), ([1, 5, 7, 1], [3, 4, 2])][int(rng.integers(0, 3))]
        om = O.MLP(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
        gm = gpu.MLPSpec(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    elif kind == O.LAW_NN_Y:
        widths, acts = [([2, 3, 10, 3, 1], [1, 1, 1, 2]), ([2, 3, 1], [1, 2])][int(rng.integers(0, 2))]
        widths, acts = _maybe_runtime_arch(seed, widths, acts)
        pre = [(-25.0, 0.0), (0.0, 500.0)]
        om = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA)
        gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA)
        interp = ("linear", int(rng.choice([6, 20, 75]))) if rng.random() < 0.6 else ("none", 75)
    elif kind == O.LAW_NN_U:
        widths, acts = _maybe_runtime_arch(seed, [2, 3, 10, 3, 1], [1, 1, 1, 2])
        pre = [(0.0, 300.0), (0.0, 0.5)]
        om = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0)
        gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0)
    if om is not None:
        th = om.init_theta(rng) + 0.05 * rng.standard_normal(om.n_params)
    fV = 0.8 if velocity and rng.random() < 0.5 else 1.0
    shapes, dxs, dys, Ts, As, gls, own, refs, mbs, laws = [], [], [], [], [], [], [], [], [], []
    geo = []
    for g in range(G):
        nx = int(rng.integers(60, 72)) if rng.random() < 0.25 else int(rng.integers(12, 45))
        ny = int(rng.integers(16, 40))
        if os.environ.get("ODINN_FUZZ_BIG"):  # several strip tiles (54 x 46 / 54 x 54 outputs) in both directions; slow oracle
            nx, ny = int(rng.integers(60, 150)), int(rng.integers(40, 120))
        dx = float(rng.choice([40.0, 50.0, 100.0]))
        dy = dx if rng.random() < 0.6 else float(np.round(dx * rng.uniform(0.7, 1.4), 1))
        x = np.linspace(-1.0, 1.0, nx)[:, None]
        y = np.linspace(-1.0, 1.0, ny)[None, :]
        r2 = ((x - rng.uniform(-0.1, 0.1)) / rng.uniform(0.6, 0.85)) ** 2 + ((y - rng.uniform(-0.1, 0.1)) / rng.uniform(0.6, 0.85)) ** 2
        H0 = rng.uniform(60.0, 220.0) * np.sqrt(np.maximum(0.0, 1.0 - r2))
        B = 1500.0 - rng.uniform(0.02, 0.12) * x * (nx * dx / 2.0) + rng.uniform(0.0, 0.05) * y * (ny * dy / 2.0) \
            + 15.0 * np.sin(3.0 * x + rng.uniform(0, 6)) * np.cos(2.5 * y + rng.uniform(0, 6))
        if rng.random() < 0.3:  # an ice-free stripe and a few isolated cells
            H0[:, ny // 2] = 0.0
            H0[rng.integers(0, nx, 3), rng.integers(0, ny, 3)] = 5.0
        H0, B = np.asfortranarray(H0), np.asfortranarray(B + np.zeros_like(H0))
        T = float(rng.uniform(-20.0, -1.0))
        A = float(rng.uniform(1e-18, 4e-17))
        if kind == O.LAW_CONST_A:
            law = O.Law(kind=kind, A=A)
        elif kind == O.LAW_NN_A_GRIDDED:
            S = B + H0
            law = O.Law(kind=kind, mlp=om, theta=th, T=np.asfortranarray(T - 6.5e-3 * (O.avg(S) - S.mean())))
        elif kind == O.LAW_NN_Y:
            law = O.Law(kind=kind, mlp=om, theta=th, T=T, interpolation=interp[0], n_interp_half=interp[1])
        elif kind == O.LAW_NN_U:
            law = O.Law(kind=kind, mlp=om, theta=th, T=T, fV=fV)
        else:
            law = O.Law(kind=kind, mlp=om, theta=th, T=T)
        geo.append((nx, ny, dx, dy, H0, B, T, A, law))
    # a time scale the explicit scheme is stable on: dt = 0.15 min(dx, dy)^2 / max D over the batch; stops every 6 dt
    dts = min(0.15 * min(q[2], q[3]) ** 2 / max(O.max_diffusivity(q[4], q[5], q[2], q[3], phs[i], q[8]), 1e-30) for i, q in enumerate(geo))
    dts = float(min(dts, 1.0 / 480.0))
    step = 6.0 * dts
    common = [T0 + j * step for j in range(k)]
    for g in range(G):
        nx, ny, dx, dy, H0, B, T, A, law = geo[g]
        shapes.append((nx, ny)); dxs.append(dx); dys.append(dy); Ts.append(T); As.append(A); laws.append(law)
        gls.append(O.Glacier(H0, B, dx, dy, phs[g]))
        ts = list(common)
        if ragged:  # own interior data times; t0 and t1 are shared (every glacier covers the same tspan)
            inner = sorted(set(float(v) for v in rng.uniform(common[0] + 0.2 * step, common[-1] - 0.2 * step, int(rng.integers(1, 4)))))
            ts = [common[0]] + inner + [common[-1]]
        own.append(ts)
        refs.append([np.asfortranarray(np.maximum(H0 * (1.0 - 0.02 * j) + (H0 > 0) * rng.normal(0.0, 1.0, H0.shape), 0.0))
                     for j in range(len(ts))])
        if with_mb:
            S0 = B + np.maximum(H0, 0.0)
            ela = np.percentile(S0[H0 > 0], 60)
            grad = 6e-3
            mbs.append(O.MassBalance(mb0=np.asfortranarray(grad * (S0 - ela) * 30.0 * step), dmb_dS=grad * 30.0 * step,
                                     S_ref=np.asfortranarray(S0), mb_max=1.2 * 30.0 * step))
        else:
            mbs.append(None)
    # mass-balance times: stops every glacier has (the DiscreteAdjoint's requirement, gradient.jl:131) -- t1, and with a common
    # table sometimes an interior stop; the ContinuousAdjoint also takes times that are nobody's stop
    mbt = []
    if with_mb:
        mbt = [common[-1]]
        if not ragged and k > 3 and rng.random() < 0.5:
            mbt = [common[k // 2], common[-1]]
        if mode == "continuous" and rng.random() < 0.4:
            mbt = sorted(set(mbt + [common[0] + 0.37 * (common[-1] - common[0])]))
    sched = SCHEDULES[int(rng.integers(0, len(SCHEDULES)))] if rng.random() < 0.5 else {}
    vel = None
    if velocity:  # LossV / LossHV: velocity maps at some of the stops (the ContinuousAdjoint interpolates them over the whole tspan)
        comp = ["xy", "abs", "log"][int(rng.integers(0, 3))]
        vel = dict(kind="V" if rng.random() < 0.5 else "HV", component="abs" if comp == "log" else comp,
                   log_eps=0.1 if comp == "log" else None, scale=bool(rng.random() < 0.5), scaling=float(rng.uniform(0.3, 3.0)),
                   tV=[], Vref=[])
        for g in range(G):
            if mode == "continuous":
                tV = list(common) if rng.random() < 0.6 else [common[0], common[-1]]
            else:
                tV = [t for j, t in enumerate(common) if j >= 1 and (j % 2 == g % 2 or j == k - 1)]
            Vx0, Vy0, _ = O.V_from_H(gls[g].H0, gls[g].B, dxs[g], dys[g], phs[g], laws[g])
            maps = []
            for _t in tV:
                Vx = np.asfortranarray(Vx0 * rng.uniform(0.8, 1.3) + 0.05 * np.abs(Vx0).max() * rng.standard_normal(Vx0.shape) * (Vx0 != 0))
                Vy = np.asfortranarray(Vy0 * rng.uniform(0.8, 1.3) + 0.05 * np.abs(Vy0).max() * rng.standard_normal(Vy0.shape) * (Vy0 != 0))
                maps.append((np.asfortranarray(np.sqrt(Vx ** 2 + Vy ** 2)), Vx, Vy))
            vel["tV"].append(tV)
            vel["Vref"].append(maps)
    return dict(G=G, ph=ph, kind=kind, mode=mode, vjp=vjp, log_eps=log_eps, common=common, own=own, ragged=ragged, mbt=mbt,
                om=om, gm=gm, th=th, interp=interp, shapes=shapes, dxs=dxs, dys=dys, Ts=Ts, As=As, gls=gls, refs=refs, mbs=mbs,
                laws=laws, sched=sched, step=step, dts=dts, vel=vel, fV=fV, phs=phs)


# cap of the CHECKER's reverse solve: the draws' reverse solves take 10 ... a few hundred steps (the three draws in 23 200 seeds on which
# the reverse ODE's step size collapses -- aggregated terms, seeds 19681, 22159, 22782 -- cost the numpy checker its full 10^6 attempts
# before both sides got the stuck-solve exit, ODINN_ERR_DTMIN: they end in a fraction of a second now)
_ORACLE_REV_MAXITERS = 30000


_CHECKER = {"stall": 0}  # longest run of attempts without progress in the checker's UNPERTURBED solve of the current draw


def _oracle_gradient_or_skip(test, seed, tag, c, nq, parts=None):
    """_oracle_gradient; a draw on which the CHECKER's own adaptive solve cannot finish (maxiters, dt <= eps(t)) is skipped (audited).
    Seen three times in 23 200 seeds (aggregated terms, seeds 19681, 22159, 22782: ContinuousAdjoint over stops a few 1e-4 yr apart -- after
    a mass-balance stop the reverse solve rejects 14 steps in a row, dt reaches the resolution of tau and never recovers); the device was run on
    all three by hand and ends the same way: ODINN_ERR_DTMIN "the reverse solve is stuck at tau = ... 256 attempts in a row without advancing tau"."""
    O.SOLVE_DIAG["max_stall"] = 0
    _CHECKER["stall"] = 0
    try:
        out = _oracle_gradient(c, nq, parts=parts)
        _CHECKER["stall"] = O.SOLVE_DIAG["max_stall"]  # of THIS solve: the perturbed solves of _ill_conditioned keep adding to SOLVE_DIAG
        return out
    except RuntimeError as e:
        if "maxiters" not in str(e) and "dtmin" not in str(e):
            raise
        _skip(test, seed, "the checker's own solve reaches maxiters / dtmin", tag)


def _device_or_skip(test, seed, tag, fn):
    """The device's gradient; a solve the device gives up as stuck (ODINN_ERR_DTMIN) on a draw whose CHECKER solve itself went through an
    episode at the resolution of its time variable (>= 6 attempts in a row without advancing it, O.SOLVE_DIAG: ordinary solves see 3 - 4 rejections in a row at most) is an audited skip: in
    that regime the error estimate alternates between 1e-8 and 1e+1 from one attempt to the next and round-off decides which side gets
    through (seed 23721: the checker does after runs of up to 9 such attempts, the device never does -- maxiters with the exit disabled)."""
    try:
        return fn()
    except Exception as e:
        if "is stuck" in str(e) and _CHECKER["stall"] >= 6:
            _skip(test, seed, "solve at the resolution of its time variable: round-off decides", tag)
        raise


def _oracle_gradient(c, nq, rel_perturbation=0.0, parts=None):
    """(loss, d loss / d theta) of the draw by the oracle, summed over the glaciers; the parameters (theta, or the glaciers' A)
    scaled by 1 + rel_perturbation."""
    import dataclasses

    mode, v = c["mode"], c["vel"]
    Lo, go = 0.0, 0.0
    for g in range(c["G"]):
        law = c["laws"][g]
        if rel_perturbation:
            law = (dataclasses.replace(law, A=law.A * (1.0 + rel_perturbation)) if law.kind == O.LAW_CONST_A
                   else dataclasses.replace(law, theta=law.theta * (1.0 + rel_perturbation)))
        cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=c["mbs"][g], mb_times=c["mbt"] if c["mbs"][g] is not None else (),
                          fixed_dt=c["dts"] if mode == "discrete_fixed" else None, h_log_eps=c["log_eps"],
                          **(c["cfg_extra"][g] if c.get("cfg_extra") else {}))
        if v is None and mode == "continuous":
            out = O.loss_and_grad_continuous(c["gls"][g], law, cfg, c["refs"][g], c["own"][g], O.ContinuousAdjointCfg(n_quadrature=nq, maxiters=_ORACLE_REV_MAXITERS),
                                             vjp=c["vjp"])
        elif v is None:
            out = O.loss_and_grad(c["gls"][g], law, cfg, c["refs"][g], c["own"][g], vjp=c["vjp"])
        else:
            vspec = O.LossVSpec(component=v["component"], scale_loss=v["scale"], log_eps=v["log_eps"])
            if mode == "continuous":
                out = O.loss_and_grad_continuous(c["gls"][g], law, cfg, c["refs"][g], c["own"][g], O.ContinuousAdjointCfg(n_quadrature=nq, maxiters=_ORACLE_REV_MAXITERS),
                                                 V_ref=v["Vref"][g], tV_ref=v["tV"][g], vspec=vspec, loss_kind=v["kind"],
                                                 scaling=v["scaling"])
            else:
                out = O.loss_and_grad_HV(c["gls"][g], law, cfg, c["refs"][g], c["own"][g], v["Vref"][g], v["tV"][g], vspec,
                                         loss_kind=v["kind"], scaling=v["scaling"])
        Lo += out[0]
        go = go + np.atleast_1d(out[1])
        if parts is not None:
            parts.append(out)
    return Lo, go


def _ill_conditioned(c, nq, go, gtol):
    """Adaptive solves: the accept / reject sequence of the PID controller is a discontinuous function of the data, and on a
    stability-limited problem (what the draw's time scale makes of it) a relative change of 1e-12 in the parameters can move
    the reference algorithm's own gradient by 1e-2.  True when the oracle's gradient at theta (1 + 1e-12) differs from its
    gradient at theta by more than a quarter of the comparison's tolerance: no two implementations agree better than that."""
    if np.linalg.norm(go) == 0:
        return False
    try:
        _, g1 = _oracle_gradient(c, nq, 1e-12)
    except RuntimeError as e:  # (the perturbed run gets stuck / runs out of attempts where the unperturbed one did not: seed 23649)
        if "maxiters" in str(e) or "dtmin" in str(e):
            return True
        raise
    return rel_l2(g1, go) > 0.25 * gtol


def _margins_agree(b, c, mode):
    """The reference's gradient is DISCONTINUOUS in the state where it tests H > 0 (the mass-balance mask, VJPs.jl:124-126; the
    factor (Hbar > 0) of dVelocity/dtheta in target :D, target_D_pure.jl:246-255): an advancing margin leaves subnormal
    thicknesses (1e-320) behind, and whether such a cell holds 1e-320 or 0 is a matter of rounding (FMA contraction).  Returns
    False when the positivity pattern of a snapshot differs between the device and the checker -- only ever on cells below
    1e-20 (asserted) -- in which case the two gradients are not comparable and the seed is skipped."""
    same = True
    for g in range(c["G"]):
        cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=c["mbs"][g], mb_times=c["mbt"] if c["mbs"][g] is not None else (),
                          fixed_dt=c["dts"] if mode == "discrete_fixed" else None)
        snaps, _, _ = O.forward(c["gls"][g], c["laws"][g], cfg)
        for j, Ho in enumerate(snaps):
            Hg = b.snapshot(g, j)
            for a, d in ((Ho, Hg), (O.avg(np.maximum(Ho, 0.0)), O.avg(np.maximum(Hg, 0.0)))):
                diff = (a > 0) != (d > 0)
                if diff.any():
                    assert max(a[diff].max(), d[diff].max()) < 1e-20, (g, j, a[diff].max(), d[diff].max())
                    same = False
    return same


def _subnormal_margin(c, mode):
    """True when a snapshot of the oracle's forward solve, or its state right before a mass balance, holds subnormal-range
    thicknesses (0 < H < 1e-290: what an advancing margin leaves in the cells it has not reached yet).  Such values carry no
    significant digits -- two correct forward solves differ in them by factors -- yet the reference's gradient tests them
    against zero (the mass-balance mask H > 0 that zeroes lambda on a cell that melts completely, VJPs.jl:124-148; the factor
    (Hbar > 0) of dVelocity/dtheta in target :D, evaluated on the linearly INTERPOLATED state, where s * 1e-323 underflows to 0
    or not): where this is the case a mismatch at the 1e-4 level is the reference algorithm's discontinuity, not an error."""
    for g in range(c["G"]):
        cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=c["mbs"][g], mb_times=c["mbt"] if c["mbs"][g] is not None else (),
                          fixed_dt=c["dts"] if mode == "discrete_fixed" else None)
        snaps, _, inc = O.forward(c["gls"][g], c["laws"][g], cfg)
        fields = list(snaps)
        for tm, d in inc.items():
            past = [j for j, t in enumerate(c["own"][g]) if t == tm]
            if past:
                fields.append(snaps[past[0]] - d)
        for H in fields:
            if np.any((H > 0) & (H < 1e-290)):
                return True
    return False


@pytest.mark.parametrize("seed", _seeds())
def test_random_batch_gradient_matches_the_oracle(gpu, monkeypatch, seed):
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB", "ODINN_FUZZ_AUDIT", "ODINN_FUZZ_BIG"):
            monkeypatch.delenv(key, raising=False)
    c = _draw(gpu, seed)
    G, ph, kind, mode = c["G"], c["ph"], c["kind"], c["mode"]
    tag = {k: c[k] for k in ("G", "kind", "mode", "vjp", "log_eps", "ragged", "mbt", "interp", "shapes", "dxs", "dys", "sched")}
    tag["sliding"] = ph.C != 0.0
    nq = 8
    dt_fixed = c["dts"]
    # ---- oracle, glacier by glacier
    per = []
    Lo, go = _oracle_gradient_or_skip("gradient", seed, tag, c, nq, parts=per)
    ill = mode != "discrete_fixed" and _ill_conditioned(c, nq, go, 2e-5)
    ILL = "the checker's own gradient moves by more than the tolerance under a 1e-12 perturbation of the parameters"
    if ill and not _audit():
        pytest.skip(ILL)
    # ---- the HIP path
    b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
    try:
        for g in range(G):
            b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
            b.set_reference(g, c["own"][g], c["refs"][g], 3)
            if c["mbs"][g] is not None:
                m = c["mbs"][g]
                b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
            if c["ragged"]:
                b.set_glacier_stops(g, c["own"][g])
        if kind != O.LAW_CONST_A:
            b.set_law(kind, c["gm"], c["th"])
            if kind == O.LAW_NN_A_GRIDDED:
                for g in range(G):
                    b.set_T_field(g, c["laws"][g].T)
            if kind == O.LAW_NN_Y:
                b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if c["interp"][0] == "linear" else gpu._lib.GRAD_INTERP_NONE,
                                         c["interp"][1])
        if c["vjp"] == "continuous":
            b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
        if c["log_eps"] is not None:
            b.set_thickness_loss_function(c["log_eps"])
        if c["sched"]:
            b.set_schedule(**c["sched"])
        union = sorted(set(t for ts in c["own"] for t in ts))
        if mode == "continuous":
            Lg, gg = _device_or_skip("gradient", seed, tag, lambda: b.loss_grad_continuous(union, theta=c["th"], mb_times=c["mbt"], reltol=1e-8, n_quadrature=nq))
        elif mode == "discrete_fixed":
            Lg, gg = b.loss_grad(union, theta=c["th"], mb_times=c["mbt"], fixed_dt=dt_fixed)
        else:
            Lg, gg = b.loss_grad(union, theta=c["th"], mb_times=c["mbt"], reltol=1e-8)
        comparable = _margins_agree(b, c, mode)
        loss_g, G_g = b.grad_parts()  # odinn_get_grad_parts: per-glacier loss and dL/dA (A-type laws)
        lam0 = [b.lambda0(g) for g in range(G)]
    finally:
        b.close()
    if ill:  # (audit runs only: the draw went through the device although the checker itself is ill-conditioned on it)
        _skip("gradient", seed, ILL, tag, Lg, Lo, gg, go)
    if not comparable:
        _skip("gradient", seed, "subnormal margin cells differ between the device and the checker: the gradient is discontinuous there", tag, Lg, Lo, gg, go)
    if mode == "discrete_fixed":  # the per-glacier results (PerGlacierModel slots, Model.jl:214-216; lambda(t0) of the IC gradient)
        for g in range(G):
            assert abs(loss_g[g] - per[g][0]) <= 1e-10 * max(abs(per[g][0]), 1e-300), (tag, g, "loss of the glacier")
            if kind == O.LAW_CONST_A and per[g][1][0] != 0.0:
                assert abs(G_g[g] - per[g][1][0]) <= 1e-8 * abs(per[g][1][0]), (tag, g, "dL/dA of the glacier")
            if np.linalg.norm(per[g][2]) > 0:
                assert rel_l2(lam0[g], per[g][2]) < (1e-6 if kind in (O.LAW_NN_Y, O.LAW_NN_U) else 1e-8), (tag, g, "lambda(t0)")
    # (Y and U laws: the reference's own finite-difference steps in dD/dH, target_D_hybrid.jl:58-71, target_D_pure.jl:105-137,
    #  amplify rounding differences to 1e-8)
    ltol, gtol = (1e-10, 1e-7 if kind in (O.LAW_NN_Y, O.LAW_NN_U) else 1e-8) if mode == "discrete_fixed" else (1e-6, 2e-5)
    assert abs(Lg - Lo) <= ltol * max(abs(Lo), 1e-300), (tag, Lg, Lo)
    if np.linalg.norm(go) > 0:
        if not rel_l2(gg, go) < gtol and rel_l2(gg, go) < 5e-3 and _subnormal_margin(c, mode):
            _skip("gradient", seed, "gradient differs at the level of the reference's own discontinuity at subnormal margin cells", tag, Lg, Lo, gg, go)
        assert rel_l2(gg, go) < gtol, (tag, rel_l2(gg, go))
    else:
        assert np.linalg.norm(gg) == 0, tag


@pytest.mark.parametrize("seed", _seeds())
def test_random_batch_velocity_loss_gradient_matches_the_oracle(gpu, monkeypatch, seed):
    """The same draw with LossV / LossHV (Losses.jl:293-440): velocity maps at some of the stops, :xy / :abs / LogSum, scaled or
    not, every law (the U law with a surface-velocity factor), both adjoints."""
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB", "ODINN_FUZZ_AUDIT", "ODINN_FUZZ_BIG"):
            monkeypatch.delenv(key, raising=False)
    c = _draw(gpu, seed, velocity=True)
    G, ph, kind, mode, v = c["G"], c["ph"], c["kind"], c["mode"], c["vel"]
    tag = {k: c[k] for k in ("G", "kind", "mode", "log_eps", "mbt", "interp", "shapes", "dxs", "dys", "sched", "fV")}
    tag.update({k: v[k] for k in ("kind", "component", "log_eps", "scale")}, sliding=ph.C != 0.0, law=kind)
    nq = 8
    Lo, go = _oracle_gradient_or_skip("velocity", seed, tag, c, nq)
    ill = mode != "discrete_fixed" and _ill_conditioned(c, nq, go, 2e-5)
    ILL = "the checker's own gradient moves by more than the tolerance under a 1e-12 perturbation of the parameters"
    if ill and not _audit():
        pytest.skip(ILL)
    b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
    try:
        for g in range(G):
            b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
            b.set_reference(g, c["own"][g], c["refs"][g], 3)
            b.set_velocity_reference(g, v["tV"][g], [m[0] for m in v["Vref"][g]], [m[1] for m in v["Vref"][g]], [m[2] for m in v["Vref"][g]])
            if c["mbs"][g] is not None:
                m = c["mbs"][g]
                b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
        if kind != O.LAW_CONST_A:
            b.set_law(kind, c["gm"], c["th"])
            if kind == O.LAW_NN_A_GRIDDED:
                for g in range(G):
                    b.set_T_field(g, c["laws"][g].T)
            if kind == O.LAW_NN_Y:
                b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if c["interp"][0] == "linear" else gpu._lib.GRAD_INTERP_NONE,
                                         c["interp"][1])
        b.set_surface_velocity_factor(c["fV"])
        b.set_loss(gpu._lib.LOSS_V if v["kind"] == "V" else gpu._lib.LOSS_HV, v["component"], v["scale"], v["scaling"])
        b.set_velocity_loss_function(v["log_eps"])
        if c["log_eps"] is not None:
            b.set_thickness_loss_function(c["log_eps"])
        if c["sched"]:
            b.set_schedule(**c["sched"])
        if mode == "continuous":
            Lg, gg = _device_or_skip("velocity", seed, tag, lambda: b.loss_grad_continuous(c["common"], theta=c["th"], mb_times=c["mbt"], reltol=1e-8, n_quadrature=nq))
        elif mode == "discrete_fixed":
            Lg, gg = b.loss_grad(c["common"], theta=c["th"], mb_times=c["mbt"], fixed_dt=c["dts"])
        else:
            Lg, gg = b.loss_grad(c["common"], theta=c["th"], mb_times=c["mbt"], reltol=1e-8)
        comparable = _margins_agree(b, c, mode)
    finally:
        b.close()
    if ill:  # (audit runs only: the draw went through the device although the checker itself is ill-conditioned on it)
        _skip("velocity", seed, ILL, tag, Lg, Lo, gg, go)
    if not comparable:
        _skip("velocity", seed, "subnormal margin cells differ between the device and the checker: the gradient is discontinuous there", tag, Lg, Lo, gg, go)
    ltol, gtol = (1e-10, 1e-7 if kind in (O.LAW_NN_Y, O.LAW_NN_U) else 1e-8) if mode == "discrete_fixed" else (1e-6, 2e-5)
    assert abs(Lg - Lo) <= ltol * max(abs(Lo), 1e-300), (tag, Lg, Lo)
    if np.linalg.norm(go) > 0:
        if not rel_l2(gg, go) < gtol and rel_l2(gg, go) < 5e-3 and _subnormal_margin(c, mode):
            _skip("velocity", seed, "gradient differs at the level of the reference's own discontinuity at subnormal margin cells", tag, Lg, Lo, gg, go)
        assert rel_l2(gg, go) < gtol, (tag, rel_l2(gg, go))
    else:
        assert np.linalg.norm(gg) == 0, tag


def _draw_seam(gpu, seed):
    rng = np.random.default_rng(77000 + seed)
    nx = int(rng.choice([3, 4, 17, 63, 64, 65, 66, 130])) if rng.random() < 0.4 else int(rng.integers(3, 100))
    ny = int(rng.choice([3, 4, 15, 16, 17, 33])) if rng.random() < 0.4 else int(rng.integers(3, 70))
    dx = float(rng.choice([25.0, 50.0, 100.0, 200.0]))
    dy = dx if rng.random() < 0.5 else float(np.round(dx * rng.uniform(0.6, 1.6), 2))
    ph = [O.Phys(), O.Phys(n=3.2, C=7e-8, q=1.0), O.Phys(n=2.5), O.Phys(eta0=0.7), O.Phys(n=3.0, C=3e-8, p=2.0, q=1.0),
          O.Phys(n=4.0, eta0=1.3)][int(rng.integers(0, 6))]
    kind = [O.LAW_CONST_A, O.LAW_NN_A_SCALAR, O.LAW_NN_A_GRIDDED, O.LAW_NN_Y, O.LAW_NN_U][int(rng.integers(0, 5))]
    x = np.linspace(-1.0, 1.0, nx)[:, None]
    y = np.linspace(-1.0, 1.0, ny)[None, :]
    style = int(rng.integers(0, 4))
    # (off-centre: on an exactly symmetric field the nodes that tie with max(Hbar) bit for bit decide the quantile knots of
    #  create_interpolation)
    H = rng.uniform(20.0, 300.0) * np.maximum(0.0, 1.0 - ((x - rng.uniform(-0.13, 0.13)) / rng.uniform(0.5, 1.5)) ** 2
                                              - ((y - rng.uniform(-0.13, 0.13)) / rng.uniform(0.5, 1.5)) ** 2) + 0.0 * x * y
    if style == 1:
        H = H + rng.normal(0.0, 5.0, H.shape)                       # negative entries where the ice is thin or absent
    elif style == 2:
        H = np.round(H / 10.0) * 10.0                               # plateaus: equal neighbours, ties in the clamp
    elif style == 3:
        H = H * (rng.random(H.shape) > 0.3)                         # holes
    B = 1000.0 + rng.uniform(-0.2, 0.2) * x * nx * dx / 2 + rng.uniform(-0.2, 0.2) * y * ny * dy / 2 \
        + rng.uniform(0.0, 30.0) * np.sin(4.0 * x) * np.cos(3.0 * y)
    if style == 2 and rng.random() < 0.5:
        B = np.round(B / 10.0) * 10.0                               # ... with S = B + H on a lattice too
    H, B = np.asfortranarray(H), np.asfortranarray(B + np.zeros_like(H))
    T = float(rng.uniform(-20.0, -1.0))
    A = float(rng.uniform(1e-18, 4e-17))
    om = gm = th = None
    interp = None
    if kind in (O.LAW_NN_A_SCALAR, O.LAW_NN_A_GRIDDED):
        widths, acts = [([1, 3, 10, 3, 1], [1, 1, 1, 2]), ([1, 16, 16, 1], [1, 1, 2]), ([1, 5, 7, 1], [3, 4, 2])][int(rng.integers(0, 3))]
        om = O.MLP(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
        gm = gpu.MLPSpec(widths, acts, None, O.POST_AFFINE, ph.minA, ph.maxA)
    elif kind == O.LAW_NN_Y:
        widths, acts = [([2, 3, 10, 3, 1], [1, 1, 1, 2]), ([2, 3, 1], [1, 2]), ([2, 16, 16, 1], [1, 1, 2])][int(rng.integers(0, 3))]
        widths, acts = _maybe_runtime_arch(seed, widths, acts)
        pre = [(-25.0, 0.0), (0.0, 500.0)]
        om = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA)
        gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, ph.maxA)
        interp = ("linear", int(rng.choice([3, 10, 75]))) if rng.random() < 0.5 else ("none", 75)
    elif kind == O.LAW_NN_U:
        widths, acts = _maybe_runtime_arch(seed, [2, 3, 10, 3, 1], [1, 1, 1, 2])
        pre = [(0.0, 300.0), (0.0, 0.5)]
        om = O.MLP(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0)
        gm = gpu.MLPSpec(widths, acts, pre, O.POST_EXPMAX, 0.0, 50.0)
        if np.random.default_rng(66000 + seed).random() < 0.35:
            # SIA2D_D_target(interpolation = :Linear): the law's node grid spans [0, 100] on both axes and does not extrapolate
            # (Laws.jl:128-131): thickness kept below 100 m
            interp = ("linear", int(np.random.default_rng(67000 + seed).choice([2, 5, 100])))
            H = np.asfortranarray(H * (95.0 / max(float(H.max()), 95.0)))
            if style == 2:  # stay on the lattice: (B + H) - B == H exactly, so that the slope clamp sees exact ties, not near-ties
                H = np.asfortranarray(np.round(H / 10.0) * 10.0)
    if om is not None:
        th = om.init_theta(rng) + 0.05 * rng.standard_normal(om.n_params)
    if kind == O.LAW_CONST_A:
        law = O.Law(kind=kind, A=A)
    elif kind == O.LAW_NN_A_GRIDDED:
        law = O.Law(kind=kind, mlp=om, theta=th, T=np.asfortranarray(T + rng.uniform(-3, 3, (nx - 1, ny - 1))))
    elif kind == O.LAW_NN_Y:
        law = O.Law(kind=kind, mlp=om, theta=th, T=T, interpolation=interp[0], n_interp_half=interp[1])
    elif kind == O.LAW_NN_U and interp is not None:
        law = O.Law(kind=kind, mlp=om, theta=th, T=T, interpolation=interp[0], n_interp_half=interp[1])
    else:
        law = O.Law(kind=kind, mlp=om, theta=th, T=T)
    lam = rng.standard_normal(H.shape)
    return dict(nx=nx, ny=ny, dx=dx, dy=dy, ph=ph, kind=kind, style=style, H=H, B=B, T=T, A=A, om=om, gm=gm, th=th, interp=interp,
                law=law, lam=lam)


def _gradS(H, B, dx, dy):
    S = B + np.maximum(H, 0.0)
    gx = O.avg_y(O.diff_x(S) / dx)
    gy = O.avg_x(O.diff_y(S) / dy)
    return np.sqrt(gx ** 2 + gy ** 2)


def _seam_seeds():
    e = os.environ.get("ODINN_FUZZ_SEEDS")
    if e:
        a, b = e.split(":")
        return list(range(int(a), int(b)))
    return list(range(40))


@pytest.mark.parametrize("seed", _seam_seeds())
def test_random_seams_match_the_oracle(gpu, monkeypatch, seed):
    """The four seams of the path on random inputs: SIA2D! (adjoint.jl:52-97), VJP_lambda_dSIA/dH discrete (:99-151) and
    continuous (:442-553), VJP_lambda_dSIA/dtheta (:178-255) -- random shapes from 3 x 3 up across the 64-lane and tile
    boundaries, rectangular cells, n != 3, sliding with several (p, q), eta0 != 1, the five laws, thickness fields with
    ice-free areas, negative entries (clamped, :52), integer values (ties in the slope clamp, inversion_utils.jl:22-43) and
    cells on the boundary ring."""
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB", "ODINN_FUZZ_AUDIT", "ODINN_FUZZ_BIG"):
            monkeypatch.delenv(key, raising=False)
    q = _draw_seam(gpu, seed)
    nx, ny, dx, dy, ph, kind, style, H, B, T, A = (q[k] for k in ("nx", "ny", "dx", "dy", "ph", "kind", "style", "H", "B", "T", "A"))
    gm, th, interp, law, lam = (q[k] for k in ("gm", "th", "interp", "law", "lam"))
    tag = dict(shape=(nx, ny), dx=dx, dy=dy, ph=ph, kind=kind, style=style, interp=interp)
    b = gpu.GlacierBatch([(nx, ny)], [dx], [dy], phys=[gpu.PhysicalParameters(**ph.__dict__)], A=[A], T=[T])
    try:
        b.set_fields(0, H, B)
        if kind != O.LAW_CONST_A:
            b.set_law(kind, gm, th)
            if kind == O.LAW_NN_A_GRIDDED:
                b.set_T_field(0, law.T)
            if kind == O.LAW_NN_Y or (kind == O.LAW_NN_U and interp is not None):
                b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if interp[0] == "linear" else gpu._lib.GRAD_INTERP_NONE, interp[1])
        dH = b.dhdt(0, H)
        vH = b.vjp_H(0, lam, H)
        vT = np.atleast_1d(b.vjp_theta(0, lam, H))
        b.set_vjp_method(gpu._lib.VJP_CONTINUOUS)
        vHc = b.vjp_H(0, lam, H)
        vTc = np.atleast_1d(b.vjp_theta(0, lam, H))
        # surface velocity (Huginn.surface_V, adjoint.jl:268-413) and the mass-balance seams (VJPs.jl:107-151)
        rng2 = np.random.default_rng(88000 + seed)
        fV = 0.8 if rng2.random() < 0.5 else 1.0
        law.fV = fV
        b.set_surface_velocity_factor(fV)
        w1, w2 = rng2.standard_normal(H.shape), rng2.standard_normal(H.shape)
        Vx, Vy = b.surface_V(0, H)
        sH = b.surface_V_vjp_H(0, w1, w2, H)
        sT = np.atleast_1d(b.surface_V_vjp_theta(0, w1, w2, H))
        S0 = B + np.maximum(H, 0.0)
        mb = O.MassBalance(mb0=np.asfortranarray(rng2.uniform(-2.0, 1.0) + 4e-3 * (S0 - S0.mean())), dmb_dS=float(rng2.choice([0.0, 4e-3])),
                           S_ref=np.asfortranarray(S0 + rng2.normal(0.0, 3.0, H.shape)), mb_max=float(rng2.choice([0.5, 1e9])))
        b.set_mass_balance(0, mb.mb0, mb.dmb_dS, mb.S_ref, mb.mb_max)
        Hn, MB = b.mb_apply(0, H)
        mV = b.mb_vjp_H(0, lam, H)
        lawv = np.atleast_1d(b.eval_law(0, H))
    finally:
        b.close()

    def close(a, r, tol, what):
        r = np.atleast_1d(r)
        if not np.all(np.isfinite(r)):
            # an exponent below 3 (n < 3, or sliding with p < 3) on a field with nodes where grad S = 0 exactly (the lattice
            # style): the reference's beta = dD/d|grad S| / |grad S| carries |grad S|^(n - 3) (target_A.jl:46-62) -- Inf, and
            # Inf * 0 = NaN in its VJP; nothing to compare (the device returns NaN or the limit, as its FMA contraction falls)
            assert ph.n < 3.0 or (ph.C > 0.0 and ph.p < 3.0), ("oracle not finite", ph, style, (nx, ny), what)
            return
        assert np.all(np.isfinite(a)), ("device not finite", tag, what)
        if np.linalg.norm(r) == 0:
            assert np.linalg.norm(a) == 0, (tag, what)
        else:
            assert rel_l2(a, r) < tol, (tag, what, rel_l2(a, r))

    # (Y and U laws: the reference's finite-difference steps inside dD/dH, dD/d|grad S| amplify rounding to 1e-8 / 1e-6)
    tolH = {O.LAW_NN_Y: 1e-7, O.LAW_NN_U: 1e-5}.get(kind, 1e-10)
    close(dH, O.sia2d_rhs(H, B, dx, dy, ph, law), 1e-11, "dhdt")
    assert np.all(dH[0, :] == 0) and np.all(dH[-1, :] == 0) and np.all(dH[:, 0] == 0) and np.all(dH[:, -1] == 0)
    close(vH, O.vjp_H(lam, H, B, dx, dy, ph, law), tolH, "vjp_H")
    close(vT, O.vjp_theta(lam, H, B, dx, dy, ph, law), 1e-9, "vjp_theta")
    close(vHc, O.vjp_H_continuous(lam, H, B, dx, dy, ph, law), tolH, "vjp_H continuous")
    close(vTc, O.vjp_theta_continuous(lam, H, B, dx, dy, ph, law), 1e-9, "vjp_theta continuous")
    vx, vy = O.surface_V(H, B, dx, dy, ph, law)
    close(Vx[:-1, :-1], vx, 1e-11, "surface_V x")
    close(Vy[:-1, :-1], vy, 1e-11, "surface_V y")
    assert np.all(Vx[-1, :] == 0) and np.all(Vx[:, -1] == 0) and np.all(Vy[-1, :] == 0) and np.all(Vy[:, -1] == 0)
    close(sH, O.vjp_surface_V_H(w1, w2, H, B, dx, dy, ph, law), tolH, "surface_V vjp_H")
    close(sT, O.vjp_surface_V_theta(w1, w2, H, B, dx, dy, ph, law), 1e-9, "surface_V vjp_theta")
    Hr, MBr = O.mb_apply(mb, np.maximum(H, 0.0) if False else H, B)
    assert np.allclose(Hn, Hr, rtol=1e-13, atol=1e-13) and np.allclose(MB, MBr, rtol=1e-12, atol=1e-14), (tag, "mb_apply")
    assert np.allclose(mV, O.vjp_mb(mb, lam, H, B), rtol=1e-13, atol=0), (tag, "mb_vjp")
    if kind == O.LAW_CONST_A:
        assert lawv.shape == (1,) and lawv[0] == A, (tag, "eval_law")
    else:
        close(lawv.ravel(), np.atleast_1d(O.law_value(law, ph, O.avg(np.maximum(H, 0.0)), _gradS(H, B, dx, dy))).ravel(), 1e-12, "eval_law")


# (seeds 49 and 78 of the fixed slice cost the numpy CHECKER 57 s and 84 s -- reverse solves of several thousand steps on the CPU; the
#  device needs a fraction of a second.  They stay in the exploration runs, ODINN_FUZZ_SEEDS=a:b, and in the other three tests.)
@pytest.mark.parametrize("seed", [s_ for s_ in _seeds() if s_ not in (49, 78) or os.environ.get("ODINN_FUZZ_SEEDS")])
def test_random_batch_time_aggregated_terms_match_the_oracle(gpu, monkeypatch, seed):
    """The draw of the first test with LossH and a random subset of the time-aggregated terms of a MultiLoss: LossDhdt
    (TimeAggregatedLosses.jl:38-113), LossAvgV (:115-258, :xy / :abs), VelocityRegularization (Regularization.jl:64-79,
    192-245), with random weights and windows, every law, both adjoints."""
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB", "ODINN_FUZZ_AUDIT", "ODINN_FUZZ_BIG"):
            monkeypatch.delenv(key, raising=False)
    c = _draw(gpu, 700000 + seed)
    rng = np.random.default_rng(31000 + seed)
    G, ph, kind, mode = c["G"], c["ph"], c["kind"], c["mode"]
    c["vjp"] = "discrete"
    if c["ragged"]:  # the windows of the aggregated terms sit on the common table
        c["ragged"] = False
        for g in range(G):
            c["own"][g] = list(c["common"])
            c["refs"][g] = [c["refs"][g][min(j, len(c["refs"][g]) - 1)] for j in range(len(c["common"]))]
    ts, k = c["common"], len(c["common"])
    terms = [t for t in ("dhdt", "avgv", "vreg") if rng.random() < 0.55] or ["dhdt"]
    w = dict(dhdt=float(rng.uniform(0.5, 5.0)), avgv=float(rng.uniform(0.5, 5.0)), vreg=float(rng.uniform(5.0, 80.0)))
    comp = "abs" if rng.random() < 0.5 else "xy"
    dist = int(rng.integers(0, 4))
    extra, avg, tVs = [], [], []
    for g in range(G):
        e = {}
        if "dhdt" in terms:
            i0 = int(rng.integers(0, k - 1)); i1 = int(rng.integers(i0 + 1, k))
            e.update(dhdt=(ts[i0], ts[i1], float(rng.uniform(-5.0, 1.0))), dhdt_weight=w["dhdt"])
        if "avgv" in terms:
            i1 = int(rng.integers(0, k - 1)); i2 = int(rng.integers(i1 + 1, k))
            Vx0, Vy0, _ = O.V_from_H(c["gls"][g].H0, c["gls"][g].B, c["dxs"][g], c["dys"][g], c["phs"][g], c["laws"][g])
            Vx, Vy = np.asfortranarray(1.15 * Vx0), np.asfortranarray(0.9 * Vy0)
            a = O.AvgVData(ts[i1], ts[i2], np.asfortranarray(np.sqrt(Vx ** 2 + Vy ** 2)), Vx, Vy, comp, c["step"])
            e.update(avgv=a, avgv_weight=w["avgv"])
            avg.append(a)
        if "vreg" in terms:
            tV = [ts[j] for j in sorted(rng.choice(k, size=int(rng.integers(2, k + 1)), replace=False))]  # (the weights are date differences)
            e.update(vreg_times=tV, vreg_distance=dist, vreg_weight=w["vreg"])
            tVs.append(tV)
        extra.append(e)
    c["cfg_extra"] = extra
    tag = {q: c[q] for q in ("G", "kind", "mode", "log_eps", "mbt", "interp", "shapes", "dxs", "dys", "sched")}
    tag.update(terms=terms, comp=comp, dist=dist, sliding=ph.C != 0.0)
    nq = 8
    Lo, go = _oracle_gradient_or_skip("aggregated", seed, tag, c, nq)
    ill = mode != "discrete_fixed" and _ill_conditioned(c, nq, go, 2e-5)
    ILL = "the checker's own gradient moves by more than the tolerance under a 1e-12 perturbation of the parameters"
    if ill and not _audit():
        pytest.skip(ILL)
    b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
    try:
        for g in range(G):
            b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
            b.set_reference(g, c["own"][g], c["refs"][g], 3)
            if c["mbs"][g] is not None:
                m = c["mbs"][g]
                b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
            if "dhdt" in terms:
                b.set_dhdt_reference(g, *extra[g]["dhdt"])
            if "avgv" in terms:
                a = avg[g]
                b.set_avgv_reference(g, a.t1, a.t2, a.Vabs, a.Vx, a.Vy)
            if "vreg" in terms:
                z = [np.zeros(c["shapes"][g])] * len(tVs[g])
                b.set_velocity_reference(g, tVs[g], z, z, z)  # only the dates matter
        if kind != O.LAW_CONST_A:
            b.set_law(kind, c["gm"], c["th"])
            if kind == O.LAW_NN_A_GRIDDED:
                for g in range(G):
                    b.set_T_field(g, c["laws"][g].T)
            if kind == O.LAW_NN_Y:
                b.set_grad_interpolation(gpu._lib.GRAD_INTERP_LINEAR if c["interp"][0] == "linear" else gpu._lib.GRAD_INTERP_NONE,
                                         c["interp"][1])
        if "dhdt" in terms:
            b.set_dhdt_loss(w["dhdt"])
        if "avgv" in terms:
            b.set_avgv_loss(w["avgv"], c["step"], comp)
        if "vreg" in terms:
            b.set_velocity_regularization(w["vreg"], dist)
        if c["log_eps"] is not None:
            b.set_thickness_loss_function(c["log_eps"])
        if c["sched"]:
            b.set_schedule(**c["sched"])
        if mode == "continuous":
            Lg, gg = _device_or_skip("aggregated", seed, tag, lambda: b.loss_grad_continuous(ts, theta=c["th"], mb_times=c["mbt"], reltol=1e-8, n_quadrature=nq))
        elif mode == "discrete_fixed":
            Lg, gg = b.loss_grad(ts, theta=c["th"], mb_times=c["mbt"], fixed_dt=c["dts"])
        else:
            Lg, gg = b.loss_grad(ts, theta=c["th"], mb_times=c["mbt"], reltol=1e-8)
        comparable = _margins_agree(b, c, mode)
    finally:
        b.close()
    if ill:  # (audit runs only: the draw went through the device although the checker itself is ill-conditioned on it)
        _skip("aggregated", seed, ILL, tag, Lg, Lo, gg, go)
    if not comparable:
        _skip("aggregated", seed, "subnormal margin cells differ between the device and the checker: the gradient is discontinuous there", tag, Lg, Lo, gg, go)
    # (the Laplacian of VelocityRegularization and the small differences of LossAvgV amplify rounding in the LOSS to 1e-9)
    ltol, gtol = (1e-8, 1e-7 if kind in (O.LAW_NN_Y, O.LAW_NN_U) else 1e-8) if mode == "discrete_fixed" else (1e-6, 2e-5)
    assert abs(Lg - Lo) <= ltol * max(abs(Lo), 1e-300), (tag, Lg, Lo)
    if np.linalg.norm(go) > 0:
        if not rel_l2(gg, go) < gtol and rel_l2(gg, go) < 5e-3 and _subnormal_margin(c, mode):
            _skip("aggregated", seed, "gradient differs at the level of the reference's own discontinuity at subnormal margin cells", tag, Lg, Lo, gg, go)
        assert rel_l2(gg, go) < gtol, (tag, rel_l2(gg, go))
    else:
        assert np.linalg.norm(gg) == 0, tag


@pytest.mark.parametrize("seed", _seeds())
def test_random_batch_forward_solve_matches_the_oracle(gpu, monkeypatch, seed):
    """The forward solve alone (_batch_iceflow_UDE, inversion_utils.jl:472-572) on the draws of the first test: adaptive
    RDPK3Sp35 under the per-stage and the fused kernel schedules with and without the ice-free shortcut (snapshots 1e-6, step
    counts within 2), the fixed-step sequence (1e-11), and the CFL-limited Euler scheme against its restatement (same step
    count, 1e-10); mass-balance times on stops and between them, per-glacier stop tables."""
    for key in list(os.environ):
        if key.startswith("ODINN_") and key not in ("ODINN_FUZZ_SEEDS", "ODINN_LIB", "ODINN_FUZZ_AUDIT", "ODINN_FUZZ_BIG"):
            monkeypatch.delenv(key, raising=False)
    c = _draw(gpu, 400000 + seed)
    rng = np.random.default_rng(52000 + seed)
    G, ph, kind = c["G"], c["ph"], c["kind"]
    how = ["adaptive", "fixed", "euler"][int(rng.integers(0, 3))]
    scheme = int(rng.choice([0, 1, 2])) if how != "euler" else gpu._lib.SCHEME_EULER_CFL
    dense = int(rng.integers(0, 2))
    cfl = float(rng.choice([0.1, 0.25, 0.5]))
    mbt = list(c["mbt"])
    if c["mbs"][0] is not None and rng.random() < 0.5:  # a mass-balance time that is nobody's stop
        mbt = sorted(set(mbt + [c["common"][0] + 0.61 * (c["common"][-1] - c["common"][0])]))
    tag = {q: c[q] for q in ("G", "kind", "ragged", "shapes", "dxs", "dys", "sched")}
    tag.update(how=how, scheme=scheme, dense=dense, mbt=mbt, sliding=ph.C != 0.0)
    b = gpu.GlacierBatch(c["shapes"], c["dxs"], c["dys"], phys=[gpu.PhysicalParameters(**q.__dict__) for q in c["phs"]], A=c["As"], T=c["Ts"])
    try:
        for g in range(G):
            b.set_fields(g, c["gls"][g].H0, c["gls"][g].B)
            if c["mbs"][g] is not None:
                m = c["mbs"][g]
                b.set_mass_balance(g, m.mb0, m.dmb_dS, m.S_ref, m.mb_max)
            if c["ragged"]:
                b.set_glacier_stops(g, c["own"][g])
        if kind != O.LAW_CONST_A:
            b.set_law(kind, c["gm"], c["th"])
            if kind == O.LAW_NN_A_GRIDDED:
                for g in range(G):
                    b.set_T_field(g, c["laws"][g].T)
        if c["sched"]:
            b.set_schedule(**c["sched"])
        union = sorted(set(t for ts in c["own"] for t in ts))
        if how == "euler":
            st = b.solve(union, mb_times=mbt, scheme=scheme, cfl=cfl)
        elif how == "fixed":
            st = b.solve(union, mb_times=mbt, fixed_dt=c["dts"], scheme=scheme, dense=dense)
        else:
            st = b.solve(union, mb_times=mbt, reltol=1e-8, scheme=scheme, dense=dense)
        got = [[b.snapshot(g, j) for j in range(len(c["own"][g]))] for g in range(G)]
    finally:
        b.close()
    for g in range(G):
        gl, law, mb = c["gls"][g], c["laws"][g], c["mbs"][g]
        mt = mbt if mb is not None else ()
        if how == "euler":
            cb = (lambda u, t, mb=mb, B=gl.B: O.mb_apply(mb, u, B)[0]) if mb is not None else None
            stops = sorted(set(c["own"][g]) | set(mt))
            snaps_all, n = O.solve_euler_cfl(gl, law, stops, cfl=cfl, callback=cb, callback_times=mt)
            snaps = [snaps_all[stops.index(t)] for t in c["own"][g]]
            assert abs(st[g].naccept - n) <= 1 and st[g].nreject == 0, (tag, g, st[g], n)
            tol = 1e-10 if st[g].naccept == n else 1e-3
        else:
            cfg = O.SimConfig(tstops=c["own"][g], reltol=1e-8, mb=mb, mb_times=mt, fixed_dt=c["dts"] if how == "fixed" else None)
            snaps, so, _ = O.forward(gl, law, cfg)
            close = lambda s_: abs(st[g].naccept - s_.naccept) <= 2 and abs(st[g].nreject - s_.nreject) <= 2
            unstable = False
            if how == "adaptive" and not close(so):
                # A step that ends within ~1e-5 of its length before a stop is followed by a sliver step onto the stop and by a dozen
                # steps in which the controller grows the step size back (dt = h * factor, factor <= 1 + pi / 2); one that reaches the
                # stop is not.  Where the error estimate sits at the round-off floor (smooth draws: err ~ 1e-14 m) the two sides'
                # step sizes differ by 1e-5 relative and can fall on either side of that edge (seed 19220: 20 steps here, 7 in the
                # oracle -- whose own count is 7, 16, 15, 11 under 1e-4 ... 1e-2 relative changes of reltol -- snapshots equal to
                # 3e-13).  The step count is then not a property of the algorithm: not compared where the ORACLE's own count moves by
                # more than the window under such a change; the snapshots still are.
                counts = [O.forward(gl, law, O.SimConfig(tstops=c["own"][g], reltol=1e-8 * rs, mb=mb, mb_times=mt))[1].naccept
                          for rs in (1.0 - 1e-4, 1.0 + 1e-4, 1.0 - 1e-3, 1.0 + 1e-3)] + [so.naccept]
                # ... or where one of its steps all but reached / all but missed a stop (SolveStats.min_stop_gap; seed 24379: 15 vs 11)
                unstable = max(counts) - min(counts) > 2 or so.min_stop_gap < 1e-9
            assert unstable or close(so), (tag, g, st[g], so)
            tol = 1e-11 if how == "fixed" else 1e-6
        assert abs(st[g].t_final - c["own"][g][-1]) < 1e-12, (tag, g)
        for j in range(len(snaps)):
            if np.linalg.norm(snaps[j]) > 0:
                assert rel_l2(got[g][j], snaps[j]) < tol, (tag, g, j, rel_l2(got[g][j], snaps[j]))
