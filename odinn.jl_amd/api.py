"""Host-side mirror of the reference's operator interface for the SIA2D(+NN) path.

The reference is Julia and its toolchain is absent from the build image, so the host
side above the C ABI is Python.  Names, argument meaning and error behaviour follow the
reference (Julia's ``f!`` is spelled ``f_b``; non-ASCII ``∂`` is spelled ``d``):

    Parameters / PhysicalParameters / SimulationParameters / SolverParameters /
    Hyperparameters / UDEparameters           src/parameters/*.jl, Sleipnir (OOT)
    NeuralNetwork                             src/models/trainable_components/NeuralNetwork.jl:18-74
    LawA / LawY / LawU / ConstantA            src/laws/Laws.jl:97-386, Huginn (OOT)
    SIA2Dmodel, Model                         Huginn (OOT), src/models/trainable_components/Model.jl:61-127
    Prediction, Inversion (= FunctionalInversion, the legacy name)
                                              Huginn (OOT), src/simulations/inversions/Inversion.jl:16-62
    run_b  (run!)                             src/simulations/inversions/inversion_utils.jl:21-88
    SIA2D_grad_b (SIA2D_grad!)                src/inverse/SIA2D/gradient.jl:6-31
    VJP_lambda_dSIAdH / VJP_lambda_dSIAdtheta src/inverse/SIA2D/VJPs.jl:2-5,30-33
    SIA2D_b (Huginn.SIA2D!)                   called at inversion_utils.jl:691-699

Everything numerical runs in libodinn_hip.so on the GPU; this module only orchestrates
(glacier sharding, the optimiser loop, the RCCL all-reduce of [loss, dtheta]).
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .batch import GlacierBatch, MLPSpec, PhysicalParameters

__all__ = [
    "Parameters", "SimulationParameters", "SolverParameters", "Hyperparameters", "UDEparameters",
    "Glacier2D", "ThicknessData", "NeuralNetwork", "LawA", "LawY", "LawU", "ConstantA", "SIA2Dmodel", "Model",
    "TrainableComponents", "InversionBinder",
    "GlacierWideInv", "GriddedInv", "LinearMB", "FieldMB", "Prediction", "Inversion", "FunctionalInversion", "DiscreteAdjoint", "ContinuousAdjoint", "DummyAdjoint", "DiscreteVJP", "ContinuousVJP", "MultiLoss", "TikhonovRegularization",
    "InitialThicknessRegularization", "RheologyRegularization", "InitialCondition", "evaluate_H0", "evaluate_dH0",
    "sigma_zang", "dsigma_zang", "TrainingResult", "save_inversion_file", "load_inversion_file", "ScalarLogger", "TBLogger", "read_event_file",
    "callback_diagnosis",
    "LossH", "LossV", "LossHV", "VelocityData", "V_from_H", "L2Sum", "LogSum", "Adam", "LBFGS", "Results", "TrainingStats", "run_b", "SIA2D_grad_b", "loss_iceflow_transient", "SIA2D_b",
    "VJP_lambda_dSIAdH", "VJP_lambda_dSIAdtheta", "define_callback_steps", "build_default_NN",
    "shard_glaciers", "init_distributed", "allreduce_loss_grad", "load_gridded_glacier", "attach_rccl_comm",
    "SIA2D_A_target", "SIA2D_D_hybrid_target", "SIA2D_D_target", "LossDhdt", "DhdtData", "LossAvgV", "VelocityRegularization",
]


# ----------------------------------------------------------------------------------------
# parameters (field names as in the reference; only what the path reads)
# ----------------------------------------------------------------------------------------
@dataclass
class SimulationParameters:
    tspan: Tuple[float, float] = (2010.0, 2015.0)
    use_MB: bool = False
    step_MB: float = 1.0 / 12.0
    multiprocessing: bool = False  # reference: Distributed workers; here: one rank per GPU
    workers: int = 1
    test_mode: bool = False  # light NN (NeuralNetwork.jl:43)
    f_surface_velocity_factor: float = 1.0  # target :D: Velocityꜛ = U / f (target_D_pure.jl:206-255; 0.8 in test_grad_loss.jl:120)


@dataclass
class SolverParameters:
    """Huginn.SolverParameters (OOT).  solver is fixed to RDPK3Sp35 (test_grad_loss.jl:143)."""

    solver: str = "RDPK3Sp35"
    reltol: float = 1e-8
    abstol: float = 1e-6
    step: float = 1.0 / 12.0
    tstops: Sequence[float] = ()
    dtmax: float = 0.0
    maxiters: int = 10 ** 6


@dataclass
class DiscreteVJP:
    """src/inverse/VJPTypes.jl:29-37"""


@dataclass
class ContinuousVJP:
    """src/inverse/VJPTypes.jl:39-50: the continuous-form stencils of adjoint.jl:442-662 (target :A)."""


@dataclass
class DiscreteAdjoint:
    """src/inverse/AdjointTypes.jl:85-91"""

    VJP_method: object = field(default_factory=DiscreteVJP)  # DiscreteVJP | ContinuousVJP
    MB_VJP: DiscreteVJP = field(default_factory=DiscreteVJP)


@dataclass
class DummyAdjoint:
    """src/inverse/AdjointTypes.jl:98-107: exercises the training workflow without a sensitivity analysis -- the loss is the
    real one, the "gradient" is grad_function(θ) or max|θ| · rand(size(θ)) (gradient.jl:540-545; the reference's
    grad_free_test, runtests.jl:72-73)."""

    grad_function: Optional[Callable] = None
    VJP_method: object = field(default_factory=DiscreteVJP)
    MB_VJP: DiscreteVJP = field(default_factory=DiscreteVJP)
    seed: int = 0  # of the random dummy gradient (reproducible here; `rand` in the reference)


@dataclass
class ContinuousAdjoint:
    """src/inverse/AdjointTypes.jl:53-67 -- the reference's default `grad` (UDEparameters.jl:63).
    VJP_method: DiscreteVJP() (default) or ContinuousVJP() (adjoint.jl:442-662, all three targets); interpolation = :Linear
    and the RDPK3Sp35 reverse solver are what the reference defines / what the device implements."""

    VJP_method: object = field(default_factory=DiscreteVJP)  # DiscreteVJP | ContinuousVJP
    solver: str = "RDPK3Sp35"
    reltol: float = 1e-8
    abstol: float = 1e-8
    dtmax: float = 1.0 / 12.0
    interpolation: str = "Linear"
    n_quadrature: int = 200
    MB_VJP: DiscreteVJP = field(default_factory=DiscreteVJP)

    def __post_init__(self):
        if self.interpolation != "Linear":
            raise ValueError("Interpolation method for continuous adjoint not defined.")  # gradient.jl:302
        if self.solver != "RDPK3Sp35":
            raise ValueError("only RDPK3Sp35 is provided for the reverse solve")


@dataclass
class L2Sum:
    """src/losses/Losses.jl: L2Sum(distance)"""

    distance: int = 3


@dataclass
class LogSum:
    """src/losses/Losses.jl:34-49,207-229: log²((a + ϵ) / (b + ϵ)) / normalization (Morlighem et al. 2010); the simple loss
    of LossV(component = :abs) -- the combination the reference tests (runtests.jl:165-167) -- or of LossH."""

    distance: int = 3
    eps: float = 0.1  # ϵ


@dataclass
class LossH:
    """src/losses/Losses.jl:250-291"""

    loss: object = field(default_factory=L2Sum)  # L2Sum | LogSum


@dataclass
class LossV:
    """src/losses/Losses.jl:66-81,293-390 (targets :A and :D)"""

    loss: object = field(default_factory=L2Sum)  # L2Sum | LogSum
    component: str = "xy"  # :xy | :abs
    scale_loss: bool = True

    def __post_init__(self):
        if isinstance(self.loss, LogSum) and self.component != "abs":
            raise ValueError("LogSum needs non-negative fields (Losses.jl:214): use LossV(loss = LogSum(), component = :abs)")


@dataclass
class LossHV:
    """src/losses/Losses.jl:86-112,395-440"""

    hLoss: LossH = field(default_factory=LossH)
    vLoss: LossV = field(default_factory=LossV)
    scaling: float = 1.0


@dataclass
class TikhonovRegularization:
    """src/losses/Regularization.jl:24-45: sum over a mask of (nabla^2 a)^2 with the staggered Laplacian
    (:330-352) and its hand-written VJP (:372-382); evaluated on the device (odinn_tikhonov)."""

    operator: str = "laplacian"
    distance: int = 3

    def __post_init__(self):
        if self.operator != "laplacian":
            raise ValueError(f"Operator named {self.operator} not implemented inside Tikhonov regularization")


@dataclass
class InitialThicknessRegularization:
    """src/losses/Regularization.jl:47-61,128-190: Tikhonov term on H0 = evaluate_H0(theta.IC) at t0."""

    reg: TikhonovRegularization = field(default_factory=TikhonovRegularization)
    t0: float = 1994.0


@dataclass
class RheologyRegularization:
    """src/losses/Regularization.jl:79-89,252-310: Tikhonov term on the gridded A of a classical inversion."""

    reg: TikhonovRegularization = field(default_factory=TikhonovRegularization)


@dataclass
class VelocityRegularization:
    """src/losses/Regularization.jl:64-79,192-245: Tikhonov penalty on the Laplacian of the predicted surface speed inside
    the glacier (distance to the margin), at the velocity-data times with their Δt weights -- the regulariser of the
    reference's documented example MultiLoss((LossH(), VelocityRegularization()), ...).  Evaluated on the device."""

    reg: TikhonovRegularization = field(default_factory=TikhonovRegularization)
    components: str = "abs"
    distance: int = 3

    def __post_init__(self):
        if self.components != "abs":
            raise ValueError(f"Regularization {self} not implemented.")  # Regularization.jl:214


@dataclass
class LossDhdt:
    """src/losses/TimeAggregatedLosses.jl:38-113: (mean_{H(t0) > 1e-2}(H(t1) - H(t0)) / (t1 - t0) - dhdt_ref)^2 with
    glacier.dhdtData = DhdtData((t0, t1), dhdt_ref) -- a time-aggregated loss, evaluated on the device together with its
    cotangent fields at t0 and t1."""


@dataclass
class LossAvgV:
    """src/losses/TimeAggregatedLosses.jl:115-258: L2Sum between the time-weighted average of the predicted surface
    velocity over velocityData's single sample [date1, date2] (time grid date1:step:date2) and that sample (component
    :xy | :abs) -- a time-aggregated loss, evaluated on the device; its cotangent is pulled back through surface_V at
    every point of the time grid."""

    loss: L2Sum = field(default_factory=L2Sum)
    component: str = "xy"
    step: float = 1.0 / 12.0


_AGGREGATED = (LossDhdt, LossAvgV, VelocityRegularization)  # evaluated on the device next to the data loss


@dataclass
class MultiLoss:
    """src/losses/MultiLoss.jl:22-35: sum_k lambdas[k] * losses[k].  One data term (LossH | LossV |
    LossHV, evaluated by the device adjoint) plus any number of regularisers."""

    losses: Tuple = field(default_factory=lambda: (LossH(),))
    lambdas: Tuple = (1.0,)  # λs

    def __post_init__(self):
        if len(self.losses) != len(self.lambdas):
            raise ValueError("You need to provide an hyperparameter for each loss term defined.")  # MultiLoss.jl:29


class _NoDataTerm(LossH):
    """Marker: the loss holds time-aggregated terms / regularisers only (MultiLoss((LossDhdt(),), (1,)) has no thickness
    term even when the glacier carries thicknessData): no reference is uploaded, so the device adds no LossH."""


def _split_loss(lf):
    """(data loss, its weight, [(regulariser, weight)]) of a loss specification."""
    if isinstance(lf, _AGGREGATED + (InitialThicknessRegularization, RheologyRegularization)):  # a bare regulariser (runtests.jl:221-223: loss = RheologyRegularization())
        return _NoDataTerm(), 1.0, [(lf, 1.0)]
    if not isinstance(lf, MultiLoss):
        return lf, 1.0, []
    data = [(l, w) for l, w in zip(lf.losses, lf.lambdas) if isinstance(l, (LossH, LossV, LossHV))]
    regs = [(l, w) for l, w in zip(lf.losses, lf.lambdas) if not isinstance(l, (LossH, LossV, LossHV))]
    if len(data) == 0 and regs:
        data = [(_NoDataTerm(), 1.0)]  # time-aggregated losses / regularisers alone: no data term
    if len(data) != 1:
        raise ValueError("MultiLoss needs exactly one data term (LossH, LossV or LossHV)")
    for r, _ in regs:
        if not isinstance(r, (InitialThicknessRegularization, RheologyRegularization, LossDhdt, LossAvgV, VelocityRegularization)):
            raise TypeError(f"loss term {type(r).__name__} is not provided")
    return data[0][0], float(data[0][1]), regs


@dataclass
class Adam:
    """Optimisers.Adam(eta, (beta1, beta2), eps)"""

    eta: float = 1e-3
    beta: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8


@dataclass
class LBFGS:
    """Optim.LBFGS stand-in (scipy L-BFGS-B drives the same loss/grad callbacks)."""

    m: int = 10


@dataclass
class Hyperparameters:
    optimizer: object = field(default_factory=Adam)
    epochs: int = 50
    batch_size: int = 0  # reference: whole simulation in one batch (ML_utils.jl:190-200)


@dataclass
class UDEparameters:
    """src/parameters/UDEparameters.jl:60-80.  grad: DiscreteAdjoint() or ContinuousAdjoint() (the
    reference's default), both with the hand-written DiscreteVJP stencils."""

    grad: object = field(default_factory=lambda: ContinuousAdjoint())  # the reference's default (UDEparameters.jl:63)
    # LossH | LossV | LossHV | MultiLoss; the reference's default is MultiLoss(losses = (LossH(),), λs = (1.0,))
    empirical_loss_function: object = field(default_factory=lambda: MultiLoss())
    target: str = "A"  # :A | :D_hybrid | :D
    optimization_method: str = "AD+AD"
    initial_condition_filter: str = "identity"  # :identity | :softplus | :Zang1980 (UDEparameters.jl:67)


@dataclass
class Parameters:
    """src/parameters/UDEparameters.jl:109-128"""

    physical: PhysicalParameters = field(default_factory=PhysicalParameters)
    simulation: SimulationParameters = field(default_factory=SimulationParameters)
    solver: SolverParameters = field(default_factory=SolverParameters)
    hyper: Hyperparameters = field(default_factory=Hyperparameters)
    UDE: UDEparameters = field(default_factory=UDEparameters)


# ----------------------------------------------------------------------------------------
# glaciers, data
# ----------------------------------------------------------------------------------------
@dataclass
class ThicknessData:
    t: Sequence[float]
    H: Sequence[np.ndarray]


@dataclass
class DhdtData:
    """glacier.dhdtData of LossDhdt: mean surface-elevation change rate `dhdt` between t = (t0, t1)
    (src/losses/TimeAggregatedLosses.jl:66-68)."""

    t: Tuple[float, float]
    dhdt: float


@dataclass
class VelocityData:
    """glacier.velocityData: absolute value and components at dates t (nx*ny fields).  date1 / date2 (decimal years): the
    acquisition window of each sample -- LossAvgV reads them (exactly one sample), LossV compares at t."""

    t: Sequence[float]
    vabs: Sequence[np.ndarray]
    vx: Sequence[np.ndarray]
    vy: Sequence[np.ndarray]
    date1: Optional[Sequence[float]] = None
    date2: Optional[Sequence[float]] = None


def _avgv_times(g, loss) -> List[float]:
    """discretePostIntegralLossSteps(::LossAvgV) (TimeAggregatedLosses.jl:355-363): date1:step:date2 without its last point."""
    v = g.velocityData
    if v is None or v.date1 is None or v.date2 is None or len(v.date1) != 1 or len(v.date2) != 1:
        raise ValueError("With LossAvgV the velocity data should contain exactly one sample.")  # :356
    t1, t2 = float(v.date1[0]), float(v.date2[0])
    n = int(round((t2 - t1) / loss.step))  # length of a Julia float range (see odinn_hip.hip, agg_tables)
    if t1 + n * loss.step > t2 + 4.0 * 2.220446049250313e-16 * max(abs(t1), abs(t2)):
        n -= 1
    return [t1 + i * loss.step for i in range(n)]


@dataclass
class Glacier2D:
    """POD stand-in for Sleipnir.Glacier2D (kwargs as test/test_grad_loss.jl:595-597)."""

    rgi_id: str
    H0: np.ndarray  # H₀
    B: np.ndarray
    dx: float  # Δx
    dy: float  # Δy
    A: float = 2.21e-18
    C: float = 0.0
    n: float = 3.0
    T: float = -5.0  # long-term air temperature (iAvgScalarTemp)
    thicknessData: Optional[ThicknessData] = None
    velocityData: Optional[VelocityData] = None
    dhdtData: Optional[DhdtData] = None
    mask: Optional[np.ndarray] = None  # True OUTSIDE the glacier (Sleipnir builds it from the outline); default H0 <= 0

    def __post_init__(self):
        if self.mask is None:
            self.mask = np.asarray(self.H0) <= 0.0

    @property
    def nx(self):
        return self.H0.shape[0]

    @property
    def ny(self):
        return self.H0.shape[1]


def load_gridded_glacier(path, rgi_id=None, thickness_vars=("consensus_ice_thickness", "distributed_thickness", "thickness"),
                         topo_vars=("topo_smoothed", "topo"), mask_var="glacier_mask", **glacier_kwargs):
    """Real-glacier ingestion: a gridded directory file in the layout OGGM writes and Sleipnir reads
    (`gridded_data.nc`: 1-D coordinates `x`, `y` in metres, 2-D variables on (y, x): surface elevation `topo` /
    `topo_smoothed`, `glacier_mask`, an ice-thickness estimate) -> `Glacier2D`.

    Own definitions (Sleipnir's glacier initialisation is out of tree): H0 = thickness with NaN -> 0 and 0 outside
    `glacier_mask`; B = surface - H0; arrays are returned as `[i, j]` = `[x, y]` with x contiguous (this
    package's layout) and both coordinates increasing; dx, dy = coordinate spacings.  Reads NetCDF-3 (classic /
    64-bit offset) files through scipy -- the image has no HDF5 stack, so a NetCDF-4 `gridded_data.nc` must be
    converted first (`nccopy -k classic`) -- and `.npz` files holding the same variable names."""
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            var = {k: np.array(z[k]) for k in z.files}
    else:
        from scipy.io import netcdf_file

        with netcdf_file(str(path), "r", mmap=False) as nc:
            var = {k: np.array(v[:], dtype=float) * getattr(v, "scale_factor", 1.0) + getattr(v, "add_offset", 0.0)
                   for k, v in nc.variables.items()}
    x, y = np.asarray(var["x"], float), np.asarray(var["y"], float)

    def pick(names):
        for n_ in names:
            if n_ in var:
                return np.asarray(var[n_], float)
        raise KeyError(f"none of {names} in {path}")

    thick, topo = pick(thickness_vars), pick(topo_vars)
    if thick.shape != (y.size, x.size) or topo.shape != thick.shape:
        raise ValueError(f"expected (y, x) = ({y.size}, {x.size}) fields, got {thick.shape} / {topo.shape}")
    H = np.nan_to_num(thick, nan=0.0)
    H = np.maximum(H, 0.0)
    if mask_var in var:
        H = np.where(np.asarray(var[mask_var]) > 0, H, 0.0)
    S = np.nan_to_num(topo, nan=float(np.nanmin(topo)))

    def orient(a):  # (y, x) -> [x, y], both increasing
        a = a[::-1, :] if y.size > 1 and y[1] < y[0] else a
        a = a[:, ::-1] if x.size > 1 and x[1] < x[0] else a
        return np.asfortranarray(a.T)

    H0, S0 = orient(H), orient(S)
    dx = float(abs(x[1] - x[0])) if x.size > 1 else 1.0
    dy = float(abs(y[1] - y[0])) if y.size > 1 else dx
    return Glacier2D(rgi_id or os.path.splitext(os.path.basename(str(path)))[0], H0, np.asfortranarray(S0 - H0), dx, dy,
                     **glacier_kwargs)


def define_callback_steps(tspan, step):
    """Huginn.define_callback_steps: tspan[0]:step:tspan[1] (end included)."""
    n = int(round((tspan[1] - tspan[0]) / step))
    ts = [tspan[0] + k * step for k in range(n + 1)]
    ts[-1] = tspan[1]
    return ts


# ----------------------------------------------------------------------------------------
# regressors and laws
# ----------------------------------------------------------------------------------------
def build_default_NN(n_input=1, lightNN=False):
    """ML_utils.jl:23-39"""
    if lightNN:
        return [n_input, 3, 1], [L.ACT_SOFTPLUS, L.ACT_SIGMOID]
    return [n_input, 3, 10, 3, 1], [L.ACT_SOFTPLUS, L.ACT_SOFTPLUS, L.ACT_SOFTPLUS, L.ACT_SIGMOID]


class NeuralNetwork:
    """Feed-forward regressor (NeuralNetwork.jl:18-74).  ``architecture`` = (widths, acts);
    theta flattened [vec(W) column-major, b] per layer.  Glorot-uniform init, numpy
    ``default_rng(seed)`` (the reference seeds MersenneTwister(666), ML_utils.jl:79)."""

    def __init__(self, params: Parameters, architecture=None, theta=None, seed=666):
        if architecture is None:
            n_in = 1 if params.UDE.target == "A" else 2
            architecture = build_default_NN(n_in, lightNN=params.simulation.test_mode)
        self.widths, self.acts = list(architecture[0]), list(architecture[1])
        if theta is None:
            rng = np.random.default_rng(seed)
            parts = []
            for l in range(len(self.acts)):
                nin, nout = self.widths[l], self.widths[l + 1]
                lim = math.sqrt(6.0 / (nin + nout))
                parts += [rng.uniform(-lim, lim, nin * nout), np.zeros(nout)]
            theta = np.concatenate(parts)
        self.theta = np.asarray(theta, dtype=np.float64).copy()

    @property
    def n_params(self):
        return self.theta.size


class GlacierWideInv:
    """One scalar parameter per glacier (PerGlacierModel, GlacierWideInv.jl): theta_g =
    atanh((A_g - minA) * 2/(maxA - minA) - 1), initialised from glacier.A."""

    def __init__(self, params: Parameters, glaciers: Sequence["Glacier2D"], var: str = "A"):
        lo, hi = params.physical.minA, params.physical.maxA
        self.sizes = [1] * len(glaciers)
        self.theta = np.array([math.atanh((getattr(g, var) - lo) * 2.0 / (hi - lo) - 1.0) for g in glaciers])

    @property
    def n_params(self):
        return self.theta.size


class GriddedInv:
    """One parameter per dual-grid node per glacier (PerGlacierModel, GriddedInv.jl)."""

    def __init__(self, params: Parameters, glaciers: Sequence["Glacier2D"], var: str = "A"):
        lo, hi = params.physical.minA, params.physical.maxA
        self.sizes = [(g.nx - 1) * (g.ny - 1) for g in glaciers]
        self.theta = np.concatenate([np.full(n, math.atanh((getattr(g, var) - lo) * 2.0 / (hi - lo) - 1.0))
                                     for n, g in zip(self.sizes, glaciers)])

    @property
    def n_params(self):
        return self.theta.size


class InitialCondition:
    """Per-glacier initial-thickness matrices as trainable parameters (InitialCondition.jl:33-76);
    initialization :Farinotti2019 = glacier.H0."""

    def __init__(self, params: Parameters, glaciers: Sequence["Glacier2D"], initialization: str = "Farinotti2019"):
        if initialization != "Farinotti2019":
            raise ValueError("Strategy for initialization of ice thicknesses not found.")
        self.sizes = [g.nx * g.ny for g in glaciers]
        self.theta = np.concatenate([np.asarray(g.H0, dtype=np.float64).ravel(order="F") for g in glaciers])

    @property
    def n_params(self):
        return self.theta.size


def sigma_zang(x, beta=2.0):
    """σ_zang (InitialCondition_utils.jl:92-100)"""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x < -beta / 2, 0.0, np.where(x < beta / 2, (x + beta / 2) ** 2 / (2 * beta), x))


def dsigma_zang(x, beta=2.0):
    """∂σ_zang (InitialCondition_utils.jl:112-120)"""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x < -beta / 2, 0.0, np.where(x < beta / 2, x / beta + 0.5, 1.0))


def evaluate_H0(theta_ic, glacier: "Glacier2D", filter: str = "identity"):
    """evaluate_H₀ (InitialCondition_utils.jl:30-46)"""
    x = np.asarray(theta_ic, dtype=np.float64).reshape((glacier.nx, glacier.ny), order="F")
    if filter == "identity":
        H0 = x.copy()
    elif filter == "softplus":
        H0 = np.log(1.0 + np.exp(x))
    elif filter == "Zang1980":
        H0 = sigma_zang(x)
    else:
        raise ValueError(f"unknown initial_condition_filter {filter}")
    H0 = np.asfortranarray(H0)
    H0[glacier.mask] = 0.0
    return H0


def evaluate_dH0(theta_ic, glacier: "Glacier2D", filter: str = "identity"):
    """evaluate_∂H₀ (InitialCondition_utils.jl:73-89)"""
    x = np.asarray(theta_ic, dtype=np.float64).reshape((glacier.nx, glacier.ny), order="F")
    if filter == "identity":
        d = np.ones_like(x)
    elif filter == "softplus":
        d = 1.0 / (1.0 + np.exp(-x))
    elif filter == "Zang1980":
        d = dsigma_zang(x)
    else:
        raise ValueError(f"unknown initial_condition_filter {filter}")
    d = np.asfortranarray(d)
    d[glacier.mask] = 0.0
    return d


@dataclass
class _Law:
    kind: int
    nn: Optional[NeuralNetwork] = None
    mlp: Optional[MLPSpec] = None
    n_H: float = -1.0
    n_gradS: float = -1.0
    value: Optional[float] = None
    classical: Optional[str] = None  # "scalar" | "gridded": LawA(params; scalar) of Laws.jl:402-460
    bounds: Tuple[float, float] = (0.0, 1.0)


def ConstantA(A=2.21e-18):
    """Huginn.ConstantA"""
    return _Law(L.LAW_CONST_A, value=A)


def LawA(nn_model, params: Optional[Parameters] = None, scalar: bool = True):
    """LawA(nn_model, params; scalar): A = minA + (maxA-minA) * NN(T)  (Laws.jl:323-386), evaluated
    once per simulation (callback_freq = 0 with the manual adjoints, Laws.jl:339-347).
    LawA(params; scalar): the classical per-glacier law A = minA + (maxA-minA)(tanh(theta)+1)/2,
    glacier-wide or gridded (Laws.jl:402-460)."""
    if isinstance(nn_model, Parameters):
        ph = nn_model.physical
        return _Law(L.LAW_CONST_A, classical="scalar" if scalar else "gridded", bounds=(ph.minA, ph.maxA))
    ph = params.physical
    mlp = MLPSpec(nn_model.widths, nn_model.acts, None, L.POST_AFFINE, ph.minA, ph.maxA)
    return _Law(L.LAW_NN_A_SCALAR if scalar else L.LAW_NN_A_GRIDDED, nn_model, mlp)


def LawY(nn_model: NeuralNetwork, params: Parameters, max_NN=None, prescale_bounds=((-25.0, 0.0), (0.0, 500.0)),
         n_H=-1.0, n_gradS=-1.0):
    """Y = max_NN*exp((y-1)/y), y = NN(norm(T), norm(Hbar))  (Laws.jl:240-273)."""
    mx = params.physical.maxA if max_NN is None else max_NN
    mlp = MLPSpec(nn_model.widths, nn_model.acts, prescale_bounds, L.POST_EXPMAX, 0.0, mx)
    return _Law(L.LAW_NN_Y, nn_model, mlp, n_H, n_gradS)


def LawU(nn_model: NeuralNetwork, params: Parameters, max_NN=None, prescale_bounds=None):
    """U = post(NN(pre(Hbar, gradS)))  (Laws.jl:97-183); D = Hbar*U."""
    kind = L.POST_EXPMAX if max_NN is not None else L.POST_NONE
    mlp = MLPSpec(nn_model.widths, nn_model.acts, prescale_bounds, kind, 0.0, max_NN if max_NN is not None else 1.0)
    return _Law(L.LAW_NN_U, nn_model, mlp)


class SIA2Dmodel:
    """Huginn.SIA2Dmodel(params; A|Y|U = law)  (call sites test/SIA2D_adjoint.jl:72-88)."""

    def __init__(self, params: Parameters, A=None, Y=None, U=None):
        given = [l for l in (A, Y, U) if l is not None]
        if len(given) > 1:
            raise ValueError("provide at most one of A, Y, U")
        self.law = given[0] if given else ConstantA()
        self.U_is_provided = U is not None
        self.Y_is_provided = Y is not None


@dataclass
class LinearMB:
    """Synthetic stand-in for Muninn.TImodel1 (climate pipeline is out of tree): mass-balance
    rate min(grad*(S-ELA), max_acc) [m/yr], applied every step_MB with the reference's
    mask/clip logic (src/inverse/SIA2D/VJPs.jl:129-139)."""

    grad: float = 6e-3
    ELA: float = 2000.0
    max_acc: float = 1.2


@dataclass
class FieldMB:
    """Prescribed MB increment per step_MB, one field per glacier."""

    fields: Sequence[np.ndarray]


@dataclass
class SIA2D_A_target:
    """src/models/target/target_A.jl:9"""


@dataclass
class SIA2D_D_hybrid_target:
    """src/models/target/target_D_hybrid.jl:12-15: how dY/dθ is evaluated over the dual grid in ∂Diffusivity∂θ --
    :Linear (default) = exact gradients on 2 n_interp_half knots of H̄ (create_interpolation), interpolated linearly;
    :None = exact backprop at every node."""

    interpolation: str = "Linear"
    n_interp_half: int = 75


@dataclass
class SIA2D_D_target:
    """src/models/target/target_D_pure.jl:34-39: :None (default) = exact backprop at every node; :Linear = gradients on the
    fixed (2 n_interp_half)² node grid of LawU's p_VJP! (Laws.jl:128-169, both axes LinRange(0, 100, ·)), interpolated
    bilinearly in (H̄, |∇S|) -- a node with H̄ > 100 m is outside the interpolant and raises, as in the reference."""

    interpolation: str = "None"
    n_interp_half: int = 100


class Model:
    """Model(iceflow=..., mass_balance=..., regressors=(; A=nn); target=nothing)  (Model.jl:61-127); the target is
    inferred from the laws when not given (:102-114)."""

    def __init__(self, iceflow: SIA2Dmodel, mass_balance=None, regressors: Optional[Dict[str, NeuralNetwork]] = None,
                 target=None):
        kind = iceflow.law.kind
        want = SIA2D_D_target if kind == L.LAW_NN_U else SIA2D_D_hybrid_target if kind == L.LAW_NN_Y else SIA2D_A_target
        if target is None:
            target = want()
        elif not isinstance(target, want):
            raise ValueError(f"The provided laws do not match with the provided target. Make sure that the target is a {want.__name__}.")
        if getattr(target, "interpolation", "None") not in ("None", "Linear"):
            raise ValueError("Method to spatially compute gradient with respect to H̄ not specified.")  # target_D_hybrid.jl:161
        self.target = target
        self.iceflow = iceflow
        self.mass_balance = mass_balance
        self.regressors = regressors or {}
        law = iceflow.law
        self.theta = None if law.nn is None else law.nn.theta.copy()  # trainable_components.θ
        self.per_glacier = None
        if law.classical is not None:
            reg = self.regressors.get("A")
            if reg is None or not hasattr(reg, "sizes"):
                raise ValueError("classical LawA needs regressors={'A': GlacierWideInv|GriddedInv}")
            self.per_glacier = reg
            self.theta = reg.theta.copy()
        # θ = (A = ..., IC = ...): the law's parameters first, then one H0 matrix per glacier (Model.jl:155-165)
        self.IC = self.regressors.get("IC")
        self.n_main = 0 if self.theta is None else self.theta.size
        if self.IC is not None:
            main = np.zeros(0) if self.theta is None else self.theta
            self.theta = np.concatenate([main, self.IC.theta])
        self.trainable_components = TrainableComponents(self)


class TrainableComponents:
    """The trainable sub-models of a Model by the law they feed and the flat parameter vector θ (Model.jl:132-181): regressor
    slots A, C, n, Y, U, IC (None = emptyTrainableModel()), `target`, and θ = (A | Y | U = ..., IC = ...) concatenated in that
    order.  A view of the Model that owns it: `model.trainable_components.θ = v` sets `model.theta` (what
    `container.simulation.model.trainable_components.θ = container.θ` does in inversion_utils.jl:483)."""

    _SLOTS = ("A", "C", "n", "Y", "U", "IC")

    def __init__(self, model: "Model"):
        self._model = model

    def __getattr__(self, name):
        if name in TrainableComponents._SLOTS:
            return self._model.regressors.get(name)
        raise AttributeError(name)

    @property
    def target(self):
        return self._model.target

    @property
    def theta(self):
        return self._model.theta

    @theta.setter
    def theta(self, v):
        v = np.asarray(v, dtype=float)
        if self._model.theta is None or v.shape != self._model.theta.shape:
            raise ValueError("θ does not match the model's parameter vector")
        self._model.theta = v.copy()

    θ = theta  # the reference's field name: `model.trainable_components.θ = v` (inversion_utils.jl:483)

    def split_theta(self, theta, glacier_idx: int):
        """splitθ (Model.jl:189-200): the part of θ one glacier's simulation sees -- a FunctionalModel's parameters whole, a
        PerGlacierModel's / the initial condition's own slot only.  Returns {key: array}."""
        m, out = self._model, {}
        theta = np.asarray(theta, dtype=float)
        if m.n_main:
            main = theta[:m.n_main]
            key = "U" if m.iceflow.law.kind == L.LAW_NN_U else "Y" if m.iceflow.law.kind == L.LAW_NN_Y else "A"
            if m.per_glacier is not None:
                offs = np.concatenate([[0], np.cumsum(m.per_glacier.sizes)])
                main = main[offs[glacier_idx]:offs[glacier_idx + 1]]
            out[key] = main
        if m.IC is not None:
            offs = np.concatenate([[0], np.cumsum(m.IC.sizes)]) + m.n_main
            out["IC"] = theta[offs[glacier_idx]:offs[glacier_idx + 1]]
        return out


@dataclass
class InversionBinder:
    """Container handed to the ODE problem: the simulation and the θ it is solved with (sciml_utils.jl:21-24).  Here it is what
    `SIA2D_grad_b` / `run_b` build internally; exposed under the reference's name for callers that construct it themselves."""

    simulation: "_Simulation"
    theta: np.ndarray

    def apply(self):
        """simulation.model.trainable_components.θ = θ (inversion_utils.jl:483)"""
        self.simulation.model.trainable_components.theta = self.theta
        return self.simulation


# ----------------------------------------------------------------------------------------
# multi-GPU: glaciers shard across ranks (pmap of gradient.jl:9-10 -> one rank per GPU)
# ----------------------------------------------------------------------------------------
def shard_glaciers(cells: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment by cell count; ties by index."""
    order = sorted(range(len(cells)), key=lambda i: (-cells[i], i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += cells[i]
    return [sorted(s) for s in out]


_DIST = {"init": False, "rank": 0, "world": 1, "device": None, "comm": None}


def attach_rccl_comm(local: int):
    """Create the library's own RCCL communicator (odinn_comm_init_rank) for this rank: rank 0 draws the unique id, the
    already-initialised torch.distributed group carries its 128 bytes to the other ranks (NCCL's bootstrap contract;
    a Julia host would use Distributed or MPI for the same step).  From then on the all-reduce of [loss, dtheta] runs
    inside libodinn_hip (ncclAllReduce on device memory), not through torch."""
    import torch.distributed as dist

    from .batch import Comm

    rank, world = dist.get_rank(), dist.get_world_size()
    box = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    _DIST["comm"] = Comm(local, world, rank, box[0])
    return _DIST["comm"]


def init_distributed(backend: Optional[str] = None):
    """One process per GPU (torchrun env: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  backend
    'nccl' is RCCL on ROCm; 'gloo' for the CPU multi-process tests."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # ODINN_DEVICE / ODINN_DIST_BACKEND: run several ranks on ONE device with gloo collectives (tests on a 1-GPU box)
    local = int(os.environ.get("ODINN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if backend is None:
        backend = os.environ.get("ODINN_DIST_BACKEND")
    if world > 1 and not _DIST["init"]:
        import torch
        import torch.distributed as dist

        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _DIST.update(init=True, device=(f"cuda:{local}" if backend == "nccl" else "cpu"))
        if backend == "nccl" and _DIST.get("comm") is None:
            attach_rccl_comm(local)
    _DIST.update(rank=rank, world=world, local=local)
    return rank, world, local


def allreduce_loss_grad(loss: float, dtheta: np.ndarray):
    """sum over ranks of [loss, dtheta...]: the single collective of the path
    (SIA2D_grad!: sum(losses), aggregate∇θ -- gradient.jl:14,25; Model.jl:208-224)."""
    if _DIST["world"] <= 1 or not _DIST["init"]:
        return loss, dtheta
    if _DIST.get("comm") is not None:  # RCCL through the C ABI (one process per GPU)
        out = _DIST["comm"].allreduce_sum(np.concatenate([[loss], dtheta]))
        return float(out[0]), out[1:].copy()
    import torch
    import torch.distributed as dist

    buf = torch.tensor(np.concatenate([[loss], dtheta]), dtype=torch.float64, device=_DIST["device"])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    return float(out[0]), out[1:].copy()


# ----------------------------------------------------------------------------------------
# simulations
# ----------------------------------------------------------------------------------------
@dataclass
class TrainingStats:
    """src/simulations/results/Results.jl:19-28"""

    retcode: Optional[str] = None
    losses: List[float] = field(default_factory=list)
    niter: int = 0
    θ: Optional[np.ndarray] = None
    θ_hist: List[np.ndarray] = field(default_factory=list)
    grad_hist: List[np.ndarray] = field(default_factory=list)  # ∇θ_hist
    initial_conditions: Optional[dict] = None
    grad_norms: List[float] = field(default_factory=list)
    time_per_iter: List[float] = field(default_factory=list)


@dataclass
class TrainingResult:
    """src/results/TrainingResults.jl:6-12: what save_inversion_file! writes."""

    θ: np.ndarray
    θ_hist: List[np.ndarray]
    grad_hist: List[np.ndarray]  # ∇θ_hist
    losses: List[float]
    params: dict


def _params_dict(p):
    import dataclasses

    def conv(o):
        if dataclasses.is_dataclass(o) and not isinstance(o, type):
            return {"__type__": type(o).__name__, **{f.name: conv(getattr(o, f.name)) for f in dataclasses.fields(o)}}
        if isinstance(o, (list, tuple)):
            return [conv(x) for x in o]
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (int, float, str, bool)) or o is None:
            return o
        return repr(o)

    return conv(p)


def save_inversion_file(theta, simulation, path: Optional[str] = None, file_name: Optional[str] = None):
    """save_inversion_file!(sol, simulation; path, file_name) (src/results/trainingresult_utils.jl:4-33).
    Same record (θ, θ_hist, ∇θ_hist, losses, params); the container is NumPy .npz with the parameters as
    JSON because JLD2 cannot be written without Julia.  Returns the file path."""
    import json
    import os

    path = path or os.path.join(os.getcwd(), "data", "results", "inversions")
    os.makedirs(path, exist_ok=True)
    file_name = file_name or "_inversion_result.npz"
    st = simulation.stats
    out = os.path.join(path, file_name)
    np.savez(out, theta=np.asarray(theta, dtype=np.float64),
             theta_hist=np.asarray(st.θ_hist, dtype=np.float64).reshape(len(st.θ_hist), -1),
             grad_hist=np.asarray(st.grad_hist, dtype=np.float64).reshape(len(st.grad_hist), -1),
             losses=np.asarray(st.losses, dtype=np.float64),
             params=np.array(json.dumps(_params_dict(simulation.parameters))))
    return out if out.endswith(".npz") else out + ".npz"


def load_inversion_file(file: str) -> TrainingResult:
    """Read a record written by save_inversion_file (checkpoint / resume: pass result.θ as the
    regressor's theta)."""
    import json

    z = np.load(file, allow_pickle=False)
    return TrainingResult(z["theta"], list(z["theta_hist"]), list(z["grad_hist"]), list(z["losses"]),
                          json.loads(str(z["params"])))


class ScalarLogger:
    """Stand-in for the TensorBoardLogger of callback_diagnosis (callback_utils.jl:84-98): the same
    tags (train/loss, train/norm_grad, train/time_per_iter) and steps, one JSON object per line."""

    def __init__(self, file: str):
        import os

        os.makedirs(os.path.dirname(os.path.abspath(file)), exist_ok=True)
        self.file = file
        self._f = open(file, "a")

    def log_value(self, tag: str, value: float, step: int):
        import json

        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()


def _crc32c_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC32C = _crc32c_table()


def _crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for x in data:
        c = _CRC32C[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _pb_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_bytes(field: int, payload: bytes) -> bytes:
    return _pb_varint(field << 3 | 2) + _pb_varint(len(payload)) + payload


class TBLogger(ScalarLogger):
    """TensorBoardLogger.TBLogger(logdir) as callback_diagnosis uses it (callback_utils.jl:84-98): scalars written as a
    TensorBoard event file (`events.out.tfevents.<time>.<host>`: TFRecord framing with masked CRC-32C, `Event{wall_time,
    step, summary{value{tag, simple_value}}}` records after the `brain.Event:2` version record), so `tensorboard --logdir`
    reads a run of this package like a run of the reference."""

    def __init__(self, logdir: str):
        import os
        import socket
        import struct

        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self.file = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}")
        self._f = open(self.file, "ab")
        self._record(struct.pack("<Bd", 0x09, time.time()) + _pb_bytes(3, b"brain.Event:2"))

    def _record(self, event: bytes):
        import struct

        head = struct.pack("<Q", len(event))
        self._f.write(head + struct.pack("<I", _masked_crc(head)) + event + struct.pack("<I", _masked_crc(event)))
        self._f.flush()

    def log_value(self, tag: str, value: float, step: int):
        import struct

        val = _pb_bytes(1, tag.encode()) + struct.pack("<Bf", 0x15, float(value))   # Summary.Value{tag = 1, simple_value = 2}
        event = struct.pack("<Bd", 0x09, time.time()) + b"\x10" + _pb_varint(int(step)) + _pb_bytes(5, _pb_bytes(1, val))
        self._record(event)


def read_event_file(file: str):
    """Scalars of a TensorBoard event file as [(tag, step, value)], every CRC checked (the reader side of TBLogger; used by
    the tests and to reload a run's curves)."""
    import struct

    def varint(b, i):
        v = s_ = 0
        while True:
            v |= (b[i] & 0x7F) << s_
            s_ += 7
            i += 1
            if not b[i - 1] & 0x80:
                return v, i

    def fields(b):
        i = 0
        while i < len(b):
            key, i = varint(b, i)
            f, w = key >> 3, key & 7
            if w == 0:
                v, i = varint(b, i)
            elif w == 1:
                v, i = b[i:i + 8], i + 8
            elif w == 5:
                v, i = b[i:i + 4], i + 4
            elif w == 2:
                n, i = varint(b, i)
                v, i = b[i:i + n], i + n
            else:
                raise ValueError(f"wire type {w}")
            yield f, w, v

    out = []
    raw = open(file, "rb").read()
    i = 0
    while i < len(raw):
        head = raw[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", raw[i + 8:i + 12])[0] != _masked_crc(head):
            raise ValueError("event file: bad length CRC")
        ev = raw[i + 12:i + 12 + n]
        if struct.unpack("<I", raw[i + 12 + n:i + 16 + n])[0] != _masked_crc(ev):
            raise ValueError("event file: bad data CRC")
        i += 16 + n
        step, summ = 0, None
        for f, w, v in fields(ev):
            if f == 2 and w == 0:
                step = v
            elif f == 5 and w == 2:
                summ = v
        if summ is None:
            continue
        for f, w, v in fields(summ):
            if f == 1 and w == 2:
                tag, val = None, None
                for f2, w2, v2 in fields(v):
                    if f2 == 1 and w2 == 2:
                        tag = v2.decode()
                    elif f2 == 2 and w2 == 5:
                        (val,) = struct.unpack("<f", v2)
                out.append((tag, step, val))
    return out


def callback_diagnosis(theta, loss, grad, simulation, save: bool = False, tbLogger: Optional[ScalarLogger] = None,
                       path: Optional[str] = None):
    """callback_diagnosis(θ, l, simulation; save, tbLogger) (callback_utils.jl:60-110)."""
    st = simulation.stats
    now = time.perf_counter()
    st.losses.append(float(loss))
    st.θ_hist.append(np.array(theta, dtype=np.float64, copy=True))
    st.grad_hist.append(np.array(grad, dtype=np.float64, copy=True))
    st.grad_norms.append(float(np.linalg.norm(grad)))
    it = len(st.losses)
    last = getattr(st, "_lastCall", None)
    if last is not None:
        st.time_per_iter.append(now - last)
    if tbLogger is not None:
        tbLogger.log_value("train/loss", loss, it)
        tbLogger.log_value("train/norm_grad", np.linalg.norm(grad), it)
        if last is not None:
            tbLogger.log_value("train/time_per_iter", now - last, it)
    st._lastCall = now
    if save:
        save_inversion_file(theta, simulation, path=path, file_name="_inversion_result.npz")


@dataclass
class Results:
    """Per-glacier result (Sleipnir.Results fields used downstream: H, t, rgi_id)."""

    rgi_id: str
    t: List[float]
    H: List[np.ndarray]
    stats: object = None


class _Simulation:
    def __init__(self, model: Model, glaciers: Sequence[Glacier2D], parameters: Parameters):
        self.model = model
        self.glaciers = list(glaciers)
        self.parameters = parameters
        self.results: List[Results] = []
        self.stats = TrainingStats()
        self._batch: Optional[GlacierBatch] = None
        self._mine: List[int] = list(range(len(self.glaciers)))

    # tstops exactly as _batch_iceflow_UDE builds them (inversion_utils.jl:487-495) -- PER GLACIER: the `step` grid and
    # solver.tstops are shared, the thickness / velocity data times and the stops of the time-aggregated losses are the
    # glacier's own (gradient.jl:96-107 rebuilds the same table for the reverse loop and asserts it equals result.t)
    def _shared_stops(self):
        p = self.parameters
        return set(define_callback_steps(p.simulation.tspan, p.solver.step)) | set(float(t) for t in p.solver.tstops)

    def tstops_glacier(self, i: int):
        """Stop table of glacier i (index into simulation.glaciers)."""
        p = self.parameters
        g = self.glaciers[i]
        ts = self._shared_stops()
        if g.thicknessData is not None:
            ts |= set(float(t) for t in g.thicknessData.t)
        if g.velocityData is not None:
            ts |= set(float(t) for t in g.velocityData.t)
        regs_ = _split_loss(p.UDE.empirical_loss_function)[2]
        if g.dhdtData is not None and any(isinstance(r, LossDhdt) for r, _ in regs_):
            ts |= set(float(t) for t in g.dhdtData.t)  # discretePostIntegralLossSteps (TimeAggregatedLosses.jl:352-354)
        for r, _ in regs_:
            if isinstance(r, LossAvgV):  # :355-363; a grid point that is an existing stop up to rounding IS that stop
                for x in _avgv_times(g, r):
                    if not any(abs(x - t) <= 1e-9 for t in ts):
                        ts.add(x)
        return sorted(t for t in ts if p.simulation.tspan[0] <= t <= p.simulation.tspan[1])

    def tstops(self):
        """Union of the glaciers' tables: THE table when all glaciers share their data times; otherwise the table handed to
        the solver for the glaciers that have no table of their own (see _push_stops)."""
        ts = set()
        for i in range(len(self.glaciers)):
            ts |= set(self.tstops_glacier(i))
        out = []
        for t in sorted(ts):  # (grid points of time-aggregated losses computed from different dates agree up to rounding only)
            if not out or t - out[-1] > 1e-9:
                out.append(t)
        return out

    def _push_stops(self):
        """Hand every glacier of this rank its own stop table (no-op entries for those that only have the shared one)."""
        b = self.batch()
        shared = self.tstops()
        for k, gi in enumerate(self._mine):
            own = self.tstops_glacier(gi)
            b.set_glacier_stops(k, None if own == shared else own)
        return shared

    def mb_times(self):
        p = self.parameters
        if not (p.simulation.use_MB and self.model.mass_balance is not None):
            return []
        return define_callback_steps(p.simulation.tspan, p.simulation.step_MB)[1:]

    def batch(self) -> GlacierBatch:
        """Device context of this rank's shard (built once; state stays in HBM)."""
        if self._batch is not None:
            return self._batch
        rank, world, local = init_distributed() if self.parameters.simulation.multiprocessing else (0, 1, 0)
        cells = [g.nx * g.ny for g in self.glaciers]
        self._mine = shard_glaciers(cells, world)[rank]
        if not self._mine:
            raise RuntimeError(f"rank {rank} received no glacier (G={len(cells)} < world={world})")
        gl = [self.glaciers[i] for i in self._mine]
        p = self.parameters
        phys = []
        for g in gl:
            ph = PhysicalParameters(**{**p.physical.__dict__})
            ph.C, ph.n = g.C, g.n
            phys.append(ph)
        b = GlacierBatch([(g.nx, g.ny) for g in gl], [g.dx for g in gl], [g.dy for g in gl], phys,
                         A=[g.A for g in gl], T=[g.T for g in gl], device=local)
        law = self.model.iceflow.law
        for k, g in enumerate(gl):
            b.set_fields(k, g.H0, g.B)
            lf, _, _ = _split_loss(p.UDE.empirical_loss_function)
            dist_ = (lf.hLoss.loss.distance if isinstance(lf, LossHV) else lf.loss.distance)
            if g.thicknessData is not None and not isinstance(lf, (_NoDataTerm, LossV)):  # (only LossH / LossHV read it)
                b.set_reference(k, g.thicknessData.t, g.thicknessData.H, dist_)
            if g.velocityData is not None and len(g.velocityData.t) > 0:
                v = g.velocityData
                b.set_velocity_reference(k, v.t, v.vabs, v.vx, v.vy)
        lf, w_data, regs_ = _split_loss(p.UDE.empirical_loss_function)
        for r, w in regs_:
            if isinstance(r, VelocityRegularization):
                if any(g.velocityData is None or len(g.velocityData.t) < 2 for g in gl):
                    raise ValueError("VelocityRegularization is weighted by the intervals between the velocity-data times: "
                                     "every glacier needs velocityData with at least two dates")
                b.set_velocity_regularization(float(w) / w_data, r.distance)
            if isinstance(r, LossAvgV):  # evaluated on the device, weight relative to the data loss
                for k, g in enumerate(gl):
                    _avgv_times(g, r)  # validates velocityData
                    v = g.velocityData
                    b.set_avgv_reference(k, v.date1[0], v.date2[0], v.vabs[0], v.vx[0], v.vy[0])
                b.set_avgv_loss(float(w) / w_data, r.step, r.component)
            if isinstance(r, LossDhdt):  # the device evaluates the term; its weight is relative to the data loss, which
                for k, g in enumerate(gl):  # SIA2D_grad_b scales by w_data afterwards
                    if g.dhdtData is None:
                        raise ValueError("LossDhdt needs glacier.dhdtData")
                    b.set_dhdt_reference(k, g.dhdtData.t[0], g.dhdtData.t[1], g.dhdtData.dhdt)
                b.set_dhdt_loss(float(w) / w_data)
        if isinstance(lf, LossHV):
            b.set_loss(L.LOSS_HV, lf.vLoss.component, lf.vLoss.scale_loss, lf.scaling)
        elif isinstance(lf, LossV):
            b.set_loss(L.LOSS_V, lf.component, lf.scale_loss)
        hl = lf.hLoss if isinstance(lf, LossHV) else lf if isinstance(lf, LossH) else None
        if hl is not None and isinstance(hl.loss, LogSum):
            b.set_thickness_loss_function(hl.loss.eps)
        vl = lf.vLoss if isinstance(lf, LossHV) else lf if isinstance(lf, LossV) else None
        if vl is not None and isinstance(vl.loss, LogSum):
            b.set_velocity_loss_function(vl.loss.eps)
        if law.classical is not None:
            self._batch = b
            self._apply_classical(self.model.theta[:self.model.n_main])
        elif law.kind == L.LAW_CONST_A:
            if law.value is not None:
                for k in range(len(gl)):
                    b.set_A(k, gl[k].A if gl[k].A is not None else law.value)
        else:
            b.set_law(law.kind, law.mlp, self.model.theta[:self.model.n_main], law.n_H, law.n_gradS)
            tg = self.model.target
            if law.kind == L.LAW_NN_U:
                b.set_surface_velocity_factor(p.simulation.f_surface_velocity_factor)
            if isinstance(tg, (SIA2D_D_hybrid_target, SIA2D_D_target)):
                b.set_grad_interpolation(L.GRAD_INTERP_LINEAR if tg.interpolation == "Linear" else L.GRAD_INTERP_NONE,
                                         tg.n_interp_half)
        mb = self.model.mass_balance
        if p.simulation.use_MB and mb is not None:
            for k, g in enumerate(gl):
                if isinstance(mb, LinearMB):
                    S0 = g.B + np.maximum(g.H0, 0.0)
                    step = p.simulation.step_MB
                    b.set_mass_balance(k, mb.grad * (S0 - mb.ELA) * step, mb.grad * step, S0, mb.max_acc * step)
                else:
                    b.set_mass_balance(k, mb.fields[self._mine[k]])
        self._batch = b
        return b

    # classical per-glacier law: theta slots of glacier i are [offs[i], offs[i+1])
    def _slots(self):
        sizes = self.model.per_glacier.sizes
        offs = np.concatenate([[0], np.cumsum(sizes)])
        return sizes, offs

    def _ic_slots(self):
        offs = np.concatenate([[0], np.cumsum(self.model.IC.sizes)]) + self.model.n_main
        return offs

    def _apply_IC(self, theta):
        """H0 of every glacier of this rank from theta.IC (simulate_iceflow_UDE!, inversion_utils.jl:591-597)."""
        offs = self._ic_slots()
        filt = self.parameters.UDE.initial_condition_filter
        for k, gi in enumerate(self._mine):
            g = self.glaciers[gi]
            self._batch.set_fields(k, evaluate_H0(theta[offs[gi]:offs[gi + 1]], g, filt), g.B)

    def _apply_classical(self, theta):
        law = self.model.iceflow.law
        lo, hi = law.bounds
        sizes, offs = self._slots()
        for k, gi in enumerate(self._mine):
            th = np.asarray(theta[offs[gi]:offs[gi + 1]])
            A = lo + (hi - lo) * (np.tanh(th) + 1.0) / 2.0
            if law.classical == "scalar":
                self._batch.set_A(k, float(A[0]))
            else:
                g = self.glaciers[gi]
                self._batch.set_A_field(k, A.reshape((g.nx - 1, g.ny - 1), order="F"))

    def _solver_opts(self):
        s = self.parameters.solver
        return dict(reltol=s.reltol, abstol=s.abstol, dtmax=s.dtmax, maxiters=s.maxiters)


class Prediction(_Simulation):
    """Forward simulation (Huginn.Prediction; run!(prediction) == docs/src/quick_start.jl:11-30)."""


class Inversion(_Simulation):
    """Functional inversion (src/simulations/inversions/Inversion.jl:16-62)."""


FunctionalInversion = Inversion  # legacy name (scripts/benchmarks/sensealg_benchmark.jl:93)


def SIA2D_b(dH: np.ndarray, H: np.ndarray, simulation: _Simulation, t: float, theta=None, glacier_idx: int = 0):
    """Huginn.SIA2D!(dH, H, simulation, t, θ): in-place RHS of glacier ``glacier_idx`` of this rank's shard."""
    b = simulation.batch()
    if theta is not None and b.P:
        b.set_theta(theta)
    dH[...] = b.dhdt(glacier_idx, H, t)
    return None


def VJP_lambda_dSIAdH(VJPMode, lam, H, theta, simulation: _Simulation, t, glacier_idx: int = 0):
    """VJP_λ_∂SIA∂H(::DiscreteVJP | ::ContinuousVJP, λ, H, θ, simulation, t) -> (λ_∂f∂H, nothing)  (VJPs.jl:2-10)."""
    b = simulation.batch()
    b.set_vjp_method(L.VJP_CONTINUOUS if isinstance(VJPMode, ContinuousVJP) else L.VJP_DISCRETE)
    if theta is not None and b.P:
        b.set_theta(theta)
    return b.vjp_H(glacier_idx, lam, H, t), None


def VJP_lambda_dSIAdtheta(VJPMode: DiscreteVJP, lam, H, theta, dH_H, simulation: _Simulation, t, glacier_idx: int = 0):
    """VJP_λ_∂SIA∂θ(::DiscreteVJP, λ, H, θ, dH_H, simulation, t)  (VJPs.jl:30-33)."""
    b = simulation.batch()
    if theta is not None and b.P:
        b.set_theta(theta)
    return b.vjp_theta(glacier_idx, lam, H, t)


def V_from_H(simulation: _Simulation, H, t, theta=None, glacier_idx: int = 0):
    """Huginn.V_from_H(simulation, H, t, θ) -> (Vx, Vy, V) on the nx*ny grid (called Losses.jl:314,358)."""
    b = simulation.batch()
    if theta is not None and b.P:
        b.set_theta(theta)
    Vx, Vy = b.surface_V(glacier_idx, H)
    return Vx, Vy, np.sqrt(Vx ** 2 + Vy ** 2)


def SIA2D_grad_b(dtheta: np.ndarray, theta: np.ndarray, simulation: Inversion, _loss_only: bool = False, _grad=None):
    """SIA2D_grad!(dθ, θ, simulation): loss and gradient over ALL glaciers of ALL ranks
    (gradient.jl:6-31).  Returns the loss; dθ is written in place.  θ = [law parameters, IC matrices].
    (_loss_only / _grad: loss_iceflow_transient's forward-only evaluation of the same loss under an adjoint of its own -- passed in,
    the simulation's parameters are not touched.)"""
    b = simulation.batch()
    model = simulation.model
    law = model.iceflow.law
    p = simulation.parameters
    grad = p.UDE.grad if _grad is None else _grad
    _, w_data, regs = _split_loss(p.UDE.empirical_loss_function)
    if isinstance(grad, ContinuousAdjoint):  # gradient.jl:276
        def loss_grad(*a, **kw):
            return b.loss_grad_continuous(*a, adj_reltol=grad.reltol, adj_abstol=grad.abstol, adj_dtmax=grad.dtmax,
                                          n_quadrature=grad.n_quadrature, adj_maxiters=p.solver.maxiters, **kw)
    elif isinstance(grad, (DiscreteAdjoint, DummyAdjoint)):  # gradient.jl:129; DummyAdjoint: for the loss only
        loss_grad = b.loss_grad
    else:
        raise TypeError(f"adjoint method {type(grad).__name__} is not provided")
    if isinstance(grad.VJP_method, ContinuousVJP):
        b.set_vjp_method(L.VJP_CONTINUOUS)
    elif isinstance(grad.VJP_method, DiscreteVJP):
        b.set_vjp_method(L.VJP_DISCRETE)
    else:
        raise TypeError(f"VJP method {type(grad.VJP_method).__name__} is not supported yet.")  # gradient.jl:311-314
    theta = np.asarray(theta, dtype=np.float64)
    th_main = theta[:model.n_main]
    dth = np.zeros_like(theta)
    lf_data = _split_loss(p.UDE.empirical_loss_function)[0]
    if _loss_only and not any(isinstance(r, _AGGREGATED) for r, _ in regs) and not isinstance(lf_data, _NoDataTerm):
        # forward solve + the loss over its snapshots (the time-aggregated terms are evaluated by the gradient drivers only)
        def loss_grad(ts, theta=None, mb_times=(), **opts):  # noqa: F811
            if theta is not None:
                b.set_theta(theta)
            b.solve(ts, mb_times=mb_times, **opts)
            return float(np.sum(b.loss())), np.zeros(max(model.n_main, 1))

        def grad_parts():
            return np.zeros(b.G), np.zeros(b.G)

        def grad_field(k):
            return np.zeros((b.shapes[k][0] - 1, b.shapes[k][1] - 1), order="F")

        def lambda0(k):
            return np.zeros(b.shapes[k], order="F")
    elif isinstance(lf_data, _NoDataTerm) and not any(isinstance(r, _AGGREGATED) for r, _ in regs):
        # regularisers on the parameters alone (loss = RheologyRegularization(), runtests.jl:221-223): no term depends on the
        # solve -- the reference integrates a zero adjoint; nothing to run on the device
        def loss_grad(*a, **kw):  # noqa: F811
            return 0.0, np.zeros(max(model.n_main, 1))

        def grad_parts():
            return np.zeros(b.G), np.zeros(b.G)

        def grad_field(k):
            return np.zeros((b.shapes[k][0] - 1, b.shapes[k][1] - 1), order="F")

        def lambda0(k):
            return np.zeros(b.shapes[k], order="F")
    else:
        grad_parts, grad_field, lambda0 = b.grad_parts, b.grad_field, b.lambda0
    if model.IC is not None:
        simulation._apply_IC(theta)
    if law.classical is not None:
        # PerGlacierModel: every theta slot has a single owner (Model.jl:214-216); dL/dtheta = dL/dA * dA/dtheta
        simulation._apply_classical(th_main)
        loss, _ = loss_grad(simulation._push_stops(), mb_times=simulation.mb_times(), **simulation._solver_opts())
        lo, hi = law.bounds
        sizes, offs = simulation._slots()
        _, Gg = grad_parts()
        for k, gi in enumerate(simulation._mine):
            th = np.asarray(theta[offs[gi]:offs[gi + 1]])
            dA = (hi - lo) / 2.0 * (1.0 - np.tanh(th) ** 2)
            dLdA = Gg[k] if law.classical == "scalar" else grad_field(k).ravel(order="F")
            dth[offs[gi]:offs[gi + 1]] = dLdA * dA
    elif model.n_main:
        loss, dth[:model.n_main] = loss_grad(simulation._push_stops(), theta=th_main, mb_times=simulation.mb_times(),
                                             **simulation._solver_opts())
    else:  # only the initial condition is trainable
        loss, _ = loss_grad(simulation._push_stops(), mb_times=simulation.mb_times(), **simulation._solver_opts())
    if model.IC is not None:  # dL/dθ.IC = λ(t0) ⊙ ∂H0/∂θ.IC  (gradient.jl:262-271, :507-516)
        offs = simulation._ic_slots()
        for k, gi in enumerate(simulation._mine):
            g = simulation.glaciers[gi]
            s0 = evaluate_dH0(theta[offs[gi]:offs[gi + 1]], g, p.UDE.initial_condition_filter)
            dth[offs[gi]:offs[gi + 1]] = (lambda0(k) * s0).ravel(order="F")
    loss *= w_data  # MultiLoss: sum_k λ_k loss_k, same weights on the gradients (MultiLoss.jl:75-98,148-180)
    dth *= w_data
    tspan = p.simulation.tspan
    for reg, w in regs:
        if isinstance(reg, _AGGREGATED):
            continue  # evaluated on the device inside loss_grad (odinn_set_dhdt_loss / odinn_set_avgv_loss)
        if isinstance(reg, InitialThicknessRegularization):
            if model.IC is None:
                raise ValueError("Regularization with respect to initial condition requires to set initial "
                                 "condition as a trainable parameter.")  # Regularization.jl:152
            if reg.t0 != tspan[0]:
                continue  # evaluated only at t == t0, which discreteLossSteps adds to the stops (:153,:189)
            offs = simulation._ic_slots()
            for k, gi in enumerate(simulation._mine):
                g = simulation.glaciers[gi]
                H0 = evaluate_H0(theta[offs[gi]:offs[gi + 1]], g, p.UDE.initial_condition_filter)
                l, gH = b.tikhonov(H0, g.dx, g.dy)  # mask = trues (:157)
                loss += w * l
                dth[offs[gi]:offs[gi + 1]] += w * gH.ravel(order="F")  # as written: no filter chain rule (:185)
        elif isinstance(reg, RheologyRegularization):
            if law.classical != "gridded":
                raise ValueError("RheologyRegularization needs the gridded classical law LawA(params; scalar=false)")
            lo, hi = law.bounds
            sizes, offs = simulation._slots()
            for k, gi in enumerate(simulation._mine):
                g = simulation.glaciers[gi]
                th = theta[offs[gi]:offs[gi + 1]]
                A = (lo + (hi - lo) * (np.tanh(th) + 1.0) / 2.0).reshape((g.nx - 1, g.ny - 1), order="F")
                l, gA = b.tikhonov(A, g.dx, g.dy)  # mask = trues(size(H) .- 1)  (:272)
                loss += w * l
                dth[offs[gi]:offs[gi + 1]] += w * gA.ravel(order="F") * (hi - lo) * (1.0 - np.tanh(th) ** 2) / 2.0
    if isinstance(grad, DummyAdjoint):  # gradient.jl:540-545 (the same dummy vector on every rank)
        if grad.grad_function is not None:
            dth = np.asarray(grad.grad_function(theta), dtype=np.float64).reshape(theta.shape) / max(_DIST.get("world", 1), 1)
        else:
            dth = np.max(np.abs(theta)) * np.random.default_rng(grad.seed).random(theta.shape) / max(_DIST.get("world", 1), 1)
    loss, dth = allreduce_loss_grad(loss, dth)
    if np.linalg.norm(dth) > 1e7:  # gradient.jl:19-24
        import warnings

        warnings.warn(f"Potential unstable gradient: ‖dθ‖={np.linalg.norm(dth):.3e}. "
                      "Try reducing the temporal stepsize Δt used for reverse simulation.")
    dtheta[...] = dth
    return loss


def loss_iceflow_transient(theta: np.ndarray, simulation: Inversion) -> float:
    """loss_iceflow_transient(θ, simulation, pmap) (inversion_utils.jl:287-296): the loss of SIA2D_grad_b(θ) alone -- forward
    solves, the empirical loss with its weights and the regularisation terms (the function the reference's test_grad_finite_diff
    differences, test/test_grad_loss.jl:262-265).  Losses without a time-aggregated term are evaluated forward-only; with one
    (LossDhdt / velocity-regularisation terms, which only the gradient drivers evaluate) the DiscreteAdjoint driver runs for its loss,
    whatever adjoint the simulation is configured with -- the loss VALUE does not depend on the adjoint.  Re-entrant: the simulation's
    parameters are not modified and the batch's VJP method is restored."""
    b = simulation.batch()
    vjp_saved = getattr(b, "vjp_method", None)
    try:
        return SIA2D_grad_b(np.zeros_like(np.asarray(theta, dtype=np.float64)), theta, simulation, _loss_only=True,
                            _grad=DummyAdjoint(grad_function=lambda th: np.zeros_like(th), VJP_method=DiscreteVJP()))
    finally:
        if vjp_saved is not None:
            b.set_vjp_method(vjp_saved)


def _run_prediction(sim: Prediction):
    b = sim.batch()
    ts = sim._push_stops()
    stats = b.solve(ts, mb_times=sim.mb_times(), **sim._solver_opts())
    sim.results = []
    for k, gi in enumerate(sim._mine):  # every glacier's result holds ITS stops (Sleipnir.create_results, inversion_utils.jl:533-538)
        tg = sim.tstops_glacier(gi)
        sim.results.append(Results(sim.glaciers[gi].rgi_id, list(tg), [b.snapshot(k, j) for j in range(len(tg))], stats[k]))
    return sim.results


def _run_inversion(sim: Inversion, callback: Optional[Callable] = None, save: bool = False,
                   tbLogger: Optional[ScalarLogger] = None, path: Optional[str] = None):
    """train_UDE! (inversion_utils.jl:112-238): Adam or LBFGS over loss/grad callbacks, with
    callback_diagnosis bookkeeping (losses, θ_hist, ∇θ_hist, scalar log, intermediate saves)."""
    hyper = sim.parameters.hyper
    theta = sim.model.theta.copy()
    dth = np.zeros_like(theta)
    st = sim.stats
    opt = hyper.optimizer
    if isinstance(opt, Adam):
        m = np.zeros_like(theta)
        v = np.zeros_like(theta)
        b1, b2 = opt.beta
        for it in range(1, hyper.epochs + 1):
            loss = SIA2D_grad_b(dth, theta, sim)
            callback_diagnosis(theta, loss, dth, sim, save=save, tbLogger=tbLogger, path=path)
            m = b1 * m + (1 - b1) * dth
            v = b2 * v + (1 - b2) * dth * dth
            theta = theta - opt.eta * (m / (1 - b1 ** it)) / (np.sqrt(v / (1 - b2 ** it)) + opt.eps)
            st.niter = it
            if callback is not None:
                callback(it, loss, theta)
        st.retcode = "Default"
    else:
        from scipy.optimize import minimize

        def fg(x):
            loss = SIA2D_grad_b(dth, x, sim)
            callback_diagnosis(x, loss, dth, sim, save=save, tbLogger=tbLogger, path=path)
            return loss, dth.copy()

        res = minimize(fg, theta, jac=True, method="L-BFGS-B",
                       options=dict(maxiter=hyper.epochs, maxcor=opt.m, ftol=0.0, gtol=0.0))
        theta = res.x
        st.niter = len(st.losses)
        st.retcode = "Success" if res.success else str(res.message)
    sim.model.theta = theta
    st.θ = theta.copy()
    nm = sim.model.n_main
    if sim.model.iceflow.law.classical is not None:
        sim.model.per_glacier.theta = theta[:nm].copy()
        sim._apply_classical(theta[:nm])
    elif nm:
        sim.model.iceflow.law.nn.theta = theta[:nm].copy()
        sim.batch().set_theta(theta[:nm])
    if sim.model.IC is not None:
        sim.model.IC.theta = theta[nm:].copy()
        sim._apply_IC(theta)
    return st


def run_b(simulation: _Simulation, callback: Optional[Callable] = None, save: bool = False,
          tbLogger: Optional[ScalarLogger] = None, path: Optional[str] = None):
    """run!(simulation; save, tbLogger) for Prediction and Inversion (inversion_utils.jl:21-88)."""
    if isinstance(simulation, Inversion):
        if simulation.model.theta is None:
            raise ValueError("Inversion needs a trainable law (LawA/LawY/LawU) or a trainable initial condition")
        st = _run_inversion(simulation, callback, save=save, tbLogger=tbLogger, path=path)
        if save:
            save_inversion_file(st.θ, simulation, path=path, file_name="_inversion_result.npz")
        return st
    return _run_prediction(simulation)
