# OdinnHIP.jl -- the reference-side binding of libodinn_hip.so (see INTEGRATION.md).
#
# This is the file a maintainer would add to ODINN.jl (src/inverse/HIP/OdinnHIP.jl): one new VJP type
# and one new adjoint type whose methods ccall the C ABI of include/odinn_hip.h.  The build image has
# no Julia toolchain, so this file is NOT executed by the test-suite; tests/test_abi.py parses every
# ccall below and checks its symbol, argument count and pointer/scalar pattern against the header, and
# the same entry points are exercised from Python (odinn.jl_amd/_lib.py) by tests/ -m gpu.
module OdinnHIP
using ODINN, Huginn, Sleipnir
const lib = "libodinn_hip"            # odinn.jl_amd/csrc/libodinn_hip.so on LD_LIBRARY_PATH

struct Phys;  rho::Float64; g::Float64; eta0::Float64; n::Float64; p::Float64; q::Float64
              C::Float64; minA::Float64; maxA::Float64; end
struct GlacierDesc; nx::Int32; ny::Int32; dx::Float64; dy::Float64; phys::Phys
                    A::Float64; T::Float64; end
# odinn_solver_opts / odinn_adjoint_opts / odinn_solve_stats, field for field (include/odinn_hip.h)
struct SolverOpts; reltol::Float64; abstol::Float64; dtmax::Float64; dt0::Float64; fixed_dt::Float64
                   maxiters::Int64; scheme::Int32; dense::Int32; cfl::Float64; end
struct AdjointOpts; reltol::Float64; abstol::Float64; dtmax::Float64; n_quadrature::Int32; reserved::Int32
                    maxiters::Int64; end
# odinn_schedule: which of the library's equivalent kernel forms run (-1 = automatic); set_schedule!(batch; adj_fused = 0, ...)
struct Schedule; step_sc::Int32; fused_tiles::Int32; dhdt_strip::Int32; vjph_strip::Int32; vjpth_strip::Int32
                 snap_on_load::Int32; interp_streams::Int32; interp_batch::Int32; lawgrad_wave::Int32; vq_onepass::Int32
                 adj_fused::Int32; adj_skip::Int32; adj_segs::Int32; adj_rows::Int32; adj_theta_fused::Int32
                 reserved::NTuple{5, Int32}; end
SolverOpts(solver) = SolverOpts(solver.reltol, 1e-6, 0.0, 0.0, 0.0, Int64(solver.maxiters), Int32(0), Int32(0), 0.0)
AdjointOpts(grad::ODINN.ContinuousAdjoint, solver) =
    AdjointOpts(grad.reltol, grad.abstol, grad.dtmax, Int32(grad.n_quadrature), Int32(0), Int64(solver.maxiters))

check(rc) = rc == 0 || error(unsafe_string(ccall((:odinn_last_error, lib), Cstring, ())))

# the scalar long-term air temperature the A / Y laws take as input, obtained the way the reference's own
# targets obtain it (src/models/target/target_D_hybrid.jl:60, src/laws/laws_utils.jl:85)
mean_temp(simulation, i) = Huginn.get_input(Huginn.iAvgScalarTemp(), simulation, i, simulation.parameters.simulation.tspan[1])

"One device context per simulation (all glaciers of this process on one GPU)."
mutable struct Batch; h::Ptr{Cvoid}; end
function Batch(simulation; device = 0)
    ph = simulation.parameters.physical
    descs = map(enumerate(simulation.glaciers)) do (i, g)
        # n, p, q, C of the SIA2D cache (the exponents compute_D reads: target_D_hybrid.jl:174-185)
        c = ODINN.init_cache(simulation.model, simulation, i, simulation.model.trainable_components.θ).iceflow
        GlacierDesc(g.nx, g.ny, g.Δx, g.Δy,
            Phys(ph.ρ, ph.g, ph.η₀, c.n.value, c.p.value, c.q.value, c.C.value, ph.minA, ph.maxA),
            g.A, mean_temp(simulation, i))
    end
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:odinn_batch_create, lib), Cint, (Cint, Cint, Ptr{GlacierDesc}, Ptr{Ptr{Cvoid}}),
                device, length(descs), descs, h))
    b = Batch(h[])
    dl = data_loss(simulation.parameters.UDE.empirical_loss_function)
    hl = dl isa ODINN.LossHV ? dl.hLoss : dl isa ODINN.LossH ? dl : nothing   # the thickness term (its L2Sum / LogSum carries `distance`)
    for (i, g) in enumerate(simulation.glaciers)
        check(ccall((:odinn_set_fields, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}),
                    b.h, i - 1, g.H₀, g.B))
        if !isnothing(g.thicknessData) && !isnothing(hl)   # (no LossH / LossHV in the loss: no thickness data term)
            tH = collect(Float64, ODINN.tdata(g.thicknessData)); Hs = reduce(hcat, vec.(g.thicknessData.H))
            check(ccall((:odinn_set_reference, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Cint),
                        b.h, i - 1, length(tH), tH, Hs, hl.loss.distance))
        end
        tV = collect(Float64, ODINN.tdata(g.velocityData, simulation.parameters.simulation.mapping))
        if length(tV) > 0   # LossV / LossHV data, and the dates VelocityRegularization is weighted by
            v = g.velocityData
            check(ccall((:odinn_set_velocity_reference, lib), Cint,
                        (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                        b.h, i - 1, length(tV), tV, reduce(hcat, vec.(v.vabs)), reduce(hcat, vec.(v.vx)), reduce(hcat, vec.(v.vy))))
        end
    end
    vl = dl isa ODINN.LossHV ? dl.vLoss : dl isa ODINN.LossV ? dl : nothing
    check(ccall((:odinn_set_loss, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cdouble), b.h,
                dl isa ODINN.LossHV ? 2 : dl isa ODINN.LossV ? 1 : 0,
                (!isnothing(vl) && vl.component == :abs) ? 1 : 0, (isnothing(vl) || vl.scale_loss) ? 1 : 0,
                dl isa ODINN.LossHV ? dl.scaling : 1.0))
    check(ccall((:odinn_set_surface_velocity_factor, lib), Cint, (Ptr{Cvoid}, Cdouble),
                b.h, simulation.parameters.simulation.f_surface_velocity_factor))   # target :D: Velocityꜛ = U / f
    set_time_aggregated_losses!(b, simulation)   # (after the velocity dates are known to the library)
    set_grad_interpolation!(b, simulation)
    set_glacier_stops!(b, simulation)
    finalizer(x -> ccall((:odinn_batch_destroy, lib), Cint, (Ptr{Cvoid},), x.h), b)
end

# the data term of the empirical loss (LossH / LossV / LossHV, alone or inside a MultiLoss); `nothing` when the loss holds
# time-aggregated terms and regularisers only (MultiLoss((LossDhdt(),), (1,)): no thickness / velocity data term)
function data_loss(lf)
    terms = lf isa ODINN.MultiLoss ? collect(lf.losses) : [lf]
    k = findfirst(l -> l isa Union{ODINN.LossH, ODINN.LossV, ODINN.LossHV}, terms)
    return isnothing(k) ? nothing : terms[k]
end

# The reference builds tstops PER GLACIER: the `step` grid and solver.tstops, shared, plus the glacier's own data times
# and the stops of its time-aggregated losses (inversion_utils.jl:487-495; gradient.jl:96-107 rebuilds the same table
# and asserts it equals result.t).  The library keeps one table per glacier: odinn_set_glacier_stops.
function glacier_tstops(simulation, i)
    params = simulation.parameters; glacier = simulation.glaciers[i]; lf = params.UDE.empirical_loss_function
    ts = unique(vcat(Huginn.define_callback_steps(params.simulation.tspan, params.solver.step), params.solver.tstops))
    return sort(unique(vcat(ts, ODINN.tdata(glacier.thicknessData), ODINN.tdata(glacier.velocityData, params.simulation.mapping),
                            unique(ODINN.discreteLossSteps(lf, params.simulation.tspan)),
                            unique(ODINN.discretePostIntegralLossSteps(lf, simulation, i)))))
end
function set_glacier_stops!(b::Batch, simulation)
    for i in eachindex(simulation.glaciers)
        ts = collect(Float64, glacier_tstops(simulation, i))
        check(ccall((:odinn_set_glacier_stops, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), b.h, i - 1, length(ts), ts))
    end
end

function set_schedule!(b; kw...)
    f = fieldnames(Schedule)[1:15]
    unknown = setdiff(keys(kw), f)
    isempty(unknown) || error("unknown schedule field(s): $(unknown)")
    sc = Ref(Schedule((Int32(get(kw, k, -1)) for k in f)..., ntuple(_ -> Int32(0), 5)))
    check(ccall((:odinn_set_schedule, lib), Cint, (Ptr{Cvoid}, Ptr{Schedule}), b.h, sc))
end

# terms of a MultiLoss the library evaluates next to the data loss (src/losses/TimeAggregatedLosses.jl, Regularization.jl:192-245):
# data and weight relative to the data loss
function set_time_aggregated_losses!(b::Batch, simulation)
    lf = simulation.parameters.UDE.empirical_loss_function
    terms = lf isa ODINN.MultiLoss ? collect(zip(lf.losses, lf.λs)) : [(lf, 1.0)]
    wdata = something(findfirst(t -> t[1] isa Union{ODINN.LossH, ODINN.LossV, ODINN.LossHV}, terms), 0)
    wdata = wdata == 0 ? 1.0 : terms[wdata][2]
    # the simple loss inside the data terms: L2Sum (default) or LogSum(ϵ) (Losses.jl:34-49)
    for (l, _) in terms
        hl = l isa ODINN.LossHV ? l.hLoss : l isa ODINN.LossH ? l : nothing
        vl = l isa ODINN.LossHV ? l.vLoss : l isa ODINN.LossV ? l : nothing
        if !isnothing(hl) && hl.loss isa ODINN.LogSum
            check(ccall((:odinn_set_thickness_loss_function, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble), b.h, 1, hl.loss.ϵ))
        end
        if !isnothing(vl) && vl.loss isa ODINN.LogSum
            check(ccall((:odinn_set_velocity_loss_function, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble), b.h, 1, vl.loss.ϵ))
        end
    end
    for (l, w) in terms
        if l isa ODINN.LossDhdt
            for (i, g) in enumerate(simulation.glaciers)
                check(ccall((:odinn_set_dhdt_reference, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cdouble),
                            b.h, i - 1, g.dhdtData.t[1], g.dhdtData.t[2], g.dhdtData.dhdt))
            end
            check(ccall((:odinn_set_dhdt_loss, lib), Cint, (Ptr{Cvoid}, Cdouble), b.h, w / wdata))
        elseif l isa ODINN.VelocityRegularization   # weighted by the intervals between the velocity-data dates (set_velocity_reference)
            check(ccall((:odinn_set_velocity_regularization, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint), b.h, w / wdata, l.distance))
        elseif l isa ODINN.LossAvgV
            for (i, g) in enumerate(simulation.glaciers)
                v = g.velocityData
                t1 = Sleipnir.datetime_to_floatyear(only(v.date1)); t2 = Sleipnir.datetime_to_floatyear(only(v.date2))
                check(ccall((:odinn_set_avgv_reference, lib), Cint,
                            (Ptr{Cvoid}, Cint, Cdouble, Cdouble, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            b.h, i - 1, t1, t2, only(v.vabs), only(v.vx), only(v.vy)))
            end
            check(ccall((:odinn_set_avgv_loss, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cint),
                        b.h, w / wdata, l.step, l.component == :abs ? 1 : 0))
        end
    end
end

# `interpolation` / `n_interp_half` of the :D_hybrid and :D targets (target_D_hybrid.jl:12-15, target_D_pure.jl:34-39);
# called again after odinn_set_law, which resets the mode to the law's default
function set_grad_interpolation!(b::Batch, simulation)
    tc = simulation.model.trainable_components
    (isnothing(tc) || !hasproperty(tc, :target) || !hasproperty(tc.target, :interpolation)) && return
    check(ccall((:odinn_set_grad_interpolation, lib), Cint, (Ptr{Cvoid}, Cint, Cint),
                b.h, tc.target.interpolation == :Linear ? 1 : 0, tc.target.n_interp_half))
end

# ---- seam 1: the ODE right-hand side  (replaces Huginn.SIA2D! in SIA2D_UDE!,
#      src/simulations/inversions/inversion_utils.jl:691-699) -----------------------------
function SIA2D_HIP!(dH::Matrix{Float64}, H::Matrix{Float64}, b::Batch, i::Integer, t::Real)
    check(ccall((:odinn_sia2d_dhdt, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cdouble, Ptr{Float64}),
                b.h, i - 1, H, t, dH))
end

# ---- seam 2: the VJP flavour  (src/inverse/VJPTypes.jl, src/inverse/SIA2D/VJPs.jl:2-59) -
struct HIPVJP <: ODINN.AbstractVJPMethod; batch::Batch; end

function ODINN.VJP_λ_∂SIA∂H(m::HIPVJP, λ, H, θ, simulation, t)
    dλ = similar(H); i = simulation.cache.iceflow.glacier_idx
    check(ccall((:odinn_sia2d_vjp_H, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}),
                m.batch.h, i - 1, λ, H, t, dλ))
    return dλ, nothing
end

function ODINN.VJP_λ_∂SIA∂θ(m::HIPVJP, λ, H, θ, dH_H, simulation, t)
    v = zeros(length(θ)); i = simulation.cache.iceflow.glacier_idx
    check(ccall((:odinn_sia2d_vjp_theta, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Float64}, Cint),
                m.batch.h, i - 1, λ, H, t, v, length(v)))
    return ODINN.Vector2ComponentVector(v, θ)
end

function ODINN.VJP_λ_∂MB∂H(m::HIPVJP, λ, H, simulation, glacier, t)
    out = similar(H); i = simulation.cache.iceflow.glacier_idx
    check(ccall((:odinn_mb_vjp_H, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.batch.h, i - 1, λ, H, out))
    return out
end

# ---- multi-GPU: one Julia process per GPU; the library owns the RCCL communicator.  Rank 0 draws the 128-byte
#      unique id, the host distributes it (Distributed.remotecall_fetch / MPI.Bcast!) -- NCCL's bootstrap contract.
mutable struct Comm; h::Ptr{Cvoid}; end
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:odinn_comm_get_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    return id
end
function Comm(device::Integer, nranks::Integer, rank::Integer, id::Vector{UInt8})
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:odinn_comm_init_rank, lib), Cint, (Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                device, nranks, rank, id, h))
    finalizer(x -> ccall((:odinn_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.h), Comm(h[]))
end

# ---- seam 3 (the fast path): the whole gradient on the device ----------------------------
#      a new adjoint type next to DiscreteAdjoint / ContinuousAdjoint (src/inverse/AdjointTypes.jl:53-91);
#      SIA2D_grad! (gradient.jl:6-31) gains the method below: this rank's batch of glaciers is solved and
#      differentiated on its GPU, and ONE ncclAllReduce of [loss, dθ] inside the library replaces
#      pmap + sum(losses) + aggregate∇θ (gradient.jl:9-25, Model.jl:208-224).
struct HIPAdjoint{G <: ODINN.AbstractAdjointMethod} <: ODINN.AbstractAdjointMethod
    batch::Batch; comm::Union{Comm, Nothing}; method::G; VJP_method::HIPVJP
end

function SIA2D_grad_HIP!(dθ, θ, simulation, adj::HIPAdjoint, tstops, tstopsMB)
    solver = simulation.parameters.solver
    opts = Ref(SolverOpts(solver))
    continuous = adj.method isa ODINN.ContinuousAdjoint
    aopts = Ref(continuous ? AdjointOpts(adj.method, solver) : AdjointOpts(0.0, 0.0, 0.0, Int32(0), Int32(0), Int64(0)))
    loss = Ref(0.0); g = zeros(length(θ)); θv = ODINN.ComponentVector2Vector(θ)
    ts = collect(Float64, tstops); tmb = collect(Float64, tstopsMB)
    check(ccall((:odinn_batch_loss_grad, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64},
                 Ptr{SolverOpts}, Ptr{AdjointOpts}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
                adj.batch.h, isnothing(adj.comm) ? C_NULL : adj.comm.h, continuous ? 1 : 0, θv, length(θv),
                length(ts), ts, length(tmb), tmb, opts, aopts, loss, g, C_NULL, C_NULL))
    dθ .= ODINN.Vector2ComponentVector(g, θ)
    return loss[]
end

# single-rank pieces of the same seam, for callers that keep the reference's own reduction:
function SIA2D_grad_batch_HIP!(θ, simulation, adj::HIPAdjoint, tstops, tstopsMB)
    opts = Ref(SolverOpts(simulation.parameters.solver))
    loss = Ref(0.0); dθ = zeros(length(θ)); θv = ODINN.ComponentVector2Vector(θ)
    ts = collect(Float64, tstops); tmb = collect(Float64, tstopsMB)
    if adj.method isa ODINN.ContinuousAdjoint   # the reference's DEFAULT gradient (gradient.jl:276-539)
        aopts = Ref(AdjointOpts(adj.method, simulation.parameters.solver))
        check(ccall((:odinn_loss_grad_continuous, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{SolverOpts},
                     Ptr{AdjointOpts}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
                    adj.batch.h, θv, length(θv), length(ts), ts, length(tmb), tmb, opts, aopts, loss, dθ, C_NULL, C_NULL))
    else                                        # DiscreteAdjoint (gradient.jl:129-275)
        check(ccall((:odinn_loss_grad, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{SolverOpts},
                     Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                    adj.batch.h, θv, length(θv), length(ts), ts, length(tmb), tmb, opts, loss, dθ, C_NULL))
    end
    return loss[], [ODINN.Vector2ComponentVector(dθ, θ)]     # same tuple SIA2D_grad_batch! returns (gradient.jl:554)
end
end # module
