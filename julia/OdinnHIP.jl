# OdinnHIP.jl -- the reference-side binding of libodinn_hip.so (see INTEGRATION.md).
#
# This is the file a maintainer would add to ODINN.jl (src/inverse/HIP/OdinnHIP.jl): one new VJP type
# and one new adjoint type whose methods ccall the C ABI of include/odinn_hip.h.  The build image has
# no Julia toolchain, so this file is NOT executed by the test-suite; tests/test_abi.py parses every
# ccall below and checks its symbol, argument count and pointer/scalar pattern against the header, and
# the same entry points are exercised from Python (odinn.jl_amd/_lib.py) by tests/ -m gpu.
module OdinnHIP
using ODINN, Huginn, Sleipnir, Lux
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
                 law_table::Int32; interp_async::Int32; adj_sc::Int32; adj_ut_fused::Int32; reserved::NTuple{1, Int32}; end
# odinn_mlp_desc: a Lux.Chain of Dense layers with ODINN's pre / post-scaling (ML_utils.jl:23-39, target_utils.jl:58-141)
struct MlpDesc; n_layers::Int32; widths::NTuple{9, Int32}; acts::NTuple{8, Int32}; has_prescale::Int32
                pre_lo::NTuple{2, Float64}; pre_hi::NTuple{2, Float64}; post_kind::Int32; post_lo::Float64; post_hi::Float64; end
struct SolveStats; naccept::Int64; nreject::Int64; nrhs::Int64; t_final::Float64; dt_last::Float64; end
# enum odinn_law_kind / odinn_act / odinn_post
const LAW_CONST_A, LAW_NN_A_SCALAR, LAW_NN_A_GRIDDED, LAW_NN_Y, LAW_NN_U = 0, 1, 2, 3, 4
const POST_NONE, POST_AFFINE, POST_EXPMAX, POST_SCALE = 0, 1, 2, 3
SolverOpts(solver) = SolverOpts(solver.reltol, 1e-6, 0.0, 0.0, 0.0, Int64(solver.maxiters), Int32(0), Int32(0), 0.0)
AdjointOpts(grad::ODINN.ContinuousAdjoint, solver) =
    AdjointOpts(grad.reltol, grad.abstol, grad.dtmax, Int32(grad.n_quadrature), Int32(0), Int64(solver.maxiters))

check(rc) = rc == 0 || error(unsafe_string(ccall((:odinn_last_error, lib), Cstring, ())))

# the scalar long-term air temperature the A / Y laws take as input, obtained the way the reference's own
# targets obtain it (src/models/target/target_D_hybrid.jl:60, src/laws/laws_utils.jl:85)
mean_temp(simulation, i) = Huginn.get_input(Huginn.iAvgScalarTemp(), simulation, i, simulation.parameters.simulation.tspan[1])

"One device context per simulation (all glaciers of this process on one GPU)."
mutable struct Batch; h::Ptr{Cvoid}; end
function Batch(simulation; device = 0, law_kw...)   # law_kw: prescale_bounds / max_NN as they were given to LawY / LawU
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
    set_law!(b, simulation; law_kw...)           # the NN_θ law of the target (LawA / LawY / LawU) with its Lux chain, scalings and θ
    set_grad_interpolation!(b, simulation)       # (after odinn_set_law, which resets the mode to the law's default)
    set_mass_balance!(b, simulation)
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
    f = fieldnames(Schedule)[1:19]
    unknown = setdiff(keys(kw), f)
    isempty(unknown) || error("unknown schedule field(s): $(unknown)")
    sc = Ref(Schedule((Int32(get(kw, k, -1)) for k in f)..., ntuple(_ -> Int32(0), 1)))
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

# ---- the NN_θ law  (src/laws/Laws.jl:97-183 LawU, :240-273 LawY, :323-386 LawA; the regressor: NeuralNetwork.jl:18-74) ----------
# NNlib activation -> enum odinn_act.  ODINN builds its default chains with anonymous wrappers (`x -> softplus.(x)`,
# ML_utils.jl:26-36), so the activation is identified by its values at two probe points, not by identity.
function act_code(f)
    val(g, x) = (y = g(x); y isa AbstractArray ? only(y) : y)
    for (code, g) in ((0, identity), (1, Lux.softplus), (2, Lux.sigmoid), (3, Lux.gelu), (4, tanh), (5, Lux.relu))
        all(x -> isapprox(val(f, x), val(g, x); rtol = 1e-12, atol = 1e-300), (-0.7, 0.3)) && return Int32(code)
    end
    error("OdinnHIP: activation $(f) is none of identity / softplus / sigmoid / gelu / tanh / relu")
end

"`odinn_mlp_desc` of a `Lux.Chain(Dense...)`; pre / post-scaling as LawA / LawY / LawU apply them around the chain."
function MlpDesc(architecture::Lux.Chain; prescale_bounds = nothing, post_kind = POST_NONE, post_lo = 0.0, post_hi = 1.0)
    layers = collect(values(architecture.layers))
    all(l -> l isa Lux.Dense, layers) || error("OdinnHIP: only chains of Dense layers are bound (got $(typeof.(layers))); " *
        "express input normalisation / output scaling through prescale_bounds / max_NN")
    nl = length(layers)
    (1 <= nl <= 8 && all(l -> l.out_dims <= 32, layers) && layers[1].in_dims <= 2 && layers[end].out_dims == 1) ||
        error("OdinnHIP: the library takes 1-2 inputs, at most 8 Dense layers of at most 32 units and one output")
    all(l -> l.use_bias isa Lux.True || l.use_bias === true, layers) || error("OdinnHIP: Dense layers without bias are not bound")
    widths = zeros(Int32, 9); acts = zeros(Int32, 8)
    widths[1] = layers[1].in_dims
    for (k, l) in enumerate(layers)
        widths[k + 1] = l.out_dims; acts[k] = act_code(l.activation)
    end
    lo = zeros(2); hi = ones(2)
    if !isnothing(prescale_bounds)
        length(prescale_bounds) == layers[1].in_dims || error("OdinnHIP: one (lo, hi) pair per network input")
        for (k, (a, b)) in enumerate(prescale_bounds); lo[k] = a; hi[k] = b; end
    end
    return MlpDesc(Int32(nl), Tuple(widths), Tuple(acts), Int32(isnothing(prescale_bounds) ? 0 : 1), Tuple(lo), Tuple(hi),
                   Int32(post_kind), Float64(post_lo), Float64(post_hi))
end

# which law the target trains, the key of its parameters in θ, and the exponents of target :D_hybrid
function law_of(simulation)
    target = ODINN.targetType(simulation.model.trainable_components.target)
    target == :A && return (simulation.model.iceflow.A, :A)
    target == :D_hybrid && return (simulation.model.iceflow.Y, :Y)
    target == :D && return (simulation.model.iceflow.U, :U)
    error("OdinnHIP: unknown target $(target)")
end

"""
    set_law!(batch, simulation; prescale_bounds, max_NN)

Hands the NN_θ law of the simulation to the library (`odinn_set_law`): the Lux chain of the regressor as an `odinn_mlp_desc`,
θ flattened as `ComponentVector2Vector` flattens it (`[vec(W), b]` per layer), and the scalings LawA / LawY / LawU wrap around
the chain.  Those scalings live inside the laws' closures; `prescale_bounds` / `max_NN` must be the values the law was
constructed with (defaults = the constructors' defaults, Laws.jl:101-103,240-244).  A model without a regressor (a
classical `LawA(params)` or a constant A) keeps the constant-A law of `odinn_batch_create`.
"""
function set_law!(b::Batch, simulation; prescale_bounds = :default, max_NN = :default)
    tc = simulation.model.trainable_components
    (isnothing(tc) || !hasproperty(tc, :regressors) || isnothing(tc.regressors)) && return b
    law, key = law_of(simulation)
    hasproperty(tc.regressors, key) || return b
    nn = getproperty(tc.regressors, key)
    nn isa ODINN.NeuralNetwork || return b            # per-glacier classical inversions: set_A! / set_A_field! per iteration
    ph = simulation.parameters.physical
    θv = collect(Float64, ODINN.ComponentVector2Vector(getproperty(tc.θ, key)))
    nH = -1.0; nS = -1.0
    if key == :A
        gridded = ODINN.inputs(law) == ODINN._inputs_A_law_gridded
        kind = gridded ? LAW_NN_A_GRIDDED : LAW_NN_A_SCALAR
        desc = MlpDesc(nn.architecture; post_kind = POST_AFFINE, post_lo = ph.minA, post_hi = ph.maxA)   # scale(·, (minA, maxA)), Laws.jl:351
        if gridded   # the law's input field: long-term air temperature on the grid D lives on
            for (i, g) in enumerate(simulation.glaciers)
                T = Huginn.get_input(Huginn.iAvgGriddedTemp(), simulation, i, simulation.parameters.simulation.tspan[1])
                size(T) == (g.nx - 1, g.ny - 1) || error("OdinnHIP: gridded temperature of glacier $(i) is $(size(T)), the dual grid is $((g.nx - 1, g.ny - 1))")
                check(ccall((:odinn_set_T_field, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), b.h, i - 1, collect(Float64, T)))
            end
        end
    elseif key == :Y
        kind = LAW_NN_Y
        bounds = prescale_bounds === :default ? [(-25.0, 0.0), (0.0, 500.0)] : prescale_bounds        # Laws.jl:244
        mx = (max_NN === :default || isnothing(max_NN)) ? ph.maxA : max_NN                            # Laws.jl:248
        desc = MlpDesc(nn.architecture; prescale_bounds = bounds, post_kind = POST_EXPMAX, post_hi = mx)
        c = ODINN.init_cache(simulation.model, simulation, 1, tc.θ).iceflow                            # n_H, n_∇S: target_D_hybrid.jl:180-185
        nH = c.n_H.value; nS = c.n_∇S.value
    else
        kind = LAW_NN_U
        bounds = prescale_bounds === :default ? nothing : prescale_bounds                              # Laws.jl:101-102
        mx = max_NN === :default ? nothing : max_NN
        desc = MlpDesc(nn.architecture; prescale_bounds = bounds, post_kind = isnothing(mx) ? POST_NONE : POST_EXPMAX,
                       post_hi = isnothing(mx) ? 1.0 : mx)
    end
    check(ccall((:odinn_set_law, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{MlpDesc}, Ptr{Float64}, Cint, Cdouble, Cdouble),
                b.h, kind, Ref(desc), θv, length(θv), nH, nS))
    return b
end

"θ of the law for the calls that follow (the gradient entry points also take θ directly)."
set_theta!(b::Batch, θv::Vector{Float64}) =
    check(ccall((:odinn_set_theta, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), b.h, θv, length(θv)))
# classical inversions (LawA(params; scalar), Laws.jl:402-460): the per-glacier A the host derived from θ.A[glacier_id]
set_A!(b::Batch, i::Integer, A::Real) = check(ccall((:odinn_set_A, lib), Cint, (Ptr{Cvoid}, Cint, Cdouble), b.h, i - 1, A))
set_A_field!(b::Batch, i::Integer, A::Matrix{Float64}) =
    check(ccall((:odinn_set_A_field, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), b.h, i - 1, A))
"value of the law at H (scalar laws: one entry; matrix laws: the dual grid) -- eval_law / callback_plots_A"
function eval_law(b::Batch, i::Integer, H::Matrix{Float64}; scalar::Bool = false)
    out = scalar ? zeros(1) : zeros(size(H, 1) - 1, size(H, 2) - 1)
    check(ccall((:odinn_eval_law, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Cint), b.h, i - 1, H, out, length(out)))
    return scalar ? out[1] : out
end

# ---- mass balance  (the PeriodicCallback of inversion_utils.jl:498-517; VJP: VJPs.jl:107-151) --------------------------------
# The library's mass-balance model is a prescribed increment per step_MB with an optional linear elevation feedback,
# mb = min(mb0 + dmb_dS (S - S_ref), mb_max), masked and clipped as apply_MB_mask! does.  TImodel1's climate downscaling
# (Muninn / OGGM climate files) stays on the host: MB_timestep! is evaluated once on the initial surface and handed over as mb0.
function set_mass_balance!(b::Batch, simulation; dmb_dS = 0.0, mb_max = Inf)
    params = simulation.parameters
    (params.simulation.use_MB && !isnothing(simulation.model.mass_balance)) || return b
    for (i, g) in enumerate(simulation.glaciers)
        cache = ODINN.init_cache(simulation.model, simulation, i, simulation.model.trainable_components.θ)
        g.S .= g.B .+ g.H₀
        Huginn.MB_timestep!(cache, simulation.model, g, params.simulation.step_MB, params.simulation.tspan[1])
        mb0 = collect(Float64, cache.iceflow.MB); S0 = collect(Float64, g.S)
        check(ccall((:odinn_set_mass_balance, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cdouble, Ptr{Float64}, Cdouble),
                    b.h, i - 1, mb0, dmb_dS, S0, mb_max))
    end
    return b
end
mb_times(simulation) = simulation.parameters.simulation.use_MB ?
    collect(Float64, Huginn.define_callback_steps(simulation.parameters.simulation.tspan, simulation.parameters.simulation.step_MB)[2:end]) : Float64[]

# ---- the H-VJP stencil inside both adjoints  (adj.method.VJP_method, VJPTypes.jl:29-50) -------------------------------------
set_vjp_method!(b::Batch, m) =
    check(ccall((:odinn_set_vjp_method, lib), Cint, (Ptr{Cvoid}, Cint), b.h, m isa ODINN.ContinuousVJP ? 1 : 0))

# ---- run!(::Prediction) / the forward solve  (_batch_iceflow_UDE + simulate_iceflow_UDE!, inversion_utils.jl:472-572) --------
"""
    solve!(batch, simulation) -> Vector of (t = tstops_i, u = [H(t) for t in tstops_i], stats)

Integrates every glacier of the batch over `tspan` on the device (RDPK3Sp35 + PID as `params.solver` asks, mass balance at
`step_MB`, a snapshot at each of the glacier's own tstops) and returns per glacier what `Sleipnir.create_results` reads
from the ODE solution.
"""
function solve!(b::Batch, simulation)
    G = length(simulation.glaciers)
    ts = collect(Float64, glacier_tstops(simulation, 1)); tmb = mb_times(simulation)   # (glaciers with own tables: set_glacier_stops!)
    opts = Ref(SolverOpts(simulation.parameters.solver)); stats = Vector{SolveStats}(undef, G)
    check(ccall((:odinn_solve, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{SolverOpts}, Ptr{SolveStats}),
                b.h, length(ts), ts, length(tmb), tmb, opts, stats))
    return map(enumerate(simulation.glaciers)) do (i, g)
        tsi = collect(Float64, glacier_tstops(simulation, i))
        u = map(eachindex(tsi)) do j
            H = zeros(g.nx, g.ny)
            check(ccall((:odinn_get_snapshot, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Float64}), b.h, i - 1, j - 1, H))
            H
        end
        (; t = tsi, u = u, stats = stats[i])
    end
end
"current state of glacier i (H₀ before a solve, the last snapshot after it)"
function get_H(b::Batch, i::Integer, nx::Integer, ny::Integer)
    H = zeros(nx, ny)
    check(ccall((:odinn_get_H, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), b.h, i - 1, H)); H
end

"run!(prediction) on the device: the forward solve of every glacier, results built by Sleipnir as in inversion_utils.jl:541-546"
function run_HIP!(simulation; device = 0, law_kw...)
    b = Batch(simulation; device = device, law_kw...)
    sols = solve!(b, simulation)
    simulation.results = map(enumerate(sols)) do (i, sol)
        Sleipnir.create_results(simulation, i, sol, sol.t)      # reads sol.t / sol.u like an ODESolution
    end
    return simulation.results
end

"batch_loss_iceflow_transient (inversion_utils.jl:383-461) on the stored snapshots: loss per glacier"
function loss(b::Batch, G::Integer)
    l = zeros(G); check(ccall((:odinn_loss, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.h, l)); l
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

# surface-velocity seams  (Huginn.V_from_H, VJP_λ_∂surface_V∂H / ∂θ: VJPs.jl:61-82 -> adjoint.jl:268-413)
function surface_V_HIP(b::Batch, i::Integer, H::Matrix{Float64})
    Vx = similar(H); Vy = similar(H)
    check(ccall((:odinn_surface_V, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.h, i - 1, H, Vx, Vy))
    return Vx, Vy
end
function ODINN.VJP_λ_∂surface_V∂H(m::HIPVJP, λx, λy, H, θ, simulation, t)
    out = similar(H); i = simulation.cache.iceflow.glacier_idx
    check(ccall((:odinn_surface_V_vjp_H, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                m.batch.h, i - 1, λx, λy, H, out))
    return out
end
function ODINN.VJP_λ_∂surface_V∂θ(m::HIPVJP, λx, λy, H, θ, simulation, t)
    v = zeros(length(θ)); i = simulation.cache.iceflow.glacier_idx
    check(ccall((:odinn_surface_V_vjp_theta, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint),
                m.batch.h, i - 1, λx, λy, H, v, length(v)))
    return ODINN.Vector2ComponentVector(v, θ)
end
# the mass-balance step itself (mb_action! of inversion_utils.jl:501-510): H after the masked / clipped increment, and the increment
function mb_apply_HIP(b::Batch, i::Integer, H::Matrix{Float64})
    Hn = similar(H); MB = similar(H)
    check(ccall((:odinn_mb_apply, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.h, i - 1, H, Hn, MB))
    return Hn, MB
end
# loss / backward_loss(::TikhonovRegularization, a, Δx, Δy, mask) (Regularization.jl:92-126) on one field
function tikhonov_HIP(b::Batch, a::Matrix{Float64}, Δx::Real, Δy::Real, mask::Union{Nothing, Matrix{Bool}} = nothing)
    l = Ref(0.0); grad = similar(a)
    # (the UInt8 copy is a local that lives across the ccall: `pointer(convert(...))` of a temporary would dangle, and
    #  GC.@preserve of the Bool matrix does not root its converted copy)
    m8 = isnothing(mask) ? nothing : convert(Matrix{UInt8}, mask)
    GC.@preserve m8 check(ccall((:odinn_tikhonov, lib), Cint,
                (Ptr{Cvoid}, Cint, Cint, Cdouble, Cdouble, Ptr{Float64}, Ptr{UInt8}, Ptr{Float64}, Ptr{Float64}),
                b.h, size(a, 1), size(a, 2), Δx, Δy, a, isnothing(m8) ? Ptr{UInt8}(C_NULL) : pointer(m8), l, grad))
    return l[], grad
end
# per-glacier pieces of the last gradient call: θ.IC (gradient.jl:262-271,507-516) and the slots of a PerGlacierModel (Model.jl:208-224)
function lambda0(b::Batch, i::Integer, nx::Integer, ny::Integer)
    λ0 = zeros(nx, ny); check(ccall((:odinn_get_lambda0, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), b.h, i - 1, λ0)); λ0
end
function grad_parts(b::Batch, G::Integer)
    l = zeros(G); dA = zeros(G)
    check(ccall((:odinn_get_grad_parts, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), b.h, l, dA)); (l, dA)
end
function grad_field(b::Batch, i::Integer, nx::Integer, ny::Integer)
    g = zeros(nx - 1, ny - 1); check(ccall((:odinn_get_grad_field, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), b.h, i - 1, g)); g
end
function get_schedule(b::Batch)
    sc = Ref(Schedule(ntuple(_ -> Int32(-1), 19)..., ntuple(_ -> Int32(0), 1)))
    check(ccall((:odinn_get_schedule, lib), Cint, (Ptr{Cvoid}, Ptr{Schedule}), b.h, sc)); sc[]
end
# state of the Y law's table (set_schedule!(b; law_table = 1)): (usable, intervals, largest relative deviation from the network, Hbar range per glacier)
function law_table(b::Batch, n_glaciers::Integer)
    use = Ref{Cint}(0); ni = Ref{Cint}(0); dev = Ref{Float64}(0.0); hmax = zeros(n_glaciers)
    check(ccall((:odinn_get_law_table, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}), b.h, use, ni, dev, hmax))
    (usable = use[] != 0, n_intervals = Int(ni[]), max_rel_dev = dev[], hmax = hmax)
end
device_count() = (n = Ref{Cint}(0); check(ccall((:odinn_device_count, lib), Cint, (Ptr{Cint},), n)); Int(n[]))

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

comm_size(c::Comm) = (r = Ref{Cint}(0); n = Ref{Cint}(0);
                      check(ccall((:odinn_comm_rank, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}), c.h, r, n)); (Int(r[]), Int(n[])))
# the same reduction for a host vector (e.g. per-glacier slots of a PerGlacierModel summed over ranks)
allreduce_sum!(c::Comm, v::Vector{Float64}) =
    (check(ccall((:odinn_comm_allreduce_sum, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), c.h, v, length(v))); v)

# ---- seam 3 (the fast path): the whole gradient on the device ----------------------------
#      a new adjoint type next to DiscreteAdjoint / ContinuousAdjoint (src/inverse/AdjointTypes.jl:53-91);
#      SIA2D_grad! (gradient.jl:6-31) gains the method below: this rank's batch of glaciers is solved and
#      differentiated on its GPU, and ONE ncclAllReduce of [loss, dθ] inside the library replaces
#      pmap + sum(losses) + aggregate∇θ (gradient.jl:9-25, Model.jl:208-224).
struct HIPAdjoint{G <: ODINN.AbstractAdjointMethod} <: ODINN.AbstractAdjointMethod
    batch::Batch; comm::Union{Comm, Nothing}; method::G; VJP_method::HIPVJP
    function HIPAdjoint(batch::Batch, comm, method::G) where {G <: ODINN.AbstractAdjointMethod}
        set_vjp_method!(batch, method.VJP_method)   # DiscreteVJP (default) or ContinuousVJP stencil inside the reverse loops
        new{G}(batch, comm, method, HIPVJP(batch))
    end
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
