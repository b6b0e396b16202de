# dump_reference.jl -- pins the oracle to the REFERENCE's own numbers (SURVEY 8(c)(6)).
#
# NOT executed in the build image (no Julia toolchain there): run it once on any machine with Julia and ODINN.jl
# installed, commit what it writes under tests/golden/reference_dump/, and tests/test_oracle_vs_reference_dump.py
# turns from "xfail: no reference dump present" into a hard 1e-12 comparison of oracle/sia2d_oracle.py with the
# Julia reference on the golden cases of tests/golden/make_golden.py.
#
#   python oracle/julia/export_inputs.py            # raw Float64 inputs -> oracle/julia/io/<case>/
#   julia --project=<ODINN.jl checkout> oracle/julia/dump_reference.jl
#
# Own code over the reference's PUBLIC API only, modelled on how its tests drive one RHS / VJP evaluation
# (test/SIA2D_adjoint.jl:98-137) on a synthetic Glacier2D (test/test_grad_loss.jl:588-600).  For every case it writes
#   dH.f64         Huginn.SIA2D!(dH, H, simulation, t, θ)
#   vjp_H.f64      VJP_λ_∂SIA∂H(DiscreteVJP(), λ, H, θ, simulation, t)[1]
#   vjp_theta.f64  VJP_λ_∂SIA∂θ(DiscreteVJP(), λ, H, θ, nothing, simulation, t)   (flattened like θ)
# and, for the A-type cases (target :A), the surface-velocity seam with the cotangents (∂Vx, ∂Vy) = (λ, reverse(λ, dims = 1)):
#   Vx.f64, Vy.f64            Huginn.V_from_H(simulation, H, t, θ)
#   vjp_surfV_H.f64           VJP_λ_∂surface_V∂H(DiscreteVJP(), ∂Vx, ∂Vy, H, θ, simulation, t)[1]
#   vjp_surfV_theta.f64       VJP_λ_∂surface_V∂θ(DiscreteVJP(), ∂Vx, ∂Vy, H, θ, simulation, t)[1]
# as raw little-endian Float64, column-major -- the layout of include/odinn_hip.h.
#
# ONE run pins the whole path: besides the single-evaluation seams above, run_solve_case() dumps a forward solve (snapshots,
# every accepted step time, step counts: the integrator + PID controller), the loss and dθ of SIA2D_grad! through the
# DiscreteAdjoint and the ContinuousAdjoint, and run_pieces() the out-of-tree helpers the oracle had to define itself
# (is_in_glacier, create_interpolation's knots, the mass-balance mask / clip).
using ODINN, Huginn, Sleipnir

const HERE = @__DIR__
const IO_DIR = joinpath(HERE, "io")
const OUT_DIR = normpath(joinpath(HERE, "..", "..", "tests", "golden", "reference_dump"))

readf64(path, dims...) = reshape(reinterpret(Float64, read(path)), dims...) |> collect
function readmeta(path)
    d = Dict{String, String}()
    for l in eachline(path)
        k, v = split(l, ' '; limit = 2)
        d[k] = v
    end
    return d
end

function run_case(case::String)
    dir = joinpath(IO_DIR, case)
    m = readmeta(joinpath(dir, "meta.txt"))
    nx, ny = parse(Int, m["nx"]), parse(Int, m["ny"])
    Δx, Δy = parse(Float64, m["dx"]), parse(Float64, m["dy"])
    H = readf64(joinpath(dir, "H.f64"), nx, ny)
    B = readf64(joinpath(dir, "B.f64"), nx, ny)
    λ = readf64(joinpath(dir, "lam.f64"), nx, ny)
    θin = readf64(joinpath(dir, "theta.f64"), parse(Int, m["P"]))
    law = m["law"]
    T = parse(Float64, m["T"]); A = parse(Float64, m["A"]); C = parse(Float64, m["C"])
    target = law == "nnY" ? :D_hybrid : law == "nnU" ? :D : :A
    grad = DiscreteAdjoint(VJP_method = DiscreteVJP())
    params = Parameters(
        simulation = SimulationParameters(tspan = (2010.0, 2011.0), multiprocessing = false, use_MB = false,
            use_iceflow = true, test_mode = false, working_dir = Huginn.root_dir),
        physical = PhysicalParameters(ρ = parse(Float64, m["rho"]), g = parse(Float64, m["g"]),
            η₀ = parse(Float64, m["eta0"]), minA = parse(Float64, m["minA"]), maxA = parse(Float64, m["maxA"])),
        UDE = UDEparameters(optim_autoAD = ODINN.NoAD(), grad = grad, optimization_method = "AD+AD", target = target),
        solver = Huginn.SolverParameters(step = 1 / 12))
    climate = Sleipnir.DummyClimate2D(longterm_temps_scalar = [T], longterm_temps_gridded = fill(T, nx - 1, ny - 1))
    glacier = Glacier2D(rgi_id = "golden-" * case, climate = climate, H₀ = H, S = B .+ max.(H, 0.0), B = B, A = A,
        n = parse(Float64, m["n"]), Δx = Δx, Δy = Δy, nx = nx, ny = ny, C = C)
    glaciers = Vector{Sleipnir.AbstractGlacier}([glacier])
    model = if law == "constA"
        # a constant A is not trainable in the reference: it is reached through the classical per-glacier law, whose
        # single parameter maps to A by the tanh rule of Laws.jl:402-460 -- dθ is converted back to d/dA below
        reg = GlacierWideInv(params, glaciers, :A)
        Model(iceflow = SIA2Dmodel(params; A = LawA(params; scalar = true)), mass_balance = nothing, regressors = (; A = reg))
    elseif law == "nnA_scalar"
        nn = NeuralNetwork(params)
        Model(iceflow = SIA2Dmodel(params; A = LawA(nn, params; precompute_VJPs = false, scalar = true)),
            mass_balance = nothing, regressors = (; A = nn))
    elseif law == "nnY"
        nn = NeuralNetwork(params)
        Model(iceflow = SIA2Dmodel(params; Y = LawY(nn, params)), mass_balance = nothing, regressors = (; A = nn))
    else
        error("case $(case): law $(law) is not dumped")
    end
    simulation = Inversion(model, glaciers, params)
    θ = simulation.model.trainable_components.θ
    if law != "constA"
        # the oracle's flattening ([vec(W) column-major, b] per layer) against Lux/ComponentArrays' own: this line is
        # the check of SURVEY Appendix B's open item
        @assert length(θ) == length(θin) "θ length: reference $(length(θ)), oracle $(length(θin))"
        θ = ODINN.Vector2ComponentVector(θin, θ)
        simulation.model.trainable_components.θ = θ
    end
    t = params.simulation.tspan[1]
    simulation.cache = init_cache(model, simulation, 1, θ)
    apply_all_callback_laws!(model.iceflow, simulation.cache.iceflow, simulation, 1, t, θ)
    dH = zero(H)
    Huginn.SIA2D!(dH, H, simulation, t, θ)
    Huginn.precompute_all_VJPs_laws!(model.iceflow, simulation.cache.iceflow, simulation, 1, t, θ)
    ∂H, = ODINN.VJP_λ_∂SIA∂H(grad.VJP_method, λ, H, θ, simulation, t)
    ∂θ = ODINN.VJP_λ_∂SIA∂θ(grad.VJP_method, λ, H, θ, nothing, simulation, t)
    ∂θv = ODINN.ComponentVector2Vector(∂θ)
    if law == "constA"   # dL/dθ = dL/dA · dA/dθ with A = minA + (maxA - minA)(tanh θ + 1)/2  ->  report dL/dA like the oracle
        lo, hi = params.physical.minA, params.physical.maxA
        ∂θv = ∂θv ./ ((hi - lo) / 2 .* (1 .- tanh.(ODINN.ComponentVector2Vector(θ)) .^ 2))
    end
    out = joinpath(OUT_DIR, case)
    mkpath(out)
    write(joinpath(out, "dH.f64"), vec(dH))
    write(joinpath(out, "vjp_H.f64"), vec(∂H))
    write(joinpath(out, "vjp_theta.f64"), ∂θv)
    if target == :A   # surface-velocity seam (adjoint.jl:268-413)
        ∂Vx, ∂Vy = λ, reverse(λ, dims = 1)
        Vx, Vy, _ = Huginn.V_from_H(simulation, H, t, θ)
        ∂HV, = ODINN.VJP_λ_∂surface_V∂H(grad.VJP_method, ∂Vx, ∂Vy, H, θ, simulation, t)
        ∂θV, = ODINN.VJP_λ_∂surface_V∂θ(grad.VJP_method, ∂Vx, ∂Vy, H, θ, simulation, t)
        ∂θVv = ODINN.ComponentVector2Vector(∂θV)
        if law == "constA"
            lo, hi = params.physical.minA, params.physical.maxA
            ∂θVv = ∂θVv ./ ((hi - lo) / 2 .* (1 .- tanh.(ODINN.ComponentVector2Vector(θ)) .^ 2))
        end
        write(joinpath(out, "Vx.f64"), vec(Vx)); write(joinpath(out, "Vy.f64"), vec(Vy))
        write(joinpath(out, "vjp_surfV_H.f64"), vec(∂HV)); write(joinpath(out, "vjp_surfV_theta.f64"), ∂θVv)
    end
    println(case, ": ‖dH‖ = ", sqrt(sum(abs2, dH)), "  ‖∂H‖ = ", sqrt(sum(abs2, ∂H)), "  ‖∂θ‖ = ", sqrt(sum(abs2, ∂θv)))
end

# ---------------------------------------------------------------------------------------------------------------------
# The WHOLE path on one case (inputs: oracle/julia/io/solve_valley_nnA/, = tests/golden/solve_valley_nnA.npz):
#   forward solve as _batch_iceflow_UDE runs it (inversion_utils.jl:472-572): RDPK3Sp35, reltol 1e-8, tstops = the data times
#       fwd_H_<j>.f64   result.H[j]        fwd_t.f64   result.t
#       fwd_steps.f64   [naccept, nreject] of the ODE solution, fwd_sol_t.f64 every accepted step time (pins OrdinaryDiffEq's
#                       PID controller and initial-step heuristic, SURVEY App. B)
#   SIA2D_grad! (gradient.jl:6-31) with DiscreteAdjoint and with ContinuousAdjoint(n_quadrature = meta), DiscreteVJP both:
#       grad_discrete.f64 / grad_continuous.f64   [loss, dθ...]   (loss = loss_iceflow_transient(θ, simulation, map))
function run_solve_case()
    case = "solve_valley_nnA"
    dir = joinpath(IO_DIR, case)
    isdir(dir) || return
    m = readmeta(joinpath(dir, "meta.txt"))
    nx, ny, k = parse(Int, m["nx"]), parse(Int, m["ny"]), parse(Int, m["k"])
    Δx, Δy, T = parse(Float64, m["dx"]), parse(Float64, m["dy"]), parse(Float64, m["T"])
    H₀ = readf64(joinpath(dir, "H0.f64"), nx, ny)
    B = readf64(joinpath(dir, "B.f64"), nx, ny)
    ts = readf64(joinpath(dir, "ts.f64"), k)
    θin = readf64(joinpath(dir, "theta.f64"), parse(Int, m["P"]))
    Href = [readf64(joinpath(dir, "ref_$(j - 1).f64"), nx, ny) for j in 1:k]
    nq = parse(Int, m["n_quadrature"])
    out = joinpath(OUT_DIR, case)
    mkpath(out)
    for (name, grad) in (("discrete", DiscreteAdjoint(VJP_method = DiscreteVJP())),
                         ("continuous", ContinuousAdjoint(VJP_method = DiscreteVJP(), n_quadrature = nq, MB_VJP = DiscreteVJP())))
        params = Parameters(
            simulation = SimulationParameters(tspan = (ts[1], ts[end]), multiprocessing = false, use_MB = false,
                use_iceflow = true, test_mode = false, working_dir = Huginn.root_dir),
            physical = PhysicalParameters(ρ = parse(Float64, m["rho"]), g = parse(Float64, m["g"]), η₀ = parse(Float64, m["eta0"]),
                minA = parse(Float64, m["minA"]), maxA = parse(Float64, m["maxA"])),
            UDE = UDEparameters(optim_autoAD = ODINN.NoAD(), grad = grad, optimization_method = "AD+AD", target = :A,
                empirical_loss_function = LossH(L2Sum(distance = parse(Int, m["distance"])))),
            # step = the whole span: the `step` grid adds only the two ends, the stops are the thickness-data times
            solver = Huginn.SolverParameters(step = ts[end] - ts[1], reltol = parse(Float64, m["reltol"])))
        climate = Sleipnir.DummyClimate2D(longterm_temps_scalar = [T], longterm_temps_gridded = fill(T, nx - 1, ny - 1))
        glacier = Glacier2D(rgi_id = "golden-" * case, climate = climate, H₀ = H₀, S = B .+ H₀, B = B, A = parse(Float64, m["A"]),
            n = parse(Float64, m["n"]), Δx = Δx, Δy = Δy, nx = nx, ny = ny, C = parse(Float64, m["C"]),
            thicknessData = Sleipnir.ThicknessData(ts, Href))
        glaciers = Vector{Sleipnir.AbstractGlacier}([glacier])
        nn = NeuralNetwork(params)
        model = Model(iceflow = SIA2Dmodel(params; A = LawA(nn, params; scalar = true)), mass_balance = nothing, regressors = (; A = nn))
        simulation = Inversion(model, glaciers, params)
        θ = ODINN.Vector2ComponentVector(θin, simulation.model.trainable_components.θ)
        simulation.model.trainable_components.θ = θ
        if name == "discrete"   # the forward solve, once
            container = ODINN.InversionBinder(simulation, θ)
            simulation.cache = init_cache(model, simulation, 1, θ)
            prob = ODINN.define_iceflow_prob(θ, simulation, 1)
            sol = ODINN.simulate_iceflow_UDE!(container, ODINN.CallbackSet(), prob, ts)
            write(joinpath(out, "fwd_steps.f64"), Float64[sol.stats.naccept, sol.stats.nreject])
            write(joinpath(out, "fwd_sol_t.f64"), collect(Float64, sol.t))
            result = ODINN._batch_iceflow_UDE(container, 1, prob)
            write(joinpath(out, "fwd_t.f64"), collect(Float64, result.t))
            for j in 1:length(result.H)
                write(joinpath(out, "fwd_H_$(j - 1).f64"), vec(result.H[j]))
            end
        end
        loss = ODINN.loss_iceflow_transient(θ, simulation, map)
        dθ = zero(θ)
        SIA2D_grad!(dθ, θ, simulation)
        write(joinpath(out, "grad_$(name).f64"), vcat([loss], ODINN.ComponentVector2Vector(dθ)))
        println(case, " ", name, ": loss = ", loss, "  ‖dθ‖ = ", sqrt(sum(abs2, dθ)))
    end
end

# ---------------------------------------------------------------------------------------------------------------------
# Out-of-tree pieces the oracle had to define itself (inputs: oracle/julia/io/pieces/):
#   mask_a.f64, mask_b.f64   Sleipnir's is_in_glacier(H, distance) as 0 / 1 doubles   (Losses.jl:122,266)
#   knots.f64                ODINN.create_interpolation(avg(max.(H_a, 0)); n_interp_half)   (target_utils.jl:245-293)
#   H_after_mb.f64           the state after apply_MB_mask! with the given MB field (the mask / clip of the mass-balance
#                            callback, inversion_utils.jl:505-507; VJPs.jl:129-139 mirrors it), mb_applied.f64 = its increment
function run_pieces()
    dir = joinpath(IO_DIR, "pieces")
    isdir(dir) || return
    m = readmeta(joinpath(dir, "meta.txt"))
    nx, ny, nxb, nyb = parse(Int, m["nx"]), parse(Int, m["ny"]), parse(Int, m["nxb"]), parse(Int, m["nyb"])
    d = parse(Int, m["distance"])
    out = joinpath(OUT_DIR, "pieces")
    mkpath(out)
    Ha = readf64(joinpath(dir, "H_a.f64"), nx, ny)
    Hb = readf64(joinpath(dir, "H_b.f64"), nxb, nyb)
    write(joinpath(out, "mask_a.f64"), Float64.(vec(ODINN.is_in_glacier(Ha, d))))
    write(joinpath(out, "mask_b.f64"), Float64.(vec(ODINN.is_in_glacier(Hb, d))))
    H̄ = Huginn.avg(max.(Ha, 0.0))
    write(joinpath(out, "knots.f64"), collect(Float64, ODINN.create_interpolation(H̄; n_interp_half = parse(Int, m["n_interp_half"]))))
    # mass-balance mask / clip on a prescribed MB field: cache.iceflow.MB is what MB_timestep! would have filled
    MB = readf64(joinpath(dir, "MB.f64"), nx, ny)
    H = readf64(joinpath(dir, "Hmb.f64"), nx, ny)
    params = Parameters(simulation = SimulationParameters(tspan = (2010.0, 2011.0), multiprocessing = false, use_MB = false,
                            use_iceflow = true, test_mode = false, working_dir = Huginn.root_dir),
                        solver = Huginn.SolverParameters(step = 1 / 12))
    climate = Sleipnir.DummyClimate2D(longterm_temps_scalar = [-2.0], longterm_temps_gridded = fill(-2.0, nx - 1, ny - 1))
    glacier = Glacier2D(rgi_id = "golden-pieces", climate = climate, H₀ = H, S = copy(H), B = zero(H), A = 2.21e-18, n = 3.0,
        Δx = 50.0, Δy = 50.0, nx = nx, ny = ny, C = 0.0)
    model = Model(iceflow = SIA2Dmodel(params), mass_balance = nothing)
    simulation = Prediction(model, Vector{Sleipnir.AbstractGlacier}([glacier]), params)
    cache = init_cache(model, simulation, 1, nothing)
    cache.iceflow.MB .= MB
    Hn = copy(H)
    Huginn.apply_MB_mask!(Hn, cache.iceflow)
    write(joinpath(out, "H_after_mb.f64"), vec(Hn))
    write(joinpath(out, "mb_applied.f64"), vec(Hn .- H))
    println("pieces: |mask_a| = ", sum(ODINN.is_in_glacier(Ha, d)), "  |mask_b| = ", sum(ODINN.is_in_glacier(Hb, d)))
end

for case in sort(readdir(IO_DIR))
    (isdir(joinpath(IO_DIR, case)) && isfile(joinpath(IO_DIR, case, "lam.f64"))) || continue
    try
        run_case(case)
    catch err
        @warn "case $(case) not dumped" err
    end
end
for (name, f) in (("solve_valley_nnA", run_solve_case), ("pieces", run_pieces))
    try
        f()
    catch err
        @warn "$(name) not dumped" err
    end
end
println("reference dump written to ", OUT_DIR)
