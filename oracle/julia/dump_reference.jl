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

for case in sort(readdir(IO_DIR))
    isdir(joinpath(IO_DIR, case)) || continue
    try
        run_case(case)
    catch err
        @warn "case $(case) not dumped" err
    end
end
println("reference dump written to ", OUT_DIR)
