# OdinnHIP.jl -- the reference-side binding of libodinn_hip.so (see INTEGRATION.md).
#
# This is the file a maintainer would add to ODINN.jl (src/inverse/HIP/OdinnHIP.jl): one new VJP type
# and one new adjoint type whose methods ccall the C ABI of include/odinn_hip.h.  The build image has
# no Julia toolchain, so this file is NOT executed by the test-suite; the same entry points are
# exercised from Python (odinn.jl_amd/_lib.py) by tests/ -m gpu.
module OdinnHIP
using ODINN, Huginn, Sleipnir
const lib = "libodinn_hip"            # odinn.jl_amd/csrc/libodinn_hip.so on LD_LIBRARY_PATH

struct Phys;  rho::Float64; g::Float64; eta0::Float64; n::Float64; p::Float64; q::Float64
              C::Float64; minA::Float64; maxA::Float64; end
struct GlacierDesc; nx::Int32; ny::Int32; dx::Float64; dy::Float64; phys::Phys
                    A::Float64; T::Float64; end

check(rc) = rc == 0 || error(unsafe_string(ccall((:odinn_last_error, lib), Cstring, ())))

"One device context per simulation (all glaciers of this process on one GPU)."
mutable struct Batch; h::Ptr{Cvoid}; end
function Batch(simulation; device = 0)
    ph = simulation.parameters.physical
    descs = [GlacierDesc(g.nx, g.ny, g.Δx, g.Δy,
                 Phys(ph.ρ, ph.g, ph.η₀, g.n, 3.0, 0.0, g.C, ph.minA, ph.maxA), g.A, mean_temp(g))
             for g in simulation.glaciers]
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:odinn_batch_create, lib), Cint, (Cint, Cint, Ptr{GlacierDesc}, Ptr{Ptr{Cvoid}}),
                device, length(descs), descs, h))
    b = Batch(h[])
    for (i, g) in enumerate(simulation.glaciers)
        check(ccall((:odinn_set_fields, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}),
                    b.h, i - 1, g.H₀, g.B))
    end
    finalizer(x -> ccall((:odinn_batch_destroy, lib), Cint, (Ptr{Cvoid},), x.h), b)
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

# ---- seam 3 (the fast path): the whole per-glacier gradient on the device --------------
#      a new adjoint type next to DiscreteAdjoint (src/inverse/AdjointTypes.jl:85-91); the
#      branch `typeof(grad) <: HIPAdjoint` in SIA2D_grad_batch! (gradient.jl:129) becomes:
struct HIPAdjoint <: ODINN.AbstractAdjointMethod; batch::Batch; VJP_method::HIPVJP; end

function SIA2D_grad_batch_HIP!(θ, simulation, adj::HIPAdjoint, tstops, tstopsMB, solver)
    opts = Ref((solver.reltol, 1e-6, 0.0, 0.0, 0.0, Int64(solver.maxiters), Int32(0), Int32(0), 0.0))   # odinn_solver_opts
    loss = Ref(0.0); dθ = zeros(length(θ)); θv = ODINN.ComponentVector2Vector(θ)
    check(ccall((:odinn_loss_grad, lib), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Cvoid},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                adj.batch.h, θv, length(θv), length(tstops), tstops, length(tstopsMB), tstopsMB, opts,
                loss, dθ, C_NULL))
    return loss[], [ODINN.Vector2ComponentVector(dθ, θ)]     # same tuple SIA2D_grad_batch! returns (:554)
end

# the reference's DEFAULT gradient (ContinuousAdjoint, gradient.jl:276-539) on the device:
# reverse RDPK3Sp35 solve + Gauss-Legendre quadrature; `grad` supplies reltol/abstol/dtmax/n_quadrature
function SIA2D_grad_batch_HIP!(θ, simulation, adj::HIPAdjoint, grad::ODINN.ContinuousAdjoint, tstops, tstopsMB, solver)
    opts  = Ref((solver.reltol, 1e-6, 0.0, 0.0, 0.0, Int64(solver.maxiters), Int32(0), Int32(0), 0.0))  # odinn_solver_opts
    aopts = Ref((grad.reltol, grad.abstol, grad.dtmax, Int32(grad.n_quadrature), Int32(0),
                 Int64(solver.maxiters)))                                                    # odinn_adjoint_opts
    loss = Ref(0.0); dθ = zeros(length(θ)); θv = ODINN.ComponentVector2Vector(θ)
    check(ccall((:odinn_loss_grad_continuous, lib), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid},
                 Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}),
                adj.batch.h, θv, length(θv), length(tstops), tstops, length(tstopsMB), tstopsMB, opts, aopts,
                loss, dθ, C_NULL, C_NULL))
    return loss[], [ODINN.Vector2ComponentVector(dθ, θ)]
end
end # module
