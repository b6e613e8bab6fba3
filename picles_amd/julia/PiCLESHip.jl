# PiCLESHip.jl — Julia-side binding of libpicles_hip.so (include/picles_hip.h).
#
# Drop-in for the 2D time step of PiCLES: a `WaveGrowth2DHIP <: Abstract2DModel` whose
# `time_step!`, `movie_time_step!`, `time_step!_advance`, `time_step!_remesh` and
# `init_particles!` methods forward to the HIP kernels through `ccall`, so `run!(sim)`
# (src/Simulations/run.jl:36-122) works unchanged.
#
# STATUS: Julia is not installed in the build container, so this file is syntax-reviewed only.
# The same C symbols are exercised end-to-end by the ctypes binding picles_amd/_capi.py
# (tests/test_gpu_*.py); the struct layouts below mirror include/picles_hip.h field by field.
module PiCLESHip

using PiCLES
using DifferentialEquations: DP5, Tsit5
using PiCLES.Architectures: Abstract2DModel
using PiCLES.Grids.CartesianGrid: TwoDCartesianGridMesh
using PiCLES.custom_structures: N_Periodic
import PiCLES.Operators.TimeSteppers: time_step!, movie_time_step!, time_step!_advance, time_step!_remesh
import PiCLES.Simulations: init_particles!

const libpicles = get(ENV, "PICLES_HIP_LIB", "libpicles_hip.so")

# ---- C structs (include/picles_hip.h) ----------------------------------------------------
struct picles_grid
    Nx::Int32; Ny::Int32
    dx::Float64; dy::Float64
    periodic_x::Int32; periodic_y::Int32
    mask::Ptr{Int8}
    j_begin::Int32; j_end::Int32
end

struct picles_phys
    r_g::Float64; C_alpha::Float64; C_phi::Float64; C_e::Float64; g::Float64
    gamma::Float64; q::Float64
    c_beta::Float64; c_D::Float64; c_e::Float64; c_alpha::Float64
    propagation::Int32; input::Int32; dissipation::Int32; peak_shift::Int32; direction::Int32
    _pad0::Int32
    dir_deadband::Float64
end

struct picles_ode
    abstol::Float64; reltol::Float64; dt0::Float64; dtmin::Float64
    force_dtmin::Int32; solver::Int32
    maxiters::Int64
    log_energy_minimum::Float64; log_energy_maximum::Float64; wind_min_squared::Float64
    timestep::Float64
end

struct picles_model
    periodic_boundary::Int32; init_type::Int32
    default_particle::NTuple{3,Float64}
    minimal_state::NTuple{2,Float64}
end

const STEP_ZERO_FIRST = Int32(1)
const STEP_MOVIE      = Int32(2)
const STEP_ATOMIC     = Int32(4)

# ---- the model type -----------------------------------------------------------------------
mutable struct WaveGrowth2DHIP{G,W,C} <: Abstract2DModel
    grid::G
    winds::W                      # (u = (x,y,t)->..., v = ...): sampled on the host, never called on device
    clock::C
    ODEsettings
    ODEdefaults
    minimal_state::Vector{Float64}
    periodic_boundary::Bool
    ocean_points::Vector
    State::Array{Float64,3}       # host mirror, refreshed after every step (what run! stores)
    MovieState::Union{Nothing,Array{Float64,3}}
    FailedCollection::Vector
    ctx::Ptr{Cvoid}
    mask::Matrix{Int8}
    winds_static::Bool
end

check(ctx, rc, what) = rc == 0 || error("$what failed (rc=$rc): " *
    unsafe_string(ccall((:picles_last_error, libpicles), Cstring, (Ptr{Cvoid},), ctx)))

"""
    WaveGrowth2DHIP(; grid, winds, ODEsys_kwargs, ODEsets, ...)

Same keywords as `WaveGrowth2D` (src/Models/WaveGrowthModels2D.jl:194-208); `ODEsys_kwargs` are the
keyword arguments that would go to `particle_equations` (γ, q, IDConstants, switches), because
the RHS itself runs inside the HIP kernel.
"""
function WaveGrowth2DHIP(; grid::TwoDCartesianGridMesh, winds, ODEsets, γ, q, IDConstants,
        propagation=true, input=true, dissipation=true, peak_shift=true, direction=true,
        ODEinit_type="wind_sea", minimal_state=nothing, periodic_boundary=true,
        clock, device::Integer=0, winds_static::Bool=false)
    st = grid.stats
    mask = Matrix{Int8}(grid.data.mask)
    ms = isnothing(minimal_state) ? PiCLES.FetchRelations.MinimalState(2, 2, ODEsets.timestep) : minimal_state
    P = ODEsets.Parameters
    per_y = st.Ny isa N_Periodic ? 1 : (nameof(typeof(st.Ny)) == :N_TripolarNorth ? 2 : 0)
    g = Ref(picles_grid(st.Nx.N, st.Ny.N, st.dx, st.dy, st.Nx isa N_Periodic, per_y,
                        pointer(mask), 0, st.Ny.N))
    p = Ref(picles_phys(P.r_g, P.C_α, P.C_φ, P.C_e, P.g, γ, q,
                        IDConstants.c_β, IDConstants.c_D, IDConstants.c_e, IDConstants.c_alpha,
                        propagation, input, dissipation, peak_shift, direction, 0, 0.0))
    solver_id = ODEsets.solver isa DP5 ? 0 : ODEsets.solver isa Tsit5 ? 1 : 2     # 2 = AutoTsit5(Rosenbrock23()), the default
    o = Ref(picles_ode(ODEsets.abstol, ODEsets.reltol, ODEsets.dt, ODEsets.dtmin, ODEsets.force_dtmin, solver_id,
                       ODEsets.maxiters, ODEsets.log_energy_minimum, ODEsets.log_energy_maximum,
                       ODEsets.wind_min_squared, ODEsets.timestep))
    fixed = !(ODEinit_type isa String)
    dp = fixed ? (ODEinit_type.lne, ODEinit_type.c̄_x, ODEinit_type.c̄_y) : (0.0, 0.0, 0.0)
    m = Ref(picles_model(periodic_boundary, fixed, dp, (ms[1], ms[2])))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve mask ccall((:picles_create, libpicles), Int32,
        (Ref{picles_grid}, Ref{picles_phys}, Ref{picles_ode}, Ref{picles_model}, Int32, Int32, Ref{Ptr{Cvoid}}),
        g, p, o, m, device, 1, ctx)
    rc == 0 || error("picles_create failed (rc=$rc): " *
        unsafe_string(ccall((:picles_last_error, libpicles), Cstring, (Ptr{Cvoid},), C_NULL)))
    Nx, Ny = st.Nx.N, st.Ny.N
    ocean = findall(mask .== 1)
    periodic_boundary && append!(ocean, findall(mask .== 3))
    model = WaveGrowth2DHIP(grid, winds, clock, ODEsets, fixed ? ODEinit_type : nothing, collect(Float64, ms),
        periodic_boundary, ocean, zeros(Nx, Ny, 3), nothing, [], ctx[], mask, winds_static)
    finalizer(mdl -> ccall((:picles_destroy, libpicles), Int32, (Ptr{Cvoid},), mdl.ctx), model)
    return model
end

# node-sample the wind closures for [t, t+Δt]; the only place user closures run
function upload_winds!(model::WaveGrowth2DHIP, t, Δt)
    x, y = model.grid.data.x, model.grid.data.y
    u0 = Float64[model.winds.u(x[I], y[I], t) for I in CartesianIndices(x)]
    v0 = Float64[model.winds.v(x[I], y[I], t) for I in CartesianIndices(x)]
    if model.winds_static
        rc = ccall((:picles_set_winds, libpicles), Int32,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64),
            model.ctx, u0, v0, t, C_NULL, C_NULL, t)
    else
        u1 = Float64[model.winds.u(x[I], y[I], t + Δt) for I in CartesianIndices(x)]
        v1 = Float64[model.winds.v(x[I], y[I], t + Δt) for I in CartesianIndices(x)]
        rc = ccall((:picles_set_winds, libpicles), Int32,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64),
            model.ctx, u0, v0, t, u1, v1, t + Δt)
    end
    check(model.ctx, rc, "picles_set_winds")
end

pull_state!(model) = check(model.ctx,
    ccall((:picles_get_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.State), "picles_get_state")

# ---- the drop-in methods --------------------------------------------------------------------
# init_particles!(model) — run.jl:199-247
function init_particles!(model::WaveGrowth2DHIP; defaults=nothing, verbose::Bool=false)
    upload_winds!(model, 0.0, model.ODEsettings.timestep)
    check(model.ctx, ccall((:picles_seed, libpicles), Int32, (Ptr{Cvoid}, Float64), model.ctx, model.clock.time), "picles_seed")
    pull_state!(model)
    nothing
end

# time_step!(model, Δt) — TimeSteppers.jl:109-166.  run! zeroes State before calling
# (run.jl:75-79); State lives on the device, so that zero-fill is requested with ZERO_FIRST when
# the host mirror is all zeros.
function time_step!(model::WaveGrowth2DHIP, Δt::Float64; callbacks=nothing, debug=false)
    upload_winds!(model, model.clock.time, Δt)
    flags = all(iszero, model.State) ? STEP_ZERO_FIRST : Int32(0)
    flags == 0 && check(model.ctx, ccall((:picles_set_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.State), "picles_set_state")
    check(model.ctx, ccall((:picles_time_step, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, flags), "picles_time_step")
    pull_state!(model)
    PiCLES.Operators.TimeSteppers.tick!(model.clock, Δt)
end

# movie_time_step!(model, Δt) — TimeSteppers.jl:212-247
function movie_time_step!(model::WaveGrowth2DHIP, Δt; callbacks=nothing, debug=false)
    upload_winds!(model, model.clock.time, Δt)
    check(model.ctx, ccall((:picles_time_step, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, STEP_MOVIE), "picles_time_step")
    model.MovieState === nothing && (model.MovieState = similar(model.State))
    check(model.ctx, ccall((:picles_get_movie_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.MovieState), "picles_get_movie_state")
    model.State .= 0.0
    PiCLES.Operators.TimeSteppers.tick!(model.clock, Δt)
end

# time_step!_advance / time_step!_remesh — TimeSteppers.jl:168-193
function time_step!_advance(model::WaveGrowth2DHIP, Δt::Float64, FailedCollection)
    upload_winds!(model, model.clock.time, Δt)
    check(model.ctx, ccall((:picles_set_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.State), "picles_set_state")
    check(model.ctx, ccall((:picles_advance, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, 0), "picles_advance")
    pull_state!(model)
end

function time_step!_remesh(model::WaveGrowth2DHIP, Δt::Float64)
    check(model.ctx, ccall((:picles_set_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.State), "picles_set_state")
    check(model.ctx, ccall((:picles_remesh, libpicles), Int32, (Ptr{Cvoid}, Float64), model.ctx, Δt), "picles_remesh")
end

end # module
