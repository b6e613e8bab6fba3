# PiCLESHip.jl — Julia-side binding of libpicles_hip.so (include/picles_hip.h).
#
# Drop-in for the 2D time step of PiCLES: a `WaveGrowth2DHIP <: Abstract2DModel` whose
# `time_step!`, `movie_time_step!`, `time_step!_advance`, `time_step!_remesh` and
# `init_particles!` methods forward to the HIP kernels through `ccall`, so `run!(sim)`
# (src/Simulations/run.jl:36-122) works unchanged.
#
# The field `State` is a `LazyState <: AbstractArray{Float64,3}`: the data lives in HBM, the host
# mirror is pulled only when somebody READS it, and the write `run!` performs every iteration —
# `sim.model.State .= 0.0` (run.jl:75-79) — is RECORDED and passed to the library as
# PICLES_STEP_ZERO_FIRST, where the zero-fill rides on the scatter's store.  An unobserved
# `State .= 0; time_step!` loop therefore stays on the fused one-launch-per-step path and moves
# nothing over PCIe; `push_state_to_storage!` / `copy(model.State)` pull once per stored step.
#
# STATUS: UNEXECUTED.  Julia is not installed in the build container or on the GPU boxes, so this file is
# syntax-reviewed only.  The same protocol is carried by picles_amd/models.py (`LazyState`),
# picles_amd/timesteppers.py and picles_amd/simulations.py, which ARE executed against the same C
# symbols (tests/test_gpu_lazy_state.py: an unobserved 10-step run! loop makes 10 fused launches,
# <= 1 stand-alone scatter and no State transfer).  Struct layouts mirror include/picles_hip.h
# field by field (ABI version 4; tests/test_capi_symbols.py holds the struct blocks below against the header).
module PiCLESHip

using PiCLES
using DifferentialEquations: DP5, Tsit5
using PiCLES.Architectures: Abstract2DModel
using PiCLES.Grids.CartesianGrid: TwoDCartesianGridMesh
using PiCLES.custom_structures: N_Periodic
import PiCLES.Operators.TimeSteppers: time_step!, movie_time_step!, time_step!_advance, time_step!_remesh
import PiCLES.Simulations: init_particles!

const libpicles = get(ENV, "PICLES_HIP_LIB", "libpicles_hip.so")
const PICLES_ABI_VERSION = Int32(4)

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

struct picles_counters
    rhs_evals::UInt64; steps_accepted::UInt64; steps_rejected::UInt64; reseeds::UInt64
    clamps::UInt64; maxiters_hits::UInt64; particles_advanced::UInt64; halo_overflow::UInt64
    max_reach::Int32; max_reach_seen::Int32
    dropped_nonfinite::UInt64
    wave_attempt_slots::UInt64
end

const STEP_ZERO_FIRST = Int32(1)
const STEP_MOVIE      = Int32(2)
const STEP_ATOMIC     = Int32(4)
# per-particle status bits (picles_get_particles)
const ST_MAXITERS, ST_RESEED_NAN, ST_RESEED_INF, ST_DTMIN, ST_NONFINITE = Int32(2), Int32(4), Int32(8), Int32(64), Int32(128)

check(ctx, rc, what) = rc == 0 || error("$what failed (rc=$rc): " *
    unsafe_string(ccall((:picles_last_error, libpicles), Cstring, (Ptr{Cvoid},), ctx)))

# ---- State: a lazy host view of the device field --------------------------------------------
"""
    LazyState

`model.State` as the rest of PiCLES sees it (`Array{Float64,3}(Nx, Ny, 3)` semantics) while the field lives in HBM.
`host_valid`: the host mirror equals the device field.  `dirty`: the mirror was written element-wise and must be
uploaded before the next step.  `zeroed`: the last write was `State .= 0` — nothing is executed, the next
`time_step!` passes `PICLES_STEP_ZERO_FIRST`.  (Python twin: picles_amd.models.LazyState.)
"""
mutable struct LazyState <: AbstractArray{Float64,3}
    host::Array{Float64,3}
    ctx::Ptr{Cvoid}
    host_valid::Bool
    dirty::Bool
    zeroed::Bool
    pulls::Int
    uploads::Int
end
LazyState(Nx::Integer, Ny::Integer) = LazyState(zeros(Nx, Ny, 3), C_NULL, false, false, false, 0, 0)

Base.size(s::LazyState) = size(s.host)
Base.IndexStyle(::Type{LazyState}) = IndexLinear()

function pull!(s::LazyState)
    if s.zeroed && !s.dirty
        s.host_valid || fill!(s.host, 0.0)
        s.host_valid = true
    elseif !s.host_valid && !s.dirty
        check(s.ctx, ccall((:picles_get_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.ctx, s.host), "picles_get_state")
        s.host_valid = true
        s.pulls += 1
    end
    return s.host
end

Base.getindex(s::LazyState, i::Int) = @inbounds pull!(s)[i]
function Base.setindex!(s::LazyState, v, i::Int)
    @inbounds pull!(s)[i] = v
    s.dirty = true; s.zeroed = false
    return v
end
# `State .= 0.0` lowers to copyto!(State, Broadcasted(identity, (0.0,))) and Base routes that to fill!(State, 0.0)
function Base.fill!(s::LazyState, x)
    if iszero(x)
        s.zeroed = true; s.dirty = false; s.host_valid = false
    else
        fill!(pull!(s), x)
        s.dirty = true; s.zeroed = false
    end
    return s
end
Base.copy(s::LazyState) = copy(pull!(s))
Base.Array(s::LazyState) = copy(pull!(s))

"called by the steppers: upload a written mirror; returns true if the step starts from a zeroed State"
function before_step!(s::LazyState)
    if s.dirty
        check(s.ctx, ccall((:picles_set_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.ctx, s.host), "picles_set_state")
        s.uploads += 1
        s.dirty = false
        return false
    end
    return s.zeroed
end
after_step!(s::LazyState) = (s.host_valid = false; s.zeroed = false; s.dirty = false; nothing)

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
    State::LazyState              # lazy host view of the device field (what run! zeroes and stores)
    MovieState::Union{Nothing,Array{Float64,3}}
    FailedCollection::Vector
    ctx::Ptr{Cvoid}
    mask::Matrix{Int8}
    winds_static::Bool
    winds_uploaded::Bool          # static winds: sampled and shipped once
    wind_level::Union{Nothing,Tuple{Float64,Matrix{Float64},Matrix{Float64}}}   # (t, u, v) sampled for the end of the last step
end

"""
    WaveGrowth2DHIP(; grid, winds, ODEsets, γ, q, IDConstants, ...)

Same keywords as `WaveGrowth2D` (src/Models/WaveGrowthModels2D.jl:194-208); `γ, q, IDConstants` and the five switches
are the keyword arguments that would go to `particle_equations`, because the RHS itself runs inside the HIP kernel.
`wind_lattice = (x, y, t, u, v)` hands gridded winds (`Utils/WindEmulator.jl:18-43`) to the device once
(`picles_set_wind_grid`): no wind data crosses PCIe inside the time loop.
"""
function WaveGrowth2DHIP(; grid::TwoDCartesianGridMesh, winds, ODEsets, γ, q, IDConstants,
        propagation=true, input=true, dissipation=true, peak_shift=true, direction=true,
        ODEinit_type="wind_sea", minimal_state=nothing, periodic_boundary=true,
        clock, device::Integer=0, winds_static::Bool=false, wind_lattice=nothing, wind_lattice_mode::Symbol=:linear)
    ccall((:picles_abi_version, libpicles), Int32, ()) == PICLES_ABI_VERSION ||
        error("libpicles_hip.so: ABI version mismatch (this binding expects $PICLES_ABI_VERSION)")
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
    state = LazyState(Nx, Ny)
    state.ctx = ctx[]
    model = WaveGrowth2DHIP(grid, winds, clock, ODEsets, fixed ? ODEinit_type : nothing, collect(Float64, ms),
        periodic_boundary, ocean, state, nothing, [], ctx[], mask, winds_static, false, nothing)
    if wind_lattice !== nothing
        x, y, t, u, v = wind_lattice            # regular knots; u, v :: Array{Float64,3}(nx, ny, nt)
        check(ctx[], ccall((:picles_set_wind_grid, libpicles), Int32,
            (Ptr{Cvoid}, Int32, Int32, Int32, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Float64}, Ptr{Float64}, Float64, Float64),
            ctx[], length(x), length(y), length(t), x[1], x[2] - x[1], y[1], y[2] - y[1], t[1], t[2] - t[1],
            u, v, grid.data.x[1, 1], grid.data.y[1, 1]), "picles_set_wind_grid")
        # time semantics inside a model step (include/picles_hip.h): :linear — the interpolant itself, a time knot inside the step is a
        # kink the RHS sees (what wind_interpolator's linear_interpolation gives the reference, Utils/WindEmulator.jl:18-43; a step
        # holding two or more knots is refused); :smooth3 — the lattice tabulates a smooth closure, three levels and the parabola
        wind_lattice_mode === :smooth3 && check(ctx[], ccall((:picles_set_wind_grid_mode, libpicles), Int32, (Ptr{Cvoid}, Int32),
            ctx[], Int32(1)), "picles_set_wind_grid_mode")
        model.winds_uploaded = true               # the device samples every step by itself
        model.winds_static = true
    end
    finalizer(mdl -> ccall((:picles_destroy, libpicles), Int32, (Ptr{Cvoid},), mdl.ctx), model)
    return model
end

# node-sample the wind closures for [t, t+Δt] by broadcast — the only place user closures run.  Static winds are
# shipped once; time-varying winds go as THREE levels (t, t+Δt/2, t+Δt: the kernels evaluate the parabola through them at
# every Runge-Kutta stage time, the stand-in for the reference calling u_wind(x,y,t) inside the RHS,
# particle_waves_v5.jl:494-495) and reuse the level sampled for the end of the previous step.
function upload_winds!(model::WaveGrowth2DHIP, t, Δt)
    model.winds_static && model.winds_uploaded && return
    x, y = model.grid.data.x, model.grid.data.y
    lvl = model.wind_level
    u0, v0 = (lvl !== nothing && lvl[1] == t) ? (lvl[2], lvl[3]) :
             (Matrix{Float64}(model.winds.u.(x, y, t)), Matrix{Float64}(model.winds.v.(x, y, t)))
    if model.winds_static
        rc = ccall((:picles_set_winds, libpicles), Int32,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Float64),
            model.ctx, u0, v0, t, C_NULL, C_NULL, t)
        model.winds_uploaded = true
    else
        um = Matrix{Float64}(model.winds.u.(x, y, t + Δt / 2))
        vm = Matrix{Float64}(model.winds.v.(x, y, t + Δt / 2))
        u1 = Matrix{Float64}(model.winds.u.(x, y, t + Δt))
        v1 = Matrix{Float64}(model.winds.v.(x, y, t + Δt))
        rc = ccall((:picles_set_winds3, libpicles), Int32,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64),
            model.ctx, u0, v0, t, um, vm, u1, v1, t + Δt)
        model.wind_level = (t + Δt, u1, v1)
    end
    check(model.ctx, rc, "picles_set_winds")
end

function counters(model::WaveGrowth2DHIP)
    c = Ref{picles_counters}()
    check(model.ctx, ccall((:picles_get_counters, libpicles), Int32, (Ptr{Cvoid}, Ref{picles_counters}), model.ctx, c), "picles_get_counters")
    return c[]
end

"""
diagnostic: `(busy, calm, order)` — the dispatch order the latest whole-grid fused step filed for its successor (the cost-ordered
dispatch of mixed calm / busy runs), or `nothing` when the run is not mixed
"""
function dispatch_order(model::WaveGrowth2DHIP)
    n = ccall((:picles_get_dispatch_order, libpicles), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int32), model.ctx, C_NULL, 0)
    n < 0 && check(model.ctx, n, "picles_get_dispatch_order")
    n == 0 && return nothing
    out = Vector{Int32}(undef, 2 + n)
    ccall((:picles_get_dispatch_order, libpicles), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int32), model.ctx, out, length(out)) == n || return nothing
    return (Int(out[1]), Int(out[2]), out[3:end])
end

"""
particles that could not be scattered: farther than the reach cap of the pull scatter (64 cells per model step for a
whole-grid context — a deliberate limit of this implementation, INTEGRATION.md; the reference wraps at any distance) or
with a non-finite position (the reference throws in `Int(floor(NaN))`).  They are counted, never silently lost.
"""
function check_dropped(model::WaveGrowth2DHIP)
    c = counters(model)
    (c.halo_overflow > 0 || c.dropped_nonfinite > 0) &&
        @warn "PiCLESHip: particles were NOT scattered" beyond_reach_cap=c.halo_overflow non_finite_position=c.dropped_nonfinite
    return c
end

"`debug=true` (TimeSteppers.jl:113-120, run.jl:84-92): particles whose last advance failed, from the per-particle status bits"
function collect_failed!(model::WaveGrowth2DHIP)
    N = length(model.mask)
    status = Vector{Int32}(undef, N)
    check(model.ctx, ccall((:picles_get_particles, libpicles), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Int32}), model.ctx, C_NULL, C_NULL, C_NULL, status), "picles_get_particles")
    bad = ST_MAXITERS | ST_DTMIN | ST_NONFINITE | ST_RESEED_NAN | ST_RESEED_INF
    model.FailedCollection = [(position_ij=Tuple(CartesianIndices(model.mask)[k]), status=status[k], time=model.clock.time)
                              for k in 1:N if (status[k] & bad) != 0]
    return model.FailedCollection
end

# ---- the drop-in methods --------------------------------------------------------------------
# init_particles!(model) — run.jl:199-247
function init_particles!(model::WaveGrowth2DHIP; defaults=nothing, verbose::Bool=false)
    model.winds_uploaded = model.winds_uploaded && model.winds_static && model.wind_level === nothing
    upload_winds!(model, 0.0, model.ODEsettings.timestep)
    check(model.ctx, ccall((:picles_seed, libpicles), Int32, (Ptr{Cvoid}, Float64), model.ctx, model.clock.time), "picles_seed")
    after_step!(model.State)
    nothing
end

# time_step!(model, Δt) — TimeSteppers.jl:109-166.  run! zeroes State before calling (run.jl:75-79); the lazy view
# has recorded that, and the zero-fill is requested with ZERO_FIRST (fused into the scatter's store).
function time_step!(model::WaveGrowth2DHIP, Δt::Float64; callbacks=nothing, debug=false)
    upload_winds!(model, model.clock.time, Δt)
    flags = before_step!(model.State) ? STEP_ZERO_FIRST : Int32(0)
    check(model.ctx, ccall((:picles_time_step, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, flags), "picles_time_step")
    after_step!(model.State)
    PiCLES.Operators.TimeSteppers.tick!(model.clock, Δt)
    if debug
        collect_failed!(model)
        check_dropped(model)
    end
    callbacks === nothing || callbacks(model)
    nothing
end

# movie_time_step!(model, Δt) — TimeSteppers.jl:212-247
function movie_time_step!(model::WaveGrowth2DHIP, Δt; callbacks=nothing, debug=false)
    upload_winds!(model, model.clock.time, Δt)
    before_step!(model.State) &&
        check(model.ctx, ccall((:picles_zero_state, libpicles), Int32, (Ptr{Cvoid},), model.ctx), "picles_zero_state")
    check(model.ctx, ccall((:picles_time_step, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, STEP_MOVIE), "picles_time_step")
    model.MovieState === nothing && (model.MovieState = Array{Float64,3}(undef, size(model.State)...))
    check(model.ctx, ccall((:picles_get_movie_state, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}), model.ctx, model.MovieState), "picles_get_movie_state")
    after_step!(model.State)                    # State is zero on the device after the remesh (:245)
    PiCLES.Operators.TimeSteppers.tick!(model.clock, Δt)
    debug && collect_failed!(model)
    nothing
end

# time_step!_advance / time_step!_remesh — TimeSteppers.jl:168-193
function time_step!_advance(model::WaveGrowth2DHIP, Δt::Float64, FailedCollection)
    upload_winds!(model, model.clock.time, Δt)
    before_step!(model.State) &&
        check(model.ctx, ccall((:picles_zero_state, libpicles), Int32, (Ptr{Cvoid},), model.ctx), "picles_zero_state")
    check(model.ctx, ccall((:picles_advance, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, 0), "picles_advance")
    after_step!(model.State)
end

function time_step!_remesh(model::WaveGrowth2DHIP, Δt::Float64)
    before_step!(model.State)                   # uploads a written mirror
    check(model.ctx, ccall((:picles_remesh, libpicles), Int32, (Ptr{Cvoid}, Float64), model.ctx, Δt), "picles_remesh")
end

"""
    run_stored!(model, Δt, n_steps, sink!; every=1, slots=3)

The stepping loop of `run!(sim; store=true)` (run.jl:72-114, storing.jl:109-119) with the device-side snapshot ring:
`picles_run_steps` enqueues `every` fused steps from C, `picles_store_push` snapshots State on the device and copies
it to pinned host memory on a side stream while the next steps run; `sink!(i, state::Array{Float64,3}, t)` receives the
snapshots in order (e.g. `(i, s, t) -> store["data"][i, :, :, :] = s`).
"""
function run_stored!(model::WaveGrowth2DHIP, Δt::Float64, n_steps::Integer, sink!; every::Integer=1, slots::Integer=3)
    upload_winds!(model, model.clock.time, Δt)
    model.winds_static || error("run_stored! needs winds the device can produce by itself (static winds or wind_lattice)")
    check(model.ctx, ccall((:picles_store_init, libpicles), Int32, (Ptr{Cvoid}, Int32), model.ctx, slots), "picles_store_init")
    buf = Array{Float64,3}(undef, size(model.State)...)
    t = Ref(0.0)
    i = 1
    drain_one() = begin
        check(model.ctx, ccall((:picles_store_pop, libpicles), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), model.ctx, buf, t), "picles_store_pop")
        sink!(i, buf, t[]); i += 1
    end
    done = 0
    while done < n_steps
        k = min(every, n_steps - done)
        check(model.ctx, ccall((:picles_run_steps, libpicles), Int32, (Ptr{Cvoid}, Float64, Int32), model.ctx, Δt, k), "picles_run_steps")
        for _ in 1:k
            PiCLES.Operators.TimeSteppers.tick!(model.clock, Δt)
        end
        done += k
        ccall((:picles_store_pending, libpicles), Int32, (Ptr{Cvoid},), model.ctx) == slots && drain_one()
        check(model.ctx, ccall((:picles_store_push, libpicles), Int32, (Ptr{Cvoid},), model.ctx), "picles_store_push")
    end
    while ccall((:picles_store_pending, libpicles), Int32, (Ptr{Cvoid},), model.ctx) > 0
        drain_one()
    end
    after_step!(model.State)
    check_dropped(model)
    nothing
end

end # module
