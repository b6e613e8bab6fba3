/*
 * picles_hip.h — C ABI of the MI355X-native PiCLES 2D particle-in-cell time step.
 *
 * This header IS the drop-in boundary.  The reference (mochell/PiCLES, pure Julia)
 * has no FFI layer of its own; the functions a replacement must provide are the
 * Julia methods of Operators.TimeSteppers / Simulations listed next to each entry
 * point below (file:line are into the reference tree).  A Julia maintainer binds
 * them with `ccall`; the in-repo Python host layer
 * (picles_amd/) binds the very same symbols with ctypes.
 *
 * Conventions
 *   - every function returns int32: 0 = OK, <0 = error (text via picles_last_error)
 *   - all arrays handed across the ABI are caller-owned HOST memory unless the
 *     name says `_dev`; the library never frees caller memory and never calls back
 *   - fields are column-major [i + Nx*(j + Ny*k)] exactly like the reference's
 *     State[Nx,Ny,3] (k = e, m_x, m_y)   (WaveGrowthModels2D.jl:136-143)
 *   - a context belongs to ONE GPU (one process per GPU) and owns the rows
 *     [j_begin, j_end) of the global grid (y-slab); single-GPU use: 0, Ny
 *   - a context is not re-entrant (the reference calls time_step! from one task)
 *   - there is NO CPU fallback: creating a context without a HIP device fails
 */
#ifndef PICLES_HIP_H
#define PICLES_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PICLES_ABI_VERSION 4

/* ---- grid: TwoDCartesianGridStatistics + mesh mask (Grids/CartesianGrid.jl:26-101,
 *      Grids/mask_utils.jl:38-55) ------------------------------------------------ */
typedef struct picles_grid {
    int32_t Nx, Ny;          /* global node counts                                   */
    double  dx, dy;          /* node spacing [m]; ProjetionKernel M = diag(1/dx,1/dy) */
    int32_t periodic_x;      /* 1: Nx isa N_Periodic, 0: N_NonPeriodic (scatter wrap/drop) */
    int32_t periodic_y;      /* 0 / 1 likewise; 2: Ny isa N_TripolarNorth — open at the south edge, corners beyond the
                                north edge fold back mirrored in x (ParticleInCell.jl:353-361, 409-428); needs periodic_x */
    const int8_t *mask;      /* Nx*Ny total mask {0 land,1 ocean,2 land bnd,3 grid bnd};
                                NULL => all ocean + make_boundaries() ring on non-periodic axes */
    int32_t j_begin, j_end;  /* rows owned by this context (slab); 0, Ny for one device */
} picles_grid;

/* ---- physics: ODEParameters NamedTuple + particle_equations kwargs
 *      (particle_waves_v5.jl:107-128,154-162,184-196,382-395) -------------------- */
typedef struct picles_phys {
    double r_g, C_alpha, C_phi, C_e, g;   /* params NamedTuple (g is carried but, as in the
                                             reference :281-287,504, NOT used by the RHS)   */
    double gamma, q;                      /* particle_equations(u,v; γ, q)                  */
    double c_beta, c_D, c_e, c_alpha;     /* IDConstants fields entering e_T_func :271      */
    int32_t propagation, input, dissipation, peak_shift, direction;  /* RHS switches :383-387 */
    int32_t _pad0;
    double  dir_deadband;   /* OPT-IN, 0 = off (reference-exact).  |sin(θ_c - θ_w)| below this value is
                               treated as exact alignment in S_dir (particle_waves_v5.jl:345-346): round-off
                               misalignment (~1e-16) otherwise excites the stiff direction mode of the
                               explicit stepper.  1e-9 changes the RHS by < 1e-9 relative. */
} picles_phys;

/* ---- ODESettings (particle_waves_v5.jl:34-75) ---------------------------------- */
typedef struct picles_ode {
    double  abstol, reltol;
    double  dt0;             /* ODESettings.dt : initial dt of a freshly built integrator   */
    double  dtmin;
    int32_t force_dtmin;
    int32_t solver;          /* 0 = DP5 (Dormand-Prince 5(4), bench06:93); 1 = Tsit5 (the explicit
                                half of the ODESettings default); 2 = AutoTsit5(Rosenbrock23()),
                                the ODESettings default itself (particle_waves_v5.jl:47): Tsit5 with
                                the stiffness test of OrdinaryDiffEq's AutoSwitch and a Rosenbrock23
                                fallback on an exact hand-written Jacobian (restated, unpinned)   */
    int64_t maxiters;        /* per model step                        */
    double  log_energy_minimum, log_energy_maximum, wind_min_squared;
    double  timestep;        /* ODESettings.timestep: seed time-scale of init_particles!    */
} picles_ode;

/* ---- model flags (WaveGrowthModels2D.jl:194-345) ------------------------------- */
typedef struct picles_model {
    int32_t periodic_boundary;   /* model flag: selects ocean_points / boundary flag        */
    int32_t init_type;           /* 0 "wind_sea" (ODEdefaults = nothing); 1 fixed ParticleDefaults */
    double  default_particle[3]; /* lne, c̄_x, c̄_y when init_type == 1                      */
    double  minimal_state[2];    /* [E_min, m²_min]  (FetchRelations.MinimalState)          */
} picles_model;

/* flags of picles_time_step */
#define PICLES_STEP_ZERO_FIRST   1  /* run!: State .= 0 before time_step!  (run.jl:75-82)    */
#define PICLES_STEP_MOVIE        2  /* movie_time_step!: snapshot State->MovieState between
                                       advance and remesh, zero State after  (TimeSteppers.jl:212-247) */
#define PICLES_STEP_ATOMIC       4  /* scatter with the LDS-tiled fp64-atomic push instead of the
                                       deterministic (bitwise reproducible) pull                 */

/* per-particle status bits (returned by picles_get_particles) */
#define PICLES_ST_STEPPED        1  /* particle is in ocean_points                              */
#define PICLES_ST_MAXITERS       2  /* internal RK loop hit maxiters in the last advance        */
#define PICLES_ST_RESEED_NAN     4  /* NaN guard re-seeded (mapping_2D.jl:196-211)              */
#define PICLES_ST_RESEED_INF     8  /* Inf guard re-seeded (:213-222)                           */
#define PICLES_ST_CLAMPED       16  /* lne clamped to log_energy_maximum (:224-235)             */
#define PICLES_ST_SWITCHED_ON   32  /* off -> on by wind test in advance (:172-185)             */
#define PICLES_ST_DTMIN         64  /* stopped: dt <= dtmin without force_dtmin                 */
#define PICLES_ST_NONFINITE    128  /* a non-finite error estimate was rejected                 */

typedef struct picles_counters {
    uint64_t rhs_evals;       /* RHS evaluations                      */
    uint64_t steps_accepted;  /* accepted internal RK steps           */
    uint64_t steps_rejected;
    uint64_t reseeds;         /* NaN/Inf guards + off->on + remesh B/C */
    uint64_t clamps;
    uint64_t maxiters_hits;
    uint64_t particles_advanced;  /* particles that ran the ODE       */
    uint64_t halo_overflow;   /* particles that travelled beyond the scatter reach the context covers — halo_rows for a
                                 slab, 64 cells per model step for a whole-grid context — and were NOT scattered */
    int32_t  max_reach;       /* max |cell offset| any scatter corner had in the last advance */
    int32_t  max_reach_seen;  /* the largest max_reach since picles_seed / picles_reset_counters (slab halos are sized from it) */
    uint64_t dropped_nonfinite; /* switched-on particles whose advanced position was NaN / Inf: not scattered.  The reference
                                 would throw in Int(floor(NaN)) (ParticleInCell.jl:58-71); here they are dropped and counted */
    uint64_t wave_attempt_slots; /* sum over wavefronts of 64 x (the largest number of internal RK attempts any lane of the wave
                                 made): (steps_accepted + steps_rejected) / wave_attempt_slots is the lane efficiency of the
                                 adaptive advance — 1 when all particles of a wave take the same number of attempts */
} picles_counters;

typedef struct picles_timing {     /* accumulated device time, ms (HIP events on the compute stream) */
    double advance_ms, scatter_ms, remesh_ms, other_ms;
    uint64_t advance_launches, scatter_launches, remesh_launches;
} picles_timing;

typedef struct picles_ctx picles_ctx;

/* WaveGrowth2D(...) constructor: allocates State, particles (WaveGrowthModels2D.jl:194-345).
 * device_id: HIP device ordinal.  halo_rows: ghost rows kept on each side of the slab
 * (>= the scatter reach in y; ignored for a single slab covering [0,Ny) ). */
int32_t picles_create(const picles_grid *g, const picles_phys *p, const picles_ode *o,
                      const picles_model *m, int32_t device_id, int32_t halo_rows,
                      picles_ctx **out);
int32_t picles_destroy(picles_ctx *ctx);
const char *picles_last_error(const picles_ctx *ctx);   /* ctx may be NULL: last create() error */
int32_t picles_abi_version(void);

/* winds: node-sampled u,v (col-major, the context's own rows only: Nx*(j_end-j_begin))
 * at two time levels; the RHS lerps in t.  u1==NULL => time-constant.
 * Replaces the Julia closures winds.u(x,y,t) (particle_waves_v5.jl:494-495). */
int32_t picles_set_winds(picles_ctx *ctx, const double *u0, const double *v0, double t0,
                         const double *u1, const double *v1, double t1);
/* The same with a third level (um, vm) at the middle of the window, (t0 + t1)/2: the RHS evaluates the parabola through the
 * three levels.  For closures that are not linear in t inside a model step (tests/T04_2D_reg_test.jl:166-167 multiplies u by
 * cos(3t/(3600 2π))) the two-level form misses the reference's u_wind(x,y,t) at the stage times (particle_waves_v5.jl:494-495)
 * by (ω Δt)²/8 of the amplitude, this one by (ω Δt)³/125.  um == NULL is picles_set_winds. */
int32_t picles_set_winds3(picles_ctx *ctx, const double *u0, const double *v0, double t0,
                          const double *um, const double *vm,
                          const double *u1, const double *v1, double t1);
/* Three levels with the middle one at a KNOT of a gridded wind, t0 < tk < t1: the RHS evaluates the two straight segments
 * (t0,u0)-(tk,uk)-(t1,u1).  This is the exact form, inside one model step, of wind_interpolator's
 * linear_interpolation((x,y,t), u) (Utils/WindEmulator.jl:18-43) when one time knot of the lattice falls inside the step: the
 * reference's RHS evaluates that interpolant at every stage time (particle_waves_v5.jl:494-495), so the solver sees the kink.
 * A window without an interior knot is picles_set_winds; one with two or more is picles_set_winds_polyline. */
int32_t picles_set_winds_knot(picles_ctx *ctx, const double *u0, const double *v0, double t0,
                              const double *uk, const double *vk, double tk,
                              const double *u1, const double *v1, double t1);
/* The general form: nlev >= 2 node-sampled levels (u[k], v[k]) at strictly increasing times[k], times[0] and times[nlev-1] the
 * ends of the window — the piecewise-linear wind through them, i.e. the reference's interpolant inside a model step that holds
 * nlev - 2 time knots of the wind lattice.  2 levels = picles_set_winds, 3 = picles_set_winds_knot; more: a POLYLINE window,
 * u(s) = u0 + s du + Σ_k max(s - s_k, 0) b_k with one term per knot (at most PICLES_MAX_KNOTS = 8 knots).  A step under a polyline
 * window runs the plain phases (stand-alone advance, general flavour; scatter + remesh behind it) instead of the fused launch —
 * the same results to the bit, one launch more per step. */
#define PICLES_MAX_KNOTS 8
int32_t picles_set_winds_polyline(picles_ctx *ctx, int32_t nlev, const double *const *u, const double *const *v, const double *times);

/* Non-Cartesian meshes (SphericalGrid.jl:207-240, spherical_grid_corrections.jl:3-21): per-node
 * projection kernel M = diag(m11, m22) of the propagation terms (particle_waves_v5.jl:536) and the
 * great-circle coefficient of PropagationCorrection, S_sphere = c̄_x * pc (:521-530); own rows,
 * col-major.  NULL pointers restore the Cartesian constants M = diag(1/dx, 1/dy), pc = 0. */
int32_t picles_set_metric(picles_ctx *ctx, const double *m11, const double *m22, const double *pc);

/* Gridded wind forcing (Utils/WindEmulator.jl:18-43 wind_interpolator =
 * Interpolations.linear_interpolation((x,y,t), u; extrapolation_bc = Periodic())): u,v on a regular
 * (x,y,t) lattice [nx*ny*nt], x fastest.  The library keeps the lattice in HBM and samples the
 * time levels of every step at the mesh nodes itself (tri-linear, periodic continuation), so no
 * wind data crosses PCIe inside the time loop.  mesh_x0/mesh_y0: coordinates of node (0,0).
 * Replaces picles_set_winds until picles_set_winds is called again.
 * Time semantics inside a model step [t, t+Δt] (picles_set_wind_grid_mode; LINEAR after every picles_set_wind_grid):
 *   PICLES_LATTICE_LINEAR  the reference's: the interpolant itself, kinks included.  No lattice knot strictly inside the step:
 *                          two levels (t, t+Δt), a straight line.  One knot inside: a third level AT the knot, two straight
 *                          segments (= picles_set_winds_knot).  Two to PICLES_MAX_KNOTS knots inside one step: a level at every
 *                          knot, the polyline through them (= picles_set_winds_polyline; such a step takes the plain phases instead
 *                          of the fused launch).  More than PICLES_MAX_KNOTS: the step is REFUSED with an error text before it has
 *                          changed anything (take shorter steps, or the mode below).
 *   PICLES_LATTICE_SMOOTH3 for a lattice that tabulates a smooth closure u(x,y,t) (knots at Δt/2 or finer): three levels
 *                          (t, t+Δt/2, t+Δt), the parabola through them (= picles_set_winds3 with the lattice sampled on the
 *                          device) — what carries tests/T04_2D_reg_test.jl:166-167's cos(3t/(3600 2π)) forcing to 1e-3 with no
 *                          host work in the time loop.  An approximation of the closure by design, not of the interpolant. */
int32_t picles_set_wind_grid(picles_ctx *ctx, int32_t nx, int32_t ny, int32_t nt,
                             double x0, double dx, double y0, double dy, double t0, double dt,
                             const double *u, const double *v, double mesh_x0, double mesh_y0);
#define PICLES_LATTICE_LINEAR  0
#define PICLES_LATTICE_SMOOTH3 1
int32_t picles_set_wind_grid_mode(picles_ctx *ctx, int32_t mode);
/* The interior knot of the lattice in the window (t, t+dt), as the LINEAR mode classifies it: returns the number of lattice
 * time knots strictly inside (0, 1, or 2 = "two or more"); *tk = the time of the first one.  A knot closer to either end than
 * 1e-9 of the lattice spacing counts as that end.  Host arithmetic only; the host layers use it to build the same windows
 * from host-sampled levels (picles_set_winds_knot).  (The periodic continuation in t has a whole number of
 * lattice intervals as its period: knots stay at whole multiples of lat_dt from lat_t0.) */
int32_t picles_lattice_knots(double lat_t0, double lat_dt, double t, double dt, double *tk);
/* All of them: returns the number of lattice time knots strictly inside (t, t+dt) by the same rule and writes the times of the first
 * `cap` into tks (tks may be NULL with cap = 0). */
int32_t picles_lattice_knot_times(double lat_t0, double lat_dt, double t, double dt, double *tks, int32_t cap);
/* node winds currently on the device (own rows); any pointer may be NULL */
int32_t picles_get_winds(picles_ctx *ctx, double *u0, double *v0, double *u1, double *v1);
/* the mid-window level of three-level winds; returns 1 (and writes nothing) when the current winds have two levels */
int32_t picles_get_winds_mid(picles_ctx *ctx, double *um, double *vm);

/* init_particles!(model) + SeedParticle (run.jl:199-247, core_2D.jl:434-488); clock := t0 */
int32_t picles_seed(picles_ctx *ctx, double t0);

/* time_step!(model, Δt) / movie_time_step! (TimeSteppers.jl:109-166,212-247) */
int32_t picles_time_step(picles_ctx *ctx, double dt, int32_t flags);
/* n consecutive run!-style steps (State .= 0; time_step!) enqueued back to back from C: the loop of
 * run! (run.jl:72-114) when nothing observes State between the steps.  Asynchronous like picles_time_step. */
int32_t picles_run_steps(picles_ctx *ctx, double dt, int32_t n_steps);
/* time_step!_advance / time_step!_remesh (TimeSteppers.jl:168-193); remesh does NOT tick */
int32_t picles_advance(picles_ctx *ctx, double dt, int32_t flags);
int32_t picles_remesh(picles_ctx *ctx, double dt);
int32_t picles_tick(picles_ctx *ctx, double dt);          /* tick!(clock, Δt) :163 */
int32_t picles_zero_state(picles_ctx *ctx);               /* State .= 0  (run.jl:75-79) */
double  picles_clock(const picles_ctx *ctx);

/* State / MovieState access (own rows; 3 planes of Nx*(j_end-j_begin)) */
int32_t picles_get_state(picles_ctx *ctx, double *state);
int32_t picles_set_state(picles_ctx *ctx, const double *state);
int32_t picles_get_movie_state(picles_ctx *ctx, double *state);

/* ---- State snapshots for run!(…; store / cash_store) (run.jl:94-112, storing.jl:109-119) ----------
 * push: stream-ordered device copy of State into a ring slot, then an asynchronous D2H copy into
 * pinned host memory on a side stream — the next time steps overlap the PCIe transfer.
 * pop: wait for the OLDEST pushed snapshot and hand it out (Nx*ny_loc*3 doubles + its model time).
 * The host writes it wherever the store lives (HDF5 waves/data[time,x,y,state] in the reference). */
int32_t picles_store_init(picles_ctx *ctx, int32_t n_slots);
int32_t picles_store_push(picles_ctx *ctx);
int32_t picles_store_pop(picles_ctx *ctx, double *state, double *time);
int32_t picles_store_pending(const picles_ctx *ctx);

/* particles (own rows; z is 5 planes: lne, c̄x, c̄y, x, y). Any pointer may be NULL.
 * The state vector of a switched-off particle (on == 0) is dead storage: its content is unspecified. */
int32_t picles_get_particles(picles_ctx *ctx, double *z, uint8_t *on, uint8_t *boundary,
                             int32_t *status);
int32_t picles_set_particles(picles_ctx *ctx, const double *z, const uint8_t *on);

int32_t picles_get_counters(picles_ctx *ctx, picles_counters *c);   /* syncs */
int32_t picles_reset_counters(picles_ctx *ctx);                    /* syncs the device; does NOT complete a pending fused step */
/* on = 1: HIP events around every kernel launch (per-launch samples); on = 2: one event pair around each picles_run_steps call —
 * its launches are back to back on one stream, so advance_ms / advance_launches is the mean launch duration with nothing
 * recorded between the launches (an event between two dependent launches costs about half a microsecond of idle GPU: 20 % of a
 * 256² step, 0.6 % of a 4096² one); picles_slab_run_steps likewise: one pair on the stream of its interior launches (the edge
 * launches and exchanges of its steps are ordered into that stream), no phase events; other launches are timed per launch as
 * with 1.  0: off. */
int32_t picles_enable_timing(picles_ctx *ctx, int32_t on);
int32_t picles_get_timing(picles_ctx *ctx, picles_timing *t);       /* syncs */
/* per-launch device durations [ms] since picles_enable_timing(1): kind 0 = step / advance launches, 1 = scatter, 2 = remesh.
 * Copies up to cap values; returns the number of samples held (>= 0) or an error (< 0). */
int32_t picles_get_timing_samples(picles_ctx *ctx, int32_t kind, double *out_ms, int32_t cap);
/* Diagnostic: the dispatch order the latest whole-grid fused step filed for its successor (cost-ordered dispatch):
 * out[0] = workgroups that did work, out[1] = workgroups with nothing to do, out[2 ...] = the logical 256-node blocks in the order
 * they will be dealt (busy ones from the front, calm ones from the back).  Copies min(cap, 2 + n) ints; returns n = the number of
 * workgroups of that launch when a complete order was filed (out[0] + out[1] == n), 0 when none was (the run is not mixed; a slab reports
 * the order of the launch over its interior rows), < 0 on error.  Syncs and completes a pending fused step, like every getter.  No counterpart in the reference. */
int32_t picles_get_dispatch_order(picles_ctx *ctx, int32_t *out, int32_t cap);
int32_t picles_sync(picles_ctx *ctx);

/* ---- split phases for the slab-partitioned (multi-GPU) step -------------------
 * One model step on rank r:
 *   picles_advance_rows(EDGE)  -> edge rows' scatter records are final
 *   [host: exchange picles_halo_send_dev -> neighbour's picles_halo_recv_dev (RCCL)]
 *   picles_advance_rows(INTERIOR)   (overlaps the exchange on another stream)
 *   picles_scatter_remesh()    -> pull-scatter own rows from own + ghost records, remesh
 * Streams are caller-provided HIP streams (void* = hipStream_t; NULL = context stream). */
#define PICLES_ROWS_ALL      0
#define PICLES_ROWS_EDGE     1
#define PICLES_ROWS_INTERIOR 2
int32_t picles_begin_step(picles_ctx *ctx, double dt, int32_t flags);
int32_t picles_advance_rows(picles_ctx *ctx, int32_t which, void *stream);
int32_t picles_scatter_remesh(picles_ctx *ctx, void *stream);  /* + tick */
/* Fused form of the same step (DESIGN.md k_step): the scatter + remesh of the PREVIOUS step ride on
 * the advance launches of the current one, so a model step is picles_step_rows(EDGE) -> exchange ||
 * picles_step_rows(INTERIOR), with no separate scatter launch.  picles_begin_fused_step returns 1
 * (and does nothing) when the step cannot be fused — time-varying or gridded winds, a per-node
 * metric — in which case the plain phases above are used.  The last step's scatter + remesh are
 * completed lazily by whichever call observes State / particles / counters. */
int32_t picles_begin_fused_step(picles_ctx *ctx, double dt);
int32_t picles_step_rows(picles_ctx *ctx, int32_t which, void *stream);
int32_t picles_end_fused_step(picles_ctx *ctx);
/* device pointers + byte count of the contiguous halo blocks (halo_rows record rows each):
 * side 0 = low-j neighbour, 1 = high-j neighbour */
int32_t picles_halo_send_dev(picles_ctx *ctx, int32_t side, void **ptr, size_t *bytes);
int32_t picles_halo_recv_dev(picles_ctx *ctx, int32_t side, void **ptr, size_t *bytes);
int32_t picles_halo_rows(const picles_ctx *ctx);
int32_t picles_set_halo_rows(picles_ctx *ctx, int32_t halo_rows);  /* re-allocates records */
/* Treat a context that owns ALL rows as a slab anyway (on = 1): the y wrap of a periodic mesh then goes through the ghost
 * rows — filled by the halo exchange with itself, the "ring of one" — instead of being resolved locally.  Results are
 * bit-identical; it exists so that ONE GPU can run, and check, the complete multi-GPU data path (received ghost rows are
 * consumed by the pull).  Call before picles_seed. */
int32_t picles_set_slab_mode(picles_ctx *ctx, int32_t on);

/* ---- native slab ring: the multi-GPU model step driven from C, RCCL send/recv over xGMI -------------------
 * One process per GPU; rank r owns slab r (rows [j_begin, j_end) of its picles_grid), neighbours r-1 / r+1, closed to a
 * ring when the mesh is periodic in y.  RCCL is bound at run time with dlopen (no link-time dependency):
 *   rank 0:  picles_slab_unique_id(id)  ->  the host distributes the 128 bytes (MPI_Bcast, a TCP store, a file ...)
 *   all:     picles_slab_comm_init(ctx, id, rank, world)       ncclCommInitRank + two HIP streams
 *   all:     picles_slab_run_steps(ctx, dt, n, flags)           n model steps, asynchronous, no host work in between:
 *              edge rows (stream E) -> ncclGroup{Send,Send,Recv,Recv} of the halo blocks in place (stream E)
 *              || interior rows (stream M) -> M waits for E
 *            flags as picles_time_step (PICLES_STEP_ZERO_FIRST = run!-style = fused k_step launches)
 *   picles_sync / picles_get_state / ... complete the last step as usual.
 * Replaces the @threads loops of time_step! (TimeSteppers.jl:144-178) across GPUs; the reference has no distributed
 * counterpart on this path. */
#define PICLES_SLAB_ID_BYTES 128
int32_t picles_slab_unique_id(void *id128);
int32_t picles_slab_comm_init(picles_ctx *ctx, const void *id128, int32_t rank, int32_t world);
int32_t picles_slab_run_steps(picles_ctx *ctx, double dt, int32_t n_steps, int32_t flags);
int32_t picles_slab_exchange(picles_ctx *ctx);        /* one synchronous halo exchange of the current records (warm-up of the RCCL channels) */
int32_t picles_slab_streams(picles_ctx *ctx, void **edge_stream, void **interior_stream);   /* hipStream_t of the ring (timing, profiling) */
int32_t picles_slab_comm_destroy(picles_ctx *ctx);
/* Where a ring step's time goes, from HIP events the ring records on its two streams while picles_enable_timing(ctx, 1) is on
 * (five per step): sums over the steps since the last call.  The diagnosis of a multi-GPU run in one look: the edge launch, the
 * exchange behind it (ncclGroup of the halo blocks, on the edge stream), the interior launch (other stream), and whether the
 * exchange had completed when the interior launch ended (it then cost nothing).  Syncs the device; clears the sums. */
typedef struct picles_slab_phases {
    uint64_t steps;            /* ring steps measured                                                          */
    uint64_t exchange_hidden;  /* ... of which the exchange had completed before the interior launch ended      */
    double   edge_ms;          /* edge-row launches (stream E)                                                  */
    double   exchange_ms;      /* end of the edge launch -> end of the send/recv group (stream E)               */
    double   interior_ms;      /* interior launches (stream M)                                                  */
    double   slack_ms;         /* interior end - exchange end, summed: > 0 = the exchange finished first        */
    double   span_ms;          /* edge begin -> the later of exchange end and interior end, summed              */
} picles_slab_phases;
int32_t picles_slab_get_phases(picles_ctx *ctx, picles_slab_phases *out);

/* ---- generic particle->mesh scatter of an arbitrary particle list --------------
 * (ParticleInCell.push_to_grid! over a list, ParticleInCell.jl:341-376,530-538):
 * counting-sort into a cell list, LDS-staged grid tiles (ds_add_f64 per corner), one global fp64 atomic per touched tile node.
 * ij: 2*n int32 (0-based birth node), xy: 2*n positions in cell units relative to ij,
 * charge: 3*n (e,mx,my), all SoA planes.  Adds into State. */
int32_t picles_scatter_particles(picles_ctx *ctx, int64_t n, const int32_t *ij,
                                 const double *xy, const double *charge);

#ifdef __cplusplus
}
#endif
#endif /* PICLES_HIP_H */
