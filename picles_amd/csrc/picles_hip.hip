/*
 * picles_hip.hip — HIP kernels (gfx950 / CDNA4) and the C ABI of include/picles_hip.h.
 *
 * One context = one GPU = one y-slab of the 2D Cartesian mesh.  Particles are born at and
 * return to their node every model step (mapping_2D.jl:279-356), so particle k IS node k:
 * the particle SoA, the State planes and the cell list share one index (identity cell list).
 *
 * HBM layout (all fp64 unless noted; i fastest, local rows jl = j - j_begin):
 *   state[3][n]      e, m_x, m_y planes              (reference State[Nx,Ny,3], col-major)
 *   movie[3][n]      MovieState snapshot
 *   z[5][n]          lne, c̄x, c̄y, x, y planes        (ParticleInstance2D.ODEIntegrator.u)
 *   qold[n], dtn[n]  PI-controller memory ln(qold), next dt (<0 => auto_dt_reset!); asw[n] i32 AutoSwitch state (solver 2)
 *   on[n] u8, pflags[n] u8 (bit0 stepped, bit1 group-2 (mask 3), bit2 boundary), status[n] i32
 *   wind u0,v0,u1,v1 [n] (+ uP,vP: level 0 of the previous window, for fused steps under device-sampled winds)
 *   rec[(ny_loc+2R)][6][Nx]   per-row scatter records e, m_x, m_y, wx_hi, wy_hi, code(list, floor x, floor y)
 *                    (+R ghost rows per side = the halo blocks exchanged between slabs; a row
 *                    block is contiguous, so halo send/recv need no pack/unpack)
 *
 * Kernels and the roofline that bounds each (DESIGN.md has the numbers):
 *   k_step           ONE launch per model step for run!-style steps: pull-scatter + remesh of the previous step,
 *                    whole adaptive advance of this one, new record.  Flavours (compile time): solver (DP5 / Tsit5 /
 *                    auto-switching), dead band, static vs time-varying winds, per-node metric.  fp64-VALU bound.
 *   k_advance        per-particle step alone: guards + adaptive RK of the 5-vector in
 *                    registers + charge/record write.  fp64-VALU bound (~10^4..10^5 flop per
 *                    particle-step against 136 B).  No MFMA: not a contraction.
 *   k_scatter        deterministic PULL scatter (each node sums its <= (2R+1)^2 candidate
 *                    sources in the reference's sequential order => bitwise reproducible),
 *                    fused with State zero-fill, MovieState snapshot and remesh. HBM bound.
 *   k_push_tiles     PUSH scatter: LDS-staged grid tile + apron, ds_add_f64 inside the tile,
 *                    one global fp64 atomic per touched tile node. HBM/atomic bound.
 *   k_remesh         stand-alone NodeToParticle! (split API).
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/picles_hip.h"
#include "physics.h"

#define PX_EXPORT extern "C" __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
struct GridP {
    int Nx, Ny;            /* global */
    int periodic_x, periodic_y;
    int tripolar;          /* N_TripolarNorth: y is not periodic; corners beyond the north edge fold back, mirrored in x */
    int j_begin, ny_loc;
    int R;                 /* ghost record rows per side (halo blocks); row offset of the records */
    int Rp;                /* reach of the pull scatter: >0 fixed (slabs: = R), 0 = read the
                              max_reach the advance kernel measured (single slab, no host sync) */
    int single_slab;       /* this context owns all rows: wrap in y is local */
    int ngroups;           /* 1, or 2 when grid-boundary (mask 3) particles are stepped */
};

/* Statistics are accumulated in NSLOTS independent slots (one 64-B line each) chosen by wave:
 * a single shared counter line serialises the 2.4 M per-launch wave atomics in one L2 channel
 * (measured: +9 ms per 4096² launch); spread over 1024 lines they are free. */
#define NSLOTS 1024
struct DevCounters {
    unsigned long long rhs, acc, rej, reseeds, clamps, maxit, adv, overflow;
    unsigned long long nonfinite, pad_[7];     /* two 64-B lines per slot */
};

struct Arrays {
    double *state, *movie;   /* 3 planes */
    double *z;               /* 5 planes */
    double *qold, *dtn;
    int *asw;                /* solver 2: AutoSwitch state (counter << 1 | rosenbrock_active; ASW_FRESH after a reinit!) */
    unsigned char *on, *pflags;
    int *status;
    double *u0, *v0, *u1, *v1;
    double *uP, *vP;         /* level-0 winds of the previous step's window (fused steps under time-varying winds) */
    double *m11, *m22, *pc;  /* per-node projection diag and great-circle coefficient (NULL: Cartesian) */
    double *rec;             /* records the scatter reads  (latest completed advance) */
    double *rec_out;         /* records the advance writes (the other buffer of the pair) */
    DevCounters *cnt;        /* [NSLOTS] */
    int *max_reach;          /* max scatter reach of the records in `rec` (read by the pull) */
    int *max_reach_out;      /* ... of the records being written to `rec_out` */
    int *max_reach_total;    /* running maximum since the last reset (slab halos are sized from it) */
    int *max_reach_next;     /* the counter the NEXT step will write: cleared by this step's advance launches (three counters
                                rotate — read / written / cleared — so no per-step memset launch sits between the steps) */
    long long n;             /* Nx * ny_loc */
};

#define PF_STEPPED 1
#define PF_GROUP2 2
#define PF_BOUNDARY 4

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o, 64);
        v = (t > v) ? t : v;
    }
    return v;
}

__device__ __forceinline__ Wind load_wind(const KParams &P, const Arrays &A, long long t)
{
    Wind w;
    w.u0 = A.u0[t];
    w.v0 = A.v0[t];
    if (P.wind_static) {
        w.du = 0.0;
        w.dv = 0.0;
    } else {
        w.du = A.u1[t] - w.u0;
        w.dv = A.v1[t] - w.v0;
    }
    return w;
}

__device__ __forceinline__ double *rec_row(const Arrays &A, const GridP &G, int row)
{
    return A.rec + (size_t)row * 6 * G.Nx;
}
__device__ __forceinline__ double *rec_row_out(const Arrays &A, const GridP &G, int row)
{
    return A.rec_out + (size_t)row * 6 * G.Nx;
}

/* scatter record plane 5: 0.0 = no contribution, else list (1 ocean, 2 grid boundary) and the
 * cell offsets (bx, by) = floor(x), floor(y) of the particle, packed into an exactly
 * representable integer-valued double (the planes stay one dtype => contiguous halo blocks) */
#define REC_BIAS 2048
/* A whole-grid context follows the scatter reach the advance measured, up to this many cells per model step; a
 * particle that travels farther (not a sea state: 64 cells are > 100 km in 10 minutes on the reference's meshes) is
 * not scattered and is counted in `halo_overflow`.  The pull visits (2R+1)² candidates per node: the cap also bounds
 * the cost of a step poisoned by one runaway particle. */
#define REACH_CAP 64
__device__ __forceinline__ double rec_encode(int grp, int bx, int by)
{
    return (double)(grp + 4 * (bx + REC_BIAS) + 4 * 4096 * (by + REC_BIAS));
}
__device__ __forceinline__ void rec_decode(double code, int &grp, int &bx, int &by)
{
    int ci = (int)code;
    grp = ci & 3;
    bx = ((ci >> 2) & 4095) - REC_BIAS;
    by = (ci >> 14) - REC_BIAS;
}

/* ------------------------------------------------------------------------------------------
 * k_seed — init_particles! / SeedParticle / InitParticleValues / init_z0_to_State!
 * (run.jl:199-247, core_2D.jl:247-288,434-488, initialize.jl:14-17)
 * ---------------------------------------------------------------------------------------- */
__global__ void __launch_bounds__(256) k_seed(KParams P, GridP G, Arrays A, const signed char *mask, double seed_T)
{
    pm_device_init();
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n) return;
    int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
    Vec5 z = {0.0, 0.0, 0.0, 0.0, 0.0};
    int on = 0;
    double e = 0.0, mx = 0.0, my = 0.0;
    if (mask[t] != 0) {
        Wind w = load_wind(P, A, t);
        double u, v;
        wind_at(P, w, 0.0, u, v);   /* winds at t = 0.0 (run.jl:213-215) */
        if (P.init_type == 0) {
            if (__builtin_sqrt(u * u + v * v) > __builtin_sqrt(2.0)) {
                seed_windsea(u, v, seed_T, z.lne, z.cx, z.cy);
                on = 1;
            } else {   /* MinimalParticle: unit-speed wind in the wind's direction (rand_sign -> +1) */
                double uu = (u == 0.0) ? 1.0 : u, vv = (v == 0.0) ? 1.0 : v;
                double am = __builtin_sqrt(uu * uu + vv * vv);
                seed_windsea(1.0 * uu / am, 1.0 * vv / am, seed_T, z.lne, z.cx, z.cy);
                on = 0;
            }
        } else {
            z.lne = P.def_lne; z.cx = P.def_cx; z.cy = P.def_cy;
            on = 1;
        }
        if (on) particle_to_charge(z.lne, z.cx, z.cy, e, mx, my);
    }
    A.z[t] = z.lne; A.z[t + A.n] = z.cx; A.z[t + 2 * A.n] = z.cy; A.z[t + 3 * A.n] = 0.0; A.z[t + 4 * A.n] = 0.0;
    A.on[t] = (unsigned char)on;
    A.qold[t] = PI_LNQOLDINIT;
    A.asw[t] = ASW_FRESH;
    A.dtn[t] = P.dt0;
    A.status[t] = 0;
    A.state[t] = e; A.state[t + A.n] = mx; A.state[t + 2 * A.n] = my;
    double *rr = rec_row(A, G, jl + G.R);
    rr[5 * G.Nx + i] = 0.0;
}

/* ------------------------------------------------------------------------------------------
 * advance! (mapping_2D.jl:118-243) of one particle held in registers: integrate / off->on test,
 * NaN / Inf / cap guards.  Shared by k_advance and the fused k_step.
 * ---------------------------------------------------------------------------------------- */
struct StepStats {
    PStats st;
    unsigned int adv, reseeds, clamps, maxit, overflow, nonfinite;
    int reach;
};

template <bool FAST, bool STATIC, bool METRIC = false, bool TSIT = false, bool AUTO = false>
__device__ __forceinline__ int advance_particle(const KParams &P, const Wind &w, Vec5 &z, int &on, double &qold,
                                                double &dtn, double t_start, double DT, StepStats &S,
                                                double m11 = 0.0, double m22 = 0.0, double pc = 0.0, int *asw = nullptr)
{
    int status = PICLES_ST_STEPPED;
    if (on) {
        S.adv = 1;
        integrate_dp5<FAST, STATIC, METRIC, TSIT, AUTO>(P, w, z, qold, dtn, t_start, DT, S.st, m11, m22, pc, asw);
        status |= S.st.status;
    } else {
        double u, v;
        wind_at(P, w, t_start + DT, u, v);
        if (u * u + v * v >= P.wind_min_sq) {
            reseed(P, u, v, DT, z);
            dtn = -1.0;
            on = 1;
            status |= PICLES_ST_SWITCHED_ON;
        }
    }
    if (pm_isnan(z.lne) || pm_isnan(z.cx) || pm_isnan(z.cy)) {
        double u, v;
        wind_at(P, w, t_start + DT, u, v);
        reseed(P, u, v, DT, z);
        dtn = -1.0;
        status |= PICLES_ST_RESEED_NAN;
    } else if (pm_isinf(z.lne) || pm_isinf(z.cx) || pm_isinf(z.cy)) {
        double u, v;
        wind_at(P, w, t_start, u, v);
        reseed(P, u, v, DT, z);
        dtn = -1.0;
        status |= PICLES_ST_RESEED_INF;
    } else if (z.lne > P.lne_max) {
        z.lne = P.lne_max;
        dtn = -1.0;
        status |= PICLES_ST_CLAMPED;
    }
    if (status & (PICLES_ST_RESEED_NAN | PICLES_ST_RESEED_INF | PICLES_ST_SWITCHED_ON)) S.reseeds = 1;
    if (status & PICLES_ST_CLAMPED) S.clamps = 1;
    if (status & PICLES_ST_MAXITERS) S.maxit = 1;
    return status;
}

/* scatter record of one advanced particle (ParticleToNode! inputs): charge, upper-node weights and
 * the packed (list, cell offset) code, into the OUT buffer */
__device__ __forceinline__ void write_record(const GridP &G, const Arrays &A, int i, int jl, unsigned char pf, int on,
                                             const Vec5 &z, StepStats &S)
{
    double *rr = rec_row_out(A, G, jl + G.R);
    double code = 0.0;
    if (on && !(pm_isfinite(z.x) && pm_isfinite(z.y))) {
        S.nonfinite = 1;         /* the reference would throw in Int(floor(NaN)) (ParticleInCell.jl:58-71): dropped and counted */
    } else if (on && !(pm_fabs(z.x) < 2047.0 && pm_fabs(z.y) < 2047.0)) {
        S.overflow = 1;          /* farther than the record code can hold (and than any int conversion should see) */
    } else if (on) {
        double e, mx, my;
        particle_to_charge(z.lne, z.cx, z.cy, e, mx, my);
        int bx, by;
        double wx, wy;
        index_weight(z.x, bx, wx);
        index_weight(z.y, by, wy);
        int r = (bx < 0) ? -bx : bx + 1;
        int ry = (by < 0) ? -by : by + 1;
        S.reach = (r > ry) ? r : ry;
        if (S.reach <= ((G.Rp > 0) ? G.Rp : REACH_CAP)) {
            rr[i] = e; rr[G.Nx + i] = mx; rr[2 * G.Nx + i] = my; rr[3 * G.Nx + i] = wx; rr[4 * G.Nx + i] = wy;
            code = rec_encode((pf & PF_GROUP2) ? 2 : 1, bx, by);
        }
        else { S.overflow = 1; S.reach = 0; }     /* not scattered, not part of the reach the pull follows */
    }
    rr[5 * G.Nx + i] = code;
}

/* statistics: one atomic per wave into the wave's slot.  The 0/1 flags are counted with a ballot +
 * scalar popcount (no cross-lane traffic), the step counters with two 64-bit butterfly sums
 * (accepted and rejected steps share one word), the reach with a ballot ladder. */
__device__ __forceinline__ void flush_stats(const Arrays &A, const StepStats &S)
{
    unsigned long long s_rhs = wave_sum_u64(S.st.rhs);
    unsigned long long s_ar = wave_sum_u64(((unsigned long long)S.st.acc << 32) | S.st.rej);
    unsigned long long b_adv = __ballot(S.adv != 0), b_res1 = __ballot(S.reseeds == 1), b_res2 = __ballot(S.reseeds >= 2);
    unsigned long long b_cl = __ballot(S.clamps != 0), b_mx = __ballot(S.maxit != 0), b_ov = __ballot(S.overflow != 0);
    unsigned long long b_nf = __ballot(S.nonfinite != 0);
    int m_reach = 0;
    if (__ballot(S.reach > 0)) {
        m_reach = 1;
        while (__ballot(S.reach > m_reach)) m_reach++;   /* reach is 1 in all but exotic steps: one extra ballot */
    }
    if ((threadIdx.x & 63) == 0) {
        DevCounters *c = A.cnt + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (NSLOTS - 1));
        unsigned long long s_acc = s_ar >> 32, s_rej = s_ar & 0xffffffffULL;
        unsigned long long s_res = (unsigned long long)__popcll(b_res1) + 2ull * __popcll(b_res2);
        if (s_rhs) atomicAdd(&c->rhs, s_rhs);
        if (s_acc) atomicAdd(&c->acc, s_acc);
        if (s_rej) atomicAdd(&c->rej, s_rej);
        if (b_adv) atomicAdd(&c->adv, (unsigned long long)__popcll(b_adv));
        if (s_res) atomicAdd(&c->reseeds, s_res);
        if (b_cl) atomicAdd(&c->clamps, (unsigned long long)__popcll(b_cl));
        if (b_mx) atomicAdd(&c->maxit, (unsigned long long)__popcll(b_mx));
        if (b_ov) atomicAdd(&c->overflow, (unsigned long long)__popcll(b_ov));
        if (b_nf) atomicAdd(&c->nonfinite, (unsigned long long)__popcll(b_nf));
        /* one address for the whole grid: only waves that would raise it touch it */
        if (m_reach > __hip_atomic_load(A.max_reach_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(A.max_reach_out, m_reach);
        if (m_reach > __hip_atomic_load(A.max_reach_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(A.max_reach_total, m_reach);
        if (blockIdx.x == 0 && threadIdx.x == 0) *A.max_reach_next = 0;     /* nobody reads or writes it during this step */
    }
}

/* XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8),
 * each with its own L2.  The pull scatter of a node reads the records of the rows above and below,
 * so vertically adjacent 256-node segments should share an L2: give every XCD one contiguous band
 * of the index space (logical block = (b % 8) * ceil(n/8) + b / 8).  Placement is a speed matter only. */
__device__ __forceinline__ unsigned int xcd_block(void)
{
    const unsigned int n = gridDim.x, b = blockIdx.x;
    const unsigned int per = (n + 7u) / 8u;
    unsigned int l = (b % 8u) * per + b / 8u;
    /* grids that are not a multiple of 8: the tail of the last bands is empty; fall back to identity there */
    return (n % 8u == 0u) ? l : b;
}

/* local rows [r0, r0+n0) ∪ [r1, r1+n1) -> particle index */
__device__ __forceinline__ bool rows_index(const GridP &G, int r0, int n0, int r1, int n1, long long &t)
{
    long long tid = (long long)xcd_block() * blockDim.x + threadIdx.x;
    long long na = (long long)n0 * G.Nx, nb = (long long)n1 * G.Nx;
    if (tid >= na + nb) return false;
    t = (tid < na) ? (long long)r0 * G.Nx + tid : (long long)r1 * G.Nx + (tid - na);
    return true;
}

/* ------------------------------------------------------------------------------------------
 * k_advance — advance! for the particles of the given rows.  One thread per particle; the whole
 * adaptive RK loop runs in registers.  Writes the particle's scatter record instead of
 * scattering: the scatter itself is k_scatter / k_step / k_push_tiles.
 * ---------------------------------------------------------------------------------------- */
template <bool FAST, bool STATIC, bool METRIC, bool TSIT, bool AUTO>
__global__ void __launch_bounds__(256) k_advance(KParams P, GridP G, Arrays A, double t_start, double DT,
                                                   int r0, int n0, int r1, int n1)
{
    dp_device_init(TSIT ? 1 : 0);
    pm_device_init();
    long long t = 0;
    bool active = rows_index(G, r0, n0, r1, n1, t);
    unsigned char pf = active ? A.pflags[t] : 0;
    active = active && (pf & PF_STEPPED);
    StepStats S = {{0u, 0u, 0u, 0}, 0u, 0u, 0u, 0u, 0u, 0u, 0};
    if (active) {
        int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
        Vec5 z;
        z.lne = A.z[t]; z.cx = A.z[t + A.n]; z.cy = A.z[t + 2 * A.n]; z.x = A.z[t + 3 * A.n]; z.y = A.z[t + 4 * A.n];
        int on = A.on[t];
        double qold = A.qold[t], dtn = A.dtn[t];
        Wind w = load_wind(P, A, t);
        int status;
        int asw = AUTO ? A.asw[t] : 0;
        if (METRIC) status = advance_particle<FAST, STATIC, true, TSIT, AUTO>(P, w, z, on, qold, dtn, t_start, DT, S, A.m11[t], A.m22[t], A.pc[t], &asw);
        else status = advance_particle<FAST, STATIC, false, TSIT, AUTO>(P, w, z, on, qold, dtn, t_start, DT, S, 0.0, 0.0, 0.0, &asw);
        if (AUTO) A.asw[t] = asw;
        A.z[t] = z.lne; A.z[t + A.n] = z.cx; A.z[t + 2 * A.n] = z.cy; A.z[t + 3 * A.n] = z.x; A.z[t + 4 * A.n] = z.y;
        A.on[t] = (unsigned char)on;
        A.qold[t] = qold;
        A.dtn[t] = dtn;
        A.status[t] = status;
        write_record(G, A, i, jl, pf, on, z, S);
    }
    flush_stats(A, S);
}

/* NodeToParticle! (mapping_2D.jl:279-356) on the node value (e,mx,my), all in registers.
 * Returns the branch: 0 = A (node -> particle), 1 = B/C (re-seed from the wind), 2 = D (off). */
__device__ __forceinline__ int remesh_regs(const KParams &P, const Wind &w, unsigned char pf, double e, double mx,
                                           double my, double clock, double DT, Vec5 &z)
{
    double u, v;
    wind_at(P, w, clock, u, v);          /* winds at model.clock.time, before tick! */
    bool bnd = (pf & PF_BOUNDARY) != 0;
    if (!bnd && (e >= P.min_e) && (mx * mx + my * my >= P.min_m2)) {
        charge_to_particle(e, mx, my, z);
        return 0;
    }
    if (u * u + v * v >= P.wind_min_sq) {
        reseed(P, u, v, DT, z);
        return 1;
    }
    return 2;
}

/* the same decision with the node wind behind pointers: it is read only by the (rare) branches that need it */
__device__ __forceinline__ int remesh_regs_lazy(const KParams &P, unsigned char pf, double e, double mx, double my,
                                                double DT, Vec5 &z, const double *pu, const double *pv)
{
    bool bnd = (pf & PF_BOUNDARY) != 0;
    if (!bnd && (e >= P.min_e) && (mx * mx + my * my >= P.min_m2)) {
        charge_to_particle(e, mx, my, z);
        return 0;
    }
    double u = *pu, v = *pv;
    if (u * u + v * v >= P.wind_min_sq) {
        reseed(P, u, v, DT, z);
        return 1;
    }
    return 2;
}

__device__ __forceinline__ void remesh_particle(const KParams &P, const Arrays &A, long long t, unsigned char pf,
                                                double e, double mx, double my, double clock, double DT,
                                                unsigned int &reseeds)
{
    Wind w = load_wind(P, A, t);
    Vec5 z;
    int br = remesh_regs(P, w, pf, e, mx, my, clock, DT, z);
    if (br <= 1) {
        A.z[t] = z.lne; A.z[t + A.n] = z.cx; A.z[t + 2 * A.n] = z.cy; A.z[t + 3 * A.n] = 0.0; A.z[t + 4 * A.n] = 0.0;
        if (br == 1) { A.qold[t] = PI_LNQOLDINIT; A.asw[t] = ASW_FRESH; reseeds = 1; }   /* reinit! */
        A.dtn[t] = -1.0;
        A.on[t] = 1;
    } else {
        A.on[t] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * k_scatter — ParticleToNode! / push_to_grid! (mapping_2D.jl:59-73, ParticleInCell.jl:341-376,
 * 504-508,530-538) as a deterministic PULL: node (i,j) visits its (2R+1)² candidate source
 * particles in the reference's sequential order (ocean list then grid-boundary list, each
 * column-major; periodic wraps sorted by their wrapped index) and adds (wx*wy)*charge of the
 * one corner that lands on it.  Drop/wrap per axis follows the GRID's periodicity.
 * Fused: State zero-fill (accum=0), MovieState snapshot + post-remesh zero (movie=1),
 * remesh (REMESH).
 * ---------------------------------------------------------------------------------------- */
/* one candidate source: record element offset `off` (plane 0 of the source particle), cell offsets
 * (di, dj) of the source relative to the node */
__device__ __forceinline__ void pull_candidate(const double *__restrict__ rec, unsigned int off, unsigned int pl, int di, int dj,
                                               int grp, bool ok, double &s0, double &s1, double &s2)
{
    /* the source feeds this node iff its group matches and its cell offset is (-di - ax, -dj - ay) with
     * ax, ay in {0, 1}: in code space that is code - code(grp, -di - 1, -dj - 1) in {0, 4, 16384, 16388}
     * (bit 2 clear = upper x node, bit 14 clear = upper y node); an empty record (code 0) gives a negative
     * difference */
    const int d = (int)rec[off + 5u * pl] - (grp + 4 * (REC_BIAS - 1 - di) + 4 * 4096 * (REC_BIAS - 1 - dj));
    if (ok && (d & ~(4 | 16384)) == 0) {
        const bool ax = !(d & 4), ay = !(d & 16384);
        double wxh = rec[off + 3u * pl], wyh = rec[off + 4u * pl];
        double w = (ax ? wxh : 1.0 - wxh) * (ay ? wyh : 1.0 - wyh);
        s0 += w * rec[off];
        s1 += w * rec[off + pl];
        s2 += w * rec[off + 2u * pl];
    }
}

/* floor(a / b) for b > 0 */
__device__ __forceinline__ int floor_div(int a, int b) { int q = a / b; return (a % b < 0) ? q - 1 : q; }

/* General form of the pull for a reach that wraps around a periodic axis (2R + 1 > N: tiny grids, huge
 * steps): several offsets d alias the same source row / column.  Sources are visited in ascending
 * wrapped index (the reference's sequential order); for each one every aliasing offset is tried (at most
 * one can match, the cell offset stored in the record decides).  O(N) per axis, only used when needed. */
__device__ __forceinline__ void pull_node_aliased(const GridP &G, const Arrays &A, int i, int jl, int R,
                                  double &s0, double &s1, double &s2)
{
    const int RO = G.R, Nx = G.Nx, Ny = G.Ny, j = jl + G.j_begin;
    const double *__restrict__ rec = A.rec;
    const unsigned int pl = (unsigned int)Nx, rowlen = 6u * pl;
    const int jlo = G.periodic_y ? 0 : max(0, j - R), jhi = G.periodic_y ? Ny - 1 : min(Ny - 1, j + R);
    const int ilo = G.periodic_x ? 0 : max(0, i - R), ihi = G.periodic_x ? Nx - 1 : min(Nx - 1, i + R);
    const bool wrap_y = G.periodic_y && G.single_slab;     /* slabs: a periodic y never aliases (checked at create) */
    for (int grp = 1; grp <= G.ngroups; grp++) {
        for (int js = jlo; js <= jhi; js++) {
            int dj0 = js - j;
            if (G.periodic_y) dj0 -= floor_div(dj0 + R, Ny) * Ny;          /* smallest alias >= -R */
            if (dj0 > R) continue;
            for (int is = ilo; is <= ihi; is++) {                          /* sources in ascending index ... */
                int di0 = is - i;
                if (G.periodic_x) di0 -= floor_div(di0 + R, Nx) * Nx;
                /* ... each with all its aliasing offsets (at most one pair can match a corner of this source) */
                for (int dj = dj0; dj <= R; dj += (G.periodic_y ? Ny : 2 * R + 1)) {
                    const int row = (wrap_y || !G.periodic_y ? js - G.j_begin : jl + dj) + RO;
                    for (int di = di0; di <= R; di += (G.periodic_x ? Nx : 2 * R + 1))
                        pull_candidate(rec, (unsigned int)row * rowlen + (unsigned int)is, pl, di, dj, grp, true, s0, s1, s2);
                }
            }
        }
    }
}

/* sum of the contributions to node (i, j); RT > 0: reach known at compile time (fully unrolled:
 * all candidate codes are loaded before any is inspected), RT == 0: runtime reach R.
 * Record elements are addressed with 32-bit offsets from one base pointer (the host falls back to
 * a single-plane-per-call layout check: (ny_loc + 2R) * 6 * Nx < 2^31 elements). */
template <int RT>
__device__ __forceinline__ void pull_node(const GridP &G, const Arrays &A, int i, int jl, int Rdyn,
                                          double &s0, double &s1, double &s2)
{
    const int R = RT ? RT : Rdyn, W = 2 * R + 1;
    const int RO = G.R;   /* row offset of the own rows inside rec */
    const int j = jl + G.j_begin, Nx = G.Nx;
    const double *__restrict__ rec = A.rec;
    const unsigned int pl = (unsigned int)Nx, rowlen = 6u * pl;
    /* interior nodes: no wrap, no drop, natural (= reference) order */
    bool interior = (i - R >= 0) && (i + R < Nx) && (j - R >= 0) && (j + R < G.Ny);
    if (interior) {
        const unsigned int base = (unsigned int)(jl + RO) * rowlen + (unsigned int)i;
        for (int grp = 1; grp <= G.ngroups; grp++) {
            if constexpr (RT == 1) {
#pragma unroll
                for (int dj = -1; dj <= 1; dj++) {
#pragma unroll
                    for (int di = -1; di <= 1; di++)
                        pull_candidate(rec, base + (unsigned int)(dj * (int)rowlen + di), pl, di, dj, grp, true, s0, s1, s2);
                }
            } else {
                /* wider reach: one row of candidates at a time keeps the register footprint of the
                 * fused step kernel at two waves per SIMD */
#pragma unroll 1
                for (int dj = -R; dj <= R; dj++) {
                    const unsigned int rb = base + (unsigned int)(dj * (int)rowlen);
                    if constexpr (RT != 0) {
#pragma unroll
                        for (int di = -RT; di <= RT; di++)
                            pull_candidate(rec, rb + (unsigned int)di, pl, di, dj, grp, true, s0, s1, s2);
                    } else {
                        for (int di = -R; di <= R; di++)
                            pull_candidate(rec, rb + (unsigned int)di, pl, di, dj, grp, true, s0, s1, s2);
                    }
                }
            }
        }
        return;
    }
    int shx = 0, shy = 0;
    if (G.periodic_x) { if (i - R < 0) shx = R - i; else if (i + R >= Nx) shx = Nx - i + R; }
    if (G.periodic_y) { if (j - R < 0) shy = R - j; else if (j + R >= G.Ny) shy = G.Ny - j + R; }
    for (int grp = 1; grp <= G.ngroups; grp++) {
#pragma unroll 1
        for (int sj = 0; sj < W; sj++) {
            int qj = sj + shy; if (qj >= W) qj -= W;
            int dj = qj - R;
            int jj = j + dj;
            bool rowok = G.periodic_y || (jj >= 0 && jj < G.Ny);
            int row;
            if (G.single_slab) {
                int jw = jj; if (jw < 0) jw += G.Ny; else if (jw >= G.Ny) jw -= G.Ny;
                row = jw + RO;
            } else {
                row = jl + dj + RO;
            }
            if (!rowok) continue;   /* beyond a non-periodic edge: dropped, and nothing is read */
#pragma unroll 1
            for (int si = 0; si < W; si++) {
                int qi = si + shx; if (qi >= W) qi -= W;
                int di = qi - R;
                int ii = i + di;
                if (ii < 0 || ii >= Nx) {
                    /* beyond a non-periodic edge (at ANY distance: the reach may exceed the grid): dropped, nothing
                     * is read.  Periodic: one wrap suffices, a reach of N/2 or more takes pull_node_aliased. */
                    if (!G.periodic_x) continue;
                    ii += (ii < 0) ? Nx : -Nx;
                }
                pull_candidate(rec, (unsigned int)row * rowlen + (unsigned int)ii, pl, di, dj, grp, true, s0, s1, s2);
            }
        }
    }
}

/* floor modulo for b > 0 */
__device__ __forceinline__ int floor_mod(int a, int b) { int r = a % b; return (r < 0) ? r + b : r; }

/* N_TripolarNorth (ParticleInCell.jl:353-361, TripolarNorthBoundary :409-428): a node of the top band (j >= Ny - R)
 * receives ordinary corners and corners folded back over the north seam, mirrored in x.  Candidate sources — rows
 * j-R .. Ny-1, columns within R of i or of the mirror column Nx-2-i — are visited in ascending index; each replays
 * its four corners in construct_loop order through the boundary rule of the push (0-based): corner (ci, cj) with
 * cj >= Ny lands on (Nx-1 - mod(ci+1, Nx), 2Ny-1-cj), with cj < 0 is dropped, otherwise on (mod(ci, Nx), cj). */
__device__ __forceinline__ void pull_node_tripolar(const GridP &G, const Arrays &A, int i, int jl, int R,
                                                   double &s0, double &s1, double &s2)
{
    const int RO = G.R, Nx = G.Nx, Ny = G.Ny, j = jl + G.j_begin, W = 2 * R + 1;
    const double *__restrict__ rec = A.rec;
    const unsigned int pl = (unsigned int)Nx, rowlen = 6u * pl;
    const int cB = floor_mod(Nx - 2 - i, Nx);
    const bool narrow = (2 * W <= Nx);       /* two disjoint-or-touching windows; otherwise scan the whole row */
    const int a0 = floor_mod(i - R, Nx), b0 = floor_mod(cB - R, Nx);
    for (int grp = 1; grp <= G.ngroups; grp++) {
        for (int js = max(0, j - R); js < Ny; js++) {
            const unsigned int rbase = (unsigned int)(js - G.j_begin + RO) * rowlen;
            /* ascending merge of the two wrapped windows [a0, a0+W) and [b0, b0+W) (mod Nx) */
            int ka = 0, kb = 0, is_full = 0;
            while (narrow ? (ka < W || kb < W) : (is_full < Nx)) {
                int is;
                if (narrow) {
                    /* element k of a wrapped window in ascending order: the wrapped-around part comes first */
                    const int wa = a0 + W - Nx, wb = b0 + W - Nx;      /* > 0: that many elements wrap to 0.. */
                    int ea = (ka < W) ? ((wa > 0) ? ((ka < wa) ? ka : a0 + (ka - wa)) : a0 + ka) : 0x7fffffff;
                    int eb = (kb < W) ? ((wb > 0) ? ((kb < wb) ? kb : b0 + (kb - wb)) : b0 + kb) : 0x7fffffff;
                    is = min(ea, eb);
                    if (ea == is) ka++;
                    if (eb == is) kb++;
                } else {
                    is = is_full++;
                }
                const int ci0 = (int)rec[rbase + 5u * pl + (unsigned int)is];
                if (ci0 == 0 || (ci0 & 3) != grp) continue;
                const int bx = ((ci0 >> 2) & 4095) - REC_BIAS, by = (ci0 >> 14) - REC_BIAS;
                const double wxh = rec[rbase + 3u * pl + (unsigned int)is], wyh = rec[rbase + 4u * pl + (unsigned int)is];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int ax = k & 1, ay = k >> 1;
                    int ci = is + bx + ax, cj = js + by + ay;
                    if (cj < 0) continue;
                    if (cj >= Ny) { ci = Nx - 1 - floor_mod(ci + 1, Nx); cj = 2 * Ny - 1 - cj; }
                    else ci = floor_mod(ci, Nx);
                    if (ci != i || cj != j) continue;
                    const double w = (ax ? wxh : 1.0 - wxh) * (ay ? wyh : 1.0 - wyh);
                    s0 += w * rec[rbase + (unsigned int)is];
                    s1 += w * rec[rbase + pl + (unsigned int)is];
                    s2 += w * rec[rbase + 2u * pl + (unsigned int)is];
                }
            }
        }
    }
}

/* reach dispatch: compile-time reach 1 and 2, runtime reach otherwise; a reach that wraps around a periodic
 * axis takes the general (aliasing-aware) form */
__device__ __forceinline__ void pull_any(const GridP &G, const Arrays &A, int i, int jl, int R,
                                         double &s0, double &s1, double &s2)
{
    const int W = 2 * R + 1;
    if (G.tripolar && jl + G.j_begin >= G.Ny - R) pull_node_tripolar(G, A, i, jl, R, s0, s1, s2);
    else if ((G.periodic_x && W > G.Nx) || (G.periodic_y && W > G.Ny)) pull_node_aliased(G, A, i, jl, R, s0, s1, s2);
    else if (R == 1) pull_node<1>(G, A, i, jl, 1, s0, s1, s2);
    else if (R == 2) pull_node<2>(G, A, i, jl, 2, s0, s1, s2);
    else if (R == 3) pull_node<3>(G, A, i, jl, 3, s0, s1, s2);     /* a fully developed sea under strong winds */
    else pull_node<0>(G, A, i, jl, R, s0, s1, s2);
}

/* reach the pull of local row jl must cover.  A whole-grid context follows the reach its own advance measured.  A slab
 * covers halo_rows for the rows that can receive from a neighbour's particles (the edge rows: the neighbour's reach is not
 * known here), and for its interior rows — fed by own particles only — again the measured reach, which is what keeps the
 * common case at 9 candidates per node instead of (2 halo_rows + 1)².  Any reach >= the true one gives the same bits. */
__device__ __forceinline__ int pull_reach(const GridP &G, const Arrays &A, int jl)
{
    if (G.Rp > 0 && (jl < G.R || jl >= G.ny_loc - G.R)) return G.Rp;
    int m = *A.max_reach;
    if (m < 1) m = 1;
    return (G.Rp > 0 && m > G.Rp) ? G.Rp : m;
}

template <bool REMESH>
__global__ void __launch_bounds__(256) k_scatter(KParams P, GridP G, Arrays A, int accum, int movie,
                                                   double clock, double DT)
{
    if (REMESH) pm_device_init();
    long long t = (long long)xcd_block() * blockDim.x + threadIdx.x;
    unsigned int reseeds = 0;
    if (t < A.n) {
        int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (accum) { s0 = A.state[t]; s1 = A.state[t + A.n]; s2 = A.state[t + 2 * A.n]; }
        pull_any(G, A, i, jl, pull_reach(G, A, jl), s0, s1, s2);
        if (movie) {
            A.movie[t] = s0; A.movie[t + A.n] = s1; A.movie[t + 2 * A.n] = s2;
            A.state[t] = 0.0; A.state[t + A.n] = 0.0; A.state[t + 2 * A.n] = 0.0;
        } else {
            A.state[t] = s0; A.state[t + A.n] = s1; A.state[t + 2 * A.n] = s2;
        }
        if (REMESH) {
            unsigned char pf = A.pflags[t];
            if (pf & PF_STEPPED) remesh_particle(P, A, t, pf, s0, s1, s2, clock, DT, reseeds);
        }
    }
    if (REMESH) {
        unsigned long long s = wave_sum_u64(reseeds);
        if ((threadIdx.x & 63) == 0 && s)
            atomicAdd(&A.cnt[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (NSLOTS - 1)].reseeds, s);
    }
}

/* ------------------------------------------------------------------------------------------
 * k_step — one whole model step per launch, for consecutive run!-style steps (State zeroed
 * before each step, time-constant winds): the thread of node/particle k
 *   1. pull-scatters the PREVIOUS step's records -> State[k]          (k_scatter)
 *   2. remeshes particle k from that node value, in registers          (NodeToParticle!)
 *   3. advances it over the CURRENT step and writes its new record into the other record buffer
 * The particle state never round-trips through HBM: per step only records (48 B), State (24 B),
 * winds (16 B), ln(qold) (8+8 B) and status move.  The last step's scatter+remesh is done by a
 * stand-alone k_scatter when somebody looks (flush()).  Results are bit-identical to the
 * k_advance + k_scatter sequence.
 * ---------------------------------------------------------------------------------------- */
/* STATIC = false: winds linear in time over the step window (u0,v0 -> u1,v1); the remesh of the previous step
 * needs the wind at ITS start-of-step clock = level 0 of the previous window, kept in (uP, vP) */
/* Occupancy: the explicit pairs fit 168 VGPRs = three waves per SIMD; the auto-switching flavour (Rosenbrock23 peaks at
 * ~220 live registers) is built for two waves with no scratch — measured equal to the spilling three-wave build, and a
 * two-phase split was measured slower (profiles/r2_two_phase_auto_experiment.md). */
template <bool FAST, bool TSIT, bool STATIC, bool METRIC, bool AUTO>
__global__ void __launch_bounds__(256, (FAST && !AUTO) ? 3 : 2) k_step(KParams P, GridP G, Arrays A, double t_prev, double DT_prev,
                                                double t_start, double DT, int r0, int n0, int r1, int n1)
{
    dp_device_init(TSIT ? 1 : 0);
    pm_device_init();
    long long t = 0;
    bool active = rows_index(G, r0, n0, r1, n1, t);
    StepStats S = {{0u, 0u, 0u, 0}, 0u, 0u, 0u, 0u, 0u, 0u, 0};
    if (active) {
        int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        pull_any(G, A, i, jl, pull_reach(G, A, jl), s0, s1, s2);
        A.state[t] = s0; A.state[t + A.n] = s1; A.state[t + 2 * A.n] = s2;
        unsigned char pf = A.pflags[t];
        if (pf & PF_STEPPED) {
            Wind w = load_wind(P, A, t);
            Vec5 z = {0.0, 0.0, 0.0, 0.0, 0.0};
            double qold = A.qold[t], dtn = -1.0;
            int br = remesh_regs_lazy(P, pf, s0, s1, s2, DT_prev, z, STATIC ? &A.u0[t] : &A.uP[t], STATIC ? &A.v0[t] : &A.vP[t]);
            int on = (br <= 1);
            int asw = AUTO ? A.asw[t] : 0;
            if (br == 1) { qold = PI_LNQOLDINIT; asw = ASW_FRESH; S.reseeds = 1; }
            unsigned int rs = S.reseeds;
            S.reseeds = 0;
            int status;
            if (METRIC) status = advance_particle<FAST, STATIC, true, TSIT, AUTO>(P, w, z, on, qold, dtn, t_start, DT, S, A.m11[t], A.m22[t], A.pc[t], &asw);
            else status = advance_particle<FAST, STATIC, false, TSIT, AUTO>(P, w, z, on, qold, dtn, t_start, DT, S, 0.0, 0.0, 0.0, &asw);
            if (AUTO) A.asw[t] = asw;
            S.reseeds += rs;
            A.qold[t] = qold;
            A.status[t] = status;
            write_record(G, A, i, jl, pf, on, z, S);
        }
    }
    flush_stats(A, S);
}

/* stand-alone time_step!_remesh (TimeSteppers.jl:182-193) */
__global__ void __launch_bounds__(256) k_remesh(KParams P, GridP G, Arrays A, double clock, double DT)
{
    pm_device_init();
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int reseeds = 0;
    if (t < A.n) {
        unsigned char pf = A.pflags[t];
        if (pf & PF_STEPPED)
            remesh_particle(P, A, t, pf, A.state[t], A.state[t + A.n], A.state[t + 2 * A.n], clock, DT, reseeds);
    }
    unsigned long long s = wave_sum_u64(reseeds);
    if ((threadIdx.x & 63) == 0 && s)
        atomicAdd(&A.cnt[(blockIdx.x * 4u + (threadIdx.x >> 6)) & (NSLOTS - 1)].reseeds, s);
}

/* ------------------------------------------------------------------------------------------
 * k_push_tiles — PUSH scatter with an LDS-staged grid tile.
 * One workgroup owns a TX×TY tile of birth nodes; its LDS holds the tile plus an apron of AP
 * nodes on every side (3 planes).  Every on-particle of the tile adds its 4 weighted corners
 * with ds_add_f64 (LDS atomics resolve same-node collisions inside the workgroup); corners
 * beyond the apron go straight to global fp64 atomics.  The LDS tile is then flushed with ONE
 * global atomic per touched node (drop / wrap by the grid's periodicity).
 * Sum order is not fixed => last-bit run-to-run differences; PICLES_STEP_ATOMIC selects it.
 * IDENTITY=true : particles = the tile's own nodes, read from the scatter records.
 * IDENTITY=false: particles = a cell-list segment of an arbitrary particle list
 *                 (picles_scatter_particles), sorted by the tile of the birth node.
 * ---------------------------------------------------------------------------------------- */
#define PT_TX 64
#define PT_TY 4
#define PT_AP 2
#define PT_LX (PT_TX + 2 * PT_AP)
#define PT_LY (PT_TY + 2 * PT_AP)

__device__ __forceinline__ void global_add3(const GridP &G, const Arrays &A, int ig, int jg, double a0, double a1, double a2)
{
    /* ig, jg: unwrapped global node indices */
    if (ig < 0 || ig >= G.Nx) { if (!G.periodic_x) return; ig %= G.Nx; if (ig < 0) ig += G.Nx; }
    if (jg < 0 || jg >= G.Ny) { if (!G.periodic_y) return; jg %= G.Ny; if (jg < 0) jg += G.Ny; }
    int jl = jg - G.j_begin;
    if (jl < 0 || jl >= G.ny_loc) return;   /* other slab: single-slab use only */
    long long t = (long long)jl * G.Nx + ig;
    unsafeAtomicAdd(&A.state[t], a0);
    unsafeAtomicAdd(&A.state[t + A.n], a1);
    unsafeAtomicAdd(&A.state[t + 2 * A.n], a2);
}

/* Wave-level pre-reduction of same-destination contributions (the "wavefront-reduced atomics" of the scatter design).
 * Every lane offers one contribution (value triple q, destination key; key < 0: nothing).  Lanes that are neighbours in
 * the wave and target the same node form a run; the run is summed with a segmented shuffle scan and its first lane
 * alone issues the LDS / global atomic.  With one particle per cell and lanes laid along x this turns the upper-x corner
 * of lane k and the lower-x corner of lane k+1 into ONE atomic (after rotating the upper corners by one lane); with a
 * cell-sorted particle list it also folds all particles of one cell.  Runs are short (2 in the identity layout), so the
 * scan stops as soon as no lane has a partner left: usually after one step. */
__device__ __forceinline__ bool wave_fold_runs(int key, double &q0, double &q1, double &q2)
{
    const int lane = threadIdx.x & 63;
    const int kl = __shfl_up(key, 1, 64);
    const bool head = (lane == 0) || (kl != key) || (key < 0);
    if (__ballot(!head) == 0) return key >= 0;             /* no two neighbouring lanes share a node */
    /* run id = number of run heads at or below this lane; two lanes belong to the same run iff the ids agree */
    const unsigned long long hb = __ballot(head);
    const int rid = __popcll(hb & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)));
    for (int o = 1; o < 64; o <<= 1) {
        const double a0 = __shfl_down(q0, o, 64), a1 = __shfl_down(q1, o, 64), a2 = __shfl_down(q2, o, 64);
        const int r2 = __shfl_down(rid, o, 64);
        const bool take = (lane + o < 64) && (r2 == rid);
        if (take) { q0 += a0; q1 += a1; q2 += a2; }
        if (__ballot(take) == 0) break;                      /* every run is folded */
    }
    return head && key >= 0;
}

template <bool IDENTITY>
__global__ void __launch_bounds__(256) k_push_tiles(GridP G, Arrays A, int ntx,
                                                      const int *seg_start, const int *perm,
                                                      const int *pij, const double *pxy, const double *pch, long long np)
{
    __shared__ double tile[3][PT_LY][PT_LX];
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int i0 = tx * PT_TX, j0 = ty * PT_TY + G.j_begin;   /* global origin of the tile */
    for (int k = threadIdx.x; k < 3 * PT_LY * PT_LX; k += blockDim.x) (&tile[0][0][0])[k] = 0.0;
    __syncthreads();

    int count, base = 0;
    if (IDENTITY) count = PT_TX * PT_TY;
    else { base = seg_start[blockIdx.x]; count = seg_start[blockIdx.x + 1] - base; }
    const int rounds = (count + (int)blockDim.x - 1) / (int)blockDim.x;
    for (int rd = 0; rd < rounds; rd++) {                 /* whole waves stay together: the fold below shuffles across lanes */
        const int k = rd * (int)blockDim.x + (int)threadIdx.x;
        bool have = k < count;
        int ib = 0, jb = 0, bx = 0, by = 0;                /* birth node (global), cell offset */
        double e = 0.0, mx = 0.0, my = 0.0, wxh = 0.0, wyh = 0.0;
        if (have) {
            if (IDENTITY) {
                ib = i0 + (k % PT_TX);
                jb = j0 + (k / PT_TX);
                const int jl = jb - G.j_begin;
                have = ib < G.Nx && jl < G.ny_loc;
                if (have) {
                    const double *rr = rec_row(A, G, jl + G.R);
                    const double code = rr[5 * G.Nx + ib];
                    have = code != 0.0;
                    if (have) {
                        int cg;
                        rec_decode(code, cg, bx, by);
                        e = rr[ib]; mx = rr[G.Nx + ib]; my = rr[2 * G.Nx + ib]; wxh = rr[3 * G.Nx + ib]; wyh = rr[4 * G.Nx + ib];
                    }
                }
            } else {
                const long long pidx = perm[base + k];
                ib = pij[pidx]; jb = pij[np + pidx];
                const double x = pxy[pidx], y = pxy[np + pidx];
                e = pch[pidx]; mx = pch[np + pidx]; my = pch[2 * np + pidx];
                have = pm_isfinite(x) && pm_isfinite(y);
                if (have) { index_weight(x, bx, wxh); index_weight(y, by, wyh); }
            }
        }
        /* the four corners in two passes of (lower-x, upper-x) per y row.  The upper-x contributions are rotated by one
         * lane (lane k offers the upper corner of lane k-1) so that, in the identity layout, it sits next to the lower
         * corner of the particle one cell to the right — the same node. */
#pragma unroll
        for (int ay = 0; ay < 2; ay++) {
            const double wy = ay ? wyh : 1.0 - wyh;
            const int jg = jb + by + ay;
            int klo = -1;
            double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
            for (int ax = 0; ax < 2; ax++) {
                const double w = (ax ? wxh : 1.0 - wxh) * wy;
                double q0 = w * e, q1 = w * mx, q2 = w * my;
                int ig = ib + bx + ax, jgc = jg;
                int li = ig - i0 + PT_AP, lj = jgc - j0 + PT_AP;
                const bool inside = have && li >= 0 && li < PT_LX && lj >= 0 && lj < PT_LY;
                if (have && !inside) global_add3(G, A, ig, jgc, q0, q1, q2);      /* beyond the apron: rare, straight to HBM */
                int key = inside ? lj * PT_LX + li : -1;
                if (ax == 1) {                                /* rotate the upper-x corner to the right-hand neighbour lane */
                    const int lane = threadIdx.x & 63;
                    const int kk = __shfl_up(key, 1, 64);
                    const double r0 = __shfl_up(q0, 1, 64), r1 = __shfl_up(q1, 1, 64), r2 = __shfl_up(q2, 1, 64);
                    const int k63 = __shfl(key, 63, 64);       /* lane 63's own upper corner wraps to lane 0 */
                    const double s0 = __shfl(q0, 63, 64), s1 = __shfl(q1, 63, 64), s2 = __shfl(q2, 63, 64);
                    key = lane ? kk : k63; q0 = lane ? r0 : s0; q1 = lane ? r1 : s1; q2 = lane ? r2 : s2;
                    /* pair it with this lane's own lower corner of the same y row: two offers per lane, folded one after
                     * the other — first the rotated upper corner joins the lower one if they agree */
                }
                if (ax == 0) { klo = key; l0 = q0; l1 = q1; l2 = q2; }
                else {
                    if (key >= 0 && key == klo) { l0 += q0; l1 += q1; l2 += q2; key = -1; }    /* same node: one offer */
                    /* lower corners (now carrying the neighbour's upper one) */
                    double f0 = l0, f1 = l1, f2 = l2;
                    if (wave_fold_runs(klo, f0, f1, f2)) {
                        const int lj2 = klo / PT_LX, li2 = klo % PT_LX;
                        atomicAdd(&tile[0][lj2][li2], f0); atomicAdd(&tile[1][lj2][li2], f1); atomicAdd(&tile[2][lj2][li2], f2);
                    }
                    /* upper corners that found no partner (different cell offset next door, tile edge) */
                    if (__ballot(key >= 0)) {
                        if (wave_fold_runs(key, q0, q1, q2)) {
                            const int lj2 = key / PT_LX, li2 = key % PT_LX;
                            atomicAdd(&tile[0][lj2][li2], q0); atomicAdd(&tile[1][lj2][li2], q1); atomicAdd(&tile[2][lj2][li2], q2);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PT_LY * PT_LX; k += blockDim.x) {
        int li = k % PT_LX, lj = k / PT_LX;
        double a0 = tile[0][lj][li], a1 = tile[1][lj][li], a2 = tile[2][lj][li];
        if (a0 == 0.0 && a1 == 0.0 && a2 == 0.0) continue;
        global_add3(G, A, i0 + li - PT_AP, j0 + lj - PT_AP, a0, a1, a2);
    }
}

/* cell list for an arbitrary particle list: histogram of birth tiles, then (after an exclusive
 * scan) a stable-enough fill of the permutation */
__global__ void k_tile_count(GridP G, int ntx, const int *pij, long long np, int *count, int *tile_of)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    int ib = pij[k], jb = pij[np + k] - G.j_begin;
    int tile = -1;
    if (ib >= 0 && ib < G.Nx && jb >= 0 && jb < G.ny_loc) {
        tile = (jb / PT_TY) * ntx + ib / PT_TX;
        atomicAdd(&count[tile], 1);
    }
    tile_of[k] = tile;
}
__global__ void k_tile_fill(long long np, const int *tile_of, const int *seg_start, int *cursor, int *perm)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    int tile = tile_of[k];
    if (tile < 0) return;
    int pos = atomicAdd(&cursor[tile], 1);
    perm[seg_start[tile] + pos] = (int)k;
}

/* ------------------------------------------------------------------------------------------
 * k_wind_sample — wind_interpolator (Utils/WindEmulator.jl:18-43): tri-linear interpolation of a
 * regular (x,y,t) lattice to the mesh nodes at time t, periodic continuation outside the lattice
 * (Interpolations.jl extrapolation_bc = Periodic(): period = last - first knot).
 * ---------------------------------------------------------------------------------------- */
struct WindGrid {
    int nx, ny, nt;
    double x0, inv_dx, y0, inv_dy, t0, inv_dt;
    double mesh_x0, mesh_y0, mesh_dx, mesh_dy;
    const double *u, *v;
};

__device__ __forceinline__ void lattice_coord(double c, int n, int &i0, double &f)
{
    /* c in lattice units; inside [0, n-1] as is, outside continued periodically */
    double per = (double)(n - 1);
    double w = (c < 0.0 || c > per) ? c - __builtin_floor(c / per) * per : c;
    double fl = __builtin_floor(w);
    i0 = (int)fl;
    if (i0 > n - 2) i0 = n - 2;   /* w == per after rounding */
    if (i0 < 0) i0 = 0;
    f = w - (double)i0;
}

__global__ void __launch_bounds__(256) k_wind_sample(GridP G, WindGrid Wg, double t, double *uo, double *vo, long long n)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int i = (int)(k % G.Nx), j = (int)(k / G.Nx) + G.j_begin;
    double x = Wg.mesh_x0 + (double)i * Wg.mesh_dx, y = Wg.mesh_y0 + (double)j * Wg.mesh_dy;
    int ix, iy, it;
    double fx, fy, ft;
    lattice_coord((x - Wg.x0) * Wg.inv_dx, Wg.nx, ix, fx);
    lattice_coord((y - Wg.y0) * Wg.inv_dy, Wg.ny, iy, fy);
    lattice_coord((t - Wg.t0) * Wg.inv_dt, Wg.nt, it, ft);
    size_t sx = 1, sy = (size_t)Wg.nx, st = (size_t)Wg.nx * Wg.ny;
    size_t b = ix * sx + iy * sy + it * st;
    const double *F[2] = {Wg.u, Wg.v};
    double out[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const double *f = F[c];
        double c00 = f[b] + (f[b + sx] - f[b]) * fx;
        double c10 = f[b + sy] + (f[b + sy + sx] - f[b + sy]) * fx;
        double c01 = f[b + st] + (f[b + st + sx] - f[b + st]) * fx;
        double c11 = f[b + st + sy] + (f[b + st + sy + sx] - f[b + st + sy]) * fx;
        double c0 = c00 + (c10 - c00) * fy;
        double c1 = c01 + (c11 - c01) * fy;
        out[c] = c0 + (c1 - c0) * ft;
    }
    uo[k] = out[0];
    vo[k] = out[1];
}

/* ------------------------------------------------------------------------------------------
 * host side
 * ---------------------------------------------------------------------------------------- */
static thread_local std::string g_create_error;

struct picles_ctx {
    picles_grid g;
    picles_phys ph;
    picles_ode od;
    picles_model md;
    KParams P;
    GridP G;
    Arrays A;
    int device;
    hipStream_t stream;
    hipEvent_t ev_edge;
    hipEvent_t ev_ctx = nullptr;        /* "everything enqueued on the context stream so far": caller streams wait for it */
    bool edge_pending = false;
    bool step_fresh = false;
    struct SlabRing *ring = nullptr;    /* native RCCL slab ring (picles_slab_*) */
    /* record buffer pair: rec_buf[cur] belongs to the step in flight / last completed advance */
    double *rec_buf[2] = {nullptr, nullptr};
    int *mr_buf[3] = {nullptr, nullptr, nullptr};     /* reach counters: one read (previous step), one written, one being cleared */
    int mr_w = 0;                                      /* index of the counter the step in flight writes */
    int cur = 0;
    /* fused stepping: the last advance's records still await their scatter + remesh */
    bool fuse_steps = true;
    bool pending = false;
    double pend_t = 0.0, pend_dt = 0.0;
    signed char *d_mask = nullptr;
    std::vector<signed char> h_mask;
    double clock = 0.0;
    /* pending step (begin_step .. scatter_remesh) */
    double step_dt = 0.0;
    int step_flags = 0;
    bool state_zero = false;      /* State known to be all zero (skip the accumulate read) */
    bool seeded = false;
    /* timing */
    bool timing = false;
    struct Ev { hipEvent_t a, b; int kind; };
    std::vector<Ev> ev_used, ev_free;
    picles_timing tim{};
    std::vector<float> tim_samples[3];   /* per-launch durations by kind (advance / scatter / remesh) */
    std::string err;
    /* snapshot ring (run! stores) */
    int store_slots = 0, store_head = 0, store_count = 0;
    std::vector<double *> store_dev, store_host;
    std::vector<hipEvent_t> store_ready, store_done;
    std::vector<double> store_time;
    hipStream_t store_stream = nullptr;
    /* gridded winds */
    bool wind_grid_on = false;
    WindGrid wg{};
    double *d_wgu = nullptr, *d_wgv = nullptr;
    double wind_t1 = 0.0;          /* time level currently held in (u1, v1) */
    bool wind_t1_valid = false;
    bool ext_streams = false;      /* a caller-provided stream has been used: order across streams with device syncs */
    bool ring_orders = false;      /* inside picles_slab_run_steps: the ring orders its streams against the context stream with events */
    /* generic scatter scratch */
    int *d_count = nullptr, *d_start = nullptr, *d_cursor = nullptr;
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
};

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
            return -10;                                                                        \
        }                                                                                      \
    } while (0)

/* hipStreamQuery without side effects on the sticky last-error state (hipErrorNotReady is not a failure) */
static bool stream_idle(hipStream_t s)
{
    hipError_t e = hipStreamQuery(s);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess;
}

static int fail(picles_ctx *c, int code, const std::string &m)
{
    c->err = m;
    return code;
}

static void timing_begin(picles_ctx *c, hipStream_t s, int kind)
{
    if (!c->timing) return;
    picles_ctx::Ev e;
    if (!c->ev_free.empty()) { e = c->ev_free.back(); c->ev_free.pop_back(); }
    else { hipEventCreate(&e.a); hipEventCreate(&e.b); }
    e.kind = kind;
    hipEventRecord(e.a, s);
    c->ev_used.push_back(e);
}
static void timing_end(picles_ctx *c, hipStream_t s)
{
    if (!c->timing) return;
    hipEventRecord(c->ev_used.back().b, s);
}
static void timing_collect(picles_ctx *c)
{
    for (auto &e : c->ev_used) {
        hipEventSynchronize(e.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e.a, e.b);
        if (e.kind >= 0 && e.kind < 3 && c->tim_samples[e.kind].size() < (1u << 20)) c->tim_samples[e.kind].push_back(ms);
        switch (e.kind) {
        case 0: c->tim.advance_ms += ms; c->tim.advance_launches++; break;
        case 1: c->tim.scatter_ms += ms; c->tim.scatter_launches++; break;
        case 2: c->tim.remesh_ms += ms; c->tim.remesh_launches++; break;
        default: c->tim.other_ms += ms;
        }
        c->ev_free.push_back(e);
    }
    c->ev_used.clear();
}

static size_t rec_bytes(const picles_ctx *c) { return (size_t)(c->G.ny_loc + 2 * c->G.R) * 6 * c->G.Nx * sizeof(double); }

/* Arrays as a kernel sees them: which record buffer is read (scatter) / written (advance) */
static Arrays arrays_for(picles_ctx *c, int read_buf, int write_buf)
{
    Arrays A = c->A;
    A.rec = c->rec_buf[read_buf];
    A.rec_out = c->rec_buf[write_buf];
    /* reach counters rotate with the steps, independently of the record pair: a launch that scatters the records it has just
     * written (read_buf == write_buf: k_scatter after k_advance) reads the counter of the step in flight */
    A.max_reach = c->mr_buf[read_buf == write_buf ? c->mr_w : (c->mr_w + 2) % 3];
    A.max_reach_out = c->mr_buf[c->mr_w];
    A.max_reach_next = c->mr_buf[(c->mr_w + 1) % 3];
    return A;
}

static int launch_scatter(picles_ctx *c, hipStream_t s, bool remesh);

/* scatter + remesh of the last fused step, if still outstanding */
static int flush(picles_ctx *c)
{
    if (!c->pending) return 0;
    c->pending = false;
    HIPCHK(c, hipDeviceSynchronize());   /* the fused launches may have run on caller-provided streams */
    double clock_save = c->clock, dt_save = c->step_dt;
    int flags_save = c->step_flags;
    c->clock = c->pend_t;           /* remesh samples the wind at the start-of-step clock */
    c->step_dt = c->pend_dt;
    c->step_flags = PICLES_STEP_ZERO_FIRST;
    int rc = launch_scatter(c, c->stream, true);
    c->clock = clock_save; c->step_dt = dt_save; c->step_flags = flags_save;
    return rc;
}

PX_EXPORT int32_t picles_abi_version(void) { return PICLES_ABI_VERSION; }

PX_EXPORT const char *picles_last_error(const picles_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

PX_EXPORT int32_t picles_destroy(picles_ctx *c);
PX_EXPORT int32_t picles_slab_comm_destroy(picles_ctx *c);

PX_EXPORT int32_t picles_create(const picles_grid *g, const picles_phys *p, const picles_ode *o,
                                const picles_model *m, int32_t device_id, int32_t halo_rows, picles_ctx **out)
{
    if (!g || !p || !o || !m || !out) { g_create_error = "null argument"; return -1; }
    *out = nullptr;
    if (g->Nx < 2 || g->Ny < 2) { g_create_error = "grid must be at least 2x2"; return -2; }
    if (g->j_begin < 0 || g->j_end > g->Ny || g->j_end <= g->j_begin) { g_create_error = "bad slab rows [j_begin,j_end)"; return -2; }
    if (o->solver < 0 || o->solver > 2) { g_create_error = "solver must be 0 (DP5), 1 (Tsit5) or 2 (AutoTsit5(Rosenbrock23()))"; return -3; }
    if (halo_rows < 1) halo_rows = 1;
    if (halo_rows > 1024) { g_create_error = "halo_rows must be <= 1024"; return -2; }
    if (!(o->abstol > 0.0) || !(o->reltol >= 0.0) || !(o->maxiters > 0) || !(o->dtmin >= 0.0)) {
        g_create_error = "ODE settings: abstol > 0, reltol >= 0, maxiters > 0, dtmin >= 0 required";
        return -3;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (this library has no CPU path)";
        return -4;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_error = "device_id out of range"; return -4; }
    if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = hipGetErrorString(e); return -4; }

    picles_ctx *c = new picles_ctx();
    memset(&c->A, 0, sizeof(c->A));
    c->stream = nullptr;
    c->ev_edge = nullptr;
    c->g = *g; c->ph = *p; c->od = *o; c->md = *m;
    c->device = device_id;
    /* derived constants: magic_fractions :87-92, e_T_func :271 (same primitives as the kernels) */
    KParams &P = c->P;
    P.r_g = p->r_g; P.inv_rg = 1.0 / p->r_g; P.C_alpha = p->C_alpha; P.C_phi = p->C_phi; P.C_e = p->C_e;
    double q = p->q;
    P.p = (-1.0 - 10.0 * q) / 2.0;
    P.n = 2.0 * q / (P.p + 4.0 * q);
    P.neg2p = -2.0 * P.p;
    double e_T = std::sqrt(p->c_e * pm_pow(p->c_alpha, -P.p / q) / pm_pow(p->gamma * p->c_beta * p->c_D, 1.0 / P.n));
    P.inv_eT = 1.0 / e_T;
    P.inv_eT4 = (P.inv_eT * P.inv_eT) * (P.inv_eT * P.inv_eT);
    P.half_inv_rg = 0.5 * P.inv_rg;
    P.two_inv_rg2 = 2.0 * (P.inv_rg * P.inv_rg);
    P.inv_dx = 1.0 / g->dx; P.inv_dy = 1.0 / g->dy;
    P.deadband2 = p->dir_deadband * p->dir_deadband;
    P.propagation = p->propagation; P.input = p->input; P.dissipation = p->dissipation;
    P.peak_shift = p->peak_shift; P.direction = p->direction; P.n_is_2 = (P.n == 2.0);
    P.abstol = o->abstol; P.reltol = o->reltol; P.dt0 = o->dt0; P.dtmin = o->dtmin;
    P.inv_abstol = 1.0 / o->abstol;
    P.maxiters = o->maxiters; P.force_dtmin = o->force_dtmin;
    P.solver = o->solver;
    P.lne_max = o->log_energy_maximum; P.wind_min_sq = o->wind_min_squared;
    P.init_type = m->init_type;
    P.def_lne = m->default_particle[0]; P.def_cx = m->default_particle[1]; P.def_cy = m->default_particle[2];
    P.min_e = m->minimal_state[0]; P.min_m2 = m->minimal_state[1];
    P.wind_static = 1; P.tw0 = 0.0; P.inv_dtw = 0.0;

    GridP &G = c->G;
    G.Nx = g->Nx; G.Ny = g->Ny; G.periodic_x = (g->periodic_x != 0); G.periodic_y = (g->periodic_y == 1);
    G.tripolar = (g->periodic_y == 2);
    if (g->periodic_y < 0 || g->periodic_y > 2 || (G.tripolar && !G.periodic_x)) {
        g_create_error = "periodic_y must be 0, 1 or 2 (tripolar north, which needs a periodic x axis)";
        delete c;
        return -2;
    }
    G.j_begin = g->j_begin; G.ny_loc = g->j_end - g->j_begin;
    G.single_slab = (g->j_begin == 0 && g->j_end == g->Ny);
    G.R = halo_rows;
    G.Rp = G.single_slab ? 0 : halo_rows;
    G.ngroups = 1;

    /* total mask (mask_utils.jl:38-55) for the local rows */
    long long n = (long long)G.Nx * G.ny_loc;
    c->h_mask.resize(n);
    for (int jl = 0; jl < G.ny_loc; jl++)
        for (int i = 0; i < G.Nx; i++) {
            int j = jl + G.j_begin;
            signed char mk;
            if (g->mask) mk = g->mask[(long long)j * G.Nx + i];
            else {
                bool ring = (!g->periodic_x && (i == 0 || i == G.Nx - 1)) || (!g->periodic_y && (j == 0 || j == G.Ny - 1));
                mk = ring ? 3 : 1;
            }
            c->h_mask[(long long)jl * G.Nx + i] = mk;
        }
    /* ocean_points (WaveGrowthModels2D.jl:256-270) and check_boundary_point (core_2D.jl:360-366) */
    std::vector<unsigned char> pf(n, 0);
    bool any3 = false;
    if (g->mask) { for (long long k = 0; k < (long long)G.Nx * G.Ny; k++) if (g->mask[k] == 3) { any3 = true; break; } }
    else any3 = (!g->periodic_x || !g->periodic_y);
    for (long long k = 0; k < n; k++) {
        signed char mk = c->h_mask[k];
        unsigned char f = 0;
        if (mk == 1) f |= PF_STEPPED;
        if (mk == 3 && m->periodic_boundary) f |= PF_STEPPED | PF_GROUP2;
        bool bnd = m->periodic_boundary ? (mk == 2) : (mk >= 2);
        if (bnd) f |= PF_BOUNDARY;
        pf[k] = f;
    }
    if (any3 && m->periodic_boundary) G.ngroups = 2;
    if ((long long)(G.ny_loc + 2 * G.R) * 6 * G.Nx >= (1LL << 31)) {
        g_create_error = "slab too large for 32-bit record offsets ((ny_loc + 2 halo_rows) * 6 * Nx must be < 2^31): use more slabs";
        delete c;
        return -2;
    }
    /* a whole-grid context follows any reach (a reach that wraps around a periodic axis takes the general
     * pull); a slab covers halo_rows of reach, and its periodic y axis must be longer than 2*halo_rows */
    if (!G.single_slab && ((G.periodic_y && G.Ny <= 2 * G.R) || G.ny_loc < G.R)) {
        g_create_error = "slab: periodic y axis not longer than 2*halo_rows, or fewer own rows than halo_rows";
        delete c;
        return -2;
    }

#define CK(call) do { hipError_t e2 = (call); if (e2 != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(e2); picles_destroy(c); return -10; } } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&c->ev_edge, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->ev_ctx, hipEventDisableTiming));
    Arrays &A = c->A;
    memset(&A, 0, sizeof(A));
    A.n = n;
    CK(hipMalloc(&A.state, 3 * n * 8)); CK(hipMalloc(&A.movie, 3 * n * 8));
    CK(hipMalloc(&A.z, 5 * n * 8));
    CK(hipMalloc(&A.qold, n * 8)); CK(hipMalloc(&A.dtn, n * 8)); CK(hipMalloc(&A.asw, n * 4)); CK(hipMemset(A.asw, 0, n * 4));
    CK(hipMalloc(&A.on, n)); CK(hipMalloc(&A.pflags, n)); CK(hipMalloc(&A.status, n * 4));
    CK(hipMalloc(&A.u0, n * 8)); CK(hipMalloc(&A.v0, n * 8)); CK(hipMalloc(&A.u1, n * 8)); CK(hipMalloc(&A.v1, n * 8));
    CK(hipMalloc(&A.cnt, NSLOTS * sizeof(DevCounters)));
    CK(hipMalloc(&A.max_reach_total, sizeof(int)));
    CK(hipMemset(A.max_reach_total, 0, sizeof(int)));
    for (int k = 0; k < 3; k++) {
        CK(hipMalloc(&c->mr_buf[k], sizeof(int)));
        CK(hipMemset(c->mr_buf[k], 0, sizeof(int)));
    }
    for (int k = 0; k < 2; k++) {
        CK(hipMalloc(&c->rec_buf[k], rec_bytes(c)));
        CK(hipMemset(c->rec_buf[k], 0, rec_bytes(c)));
    }
    CK(hipMalloc(&c->d_mask, n));
    CK(hipMemset(A.state, 0, 3 * n * 8)); CK(hipMemset(A.movie, 0, 3 * n * 8)); CK(hipMemset(A.z, 0, 5 * n * 8));
    CK(hipMemset(A.qold, 0, n * 8)); CK(hipMemset(A.dtn, 0, n * 8)); CK(hipMemset(A.on, 0, n)); CK(hipMemset(A.status, 0, n * 4));
    CK(hipMemset(A.u0, 0, n * 8)); CK(hipMemset(A.v0, 0, n * 8)); CK(hipMemset(A.u1, 0, n * 8)); CK(hipMemset(A.v1, 0, n * 8));
    CK(hipMemset(A.cnt, 0, NSLOTS * sizeof(DevCounters)));
    CK(hipMemcpy(A.pflags, pf.data(), n, hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_mask, c->h_mask.data(), n, hipMemcpyHostToDevice));
#undef CK
    c->state_zero = true;
    *out = c;
    return 0;
}

PX_EXPORT int32_t picles_destroy(picles_ctx *c)
{
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->ring) picles_slab_comm_destroy(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    Arrays &A = c->A;
    hipFree(A.state); hipFree(A.movie); hipFree(A.z); hipFree(A.qold); hipFree(A.dtn); hipFree(A.asw); hipFree(A.on);
    hipFree(A.pflags); hipFree(A.status); hipFree(A.u0); hipFree(A.v0); hipFree(A.u1); hipFree(A.v1);
    if (A.uP) { hipFree(A.uP); hipFree(A.vP); }
    hipFree(A.cnt); hipFree(A.max_reach_total); hipFree(c->d_mask);
    if (A.m11) { hipFree(A.m11); hipFree(A.m22); hipFree(A.pc); }
    for (int k = 0; k < 2; k++) hipFree(c->rec_buf[k]);
    for (int k = 0; k < 3; k++) hipFree(c->mr_buf[k]);
    for (auto p : c->store_dev) hipFree(p);
    for (auto p : c->store_host) hipHostFree(p);
    for (auto e : c->store_ready) hipEventDestroy(e);
    for (auto e : c->store_done) hipEventDestroy(e);
    if (c->store_stream) hipStreamDestroy(c->store_stream);
    if (c->d_wgu) hipFree(c->d_wgu);
    if (c->d_wgv) hipFree(c->d_wgv);
    if (c->d_count) hipFree(c->d_count);
    if (c->d_start) hipFree(c->d_start);
    if (c->d_cursor) hipFree(c->d_cursor);
    if (c->d_scan_tmp) hipFree(c->d_scan_tmp);
    for (auto &e : c->ev_used) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto &e : c->ev_free) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    if (c->ev_edge) hipEventDestroy(c->ev_edge);
    if (c->ev_ctx) hipEventDestroy(c->ev_ctx);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

PX_EXPORT int32_t picles_sync(picles_ctx *c)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());
    return 0;
}

PX_EXPORT double picles_clock(const picles_ctx *c) { return c ? c->clock : 0.0; }

PX_EXPORT int32_t picles_set_winds(picles_ctx *c, const double *u0, const double *v0, double t0,
                                   const double *u1, const double *v1, double t1)
{
    if (!c || !u0 || !v0) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());   /* the previous step may still read the wind planes */
    c->wind_grid_on = false;
    size_t b = (size_t)c->A.n * 8;
    HIPCHK(c, hipMemcpyAsync(c->A.u0, u0, b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->A.v0, v0, b, hipMemcpyHostToDevice, c->stream));
    if (u1 && v1 && t1 != t0) {
        HIPCHK(c, hipMemcpyAsync(c->A.u1, u1, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->A.v1, v1, b, hipMemcpyHostToDevice, c->stream));
        c->P.wind_static = 0;
        c->P.tw0 = t0;
        c->P.inv_dtw = 1.0 / (t1 - t0);
    } else {
        c->P.wind_static = 1;
        c->P.tw0 = t0;
        c->P.inv_dtw = 0.0;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   /* caller may reuse its host buffers */
    return 0;
}

static inline unsigned nblocks(long long n, int b) { return (unsigned)((n + b - 1) / b); }

/* per-node ProjetionKernel diagonal + PropagationCorrection coefficient */
PX_EXPORT int32_t picles_set_metric(picles_ctx *c, const double *m11, const double *m22, const double *pc)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());
    Arrays &A = c->A;
    if (A.m11) { hipFree(A.m11); hipFree(A.m22); hipFree(A.pc); A.m11 = A.m22 = A.pc = nullptr; }
    if (!m11 || !m22 || !pc) return 0;   /* back to the Cartesian constants */
    size_t b = (size_t)A.n * 8;
    HIPCHK(c, hipMalloc(&A.m11, b)); HIPCHK(c, hipMalloc(&A.m22, b)); HIPCHK(c, hipMalloc(&A.pc, b));
    HIPCHK(c, hipMemcpy(A.m11, m11, b, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(A.m22, m22, b, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(A.pc, pc, b, hipMemcpyHostToDevice));
    return 0;
}

PX_EXPORT int32_t picles_set_wind_grid(picles_ctx *c, int32_t nx, int32_t ny, int32_t nt,
                                       double x0, double dx, double y0, double dy, double t0, double dt,
                                       const double *u, const double *v, double mesh_x0, double mesh_y0)
{
    if (!c || !u || !v) return -1;
    if (nx < 2 || ny < 2 || nt < 2 || !(dx > 0) || !(dy > 0) || !(dt > 0)) return fail(c, -2, "wind lattice needs >= 2 knots per axis and positive spacing");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());
    if (c->d_wgu) { hipFree(c->d_wgu); hipFree(c->d_wgv); c->d_wgu = c->d_wgv = nullptr; }
    size_t nb = (size_t)nx * ny * nt * 8;
    HIPCHK(c, hipMalloc(&c->d_wgu, nb));
    HIPCHK(c, hipMalloc(&c->d_wgv, nb));
    HIPCHK(c, hipMemcpy(c->d_wgu, u, nb, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_wgv, v, nb, hipMemcpyHostToDevice));
    WindGrid &w = c->wg;
    w.nx = nx; w.ny = ny; w.nt = nt;
    w.x0 = x0; w.inv_dx = 1.0 / dx; w.y0 = y0; w.inv_dy = 1.0 / dy; w.t0 = t0; w.inv_dt = 1.0 / dt;
    w.mesh_x0 = mesh_x0; w.mesh_y0 = mesh_y0; w.mesh_dx = c->g.dx; w.mesh_dy = c->g.dy;
    w.u = c->d_wgu; w.v = c->d_wgv;
    c->wind_grid_on = true;
    c->wind_t1_valid = false;
    return 0;
}

/* sample the lattice for the step [t, t+dt] into (u0,v0) / (u1,v1); reuses the level the previous
 * step left in (u1,v1) by swapping the plane pointers */
static int wind_grid_prepare(picles_ctx *c, double t, double dt, hipStream_t s)
{
    Arrays &A = c->A;
    dim3 grid(nblocks(A.n, 256)), block(256);
    if (c->wind_t1_valid && c->wind_t1 == t) {
        std::swap(A.u0, A.u1);
        std::swap(A.v0, A.v1);
    } else {
        hipLaunchKernelGGL(k_wind_sample, grid, block, 0, s, c->G, c->wg, t, A.u0, A.v0, A.n);
    }
    hipLaunchKernelGGL(k_wind_sample, grid, block, 0, s, c->G, c->wg, t + dt, A.u1, A.v1, A.n);
    HIPCHK(c, hipGetLastError());
    c->wind_t1 = t + dt;
    c->wind_t1_valid = true;
    c->P.wind_static = 0;
    c->P.tw0 = t;
    c->P.inv_dtw = 1.0 / dt;
    return 0;
}

PX_EXPORT int32_t picles_get_winds(picles_ctx *c, double *u0, double *v0, double *u1, double *v1)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    size_t b = (size_t)c->A.n * 8;
    if (u0) HIPCHK(c, hipMemcpy(u0, c->A.u0, b, hipMemcpyDeviceToHost));
    if (v0) HIPCHK(c, hipMemcpy(v0, c->A.v0, b, hipMemcpyDeviceToHost));
    if (u1) HIPCHK(c, hipMemcpy(u1, c->A.u1, b, hipMemcpyDeviceToHost));
    if (v1) HIPCHK(c, hipMemcpy(v1, c->A.v1, b, hipMemcpyDeviceToHost));
    return 0;
}

PX_EXPORT int32_t picles_seed(picles_ctx *c, double t0)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->clock = t0;
    if (c->wind_grid_on) {   /* winds at t = 0.0 seed the particles (run.jl:213-215) */
        c->wind_t1_valid = false;
        int rc = wind_grid_prepare(c, 0.0, c->od.timestep, c->stream);
        if (rc) return rc;
    }
    c->pending = false;
    c->cur = 0;
    c->mr_w = 0;
    for (int k = 0; k < 2; k++) {
        HIPCHK(c, hipMemsetAsync(c->rec_buf[k], 0, rec_bytes(c), c->stream));
        HIPCHK(c, hipMemsetAsync(c->mr_buf[k], 0, sizeof(int), c->stream));
        if (k == 0) HIPCHK(c, hipMemsetAsync(c->mr_buf[2], 0, sizeof(int), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->A.cnt, 0, NSLOTS * sizeof(DevCounters), c->stream));
    HIPCHK(c, hipMemsetAsync(c->A.max_reach_total, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_seed, dim3(nblocks(c->A.n, 256)), dim3(256), 0, c->stream, c->P, c->G, arrays_for(c, 0, 0), c->d_mask, c->od.timestep);
    HIPCHK(c, hipGetLastError());
    c->state_zero = false;
    c->seeded = true;
    return 0;
}

PX_EXPORT int32_t picles_zero_state(picles_ctx *c)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipMemsetAsync(c->A.state, 0, 3 * c->A.n * 8, c->stream));
    c->state_zero = true;
    return 0;
}

PX_EXPORT int32_t picles_tick(picles_ctx *c, double dt)
{
    if (!c) return -1;
    c->clock += dt;
    return 0;
}

PX_EXPORT int32_t picles_begin_step(picles_ctx *c, double dt, int32_t flags)
{
    if (!c) return -1;
    if (!(dt > 0.0)) return fail(c, -2, "dt must be positive");
    { int rc = flush(c); if (rc) return rc; }
    c->step_dt = dt;
    c->step_flags = flags;
    c->edge_pending = false;
    c->cur ^= 1;            /* this step's records go to (and are scattered from) rec_buf[cur] */
    c->mr_w = (c->mr_w + 1) % 3;
    c->step_fresh = true;   /* the first advance_rows of the step clears max_reach on ITS stream */
    if (c->wind_grid_on) {
        HIPCHK(c, hipSetDevice(c->device));
        if (c->ext_streams && !c->ring_orders) HIPCHK(c, hipDeviceSynchronize());   /* previous step (any stream) done with the wind planes */
        int rc = wind_grid_prepare(c, c->clock, dt, c->stream);
        if (rc) return rc;
        /* caller-stream launches wait for the sampler through ev_ctx (step_prologue) */
    }
    return 0;
}

static int select_rows(picles_ctx *c, int which, int &r0, int &n0, int &r1, int &n1)
{
    const GridP &G = c->G;
    int R = G.R;
    r0 = n0 = r1 = n1 = 0;
    bool small = G.ny_loc <= 2 * R;
    if (which == PICLES_ROWS_ALL) { n0 = G.ny_loc; }
    else if (which == PICLES_ROWS_EDGE) {
        if (small) n0 = G.ny_loc;
        else { n0 = R; r1 = G.ny_loc - R; n1 = R; }
    } else if (which == PICLES_ROWS_INTERIOR) {
        if (!small) { r0 = R; n0 = G.ny_loc - 2 * R; }
    } else return fail(c, -2, "bad row selector");
    return 0;
}

/* Launches of one step may come on caller-provided streams.  Whatever the library enqueued on its own stream before (the
 * scatter + remesh of a flushed step, a wind-lattice sample, the seed) must be complete before a caller-stream kernel reads
 * it: the caller stream waits for an event recorded on the context stream.  (The step's reach counter needs no clearing
 * here: three counters rotate and the previous step's launches cleared this one — Arrays::max_reach_next.) */
static int step_prologue(picles_ctx *c, hipStream_t s)
{
    if (s != c->stream && !stream_idle(c->stream)) {     /* an idle context stream has nothing to wait for */
        HIPCHK(c, hipEventRecord(c->ev_ctx, c->stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_ctx, 0));
    }
    return 0;
}

PX_EXPORT int32_t picles_advance_rows(picles_ctx *c, int32_t which, void *stream)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (stream) c->ext_streams = true;
    int r0, n0, r1, n1;
    int rc = select_rows(c, which, r0, n0, r1, n1);
    if (rc) return rc;
    long long nt = (long long)(n0 + n1) * c->G.Nx;
    if (nt == 0) return 0;
    if ((rc = step_prologue(c, s))) return rc;
    timing_begin(c, s, 0);
    {
        const KParams &P = c->P;
        bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.deadband2 == 0.0;
        dim3 grid(nblocks(nt, 256)), block(256);
        Arrays A = arrays_for(c, c->cur, c->cur);
#define LAUNCH_ADV2(F, S, M) do { if (P.solver == 2) hipLaunchKernelGGL((k_advance<F, S, M, true, true>), grid, block, 0, s, c->P, c->G, A, c->clock, c->step_dt, r0, n0, r1, n1); \
                                   else if (P.solver) hipLaunchKernelGGL((k_advance<F, S, M, true, false>), grid, block, 0, s, c->P, c->G, A, c->clock, c->step_dt, r0, n0, r1, n1); \
                                   else hipLaunchKernelGGL((k_advance<F, S, M, false, false>), grid, block, 0, s, c->P, c->G, A, c->clock, c->step_dt, r0, n0, r1, n1); } while (0)
#define LAUNCH_ADV(F, S, M) LAUNCH_ADV2(F, S, M)
        if (c->A.pc) LAUNCH_ADV(false, false, true);   /* per-node metric: the general code path */
        else if (fast && P.wind_static) LAUNCH_ADV(true, true, false);
        else if (fast) LAUNCH_ADV(true, false, false);
        else LAUNCH_ADV(false, false, false);          /* general physics; static winds are the du = dv = 0 case of the same code (same bits) */
#undef LAUNCH_ADV2
#undef LAUNCH_ADV
    }
    timing_end(c, s);
    HIPCHK(c, hipGetLastError());
    if (which == PICLES_ROWS_EDGE && s != c->stream) {
        HIPCHK(c, hipEventRecord(c->ev_edge, s));
        c->edge_pending = true;
    }
    return 0;
}

/* can this step ride on fused k_step launches? (run!-style: State zeroed first, static winds) */
static bool step_fusable(const picles_ctx *c, int flags)
{
    if (flags != PICLES_STEP_ZERO_FIRST || !c->fuse_steps) return false;
    const KParams &P = c->P;
    const bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.deadband2 == 0.0;
    /* the time-varying-wind and per-node-metric flavours of the fused kernel exist for the specialised physics */
    if (c->wind_grid_on) return fast;
    if (c->A.pc) return fast && P.wind_static != 0;
    if (P.solver == 2) return fast && P.wind_static != 0;   /* the auto-switching flavour exists for the specialised physics */
    return P.wind_static != 0;
}

/* fused phase launcher: scatter+remesh of the pending step and advance of the current one for the
 * selected rows (records of the pending step: rec_buf[1-cur], of this step: rec_buf[cur]) */
static int launch_step_rows(picles_ctx *c, int which, hipStream_t s)
{
    int r0, n0, r1, n1;
    int rc = select_rows(c, which, r0, n0, r1, n1);
    if (rc) return rc;
    long long nt = (long long)(n0 + n1) * c->G.Nx;
    if (nt == 0) return 0;
    if ((rc = step_prologue(c, s))) return rc;
    const KParams &P = c->P;
    Arrays A = arrays_for(c, c->cur ^ 1, c->cur);
    /* specialised variant: every physics switch on and n = 2 (all reference scripts) */
    bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.deadband2 == 0.0;
    dim3 grid(nblocks(nt, 256)), block(256);
    timing_begin(c, s, 0);
#define LAUNCH_STEP(F, T, S, M, AU) hipLaunchKernelGGL((k_step<F, T, S, M, AU>), grid, block, 0, s, c->P, c->G, A, c->pend_t, c->pend_dt, c->clock, c->step_dt, r0, n0, r1, n1)
    if (fast) {     /* solver (DP5 / Tsit5 / auto-switching) x static winds x per-node metric: 12 flavours */
        const int key = (P.solver == 2 ? 8 : (P.solver ? 4 : 0)) | (P.wind_static ? 2 : 0) | (c->A.pc ? 1 : 0);
        switch (key) {
#define CASE_STEP(k, T, S, M, AU) case k: LAUNCH_STEP(true, T, S, M, AU); break;
            CASE_STEP(0, false, false, false, false) CASE_STEP(1, false, false, true, false)
            CASE_STEP(2, false, true, false, false)  CASE_STEP(3, false, true, true, false)
            CASE_STEP(4, true, false, false, false)  CASE_STEP(5, true, false, true, false)
            CASE_STEP(6, true, true, false, false)   CASE_STEP(7, true, true, true, false)
            CASE_STEP(8, true, false, false, true)   CASE_STEP(9, true, false, true, true)
            CASE_STEP(10, true, true, false, true)   CASE_STEP(11, true, true, true, true)
#undef CASE_STEP
        }
    }
    else if (P.solver) LAUNCH_STEP(false, true, true, false, false);      /* general physics: static winds, Cartesian (step_fusable) */
    else LAUNCH_STEP(false, false, true, false, false);
#undef LAUNCH_STEP
    timing_end(c, s);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* Slab form of the fused step.  Per model step:
 *   picles_begin_fused_step(dt)            (returns 1 if the step cannot be fused: use the plain phases)
 *   picles_step_rows(EDGE, stream_E)  ->  exchange halo blocks  ||  picles_step_rows(INTERIOR, stream_M)
 *   picles_end_fused_step()                (ticks the clock; scatter+remesh of this step stay pending) */
PX_EXPORT int32_t picles_begin_fused_step(picles_ctx *c, double dt)
{
    if (!c) return -1;
    if (!(dt > 0.0)) return fail(c, -2, "dt must be positive");
    if (!step_fusable(c, PICLES_STEP_ZERO_FIRST)) return 1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->wind_grid_on) {
        /* device-sampled winds.  With a step pending, its remesh (done by this step's launches) needs the wind
         * at ITS clock: level 0 of its window.  The three level planes rotate — previous level 0 -> (uP, vP),
         * previous level 1 -> level 0 — and only the new level 1 is sampled. */
        Arrays &A = c->A;
        if (c->pending && !(c->wind_t1_valid && c->wind_t1 == c->clock && c->P.tw0 == c->pend_t && !c->P.wind_static))
            return 1;   /* the windows are not contiguous: take the plain phases (they flush first) */
        /* earlier launches on other streams read the planes (the native slab ring orders its own streams with events) */
        if (c->ext_streams && !c->ring_orders) HIPCHK(c, hipDeviceSynchronize());
        if (c->pending) {
            if (!A.uP) {
                HIPCHK(c, hipMalloc(&A.uP, (size_t)A.n * 8));
                HIPCHK(c, hipMalloc(&A.vP, (size_t)A.n * 8));
            }
            double *tu = A.uP, *tv = A.vP;
            A.uP = A.u0; A.vP = A.v0;
            A.u0 = A.u1; A.v0 = A.v1;
            A.u1 = tu; A.v1 = tv;
            dim3 grid(nblocks(A.n, 256)), block(256);
            hipLaunchKernelGGL(k_wind_sample, grid, block, 0, c->stream, c->G, c->wg, c->clock + dt, A.u1, A.v1, A.n);
            HIPCHK(c, hipGetLastError());
            c->wind_t1 = c->clock + dt;
            c->P.tw0 = c->clock;
            c->P.inv_dtw = 1.0 / dt;
        } else {
            int rc = wind_grid_prepare(c, c->clock, dt, c->stream);
            if (rc) return rc;
        }
        if (c->ext_streams && !c->ring_orders) HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->step_dt = dt;
    c->step_flags = PICLES_STEP_ZERO_FIRST;
    c->edge_pending = false;
    c->cur ^= 1;
    c->mr_w = (c->mr_w + 1) % 3;
    c->step_fresh = true;
    return 0;
}

PX_EXPORT int32_t picles_step_rows(picles_ctx *c, int32_t which, void *stream)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (stream) c->ext_streams = true;
    if (!c->pending) return picles_advance_rows(c, which, stream);   /* first step: nothing to scatter yet */
    return launch_step_rows(c, which, s);
}

PX_EXPORT int32_t picles_end_fused_step(picles_ctx *c)
{
    if (!c) return -1;
    c->pending = true;
    c->pend_t = c->clock;
    c->pend_dt = c->step_dt;
    c->state_zero = false;
    c->edge_pending = false;
    c->clock += c->step_dt;
    return 0;
}

static int launch_scatter(picles_ctx *c, hipStream_t s, bool remesh)
{
    Arrays A = arrays_for(c, c->cur, c->cur);
    int flags = c->step_flags;
    bool movie = (flags & PICLES_STEP_MOVIE) != 0;
    bool zero_first = (flags & PICLES_STEP_ZERO_FIRST) != 0;
    int accum = (zero_first || c->state_zero) ? 0 : 1;
    if (flags & PICLES_STEP_ATOMIC) {
        if (!c->G.single_slab) return fail(c, -5, "PICLES_STEP_ATOMIC is single-slab only");
        if (c->G.tripolar) return fail(c, -5, "PICLES_STEP_ATOMIC does not implement the tripolar fold: use the deterministic pull");
        if (!accum) HIPCHK(c, hipMemsetAsync(c->A.state, 0, 3 * c->A.n * 8, s));
        int ntx = (c->G.Nx + PT_TX - 1) / PT_TX, nty = (c->G.ny_loc + PT_TY - 1) / PT_TY;
        timing_begin(c, s, 1);
        hipLaunchKernelGGL(k_push_tiles<true>, dim3(ntx * nty), dim3(256), 0, s, c->G, A, ntx,
                           (const int *)nullptr, (const int *)nullptr, (const int *)nullptr,
                           (const double *)nullptr, (const double *)nullptr, 0LL);
        timing_end(c, s);
        HIPCHK(c, hipGetLastError());
        if (movie) HIPCHK(c, hipMemcpyAsync(c->A.movie, c->A.state, 3 * c->A.n * 8, hipMemcpyDeviceToDevice, s));
        if (remesh) {
            timing_begin(c, s, 2);
            hipLaunchKernelGGL(k_remesh, dim3(nblocks(c->A.n, 256)), dim3(256), 0, s, c->P, c->G, A, c->clock, c->step_dt);
            timing_end(c, s);
            HIPCHK(c, hipGetLastError());
        }
        if (movie && remesh) HIPCHK(c, hipMemsetAsync(c->A.state, 0, 3 * c->A.n * 8, s));
        c->state_zero = movie && remesh;
        return 0;
    }
    timing_begin(c, s, 1);
    if (remesh)
        hipLaunchKernelGGL(k_scatter<true>, dim3(nblocks(c->A.n, 256)), dim3(256), 0, s, c->P, c->G, A, accum, movie ? 1 : 0, c->clock, c->step_dt);
    else
        hipLaunchKernelGGL(k_scatter<false>, dim3(nblocks(c->A.n, 256)), dim3(256), 0, s, c->P, c->G, A, accum, 0, c->clock, c->step_dt);
    timing_end(c, s);
    HIPCHK(c, hipGetLastError());
    c->state_zero = movie && remesh;
    return 0;
}

PX_EXPORT int32_t picles_scatter_remesh(picles_ctx *c, void *stream)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->edge_pending) { HIPCHK(c, hipStreamWaitEvent(s, c->ev_edge, 0)); c->edge_pending = false; }
    if (s != c->stream && !stream_idle(c->stream)) {   /* ordered behind whatever the library enqueued on its own stream (see step_prologue) */
        HIPCHK(c, hipEventRecord(c->ev_ctx, c->stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_ctx, 0));
    }
    int rc = launch_scatter(c, s, true);
    if (rc) return rc;
    c->clock += c->step_dt;
    return 0;
}

PX_EXPORT int32_t picles_time_step(picles_ctx *c, double dt, int32_t flags)
{
    if (!c) return -1;
    if (!c->G.single_slab) return fail(c, -5, "picles_time_step needs the whole grid; slabs use begin_step/advance_rows/scatter_remesh");
    if (!(dt > 0.0)) return fail(c, -2, "dt must be positive");
    if (step_fusable(c, flags)) {
        /* run!-style consecutive steps: one launch per step (k_step), the scatter + remesh of the
         * previous step ride along; the last one is flushed when somebody looks */
        int rc0 = picles_begin_fused_step(c, dt);
        if (rc0) return rc0 < 0 ? rc0 : fail(c, -6, "internal: fusable step refused");
        if ((rc0 = picles_step_rows(c, PICLES_ROWS_ALL, nullptr))) return rc0;
        return picles_end_fused_step(c);
    }
    int rc = picles_begin_step(c, dt, flags);
    if (rc) return rc;
    rc = picles_advance_rows(c, PICLES_ROWS_ALL, nullptr);
    if (rc) return rc;
    return picles_scatter_remesh(c, nullptr);
}

PX_EXPORT int32_t picles_run_steps(picles_ctx *c, double dt, int32_t n_steps)
{
    if (!c || n_steps < 0) return -1;
    for (int k = 0; k < n_steps; k++) {
        int rc = picles_time_step(c, dt, PICLES_STEP_ZERO_FIRST);
        if (rc) return rc;
    }
    return 0;
}

PX_EXPORT int32_t picles_advance(picles_ctx *c, double dt, int32_t flags)
{
    if (!c) return -1;
    if (!c->G.single_slab) return fail(c, -5, "picles_advance needs the whole grid");
    int rc = picles_begin_step(c, dt, flags & ~(PICLES_STEP_MOVIE));
    if (rc) return rc;
    rc = picles_advance_rows(c, PICLES_ROWS_ALL, nullptr);
    if (rc) return rc;
    return launch_scatter(c, c->stream, false);
}

PX_EXPORT int32_t picles_remesh(picles_ctx *c, double dt)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    timing_begin(c, c->stream, 2);
    hipLaunchKernelGGL(k_remesh, dim3(nblocks(c->A.n, 256)), dim3(256), 0, c->stream, c->P, c->G, arrays_for(c, c->cur, c->cur), c->clock, dt);
    timing_end(c, c->stream);
    HIPCHK(c, hipGetLastError());
    return 0;
}

/* ---- data access ---- */
static int d2h(picles_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());   /* kernels may have run on caller-provided streams */
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
static int h2d(picles_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

PX_EXPORT int32_t picles_get_state(picles_ctx *c, double *s) { return (c && s) ? d2h(c, s, c->A.state, 3 * c->A.n * 8) : -1; }
PX_EXPORT int32_t picles_get_movie_state(picles_ctx *c, double *s) { return (c && s) ? d2h(c, s, c->A.movie, 3 * c->A.n * 8) : -1; }
PX_EXPORT int32_t picles_set_state(picles_ctx *c, const double *s)
{
    if (!c || !s) return -1;
    c->state_zero = false;
    return h2d(c, c->A.state, s, 3 * c->A.n * 8);
}

PX_EXPORT int32_t picles_get_particles(picles_ctx *c, double *z, uint8_t *on, uint8_t *boundary, int32_t *status)
{
    if (!c) return -1;
    int rc = 0;
    if (z && (rc = d2h(c, z, c->A.z, 5 * c->A.n * 8))) return rc;
    if (on && (rc = d2h(c, on, c->A.on, c->A.n))) return rc;
    if (status && (rc = d2h(c, status, c->A.status, c->A.n * 4))) return rc;
    if (boundary) {
        std::vector<unsigned char> pf(c->A.n);
        if ((rc = d2h(c, pf.data(), c->A.pflags, c->A.n))) return rc;
        for (long long k = 0; k < c->A.n; k++) boundary[k] = (pf[k] & PF_BOUNDARY) ? 1 : 0;
    }
    return 0;
}

PX_EXPORT int32_t picles_set_particles(picles_ctx *c, const double *z, const uint8_t *on)
{
    if (!c) return -1;
    int rc = 0;
    if (z && (rc = h2d(c, c->A.z, z, 5 * c->A.n * 8))) return rc;
    if (on && (rc = h2d(c, c->A.on, on, c->A.n))) return rc;
    std::vector<double> neg(c->A.n, -1.0);   /* auto_dt_reset! on the next advance */
    return h2d(c, c->A.dtn, neg.data(), c->A.n * 8);
}

PX_EXPORT int32_t picles_get_counters(picles_ctx *c, picles_counters *out)
{
    if (!c || !out) return -1;
    std::vector<DevCounters> d(NSLOTS);
    int rc = d2h(c, d.data(), c->A.cnt, NSLOTS * sizeof(DevCounters));
    if (rc) return rc;
    int mr = 0, mrt = 0;
    if ((rc = d2h(c, &mr, c->mr_buf[c->mr_w], sizeof(int)))) return rc;
    if ((rc = d2h(c, &mrt, c->A.max_reach_total, sizeof(int)))) return rc;
    memset(out, 0, sizeof(*out));
    for (const DevCounters &k : d) {
        out->rhs_evals += k.rhs; out->steps_accepted += k.acc; out->steps_rejected += k.rej;
        out->reseeds += k.reseeds; out->clamps += k.clamps; out->maxiters_hits += k.maxit;
        out->particles_advanced += k.adv; out->halo_overflow += k.overflow;
        out->dropped_nonfinite += k.nonfinite;
    }
    out->max_reach = mr;
    out->max_reach_seen = mrt;
    return 0;
}

PX_EXPORT int32_t picles_reset_counters(picles_ctx *c)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemsetAsync(c->A.cnt, 0, NSLOTS * sizeof(DevCounters), c->stream));
    HIPCHK(c, hipMemsetAsync(c->A.max_reach_total, 0, sizeof(int), c->stream));
    return 0;
}

PX_EXPORT int32_t picles_enable_timing(picles_ctx *c, int32_t on)
{
    if (!c) return -1;
    { int rc = flush(c); if (rc) return rc; }
    timing_collect(c);
    c->timing = on != 0;
    if (on) { memset(&c->tim, 0, sizeof(c->tim)); for (auto &v : c->tim_samples) v.clear(); }
    return 0;
}

PX_EXPORT int32_t picles_get_timing(picles_ctx *c, picles_timing *t)
{
    if (!c || !t) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    timing_collect(c);
    *t = c->tim;
    return 0;
}

PX_EXPORT int32_t picles_get_timing_samples(picles_ctx *c, int32_t kind, double *out_ms, int32_t cap)
{
    if (!c || kind < 0 || kind > 2 || cap < 0 || (cap > 0 && !out_ms)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    timing_collect(c);
    const std::vector<float> &v = c->tim_samples[kind];
    int n = (int)std::min<size_t>(v.size(), (size_t)cap);
    for (int k = 0; k < n; k++) out_ms[k] = v[k];
    return (int32_t)v.size();
}

/* ---- snapshot ring ---- */
PX_EXPORT int32_t picles_store_init(picles_ctx *c, int32_t n_slots)
{
    if (!c || n_slots < 1) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (c->store_slots) return fail(c, -2, "store already initialised");
    size_t b = 3 * (size_t)c->A.n * 8;
    HIPCHK(c, hipStreamCreateWithFlags(&c->store_stream, hipStreamNonBlocking));
    for (int k = 0; k < n_slots; k++) {
        double *d = nullptr, *h = nullptr;
        hipEvent_t e1, e2;
        HIPCHK(c, hipMalloc(&d, b));
        HIPCHK(c, hipHostMalloc(&h, b, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
        c->store_dev.push_back(d); c->store_host.push_back(h);
        c->store_ready.push_back(e1); c->store_done.push_back(e2);
    }
    c->store_time.assign(n_slots, 0.0);
    c->store_slots = n_slots; c->store_head = 0; c->store_count = 0;
    return 0;
}

PX_EXPORT int32_t picles_store_pending(const picles_ctx *c) { return c ? c->store_count : -1; }

PX_EXPORT int32_t picles_store_push(picles_ctx *c)
{
    if (!c) return -1;
    if (!c->store_slots) return fail(c, -2, "picles_store_init first");
    if (c->store_count == c->store_slots) return fail(c, -3, "snapshot ring full: pop first");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    int slot = (c->store_head + c->store_count) % c->store_slots;
    size_t b = 3 * (size_t)c->A.n * 8;
    /* stream-ordered behind the step that produced State; the D2H leg runs beside the next steps */
    HIPCHK(c, hipMemcpyAsync(c->store_dev[slot], c->A.state, b, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->store_ready[slot], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->store_stream, c->store_ready[slot], 0));
    HIPCHK(c, hipMemcpyAsync(c->store_host[slot], c->store_dev[slot], b, hipMemcpyDeviceToHost, c->store_stream));
    HIPCHK(c, hipEventRecord(c->store_done[slot], c->store_stream));
    c->store_time[slot] = c->clock;
    c->store_count++;
    return 0;
}

PX_EXPORT int32_t picles_store_pop(picles_ctx *c, double *state, double *time)
{
    if (!c || !state) return -1;
    if (!c->store_count) return fail(c, -3, "no snapshot pending");
    HIPCHK(c, hipSetDevice(c->device));
    int slot = c->store_head;
    HIPCHK(c, hipEventSynchronize(c->store_done[slot]));
    memcpy(state, c->store_host[slot], 3 * (size_t)c->A.n * 8);
    if (time) *time = c->store_time[slot];
    c->store_head = (c->store_head + 1) % c->store_slots;
    c->store_count--;
    return 0;
}

/* ---- halo blocks ---- */
PX_EXPORT int32_t picles_halo_rows(const picles_ctx *c) { return c ? c->G.R : -1; }

PX_EXPORT int32_t picles_set_halo_rows(picles_ctx *c, int32_t r)
{
    if (!c || r < 1 || r > 1024) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->G.single_slab && ((c->G.periodic_y && c->G.Ny <= 2 * r) || c->G.ny_loc < r))
        return fail(c, -2, "slab: periodic y axis not longer than 2*halo_rows, or fewer own rows than halo_rows");
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    /* keep the records of the own rows: re-pack into the new ghost-row geometry */
    int oldR = c->G.R;
    size_t row_b = (size_t)6 * c->G.Nx * 8;
    c->G.R = r;
    c->G.Rp = c->G.single_slab ? 0 : r;
    for (int k = 0; k < 2; k++) {
        double *old = c->rec_buf[k];
        c->rec_buf[k] = nullptr;
        HIPCHK(c, hipMalloc(&c->rec_buf[k], rec_bytes(c)));
        HIPCHK(c, hipMemsetAsync(c->rec_buf[k], 0, rec_bytes(c), c->stream));
        HIPCHK(c, hipMemcpyAsync((char *)c->rec_buf[k] + (size_t)r * row_b, (char *)old + (size_t)oldR * row_b,
                                 (size_t)c->G.ny_loc * row_b, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(old));
    }
    return 0;
}

PX_EXPORT int32_t picles_set_slab_mode(picles_ctx *c, int32_t on)
{
    if (!c) return -1;
    GridP &G = c->G;
    if (!(G.j_begin == 0 && G.ny_loc == G.Ny)) return on ? 0 : fail(c, -2, "a partial slab cannot resolve the y wrap locally");
    if (c->pending || c->seeded) return fail(c, -2, "picles_set_slab_mode: call before picles_seed");
    if (on) {
        if ((G.periodic_y && G.Ny <= 2 * G.R) || G.ny_loc < G.R)
            return fail(c, -2, "slab: periodic y axis not longer than 2*halo_rows, or fewer own rows than halo_rows");
        if (G.tripolar) return fail(c, -5, "slab mode on a whole tripolar grid is not supported");
        G.single_slab = 0;
        G.Rp = G.R;
    } else {
        G.single_slab = 1;
        G.Rp = 0;
    }
    return 0;
}

static int halo_ptr(picles_ctx *c, int side, bool send, void **ptr, size_t *bytes)
{
    if (!c || !ptr || !bytes || side < 0 || side > 1) return -1;
    const GridP &G = c->G;
    if (G.ny_loc < G.R) return fail(c, -2, "slab has fewer rows than halo_rows");
    size_t row_b = (size_t)6 * G.Nx * 8;
    int row;
    if (send) row = (side == 0) ? G.R : G.ny_loc;          /* own first R rows / own last R rows */
    else row = (side == 0) ? 0 : G.ny_loc + G.R;            /* ghost rows below / above */
    *ptr = (char *)c->rec_buf[c->cur] + (size_t)row * row_b;   /* the step in flight (after picles_begin_step) */
    *bytes = (size_t)G.R * row_b;
    return 0;
}
PX_EXPORT int32_t picles_halo_send_dev(picles_ctx *c, int32_t side, void **ptr, size_t *bytes) { return halo_ptr(c, side, true, ptr, bytes); }
PX_EXPORT int32_t picles_halo_recv_dev(picles_ctx *c, int32_t side, void **ptr, size_t *bytes) { return halo_ptr(c, side, false, ptr, bytes); }

/* ------------------------------------------------------------------------------------------
 * Native slab ring: the multi-GPU model step with no interpreter in the loop.
 * RCCL is bound at run time (dlopen of librccl.so.1 — the copy already in the process if the host has
 * loaded one, e.g. PyTorch's): the library has no link-time dependency on it and single-GPU hosts never
 * touch it.  One communicator per context = per GPU = per process; rank r owns slab r of a y-ring.
 * Per model step:
 *     edge rows   k_step on stream E  ->  ncclGroup{Send hi->next, Send lo->prev, Recv lo<-prev, Recv hi<-next} on E
 *     interior    k_step on stream M      (overlaps the exchange)
 *     M waits for E (the halo has landed before the next step's launches pull from it)
 * The halo blocks are contiguous row ranges of the record buffer: sent and received in place.
 * The reference has no counterpart (TimeSteppers.jl:144-178 is a shared-memory @threads loop).
 * ---------------------------------------------------------------------------------------- */
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommFinalize) CommFinalize = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void *handle = nullptr;
};

static RcclApi *rccl_api(std::string &err)
{
    static RcclApi api;
    static bool tried = false;
    if (api.handle) return &api;
    if (tried) { err = "RCCL could not be loaded"; return nullptr; }
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;    /* the copy already loaded, if any */
    if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { err = std::string("dlopen(librccl.so.1): ") + dlerror(); return nullptr; }
#define RSYM(f) do { api.f = (decltype(api.f))dlsym(h, "nccl" #f); if (!api.f) { err = "librccl lacks nccl" #f; return nullptr; } } while (0)
    RSYM(GetUniqueId); RSYM(CommInitRank); RSYM(CommDestroy); RSYM(GroupStart); RSYM(GroupEnd); RSYM(Send); RSYM(Recv);
    RSYM(GetErrorString);
    api.CommFinalize = (decltype(api.CommFinalize))dlsym(h, "ncclCommFinalize");     /* optional (NCCL >= 2.14) */
#undef RSYM
    api.handle = h;
    return &api;
}

struct SlabRing {
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, prev = -1, next = -1;      /* -1: no neighbour on that side (open y axis) */
    hipStream_t sE = nullptr, sM = nullptr;
    hipEvent_t evE = nullptr, evM = nullptr;
    unsigned long long steps = 0, exchanged_bytes = 0;
};

#define NCCLCHK(c, R, call)                                                                     \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + (R)->api->GetErrorString(r_);                \
            return -11;                                                                         \
        }                                                                                       \
    } while (0)

PX_EXPORT int32_t picles_slab_unique_id(void *id128)
{
    if (!id128) return -1;
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) { g_create_error = err; return -11; }
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return -11; }
    static_assert(sizeof(id) == PICLES_SLAB_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

PX_EXPORT int32_t picles_slab_comm_destroy(picles_ctx *c)
{
    if (!c) return -1;
    SlabRing *R = c->ring;
    if (!R) return 0;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (R->comm) {
        const char *mode = getenv("PICLES_RING_TEARDOWN");      /* destroy (default) | finalize | keep — diagnostics */
        if (mode && !strcmp(mode, "keep")) { /* leave the communicator to process exit */ }
        else {
            if (mode && !strcmp(mode, "finalize") && R->api->CommFinalize) R->api->CommFinalize(R->comm);
            R->api->CommDestroy(R->comm);
        }
    }
    if (R->evE) hipEventDestroy(R->evE);
    if (R->evM) hipEventDestroy(R->evM);
    if (R->sE) hipStreamDestroy(R->sE);
    if (R->sM) hipStreamDestroy(R->sM);
    delete R;
    c->ring = nullptr;
    return 0;
}

PX_EXPORT int32_t picles_slab_comm_init(picles_ctx *c, const void *id128, int32_t rank, int32_t world)
{
    if (!c || !id128) return -1;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, -2, "slab ring: need 0 <= rank < world");
    if (c->ring) return fail(c, -2, "slab ring already initialised");
    if (c->G.single_slab && world > 1) return fail(c, -2, "slab ring of several ranks needs slab contexts (j_begin, j_end)");
    if (c->G.tripolar) return fail(c, -5, "slab ring: the tripolar fold is not wired into the native ring (use picles_amd.parallel)");
    HIPCHK(c, hipSetDevice(c->device));
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(c, -11, err);
    SlabRing *R = new SlabRing();
    R->api = api; R->rank = rank; R->world = world;
    const bool per = c->G.periodic_y;
    R->prev = (rank > 0) ? rank - 1 : (per ? world - 1 : -1);
    R->next = (rank < world - 1) ? rank + 1 : (per ? 0 : -1);
    if (c->G.single_slab) R->prev = R->next = -1;       /* a whole-grid context wraps locally: nothing to exchange */
    c->ring = R;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = api->CommInitRank(&R->comm, world, id, rank);
    if (r != ncclSuccess) {
        c->err = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
        R->comm = nullptr;
        picles_slab_comm_destroy(c);
        return -11;
    }
    /* (stream priorities were tried for the edge chain and measured slower on MI355X: 0.65 vs 0.46 ms per step at 1448²) */
    HIPCHK(c, hipStreamCreateWithFlags(&R->sE, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&R->sM, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&R->evE, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&R->evM, hipEventDisableTiming));
    return 0;
}

/* the halo exchange of the step in flight, in place, on stream s */
static int ring_exchange(picles_ctx *c, SlabRing *R, hipStream_t s)
{
    if (R->prev < 0 && R->next < 0) return 0;
    void *s_lo, *s_hi, *r_lo, *r_hi;
    size_t b = 0;
    int rc;
    if ((rc = halo_ptr(c, 0, true, &s_lo, &b)) || (rc = halo_ptr(c, 1, true, &s_hi, &b)) ||
        (rc = halo_ptr(c, 0, false, &r_lo, &b)) || (rc = halo_ptr(c, 1, false, &r_hi, &b))) return rc;
    const size_t n = b / 8;
    /* order matters when prev == next (two ranks on a periodic axis, or the ring of one): sends [hi -> next, lo -> prev]
     * pair with the peer's recvs [lo <- prev, hi <- next] */
    NCCLCHK(c, R, R->api->GroupStart());
    if (R->next >= 0) NCCLCHK(c, R, R->api->Send(s_hi, n, ncclDouble, R->next, R->comm, s));
    if (R->prev >= 0) NCCLCHK(c, R, R->api->Send(s_lo, n, ncclDouble, R->prev, R->comm, s));
    if (R->prev >= 0) NCCLCHK(c, R, R->api->Recv(r_lo, n, ncclDouble, R->prev, R->comm, s));
    if (R->next >= 0) NCCLCHK(c, R, R->api->Recv(r_hi, n, ncclDouble, R->next, R->comm, s));
    NCCLCHK(c, R, R->api->GroupEnd());
    R->exchanged_bytes += (size_t)((R->next >= 0) + (R->prev >= 0)) * b;
    return 0;
}

PX_EXPORT int32_t picles_slab_exchange(picles_ctx *c)
{
    if (!c) return -1;
    SlabRing *R = c->ring;
    if (!R) return fail(c, -2, "picles_slab_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    int rc = ring_exchange(c, R, R->sE);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(R->sE));
    return 0;
}

PX_EXPORT int32_t picles_slab_run_steps(picles_ctx *c, double dt, int32_t n_steps, int32_t flags)
{
    if (!c || n_steps < 0) return -1;
    SlabRing *R = c->ring;
    if (!R) return fail(c, -2, "picles_slab_comm_init first");
    if (!(dt > 0.0)) return fail(c, -2, "dt must be positive");
    if (flags & PICLES_STEP_ATOMIC) return fail(c, -5, "PICLES_STEP_ATOMIC is single-slab only");
    HIPCHK(c, hipSetDevice(c->device));
    c->ext_streams = true;
    c->ring_orders = true;
    struct Off { picles_ctx *c; ~Off() { c->ring_orders = false; } } off{c};
    for (int k = 0; k < n_steps; k++) {
        /* the context stream (wind-lattice sampler of this step) must come after the previous step's launches on E and
         * M, and this step's launches after it (step_prologue: ev_ctx); a flush synchronises the device by itself */
        HIPCHK(c, hipEventRecord(R->evM, R->sM));
        if (c->wind_grid_on) {
            HIPCHK(c, hipEventRecord(R->evE, R->sE));
            HIPCHK(c, hipStreamWaitEvent(c->stream, R->evM, 0));
            HIPCHK(c, hipStreamWaitEvent(c->stream, R->evE, 0));
        }
        int fused = (flags == PICLES_STEP_ZERO_FIRST) ? picles_begin_fused_step(c, dt) : 1;
        if (fused < 0) return fused;
        if (fused == 1) { int rc = picles_begin_step(c, dt, flags); if (rc) return rc; }
        /* the previous step's interior launch (stream M) wrote / read what the edge launch touches */
        HIPCHK(c, hipStreamWaitEvent(R->sE, R->evM, 0));
        int rc = (fused == 0) ? picles_step_rows(c, PICLES_ROWS_EDGE, R->sE) : picles_advance_rows(c, PICLES_ROWS_EDGE, R->sE);
        if (rc) return rc;
        if ((rc = ring_exchange(c, R, R->sE))) return rc;
        rc = (fused == 0) ? picles_step_rows(c, PICLES_ROWS_INTERIOR, R->sM) : picles_advance_rows(c, PICLES_ROWS_INTERIOR, R->sM);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(R->evE, R->sE));
        HIPCHK(c, hipStreamWaitEvent(R->sM, R->evE, 0));      /* stream M waits for the edge rows and the halo */
        c->edge_pending = false;
        rc = (fused == 0) ? picles_end_fused_step(c) : picles_scatter_remesh(c, R->sM);
        if (rc) return rc;
        R->steps++;
    }
    return 0;
}

PX_EXPORT int32_t picles_slab_streams(picles_ctx *c, void **edge, void **interior)
{
    if (!c || !c->ring) return -1;
    if (edge) *edge = c->ring->sE;
    if (interior) *interior = c->ring->sM;
    return 0;
}

/* ---- generic push_to_grid! of a particle list ---- */
PX_EXPORT int32_t picles_scatter_particles(picles_ctx *c, int64_t np, const int32_t *ij, const double *xy, const double *charge)
{
    if (!c || np < 0 || (np > 0 && (!ij || !xy || !charge))) return -1;
    if (!c->G.single_slab) return fail(c, -5, "picles_scatter_particles is single-slab only");
    if (c->G.tripolar) return fail(c, -5, "picles_scatter_particles does not implement the tripolar fold");
    if (np == 0) return 0;
    if (np > 0x7fffffffLL) return fail(c, -2, "too many particles for one call");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    hipStream_t s = c->stream;
    int ntx = (c->G.Nx + PT_TX - 1) / PT_TX, nty = (c->G.ny_loc + PT_TY - 1) / PT_TY;
    int ntiles = ntx * nty;
    int *d_ij = nullptr, *d_tile = nullptr, *d_perm = nullptr;
    double *d_xy = nullptr, *d_ch = nullptr;
    struct Tmp {   /* released on every exit path */
        int *&a, *&b, *&c2; double *&d, *&e;
        ~Tmp() { hipFree(a); hipFree(b); hipFree(c2); hipFree(d); hipFree(e); }
    } tmp{d_ij, d_tile, d_perm, d_xy, d_ch};
    HIPCHK(c, hipMalloc(&d_ij, 2 * np * 4)); HIPCHK(c, hipMalloc(&d_tile, np * 4)); HIPCHK(c, hipMalloc(&d_perm, np * 4));
    HIPCHK(c, hipMalloc(&d_xy, 2 * np * 8)); HIPCHK(c, hipMalloc(&d_ch, 3 * np * 8));
    if (!c->d_count) {
        HIPCHK(c, hipMalloc(&c->d_count, (ntiles + 1) * 4));
        HIPCHK(c, hipMalloc(&c->d_start, (ntiles + 1) * 4));
        HIPCHK(c, hipMalloc(&c->d_cursor, (ntiles + 1) * 4));
        hipcub::DeviceScan::ExclusiveSum(nullptr, c->scan_tmp_bytes, c->d_count, c->d_start, ntiles + 1, s);
        HIPCHK(c, hipMalloc(&c->d_scan_tmp, c->scan_tmp_bytes));
    }
    HIPCHK(c, hipMemcpyAsync(d_ij, ij, 2 * np * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_xy, xy, 2 * np * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(d_ch, charge, 3 * np * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(c->d_count, 0, (ntiles + 1) * 4, s));
    HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, (ntiles + 1) * 4, s));
    hipLaunchKernelGGL(k_tile_count, dim3(nblocks(np, 256)), dim3(256), 0, s, c->G, ntx, d_ij, (long long)np, c->d_count, d_tile);
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->d_scan_tmp, c->scan_tmp_bytes, c->d_count, c->d_start, ntiles + 1, s));
    hipLaunchKernelGGL(k_tile_fill, dim3(nblocks(np, 256)), dim3(256), 0, s, (long long)np, d_tile, c->d_start, c->d_cursor, d_perm);
    timing_begin(c, s, 1);
    hipLaunchKernelGGL(k_push_tiles<false>, dim3(ntiles), dim3(256), 0, s, c->G, arrays_for(c, c->cur, c->cur), ntx, c->d_start, d_perm, d_ij, d_xy, d_ch, (long long)np);
    timing_end(c, s);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    c->state_zero = false;
    return 0;
}
