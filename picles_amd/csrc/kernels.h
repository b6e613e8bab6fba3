/*
 * kernels.h — device-side code shared by the translation units of libpicles_hip.so: the kernel argument structs (GridP, Arrays),
 * the per-particle step (advance_core, advance_guards, write_record, flush_stats), NodeToParticle! in registers and the deterministic pull
 * scatter.  Everything here is __device__ __forceinline__ or a plain struct; the kernels themselves live in
 *   picles_hip.hip      k_seed, k_scatter, k_remesh, k_push_tiles, the cell list, k_wind_sample + the host side / C ABI
 *   k_step_explicit.hip  the fused step, DP5 and Tsit5 flavours          k_step_auto.hip   ... AutoTsit5(Rosenbrock23())
 *   k_advance.hip        the stand-alone advance
 * (split so that the flavours compile in parallel: `make -j`).
 */
#ifndef PICLES_KERNELS_H
#define PICLES_KERNELS_H

#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/picles_hip.h"
#include "physics.h"

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
    unsigned long long nonfinite, wslots, pad_[6];     /* two 64-B lines per slot; wslots: 64 x the wave's largest attempt count */
};

struct Arrays {
    double *state, *movie;   /* 3 planes */
    double *z;               /* 5 planes */
    double *qold, *dtn;
    int *asw;                /* solver 2: AutoSwitch state (counter << 1 | rosenbrock_active; ASW_FRESH after a reinit!) */
    unsigned char *on, *pflags;
    int *status;
    double *u0, *v0, *u1, *v1;
    double *um, *vm;         /* mid-window level of three-level winds (picles_set_winds3); NULL: two levels, linear in t */
    double *uP, *vP;         /* level-0 winds of the previous step's window (fused steps under time-varying winds) */
    double *m11, *m22, *pc;  /* per-node projection diag and great-circle coefficient (NULL: Cartesian) */
    double *rec;             /* records the scatter reads  (latest completed advance) */
    double *rec_out;         /* records the advance writes (the other buffer of the pair) */
    DevCounters *cnt;        /* [NSLOTS] */
    /* scatter-reach counters: ONE base pointer and a packed index word instead of four pointers — every pointer that is live
     * across the RK loop costs two of the ~100 scalar registers, and the spill code of scalars is VALU work inside the loop
     * (measured: 44 -> 55 spilled SGPRs made the BASELINE step 2.3 % slower).  The counters sit behind the statistics slots
     * (reach_counters(A) = (int *)(A.cnt + NSLOTS)): mr[0..4] rotate with the steps, mr[5] is the running maximum since the last
     * reset; mr_idx = read | written << 4 | cleared << 8. */
    int mr_idx;
    /* local scatter reach: one int per 64 consecutive particles (index space: tile = t >> 6), the largest reach any of them had
     * in the advance that wrote the records — what lets the pull of a node look at 9 candidates where its neighbourhood is calm
     * or slow and at (2R+1)² only where particles really travel R cells (BASELINE config 5: reach 4 in the strong-wind corner,
     * 1 in the calm half).  Five buffers of ntile ints rotating with the reach counters above (same three indices). */
    int ntile;
    int *rmap;
    /* cost-ordered dispatch of the fused step (whole-grid launches).  Every workgroup files its 256-node block at the end of a
     * step: blocks that integrated or re-seeded somebody from the front of a permutation, blocks with nothing to do from the back.
     * The next launch hands the blocks out in that order — busy ones first, calm ones last — when at least an eighth of them was
     * calm.  The hardware deals workgroups to its shader engines in turn and waits for the engine whose turn it is: with waves of
     * 10 and 45 µs mixed four-and-four along every row (BASELINE config 5) the wave slots were 1.8 of 3 filled on average; with
     * like next to like the deal runs smoothly.  Five rotating buffers of (busy count, calm count, order[nblk]) ints, indices as
     * for the reach counters; ord_on = 0: natural order (the previous step built no permutation: row ranges, stand-alone advance). */
    int ord_on, nblk;
    int *ord;
    long long n;             /* Nx * ny_loc */
};
__device__ __forceinline__ int *order_buf(const Arrays &A, int which) { return A.ord + (size_t)which * (size_t)(2 + A.nblk); }

__device__ __forceinline__ int *reach_counters(const Arrays &A) { return (int *)(A.cnt + NSLOTS); }

#define PF_STEPPED 1
#define PF_GROUP2 2
#define PF_BOUNDARY 4

/* sum over the 64 lanes of a wave (callers run with every lane active).  Where all lanes hold the same value — the statistics of a
 * wave whose particles took the same number of attempts: every wave of a homogeneous sea — the sum is 64 times it: two lane reads,
 * a compare and a scalar product instead of six rounds of cross-lane additions (~30 issue slots per sum, two sums per wave and step) */
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    const unsigned int lo0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
    const unsigned int hi0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32));
    const unsigned long long v0 = ((unsigned long long)hi0 << 32) | lo0;
    if (__all(v == v0)) return v0 * 64ull;
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

/* POLY: the caller can meet a polyline window (KParams::wind_nk >= 2; physics.h, wind_eval): the first segment's slope and the first
 * knot's jump come from the planes k_wind_poly laid down */
template <bool POLY = false>
__device__ __forceinline__ Wind load_wind(const KParams &P, const Arrays &A, long long t)
{
    Wind w;
    w.u0 = A.u0[t];
    w.v0 = A.v0[t];
    w.bu = 0.0;
    w.bv = 0.0;
    w.xi = (unsigned int)t;
    if (POLY && !P.wind_static && P.wind_nk > 1) {
        const double *pl = P.wind_xb + PICLES_MAX_KNOTS + t;
        w.du = pl[0]; w.dv = pl[P.wind_xn]; w.bu = pl[2 * P.wind_xn]; w.bv = pl[3 * P.wind_xn];
        return w;
    }
    if (P.wind_static) {
        w.du = 0.0;
        w.dv = 0.0;
    } else {
        const double u1 = A.u1[t], v1 = A.v1[t];
        w.du = u1 - w.u0;
        w.dv = v1 - w.v0;
        if (A.um) {
            const double um = A.um[t], vm = A.vm[t];
            if (P.wind_sk > 0.0) {   /* three levels, the middle one at a knot of the wind lattice: two straight segments (physics.h, Wind) */
                w.du = (um - w.u0) * P.wind_isk;
                w.dv = (vm - w.v0) * P.wind_isk;
                w.bu = (u1 - um) * P.wind_i1sk - w.du;
                w.bv = (v1 - vm) * P.wind_i1sk - w.dv;
            } else {                 /* three levels: curvature term of the parabola through them */
                w.bu = 2.0 * ((w.u0 + u1) - 2.0 * um);
                w.bv = 2.0 * ((w.v0 + v1) - 2.0 * vm);
            }
        }
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
 * advance! (mapping_2D.jl:118-243) of one particle held in registers: integrate / off->on test,
 * NaN / Inf / cap guards.  Shared by k_advance and the fused k_step.
 * ---------------------------------------------------------------------------------------- */
struct StepStats {
    PStats st;
    unsigned int adv, reseeds, clamps, maxit, overflow, nonfinite;
    int reach;
};

/* advance! in two halves: the step itself (on: the whole adaptive integration; off: the wind test) and the guards behind it.
 * k_step re-reads its kernel arguments between the two (kargs_reload below). */
template <bool FAST, bool STATIC, bool METRIC = false, bool TSIT = false, bool AUTO = false, bool TABS = true>
__device__ __forceinline__ int advance_core(const KParams &P, const Wind &w, Vec5 &z, int &on, double &qold,
                                            double &dtn, double t_start, double DT, StepStats &S,
                                            double m11 = 0.0, double m22 = 0.0, double pc = 0.0, int *asw = nullptr)
{
    int status = PICLES_ST_STEPPED;
    if (on) {
        S.adv = 1;
        integrate_dp5<FAST, STATIC, METRIC, TSIT, AUTO, TABS>(P, w, z, qold, dtn, t_start, DT, S.st, m11, m22, pc, asw);
        status |= S.st.status;
    } else {
        double u, v;
        wind_at<(!FAST && !STATIC)>(P, w, t_start + DT, u, v);
        if (u * u + v * v >= P.wind_min_sq) {
            reseed(P, u, v, DT, z);
            dtn = -1.0;
            on = 1;
            status |= PICLES_ST_SWITCHED_ON;
        }
    }
    return status;
}

/* WF: () -> Wind.  k_step hands in a loader instead of the wind itself: the node wind is needed behind the RK loop only by the
 * (rare) re-seeding guards, and kept in registers across the loop it is what the 128-register build spilled to scratch */
template <bool POLY = false, class WF>
__device__ __forceinline__ int advance_guards(const KParams &P, WF wind, Vec5 &z, double &dtn, double t_start, double DT,
                                              int status, StepStats &S)
{
    if (pm_isnan(z.lne) || pm_isnan(z.cx) || pm_isnan(z.cy)) {
        double u, v;
        const Wind w = wind();
        wind_at<POLY>(P, w, t_start + DT, u, v);
        reseed(P, u, v, DT, z);
        dtn = -1.0;
        status |= PICLES_ST_RESEED_NAN;
    } else if (pm_isinf(z.lne) || pm_isinf(z.cx) || pm_isinf(z.cy)) {
        double u, v;
        const Wind w = wind();
        wind_at<POLY>(P, w, t_start, u, v);
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

/* The kernel arguments of k_step as the kernarg segment lays them out (same order, natural alignment = the C struct rule;
 * tests/test_kernel_budget.py holds the offsets against the compiler's own metadata).  k_step needs about forty scalars inside
 * the RK loop and another thirty only behind it (record pointers, grid shape, guard thresholds); kept in SGPRs across the loop
 * those overflow the scalar file and are spilled into VGPR lanes — v_writelane before the loop, v_readlane behind it: VALU slots,
 * ~2.5 % of the step.  kargs_reload() hands out the argument block again behind an opaque barrier, so everything after the loop
 * is loaded from the (scalar-cached) kernarg segment when it is used instead of living through the loop. */
struct KStepArgs {
    KParams P;
    GridP G;
    Arrays A;
    double t_prev, DT_prev, t_start, DT;
    int r0, n0, r1, n1;
};
static_assert(__builtin_offsetof(KStepArgs, G) == sizeof(KParams) && __builtin_offsetof(KStepArgs, A) % 8 == 0 &&
              __builtin_offsetof(KStepArgs, t_prev) == __builtin_offsetof(KStepArgs, A) + sizeof(Arrays) &&
              __builtin_offsetof(KStepArgs, r0) == __builtin_offsetof(KStepArgs, DT) + 8,
              "KStepArgs must mirror the kernarg layout of k_step");
typedef const KStepArgs *KStepArgsPtr;
__device__ __forceinline__ KStepArgsPtr kargs_reload(void)      /* k_advance casts the result to its own argument struct */
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();      /* constant address space */
    __asm__ volatile("" : "+s"(p));      /* a pointer the compiler knows nothing about: no load through it moves above this line */
    return (KStepArgsPtr)p;              /* the compiler's address-space inference turns the loads back into scalar loads */
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
/* rtile: the reach-map tile (t >> 6) of a lane whose particle left a record this step, -1 otherwise */
__device__ __forceinline__ void flush_stats(const Arrays &A, const StepStats &S, int rtile = -1)
{
    unsigned long long s_rhs = wave_sum_u64(S.st.rhs);
    unsigned long long s_ar = wave_sum_u64(((unsigned long long)S.st.acc << 32) | S.st.rej);
    unsigned long long b_adv = __ballot(S.adv != 0), b_res1 = __ballot(S.reseeds == 1), b_res2 = __ballot(S.reseeds >= 2);
    unsigned long long b_cl = __ballot(S.clamps != 0), b_mx = __ballot(S.maxit != 0), b_ov = __ballot(S.overflow != 0);
    unsigned long long b_nf = __ballot(S.nonfinite != 0);
    /* the attempts this wave ran = its slowest lane's: a ballot ladder from lane 0's count (no cross-lane data movement; a shuffle
     * reduction is six dependent LDS permutes at the very end of the wave's life, which the three-wave kernels do not hide) */
    const int att_ = (int)(S.st.acc + S.st.rej);
    int m_att = __builtin_amdgcn_readfirstlane(att_);
    while (__ballot(att_ > m_att)) m_att++;
    int m_reach = 0;
    if (__ballot(S.reach > 0)) {
        m_reach = 1;
        while (__ballot(S.reach > m_reach)) m_reach++;   /* reach is 1 in all but exotic steps: one extra ballot */
    }
    if (m_reach > 0) {
        /* the wave's reach into the map entry of its tile.  A wave covers 64 consecutive particles: one tile when the launch is
         * aligned with the tiles (the row length a multiple of 64), two otherwise — the loop runs once or twice */
        int *const rm = A.rmap + (size_t)((A.mr_idx >> 4) & 15) * (size_t)A.ntile;
        int mine = (S.reach > 0) ? rtile : -1;
        unsigned long long todo = __ballot(mine >= 0);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int tile0 = __builtin_amdgcn_readlane(mine, lead);
            const bool in = (mine == tile0);
            int r0 = 1;
            while (__ballot(in && S.reach > r0)) r0++;
            /* fire and forget: no load first — every tile has its own address (a cold line: the round trip would sit at the
             * end of every wave's life: measured +5 % on the three-wave kernels), and at most two waves write an entry */
            if ((int)(threadIdx.x & 63) == lead) atomicMax(rm + tile0, r0);
            if (in) mine = -1;
            todo = __ballot(mine >= 0);
        }
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
        if (m_att) atomicAdd(&c->wslots, 64ull * (unsigned long long)m_att);
        /* one address for the whole grid: only waves that would raise it touch it */
        int *const mr = reach_counters(A), *const mr_out = mr + ((A.mr_idx >> 4) & 15);
        /* (a wave without a record has nothing to raise: it does not look — the load would sit at the very end of its life) */
        if (m_reach > 0 && m_reach > __hip_atomic_load(mr_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            atomicMax(mr_out, m_reach);
            atomicMax(mr + 5, m_reach);                                      /* the running maximum can only rise when this one does */
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {       /* nobody reads or writes these during this step */
            mr[(A.mr_idx >> 8) & 15] = 0;
            mr[8 + ((A.mr_idx >> 8) & 15)] = 0;
        }
        /* "this step had waves with nothing to do" (order_wanted below): a wave that advanced nobody and re-seeded nobody says so */
        if (!b_adv && !s_res && mr[8 + ((A.mr_idx >> 4) & 15)] == 0) mr[8 + ((A.mr_idx >> 4) & 15)] = 1;
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

/* every kernel of the step families is launched with 256 threads per workgroup (StepLaunch.block, __launch_bounds__(256)) */
#define PICLES_BLOCK 256
/* the 256-node block of this workgroup: the permutation the previous step filed, or the XCD-banded natural order.  In two halves:
 * the counts are asked for BEFORE the kernel's first barrier (a scalar load the compiler would not move above it) and looked at
 * behind it, so their round trip overlaps the table set-up */
__device__ __forceinline__ long long order_counts(const Arrays &A)
{
    return A.ord_on ? *(const long long *)order_buf(A, A.mr_idx & 15) : 0ll;
}
/* ... and this workgroup's entry of the permutation with them, before it is known whether the permutation is complete (its address
 * depends on nothing but the block index): one round trip instead of two in the chain counts -> entry -> block -> every other load */
__device__ __forceinline__ int order_entry(const Arrays &A)
{
    return (A.ord_on && (int)blockIdx.x < A.nblk) ? order_buf(A, A.mr_idx & 15)[2 + blockIdx.x] : 0;
}
__device__ __forceinline__ unsigned int ordered_block(const Arrays &A, long long counts, int entry)
{
    if (A.ord_on) {
        const int busy = (int)(counts & 0xffffffffll), calm = (int)(counts >> 32);
        if (busy + calm == A.nblk && (int)gridDim.x == A.nblk && 8 * calm >= A.nblk) return (unsigned int)entry;
    }
    return xcd_block();
}
/* file this block for the next step.  No barrier: every wave signs off in an LDS word (order_sign_init() zeroes it ahead of the
 * kernel's first barrier) and the last one to do so — it sees the other three, and whether any of them was busy — files the block */
__device__ __forceinline__ int *order_sign(void)
{
    __shared__ int sign_;
    return &sign_;
}
__device__ __forceinline__ void order_sign_init(void) { if (threadIdx.x == 0) *order_sign() = 0; }
/* filing costs every workgroup an atomic with a returned value at the very end of its life (+1-2 % on the homogeneous box,
 * measured): it is done only while the runs are mixed — the previous step had calm waves (a word beside the reach counter) */
__device__ __forceinline__ bool order_wanted(const Arrays &A) { return reach_counters(A)[8 + (A.mr_idx & 15)] != 0; }
__device__ __forceinline__ void order_file(const Arrays &A, unsigned int lblock, bool wave_busy)
{
    if ((threadIdx.x & 63) != 0) return;
    const int v = atomicAdd(order_sign(), 1 + (wave_busy ? 256 : 0));
    if ((v & 255) != (int)(PICLES_BLOCK / 64) - 1) return;
    const bool busy = wave_busy || (v >> 8) != 0;
    int *const wb = order_buf(A, (A.mr_idx >> 4) & 15);
    const int pos = busy ? atomicAdd(wb, 1) : A.nblk - 1 - atomicAdd(wb + 1, 1);
    if (pos >= 0 && pos < A.nblk) wb[2 + pos] = (int)lblock;
    if (blockIdx.x == 0) { int *const cb = order_buf(A, (A.mr_idx >> 8) & 15); cb[0] = 0; cb[1] = 0; }
}

/* local rows [r0, r0+n0) ∪ [r1, r1+n1) -> particle index; lblock: the 256-node block the workgroup works on */
__device__ __forceinline__ bool rows_index(const GridP &G, int r0, int n0, int r1, int n1, long long &t, unsigned int lblock)
{
    /* (PICLES_BLOCK, not blockDim.x: the run-time value is a load from the dispatch packet, a memory round trip at the head of
     * every wave's chain of dependent prologue loads — the phase clock put that chain at a third of a wave's life) */
    /* One contiguous row range of whole 64-column strips, four rows at a time: the block is 64 columns x 4 rows, one wave per row,
     * instead of 256 consecutive nodes of one row.  A workgroup hands its wave slots on when the LAST of its four waves is done; four
     * vertically adjacent 64-node pieces meet about the same wind and sea where a 256-node strip along a row does not (the
     * wind ramp of BASELINE config 5: the waves of a strip took 30 to 57 µs and the launch ran 2.4 of 3 slots per SIMD filled).
     * Same number of blocks, same particles, each exactly once; the block number stays a label (XCD bands, cost-ordered dispatch). */
    if (n1 == 0 && (G.Nx & 63) == 0 && (n0 & 3) == 0) {
        const unsigned int nbx = (unsigned int)G.Nx >> 6;
        const unsigned int by = lblock / nbx, bx = lblock - by * nbx;
        if (by >= ((unsigned int)n0 >> 2)) return false;
        t = ((long long)r0 + 4ll * by + (threadIdx.x >> 6)) * G.Nx + (64ll * bx + (threadIdx.x & 63));
        return true;
    }
    long long tid = (long long)lblock * PICLES_BLOCK + threadIdx.x;
    long long na = (long long)n0 * G.Nx, nb = (long long)n1 * G.Nx;
    if (tid >= na + nb) return false;
    t = (tid < na) ? (long long)r0 * G.Nx + tid : (long long)r1 * G.Nx + (tid - na);
    return true;
}

/* NodeToParticle! (mapping_2D.jl:279-356) on the node value (e,mx,my), all in registers.
 * Returns the branch: 0 = A (node -> particle), 1 = B/C (re-seed from the wind), 2 = D (off). */
__device__ __forceinline__ int remesh_regs(const KParams &P, const Wind &w, unsigned char pf, double e, double mx,
                                           double my, double clock, double DT, Vec5 &z)
{
    double u, v;
    wind_at<true>(P, w, clock, u, v);    /* winds at model.clock.time, before tick!  (the stand-alone remesh: any window) */
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
    Wind w = load_wind<true>(P, A, t);
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

/* sum of the contributions to node (i, j); RT > 0: reach known at compile time (RT = 1, 2: the two-phase window above; RT = 3, 4: a row
 * at a time, unrolled along the row), RT == 0: runtime reach R.
 * Record elements are addressed with 32-bit offsets from one base pointer (the host falls back to
 * a single-plane-per-call layout check: (ny_loc + 2R) * 6 * Nx < 2^31 elements). */
/* The (2R+1)² window of an interior node in two phases.  Phase 1: the codes of ALL candidates are loaded before any is inspected
 * (one memory round trip); each lane notes which candidates land on its node, and on which corner, in three bit masks.  Phase 2: the
 * lane walks its matches in candidate order, four at a time, and issues the value loads of all four back to back before the first
 * is consumed — one more round trip, where a branch per matching candidate took one each (a node has four matches under a
 * uniform flow: the pull was a chain of 1 + 4 dependent round trips at reach 1, 5 + 4 at reach 2, and at reach 2 the waves spent
 * so long in it that the issue port ran 89 % busy instead of 97 %).  Sums are formed in candidate order: the same bits.  The
 * addresses come from an opaque copy of the base (hoisted out of the loop over the particle lists they spill by the hundred). */
template <int R, int DJ0 = -R, int NJ = 2 * R + 1>      /* rows DJ0 .. DJ0 + NJ - 1 of the window: a band of at most 32 candidates */
__device__ __forceinline__ void pull_window_2p(const double *__restrict__ rec, unsigned int base, unsigned int pl, unsigned int rowlen,
                                               int grp, double &s0, double &s1, double &s2)
{
    constexpr int W = 2 * R + 1, NC = W * NJ;
    static_assert(NC <= 32, "one bit per candidate");
    __asm__ volatile("" : "+v"(base));
    unsigned int m = 0u, ax = 0u, ay = 0u;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int di = c % W - R, dj = c / W + DJ0;
        const int d = (int)rec[base + (unsigned int)(dj * (int)rowlen + di) + 5u * pl] - (grp + 4 * (REC_BIAS - 1 - di) + 4 * 4096 * (REC_BIAS - 1 - dj));
        m |= ((d & ~(4 | 16384)) == 0 ? 1u : 0u) << c;
        ax |= ((d & 4) ? 0u : 1u) << c;
        ay |= ((d & 16384) ? 0u : 1u) << c;
    }
    /* under a locally uniform flow every lane of the wave has the same set of matching candidates: the walk is then done once, on
     * scalars (which candidate, its offset), and only the corner bits and the values stay per lane */
    const unsigned int m0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)m);
    if (!__ballot(m != m0)) {
        unsigned int mu = m0;
        while (mu != 0u) {
            double e[4], mx[4], my[4], wx[4], wy[4];
            bool on[4], hx[4], hy[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                on[k] = (mu != 0u);
                const int c = on[k] ? __builtin_ctz(mu) : 0;
                mu &= mu - 1u;
                hx[k] = (ax >> c) & 1u; hy[k] = (ay >> c) & 1u;
                const unsigned int off = base + (unsigned int)((c / W + DJ0) * (int)rowlen + (c % W - R));
                e[k] = mx[k] = my[k] = wx[k] = wy[k] = 0.0;
                if (on[k]) {
                    wx[k] = rec[off + 3u * pl]; wy[k] = rec[off + 4u * pl];
                    e[k] = rec[off]; mx[k] = rec[off + pl]; my[k] = rec[off + 2u * pl];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (on[k]) {
                    const double w = (hx[k] ? wx[k] : 1.0 - wx[k]) * (hy[k] ? wy[k] : 1.0 - wy[k]);
                    s0 += w * e[k];
                    s1 += w * mx[k];
                    s2 += w * my[k];
                }
            }
        }
        return;
    }
    while (__ballot(m != 0u)) {
        double e[4], mx[4], my[4], wx[4], wy[4];
        bool on[4], hx[4], hy[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            on[k] = (m != 0u);
            const int c = on[k] ? __builtin_ctz(m) : 0;
            m &= m - 1u;
            hx[k] = (ax >> c) & 1u; hy[k] = (ay >> c) & 1u;
            const unsigned int off = base + (unsigned int)((c / W + DJ0) * (int)rowlen + (c % W - R));
            e[k] = mx[k] = my[k] = wx[k] = wy[k] = 0.0;
            if (on[k]) {
                wx[k] = rec[off + 3u * pl]; wy[k] = rec[off + 4u * pl];
                e[k] = rec[off]; mx[k] = rec[off + pl]; my[k] = rec[off + 2u * pl];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (on[k]) {
                const double w = (hx[k] ? wx[k] : 1.0 - wx[k]) * (hy[k] ? wy[k] : 1.0 - wy[k]);
                s0 += w * e[k];
                s1 += w * mx[k];
                s2 += w * my[k];
            }
        }
    }
}
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
            if constexpr (RT == 1 || RT == 2) {
                /* reach 1 and 2 — the first steps of a run and the developed sea.  (Bands of rows for reach 3 and 4 were built too:
                 * the compiler answered with 124 instead of 30 lane moves inside the RK loop of the time-varying default-solver
                 * kernel, which is the one that meets such reaches; they keep the row loop below.) */
                pull_window_2p<RT>(rec, base, pl, rowlen, grp, s0, s1, s2);
            } else {
                /* wider reach: one row of candidates at a time */
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
    if (R == 0) return;       /* the reach map says that nobody around left a record (pull_reach_local): the node value stays 0 */
    if (G.tripolar && jl + G.j_begin >= G.Ny - R) pull_node_tripolar(G, A, i, jl, R, s0, s1, s2);
    else if ((G.periodic_x && W > G.Nx) || (G.periodic_y && W > G.Ny)) pull_node_aliased(G, A, i, jl, R, s0, s1, s2);
    else if (R == 1) pull_node<1>(G, A, i, jl, 1, s0, s1, s2);
    else if (R == 2) pull_node<2>(G, A, i, jl, 2, s0, s1, s2);
    else if (R == 3) pull_node<3>(G, A, i, jl, 3, s0, s1, s2);     /* a fully developed sea under strong winds */
    else if (R == 4) pull_node<4>(G, A, i, jl, 4, s0, s1, s2);     /* 20-minute steps (BASELINE config 5) */
    else pull_node<0>(G, A, i, jl, R, s0, s1, s2);
}

/* reach the pull of local row jl must cover.  A whole-grid context follows the reach its own advance measured.  A slab
 * covers halo_rows for the rows that can receive from a neighbour's particles (the edge rows: the neighbour's reach is not
 * known here), and for its interior rows — fed by own particles only — again the measured reach, which is what keeps the
 * common case at 9 candidates per node instead of (2 halo_rows + 1)².  Any reach >= the true one gives the same bits. */
/* the counter itself is asked for at the very start of the kernel (pull_reach_early: its address depends on the arguments alone,
 * and the compiler does not move a load above the first barrier by itself): one link less in the chain of dependent round trips
 * reach -> reach map -> candidates that a wave with nothing else to do spends its life in */
__device__ __forceinline__ int pull_reach_early(const Arrays &A) { return reach_counters(A)[A.mr_idx & 15]; }
__device__ __forceinline__ int pull_reach(const GridP &G, const Arrays &A, int jl, int m)
{
    if (G.Rp > 0 && (jl < G.R || jl >= G.ny_loc - G.R)) return G.Rp;
    if (m < 1) m = 1;
    return (G.Rp > 0 && m > G.Rp) ? G.Rp : m;
}
__device__ __forceinline__ int pull_reach(const GridP &G, const Arrays &A, int jl) { return pull_reach(G, A, jl, pull_reach_early(A)); }


/* every tile's first particle clears the tile's entry in the map two steps ahead (nobody reads or writes that buffer during this
 * step: the rotation of Arrays::mr_idx); every node is covered by exactly one launch per step, so every entry is cleared once */
__device__ __forceinline__ void rmap_clear_ahead(const Arrays &A, long long t)
{
    if ((t & 63) == 0) A.rmap[(size_t)((A.mr_idx >> 8) & 15) * (size_t)A.ntile + (size_t)(t >> 6)] = 0;
}

/* The reach the pull of node (i, jl) must cover: the largest reach of any particle that can land on it.  Rg (pull_reach: the
 * grid-wide maximum of the last advance, or halo_rows for the edge rows of a slab) bounds where such particles sit; among the
 * tiles that intersect that window the map knows how far particles really went.  Any reach >= the true one gives the same
 * bits (candidates that do not land on the node never match; the visiting order of those that do is unchanged), so the result is
 * made wave-uniform — the maximum over the ACTIVE lanes of the wave, by a ballot ladder (callers sit inside `if (active)`: a
 * shuffle would read the registers of inactive lanes) — and the reach dispatch of pull_any stays a scalar branch.  Nodes whose
 * window touches a wrap, an open edge or ghost rows keep Rg. */
__device__ __forceinline__ int pull_reach_local(const GridP &G, const Arrays &A, int i, int jl, int Rg)
{
    int R = Rg;
    const int j = jl + G.j_begin;
    const bool edge_row = G.Rp > 0 && (jl < G.R || jl >= G.ny_loc - G.R);          /* fed by a neighbour's particles: no local knowledge */
    if (Rg >= 2 && Rg <= 31 && !edge_row && i - Rg >= 0 && i + Rg < G.Nx && jl - Rg >= 0 && jl + Rg < G.ny_loc &&
        j - Rg >= 0 && j + Rg < G.Ny) {
        const int *const rm = A.rmap + (size_t)(A.mr_idx & 15) * (size_t)A.ntile;
        int m = 0;                        /* 0: no particle of any tile in the window left a record — nothing can land here */
        long long t0 = (long long)(jl - Rg) * G.Nx + (i - Rg);
        for (int dj = -Rg; dj <= Rg; dj++, t0 += G.Nx) {                           /* 2 Rg + 1 <= 63 columns: at most two tiles per row */
            const int a = rm[t0 >> 6], b = rm[(t0 + 2 * Rg) >> 6];
            m = max(m, max(a, b));
        }
        R = min(m, Rg);
    }
    if (Rg < 2) return Rg;
    int m = 0;
    while (__ballot(R > m)) m++;          /* <= Rg rounds */
    return m;
}

/* launchers of the kernel families that live in their own translation units (k_step_*.hip, k_advance.hip) */
struct StepLaunch {
    dim3 grid, block;
    hipStream_t stream;
    const KParams *P;
    const GridP *G;
    const Arrays *A;
    double t_prev, DT_prev, t_start, DT;
    int r0, n0, r1, n1;
};
/* fused step: fast = specialised physics; solver 0 DP5, 1 Tsit5, 2 auto-switching; wind_static; metric */
void launch_k_step_explicit(const StepLaunch &L, bool fast, int solver, bool wind_static, bool metric);
void launch_k_step_auto(const StepLaunch &L, bool wind_static, bool metric);
void launch_k_advance(const StepLaunch &L, bool fast, int solver, bool wind_static, bool metric);

#endif /* PICLES_KERNELS_H */
