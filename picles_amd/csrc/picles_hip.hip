/*
 * picles_hip.hip — the C ABI of include/picles_hip.h (host side) and the HIP kernels (gfx950 / CDNA4) that are not
 * flavoured: k_seed, k_scatter, k_remesh, k_push_tiles, the cell list, k_wind_sample.  The flavoured kernel families live in
 * their own translation units so that they compile in parallel — k_step_explicit.hip, k_step_auto.hip (template in
 * k_step.inc), k_advance.hip — and kernels.h holds the device code all of them share.
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
#include <mutex>
#include <string>
#include <vector>

#include "../../include/picles_hip.h"
#include "physics.h"

#define PX_EXPORT extern "C" __attribute__((visibility("default")))

#include "kernels.h"

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
        Wind w = load_wind<true>(P, A, t);
        double u, v;
        wind_at<true>(P, w, 0.0, u, v);   /* winds at t = 0.0 (run.jl:213-215) */
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

template <bool REMESH>
__global__ void __launch_bounds__(256) k_scatter(KParams P, GridP G, Arrays A, int accum, int movie,
                                                   double clock, double DT)
{
    if (REMESH) pm_device_init();
    long long t = (long long)xcd_block() * PICLES_BLOCK + threadIdx.x;
    unsigned int reseeds = 0;
    const bool active = t < A.n;
    if (active) {
        int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (accum) { s0 = A.state[t]; s1 = A.state[t + A.n]; s2 = A.state[t + 2 * A.n]; }
        pull_any(G, A, i, jl, pull_reach_local(G, A, i, jl, pull_reach(G, A, jl)), s0, s1, s2);
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

/* NT time levels per launch (the spatial cell and its weights are shared): level q at time tq[q] goes to (uo[q], vo[q]) */
struct WindSampleOut { double t[2]; double *u[2], *v[2]; };
template <int NT>
__global__ void __launch_bounds__(256) k_wind_sample(GridP G, WindGrid Wg, WindSampleOut O, long long n)
{
    long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int i = (int)(k % G.Nx), j = (int)(k / G.Nx) + G.j_begin;
    double x = Wg.mesh_x0 + (double)i * Wg.mesh_dx, y = Wg.mesh_y0 + (double)j * Wg.mesh_dy;
    int ix, iy;
    double fx, fy;
    lattice_coord((x - Wg.x0) * Wg.inv_dx, Wg.nx, ix, fx);
    lattice_coord((y - Wg.y0) * Wg.inv_dy, Wg.ny, iy, fy);
    size_t sx = 1, sy = (size_t)Wg.nx, st = (size_t)Wg.nx * Wg.ny;
    const double *F[2] = {Wg.u, Wg.v};
#pragma unroll
    for (int q = 0; q < NT; q++) {
        int it;
        double ft;
        lattice_coord((O.t[q] - Wg.t0) * Wg.inv_dt, Wg.nt, it, ft);
        size_t b = ix * sx + iy * sy + it * st;
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
        O.u[q][k] = out[0];
        O.v[q][k] = out[1];
    }
}

/* ------------------------------------------------------------------------------------------
 * k_wind_poly — coefficients of a polyline window (KParams::wind_nk >= 2; physics.h, wind_eval): from the node's levels
 * L_0 (u0 plane), L_1 .. L_nk (lv planes, one pair per knot), L_nk+1 (u1 plane) the segment slopes per unit s,
 * sl_j = (L_j+1 - L_j) ilen_j, and their jumps at the knots:  du = sl_0,  b_k = sl_k - sl_k-1.  xb: [0, PICLES_MAX_KNOTS) the
 * knots' s, then planes du, dv, b_1u, b_1v, b_2u, ...
 * ---------------------------------------------------------------------------------------- */
struct WindPolyForm { int nk; double sk[PICLES_MAX_KNOTS]; double ilen[PICLES_MAX_KNOTS + 1]; };
__global__ void __launch_bounds__(256) k_wind_poly(WindPolyForm F, const double *u0, const double *v0, const double *u1, const double *v1,
                                                   const double *lv, double *xb, long long n)
{
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < PICLES_MAX_KNOTS) xb[t] = (t < F.nk) ? F.sk[t] : 2.0;    /* (beyond the window: never reached) */
    if (t >= n) return;
    double *pl = xb + PICLES_MAX_KNOTS;
    for (int c = 0; c < 2; c++) {
        double prev = c ? v0[t] : u0[t], slp = 0.0;
        for (int j = 0; j <= F.nk; j++) {
            const double next = (j < F.nk) ? lv[(size_t)(2 * j + c) * n + t] : (c ? v1[t] : u1[t]);
            const double sl = (next - prev) * F.ilen[j];
            pl[(size_t)(2 * j + c) * n + t] = j ? sl - slp : sl;
            slp = sl;
            prev = next;
        }
    }
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
    /* reach counters, rotating with the steps: the previous step's is read by the pull, the current one is written, the one two
     * steps ahead is cleared (five, not three: row blocks of a pipelined run may be one step apart, see run_pipelined) */
    int *mr_buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int mr_w = 0;                                      /* index of the counter the step in flight writes */
    struct Pipe *pipe = nullptr;                       /* space-time pipelined run (picles_set_pipeline) */
    int pipe_nb = 0;                                   /* requested row blocks per step (0 / 1: off) */
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
    int timing_mode = 0;          /* 1: events around every launch; 2: one pair around each picles_run_steps call (its launches are
                                     back to back on one stream: region / launches = the mean launch, with no event between them) */
    bool in_region = false;       /* inside a mode-2 region: the per-launch events are skipped */
    int region_launches = 0;
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
    int wind_grid_mode = PICLES_LATTICE_LINEAR;
    double wg_t0 = 0.0, wg_dt = 0.0;   /* the lattice's first time knot and spacing as the caller gave them (knot times are computed from these) */
    bool ord_valid = false;        /* the previous fused step filed a dispatch order (kernels.h: Arrays::ord) with ... */
    int ord_nblk = 0;              /* ... this many workgroups: the whole grid, or the interior rows of a slab */
    double *um_buf = nullptr, *vm_buf = nullptr;   /* mid-window wind level (picles_set_winds3); A.um / A.vm point here while in use */
    double *lv_buf = nullptr, *xb_buf = nullptr;   /* polyline windows: the levels at the knots (2 planes per knot) and the coefficient planes (k_wind_poly) */
    int poly_cap = 0;                              /* knots the two buffers hold */
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
    if (c->in_region) { if (kind == 0) c->region_launches++; return; }
    picles_ctx::Ev e;
    if (!c->ev_free.empty()) { e = c->ev_free.back(); c->ev_free.pop_back(); }
    else { hipEventCreate(&e.a); hipEventCreate(&e.b); }
    e.kind = kind;
    hipEventRecord(e.a, s);
    c->ev_used.push_back(e);
}
static void timing_end(picles_ctx *c, hipStream_t s)
{
    if (!c->timing || c->in_region) return;
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
        case 3: c->tim.advance_ms += ms; break;      /* a mode-2 region: its launches were counted when they were enqueued */
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
    A.mr_idx = (read_buf == write_buf ? c->mr_w : (c->mr_w + 4) % 5) | (c->mr_w << 4) | (((c->mr_w + 2) % 5) << 8);
    return A;
}

static int launch_scatter(picles_ctx *c, hipStream_t s, bool remesh);

/* scatter + remesh of the last fused step, if still outstanding */
static int flush(picles_ctx *c)
{
    if (!c->pending) return 0;
    HIPCHK(c, hipDeviceSynchronize());   /* the fused launches may have run on caller-provided streams */
    c->pending = false;                  /* only now: a failed synchronisation leaves the step pending */
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
    {
        const double k1 = 0.25 * PK_G0, k2 = k1 * k1, K = k2 * k2;      /* k_p⁴ = K (1/c_gp)⁸ */
        P.KeT4 = K * P.inv_eT4;
        P.KrCa = (K * P.r_g) * P.C_alpha;
        P.Cdir = P.C_phi * P.two_inv_rg2;
        /* the y = 1/|c̄| form of the RHS (physics.h rhs3): the powers of r_g in the constants */
        P.rg2 = P.r_g * P.r_g;
        const double rg4 = P.rg2 * P.rg2, rg8 = rg4 * rg4;
        P.Cw = (0.5 * PK_G0) * P.r_g;
        P.Chrh = -0.25 * P.r_g;
        P.ymax = 10.0 / P.r_g;
        P.sgmax = 1e8 / P.rg2;
        P.KeT4y = P.KeT4 * rg8;
        P.KrCay = P.KrCa * rg8;
        P.Cs = (0.5 * P.C_phi) * P.rg2;
        P.Cdir2 = 2.0 * P.C_phi;
        P.g4rg2 = k1 * P.rg2;
        P.qU2r_max = 249999.0 / (P.ymax * P.ymax);
    }
    P.inv_dx = 1.0 / g->dx; P.inv_dy = 1.0 / g->dy;
    P.deadband2 = p->dir_deadband * p->dir_deadband;
    P.propagation = p->propagation; P.input = p->input; P.dissipation = p->dissipation;
    P.peak_shift = p->peak_shift; P.direction = p->direction; P.n_is_2 = (P.n == 2.0);
    P.p_is_075 = (P.p == 0.75);
    P.fast_phys = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.p_is_075 && P.deadband2 == 0.0;
    P.abstol = o->abstol; P.reltol = o->reltol; P.dt0 = o->dt0; P.dtmin = o->dtmin;
    P.inv_abstol = 1.0 / o->abstol;
    P.maxiters = o->maxiters; P.force_dtmin = o->force_dtmin;
    P.solver = o->solver;
    P.lne_max = o->log_energy_maximum; P.wind_min_sq = o->wind_min_squared;
    P.init_type = m->init_type;
    P.def_lne = m->default_particle[0]; P.def_cx = m->default_particle[1]; P.def_cy = m->default_particle[2];
    P.min_e = m->minimal_state[0]; P.min_m2 = m->minimal_state[1];
    P.wind_static = 1; P.tw0 = 0.0; P.inv_dtw = 0.0; P.wind_sk = P.wind_isk = P.wind_i1sk = 0.0;

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
    CK(hipMalloc(&A.cnt, NSLOTS * sizeof(DevCounters) + 16 * sizeof(int)));     /* + the reach counters (kernels.h: reach_counters) */
    for (int k = 0; k < 5; k++) c->mr_buf[k] = (int *)(A.cnt + NSLOTS) + k;     /* five rotating reach counters; [5] = the running maximum */
    CK(hipMemset(c->mr_buf[0], 0, 16 * sizeof(int)));
    for (int k = 0; k < 2; k++) {
        CK(hipMalloc(&c->rec_buf[k], rec_bytes(c)));
        CK(hipMemset(c->rec_buf[k], 0, rec_bytes(c)));
    }
    A.nblk = (int)((n + 255) / 256);
    A.ord_on = 0;
    A.ord = nullptr;
    CK(hipMalloc(&A.ord, (size_t)5 * (2 + A.nblk) * sizeof(int)));      /* (a slab orders the launch over its interior rows: fewer workgroups) */
    CK(hipMemset(A.ord, 0, (size_t)5 * (2 + A.nblk) * sizeof(int)));
    A.ntile = (int)((n + 63) / 64);
    CK(hipMalloc(&A.rmap, (size_t)5 * A.ntile * sizeof(int)));      /* local reach map, five rotating buffers (kernels.h: Arrays::rmap) */
    CK(hipMemset(A.rmap, 0, (size_t)5 * A.ntile * sizeof(int)));
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
    if (c->um_buf) { hipFree(c->um_buf); hipFree(c->vm_buf); }
    if (c->lv_buf) { hipFree(c->lv_buf); hipFree(c->xb_buf); }
    hipFree(A.cnt); hipFree(A.rmap); hipFree(c->d_mask);
    if (A.ord) hipFree(A.ord);
    if (A.m11) { hipFree(A.m11); hipFree(A.m22); hipFree(A.pc); }
    for (int k = 0; k < 2; k++) hipFree(c->rec_buf[k]);

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

static inline unsigned nblocks(long long n, int b) { return (unsigned)((n + b - 1) / b); }

/* form of a three-level window: the parabola through (t0, (t0+t1)/2, t1), or two straight segments meeting at the knot tk */
static void wind_window_form(picles_ctx *c, double t0, double t1, bool knot, double tk)
{
    KParams &P = c->P;
    P.tw0 = t0;
    P.inv_dtw = 1.0 / (t1 - t0);
    if (knot) {
        P.wind_sk = (tk - t0) * P.inv_dtw;
        P.wind_isk = 1.0 / P.wind_sk;
        P.wind_i1sk = 1.0 / (1.0 - P.wind_sk);
    } else {
        P.wind_sk = P.wind_isk = P.wind_i1sk = 0.0;
    }
    P.wind_nk = knot ? 1 : 0;
}

/* a polyline window over [t0, t1] with knots tk[0 .. nk) (nk >= 2): the levels at the knots are in lv_buf, levels 0 and nk+1 in the
 * (u0, v0) / (u1, v1) planes; lays down the coefficient planes and sets the window's form.  (um, vm) = the first knot's level and the
 * one-knot scalars describe the first knot: whatever evaluates the window at its START without knowing about polylines — the fused
 * step's remesh reads level 0 from its plane — still gets level 0. */
static int wind_poly_buffers(picles_ctx *c, int nk)
{
    if (nk <= c->poly_cap) return 0;
    if (c->lv_buf) { hipFree(c->lv_buf); hipFree(c->xb_buf); c->lv_buf = c->xb_buf = nullptr; c->poly_cap = 0; }
    const size_t n = (size_t)c->A.n;
    HIPCHK(c, hipMalloc(&c->lv_buf, (size_t)2 * nk * n * 8));
    HIPCHK(c, hipMalloc(&c->xb_buf, ((size_t)PICLES_MAX_KNOTS + (size_t)2 * (nk + 1) * n) * 8));
    c->poly_cap = nk;
    return 0;
}
static int wind_window_poly(picles_ctx *c, double t0, double t1, int nk, const double *tk, hipStream_t s)
{
    KParams &P = c->P;
    Arrays &A = c->A;
    wind_window_form(c, t0, t1, true, tk[0]);
    WindPolyForm F;
    F.nk = nk;
    double sprev = 0.0;
    for (int k = 0; k < PICLES_MAX_KNOTS; k++) F.sk[k] = 2.0;
    for (int k = 0; k <= nk; k++) {
        const double sn = (k < nk) ? (tk[k] - t0) * P.inv_dtw : 1.0;
        if (k < nk) F.sk[k] = sn;
        F.ilen[k] = 1.0 / (sn - sprev);
        sprev = sn;
    }
    for (int k = nk + 1; k <= PICLES_MAX_KNOTS; k++) F.ilen[k] = 0.0;
    hipLaunchKernelGGL(k_wind_poly, dim3(nblocks(A.n, 256)), dim3(256), 0, s, F, A.u0, A.v0, A.u1, A.v1, c->lv_buf, c->xb_buf, A.n);
    HIPCHK(c, hipGetLastError());
    A.um = c->lv_buf; A.vm = c->lv_buf + A.n;
    P.wind_nk = nk;
    P.wind_xb = c->xb_buf;
    P.wind_xn = A.n;
    return 0;
}

static int set_wind_levels(picles_ctx *c, const double *u0, const double *v0, double t0,
                           const double *um, const double *vm, bool knot, double tk,
                           const double *u1, const double *v1, double t1)
{
    if (!c || !u0 || !v0) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = flush(c); if (rc) return rc; }
    HIPCHK(c, hipDeviceSynchronize());   /* the previous step may still read the wind planes */
    c->wind_grid_on = false;
    Arrays &A = c->A;
    size_t b = (size_t)A.n * 8;
    HIPCHK(c, hipMemcpyAsync(A.u0, u0, b, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(A.v0, v0, b, hipMemcpyHostToDevice, c->stream));
    const bool two = u1 && v1 && t1 != t0;
    const bool three = two && um && vm;
    if (two) {
        HIPCHK(c, hipMemcpyAsync(A.u1, u1, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(A.v1, v1, b, hipMemcpyHostToDevice, c->stream));
        c->P.wind_static = 0;
        wind_window_form(c, t0, t1, three && knot, tk);
    } else {
        c->P.wind_static = 1;
        c->P.tw0 = t0;
        c->P.inv_dtw = 0.0;
        c->P.wind_sk = c->P.wind_isk = c->P.wind_i1sk = 0.0;
        c->P.wind_nk = 0;
    }
    if (three) {                         /* third level: the planes exist from the first three-level call on */
        if (!c->um_buf) {
            HIPCHK(c, hipMalloc(&c->um_buf, b));
            HIPCHK(c, hipMalloc(&c->vm_buf, b));
        }
        HIPCHK(c, hipMemcpyAsync(c->um_buf, um, b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->vm_buf, vm, b, hipMemcpyHostToDevice, c->stream));
        A.um = c->um_buf; A.vm = c->vm_buf;
    } else {
        A.um = A.vm = nullptr;           /* two levels: linear in t (bit for bit the two-level arithmetic) */
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));   /* caller may reuse its host buffers */
    return 0;
}

PX_EXPORT int32_t picles_set_winds3(picles_ctx *c, const double *u0, const double *v0, double t0,
                                    const double *um, const double *vm,
                                    const double *u1, const double *v1, double t1)
{
    return set_wind_levels(c, u0, v0, t0, um, vm, false, 0.0, u1, v1, t1);
}

PX_EXPORT int32_t picles_set_winds(picles_ctx *c, const double *u0, const double *v0, double t0,
                                   const double *u1, const double *v1, double t1)
{
    return set_wind_levels(c, u0, v0, t0, nullptr, nullptr, false, 0.0, u1, v1, t1);
}

PX_EXPORT int32_t picles_set_winds_knot(picles_ctx *c, const double *u0, const double *v0, double t0,
                                        const double *uk, const double *vk, double tk,
                                        const double *u1, const double *v1, double t1)
{
    if (!c) return -1;
    if (!uk || !vk || !u1 || !v1) return fail(c, -2, "picles_set_winds_knot needs all three levels");
    if (!(t0 < tk && tk < t1)) return fail(c, -2, "picles_set_winds_knot: the knot must lie strictly inside the window, t0 < tk < t1");
    return set_wind_levels(c, u0, v0, t0, uk, vk, true, tk, u1, v1, t1);
}

/* nlev >= 2 node-sampled levels at strictly increasing times: the piecewise-linear wind through them (a gridded wind sampled by the
 * host at every time knot inside the step and at its ends).  2 levels = picles_set_winds, 3 = picles_set_winds_knot; more: the
 * polyline window (at most PICLES_MAX_KNOTS levels between the ends). */
PX_EXPORT int32_t picles_set_winds_polyline(picles_ctx *c, int32_t nlev, const double *const *u, const double *const *v, const double *times)
{
    if (!c) return -1;
    if (nlev < 2 || !u || !v || !times) return fail(c, -2, "picles_set_winds_polyline needs at least two levels");
    for (int k = 0; k < nlev; k++) if (!u[k] || !v[k]) return fail(c, -2, "picles_set_winds_polyline: a level is missing");
    for (int k = 1; k < nlev; k++) if (!(times[k - 1] < times[k])) return fail(c, -2, "picles_set_winds_polyline: the level times must increase strictly");
    if (nlev == 2) return set_wind_levels(c, u[0], v[0], times[0], nullptr, nullptr, false, 0.0, u[1], v[1], times[1]);
    if (nlev == 3) return set_wind_levels(c, u[0], v[0], times[0], u[1], v[1], true, times[1], u[2], v[2], times[2]);
    const int nk = nlev - 2;
    if (nk > PICLES_MAX_KNOTS) {
        char buf[160];
        snprintf(buf, sizeof buf, "picles_set_winds_polyline: %d levels inside the window, at most %d are carried", nk, PICLES_MAX_KNOTS);
        return fail(c, -2, buf);
    }
    /* levels 0, 1 and the last through the three-level path (the one-knot form of the first knot), then the further knots */
    int rc = set_wind_levels(c, u[0], v[0], times[0], u[1], v[1], true, times[1], u[nlev - 1], v[nlev - 1], times[nlev - 1]);
    if (rc) return rc;
    if ((rc = wind_poly_buffers(c, nk))) return rc;
    const size_t b = (size_t)c->A.n * 8;
    for (int k = 0; k < nk; k++) {
        HIPCHK(c, hipMemcpyAsync(c->lv_buf + (size_t)(2 * k) * c->A.n, u[k + 1], b, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->lv_buf + (size_t)(2 * k + 1) * c->A.n, v[k + 1], b, hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = wind_window_poly(c, times[0], times[nlev - 1], nk, times + 1, c->stream))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));   /* caller may reuse its host buffers */
    return 0;
}


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
    c->wg_t0 = t0; c->wg_dt = dt;
    c->A.um = c->A.vm = nullptr;
    c->P.wind_sk = c->P.wind_isk = c->P.wind_i1sk = 0.0;
    c->P.wind_nk = 0;
    c->wind_grid_on = true;
    c->wind_grid_mode = PICLES_LATTICE_LINEAR;
    c->wind_t1_valid = false;
    return 0;
}

PX_EXPORT int32_t picles_set_wind_grid_mode(picles_ctx *c, int32_t mode)
{
    if (!c) return -1;
    if (mode != PICLES_LATTICE_LINEAR && mode != PICLES_LATTICE_SMOOTH3) return fail(c, -2, "unknown wind lattice mode");
    if (!c->wind_grid_on) return fail(c, -2, "picles_set_wind_grid first");
    { int rc = flush(c); if (rc) return rc; }
    c->wind_grid_mode = mode;
    c->wind_t1_valid = false;
    return 0;
}

/* time knots of the lattice strictly inside (t, t + dt): 0, 1 or 2 (= two or more); *tk = the first one.  Knots sit at whole
 * multiples of lat_dt from lat_t0 (the periodic continuation's period is a whole number of intervals); one closer to an end of the
 * window than 1e-9 intervals is that end (600-second steps against 900-second knots must not see a "knot" 1e-13 s before t + dt). */
PX_EXPORT int32_t picles_lattice_knots(double lat_t0, double lat_dt, double t, double dt, double *tk)
{
    const double eps = 1e-9;
    const double c0 = (t - lat_t0) / lat_dt, c1 = (t + dt - lat_t0) / lat_dt;
    const double k0 = __builtin_floor(c0 + eps) + 1.0;
    if (!(k0 < c1 - eps)) return 0;
    if (tk) *tk = lat_t0 + k0 * lat_dt;
    return (k0 + 1.0 < c1 - eps) ? 2 : 1;
}

/* all of them: the number of time knots strictly inside (t, t + dt) (same rule), the first `cap` of their times in tks */
PX_EXPORT int32_t picles_lattice_knot_times(double lat_t0, double lat_dt, double t, double dt, double *tks, int32_t cap)
{
    const double eps = 1e-9;
    const double c0 = (t - lat_t0) / lat_dt, c1 = (t + dt - lat_t0) / lat_dt;
    int n = 0;
    for (double k = __builtin_floor(c0 + eps) + 1.0; k < c1 - eps; k += 1.0) {
        if (tks && n < cap) tks[n] = lat_t0 + k * lat_dt;
        if (++n == 0x7fffffff) break;
    }
    return n;
}

/* what the step window [t, t + dt] over the lattice looks like: levels, form, time of the middle level; nk >= 2: a polyline window
 * with knots at tk[0 .. nk) */
struct WindowPlan { bool three; bool knot; double tm; int nk; double tk[PICLES_MAX_KNOTS]; };
static int wind_window_plan(picles_ctx *c, double t, double dt, WindowPlan &W)
{
    W = WindowPlan{};
    if (c->wind_grid_mode == PICLES_LATTICE_SMOOTH3) {
        W.three = true;
        W.tm = t + 0.5 * dt;
        return 0;
    }
    const int nk = picles_lattice_knot_times(c->wg_t0, c->wg_dt, t, dt, W.tk, PICLES_MAX_KNOTS);
    if (nk > PICLES_MAX_KNOTS) {
        char buf[360];
        snprintf(buf, sizeof buf, "the model step [%.17g, %.17g] contains %d time knots of the wind lattice (spacing %.17g s); a window carries at most %d: "
                 "take shorter model steps, or picles_set_wind_grid_mode(PICLES_LATTICE_SMOOTH3) if the lattice tabulates a smooth closure",
                 t, t + dt, nk, c->wg_dt, PICLES_MAX_KNOTS);
        return fail(c, -7, buf);
    }
    if (nk >= 1) { W.three = true; W.knot = true; W.tm = W.tk[0]; }
    W.nk = nk;
    return 0;
}

/* sample the levels of the window [t, t + dt] that are not on the device yet: level 1 at t + dt into (u1, v1), the middle level
 * (the knot, or t + dt/2) into (um, vm) when the window has one, level 0 into (u0, v0) when `with0`; sets the window's form */
static int wind_window_sample(picles_ctx *c, const WindowPlan &W, double t, double dt, bool with0, hipStream_t s)
{
    Arrays &A = c->A;
    dim3 grid(nblocks(A.n, 256)), block(256);
    if (W.three && W.nk < 2 && !c->um_buf) {
        HIPCHK(c, hipMalloc(&c->um_buf, (size_t)A.n * 8));
        HIPCHK(c, hipMalloc(&c->vm_buf, (size_t)A.n * 8));
    }
    if (with0) {
        WindSampleOut O = {{t, 0.0}, {A.u0, nullptr}, {A.v0, nullptr}};
        hipLaunchKernelGGL(k_wind_sample<1>, grid, block, 0, s, c->G, c->wg, O, A.n);
    }
    if (W.nk >= 2) {       /* a polyline: the levels at the knots, the level at the window's end, then the coefficient planes */
        { int rc = wind_poly_buffers(c, W.nk); if (rc) return rc; }
        const size_t n = (size_t)A.n;
        for (int k = 0; k < W.nk; k += 2) {
            const bool two = k + 1 < W.nk;
            WindSampleOut O = {{W.tk[k], two ? W.tk[k + 1] : t + dt},
                               {c->lv_buf + (size_t)(2 * k) * n, two ? c->lv_buf + (size_t)(2 * k + 2) * n : A.u1},
                               {c->lv_buf + (size_t)(2 * k + 1) * n, two ? c->lv_buf + (size_t)(2 * k + 3) * n : A.v1}};
            hipLaunchKernelGGL(k_wind_sample<2>, grid, block, 0, s, c->G, c->wg, O, A.n);
        }
        if (W.nk % 2 == 0) {
            WindSampleOut O = {{t + dt, 0.0}, {A.u1, nullptr}, {A.v1, nullptr}};
            hipLaunchKernelGGL(k_wind_sample<1>, grid, block, 0, s, c->G, c->wg, O, A.n);
        }
        HIPCHK(c, hipGetLastError());
        c->wind_t1 = t + dt;
        c->wind_t1_valid = true;
        c->P.wind_static = 0;
        return wind_window_poly(c, t, t + dt, W.nk, W.tk, s);
    }
    if (W.three) {
        WindSampleOut O = {{W.tm, t + dt}, {c->um_buf, A.u1}, {c->vm_buf, A.v1}};
        hipLaunchKernelGGL(k_wind_sample<2>, grid, block, 0, s, c->G, c->wg, O, A.n);
        A.um = c->um_buf; A.vm = c->vm_buf;
    } else {
        WindSampleOut O = {{t + dt, 0.0}, {A.u1, nullptr}, {A.v1, nullptr}};
        hipLaunchKernelGGL(k_wind_sample<1>, grid, block, 0, s, c->G, c->wg, O, A.n);
        A.um = A.vm = nullptr;
    }
    HIPCHK(c, hipGetLastError());
    c->wind_t1 = t + dt;
    c->wind_t1_valid = true;
    c->P.wind_static = 0;
    wind_window_form(c, t, t + dt, W.knot, W.tm);
    return 0;
}

/* sample the lattice for the step [t, t+dt] into (u0,v0) / (u1,v1) (+ the middle level); reuses the level the previous
 * step left in (u1,v1) by swapping the plane pointers */
static int wind_grid_prepare(picles_ctx *c, double t, double dt, hipStream_t s)
{
    Arrays &A = c->A;
    WindowPlan W;
    { int rc = wind_window_plan(c, t, dt, W); if (rc) return rc; }
    const bool reuse = c->wind_t1_valid && c->wind_t1 == t;
    if (reuse) {
        std::swap(A.u0, A.u1);
        std::swap(A.v0, A.v1);
    }
    return wind_window_sample(c, W, t, dt, !reuse, s);
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

PX_EXPORT int32_t picles_get_winds_mid(picles_ctx *c, double *um, double *vm)
{
    if (!c) return -1;
    if (!c->A.um) return 1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    size_t b = (size_t)c->A.n * 8;
    if (um) HIPCHK(c, hipMemcpy(um, c->A.um, b, hipMemcpyDeviceToHost));
    if (vm) HIPCHK(c, hipMemcpy(vm, c->A.vm, b, hipMemcpyDeviceToHost));
    return 0;
}

PX_EXPORT int32_t picles_seed(picles_ctx *c, double t0)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ext_streams) HIPCHK(c, hipDeviceSynchronize());   /* earlier steps may still run on caller / ring streams */
    c->clock = t0;
    if (c->wind_grid_on) {   /* winds at t = 0.0 seed the particles (run.jl:213-215) */
        /* only level 0 is read (k_seed evaluates the window at its start); the window's end is sampled at the seed time scale, whatever
         * lies between: a plain two-level window, replaced by the first step's own */
        c->wind_t1_valid = false;
        const WindowPlan two = WindowPlan{};
        int rc = wind_window_sample(c, two, 0.0, c->od.timestep, true, c->stream);
        if (rc) return rc;
    }
    c->pending = false;
    c->ord_valid = false;
    c->cur = 0;
    c->mr_w = 0;
    for (int k = 0; k < 2; k++) {
        HIPCHK(c, hipMemsetAsync(c->rec_buf[k], 0, rec_bytes(c), c->stream));
        HIPCHK(c, hipMemsetAsync(c->mr_buf[k], 0, sizeof(int), c->stream));
        if (k == 0) for (int q = 2; q < 5; q++) HIPCHK(c, hipMemsetAsync(c->mr_buf[q], 0, sizeof(int), c->stream));
    }
    HIPCHK(c, hipMemsetAsync(c->A.cnt, 0, NSLOTS * sizeof(DevCounters), c->stream));
    HIPCHK(c, hipMemsetAsync(c->mr_buf[0] + 5, 0, 11 * sizeof(int), c->stream));      /* running maximum + the "calm waves" words (kernels.h: order_wanted) */
    HIPCHK(c, hipMemsetAsync(c->A.rmap, 0, (size_t)5 * c->A.ntile * sizeof(int), c->stream));
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
    /* launches on caller / ring streams (the un-fused slab phases: k_scatter on stream M) may still be writing State */
    if (c->ext_streams) HIPCHK(c, hipDeviceSynchronize());
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
    if (c->wind_grid_on) {  /* (first: a window the lattice cannot carry is refused before the step has changed anything) */
        HIPCHK(c, hipSetDevice(c->device));
        if (c->ext_streams && !c->ring_orders) HIPCHK(c, hipDeviceSynchronize());   /* previous step (any stream) done with the wind planes */
        int rc = wind_grid_prepare(c, c->clock, dt, c->stream);
        if (rc) return rc;
        /* caller-stream launches wait for the sampler through ev_ctx (step_prologue) */
    }
    c->step_dt = dt;
    c->step_flags = flags;
    c->edge_pending = false;
    c->cur ^= 1;            /* this step's records go to (and are scattered from) rec_buf[cur] */
    c->mr_w = (c->mr_w + 1) % 5;
    c->step_fresh = true;   /* the first advance_rows of the step clears max_reach on ITS stream */
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
    c->ord_valid = false;        /* the stand-alone advance files no dispatch order */
    timing_begin(c, s, 0);
    {
        const KParams &P = c->P;
        bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.p_is_075 && P.deadband2 == 0.0;
        if (!P.wind_static && P.wind_nk > 1) fast = false;     /* a polyline window: the general flavours carry it (same bits: the oracle's one arithmetic) */
        Arrays A = arrays_for(c, c->cur, c->cur);
        StepLaunch L = {dim3(nblocks(nt, 256)), dim3(256), s, &c->P, &c->G, &A, 0.0, 0.0, c->clock, c->step_dt, r0, n0, r1, n1};
        launch_k_advance(L, fast, P.solver, P.wind_static != 0, c->A.pc != nullptr);      /* k_advance.hip */
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
static bool step_fusable(const picles_ctx *c, int flags, double dt)
{
    if (flags != PICLES_STEP_ZERO_FIRST || !c->fuse_steps) return false;
    const KParams &P = c->P;
    /* a polyline window (two or more lattice knots inside the step; host levels set by picles_set_winds_polyline) takes the plain
     * phases: only the general flavours of the stand-alone advance evaluate one */
    if (c->wind_grid_on ? (c->wind_grid_mode != PICLES_LATTICE_SMOOTH3 && picles_lattice_knot_times(c->wg_t0, c->wg_dt, c->clock, dt, nullptr, 0) >= 2)
                        : (!P.wind_static && P.wind_nk > 1)) return false;
    const bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.p_is_075 && P.deadband2 == 0.0;
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
    bool fast = P.propagation && P.input && P.dissipation && P.peak_shift && P.direction && P.n_is_2 && P.p_is_075 && P.deadband2 == 0.0;
    /* cost-ordered dispatch (kernels.h): the launch that covers (nearly) everything — the whole grid of a plain context, the interior
     * rows of a slab — files an order over ITS workgroups and follows the one its predecessor of the same shape filed.  The edge
     * launch of a slab (a handful of workgroups, on the other stream) neither reads nor files one and leaves the chain alone; a
     * stand-alone advance, a re-seed or a change of the launch shape (halo resize) breaks it (its counters are cleared on the way). */
    const bool orders = c->A.ord && (c->G.single_slab ? which == PICLES_ROWS_ALL : which == PICLES_ROWS_INTERIOR);
    if (orders) {
        A.nblk = (int)nblocks(nt, 256);
        const bool follows = c->ord_valid && c->ord_nblk == A.nblk;
        if (!follows) HIPCHK(c, hipMemsetAsync(c->A.ord, 0, (size_t)5 * (2 + c->A.nblk) * sizeof(int), s));
        A.ord_on = follows ? 1 : 0;
        c->ord_valid = true;
        c->ord_nblk = A.nblk;
    } else {
        A.ord = nullptr;            /* (this launch neither reads nor files an order) */
        if (c->G.single_slab || which != PICLES_ROWS_EDGE) c->ord_valid = false;
    }
    timing_begin(c, s, 0);
    {
        StepLaunch L = {dim3(nblocks(nt, 256)), dim3(256), s, &c->P, &c->G, &A, c->pend_t, c->pend_dt, c->clock, c->step_dt, r0, n0, r1, n1};
        if (fast && P.solver == 2) launch_k_step_auto(L, P.wind_static != 0, c->A.pc != nullptr);          /* k_step_auto.hip */
        else launch_k_step_explicit(L, fast, P.solver, P.wind_static != 0, c->A.pc != nullptr);             /* k_step_explicit.hip */
    }
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
    if (!step_fusable(c, PICLES_STEP_ZERO_FIRST, dt)) return 1;
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
            WindowPlan W;
            { int rc = wind_window_plan(c, c->clock, dt, W); if (rc) return rc; }
            if (!A.uP) {
                HIPCHK(c, hipMalloc(&A.uP, (size_t)A.n * 8));
                HIPCHK(c, hipMalloc(&A.vP, (size_t)A.n * 8));
            }
            double *tu = A.uP, *tv = A.vP;
            A.uP = A.u0; A.vP = A.v0;
            A.u0 = A.u1; A.v0 = A.v1;
            A.u1 = tu; A.v1 = tv;
            int rc = wind_window_sample(c, W, c->clock, dt, false, c->stream);
            if (rc) return rc;
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
    c->mr_w = (c->mr_w + 1) % 5;
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
    if (step_fusable(c, flags, dt)) {
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
    const bool region = c->timing && c->timing_mode == 2 && n_steps > 0;
    if (region) {      /* one event pair around the whole call, on the stream the launches go to */
        HIPCHK(c, hipSetDevice(c->device));
        timing_begin(c, c->stream, 3);
        c->in_region = true;
        c->region_launches = 0;
    }
    int rc = 0;
    for (int k = 0; k < n_steps && rc == 0; k++) rc = picles_time_step(c, dt, PICLES_STEP_ZERO_FIRST);
    if (region) {
        c->in_region = false;
        timing_end(c, c->stream);
        c->tim.advance_launches += (uint64_t)c->region_launches;
    }
    return rc;
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
    /* a BLOCKING copy into the caller's (pageable) buffer: when this returns the runtime has written the last byte of it and
     * holds no staging reference to it any more — the caller may free the buffer at once (ctypes / ccall hosts do) */
    HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
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
    if ((rc = d2h(c, &mrt, c->mr_buf[0] + 5, sizeof(int)))) return rc;
    memset(out, 0, sizeof(*out));
    for (const DevCounters &k : d) {
        out->rhs_evals += k.rhs; out->steps_accepted += k.acc; out->steps_rejected += k.rej;
        out->reseeds += k.reseeds; out->clamps += k.clamps; out->maxiters_hits += k.maxit;
        out->particles_advanced += k.adv; out->halo_overflow += k.overflow;
        out->dropped_nonfinite += k.nonfinite;
        out->wave_attempt_slots += k.wslots;
    }
#ifdef PICLES_PHASE_CLOCK
    {
        unsigned long long p0 = 0, p1 = 0, p2 = 0, nw = 0;
        for (const DevCounters &k : d) { p0 += k.pad_[0]; p1 += k.pad_[1]; p2 += k.pad_[2]; nw += k.pad_[3]; }
        if (nw) fprintf(stderr, "PHASE_CLOCK waves %llu: prologue+pull %.0f, RK+guards %.0f, record+stats %.0f cycles per wave (lane 0's view)\n",
                        nw, (double)p0 / nw, (double)p1 / nw, (double)p2 / nw);
    }
#endif
    out->max_reach = mr;
    out->max_reach_seen = mrt;
    return 0;
}

PX_EXPORT int32_t picles_get_dispatch_order(picles_ctx *c, int32_t *out, int32_t cap)
{
    if (!c) return -1;
    if (cap < 0 || (cap > 0 && !out)) return fail(c, -2, "picles_get_dispatch_order: bad buffer");
    if (!c->A.ord || !c->ord_valid) return 0;
    const int n = c->ord_nblk;
    std::vector<int> h((size_t)2 + n);
    /* the buffer the latest step wrote: the counters rotate at the end of a step */
    int rc = d2h(c, h.data(), c->A.ord + (size_t)((c->mr_w + 4) % 5) * (size_t)(2 + n), h.size() * sizeof(int));
    if (rc) return rc;
    if (h[0] + h[1] != n) return 0;
    const int m = cap < 2 + n ? cap : 2 + n;
    for (int k = 0; k < m; k++) out[k] = h[k];
    return n;
}

PX_EXPORT int32_t picles_reset_counters(picles_ctx *c)
{
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    /* no flush: a pending fused step stays pending (its scatter + remesh ride on the next step's launch as usual; the
     * re-seeds of that remesh are then counted in the new window — as the last step of the window leaves its own behind) */
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemsetAsync(c->A.cnt, 0, NSLOTS * sizeof(DevCounters), c->stream));
    HIPCHK(c, hipMemsetAsync(c->mr_buf[0] + 5, 0, sizeof(int), c->stream));
    return 0;
}

PX_EXPORT int32_t picles_enable_timing(picles_ctx *c, int32_t on)
{
    if (!c) return -1;
    timing_collect(c);          /* (does not flush a pending fused step: see picles_reset_counters) */
    c->timing = on != 0;
    c->timing_mode = (on == 2) ? 2 : (on ? 1 : 0);
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
    if (c->ext_streams) HIPCHK(c, hipDeviceSynchronize());   /* an un-fused slab step leaves its scatter on the ring's stream M */
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
    if (!c->G.single_slab && ((c->G.periodic_y && c->G.Ny <= 2 * r) || c->G.ny_loc < r))
        return fail(c, -2, "slab: periodic y axis not longer than 2*halo_rows, or fewer own rows than halo_rows");
    { int rc = flush(c); if (rc) return rc; }
    /* the whole device, not only the context stream: after picles_slab_run_steps the scatter of the last (un-fused) step is
     * still queued on the ring's streams, and the records it reads are about to be re-packed and freed */
    HIPCHK(c, hipDeviceSynchronize());
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
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void *handle = nullptr;
};

/* RCCL by default (the copy torch.distributed has loaded, if any).  PICLES_CCL_LIB names another library that exports the same
 * eight entry points — an MPI-backed shim, or the thread loopback the tests use to run the ring with several ranks on one GPU. */
static RcclApi *rccl_api(std::string &err)
{
    static RcclApi api;
    static bool tried = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (api.handle) return &api;
    if (tried) { err = "the communication library could not be loaded"; return nullptr; }
    tried = true;
    void *h = nullptr;
    const char *over = getenv("PICLES_CCL_LIB");
    if (over && *over) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) { err = std::string("dlopen(PICLES_CCL_LIB=") + over + "): " + dlerror(); return nullptr; }
    } else {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;    /* the copy already loaded, if any */
        if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { err = std::string("dlopen(librccl.so.1): ") + dlerror(); return nullptr; }
    }
#define RSYM(f) do { api.f = (decltype(api.f))dlsym(h, "nccl" #f); if (!api.f) { err = "the communication library lacks nccl" #f; return nullptr; } } while (0)
    RSYM(GetUniqueId); RSYM(CommInitRank); RSYM(CommDestroy); RSYM(GroupStart); RSYM(GroupEnd); RSYM(Send); RSYM(Recv);
    RSYM(GetErrorString);
#undef RSYM
    api.handle = h;
    return &api;
}

struct SlabRing {
    RcclApi *api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, prev = -1, next = -1;      /* -1: no neighbour on that side (open y axis) */
    hipStream_t sE = nullptr, sM = nullptr;
    hipEvent_t evE = nullptr, evM = nullptr, evEd = nullptr;      /* evEd: this step's edge launch alone (without the exchange behind it) */
    unsigned long long steps = 0, exchanged_bytes = 0;
    /* phase timing (picles_slab_get_phases): five events per step while per-launch timing is on */
    struct PhaseEv { hipEvent_t e0, e1, x1, m0, m1; };
    std::vector<PhaseEv> ph_used, ph_free;
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
    if (R->comm) R->api->CommDestroy(R->comm);
    if (R->evE) hipEventDestroy(R->evE);
    if (R->evM) hipEventDestroy(R->evM);
    if (R->evEd) hipEventDestroy(R->evEd);
    for (auto *v : {&R->ph_used, &R->ph_free})
        for (auto &e : *v) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); hipEventDestroy(e.x1); hipEventDestroy(e.m0); hipEventDestroy(e.m1); }
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
    HIPCHK(c, hipEventCreateWithFlags(&R->evEd, hipEventDisableTiming));
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
    /* timing mode 2: ONE event pair around the whole call, on the stream of the interior launches (every step's edge launch and
     * exchange are ordered into it by events: the pair spans the steps); its launches are counted, not bracketed, and the phase
     * events stay away — per-launch pairs and the five phase events of mode 1 cost a slab of 2 M particles 9 % of its step
     * (0.306 -> 0.333 ms, DESIGN.md lab notes of round 4): the diagnosis runs in a pass of its own, outside the timed steps */
    const bool region = c->timing && c->timing_mode == 2 && n_steps > 0;
    if (region) {
        timing_begin(c, R->sM, 3);
        c->in_region = true;
        c->region_launches = 0;
    }
    struct Off {
        picles_ctx *c; SlabRing *R; bool region;
        ~Off() {
            c->ring_orders = false;
            if (region) {      /* also on an error path: the pair is closed and the context leaves the region */
                c->in_region = false;
                timing_end(c, R->sM);
                c->tim.advance_launches += (uint64_t)c->region_launches;
            }
        }
    } off{c, R, region};
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
        const bool phases = c->timing && c->timing_mode == 1 && R->ph_used.size() < (1u << 16);
        SlabRing::PhaseEv pe{};
        if (phases) {
            if (!R->ph_free.empty()) { pe = R->ph_free.back(); R->ph_free.pop_back(); }
            else { hipEventCreate(&pe.e0); hipEventCreate(&pe.e1); hipEventCreate(&pe.x1); hipEventCreate(&pe.m0); hipEventCreate(&pe.m1); }
            HIPCHK(c, hipEventRecord(pe.e0, R->sE));
        }
        int rc = (fused == 0) ? picles_step_rows(c, PICLES_ROWS_EDGE, R->sE) : picles_advance_rows(c, PICLES_ROWS_EDGE, R->sE);
        if (rc) return rc;
        if (fused == 0) HIPCHK(c, hipEventRecord(R->evEd, R->sE));
        if (phases) HIPCHK(c, hipEventRecord(pe.e1, R->sE));
        if ((rc = ring_exchange(c, R, R->sE))) return rc;
        if (phases) { HIPCHK(c, hipEventRecord(pe.x1, R->sE)); HIPCHK(c, hipEventRecord(pe.m0, R->sM)); }
        rc = (fused == 0) ? picles_step_rows(c, PICLES_ROWS_INTERIOR, R->sM) : picles_advance_rows(c, PICLES_ROWS_INTERIOR, R->sM);
        if (rc) return rc;
        if (phases) { HIPCHK(c, hipEventRecord(pe.m1, R->sM)); R->ph_used.push_back(pe); }
        if (fused == 0) {
            /* fused steps: what runs next on M is the next step's interior launch.  Its pull reads own rows only — the records the
             * edge launch of THIS step wrote among them — never the ghost rows: it waits for the edge kernel, not for the exchange
             * behind it.  The communication is off the interior's critical path altogether; only the next edge launch (same
             * stream as the exchange) consumes the received rows. */
            HIPCHK(c, hipStreamWaitEvent(R->sM, R->evEd, 0));
        } else {
            HIPCHK(c, hipEventRecord(R->evE, R->sE));
            HIPCHK(c, hipStreamWaitEvent(R->sM, R->evE, 0));      /* the scatter on M reads the edge rows and the received halo */
        }
        c->edge_pending = false;
        rc = (fused == 0) ? picles_end_fused_step(c) : picles_scatter_remesh(c, R->sM);
        if (rc) return rc;
        R->steps++;
    }
    return 0;
}

PX_EXPORT int32_t picles_slab_get_phases(picles_ctx *c, picles_slab_phases *out)
{
    if (!c || !out) return -1;
    SlabRing *R = c->ring;
    if (!R) return fail(c, -2, "picles_slab_comm_init first");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    picles_slab_phases P{};
    for (auto &e : R->ph_used) {
        float ed = 0.f, ex = 0.f, in = 0.f, sl = 0.f, se = 0.f, sm = 0.f;
        hipEventElapsedTime(&ed, e.e0, e.e1);
        hipEventElapsedTime(&ex, e.e1, e.x1);
        hipEventElapsedTime(&in, e.m0, e.m1);
        hipEventElapsedTime(&sl, e.x1, e.m1);      /* > 0: the interior launch ended after the exchange had completed */
        hipEventElapsedTime(&se, e.e0, e.x1);
        hipEventElapsedTime(&sm, e.e0, e.m1);
        P.steps++;
        if (sl >= 0.f) P.exchange_hidden++;
        P.edge_ms += ed; P.exchange_ms += ex; P.interior_ms += in; P.slack_ms += sl;
        P.span_ms += (se > sm) ? se : sm;
        R->ph_free.push_back(e);
    }
    R->ph_used.clear();
    *out = P;
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
