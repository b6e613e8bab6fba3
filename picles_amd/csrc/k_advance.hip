/* k_advance.hip — the stand-alone advance kernel and its instantiations */
/* every flavour parks (particle index, node, flags) in LDS and reads its arguments again behind the RK loop instead of keeping
 * them in scalar registers across it (k_step.inc has the reasons): scalar spills 61-105 -> 12-18 in the general-physics and
 * auto-switching flavours (the general-physics auto-switching pair keeps ~100: its Jacobian reads most of KParams).  The Butcher
 * tableau goes through scalar loads in the four-wave flavours (FAST && !AUTO), through LDS in the others (physics.h, TABS). */
#include "kernels.h"

/* the kernel arguments as the kernarg segment lays them out (kargs_reload, kernels.h: behind the RK loop they are read again
 * instead of living through it in SGPRs) */
struct KAdvArgs {
    KParams P;
    GridP G;
    Arrays A;
    double t_start, DT;
    int r0, n0, r1, n1;
};
static_assert(__builtin_offsetof(KAdvArgs, G) == sizeof(KParams) && __builtin_offsetof(KAdvArgs, A) % 8 == 0 &&
              __builtin_offsetof(KAdvArgs, t_start) == __builtin_offsetof(KAdvArgs, A) + sizeof(Arrays) &&
              __builtin_offsetof(KAdvArgs, r0) == __builtin_offsetof(KAdvArgs, DT) + 8,
              "KAdvArgs must mirror the kernarg layout of k_advance");

/* ------------------------------------------------------------------------------------------
 * k_advance — advance! for the particles of the given rows.  One thread per particle; the whole
 * adaptive RK loop runs in registers.  Writes the particle's scatter record instead of
 * scattering: the scatter itself is k_scatter / k_step / k_push_tiles.
 * ---------------------------------------------------------------------------------------- */
template <bool FAST, bool STATIC, bool METRIC, bool TSIT, bool AUTO>
__global__ void __launch_bounds__(256, (FAST && !AUTO) ? 4 : 2) k_advance(KParams P, GridP G, Arrays A, double t_start, double DT,
                                                   int r0, int n0, int r1, int n1)
{
    dp_device_init(TSIT ? 1 : 0);
    pm_device_init();
    long long t = 0;
    bool active = rows_index(G, r0, n0, r1, n1, t, xcd_block());
    if (active) rmap_clear_ahead(A, t);      /* every node of the rows, stepped or not: the clear must reach every tile */
    unsigned char pf = active ? A.pflags[t] : 0;
    active = active && (pf & PF_STEPPED);
    StepStats S = {{0u, 0u, 0u, 0}, 0u, 0u, 0u, 0u, 0u, 0u, 0};
    int rtile = -1;
    if (active) {
        int i = (int)(t % G.Nx), jl = (int)(t / G.Nx);
        Vec5 z;
        z.lne = A.z[t]; z.cx = A.z[t + A.n]; z.cy = A.z[t + 2 * A.n]; z.x = A.z[t + 3 * A.n]; z.y = A.z[t + 4 * A.n];
        int on = A.on[t];
        double qold = A.qold[t], dtn = A.dtn[t];
        constexpr bool POLY = !FAST && !STATIC;      /* the general time-varying flavours carry polyline windows (physics.h, wind_eval) */
        Wind w = load_wind<POLY>(P, A, t);
        int status;
        int asw = AUTO ? A.asw[t] : 0;
        /* behind the RK loop: guards, particle store, scatter record */
        auto finish = [&](const KParams &Pq, const GridP &Gq, const Arrays &Aq, long long tq, int iq, int jlq, unsigned char pfq,
                          double ts, double dt_step, int status) {
            status = advance_guards<POLY>(Pq, [&]() { return load_wind<POLY>(Pq, Aq, tq); }, z, dtn, ts, dt_step, status, S);
            if (AUTO) Aq.asw[tq] = asw;
            Aq.z[tq] = z.lne; Aq.z[tq + Aq.n] = z.cx; Aq.z[tq + 2 * Aq.n] = z.cy; Aq.z[tq + 3 * Aq.n] = z.x; Aq.z[tq + 4 * Aq.n] = z.y;
            Aq.on[tq] = (unsigned char)on;
            Aq.qold[tq] = qold;
            Aq.dtn[tq] = dtn;
            Aq.status[tq] = status;
            write_record(Gq, Aq, iq, jlq, pfq, on, z, S);
            rtile = (int)(tq >> 6);
        };
        {
            /* what is needed again only behind the loop waits in LDS, and the kernel arguments are read
             * again from the kernarg segment (k_step.inc has the reasons) */
            constexpr bool TABS = FAST && !AUTO;
            __shared__ int stash_[5][256];
            const int tid_ = threadIdx.x;
            stash_[0][tid_] = (int)(unsigned int)((unsigned long long)t & 0xffffffffull); stash_[1][tid_] = (int)((unsigned long long)t >> 32);
            stash_[2][tid_] = i; stash_[3][tid_] = jl; stash_[4][tid_] = (int)pf;
            if (METRIC) status = advance_core<FAST, STATIC, true, TSIT, AUTO, TABS>(P, w, z, on, qold, dtn, t_start, DT, S, A.m11[t], A.m22[t], A.pc[t], &asw);
            else status = advance_core<FAST, STATIC, false, TSIT, AUTO, TABS>(P, w, z, on, qold, dtn, t_start, DT, S, 0.0, 0.0, 0.0, &asw);
            const KAdvArgs *K = (const KAdvArgs *)kargs_reload();
            const KParams Pb = K->P;
            const GridP Gb = K->G;
            const Arrays Ab = K->A;
            __asm__ volatile("" ::: "memory");
            const int tid2_ = threadIdx.x;
            const long long tb = (long long)(((unsigned long long)(unsigned int)stash_[1][tid2_] << 32) | (unsigned int)stash_[0][tid2_]);
            finish(Pb, Gb, Ab, tb, stash_[2][tid2_], stash_[3][tid2_], (unsigned char)(stash_[4][tid2_] & 0xff), K->t_start, K->DT, status);
        }
    }
    flush_stats(A, S, rtile);
}

#define LAUNCH_ADV(F, S, M) do { if (solver == 2) hipLaunchKernelGGL((k_advance<F, S, M, true, true>), L.grid, L.block, 0, L.stream, *L.P, *L.G, *L.A, L.t_start, L.DT, L.r0, L.n0, L.r1, L.n1); \
                                 else if (solver) hipLaunchKernelGGL((k_advance<F, S, M, true, false>), L.grid, L.block, 0, L.stream, *L.P, *L.G, *L.A, L.t_start, L.DT, L.r0, L.n0, L.r1, L.n1); \
                                 else hipLaunchKernelGGL((k_advance<F, S, M, false, false>), L.grid, L.block, 0, L.stream, *L.P, *L.G, *L.A, L.t_start, L.DT, L.r0, L.n0, L.r1, L.n1); } while (0)
void launch_k_advance(const StepLaunch &L, bool fast, int solver, bool wind_static, bool metric)
{
    if (metric) LAUNCH_ADV(false, false, true);   /* per-node metric: the general code path */
    else if (fast && wind_static) LAUNCH_ADV(true, true, false);
    else if (fast) LAUNCH_ADV(true, false, false);
    else LAUNCH_ADV(false, false, false);          /* general physics; static winds are the du = dv = 0 case of the same code (same bits) */
}
