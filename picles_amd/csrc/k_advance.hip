/* k_advance.hip — the stand-alone advance kernel and its instantiations */
#include "kernels.h"

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
