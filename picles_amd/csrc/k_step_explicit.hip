/* k_step_explicit.hip — instantiations of the fused step kernel for the explicit 5(4) pairs (DP5, Tsit5) */
#include "kernels.h"
#include "k_step.inc"

#define LAUNCH_STEP(F, T, S, M) hipLaunchKernelGGL((k_step<F, T, S, M, false>), L.grid, L.block, 0, L.stream, *L.P, *L.G, *L.A, L.t_prev, L.DT_prev, L.t_start, L.DT, L.r0, L.n0, L.r1, L.n1)
void launch_k_step_explicit(const StepLaunch &L, bool fast, int solver, bool wind_static, bool metric)
{
    if (!fast) {     /* general physics: static winds, Cartesian (step_fusable) */
        if (solver) LAUNCH_STEP(false, true, true, false);
        else LAUNCH_STEP(false, false, true, false);
        return;
    }
    const int key = (solver ? 4 : 0) | (wind_static ? 2 : 0) | (metric ? 1 : 0);
    switch (key) {
#define CASE_STEP(k, T, S, M) case k: LAUNCH_STEP(true, T, S, M); break;
        CASE_STEP(0, false, false, false) CASE_STEP(1, false, false, true) CASE_STEP(2, false, true, false) CASE_STEP(3, false, true, true)
        CASE_STEP(4, true, false, false)  CASE_STEP(5, true, false, true)  CASE_STEP(6, true, true, false)  CASE_STEP(7, true, true, true)
#undef CASE_STEP
    }
}
