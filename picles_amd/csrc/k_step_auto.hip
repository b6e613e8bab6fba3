/* k_step_auto.hip — instantiations of the fused step kernel for AutoTsit5(Rosenbrock23()), specialised physics */
#include "kernels.h"
#include "k_step.inc"

#define LAUNCH_STEP(S, M) hipLaunchKernelGGL((k_step<true, true, S, M, true>), L.grid, L.block, 0, L.stream, *L.P, *L.G, *L.A, L.t_prev, L.DT_prev, L.t_start, L.DT, L.r0, L.n0, L.r1, L.n1)
void launch_k_step_auto(const StepLaunch &L, bool wind_static, bool metric)
{
    switch ((wind_static ? 2 : 0) | (metric ? 1 : 0)) {
        case 0: LAUNCH_STEP(false, false); break;
        case 1: LAUNCH_STEP(false, true); break;
        case 2: LAUNCH_STEP(true, false); break;
        case 3: LAUNCH_STEP(true, true); break;
    }
}
