/* k_step_auto.hip — instantiations of the fused step kernel for AutoTsit5(Rosenbrock23()), specialised physics */
#define PICLES_TABLEAU_SMEM(FAST, AUTO) true      /* Butcher tableau through scalar loads (physics.h): 100 -> 60 B/lane of scratch at three waves */
#define PICLES_PREFETCH 1                           /* flags, wind, controller memory are loaded ahead of the pull (k_step.inc): -1 % at three waves per SIMD, +4 % at four */
#define PICLES_ROS_KARGS 1                          /* the Rosenbrock23 branch re-reads KParams from the kernarg segment (physics.h) */
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
