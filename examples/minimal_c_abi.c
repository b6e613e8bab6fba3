/*
 * minimal_c_abi.c — examples/example_00_minimal.jl driven through the C ABI alone (no Python, no torch):
 * 51x51 box of 100 km, constant winds (10,10), 13 steps of 10 minutes, run!-style (State zeroed first).
 *
 *   gcc -O2 -I include examples/minimal_c_abi.c -o /tmp/minimal_c_abi -L picles_amd/csrc -lpicles_hip \
 *       -Wl,-rpath,$PWD/picles_amd/csrc -lm && /tmp/minimal_c_abi
 *
 * The parameter values are those of particle_waves_v5.jl ODEParameters(r_g = 0.85) / IDConstants() and of
 * FetchRelations.MinimalState(2, 2, 600) (what WaveGrowth2D defaults to, WaveGrowthModels2D.jl:234-246).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "picles_hip.h"

#define CHECK(call)                                                                           \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != 0) {                                                                       \
            fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, picles_last_error(ctx));   \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(void)
{
    const int N = 51;
    const double L = 100e3, DT = 600.0;
    picles_ctx *ctx = NULL;

    picles_grid g = {N, N, L / (N - 1), L / (N - 1), 0, 0, NULL, 0, N};
    /* IDConstants(): c_D 2e-3, c_beta 4e-2, c_e 1.3e-6, c_alpha 11.8, r_w 2.35, q -1/4 */
    const double r_g = 0.85, c_D = 2e-3, c_beta = 4e-2, c_e = 1.3e-6, c_alpha = 11.8, r_w = 2.35, q = -0.25;
    const double p = (-1 - 10 * q) / 2;
    const double C_e = r_w * c_beta * c_D / r_g;
    const double gamma = 1 - (p - q) / (pow(c_alpha, 4) * C_e * 2);
    picles_phys ph = {r_g, -1.41, 1.81e-5, C_e, 9.81, gamma, q, c_beta, c_D, c_e, c_alpha, 1, 1, 1, 1, 1, 0, 0.0};
    picles_ode od = {1e-4, 1e-3, 1e-3, 1e-4, 1, /*solver: Tsit5*/ 1, 10000, -13.589885017354083, log(17.0), 4.0, DT};
    picles_model md = {0, 0, {0, 0, 0}, {1.253106339976604e-6, 1.2821164e-9}};

    int rc = picles_create(&g, &ph, &od, &md, 0, 1, &ctx);
    if (rc) { fprintf(stderr, "picles_create failed (rc=%d): %s\n", rc, picles_last_error(NULL)); return 1; }

    double *u = malloc(sizeof(double) * N * N), *v = malloc(sizeof(double) * N * N);
    double *state = malloc(sizeof(double) * N * N * 3);
    for (int k = 0; k < N * N; k++) { u[k] = 10.0; v[k] = 10.0; }
    CHECK(picles_set_winds(ctx, u, v, 0.0, NULL, NULL, 0.0));
    CHECK(picles_seed(ctx, 0.0));
    for (int step = 0; step < 13; step++) CHECK(picles_time_step(ctx, DT, PICLES_STEP_ZERO_FIRST));
    CHECK(picles_get_state(ctx, state));
    picles_counters c;
    CHECK(picles_get_counters(ctx, &c));
    double e = state[25 + N * 25];
    printf("clock %.0f s, Hs(centre) = %.6f m, RHS evaluations %llu, accepted RK steps %llu\n",
           picles_clock(ctx), 4 * sqrt(e), (unsigned long long)c.rhs_evals, (unsigned long long)c.steps_accepted);
    picles_destroy(ctx);
    free(u); free(v); free(state);
    return !(e > 0.2 && e < 0.3);
}
