/*
 * physics.h — per-particle arithmetic of the PiCLES 2D time step for gfx950 (device inline code).
 *
 * Everything one thread does to one particle lives here, in registers:
 *   seed_windsea      FetchRelations.get_initial_windsea            (FetchRelations.jl:314-359)
 *   rhs               particle_system(dz,z,params,t), Cartesian mesh (particle_waves_v5.jl:479-556)
 *   integrate_dp5     step!(integrator, DT, true): DP5 / Tsit5 / AutoTsit5(Rosenbrock23()) + PI controller + Hairer
 *                     initial dt (call site mapping_2D.jl:152; OrdinaryDiffEq semantics: SURVEY Appendix C)
 *   rhs3_jac_plain, rhs3_jvp, ros23_try   exact Jacobian of the RHS (structured / forward mode) and the Rosenbrock23 attempt of the auto-switching solver
 *   particle_to_charge / charge_to_particle   core_2D.jl:69-78 / :121-128
 *   index_weight      get_absolute_i_and_w(z, i_node)                 (ParticleInCell.jl:58-71)
 *
 * The evaluation order ("kernel order") is chosen for the CDNA4 fp64 VALU, where every instruction
 * costs one issue slot: y = 1/|c̄| through a deterministic reciprocal square root (no sqrt, no division) feeds k_p, ω_p,
 * α², α_p and the direction term, the powers of r_g ride in constants; |g| ≡ c_gp; sin 2(θ_c-θ_w) = 2·cross·dot·y²/U²;
 * the reference's guards (speed floors, the cap on α, zero tests) sit behind ONE wave-uniform test per evaluation ("plain"
 * particles); H_β and Δ_β share one exponential and one reciprocal; the error norm takes one reciprocal and no square root;
 * the tableau is read through scalar loads at the point of use; every multiply-add that is fused is written as fma() and
 * the TU is compiled with -ffp-contract=off, so results are bit-identical to the CPU oracle built with the same
 * primitives (oracle order 1).  No MFMA: nothing here is a contraction.
 */
#ifndef PICLES_PHYSICS_H
#define PICLES_PHYSICS_H

#include "pmath.h"

#define PK_G0 9.81
#define PK_PI 3.14159265358979323846

/* kernel parameters: lives in SGPRs / scalar cache (passed by value as a kernel argument) */
struct KParams {
    /* physics */
    double r_g, inv_rg, C_alpha, C_phi, C_e;
    double p, n, neg2p, inv_eT;
    double inv_eT4, half_inv_rg, two_inv_rg2;   /* (1/e_T)⁴, 1/(2 r_g), 2/r_g² */
    double KeT4, KrCa, Cdir;                    /* K/e_T⁴, K r_g C_α with K = (g/4)⁴ (k_p⁴ = K·(1/c_gp)⁸); 2 C_φ/r_g² */
    /* the RHS works on y = 1/|c̄| (1/c_gp = r_g y): every power of r_g rides in a constant (rhs3) */
    double rg2;                                 /* r_g² */
    double Cw, Chrh;                            /* ω_p = Cw·min(y, ymax), Cw = (g/2) r_g;  -α_p/2 = (Chrh·c̄·u)·min(y², sgmax), Chrh = -r_g/4 */
    double ymax, sgmax;                         /* 10/r_g, 1e8/r_g²: the reference's speed floors 0.1 and 1e-4 as ceilings on y and y² */
    double KeT4y, KrCay;                        /* KeT4 r_g⁸, KrCa r_g⁸ */
    double Cs, Cdir2;                           /* C_φ r_g²/2 (direction term of a plain particle), 2 C_φ (… of a clamped one, times 1/U²) */
    double g4rg2;                               /* (g/4) r_g²: k_p = g4rg2·min(y, ymax)² (general n) */
    double qU2r_max;                            /* a plain wind has (U²/4) r_g² <= qU2r_max: α² = (U²/4) r_g² y² stays below 500² for every y <= ymax */
    double inv_dx, inv_dy;
    double deadband2;       /* dir_deadband² (0 = off) */
    int propagation, input, dissipation, peak_shift, direction, n_is_2;
    int p_is_075;           /* 2p = 3/2 (q = -1/4, every reference script): exp(-2p y) and exp(-20|y|) are powers of one exponential */
    int fast_phys;          /* the specialised physics (every switch on, n = 2, 2p = 3/2, no dead band): what the FAST flavours are built for.
                             * A general-physics kernel that runs it (the per-node-metric flavour of the stand-alone advance) must take the
                             * same arithmetic wherever FAST chooses between FORMS — the Jacobian of the Rosenbrock23 attempt — or an observed
                             * step (k_advance) and a fused one (k_step) of the same model would differ in the last bits */
    /* ODE settings */
    double abstol, reltol, dt0, dtmin;
    double inv_abstol;       /* 1/abstol: the error scale of a component that is exactly 0 */
    long long maxiters;
    int force_dtmin;
    int solver;              /* 0 DP5, 1 Tsit5 (selects the kernel instantiation) */
    double lne_max, wind_min_sq;
    /* model */
    int init_type;
    double def_lne, def_cx, def_cy;
    double min_e, min_m2;
    /* winds */
    int wind_static;
    double tw0, inv_dtw;
    /* three-level windows come in two forms (physics.h, Wind): wind_sk == 0: the parabola through (t0, mid, t1) — node samples of a
     * smooth closure; 0 < wind_sk < 1: the piecewise-linear function with ONE knot at s = wind_sk — the exact form of a gridded wind
     * (Utils/WindEmulator.jl:18-43, linear_interpolation in t) one of whose time knots falls inside the model step */
    double wind_sk, wind_isk, wind_i1sk;   /* s of the knot, 1/sk, 1/(1 - sk) */
    /* a window with TWO OR MORE knots inside the step (wind_nk >= 2): the polyline  u(s) = u0 + s du + Σ_k max(s - s_k, 0) b_k  with
     * one term per knot.  Its coefficients are laid down once per step by k_wind_poly (picles_hip.hip) in wind_xb:
     * [0 .. PICLES_MAX_KNOTS) the s_k, then planes of wind_xn doubles: du, dv, b_1u, b_1v, b_2u, b_2v, ...  Only the general
     * flavours of the stand-alone phases carry such a window (wind_eval<true>); the fused step never meets one (step_fusable). */
    int wind_nk, wind_pad_;
    const double *wind_xb;
    long long wind_xn;
};
#ifndef PICLES_MAX_KNOTS
#define PICLES_MAX_KNOTS 8
#endif

/* Dormand–Prince 5(4) */
#define DP_A21 (1.0 / 5.0)
#define DP_A31 (3.0 / 40.0)
#define DP_A32 (9.0 / 40.0)
#define DP_A41 (44.0 / 45.0)
#define DP_A42 (-56.0 / 15.0)
#define DP_A43 (32.0 / 9.0)
#define DP_A51 (19372.0 / 6561.0)
#define DP_A52 (-25360.0 / 2187.0)
#define DP_A53 (64448.0 / 6561.0)
#define DP_A54 (-212.0 / 729.0)
#define DP_A61 (9017.0 / 3168.0)
#define DP_A62 (-355.0 / 33.0)
#define DP_A63 (46732.0 / 5247.0)
#define DP_A64 (49.0 / 176.0)
#define DP_A65 (-5103.0 / 18656.0)
#define DP_A71 (35.0 / 384.0)
#define DP_A73 (500.0 / 1113.0)
#define DP_A74 (125.0 / 192.0)
#define DP_A75 (-2187.0 / 6784.0)
#define DP_A76 (11.0 / 84.0)
#define DP_C2 (1.0 / 5.0)
#define DP_C3 (3.0 / 10.0)
#define DP_C4 (4.0 / 5.0)
#define DP_C5 (8.0 / 9.0)
#define DP_E1 (-71.0 / 57600.0)
#define DP_E3 (71.0 / 16695.0)
#define DP_E4 (-71.0 / 1920.0)
#define DP_E5 (17253.0 / 339200.0)
#define DP_E6 (-22.0 / 525.0)
#define DP_E7 (1.0 / 40.0)

/* PI controller (OrdinaryDiffEq defaults: DP5 beta2 = 4//100, beta1 = 1//5 - 3beta2/4; Tsit5 and
 * every other order-5 method beta1 = 7//50, beta2 = 2//25) */
#define PI_BETA1 0.17
#define PI_BETA2 0.04
#define PI_BETA1_TSIT 0.14
#define PI_BETA2_TSIT 0.08
#define PI_GAMMA 0.9
#define PI_QMIN 0.2
#define PI_QMAX 10.0
#define PI_QOLDINIT 1e-4
#define PI_LNQOLDINIT (-9.210340371976182) /* ln(1e-4): the carried controller memory is ln(qold) */

/* Tsitouras 5(4) (OrdinaryDiffEq Tsit5; Tsitouras 2011).  The coefficients satisfy all 17 order-5
 * conditions for b = a7* and the order-4 conditions for b - btilde to 1e-15 (tests/test_tableaux.py). */
#define TS_C2 0.161
#define TS_C3 0.327
#define TS_C4 0.9
#define TS_C5 0.9800255409045097
#define TS_A21 0.161
#define TS_A31 (-0.008480655492356989)
#define TS_A32 0.335480655492357
#define TS_A41 2.8971530571054935
#define TS_A42 (-6.359448489975075)
#define TS_A43 4.3622954328695815
#define TS_A51 5.325864828439257
#define TS_A52 (-11.748883564062828)
#define TS_A53 7.4955393428898365
#define TS_A54 (-0.09249506636175525)
#define TS_A61 5.86145544294642
#define TS_A62 (-12.92096931784711)
#define TS_A63 8.159367898576159
#define TS_A64 (-0.071584973281401)
#define TS_A65 (-0.028269050394068383)
#define TS_A71 0.09646076681806523
#define TS_A72 0.01
#define TS_A73 0.4798896504144996
#define TS_A74 1.379008574103742
#define TS_A75 (-3.290069515436081)
#define TS_A76 2.324710524099774
#define TS_E1 (-0.00178001105222577714)
#define TS_E2 (-0.0008164344596567469)
#define TS_E3 0.007880878010261995
#define TS_E4 (-0.1447110071732629)
#define TS_E5 0.5823571654525552
#define TS_E6 (-0.45808210592918697)
#define TS_E7 0.015151515151515152

/* The tableau as one object.  On the device it is read once per thread from LDS into VGPRs
 * before the RK loop: as 64-bit literals the constants would need > 60 SGPRs, overflow the
 * scalar file and be spilled to VGPR lanes (v_readlane/v_writelane = VALU slots in the hot loop). */
struct DPTab {
    double a21, a31, a32, a41, a42, a43, a51, a52, a53, a54, a61, a62, a63, a64, a65;
    double a71, a72, a73, a74, a75, a76, c2, c3, c4, c5, e1, e2, e3, e4, e5, e6, e7;
};
#define DPTAB_DP5                                                                                    \
    {DP_A21, DP_A31, DP_A32, DP_A41, DP_A42, DP_A43, DP_A51, DP_A52, DP_A53, DP_A54, DP_A61, DP_A62,  \
     DP_A63, DP_A64, DP_A65, DP_A71, 0.0, DP_A73, DP_A74, DP_A75, DP_A76, DP_C2, DP_C3, DP_C4, DP_C5, \
     DP_E1, 0.0, DP_E3, DP_E4, DP_E5, DP_E6, DP_E7}
#define DPTAB_TSIT5                                                                                  \
    {TS_A21, TS_A31, TS_A32, TS_A41, TS_A42, TS_A43, TS_A51, TS_A52, TS_A53, TS_A54, TS_A61, TS_A62,  \
     TS_A63, TS_A64, TS_A65, TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76, TS_C2, TS_C3, TS_C4, TS_C5, \
     TS_E1, TS_E2, TS_E3, TS_E4, TS_E5, TS_E6, TS_E7}
#define DPTAB_N 32

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const double DPTAB_C[2][DPTAB_N] = {DPTAB_DP5, DPTAB_TSIT5};
__device__ __forceinline__ double *dp_lds_tab(void)
{
    __shared__ double tab[DPTAB_N];
    return tab;
}
__device__ __forceinline__ void dp_device_init(int solver)   /* call before pm_device_init (shares its barrier) */
{
    if (threadIdx.x >= 32 && threadIdx.x < 32 + DPTAB_N) dp_lds_tab()[threadIdx.x - 32] = DPTAB_C[solver][threadIdx.x - 32];
}
__device__ __forceinline__ void dp_load(DPTab &T, int)
{
    const double *t = dp_lds_tab();
    double *o = &T.a21;
#pragma unroll
    for (int k = 0; k < DPTAB_N; k++) o[k] = t[k];
}
#else
#if defined(__HIPCC__)
__device__ __forceinline__ void dp_device_init(int) {}
#endif
PM_HD void dp_load(DPTab &T, int solver)
{
    const DPTab c0 = DPTAB_DP5, c1 = DPTAB_TSIT5;
    T = solver ? c1 : c0;
}
#endif

/* tableau access inside the RK loop, three forms.  The stand-alone advance kernel: every use is an LDS read (ds_read_b64 with a
 * wave-uniform address: the LDS port is otherwise idle, and the 26 live constants would cost 52 VGPRs); the fused step kernels
 * (k_step_explicit.hip, k_step_auto.hip): scalar loads (below); on the host a plain struct */
#if defined(__HIP_DEVICE_COMPILE__)
/* TABS (template parameter of integrate_dp5 / advance_core): the instantiation takes the scalar-load form.
 * Scalar-load form: every use is a SCALAR load from constant memory, issued stage by stage behind an opaque copy of the table
 * pointer (TT_STAGE: the loads of a stage cannot be hoisted above it, so at most one stage's coefficients are live: ~16 SGPRs), and
 * enters the fma as its scalar operand.  No VGPR ever holds a coefficient — with the LDS form the constants in flight cost ~30
 * VGPRs — which is what lets the DP5 / Tsit5 kernels fit 128 registers = FOUR waves per SIMD (2.40 -> 2.28 ms on the BASELINE box,
 * same instruction count; the default-solver step kernels: 100 -> 60 B/lane of scratch at three waves).  As literals (s_mov at the
 * point of use) the same constants overflowed the scalar file: 44 spilled SGPRs, +40 VALU slots per attempt, +5 %.  Every barrier
 * starts from the table's address again, not from the previous copy: no value is carried across the (divergent) branches of the
 * auto-switching kernel, whose merge would otherwise turn the pointer into a per-lane value.  The form needs scalar registers to
 * spare: every fused step kernel takes it; of the stand-alone advance kernels only the four-wave flavours (specialised physics,
 * explicit pair) — the general-physics and auto-switching ones keep all their arguments live and stay with LDS. */
typedef const __attribute__((opencl_constant)) double *dp_cptr;
__device__ __forceinline__ dp_cptr dp_launder(dp_cptr p) { __asm__ volatile("" : "+s"(p)); return p; }
#define DP_TAB_DECL(solver)                                                                                      \
    constexpr int dp_solver_ = (solver); constexpr bool dp_smem_ = TABS;                                          \
    dp_cptr dp_ctab_ = (dp_cptr)&DPTAB_C[dp_solver_][0]; const double *const dp_ltab_ = dp_lds_tab()
#define TT_STAGE() do { if constexpr (dp_smem_) dp_ctab_ = dp_launder((dp_cptr)&DPTAB_C[dp_solver_][0]); } while (0)
#define TT(f) (dp_smem_ ? dp_ctab_[__builtin_offsetof(DPTab, f) / 8] : dp_ltab_[__builtin_offsetof(DPTab, f) / 8])
#else
#define DP_TAB_DECL(solver) DPTab T; dp_load(T, solver)
#define TT_STAGE() ((void)0)
#define TT(f) (T.f)
#endif

struct Vec5 {
    double lne, cx, cy, x, y;
};

/* node wind over the step window [tw0, tw1], s = (t - tw0)/(tw1 - tw0):  u(s) = u0 + s (du + (s - 1) bu)  — the parabola through
 * the three levels u0 = u(tw0), um = u((tw0+tw1)/2), u1 = u(tw1) in Newton form, du = u1 - u0, bu = 2 ((u0 + u1) - 2 um).
 * Two-level winds (picles_set_winds, a lattice window without a knot) have bu = bv = 0 and are the straight line, bit for bit what
 * the two-level code computed: fma(0, s - 1, du) = du.  The reference calls the closures u(x,y,t), v(x,y,t) at every stage time
 * (particle_waves_v5.jl:494-495); for a forcing of angular frequency ω the line misses it by (ω Δt)²/8 of its amplitude at
 * mid-step, the parabola by (ω Δt)³/125 (T04_2D_reg_test.jl:167 with Δt = 20 min: 3.2e-3 against 3.2e-5). */
/* The other three-level form (P.wind_sk in (0,1)): a gridded wind is piecewise linear in t with kinks at the lattice's time knots, and
 * the RHS sees the kink when a knot falls inside the step (the reference evaluates linear_interpolation((x,y,t), u) at every stage
 * time).  With the knot at s = sk and the node's level there, uk:  u(s) = u0 + s du + max(s - sk, 0) bu,  du = (uk - u0)/sk the
 * first segment's slope and bu = (u1 - uk)/(1 - sk) - du the change of slope at the knot.  Same six numbers per node, one kernel-
 * uniform (scalar) branch per evaluation; like the parabola it returns level 0 itself at s = 0 (the fused step's remesh reads that
 * level from its plane, the stand-alone remesh evaluates the window at its start: the two must agree to the bit). */
struct Wind {
    double u0, v0, du, dv; /* level 0 and (level1 - level0) [knot form: the first segment's slope per unit s] */
    double bu, bv;         /* curvature term of the three-level window (0: linear in t) [knot form: the slope's jump at the knot] */
    unsigned int xi;       /* the node (polyline windows read the terms of their further knots from wind_xb; dead in every other flavour) */
};
/* The parabola keeps the arithmetic of the two- and three-level windows of rounds 1-3, u0 + s (du + (s - 1) bu); the knot form sits
 * behind ONE kernel-uniform (scalar) branch per evaluation that covers both components.  Measured on MI355X, default-solver flavour,
 * same box (gpurun_out/r4c/ab_wind_form*.log): this shape costs the time-varying kernels 1.3 % over the code without a knot form;
 * a common formula with a shape function g(s) selected per stage cost 8-12 % (the 168-register kernel sits at the edge of its
 * allocation: one more live value across the stage evaluations moved spill code into the RK loop). */
PM_HD void wind_eval2(const KParams &P, const Wind &w, double s, double &u, double &v)
{
    if (P.wind_sk > 0.0) {
        double sp = s - P.wind_sk;
        sp = (sp > 0.0) ? sp : 0.0;
        u = PM_FMA(w.bu, sp, PM_FMA(w.du, s, w.u0));
        v = PM_FMA(w.bv, sp, PM_FMA(w.dv, s, w.v0));
    } else {
        const double s1 = s - 1.0;
        u = PM_FMA(PM_FMA(w.bu, s1, w.du), s, w.u0);
        v = PM_FMA(PM_FMA(w.bv, s1, w.dv), s, w.v0);
    }
}
/* d/ds of the same */
PM_HD double wind_slope(const KParams &P, double du, double bu, double s)
{
    if (P.wind_sk > 0.0) return (s >= P.wind_sk) ? du + bu : du;
    return PM_FMA(bu, PM_FMA(2.0, s, -1.0), du);
}
/* The polyline (KParams::wind_nk >= 2): a gridded wind with several of its time knots inside the model step — the reference's
 * linear_interpolation((x,y,t), u) (Utils/WindEmulator.jl:18-43) is piecewise linear in t with a kink at every one of them.  The knot
 * form with one more term per further knot, summed in the order of the knots; w.du, w.bu hold the first segment's slope and the first
 * knot's jump, the jumps of the further knots are read from wind_xb when they are needed (a rare window: nothing is kept in
 * registers for it).  POLY == false compiles to wind_eval2: the fused kernels and the specialised flavours do not carry the branch. */
template <bool POLY>
PM_HD void wind_eval(const KParams &P, const Wind &w, double s, double &u, double &v)
{
    if (POLY && P.wind_nk > 1) {
        const double *xb = P.wind_xb;
        double sp = s - xb[0];
        sp = (sp > 0.0) ? sp : 0.0;
        u = PM_FMA(w.bu, sp, PM_FMA(w.du, s, w.u0));
        v = PM_FMA(w.bv, sp, PM_FMA(w.dv, s, w.v0));
        for (int k = 1; k < P.wind_nk; k++) {
            sp = s - xb[k];
            sp = (sp > 0.0) ? sp : 0.0;
            const double *pl = xb + PICLES_MAX_KNOTS + (long long)(2 * (k + 1)) * P.wind_xn + w.xi;
            u = PM_FMA(pl[0], sp, u);
            v = PM_FMA(pl[P.wind_xn], sp, v);
        }
        return;
    }
    wind_eval2(P, w, s, u, v);
}
/* d/ds of a polyline window, both components */
PM_HD void wind_slopes_poly(const KParams &P, const Wind &w, double s, double &su, double &sv)
{
    const double *xb = P.wind_xb;
    su = (s >= xb[0]) ? w.du + w.bu : w.du;
    sv = (s >= xb[0]) ? w.dv + w.bv : w.dv;
    for (int k = 1; k < P.wind_nk; k++) {
        const double *pl = xb + PICLES_MAX_KNOTS + (long long)(2 * (k + 1)) * P.wind_xn + w.xi;
        if (s >= xb[k]) { su = su + pl[0]; sv = sv + pl[P.wind_xn]; }
    }
}

struct PStats {
    unsigned int rhs, acc, rej;
    int status;
};

/* node wind at absolute time t plus the quantities every RHS evaluation derives from it.
 * For time-constant winds they are computed once per particle-step (same arithmetic, hoisted). */
struct WindD {
    double u, v;
    double qU2r;                   /* (U²/4) r_g²: α² = qU2r y² — the wind speed itself is never needed (α enters as α²) */
    double sh;                     /* pm_exp's shifter in a vector register pair (PM_EXP_SHIFTER(), set once by the integrator) */
    double ymaxw;                  /* WAVE-uniform: ymax where every lane's wind is plain (wind_is_plain), -1 otherwise — so that the
                                    * one test "every lane has y <= ymaxw" of rhs3 covers the wind as well */
};
/* a PLAIN wind: U² an ordinary positive number and small enough that α = U/(2 c_gp) cannot reach its cap of 500 while c_gp is above its
 * floor (y <= ymax); with a plain y as well (rhs3) none of the reference's guards acts on the particle */
PM_HD bool wind_is_plain(const KParams &P, double U2, double qU2r) { return U2 >= 1e-290 && qU2r <= P.qU2r_max; }
PM_HD void wind_derive(const KParams &P, double u, double v, WindD &d)
{
    d.u = u;
    d.v = v;
    const double U2 = PM_FMA(u, u, v * v);       /* recomputed from (u, v) where a rare path needs it: not kept */
    d.qU2r = (0.25 * U2) * P.rg2;
    d.ymaxw = PM_WAVE_ALL(wind_is_plain(P, U2, d.qU2r)) ? P.ymax : -1.0;
}
/* the same for the stage winds of a time-varying window (seven times per RK attempt) */
PM_HD void wind_derive_stage(const KParams &P, double u, double v, WindD &d) { wind_derive(P, u, v, d); }
template <bool POLY = false>
PM_HD void wind_at(const KParams &P, const Wind &w, double t, double &u, double &v)
{
    if (P.wind_static) {
        u = w.u0;
        v = w.v0;
    } else {
        double s = (t - P.tw0) * P.inv_dtw;
        wind_eval<POLY>(P, w, s, u, v);
    }
}
template <bool STATIC, bool POLY = false>
PM_HD void wind_stage(const KParams &P, const Wind &w, double t, WindD &d)
{
    if (!STATIC) {
        double s = (t - P.tw0) * P.inv_dtw;
        double u, v;
        wind_eval<POLY>(P, w, s, u, v);
        wind_derive_stage(P, u, v, d);
    }
}

/* FetchRelations.get_initial_windsea(U10, V10, T; particle_state=true) */
PM_HD void seed_windsea(double U10, double V10, double T, double &lne, double &cx, double &cy)
{
    const double A = 22.8013, xi0 = 2.4097, qx = 0.2748;
    double Ua = __builtin_sqrt(U10 * U10 + V10 * V10);
    Ua = (Ua < 0.1) ? 0.1 : Ua;
    T = pm_fabs(T);
    double tau = 9.81 * T / pm_fabs(Ua);
    double X = pm_pow(tau / (A * xi0), 1.0 / (1.0 - qx));
    double fm = 3.5 * (9.81 / Ua) * pm_pow(X, -0.33);
    double aj = 0.033 * pm_pow(fm * Ua / 9.81, 0.67);
    double w = fm * 2.0 * PK_PI;
    double wi = 1.0 / w;
    double E = 0.31 * (9.81 * 9.81) * aj * ((wi * wi) * (wi * wi));
    double f_peak = fm * 9.81 / Ua;
    double T_bar = 0.9 * (1.0 / f_peak);
    double cg = 9.81 * T_bar / (4.0 * PK_PI);
    lne = pm_log(E);
    cx = cg * U10 / Ua;
    cy = cg * V10 / Ua;
}

/* ResetParticleValues: windsea seed or the fixed default particle, at relative position (0,0) */
PM_HD void reseed(const KParams &P, double u, double v, double T, Vec5 &z)
{
    if (P.init_type == 0) {
        seed_windsea(u, v, T, z.lne, z.cx, z.cy);
    } else {
        z.lne = P.def_lne;
        z.cx = P.def_cx;
        z.cy = P.def_cy;
    }
    z.x = 0.0;
    z.y = 0.0;
}

/* GetParticleEnergyMomentum (kernel order: |c̄|² without the square root, one division) */
PM_HD void particle_to_charge(double lne, double cx, double cy, double &e, double &mx, double &my)
{
    e = pm_exp(lne);
    double q = (0.5 * e) / PM_FMA(cx, cx, cy * cy);
    mx = cx * q;
    my = cy * q;
}

/* GetVariablesAtVertex (kernel order: |m|² without the square root, one division) */
PM_HD void charge_to_particle(double e, double mx, double my, Vec5 &z)
{
    double q = e / (2.0 * PM_FMA(mx, mx, my * my));
    z.lne = pm_log(e);
    z.cx = mx * q;
    z.cy = my * q;
    z.x = 0.0;
    z.y = 0.0;
}

/* get_absolute_i_and_w: cell offset b = floor(z) relative to the birth node and the weight of
 * the upper node, round(z - b, digits = 6) */
PM_HD void index_weight(double zp, int &b, double &w_hi)
{
    double fb = __builtin_floor(zp);
    b = (int)fb;
    w_hi = pm_div_1e6(__builtin_rint((zp - fb) * 1e6));      /* the bits of rint(...) / 1e6 */
}

/* Rare paths read their parameters afresh from the kernarg segment: the Rosenbrock23 attempt and the guarded forms of rhs3
 * (every kernel that reaches them — k_step, k_advance — takes KParams as its FIRST argument).
 * The Rosenbrock23 attempt reads its parameters afresh from the kernarg segment (every kernel that reaches it — k_step, k_advance —
 * takes KParams as its FIRST argument).  The branch is entered by a minority of attempts, but what it alone needs — C_φ, 2/r_g²,
 * 1/e_T⁴, C_α, the unfolded constants of the Jacobian — would otherwise sit in scalar registers through all seven Tsit5 stages of
 * every attempt; the scalar file overflows there and its spill code (v_readlane / v_writelane) is VALU work inside the loop.
 * Behind an opaque copy of the segment pointer the loads stay inside the branch (scalar loads, K$ hits). */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const KParams &ros_params(const KParams &)
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    __asm__ volatile("" : "+s"(p));
    return *(const KParams *)p;
}
#else
PM_HD const KParams &ros_params(const KParams &P) { return P; }
#endif

/* RHS in kernel order: d(lne), d(c̄x), d(c̄y).  The position tendencies are c̄x/Δx, c̄y/Δy
 * (linear in the state, independent of x,y) and are folded into the stepper.
 * FAST = all five physics switches on and n == 2 (every reference script). */
struct Vec3 {
    double lne, cx, cy;
};

template <bool FAST, bool METRIC = false>
PM_HD void rhs3(const KParams &P, double lne, double cx, double cy, const WindD &W, Vec3 &d, double pc = 0.0)
{
    const double u = W.u, v = W.v;
    const double c2 = PM_FMA(cx, cx, cy * cy);
    /* y = 1/|c̄| through the deterministic reciprocal square root (no sqrt, no division); 1/c_gp = r_g y, and every power of r_g
     * rides in a constant.  The reference's guards — the speed floors 0.1 (ω_p, k_p) and 1e-4 (α_p) = ceilings ymax, sgmax on y, y²;
     * the cap α <= 500; the zero tests of the direction term — are the identity for a PLAIN particle: y <= ymax under a plain wind
     * (wind_is_plain).  Where every lane of the wave is plain — one compare and a scalar branch — none of them is evaluated; the
     * rare path evaluates them lane by lane (a NaN y, |c̄| = 0, inf or NaN, is not plain and takes the guarded forms).  The plain
     * values are computed first and overwritten on the rare path (no copies on the common one). */
    const double y = pm_rsqrt(c2);
    const double y2 = y * y;
    /* dot and cross products on the raw c̄; the cross product is two rounded products and one subtraction: exactly 0 for
     * c̄x = c̄y, u = v */
    const double dotc = PM_FMA(u, cx, v * cy);
    const double crsc = u * cy - v * cx;
    const bool plain = PM_WAVE_ALL(y <= W.ymaxw);
    double wp = P.Cw * y;                    /* ω_p = (g/2) r_g min(y, ymax) */
    double aph = (P.Chrh * dotc) * y2;       /* -α_p/2 = -(r_g/4)(c̄·u) min(y², sgmax) */
    double m2 = y2;                          /* min(y, ymax)²: k_p = (g/4) r_g² m2 */
    double m4 = y2 * y2;
    bool y_plain = true;
    if (!plain) {
        PM_RARE_PATH();
        const KParams &R = ros_params(P);    /* ymax, sgmax, ...: loaded here, not held in scalar registers through the loop */
        y_plain = (y <= R.ymax);
        const double ym = pm_fmin(y, R.ymax);
        wp = R.Cw * ym;
        aph = (R.Chrh * dotc) * pm_fmin(y2, R.sgmax);
        /* beyond ±699 the exponential below underflows in its second power either way (w² = 0: s = t = 0, H and Δ the same bits):
         * clamped (a NaN stays), so that it never needs a range test */
        aph = (aph > 699.0) ? 699.0 : aph;
        aph = (aph < -699.0) ? -699.0 : aph;
        m2 = ym * ym;
        m4 = m2 * m2;
        PM_RARE_VALUE(wp); PM_RARE_VALUE(aph); PM_RARE_VALUE(m4);
        if (!FAST) PM_RARE_VALUE(m2);
    }
    /* yh = -ya/2 = -α_p/2 + 0.425 (the halving is exact): w = exp(-|ya|/2) = exp(-|yh|).  |yh| <= 700 always: for a plain particle
     * |α_p/2| = (r_g/4)|c̄·u| y² <= (r_g/4) U ymax <= 250 whatever r_g is (that is what qU2r_max says); the others were clamped above */
    const double yh = aph + 0.425;
    /* H_β = 1/(1+eH), eH = exp(-2p ya);  Δ_β = 1 - 1.25 sech²(10 ya) = 1 - 5t/(1+t)², t = exp(-20|ya|).
     * One reciprocal serves both: r = 1/(hp (1+t)²), H = (1+t)² r, Δ = 1 - 5t hp r. */
    double hp, t, H, rHD, t12;
    if (FAST || P.p_is_075) {
        /* 2p = 3/2: both exponentials are powers of w = exp(-|ya|/2) — eH = w^(±3), t = w^40 — so ONE exponential and six
         * multiplications (1, 2, 3, 5, 10, 20, 40) serve both.  With s = w³ <= 1: for ya >= 0, eH = s and H = 1/(1+s); for ya < 0,
         * eH = 1/s and H = s/(1+s): the reciprocal is that of hp = 1 + s in [1, 2] either way, and Δ, which depends on t alone,
         * keeps its form.  For 20|ya| >= 40, 5t < 2^-54: 1 + t and Δ round to exactly 1. */
        const double w = pm_exp_negabs_inrange(yh, W.sh);
        double w2 = w * w, s3 = w2 * w;
        double w5 = s3 * w2, w10 = w5 * w5, w20 = w10 * w10;
        t = w20 * w20;
        hp = 1.0 + s3;
        double t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = pm_rcp_plain(hp * t12);            /* a plain operand in [1, 8] (or NaN) */
        H = (t12 * rHD) * ((yh <= 0.0) ? 1.0 : s3);
    } else {
        /* general p: two exponentials.  For 20|ya| >= 40 t is taken as 0 (no exp), which also makes H = 1/(1+eH) to the last
         * bit; eH is kept finite so that 0·eH stays 0 */
        const double ya = -2.0 * yh;
        hp = 1.0 + pm_exp_finite(P.neg2p * ya);
        double targ = -20.0 * pm_fabs(ya);
        t = 0.0;
        if (!(targ <= -40.0)) t = pm_exp_bounded(targ);
        double t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = pm_rcp_plain(hp * t12);            /* hp in [1, 1 + e^700], t12 in [1, 4]: a plain operand (or NaN) */
        H = t12 * rHD;
    }
    double D = PM_FMA(-((5.0 * t) * hp), rHD, 1.0);
    /* α² H = (U²/4) r_g² y² H and the direction term S_dir = C_φ α² H sin 2(θ_c - θ_w), sin 2(θ_c - θ_w) = 2 (c̄ × u)(c̄ · u) y²/U²:
     * for a plain particle the wind speed cancels, S_dir = (C_φ r_g²/2)·(c̄ × u)(c̄ · u)·y²·(y² H) — no reciprocal of U² on the path
     * (the stage winds of a time-varying window paid one per stage).  A particle that is not plain (rare path, lane by lane) takes
     * the capped α² and the guarded form with 1/U²: 0 for a vanishing wind or c̄. */
    const double y2H = y2 * H;
    double aH = W.qU2r * y2H;
    double Sd = 0.0;
    if (FAST || P.direction) Sd = ((crsc * dotc) * (y2 * y2H)) * P.Cs;
    if (!plain) {
        PM_RARE_PATH();
        const KParams &R = ros_params(P);
        const double U2 = PM_FMA(u, u, v * v);
        if (!(y_plain && wind_is_plain(R, U2, W.qU2r))) {
            aH = pm_fmin(W.qU2r * y2, 250000.0) * H;         /* α² = min(U/(2 c_gp), 500)² */
            if (FAST || P.direction)
                Sd = (U2 == 0.0 || PM_FMA(cx, cx, cy * cy) == 0.0) ? 0.0 : ((crsc * dotc) * (y2 * aH)) * (R.Cdir2 * (1.0 / U2));
        }
    }
    /* opt-in dead band: sin²(θ_c-θ_w) = crs²/(U c_gp)² below dir_deadband² counts as aligned.  Only the general-physics
     * kernels carry it (a run-time test here costs the specialised kernel 2.6 %, measured; a compile-time flavour of it
     * doubled the kernel count): a context with a dead band runs the general kernels */
    if (!FAST && P.direction && P.deadband2 > 0.0) {
        if (crsc * crsc <= P.deadband2 * (PM_FMA(u, u, v * v) * PM_FMA(cx, cx, cy * cy))) Sd = 0.0;
    }
    /* e² k_p⁴ / (K r_g⁸) = e² min(y, ymax)⁸, K = (g/4)⁴: shared by the dissipation and the peak shift, which carry K r_g⁸ in their constants */
    double Ek8 = 0.0;
    if (FAST || (P.dissipation && P.n_is_2) || P.peak_shift) Ek8 = pm_exp_sat(lne + lne, W.sh) * (m4 * m4);
    /* Ĩ - D̃ = C_e α² H - e² k_p⁴/e_T⁴ in one fused step (a switched-off term is an exact zero: C_e -> 0, D̃ -> 0) */
    double Dt = 0.0;
    if (FAST || P.dissipation) {
        if (FAST || P.n_is_2) {
            Dt = Ek8 * P.KeT4y;
        } else {
            double ke = (P.g4rg2 * m2) * P.inv_eT;
            Dt = pm_exp(P.n * lne) * pm_pow(ke, 2.0 * P.n);
        }
    }
    const double IDt = PM_FMA((FAST || P.input) ? P.C_e : 0.0, aH, -Dt);
    /* ω_p r_g S_cg = ω_p r_g C_α Δ e² k_p⁴ */
    double wrS = 0.0;
    if (FAST || P.peak_shift) wrS = (wp * D) * (Ek8 * P.KrCay);
    if (METRIC) Sd = Sd + cx * pc;   /* great-circle term S_sphere = PC(c̄x) = c̄x·coef rides on S_dir */
    d.lne = PM_FMA(wp, IDt, wrS);
    d.cx = PM_FMA(cy, Sd, -(cx * wrS));
    d.cy = -PM_FMA(cx, Sd, cy * wrS);
}

/* ------------------------------------------------------------------------------------------
 * solver 2 = AutoTsit5(Rosenbrock23()), the reference's default (particle_waves_v5.jl:47); semantics restated
 * from OrdinaryDiffEq.jl (unpinned; the full statement of the semantics is in DESIGN.md §2 and above integrate_dp5).
 * rhs3_jvp: directional derivatives of the kernel-order RHS along NS seed directions
 * (d lne, d c̄x, d c̄y, du, dv) — the exact Jacobian (and ∂f/∂t through the node wind) written out by hand.
 * ---------------------------------------------------------------------------------------- */
#define ROS_D 0.29289321881345254   /* 1/(2+sqrt 2) */
#define ROS_E32 7.414213562373095   /* 6+sqrt 2 */
#define ASW_STABILITY 3.5068        /* alg_stability_size(Tsit5()) */
#define ASW_FRESH (-2147483647 - 1)

struct Seed5 {
    double dL, dcx, dcy, du, dv;
};

template <bool FAST, bool METRIC, int NS>
PM_HD void rhs3_jvp(const KParams &P, double lne, double cx, double cy, const WindD &W, double pc,
                    const Seed5 (&seeds)[NS], Vec3 (&df)[NS])
{
    const double u = W.u, v = W.v;
    double c2 = PM_FMA(cx, cx, cy * cy);
    double y = pm_rsqrt(c2);
    double rc = P.r_g * y;
    double ic2 = y * y;
    double minv = pm_fmin(rc, 10.0);
    double wp = (0.5 * PK_G0) * minv;
    double kp = (0.25 * PK_G0) * (minv * minv);
    double rc2 = rc * rc;
    const double W_U2 = PM_FMA(u, u, v * v);
    const double W_qU2 = 0.25 * W_U2;
    double W_invU2;                                  /* 1/U²: the bits of the division (plain-range reciprocal behind a wave-uniform range test) */
    if (PM_WAVE_ALL(W_U2 >= 1e-290 && W_U2 <= 1e290)) {
        W_invU2 = pm_rcp_plain(W_U2);
    } else {
        PM_RARE_PATH();
        W_invU2 = 1.0 / W_U2;
    }
    double a2 = W_qU2 * rc2;
    double alpha2 = pm_fmin(a2, 250000.0);
    double dotc = PM_FMA(u, cx, v * cy);
    double crsc = u * cy - v * cx;
    double sginv2 = pm_fmin(rc2, 1e8);
    double ap = (P.half_inv_rg * dotc) * sginv2;
    double ya = ap - 0.85;
    /* H_β, Δ_β and the slope gH = dH/dya = 2p H (1 - H) (no e^(-2p ya) needed: H² e^(-2p ya) = H (1 - H)); as in rhs3 one
     * exponential serves both when 2p = 3/2 */
    double hp, t, t1, t12, rHD, H, gH;
    if (FAST || P.p_is_075) {
        double w = pm_exp_sh(-0.5 * pm_fabs(ya), W.sh);
        double w2 = w * w, s3 = w2 * w;
        double w4 = w2 * w2, w5 = w4 * w, w10 = w5 * w5, w20 = w10 * w10;
        t = w20 * w20;
        hp = 1.0 + s3;
        t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = pm_rcp_plain(hp * t12);
        H = (t12 * rHD) * ((ya >= 0.0) ? 1.0 : s3);
        gH = (-P.neg2p) * (H * (1.0 - H));
    } else {
        double harg = P.neg2p * ya;
        double eH = pm_exp((harg > 700.0) ? 700.0 : harg);
        hp = 1.0 + eH;
        double targ = -20.0 * pm_fabs(ya);
        t = (targ <= -40.0) ? 0.0 : pm_exp(targ);
        t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = pm_rcp_plain(hp * t12);
        H = t12 * rHD;
        gH = (harg > 700.0) ? 0.0 : -((H * H) * (eH * P.neg2p));
    }
    double D = PM_FMA(-((5.0 * t) * hp), rHD, 1.0);
    const double r13 = pm_rcp_plain(t12 * t1);       /* (1 + t)³ in [1, 8] */
    const double rminv = pm_rcp_plain(minv);         /* d k_p / k_p = 2 d(1/c_gp)/(1/c_gp): 1/c_gp in (0, 10] */
    double aH = alpha2 * H;
    const bool n2 = FAST || P.n_is_2;
    const bool s_in = FAST || P.input, s_di = FAST || P.dissipation, s_ps = FAST || P.peak_shift, s_dr = FAST || P.direction;
    double E2 = pm_exp_sh(2.0 * lne, W.sh);
    double k2 = kp * kp, k4 = k2 * k2;
    double Ek = E2 * k4;
    double It = s_in ? P.C_e * aH : 0.0;
    double Dt = (s_di && n2) ? Ek * P.inv_eT4 : 0.0;
    if (s_di && !n2) Dt = pm_exp(P.n * lne) * pm_pow(kp * P.inv_eT, 2.0 * P.n);
    double Scg = s_ps ? (P.C_alpha * D) * Ek : 0.0;
    bool calm = (W_U2 == 0.0 || c2 == 0.0);
    double rU = rc2 * W_invU2;
    double cd = (P.two_inv_rg2 * crsc) * dotc;
    double s2 = calm ? 0.0 : cd * rU;
    bool dead = !FAST && (P.deadband2 > 0.0 && crsc * crsc <= P.deadband2 * (W_U2 * c2));
    if (dead) s2 = 0.0;
    double Sd = s_dr ? (P.C_phi * aH) * s2 : 0.0;
    double wrS = (wp * P.r_g) * Scg;
    double Sdm = Sd + (METRIC ? cx * pc : 0.0);
#pragma unroll
    for (int q = 0; q < NS; q++) {
        const double dL = seeds[q].dL, dcx = seeds[q].dcx, dcy = seeds[q].dcy, du = seeds[q].du, dv = seeds[q].dv;
        double dc2 = 2.0 * PM_FMA(cx, dcx, cy * dcy);
        double drc = -0.5 * ((rc * ic2) * dc2);
        double dminv = (rc <= 10.0) ? drc : 0.0;
        double dwp = (0.5 * PK_G0) * dminv;
        double drc2 = 2.0 * (rc * drc);
        double dU2 = 2.0 * PM_FMA(u, du, v * dv);
        double dqU2 = 0.25 * dU2;
        double dinvU2 = -((W_invU2 * W_invU2) * dU2);
        double dalpha2 = (a2 <= 250000.0) ? PM_FMA(W_qU2, drc2, rc2 * dqU2) : 0.0;
        double ddot = PM_FMA(u, dcx, v * dcy) + PM_FMA(cx, du, cy * dv);
        double dcrs = (u * dcy - v * dcx) + (cy * du - cx * dv);
        double dsg = (rc2 <= 1e8) ? drc2 : 0.0;
        double dya = P.half_inv_rg * PM_FMA(ddot, sginv2, dotc * dsg);
        double sgn = (ya < 0.0) ? 20.0 : -20.0;
        double dt_ = (t * sgn) * dya;
        double dH = gH * dya;
        double dD = -5.0 * ((dt_ * (1.0 - t)) * r13);
        double daH = PM_FMA(dalpha2, H, alpha2 * dH);
        double dlk = dminv * rminv;                     /* d k_p / k_p = 2 dlk */
        double dEk = ((s_di && n2) || s_ps) ? Ek * PM_FMA(2.0, dL, 8.0 * dlk) : 0.0;
        double dIt = s_in ? P.C_e * daH : 0.0;
        double dDt = (s_di && n2) ? dEk * P.inv_eT4 : 0.0;
        if (s_di && !n2) dDt = Dt * PM_FMA(P.n, dL, (4.0 * P.n) * dlk);
        double dScg = s_ps ? P.C_alpha * PM_FMA(dD, Ek, D * dEk) : 0.0;
        double ds2 = 0.0;
        if (!calm && !dead) {
            double dcd = P.two_inv_rg2 * PM_FMA(dcrs, dotc, crsc * ddot);
            double drU = PM_FMA(drc2, W_invU2, rc2 * dinvU2);
            ds2 = PM_FMA(dcd, rU, cd * drU);
        }
        double dSd = s_dr ? P.C_phi * PM_FMA(daH, s2, aH * ds2) : 0.0;
        if (METRIC) dSd = dSd + dcx * pc;
        double dwrS = P.r_g * PM_FMA(dwp, Scg, wp * dScg);
        df[q].lne = PM_FMA(dwp, It - Dt, wp * (dIt - dDt)) + dwrS;
        df[q].cx = PM_FMA(dcy, Sdm, cy * dSd) - PM_FMA(dcx, wrS, cx * dwrS);
        df[q].cy = -(PM_FMA(dcx, Sdm, cx * dSd) + PM_FMA(dcy, wrS, cy * dwrS));
    }
}

/* The Jacobian of the specialised-physics RHS for a PLAIN particle (rhs3: none of the reference's guards acts), written out along
 * the structure of the RHS instead of pushed through it direction by direction.  (c̄x, c̄y) enter through three scalars only —
 * c² = |c̄|², the dot and the cross product with the wind — besides their explicit places in the momentum equations:
 *     F  = d ln e/dt  = ω_p (Ĩ - D̃) + ω_p r_g S_cg      F(L, c², c̄·u)            L = ln e
 *     S  = ω_p r_g S_cg = Q Δ_β                           S(L, c², c̄·u)            Q = ω_p r_g C_α e² k_p⁴
 *     Sd = S_dir = (C_φ r_g²/2)(c̄×u)(c̄·u) y⁴ H_β         Sd(c², c̄·u, c̄×u)        y = 1/|c̄|
 *     f  = (F,  c̄y Sd - c̄x S,  -(c̄x Sd + c̄y S))
 * so nine partials — of (F, S, Sd) with respect to (L, c², c̄·u, c̄×u), mostly logarithmic derivatives (y^k: -k/2 y² per unit c²;
 * e²: 2 per unit L) — and the chain rule ∂/∂c̄x = 2 c̄x ∂/∂c² + u ∂/∂(c̄·u) - v ∂/∂(c̄×u), ∂/∂c̄y = 2 c̄y ∂/∂c² + v ∂/∂(c̄·u) + u ∂/∂(c̄×u)
 * give all nine entries: about 180 issue slots where three (with ∂f/∂t: four) passes of the forward-mode rhs3_jvp took 370.  H_β and
 * Δ_β depend on ya = α_p - 0.85 = -2 yh alone: dH/dya = 2p H (1 - H), dΔ/dya = 100 sgn(ya) t (1 - t)/(1 + t)³, and
 * 1/(1 + t)³ = (1 + t)(hp r)² reuses the reciprocal r = 1/(hp (1 + t)²) the primal has (no second one).  TV: ∂f/∂t through the node
 * wind's slope (du, dv) — for a plain particle U² enters F only (α² ∝ U²; it cancels out of S_dir).
 * A particle that is not plain takes rhs3_jvp (rare path of ros23_try, lane by lane).  J row-major: J[3 r + c] = ∂f_r/∂u_c. */
template <bool METRIC, bool TV>
PM_HD double rhs3_jac_plain(const KParams &P, double L, double cx, double cy, const WindD &W, double pc, double du, double dv,
                            double (&J)[9], Vec3 &dT)      /* returns y = 1/|c̄| (the caller's plain test) */
{
    const double u = W.u, v = W.v;
    const double c2 = PM_FMA(cx, cx, cy * cy);
    const double y = pm_rsqrt(c2);
    const double y2 = y * y;                          /* 1/c²: the unit of the logarithmic derivatives with respect to c² */
    const double dotc = PM_FMA(u, cx, v * cy);
    const double crsc = u * cy - v * cx;
    const double wp = P.Cw * y;
    const double aph = (P.Chrh * dotc) * y2;
    const double m4 = y2 * y2;
    const double yh = aph + 0.425;
    const double w = pm_exp_negabs_inrange(yh, W.sh);
    const double w2 = w * w, s3 = w2 * w;
    const double w5 = s3 * w2, w10 = w5 * w5, w20 = w10 * w10;
    const double t = w20 * w20;
    const double hp = 1.0 + s3, t1 = 1.0 + t;
    const double t12 = t1 * t1;
    const double rHD = pm_rcp_plain(hp * t12);
    const double H = (t12 * rHD) * ((yh <= 0.0) ? 1.0 : s3);
    const double D = PM_FMA(-((5.0 * t) * hp), rHD, 1.0);
    /* derivatives with respect to yh = -ya/2:  dH/dyh = -2·(3/2) H (1 - H),  dΔ/dyh = 200 sgn(yh) t (1 - t)/(1 + t)³ */
    const double DH = -3.0 * (H * (1.0 - H));
    const double iq = hp * rHD;                       /* 1/(1 + t)² */
    const double DD = __builtin_copysign(200.0 * ((t * (1.0 - t)) * (t1 * (iq * iq))), yh);
    const double yh_dot = P.Chrh * y2, yh_c2 = -(aph * y2);
    const double H_dot = DH * yh_dot, H_c2 = DH * yh_c2, D_dot = DD * yh_dot, D_c2 = DD * yh_c2;
    const double alpha2 = W.qU2r * y2;
    const double aH = alpha2 * H;
    const double aH_dot = alpha2 * H_dot;
    const double aH_c2 = alpha2 * PM_FMA(-y2, H, H_c2);             /* α² ∝ y²: -y² per unit c² */
    const double Ek8 = pm_exp_sat(L + L, W.sh) * (m4 * m4);
    const double Dt = Ek8 * P.KeT4y;
    const double IDt = PM_FMA(P.C_e, aH, -Dt);
    const double Q = wp * (Ek8 * P.KrCay);
    const double S = Q * D;
    const double wpDt = wp * Dt, wpCe = wp * P.C_e;
    /* F */
    const double F_L = 2.0 * (S - wpDt);
    const double S_dot = Q * D_dot;
    const double F_dot = PM_FMA(wpCe, aH_dot, S_dot);
    const double S_c2 = PM_FMA(Q, D_c2, -4.5 * (y2 * S));            /* Q ∝ y⁹ */
    /* ω_p ∝ y: -y²/2;  D̃ ∝ y⁸: -4 y² */
    const double F_c2 = PM_FMA(-0.5 * y2, wp * IDt, PM_FMA(4.0 * y2, wpDt, PM_FMA(wpCe, aH_c2, S_c2)));
    /* S_dir = (c̄×u)(c̄·u) Ba H,  Ba = (C_φ r_g²/2) y⁴ */
    const double Ba = P.Cs * m4;
    const double B = Ba * H;
    const double Pcd = crsc * dotc;
    const double Sd = Pcd * B;
    const double PBa = Pcd * Ba;
    const double Sd_crs = dotc * B;
    const double Sd_dot = PM_FMA(PBa, H_dot, crsc * B);
    const double Sd_c2 = PBa * PM_FMA(-2.0 * y2, H, H_c2);
    /* chain rule */
    const double ax = cx + cx, ay = cy + cy;
    const double F_x = PM_FMA(ax, F_c2, u * F_dot), F_y = PM_FMA(ay, F_c2, v * F_dot);
    const double S_x = PM_FMA(ax, S_c2, u * S_dot), S_y = PM_FMA(ay, S_c2, v * S_dot);
    double Sd_x = PM_FMA(ax, Sd_c2, PM_FMA(u, Sd_dot, -(v * Sd_crs)));
    const double Sd_y = PM_FMA(ay, Sd_c2, PM_FMA(v, Sd_dot, u * Sd_crs));
    double Sdm = Sd;
    if (METRIC) { Sdm = Sd + cx * pc; Sd_x = Sd_x + pc; }           /* great-circle term c̄x·coef rides on S_dir */
    const double S_L = S + S;
    J[0] = F_L; J[1] = F_x; J[2] = F_y;
    J[3] = -(cx * S_L);
    J[4] = PM_FMA(cy, Sd_x, -PM_FMA(cx, S_x, S));
    J[5] = PM_FMA(cy, Sd_y, Sdm) - cx * S_y;
    J[6] = -(cy * S_L);
    J[7] = -(PM_FMA(cx, Sd_x, Sdm) + cy * S_x);
    J[8] = -PM_FMA(cx, Sd_y, PM_FMA(cy, S_y, S));
    if (TV) {
        const double dot_t = PM_FMA(cx, du, cy * dv), crs_t = PM_FMA(cy, du, -(cx * dv));
        const double U2_t = 2.0 * PM_FMA(u, du, v * dv);
        const double F_U2 = wpCe * ((0.25 * P.rg2) * (y2 * H));      /* α² H per unit U² */
        const double F_t = PM_FMA(F_dot, dot_t, F_U2 * U2_t);
        const double S_t = S_dot * dot_t;
        const double Sd_t = PM_FMA(Sd_dot, dot_t, Sd_crs * crs_t);
        dT.lne = F_t;
        dT.cx = PM_FMA(cy, Sd_t, -(cx * S_t));
        dT.cy = -PM_FMA(cx, Sd_t, cy * S_t);
    }
    return y;
}

/* one attempted Rosenbrock23 step of size h from (z, f0) at absolute time t: returns EEst² (kernel-order norm),
 * un = the new state, f2 = f(un, t+h) (the next step's FSAL), eig = ||J||_inf */
template <bool FAST, bool STATIC, bool METRIC>
PM_HD double ros23_try(const KParams &P, const Wind &w, WindD &W, const Vec5 &z, const Vec3 &f0, double t, double h,
                       double ipx, double ipy, double pc, Vec5 &un, Vec3 &f2, double &eig, PStats &st)
{
    constexpr bool POLY = !FAST && !STATIC;      /* the general time-varying flavours carry polyline windows (wind_eval) */
    const bool tv = !STATIC && !P.wind_static;   /* a time-varying instantiation may run with static winds: no dT terms then */
    double dudt = 0.0, dvdt = 0.0;
    if (!STATIC) {      /* du/dt, dv/dt of the window's interpolant at t: parabola (du + (2 s - 1) bu) / (tw1 - tw0); knot form: the segment's slope */
        const double s_ = (t - P.tw0) * P.inv_dtw;
        if (POLY && P.wind_nk > 1) {
            double su_, sv_;
            wind_slopes_poly(P, w, s_, su_, sv_);
            dudt = su_ * P.inv_dtw;
            dvdt = sv_ * P.inv_dtw;
        } else {
            dudt = wind_slope(P, w.du, w.bu, s_) * P.inv_dtw;
            dvdt = wind_slope(P, w.dv, w.bv, s_) * P.inv_dtw;
        }
    }
    wind_stage<STATIC, POLY>(P, w, t, W);
    double Jm[9];
    Vec3 dT = {0.0, 0.0, 0.0};
    /* the specialised physics takes the structured Jacobian (rhs3_jac_plain); a lane whose particle is not plain — and every lane of
     * the general-physics kernels — the forward-mode rhs3_jvp along the unit directions */
    const bool structured = FAST || P.fast_phys;      /* by the PHYSICS, not by the flavour (KParams::fast_phys) */
    bool fwd = !structured;
    double yj = 0.0;
    if (structured) {
        yj = rhs3_jac_plain<METRIC, !STATIC>(P, z.lne, z.cx, z.cy, W, pc, dudt, dvdt, Jm, dT);
        fwd = !PM_WAVE_ALL(yj <= W.ymaxw);
    }
    if (fwd) {
        if (FAST) PM_RARE_PATH();
        constexpr int NS = STATIC ? 3 : 4;
        Seed5 seeds[NS];
        Vec3 dfs[NS];
        seeds[0] = {1.0, 0.0, 0.0, 0.0, 0.0};
        seeds[1] = {0.0, 1.0, 0.0, 0.0, 0.0};
        seeds[2] = {0.0, 0.0, 1.0, 0.0, 0.0};
        if (!STATIC) seeds[NS - 1] = {0.0, 0.0, 0.0, dudt, dvdt};
        rhs3_jvp<FAST, METRIC, NS>(P, z.lne, z.cx, z.cy, W, pc, seeds, dfs);
        bool mine = true;         /* this lane takes the forward-mode result */
        if (structured) mine = !(wind_is_plain(P, PM_FMA(W.u, W.u, W.v * W.v), W.qU2r) && yj <= P.ymax);
        if (mine) {
            /* J[r][c] = d f_r / d u_c = dfs[c].r */
            Jm[0] = dfs[0].lne; Jm[1] = dfs[1].lne; Jm[2] = dfs[2].lne;
            Jm[3] = dfs[0].cx; Jm[4] = dfs[1].cx; Jm[5] = dfs[2].cx;
            Jm[6] = dfs[0].cy; Jm[7] = dfs[1].cy; Jm[8] = dfs[2].cy;
            if (!STATIC) dT = dfs[NS - 1];
        }
    }
    if (!tv) dT = {0.0, 0.0, 0.0};
    st.rhs += tv ? 4 : 3;
    const double J00 = Jm[0], J01 = Jm[1], J02 = Jm[2];
    const double J10 = Jm[3], J11 = Jm[4], J12 = Jm[5];
    const double J20 = Jm[6], J21 = Jm[7], J22 = Jm[8];
    {
        double m = 0.0;
        m = pm_fmax(m, (pm_fabs(J00) + pm_fabs(J01)) + pm_fabs(J02));
        m = pm_fmax(m, (pm_fabs(J10) + pm_fabs(J11)) + pm_fabs(J12));
        m = pm_fmax(m, (pm_fabs(J20) + pm_fabs(J21)) + pm_fabs(J22));
        m = pm_fmax(m, pm_fabs(ipx));
        m = pm_fmax(m, pm_fabs(ipy));
        eig = m;
    }
    const double g = h * ROS_D;
    const double W00 = 1.0 - g * J00, W01 = 0.0 - g * J01, W02 = 0.0 - g * J02;
    const double W10 = 0.0 - g * J10, W11 = 1.0 - g * J11, W12 = 0.0 - g * J12;
    const double W20 = 0.0 - g * J20, W21 = 0.0 - g * J21, W22 = 1.0 - g * J22;
    const double A00 = W11 * W22 - W12 * W21, A01 = W02 * W21 - W01 * W22, A02 = W01 * W12 - W02 * W11;
    const double A10 = W12 * W20 - W10 * W22, A11 = W00 * W22 - W02 * W20, A12 = W02 * W10 - W00 * W12;
    const double A20 = W10 * W21 - W11 * W20, A21 = W01 * W20 - W00 * W21, A22 = W00 * W11 - W01 * W10;
    const double det = PM_FMA(W00, A00, PM_FMA(W01, A10, W02 * A20));
    double idet;                    /* 1/det: the bits of the division (plain-range reciprocal behind a wave-uniform range test) */
    if (PM_WAVE_ALL(pm_fabs(det) >= 1e-290 && pm_fabs(det) <= 1e290)) {
        idet = pm_rcp_plain(det);
    } else {
        PM_RARE_PATH();
        idet = 1.0 / det;
    }
    const double gx = g * ipx, gy = g * ipy;
#define WSOLVE(b0, b1, b2, b3, b4, o)                                   \
    do {                                                                \
        double q0 = PM_FMA(A00, (b0), PM_FMA(A01, (b1), A02 * (b2))) * idet; \
        double q1 = PM_FMA(A10, (b0), PM_FMA(A11, (b1), A12 * (b2))) * idet; \
        double q2 = PM_FMA(A20, (b0), PM_FMA(A21, (b1), A22 * (b2))) * idet; \
        o.lne = q0; o.cx = q1; o.cy = q2;                               \
        o.x = PM_FMA(gx, q1, (b3)); o.y = PM_FMA(gy, q2, (b4));          \
    } while (0)
    const double f0x = z.cx * ipx, f0y = z.cy * ipy;
    Vec5 k1, k2, k3;
    if (!tv) WSOLVE(f0.lne, f0.cx, f0.cy, f0x, f0y, k1);
    else WSOLVE(PM_FMA(g, dT.lne, f0.lne), PM_FMA(g, dT.cx, f0.cx), PM_FMA(g, dT.cy, f0.cy), f0x, f0y, k1);
    const double h2 = 0.5 * h;
    const double sl = PM_FMA(h2, k1.lne, z.lne), sx = PM_FMA(h2, k1.cx, z.cx), sy = PM_FMA(h2, k1.cy, z.cy);
    Vec3 f1;
    wind_stage<STATIC, POLY>(P, w, t + h2, W);
    rhs3<FAST, METRIC>(P, sl, sx, sy, W, f1, pc);
    const double f1x = sx * ipx, f1y = sy * ipy;
    WSOLVE(f1.lne - k1.lne, f1.cx - k1.cx, f1.cy - k1.cy, f1x - k1.x, f1y - k1.y, k2);
    k2.lne = k2.lne + k1.lne; k2.cx = k2.cx + k1.cx; k2.cy = k2.cy + k1.cy; k2.x = k2.x + k1.x; k2.y = k2.y + k1.y;
    un.lne = PM_FMA(h, k2.lne, z.lne); un.cx = PM_FMA(h, k2.cx, z.cx); un.cy = PM_FMA(h, k2.cy, z.cy);
    un.x = PM_FMA(h, k2.x, z.x); un.y = PM_FMA(h, k2.y, z.y);
    wind_stage<STATIC, POLY>(P, w, t + h, W);
    rhs3<FAST, METRIC>(P, un.lne, un.cx, un.cy, W, f2, pc);
    st.rhs += 2;
    const double f2x = un.cx * ipx, f2y = un.cy * ipy;
#define ROSB(F2, K2, F1, K1, F0) (((F2) - ROS_E32 * ((K2) - (F1))) - 2.0 * ((K1) - (F0)))
    double b0 = ROSB(f2.lne, k2.lne, f1.lne, k1.lne, f0.lne), b1 = ROSB(f2.cx, k2.cx, f1.cx, k1.cx, f0.cx);
    double b2 = ROSB(f2.cy, k2.cy, f1.cy, k1.cy, f0.cy);
    double b3 = ROSB(f2x, k2.x, f1x, k1.x, f0x), b4 = ROSB(f2y, k2.y, f1y, k1.y, f0y);
#undef ROSB
    if (tv) { b0 = b0 + h * dT.lne; b1 = b1 + h * dT.cx; b2 = b2 + h * dT.cy; }
    WSOLVE(b0, b1, b2, b3, b4, k3);
#undef WSOLVE
    const double h6 = h * (1.0 / 6.0);
#define ROSE(c) (h6 * ((k1.c - 2.0 * k2.c) + k3.c))
#define ERRS(a, b) PM_FMA(pm_fmax_abs(a, b), P.reltol, P.abstol)
    double s0 = ERRS(z.lne, un.lne), s1 = ERRS(z.cx, un.cx), s2 = ERRS(z.cy, un.cy);
    double s3 = ERRS(z.x, un.x), s4 = ERRS(z.y, un.y);
    double p2 = s0 * s1, p3 = p2 * s2, p4 = p3 * s3, pp = p4 * s4;
    double q2 = s3 * s4, q1 = s2 * q2, q0 = s1 * q1;
    double n0 = ROSE(lne) * q0, n1 = (ROSE(cx) * s0) * q1, n2 = (ROSE(cy) * p2) * q2;
    double n3 = (ROSE(x) * p3) * s4, n4 = ROSE(y) * p4;
#undef ROSE
#undef ERRS
    double S = n0 * n0;
    S = PM_FMA(n1, n1, S);
    S = PM_FMA(n2, n2, S);
    S = PM_FMA(n3, n3, S);
    S = PM_FMA(n4, n4, S);
    double rp = pm_rcp_plain(pp);     /* see the explicit pair's error norm */
    return (S * 0.2) * (rp * rp);
}

/* mean square of five numbers (the squared RMS norm) */
PM_HD double ms5(double a0, double a1, double a2, double a3, double a4)
{
    double s = a0 * a0;
    s = PM_FMA(a1, a1, s);
    s = PM_FMA(a2, a2, s);
    s = PM_FMA(a3, a3, s);
    s = PM_FMA(a4, a4, s);
    return s * 0.2;
}

PM_HD double rms5(double a0, double a1, double a2, double a3, double a4)
{
    double s = a0 * a0;
    s = PM_FMA(a1, a1, s);
    s = PM_FMA(a2, a2, s);
    s = PM_FMA(a3, a3, s);
    s = PM_FMA(a4, a4, s);
    return __builtin_sqrt(s * 0.2);
}

/* ode_determine_initdt (Hairer–Wanner), = auto_dt_reset! after every remesh.
 * f0 = (k1, kx, ky) is the RHS at (u0, t). */
template <bool FAST, bool STATIC, bool METRIC>
PM_HD double init_dt(const KParams &P, const Wind &w, WindD &W, const Vec5 &u0, const Vec3 &k1, double kx, double ky,
                     double ipx, double ipy, double pc, double t, PStats &st)
{
    constexpr bool POLY = !FAST && !STATIC;
    /* kernel order: 1/sk for (lne, c̄x, c̄y) from ONE reciprocal of the product of the three scales;
     * the RMS norms stay squared (S = d²): dt0 = 0.01 d0/d1 = 0.01 S0 / sqrt(S0 S1) through the
     * deterministic rsqrt, and the second-derivative estimate works on max(d1², d2²) with the coarse
     * logarithm (this is a first guess of the step, refined by the controller). */
    double s0 = PM_FMA(pm_fabs(u0.lne), P.reltol, P.abstol);
    double s1 = PM_FMA(pm_fabs(u0.cx), P.reltol, P.abstol);
    double s2 = PM_FMA(pm_fabs(u0.cy), P.reltol, P.abstol);
    double p01 = s0 * s1;
    double rp = 1.0 / (p01 * s2);
    double r0 = (s1 * s2) * rp, r1 = (s0 * s2) * rp, r2 = p01 * rp;
    /* a freshly remeshed particle sits on its node: x = y = 0, scale = abstol (same bits, no division) */
    double r3 = P.inv_abstol, r4 = P.inv_abstol;
    if (u0.x != 0.0 || u0.y != 0.0) {
        r3 = 1.0 / PM_FMA(pm_fabs(u0.x), P.reltol, P.abstol);
        r4 = 1.0 / PM_FMA(pm_fabs(u0.y), P.reltol, P.abstol);
    }
    double S0 = ms5(u0.lne * r0, u0.cx * r1, u0.cy * r2, u0.x * r3, u0.y * r4);
    double S1 = ms5(k1.lne * r0, k1.cx * r1, k1.cy * r2, kx * r3, ky * r4);
    double dt0;
    if (S0 < 1e-10 || S1 < 1e-10) dt0 = 1e-6;
    else dt0 = (0.01 * S0) * pm_rsqrt(S0 * S1);
    if (dt0 < 10.0 * 2.220446049250313e-16) return 1e-6;
    double l1 = PM_FMA(dt0, k1.lne, u0.lne), cx1 = PM_FMA(dt0, k1.cx, u0.cx), cy1 = PM_FMA(dt0, k1.cy, u0.cy);
    Vec3 f1;
    wind_stage<STATIC, POLY>(P, w, t + dt0, W);
    rhs3<FAST, METRIC>(P, l1, cx1, cy1, W, f1, pc);
    st.rhs++;
    double f1x = cx1 * ipx, f1y = cy1 * ipy;
    double S2 = ms5((f1.lne - k1.lne) * r0, (f1.cx - k1.cx) * r1, (f1.cy - k1.cy) * r2,
                    (f1x - kx) * r3, (f1y - ky) * r4) / (dt0 * dt0);
    double m2 = (S1 > S2) ? S1 : S2;
    double dt1;
    if (!(m2 > 1e-30)) {   /* flat (or non-finite) second-derivative estimate */
        double c = dt0 * 1e-3;
        dt1 = (1e-6 > c) ? 1e-6 : c;
    } else {
        /* 10^(-(2 + log10 m)/5) = exp(-0.2 ln 100 - 0.1 ln m²) */
        dt1 = pm_exp(PM_FMA(-0.1, pm_log_coarse(m2), -0.92103403719761827));
    }
    double h = 100.0 * dt0;
    if (dt1 < h) h = dt1;
    if (!(h == h)) h = 1e-6;
    return (P.dtmin > h) ? P.dtmin : h;
}

/* step!(integrator, DT, true): integrate z over [t_start, t_start+DT] with DP5(4).
 * Only the stage derivatives of (lne, c̄x, c̄y) are kept; the x,y rows of the tableau are
 * accumulated as the stages appear (same fma order as the full Butcher sums). */
/* AUTO (solver 2, implies TSIT): AutoTsit5(Rosenbrock23()) — after every attempt the AutoSwitch tests
 * |eigen_est·dt_next/3.5068| > 0.9; more than 10 successive positives hand over to Rosenbrock23 (dt·2), more than 3
 * successive negatives hand back (dt/2).  *asw carries (counter << 1 | rosenbrock_active) across model steps. */
template <bool FAST, bool STATIC, bool METRIC = false, bool TSIT = false, bool AUTO = false, bool TABS = true>
PM_HD void integrate_dp5(const KParams &P, const Wind &w, Vec5 &z, double &lq, double &dtn,
                         double t_start, double DT, PStats &st, double m11 = 0.0, double m22 = 0.0, double pc = 0.0,
                         int *asw = nullptr)
{
    /* projection M = diag(ipx, ipy): 1/Δx, 1/Δy on the Cartesian mesh, per node otherwise */
    const double ipx = (FAST || P.propagation) ? (METRIC ? m11 : P.inv_dx) : 0.0;
    const double ipy = (FAST || P.propagation) ? (METRIC ? m22 : P.inv_dy) : 0.0;
    constexpr bool POLY = !FAST && !STATIC;      /* the general time-varying flavours carry polyline windows (wind_eval) */
    Vec3 k1, k2, k3, k4, k5, k6, k7;
    WindD W;
    W.sh = PM_EXP_SHIFTER();
    DP_TAB_DECL(TSIT ? 1 : 0);
    constexpr bool has2 = TSIT;   /* Tsit5: a72, e2 != 0 (compile-time: the DP5 instruction stream is untouched) */
    constexpr double beta1 = TSIT ? PI_BETA1_TSIT : PI_BETA1, beta2 = TSIT ? PI_BETA2_TSIT : PI_BETA2;
    double tr = 0.0;
    if (STATIC) wind_derive(P, w.u0, w.v0, W);
    else wind_stage<false, POLY>(P, w, t_start, W);
    rhs3<FAST, METRIC>(P, z.lne, z.cx, z.cy, W, k1, pc);
    st.rhs++;
    double dt = dtn;
    if (!(dt > 0.0)) dt = init_dt<FAST, STATIC, METRIC>(P, w, W, z, k1, z.cx * ipx, z.cy * ipy, ipx, ipy, pc, t_start, st);
    long long iter = 0;
    bool as_fresh = false, as_stiff = false, have_eig = false;
    int as_count = 0;
    double eig = 0.0, eig_nu = 0.0, eig_nd = 1.0;   /* eigen_est² = eig_nu / eig_nd */
    if (AUTO) {
        as_fresh = (*asw == ASW_FRESH);
        as_stiff = as_fresh ? false : ((*asw & 1) != 0);
        as_count = as_fresh ? 0 : (*asw >> 1);
    }
    while (tr < DT) {
        iter++;
        if (iter > P.maxiters) { st.status |= 2 /*PICLES_ST_MAXITERS*/; break; }
        if (AUTO) {   /* choose_algorithm!: the estimate of the previous attempt against the proposed dt */
            if (as_fresh) {
                as_fresh = false;
            } else if (have_eig) {
                /* |eigen_est dt / 3.5068| > 0.9  <=>  nu dt² > (0.9·3.5068)² nd  (a NaN on either side: not stiff) */
                bool pos = eig_nu * (dt * dt) > ((0.9 * ASW_STABILITY) * (0.9 * ASW_STABILITY)) * eig_nd;
                as_count = pos ? (as_count < 0 ? 1 : as_count + 1) : (as_count > 0 ? -1 : as_count - 1);
                if (!as_stiff && as_count > 10) { dt = dt * 2.0; as_stiff = true; }
                else if (as_stiff && as_count < -3) { dt = dt * 0.5; as_stiff = false; }
            }
        }
        if (dt < P.dtmin) dt = P.dtmin;
        double rem = DT - tr;
        bool last = !(dt < rem);
        double h = last ? rem : dt;
        double t = t_start + tr;
        Vec5 un;
        double EE2;
        if (AUTO && as_stiff) {
            EE2 = ros23_try<FAST, STATIC, METRIC>(ros_params(P), w, W, z, k1, t, h, ipx, ipy, pc, un, k7, eig, st);
            eig_nu = eig * eig; eig_nd = 1.0;     /* ||J||_inf */
        } else {
        double gl, gx, gy;      /* stage state (lne, c̄x, c̄y) */
        /* x,y tendencies are c̄x/Δx, c̄y/Δy of the stage state: their tableau sums run on the stage
         * c̄ itself (Σ a7i c̄_i, Σ e_i c̄_i) and meet the projection 1/Δx, 1/Δy once, at the end */
        double ax, ay, ex, ey;
        TT_STAGE();
        ax = TT(a71) * z.cx; ay = TT(a71) * z.cy;
        ex = TT(e1) * z.cx; ey = TT(e1) * z.cy;
        double s6x = 0.0, s6y = 0.0;   /* AUTO: Σ a6j c̄_j, the x,y position of stage 6 (stiffness estimate) */
        if (AUTO) { s6x = TT(a61) * z.cx; s6y = TT(a61) * z.cy; }
#define S6ACC(a6) do { if (AUTO) { s6x = PM_FMA(TT(a6), gx, s6x); s6y = PM_FMA(TT(a6), gy, s6y); } } while (0)
        {
            double a21h = h * TT(a21);
            gl = PM_FMA(a21h, k1.lne, z.lne); gx = PM_FMA(a21h, k1.cx, z.cx); gy = PM_FMA(a21h, k1.cy, z.cy);
        }
        wind_stage<STATIC, POLY>(P, w, PM_FMA(TT(c2), h, t), W);
        rhs3<FAST, METRIC>(P, gl, gx, gy, W, k2, pc);
        if (has2) {
            ax = PM_FMA(TT(a72), gx, ax); ay = PM_FMA(TT(a72), gy, ay);
            ex = PM_FMA(TT(e2), gx, ex); ey = PM_FMA(TT(e2), gy, ey);
        }
        S6ACC(a62);
#define ST3(c) PM_FMA(h, PM_FMA(TT(a32), k2.c, TT(a31) * k1.c), z.c)
        TT_STAGE();
        gl = ST3(lne); gx = ST3(cx); gy = ST3(cy);
        wind_stage<STATIC, POLY>(P, w, PM_FMA(TT(c3), h, t), W);
        rhs3<FAST, METRIC>(P, gl, gx, gy, W, k3, pc);
        ax = PM_FMA(TT(a73), gx, ax); ay = PM_FMA(TT(a73), gy, ay);
        ex = PM_FMA(TT(e3), gx, ex); ey = PM_FMA(TT(e3), gy, ey);
        S6ACC(a63);
#define ST4(c) PM_FMA(h, PM_FMA(TT(a43), k3.c, PM_FMA(TT(a42), k2.c, TT(a41) * k1.c)), z.c)
        TT_STAGE();
        gl = ST4(lne); gx = ST4(cx); gy = ST4(cy);
        wind_stage<STATIC, POLY>(P, w, PM_FMA(TT(c4), h, t), W);
        rhs3<FAST, METRIC>(P, gl, gx, gy, W, k4, pc);
        ax = PM_FMA(TT(a74), gx, ax); ay = PM_FMA(TT(a74), gy, ay);
        ex = PM_FMA(TT(e4), gx, ex); ey = PM_FMA(TT(e4), gy, ey);
        S6ACC(a64);
#define ST5(c) PM_FMA(h, PM_FMA(TT(a54), k4.c, PM_FMA(TT(a53), k3.c, PM_FMA(TT(a52), k2.c, TT(a51) * k1.c))), z.c)
        TT_STAGE();
        gl = ST5(lne); gx = ST5(cx); gy = ST5(cy);
        wind_stage<STATIC, POLY>(P, w, PM_FMA(TT(c5), h, t), W);
        rhs3<FAST, METRIC>(P, gl, gx, gy, W, k5, pc);
        ax = PM_FMA(TT(a75), gx, ax); ay = PM_FMA(TT(a75), gy, ay);
        ex = PM_FMA(TT(e5), gx, ex); ey = PM_FMA(TT(e5), gy, ey);
        S6ACC(a65);
#undef S6ACC
#define ST6(c) PM_FMA(h, PM_FMA(TT(a65), k5.c, PM_FMA(TT(a64), k4.c, PM_FMA(TT(a63), k3.c, PM_FMA(TT(a62), k2.c, TT(a61) * k1.c)))), z.c)
        TT_STAGE();
        gl = ST6(lne); gx = ST6(cx); gy = ST6(cy);
        wind_stage<STATIC, POLY>(P, w, t + h, W);
        rhs3<FAST, METRIC>(P, gl, gx, gy, W, k6, pc);
        ax = PM_FMA(TT(a76), gx, ax); ay = PM_FMA(TT(a76), gy, ay);
        ex = PM_FMA(TT(e6), gx, ex); ey = PM_FMA(TT(e6), gy, ey);
#define S72(c) (has2 ? PM_FMA(TT(a72), k2.c, TT(a71) * k1.c) : TT(a71) * k1.c)
#define ST7(c) PM_FMA(h, PM_FMA(TT(a76), k6.c, PM_FMA(TT(a75), k5.c, PM_FMA(TT(a74), k4.c, PM_FMA(TT(a73), k3.c, S72(c))))), z.c)
        TT_STAGE();
        un.lne = ST7(lne); un.cx = ST7(cx); un.cy = ST7(cy);
        un.x = PM_FMA(h, ax * ipx, z.x); un.y = PM_FMA(h, ay * ipy, z.y);
        rhs3<FAST, METRIC>(P, un.lne, un.cx, un.cy, W, k7, pc);
        st.rhs += 6;
        if (AUTO) {   /* Tsit5 inside the composite: eigen_est = ||k7 - k6|| / ||u - g6||, RMS over the 5 components */
            const double g6x = PM_FMA(h, s6x * ipx, z.x), g6y = PM_FMA(h, s6y * ipy, z.y);
            double a, b, nu = 0.0, nd = 0.0;
            a = k7.lne - k6.lne; b = un.lne - gl; nu = PM_FMA(a, a, nu); nd = PM_FMA(b, b, nd);
            a = k7.cx - k6.cx; b = un.cx - gx; nu = PM_FMA(a, a, nu); nd = PM_FMA(b, b, nd);
            a = k7.cy - k6.cy; b = un.cy - gy; nu = PM_FMA(a, a, nu); nd = PM_FMA(b, b, nd);
            a = un.cx * ipx - gx * ipx; b = un.x - g6x; nu = PM_FMA(a, a, nu); nd = PM_FMA(b, b, nd);
            a = un.cy * ipy - gy * ipy; b = un.y - g6y; nu = PM_FMA(a, a, nu); nd = PM_FMA(b, b, nd);
            eig_nu = nu; eig_nd = nd;     /* the test below needs eigen_est² = nu/nd only: no sqrt, no division */
        }
        TT_STAGE();
        ex = PM_FMA(TT(e7), un.cx, ex) * ipx; ey = PM_FMA(TT(e7), un.cy, ey) * ipy;
#define E12(c) (has2 ? PM_FMA(TT(e2), k2.c, TT(e1) * k1.c) : TT(e1) * k1.c)
#define ERRN(c) (h * PM_FMA(TT(e7), k7.c, PM_FMA(TT(e6), k6.c, PM_FMA(TT(e5), k5.c, PM_FMA(TT(e4), k4.c, PM_FMA(TT(e3), k3.c, E12(c)))))))
#define ERRS(a, b) PM_FMA(pm_fmax_abs(a, b), P.reltol, P.abstol)
        /* EEst² = (1/5) Σ (e_i/s_i)² with ONE reciprocal (kernel order): every numerator is
         * multiplied by the other four scales, the sum is divided by (Π s_i)².  The controller
         * works on ln EEst = ½ ln EEst², so no square root is taken either. */
        {
            double s0 = ERRS(z.lne, un.lne), s1 = ERRS(z.cx, un.cx), s2 = ERRS(z.cy, un.cy);
            double s3 = ERRS(z.x, un.x), s4 = ERRS(z.y, un.y);
            double p2 = s0 * s1, p3 = p2 * s2, p4 = p3 * s3, pp = p4 * s4;
            double q2 = s3 * s4, q1 = s2 * q2, q0 = s1 * q1;
            double n0 = ERRN(lne) * q0, n1 = (ERRN(cx) * s0) * q1, n2 = (ERRN(cy) * p2) * q2;
            double n3 = ((h * ex) * p3) * s4, n4 = (h * ey) * p4;
            double S = n0 * n0;
            S = PM_FMA(n1, n1, S);
            S = PM_FMA(n2, n2, S);
            S = PM_FMA(n3, n3, S);
            S = PM_FMA(n4, n4, S);
            /* every scale is >= abstol, so pp >= abstol⁵ is a plain operand while the state is finite and moderate.  Beyond
             * that (pp = inf or NaN, or within a factor four of overflow) S has overflowed as well — each n_i carries four of
             * the five scales — and EE2 comes out NaN (inf·0) or 0 from either form of the reciprocal */
            double rp = pm_rcp_plain(pp);
            EE2 = (S * 0.2) * (rp * rp);
        }
#undef ST3
#undef ST4
#undef ST5
#undef ST6
#undef ST7
#undef ERRN
#undef ERRS
#undef S72
#undef E12
        }   /* explicit pair */
        if (AUTO) have_eig = true;
        if (!(EE2 == EE2)) { EE2 = pm_inf(); st.status |= 128 /*PICLES_ST_NONFINITE*/; }
        /* PI controller in log space (kernel order): 1/q = γ·qold^β2 / EEst^β1, clamped to
         * [qmin, qmax]; lq = ln(qold) is the carried controller memory. One log + one exp per step. */
        double le = 0.5 * pm_log_coarse(EE2);
        /* (force_dtmin as a threshold, -1 when off — h > 0: a scalar double instead of a lane mask held across the loop) */
        bool accept = (EE2 <= 1.0) || (h <= (P.force_dtmin ? P.dtmin : -1.0));
        if (accept) {
            st.acc++;
            double qi = pm_exp_sat(PM_FMA(beta2, lq, -(beta1 * le)), W.sh) * PI_GAMMA;   /* clamped to [qmin, qmax] right below: saturation is the identity here */
            qi = pm_fmax_c(pm_fmin_c(qi, PM_SCI(PI_QMAX)), PM_SCI(PI_QMIN));   /* the bounds created at their use (two s_mov): as loop invariants they were spilled scalars */
            lq = pm_fmax_c(le, PM_SCI(PI_LNQOLDINIT));
            dt = h * qi;
            z = un;
                    k1 = k7;
            tr = last ? DT : tr + h;
            if (z.lne != z.lne || z.cx != z.cx || z.cy != z.cy || z.x != z.x || z.y != z.y) break;
        } else {
            st.rej++;
            double r = PI_GAMMA * pm_exp_inrange(-(beta1 * le), W.sh);   /* |le| <= 355 (pm_log_coarse of a number in [0, inf]) */
            r = pm_fmax_c(r, PM_SCI(PI_QMIN));
            dt = h * r;
            if (h <= (P.force_dtmin ? -1.0 : P.dtmin)) { st.status |= 64 /*PICLES_ST_DTMIN*/; break; }
        }
    }
    dtn = dt;
    if (AUTO) *asw = (int)((unsigned int)as_count << 1) | (as_stiff ? 1 : 0);
}

#endif /* PICLES_PHYSICS_H */
