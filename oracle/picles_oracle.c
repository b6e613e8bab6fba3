/*
 * picles_oracle.c — CPU restatement of the PiCLES 2D particle-in-cell time step.
 *
 * *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (picles_amd/, libpicles_hip.so) never imports, links or calls it.
 *
 * PARITY STATUS: "parity unpinned" against the true Julia reference for the ODE stepping.
 * The reference holds no assertions and exactly one numeric literal on this path
 * (benchmark/bench02_PW5_allocation.jl:49-50, the seed state for winds (0.1,-0.1), T=300 s),
 * which this file reproduces (tests/test_oracle_golden.py).  Julia is not installed here and
 * the time integration is an unvendored, unpinned third-party dependency
 * (OrdinaryDiffEq.jl DP5 / AutoTsit5, reference Project.toml:6-46); its published
 * algorithm (Dormand-Prince 5(4) + PI controller + Hairer initial step) is restated below and
 * cross-checked against SciPy's RK45/DOP853 in tests/.
 *
 * Every function cites the reference file:line it follows (paths into mochell/PiCLES).
 *
 * Two arithmetic "orders" are provided (runtime switch):
 *   order 0  LITERAL  : the reference's evaluation order, no fused multiply-adds.
 *   order 1  KERNEL   : the same formulas re-associated exactly as the HIP kernels evaluate
 *                       them (shared reciprocals, explicit fma, cross-product form of
 *                       sin 2(a-b)); independent restatement of DESIGN.md section "Kernel order".
 * Two math back-ends (compile-time): glibc libm (default) or -DPO_PMATH: the deterministic
 * primitives of picles_amd/csrc/pmath.h, which make GPU-vs-oracle a bitwise comparison.
 *
 * Build: see oracle/Makefile  (-O2 -ffp-contract=off [-fopenmp]).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#include "../include/picles_hip.h"

#ifdef PO_PMATH
#include "../picles_amd/csrc/pmath.h"
#define o_exp pm_exp
/* the kernels' saturating exponential of the energy, exp(min(max(x, -700), 700)) (pmath.h pm_exp_sat) */
static inline double o_exp_sat(double x)
{
    x = (x > 700.0) ? 700.0 : x;
    x = (x < -700.0) ? -700.0 : x;
    return pm_exp(x);
}
#define o_log pm_log
#define o_log_coarse pm_log_coarse
#define o_rsqrt pm_rsqrt
#define o_pow pm_pow
#define o_tanh pm_tanh
#define o_cosh pm_cosh
static inline double o_log10(double x) { return pm_log(x) * 0.43429448190325182765; }
static inline double o_exp10(double x) { return pm_exp(x * 2.30258509299404568402); }
#define PO_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define o_exp exp
#define o_exp_sat exp
#define o_log log
#define o_log_coarse log
static inline double o_rsqrt(double x) { return 1.0 / sqrt(x); }
#define o_pow pow
#define o_tanh tanh
#define o_cosh cosh
static inline double o_log10(double x) { return log10(x); }
static inline double o_exp10(double x) { return pow(10.0, x); }
#define PO_FMA(a, b, c) fma((a), (b), (c))
#endif

#define PO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * constants
 * ---------------------------------------------------------------------------------------- */
#define G0 9.81 /* hard default of c_g_conversions_vector, particle_waves_v5.jl:281-287 */

/* Dormand-Prince 5(4) tableau (OrdinaryDiffEq DP5 == scipy.integrate RK45 A,B,C,E): PO_DP5 in po_dp5_try */

/* Tsitouras 5(4) tableau (OrdinaryDiffEq Tsit5 constant cache; Ch. Tsitouras, Comput. Math. Appl. 62
 * (2011) 770-775).  Checked against the order conditions in tests/test_tableaux.py. */
typedef struct po_tab {
    double a21, a31, a32, a41, a42, a43, a51, a52, a53, a54, a61, a62, a63, a64, a65;
    double a71, a72, a73, a74, a75, a76, c2, c3, c4, c5, e1, e2, e3, e4, e5, e6, e7;
    double beta1, beta2;
    int has2;
} po_tab;

static const po_tab PO_TSIT5 = {
    0.161, -0.008480655492356989, 0.335480655492357,
    2.8971530571054935, -6.359448489975075, 4.3622954328695815,
    5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525,
    5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383,
    0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774,
    0.161, 0.327, 0.9, 0.9800255409045097,
    -0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
    0.5823571654525552, -0.45808210592918697, 0.015151515151515152,
    0.14, 0.08, 1};

/* OrdinaryDiffEq PI controller defaults for DP5 (alg_utils.jl: beta2 = 4//100,
 * beta1 = 1//5 - 3beta2/4, gamma = 9//10, qmin = 1//5, qmax = 10, qoldinit = 1//10^4) */
#define CTRL_BETA1 0.17
#define CTRL_BETA2 0.04
#define CTRL_GAMMA 0.9
#define CTRL_QMIN 0.2
#define CTRL_QMAX 10.0
#define CTRL_QOLDINIT 1e-4
#define CTRL_LNQOLDINIT (-9.210340371976182) /* ln(1e-4); order 1 carries ln(qold) */
#define PO_QOLD_RESET(M) (((M)->order || (M)->od.solver == 2) ? CTRL_LNQOLDINIT : CTRL_QOLDINIT)   /* solver 2 always runs the log-space controller */

typedef struct po_consts {
    double p, n, e_T, inv_eT, inv_rg, inv_dx, inv_dy;
} po_consts;

#define PO_MAX_KNOTS 8      /* the product's PICLES_MAX_KNOTS */
typedef struct po_model {
    picles_grid g;
    picles_phys ph;
    picles_ode od;
    picles_model md;
    po_consts k;
    int order;
    int nthreads;
    int Nx, Ny;                /* global node counts */
    int j0, nyl;               /* slab: first owned row, number of owned rows */
    int R;                     /* ghost record rows per side (slab stepping) */
    int single_slab;
    int ngroups;
    int64_t N;                 /* Nx * nyl */
    int8_t *mask;              /* local rows */
    uint8_t *grp;              /* 0 not stepped, 1 ocean list, 2 grid-boundary list */
    double *rec;               /* [(nyl+2R)][6][Nx] scatter records (pull / slab stepping) */
    double step_dt;
    int step_flags;
    double *state, *movie;     /* 3 planes */
    double *z;                 /* 5 planes */
    double *qold, *dtn;        /* controller memory, next dt (<0: auto_dt_reset!) */
    int32_t *asw;              /* solver 2: AutoSwitch state (po_integrate_auto) */
    uint8_t *on, *bnd;
    int32_t *status;
    int64_t n_step;
    int64_t *steplist;         /* ocean_points in reference order */
    double *u0, *v0, *u1, *v1;
    double *um, *vm;           /* mid-window level of three-level winds (picles_oracle_set_winds3); NULL: two levels */
    double *m11, *m22, *pc;    /* per-node projection M = diag(m11, m22) and great-circle coefficient (NULL: Cartesian) */
    double tw0, tw1;
    int wind_static;
    int wind_knot;             /* three levels, the middle one at the knot twk of a gridded wind (picles_oracle_set_winds_knot): two straight segments */
    double twk;
    /* a polyline window (picles_oracle_set_winds_polyline): wind_nk >= 2 knots at ptk[0 .. nk) inside (tw0, tw1), the levels there in
     * plu[k] / plv[k]; order 1 also keeps what the product's k_wind_poly lays down: the knots' s, the first segment's slope and the
     * jumps of the slope at the knots (pcu[0] = du, pcu[k] = b_k) */
    int wind_nk;
    double ptk[PO_MAX_KNOTS], psk[PO_MAX_KNOTS];
    double *plu[PO_MAX_KNOTS], *plv[PO_MAX_KNOTS];
    double *pcu[PO_MAX_KNOTS + 1], *pcv[PO_MAX_KNOTS + 1];
    double clock;
    picles_counters cnt;
    char err[256];
} po_model;

/* ------------------------------------------------------------------------------------------
 * derived constants: magic_fractions (particle_waves_v5.jl:87-92), e_T_func (:271)
 * ---------------------------------------------------------------------------------------- */
static void po_derive(const picles_phys *ph, double dx, double dy, po_consts *k)
{
    double q = ph->q;
    k->p = (-1.0 - 10.0 * q) / 2.0;
    k->n = 2.0 * q / (k->p + 4.0 * q);
    /* e_T = sqrt(c_e * c_α^(-p/q) / (γ*c_β*c_D)^(1/n)) */
    k->e_T = sqrt(ph->c_e * o_pow(ph->c_alpha, -k->p / q) /
                  o_pow(ph->gamma * ph->c_beta * ph->c_D, 1.0 / k->n));
    k->inv_eT = 1.0 / k->e_T;
    k->inv_rg = 1.0 / ph->r_g;
    k->inv_dx = 1.0 / dx; /* ProjetionKernel: M = [1/dx 0; 0 1/dy], CartesianGrid.jl:115-121 */
    k->inv_dy = 1.0 / dy;
}

/* ------------------------------------------------------------------------------------------
 * FetchRelations.get_initial_windsea(U10,V10,time_scale; particle_state=true)
 * FetchRelations.jl:314-359 with X_tilde_from_tau :128-130, fₘ_from_X_tilde :165-167,
 * alpha_j :184-186, E_JONSWAP :201-203.
 * ---------------------------------------------------------------------------------------- */
static void po_windsea(double U10, double V10, double T, double out[3])
{
    const double A = 22.8013, xi0 = 2.4097, qx = 0.2748; /* Dulov_fetch_constants :107-111 */
    double Ua = sqrt(U10 * U10 + V10 * V10);
    Ua = (Ua < 0.1) ? 0.1 : Ua;
    T = fabs(T);
    double tau = 9.81 * T / fabs(Ua);
    double X = o_pow(tau / (A * xi0), 1.0 / (1.0 - qx));
    double fm = 3.5 * (9.81 / Ua) * o_pow(X, -0.33);
    double aj = 0.033 * o_pow(fm * Ua / 9.81, 0.67);
    double w = fm * 2.0 * M_PI;
    double wi = 1.0 / w;
    double E = 0.31 * (9.81 * 9.81) * aj * ((wi * wi) * (wi * wi));
    double f_peak = fm * 9.81 / Ua;
    double T_bar = 0.9 * (1.0 / f_peak);
    double cg = 9.81 * T_bar / (4.0 * M_PI);
    out[0] = o_log(E);
    out[1] = cg * U10 / Ua;
    out[2] = cg * V10 / Ua;
}

/* ResetParticleValues (core_2D.jl:307-343) / InitParticleValues default branch: the state a
 * re-seeded particle gets at relative position (0,0) */
static void po_reseed(const po_model *M, double u, double v, double T, double z[5])
{
    if (M->md.init_type == 0) {
        double s[3];
        po_windsea(u, v, T, s);
        z[0] = s[0]; z[1] = s[1]; z[2] = s[2];
    } else {
        z[0] = M->md.default_particle[0];
        z[1] = M->md.default_particle[1];
        z[2] = M->md.default_particle[2];
    }
    z[3] = 0.0;
    z[4] = 0.0;
}

/* GetParticleEnergyMomentum (core_2D.jl:69-78): m = c̄ e / |c̄|² / 2 */
static void po_particle_to_charge(int K, const double z[5], double c[3])
{
    double e = o_exp(z[0]);
    c[0] = e;
    if (K) {   /* kernel order: |c̄|² without the square root, one division */
        double q = (0.5 * e) / PO_FMA(z[1], z[1], z[2] * z[2]);
        c[1] = z[1] * q;
        c[2] = z[2] * q;
        return;
    }
    double sp = sqrt(z[1] * z[1] + z[2] * z[2]);
    c[1] = z[1] * e / (sp * sp) / 2.0;
    c[2] = z[2] * e / (sp * sp) / 2.0;
}
static void po_charge_to_particle(int K, const double c[3], double z[5])
{
    double e = c[0], mx = c[1], my = c[2];
    z[0] = o_log(e);
    z[3] = 0.0;
    z[4] = 0.0;
    if (K) {   /* kernel order: |m|² without the square root, one division */
        double q = e / (2.0 * PO_FMA(mx, mx, my * my));
        z[1] = mx * q;
        z[2] = my * q;
        return;
    }
    double ma = sqrt(mx * mx + my * my);
    z[1] = mx * e / (2.0 * (ma * ma));
    z[2] = my * e / (2.0 * (ma * ma));
}

/* ------------------------------------------------------------------------------------------
 * RHS — particle_system(dz,z,params,t), particle_waves_v5.jl:479-556 (Cartesian: PC ≡ 0)
 * ---------------------------------------------------------------------------------------- */
static void po_rhs_literal(const po_model *M, int64_t idx, const double z[5], double u, double v, double dz[5])
{
    const picles_phys *ph = &M->ph;
    const po_consts *k = &M->k;
    double lne = z[0], cx = z[1], cy = z[2];
    double r_g = ph->r_g;

    double cbar = sqrt(cx * cx + cy * cy);                 /* speed() :297 */
    double U = sqrt(u * u + v * v);
    /* c_g_conversions_vector(abs(c̄), r_g) :281-287 (g = 9.81 hard default) */
    double cgp = fabs(cbar) / r_g;
    double cgp2 = cgp * cgp;
    double kp = G0 / (4.0 * (cgp2 > 1e-2 ? cgp2 : 1e-2));
    double acg = fabs(cgp);
    double wp = G0 / (2.0 * (acg > 0.1 ? acg : 0.1));
    double gx = cx / r_g, gy = cy / r_g;                   /* c_g_conversions :289-295 */
    /* α_func :215-218 */
    double a = U / (2.0 * cgp);
    double alpha = (a > 500.0) ? 500.0 : a;
    /* αₚ :212 */
    double sg = sqrt(gx * gx + gy * gy);
    double sgm = (sg > 1e-4) ? sg : 1e-4;
    double ap = (u * gx + v * gy) / (2.0 * (sgm * sgm));
    /* H_β :274, Δ_β :275 */
    double H = 0.5 * (1.0 + o_tanh(k->p * (ap - 0.85)));
    double sech = 1.0 / o_cosh(10.0 * (ap - 0.85));
    double D = 1.0 - 1.25 * (sech * sech);

    double It = 0.0, Dt = 0.0, Scg = 0.0, Sd = 0.0;
    if (ph->input) It = ph->C_e * H * (alpha * alpha);        /* Ĩ_func :317-321 */
    if (ph->dissipation) {                                 /* D̃_func_lne :331-335 */
        double t = kp / k->e_T;
        double pw = (2.0 * k->n == 4.0) ? (t * t) * (t * t) : o_pow(t, 2.0 * k->n);
        Dt = o_exp(k->n * lne) * pw;
    }
    if (ph->peak_shift)                                    /* S_cg :340 */
        Scg = ph->C_alpha * D * ((kp * kp) * (kp * kp)) * o_exp(2.0 * lne);
    if (ph->direction) {                                   /* S_dir :345-346, sin2_a_min_b :242-249 */
        double a2 = U / (2.0 * sg);
        a2 = (a2 > 500.0) ? 500.0 : a2;
        double UG = U * sg;
        double s2;
        if (UG == 0.0)
            s2 = 0.0;
        else
            s2 = (2.0 / (UG * UG)) *
                 (u * v * (2.0 * (gy * gy) - sg * sg) - gx * gy * (2.0 * (v * v) - U * U));
        if (ph->dir_deadband > 0.0) {   /* opt-in dead band on |sin(θ_c - θ_w)| = |u gy - v gx| / (U |g|) */
            double crs = u * gy - v * gx;
            if (crs * crs <= ph->dir_deadband * ph->dir_deadband * (UG * UG)) s2 = 0.0;
        }
        Sd = a2 * a2 * ph->C_phi * H * s2;
    }
    dz[0] = wp * r_g * Scg + wp * (It - Dt);               /* :526 */
    if (M->pc) {
        /* S_sphere_tilde = PropagationCorrection(c̄_x) = c̄_x * coefficient (:521-530,
         * spherical_grid_corrections.jl:3-21) */
        double Ss = cx * M->pc[idx];
        dz[1] = -cx * wp * r_g * Scg + cy * Sd + cy * Ss;
        dz[2] = -cy * wp * r_g * Scg - cx * Sd - cx * Ss;
    } else {
        dz[1] = -cx * wp * r_g * Scg + cy * Sd;            /* :529 (PC ≡ 0 on the Cartesian mesh) */
        dz[2] = -cy * wp * r_g * Scg - cx * Sd;            /* :530 */
    }
    if (ph->propagation) {                                 /* :536, M*[c̄x,c̄y]; M = diag(1/dx,1/dy) or per node */
        dz[3] = (M->m11 ? M->m11[idx] : k->inv_dx) * cx;
        dz[4] = (M->m22 ? M->m22[idx] : k->inv_dy) * cy;
    } else {
        dz[3] = 0.0;
        dz[4] = 0.0;
    }
}

/* KERNEL order: the same RHS as the HIP kernels evaluate it (DESIGN.md "Kernel order", physics.h rhs3).
 * Everything is expressed through y = 1/|c̄| (1/c_gp = r_g y; one deterministic reciprocal square root feeds k_p, ω_p, α, α_p and
 * sin2, the powers of r_g ride in the constants); |g| is taken as c_gp (identical in exact arithmetic); sin 2(θ_c-θ_w) is
 * evaluated as 2·cross·dot/(U|g|)² (algebraically equal to sin2_a_min_b :242-249, exactly zero for aligned vectors); H_β and the
 * sech² of Δ_β from one exponential and one reciprocal; fused multiply-adds written out.
 * The reference's guards — max(c_gp, 0.1) in ω_p and k_p (:331-336), max(c_gp, 1e-4) in α_p (:505-507), min(α, 500) (:340), the
 * zero tests of sin2_a_min_b — are ceilings ymax, sgmax on y, y², a cap on α² and the same zero tests; for a PLAIN particle
 * (y <= ymax under a wind with 1e-290 <= U² and (U²/4) r_g² <= qU2r_max) none of them acts, the wind speed cancels out of the
 * direction term, and the kernels evaluate the shorter form below; every other particle takes the guarded form. */
static void po_rhs_kernel(const po_model *M, int64_t idx, const double z[5], double u, double v, double dz[5])
{
    const picles_phys *ph = &M->ph;
    const po_consts *k = &M->k;
    double lne = z[0], cx = z[1], cy = z[2];
    /* constants of the kernel order (picles_hip.hip, picles_create) */
    const double r_g = ph->r_g;
    const double g4 = 0.25 * G0, g42 = g4 * g4, K = g42 * g42;
    const double ie2 = k->inv_eT * k->inv_eT, inv_eT4 = ie2 * ie2;
    const double rg2 = r_g * r_g, rg4 = rg2 * rg2, rg8 = rg4 * rg4;
    const double Cw = (0.5 * G0) * r_g, Chrh = -0.25 * r_g;
    const double ymax = 10.0 / r_g, sgmax = 1e8 / rg2;
    const double KeT4y = (K * inv_eT4) * rg8, KrCay = ((K * r_g) * ph->C_alpha) * rg8;
    const double Cs = (0.5 * ph->C_phi) * rg2, Cdir2 = 2.0 * ph->C_phi;
    const double g4rg2 = g4 * rg2;
    const double qU2r_max = 249999.0 / (ymax * ymax);

    double c2 = PO_FMA(cx, cx, cy * cy);
    double U2 = PO_FMA(u, u, v * v);
    double qU2r = (0.25 * U2) * rg2;
    double y = o_rsqrt(c2);                 /* a NaN y (|c̄| = 0, inf or NaN) is not plain and takes the ceilings (IEEE minNum) */
    double y2 = y * y;
    double dotc = PO_FMA(u, cx, v * cy);
    double crsc = u * cy - v * cx;
    int plain = (U2 >= 1e-290 && qU2r <= qU2r_max) && (y <= ymax);
    double ym = fmin(y, ymax);              /* the identity for a plain particle, like the next three */
    double wp = Cw * ym;
    double aph = (Chrh * dotc) * fmin(y2, sgmax);      /* -α_p/2 */
    if (!plain) {   /* beyond ±699 the exponential below underflows in its second power either way (same H, Δ): the kernels clamp, a NaN stays */
        aph = (aph > 699.0) ? 699.0 : aph;
        aph = (aph < -699.0) ? -699.0 : aph;
    }
    double m2 = ym * ym;
    double m4 = m2 * m2;
    double yh = aph + 0.425;                /* -ya/2, ya = α_p - 0.85 (exact halving) */
    /* one reciprocal for H_β and Δ_β: r = 1/(hp (1+t)²), H = (1+t)² r, Δ = 1 - 5t hp r */
    double hp, t, H, rHD, t12;
    if (k->p == 0.75) {
        /* 2p = 3/2: eH = exp(-2p ya) = w^(±3) and t = exp(-20|ya|) = w^40 with w = exp(-|ya|/2) = exp(-|yh|); s = w³, hp = 1 + s,
         * H = 1/(1+s) for ya >= 0 and s/(1+s) below; the powers along 1, 2, 3, 5, 10, 20, 40 (physics.h rhs3) */
        double w = o_exp(-fabs(yh));
        double w2 = w * w, s3 = w2 * w;
        double w5 = s3 * w2, w10 = w5 * w5, w20 = w10 * w10;
        t = w20 * w20;
        hp = 1.0 + s3;
        double t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = 1.0 / (hp * t12);
        H = (t12 * rHD) * ((yh <= 0.0) ? 1.0 : s3);
    } else {
        /* general p: t = exp(-20|ya|) is taken as 0 once 5t < 2^-54 (Δ rounds to 1); eH stays finite (argument <= 700) */
        double ya = -2.0 * yh;
        double harg = (-2.0 * k->p) * ya;
        hp = 1.0 + o_exp((harg > 700.0) ? 700.0 : harg);
        double targ = -20.0 * fabs(ya);
        t = (targ <= -40.0) ? 0.0 : o_exp(targ);
        double t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = 1.0 / (hp * t12);
        H = t12 * rHD;
    }
    double D = PO_FMA(-((5.0 * t) * hp), rHD, 1.0);

    double y2H = y2 * H;
    double aH, Sd = 0.0;
    if (plain) {
        aH = qU2r * y2H;                                             /* α² H, α² = (U²/4) r_g² y² below its cap */
        if (ph->direction) Sd = ((crsc * dotc) * (y2 * y2H)) * Cs;   /* the wind speed cancels: no 1/U² */
    } else {
        aH = fmin(qU2r * y2, 250000.0) * H;                          /* α² = min(U/(2 c_gp), 500)² */
        if (ph->direction && !(U2 == 0.0 || c2 == 0.0)) Sd = ((crsc * dotc) * (y2 * aH)) * (Cdir2 * (1.0 / U2));
    }
    if (ph->direction) {   /* opt-in dead band (picles_phys.dir_deadband) */
        double db2 = ph->dir_deadband * ph->dir_deadband;
        if (db2 > 0.0 && crsc * crsc <= db2 * (U2 * c2)) Sd = 0.0;
    }
    int n_is_2 = (k->n == 2.0);
    double Ek8 = 0.0;
    if ((ph->dissipation && n_is_2) || ph->peak_shift) Ek8 = o_exp_sat(lne + lne) * (m4 * m4);
    double Dt = 0.0;
    if (ph->dissipation) {
        if (n_is_2) {
            Dt = Ek8 * KeT4y;
        } else {
            double ke = (g4rg2 * m2) * k->inv_eT;
            Dt = o_exp(k->n * lne) * o_pow(ke, 2.0 * k->n);
        }
    }
    double IDt = PO_FMA(ph->input ? ph->C_e : 0.0, aH, -Dt);
    double wrS = 0.0;
    if (ph->peak_shift) wrS = (wp * D) * (Ek8 * KrCay);
    if (M->pc) Sd = Sd + cx * M->pc[idx];   /* great-circle term rides on the direction term */
    dz[0] = PO_FMA(wp, IDt, wrS);
    dz[1] = PO_FMA(cy, Sd, -(cx * wrS));
    dz[2] = -PO_FMA(cx, Sd, cy * wrS);
    if (ph->propagation) {
        dz[3] = cx * (M->m11 ? M->m11[idx] : k->inv_dx);
        dz[4] = cy * (M->m22 ? M->m22[idx] : k->inv_dy);
    } else {
        dz[3] = 0.0;
        dz[4] = 0.0;
    }
}

static inline void po_rhs(const po_model *M, int64_t idx, const double z[5], double u, double v, double dz[5])
{
    if (M->order == 0) po_rhs_literal(M, idx, z, u, v, dz);
    else po_rhs_kernel(M, idx, z, u, v, dz);
}

/* node wind at absolute time t: the boundary's replacement of the closures u(x,y,t), v(x,y,t)
 * (particle_waves_v5.jl:494-495): the interpolant through the node-sampled levels of the step window — two levels: linear in t;
 * three levels (t0, (t0+t1)/2, t1): the parabola.  Order 0 writes the Lagrange form, order 1 the kernels' Newton form
 * u0 + s (du + (s - 1) bu), bu = 2 ((u0 + u1) - 2 um) (physics.h, Wind). */
static inline void po_wind(const po_model *M, int64_t idx, double t, double *u, double *v)
{
    if (M->wind_static) {
        *u = M->u0[idx];
        *v = M->v0[idx];
        return;
    }
    if (M->wind_nk > 1) {
        if (M->order == 0) {
            /* several time knots inside the step: the lerp of the segment t falls into (Utils/WindEmulator.jl:18-43) */
            const int nk = M->wind_nk;
            int j = 0;
            while (j < nk && t >= M->ptk[j]) j++;                 /* segment j: [T_j, T_j+1), T_0 = tw0, T_k = ptk[k-1], T_nk+1 = tw1 */
            const double ta = j ? M->ptk[j - 1] : M->tw0, tb = (j < nk) ? M->ptk[j] : M->tw1;
            const double ua = j ? M->plu[j - 1][idx] : M->u0[idx], ub = (j < nk) ? M->plu[j][idx] : M->u1[idx];
            const double va = j ? M->plv[j - 1][idx] : M->v0[idx], vb = (j < nk) ? M->plv[j][idx] : M->v1[idx];
            const double f = (t - ta) / (tb - ta);
            *u = ua + (ub - ua) * f;
            *v = va + (vb - va) * f;
        } else {
            /* kernel order (physics.h, wind_eval<true>): u0 + s du + Σ_k max(s - s_k, 0) b_k in the order of the knots */
            const double idt = 1.0 / (M->tw1 - M->tw0);
            const double s = (t - M->tw0) * idt;
            double uu = PO_FMA(M->pcu[0][idx], s, M->u0[idx]), vv = PO_FMA(M->pcv[0][idx], s, M->v0[idx]);
            for (int k = 0; k < M->wind_nk; k++) {
                double sp = s - M->psk[k];
                sp = (sp > 0.0) ? sp : 0.0;
                uu = PO_FMA(M->pcu[k + 1][idx], sp, uu);
                vv = PO_FMA(M->pcv[k + 1][idx], sp, vv);
            }
            *u = uu;
            *v = vv;
        }
        return;
    }
    if (M->order == 0) {
        if (M->um && M->wind_knot) {
            /* a gridded wind with one of its time knots inside the step (Utils/WindEmulator.jl:18-43: linear_interpolation in t,
             * evaluated by the RHS at every stage time, particle_waves_v5.jl:494-495): the lerp of the segment t falls into */
            if (t < M->twk) {
                double f = (t - M->tw0) / (M->twk - M->tw0);
                *u = M->u0[idx] + (M->um[idx] - M->u0[idx]) * f;
                *v = M->v0[idx] + (M->vm[idx] - M->v0[idx]) * f;
            } else {
                double f = (t - M->twk) / (M->tw1 - M->twk);
                *u = M->um[idx] + (M->u1[idx] - M->um[idx]) * f;
                *v = M->vm[idx] + (M->v1[idx] - M->vm[idx]) * f;
            }
            return;
        }
        double s = (t - M->tw0) / (M->tw1 - M->tw0);
        if (M->um) {
            double l0 = (2.0 * s - 1.0) * (s - 1.0), lm = 4.0 * s * (1.0 - s), l1 = s * (2.0 * s - 1.0);
            *u = M->u0[idx] * l0 + M->um[idx] * lm + M->u1[idx] * l1;
            *v = M->v0[idx] * l0 + M->vm[idx] * lm + M->v1[idx] * l1;
        } else {
            *u = M->u0[idx] + (M->u1[idx] - M->u0[idx]) * s;
            *v = M->v0[idx] + (M->v1[idx] - M->v0[idx]) * s;
        }
    } else {
        /* kernel order (physics.h, wind_eval2; kernels.h, load_wind): the parabola in Newton form, u0 + s (du + (s - 1) bu) with
         * bu = 2 ((u0 + u1) - 2 um) (two levels: bu = 0), or the knot form u0 + s du + max(s - sk, 0) bu with the first segment's
         * slope du = (uk - u0)/sk and the jump of the slope bu = (u1 - uk)/(1 - sk) - du */
        const double idt = 1.0 / (M->tw1 - M->tw0);
        double s = (t - M->tw0) * idt;
        if (M->um && M->wind_knot) {
            const double sk = (M->twk - M->tw0) * idt, isk = 1.0 / sk, i1sk = 1.0 / (1.0 - sk);
            const double du = (M->um[idx] - M->u0[idx]) * isk, dv = (M->vm[idx] - M->v0[idx]) * isk;
            const double bu = (M->u1[idx] - M->um[idx]) * i1sk - du, bv = (M->v1[idx] - M->vm[idx]) * i1sk - dv;
            double sp = s - sk;
            sp = (sp > 0.0) ? sp : 0.0;
            *u = PO_FMA(bu, sp, PO_FMA(du, s, M->u0[idx]));
            *v = PO_FMA(bv, sp, PO_FMA(dv, s, M->v0[idx]));
            return;
        }
        double s1 = s - 1.0;
        double bu = 0.0, bv = 0.0;
        if (M->um) {
            bu = 2.0 * ((M->u0[idx] + M->u1[idx]) - 2.0 * M->um[idx]);
            bv = 2.0 * ((M->v0[idx] + M->v1[idx]) - 2.0 * M->vm[idx]);
        }
        *u = PO_FMA(PO_FMA(bu, s1, M->u1[idx] - M->u0[idx]), s, M->u0[idx]);
        *v = PO_FMA(PO_FMA(bv, s1, M->v1[idx] - M->v0[idx]), s, M->v0[idx]);
    }
}

/* d/dt of the same interpolant at t (the explicit time derivative the Rosenbrock23 attempt needs) */
static inline void po_wind_dt(const po_model *M, int64_t idx, double t, double *dudt, double *dvdt)
{
    *dudt = *dvdt = 0.0;
    if (M->wind_static) return;
    const double idt = 1.0 / (M->tw1 - M->tw0);
    if (M->wind_nk > 1) {
        const int nk = M->wind_nk;
        if (M->order == 0) {
            int j = 0;
            while (j < nk && t >= M->ptk[j]) j++;
            const double ta = j ? M->ptk[j - 1] : M->tw0, tb = (j < nk) ? M->ptk[j] : M->tw1;
            const double ua = j ? M->plu[j - 1][idx] : M->u0[idx], ub = (j < nk) ? M->plu[j][idx] : M->u1[idx];
            const double va = j ? M->plv[j - 1][idx] : M->v0[idx], vb = (j < nk) ? M->plv[j][idx] : M->v1[idx];
            *dudt = (ub - ua) / (tb - ta);
            *dvdt = (vb - va) / (tb - ta);
            return;
        }
        const double s = (t - M->tw0) * idt;
        double su = M->pcu[0][idx], sv = M->pcv[0][idx];
        su = (s >= M->psk[0]) ? su + M->pcu[1][idx] : su;
        sv = (s >= M->psk[0]) ? sv + M->pcv[1][idx] : sv;
        for (int k = 1; k < nk; k++)
            if (s >= M->psk[k]) { su = su + M->pcu[k + 1][idx]; sv = sv + M->pcv[k + 1][idx]; }
        *dudt = su * idt;
        *dvdt = sv * idt;
        return;
    }
    if (M->um && M->wind_knot) {
        if (M->order == 0) {
            if (t < M->twk) {
                *dudt = (M->um[idx] - M->u0[idx]) / (M->twk - M->tw0);
                *dvdt = (M->vm[idx] - M->v0[idx]) / (M->twk - M->tw0);
            } else {
                *dudt = (M->u1[idx] - M->um[idx]) / (M->tw1 - M->twk);
                *dvdt = (M->v1[idx] - M->vm[idx]) / (M->tw1 - M->twk);
            }
            return;
        }
        const double s = (t - M->tw0) * idt;
        const double sk = (M->twk - M->tw0) * idt, isk = 1.0 / sk, i1sk = 1.0 / (1.0 - sk);
        const double du = (M->um[idx] - M->u0[idx]) * isk, dv = (M->vm[idx] - M->v0[idx]) * isk;
        const double bu = (M->u1[idx] - M->um[idx]) * i1sk - du, bv = (M->v1[idx] - M->vm[idx]) * i1sk - dv;
        *dudt = ((s >= sk) ? du + bu : du) * idt;
        *dvdt = ((s >= sk) ? dv + bv : dv) * idt;
        return;
    }
    /* parabola (two levels: bu = 0): (du + (2 s - 1) bu) / (tw1 - tw0) */
    double s21 = PO_FMA(2.0, (t - M->tw0) * idt, -1.0);
    double bu = 0.0, bv = 0.0;
    if (M->um) {
        bu = 2.0 * ((M->u0[idx] + M->u1[idx]) - 2.0 * M->um[idx]);
        bv = 2.0 * ((M->v0[idx] + M->v1[idx]) - 2.0 * M->vm[idx]);
    }
    *dudt = PO_FMA(bu, s21, M->u1[idx] - M->u0[idx]) * idt;
    *dvdt = PO_FMA(bv, s21, M->v1[idx] - M->v0[idx]) * idt;
}

/* ------------------------------------------------------------------------------------------
 * Time integration: step!(integrator, DT, true) (call site mapping_2D.jl:152).
 * Third-party semantics (OrdinaryDiffEq.jl v6, unpinned): DP5 perform_step!, ODE_DEFAULT_NORM
 * (RMS), calculate_residuals, PIController, ode_determine_initdt, fix_dt_at_bounds!,
 * modify_dt_for_tstops!.  See SURVEY.md Appendix C.
 * ---------------------------------------------------------------------------------------- */
typedef struct po_pstats {
    uint64_t rhs, acc, rej;
    int status;
} po_pstats;

static inline double po_norm5_lit(const double a[5])
{
    double s = 0.0;
    for (int i = 0; i < 5; i++) s += a[i] * a[i];
    return sqrt(s / 5.0);
}
static inline double po_norm5_k(const double a[5])
{
    double s = a[0] * a[0];
    for (int i = 1; i < 5; i++) s = PO_FMA(a[i], a[i], s);
    return sqrt(s * 0.2);
}

/* ode_determine_initdt (OrdinaryDiffEq initdt.jl, out-of-place form), = auto_dt_reset!
 * (mapping_2D.jl:95,103,110).  f0 = f(u0,t) is passed in (it is also the FSAL k1). */
static double po_initdt_kernel(const po_model *M, int64_t idx, const double u0[5], const double f0[5],
                               double t, po_pstats *st);
static double po_initdt(const po_model *M, int64_t idx, const double u0[5], const double f0[5],
                        double t, po_pstats *st)
{
    const picles_ode *od = &M->od;
    if (M->order) return po_initdt_kernel(M, idx, u0, f0, t, st);
    double sk[5], a0[5], a1[5];
    for (int i = 0; i < 5; i++) {
        sk[i] = od->abstol + fabs(u0[i]) * od->reltol;
        a0[i] = u0[i] / sk[i];
        a1[i] = f0[i] / sk[i];
    }
    double d0 = po_norm5_lit(a0);
    double d1 = po_norm5_lit(a1);
    double dt0;
    if (d0 < 1e-5 || d1 < 1e-5) dt0 = 1e-6;
    else dt0 = (d0 / d1) / 100.0;
    if (dt0 < 10.0 * 2.220446049250313e-16) return 1e-6;
    double u1[5], f1[5], uw, vw;
    for (int i = 0; i < 5; i++) u1[i] = u0[i] + dt0 * f0[i];
    po_wind(M, idx, t + dt0, &uw, &vw);
    po_rhs(M, idx, u1, uw, vw, f1);
    st->rhs++;
    for (int i = 0; i < 5; i++) a1[i] = (f1[i] - f0[i]) / sk[i];
    double d2 = po_norm5_lit(a1) / dt0;
    double m = (d1 > d2) ? d1 : d2;
    double dt1;
    if (m <= 1e-15) {
        double c = dt0 * 1e-3;
        dt1 = (1e-6 > c) ? 1e-6 : c;
    } else {
        dt1 = o_exp10(-(2.0 + o_log10(m)) / 5.0);
    }
    double h = 100.0 * dt0;
    if (dt1 < h) h = dt1;
    /* a NaN estimate (non-finite f) must not poison dt: fall back to the smallest step */
    if (!(h == h)) h = 1e-6;
    return (od->dtmin > h) ? od->dtmin : h;
}

/* the same estimate in KERNEL order (physics.h init_dt): one reciprocal for the three state scales,
 * squared norms, dt0 through the deterministic rsqrt, coarse logarithm for the second-derivative term */
static inline double po_ms5(const double a[5])
{
    double s = a[0] * a[0];
    for (int i = 1; i < 5; i++) s = PO_FMA(a[i], a[i], s);
    return s * 0.2;
}
static double po_initdt_kernel(const po_model *M, int64_t idx, const double u0[5], const double f0[5],
                               double t, po_pstats *st)
{
    const picles_ode *od = &M->od;
    double r[5], a0[5], a1[5];
    double s0 = PO_FMA(fabs(u0[0]), od->reltol, od->abstol);
    double s1 = PO_FMA(fabs(u0[1]), od->reltol, od->abstol);
    double s2 = PO_FMA(fabs(u0[2]), od->reltol, od->abstol);
    double p01 = s0 * s1;
    double rp = 1.0 / (p01 * s2);
    r[0] = (s1 * s2) * rp; r[1] = (s0 * s2) * rp; r[2] = p01 * rp;
    r[3] = 1.0 / PO_FMA(fabs(u0[3]), od->reltol, od->abstol);
    r[4] = 1.0 / PO_FMA(fabs(u0[4]), od->reltol, od->abstol);
    for (int i = 0; i < 5; i++) { a0[i] = u0[i] * r[i]; a1[i] = f0[i] * r[i]; }
    double S0 = po_ms5(a0), S1 = po_ms5(a1);
    double dt0;
    if (S0 < 1e-10 || S1 < 1e-10) dt0 = 1e-6;
    else dt0 = (0.01 * S0) * o_rsqrt(S0 * S1);
    if (dt0 < 10.0 * 2.220446049250313e-16) return 1e-6;
    double u1[5], f1[5], uw, vw;
    for (int i = 0; i < 5; i++) u1[i] = PO_FMA(dt0, f0[i], u0[i]);
    po_wind(M, idx, t + dt0, &uw, &vw);
    po_rhs(M, idx, u1, uw, vw, f1);
    st->rhs++;
    for (int i = 0; i < 5; i++) a1[i] = (f1[i] - f0[i]) * r[i];
    double S2 = po_ms5(a1) / (dt0 * dt0);
    double m2 = (S1 > S2) ? S1 : S2;
    double dt1;
    if (!(m2 > 1e-30)) {
        double c = dt0 * 1e-3;
        dt1 = (1e-6 > c) ? 1e-6 : c;
    } else {
        dt1 = o_exp(PO_FMA(-0.1, o_log_coarse(m2), -0.92103403719761827));
    }
    double h = 100.0 * dt0;
    if (dt1 < h) h = dt1;
    if (!(h == h)) h = 1e-6;
    return (od->dtmin > h) ? od->dtmin : h;
}

/* One attempted DP5 step of size h from (u0,k1) at absolute time t. Returns EEst. */
static double po_dp5_try_e(const po_model *M, int64_t idx, const double u0[5], const double k1[5],
                           double t, double h, double unew[5], double k7[5], po_pstats *st, double *eigen_est);
static double po_dp5_try(const po_model *M, int64_t idx, const double u0[5], const double k1[5],
                         double t, double h, double unew[5], double k7[5], po_pstats *st)
{
    return po_dp5_try_e(M, idx, u0, k1, t, h, unew, k7, st, NULL);
}
static double po_dp5_try_e(const po_model *M, int64_t idx, const double u0[5], const double k1[5],
                           double t, double h, double unew[5], double k7[5], po_pstats *st, double *eigen_est)
{
    const picles_ode *od = &M->od;
    /* tableau of the selected solver (od->solver: 0 DP5, 1 Tsit5); same 7-stage FSAL structure */
    static const po_tab PO_DP5 = {1.0 / 5.0, 3.0 / 40.0, 9.0 / 40.0, 44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0,
                                  19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0,
                                  9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0,
                                  35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0,
                                  1.0 / 5.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0,
                                  -71.0 / 57600.0, 0.0, 71.0 / 16695.0, -71.0 / 1920.0, 17253.0 / 339200.0, -22.0 / 525.0, 1.0 / 40.0,
                                  0.17, 0.04, 0};
    const po_tab *T = od->solver ? &PO_TSIT5 : &PO_DP5;
#define A21 T->a21
#define A31 T->a31
#define A32 T->a32
#define A41 T->a41
#define A42 T->a42
#define A43 T->a43
#define A51 T->a51
#define A52 T->a52
#define A53 T->a53
#define A54 T->a54
#define A61 T->a61
#define A62 T->a62
#define A63 T->a63
#define A64 T->a64
#define A65 T->a65
#define A71 T->a71
#define A72 T->a72
#define A73 T->a73
#define A74 T->a74
#define A75 T->a75
#define A76 T->a76
#define C2 T->c2
#define C3 T->c3
#define C4 T->c4
#define C5 T->c5
#define E1 T->e1
#define E2 T->e2
#define E3 T->e3
#define E4 T->e4
#define E5 T->e5
#define E6 T->e6
#define E7 T->e7
    double k2[5], k3[5], k4[5], k5[5], k6[5], g[5], uw, vw;
    int K = M->order;
#define STAGE(expr_lit, expr_k) for (int i = 0; i < 5; i++) g[i] = K ? (expr_k) : (expr_lit)
    if (!K) {
        double a = h * A21;
        for (int i = 0; i < 5; i++) g[i] = u0[i] + a * k1[i];
    } else {
        double a = h * A21;
        for (int i = 0; i < 5; i++) g[i] = PO_FMA(a, k1[i], u0[i]);
    }
    /* kernel order, x and y: the tendencies are c̄x/Δx, c̄y/Δy of the stage state, so the tableau sums
     * Σ a7i c̄_i and Σ e_i c̄_i run on the stage c̄ and meet the projection once, at the end */
    double sx = A71 * u0[1], sy = A71 * u0[2], ex = E1 * u0[1], ey = E1 * u0[2];
    const double ipx = M->ph.propagation ? (M->m11 ? M->m11[idx] : M->k.inv_dx) : 0.0;
    const double ipy = M->ph.propagation ? (M->m22 ? M->m22[idx] : M->k.inv_dy) : 0.0;
#define XYACC(a7, e) do { sx = PO_FMA(a7, g[1], sx); sy = PO_FMA(a7, g[2], sy); ex = PO_FMA(e, g[1], ex); ey = PO_FMA(e, g[2], ey); } while (0)
    /* Σ a6j c̄_j: the x,y position of stage 6, needed only for the stiffness estimate of the auto-switching solver */
    double s6x = A61 * u0[1], s6y = A61 * u0[2];
#define S6ACC(a6) do { s6x = PO_FMA(a6, g[1], s6x); s6y = PO_FMA(a6, g[2], s6y); } while (0)
    po_wind(M, idx, K ? PO_FMA(C2, h, t) : t + C2 * h, &uw, &vw);
    po_rhs(M, idx, g, uw, vw, k2);
    if (K && T->has2) XYACC(A72, E2);
    S6ACC(A62);
    STAGE(u0[i] + h * (A31 * k1[i] + A32 * k2[i]),
          PO_FMA(h, PO_FMA(A32, k2[i], A31 * k1[i]), u0[i]));
    po_wind(M, idx, K ? PO_FMA(C3, h, t) : t + C3 * h, &uw, &vw);
    po_rhs(M, idx, g, uw, vw, k3);
    XYACC(A73, E3);
    S6ACC(A63);
    STAGE(u0[i] + h * (A41 * k1[i] + A42 * k2[i] + A43 * k3[i]),
          PO_FMA(h, PO_FMA(A43, k3[i], PO_FMA(A42, k2[i], A41 * k1[i])), u0[i]));
    po_wind(M, idx, K ? PO_FMA(C4, h, t) : t + C4 * h, &uw, &vw);
    po_rhs(M, idx, g, uw, vw, k4);
    XYACC(A74, E4);
    S6ACC(A64);
    STAGE(u0[i] + h * (A51 * k1[i] + A52 * k2[i] + A53 * k3[i] + A54 * k4[i]),
          PO_FMA(h, PO_FMA(A54, k4[i], PO_FMA(A53, k3[i], PO_FMA(A52, k2[i], A51 * k1[i]))), u0[i]));
    po_wind(M, idx, K ? PO_FMA(C5, h, t) : t + C5 * h, &uw, &vw);
    po_rhs(M, idx, g, uw, vw, k5);
    XYACC(A75, E5);
    S6ACC(A65);
    STAGE(u0[i] + h * (A61 * k1[i] + A62 * k2[i] + A63 * k3[i] + A64 * k4[i] + A65 * k5[i]),
          PO_FMA(h, PO_FMA(A65, k5[i], PO_FMA(A64, k4[i], PO_FMA(A63, k3[i], PO_FMA(A62, k2[i], A61 * k1[i])))), u0[i]));
    po_wind(M, idx, t + h, &uw, &vw);
    po_rhs(M, idx, g, uw, vw, k6);
    XYACC(A76, E6);
#undef XYACC
#undef S6ACC
    const double g6x = PO_FMA(h, s6x * ipx, u0[3]), g6y = PO_FMA(h, s6y * ipy, u0[4]);
    for (int i = 0; i < 5; i++)
        if (K) {
            double s72 = T->has2 ? PO_FMA(A72, k2[i], A71 * k1[i]) : A71 * k1[i];
            unew[i] = PO_FMA(h, PO_FMA(A76, k6[i], PO_FMA(A75, k5[i], PO_FMA(A74, k4[i], PO_FMA(A73, k3[i], s72)))), u0[i]);
            if (i == 3) unew[3] = PO_FMA(h, sx * ipx, u0[3]);
            if (i == 4) unew[4] = PO_FMA(h, sy * ipy, u0[4]);
        } else if (T->has2) {   /* Tsit5 perform_step!: a71 k1 + a72 k2 + ... */
            unew[i] = u0[i] + h * (A71 * k1[i] + A72 * k2[i] + A73 * k3[i] + A74 * k4[i] + A75 * k5[i] + A76 * k6[i]);
        } else {                /* DP5 perform_step!: the zero a72 term does not appear */
            unew[i] = u0[i] + h * (A71 * k1[i] + A73 * k3[i] + A74 * k4[i] + A75 * k5[i] + A76 * k6[i]);
        }
    po_rhs(M, idx, unew, uw, vw, k7);
    st->rhs += 6;
    if (eigen_est) {
        /* Tsit5 inside a CompositeAlgorithm: eigen_est = ||k7 - k6|| / ||u - g6|| (RMS norms over the 5 components);
         * the x,y components of g6 follow the same stage formula (kernel order: sums on the stage velocities) */
        double nu = 0.0, nd = 0.0;
        for (int i = 0; i < 5; i++) {
            double a = k7[i] - k6[i], b = unew[i] - ((i == 3) ? g6x : (i == 4) ? g6y : g[i]);
            nu = PO_FMA(a, a, nu);
            nd = PO_FMA(b, b, nd);
        }
        eigen_est[0] = nu; eigen_est[1] = nd;      /* eigen_est² = nu/nd: the stiffness test needs no more */
    }
#undef STAGE
    double at[5], sc[5];
    for (int i = 0; i < 5; i++) {
        double ut, m0 = fabs(u0[i]), m1 = fabs(unew[i]);
        double mm = (m0 > m1) ? m0 : m1;
        if (!K) {
            if (T->has2) ut = h * (E1 * k1[i] + E2 * k2[i] + E3 * k3[i] + E4 * k4[i] + E5 * k5[i] + E6 * k6[i] + E7 * k7[i]);
            else ut = h * (E1 * k1[i] + E3 * k3[i] + E4 * k4[i] + E5 * k5[i] + E6 * k6[i] + E7 * k7[i]);
            at[i] = ut / (od->abstol + mm * od->reltol);
        } else {
            double e12 = T->has2 ? PO_FMA(E2, k2[i], E1 * k1[i]) : E1 * k1[i];
            at[i] = h * PO_FMA(E7, k7[i], PO_FMA(E6, k6[i], PO_FMA(E5, k5[i], PO_FMA(E4, k4[i], PO_FMA(E3, k3[i], e12)))));
            if (i == 3) at[3] = h * (PO_FMA(E7, unew[1], ex) * ipx);
            if (i == 4) at[4] = h * (PO_FMA(E7, unew[2], ey) * ipy);
            sc[i] = PO_FMA(fmax(m0, m1), od->reltol, od->abstol);
        }
    }
    if (!K) return po_norm5_lit(at);
    /* kernel order: returns EEst² = (1/5) Σ (e_i/s_i)² through ONE reciprocal of Π s_i */
    double p2 = sc[0] * sc[1], p3 = p2 * sc[2], p4 = p3 * sc[3], pp = p4 * sc[4];
    double q2 = sc[3] * sc[4], q1 = sc[2] * q2, q0 = sc[1] * q1;
    double n0 = at[0] * q0, n1 = (at[1] * sc[0]) * q1, n2 = (at[2] * p2) * q2;
    double n3 = (at[3] * p3) * sc[4], n4 = at[4] * p4;
    double S = n0 * n0;
    S = PO_FMA(n1, n1, S);
    S = PO_FMA(n2, n2, S);
    S = PO_FMA(n3, n3, S);
    S = PO_FMA(n4, n4, S);
    double rp = 1.0 / pp;
    return (S * 0.2) * (rp * rp);
#undef A21
#undef A31
#undef A32
#undef A41
#undef A42
#undef A43
#undef A51
#undef A52
#undef A53
#undef A54
#undef A61
#undef A62
#undef A63
#undef A64
#undef A65
#undef A71
#undef A72
#undef A73
#undef A74
#undef A75
#undef A76
#undef C2
#undef C3
#undef C4
#undef C5
#undef E1
#undef E2
#undef E3
#undef E4
#undef E5
#undef E6
#undef E7
}

/* ------------------------------------------------------------------------------------------
 * solver 2: AutoTsit5(Rosenbrock23()) — the reference's DEFAULT (particle_waves_v5.jl:47).
 * Third-party semantics restated from OrdinaryDiffEq.jl v6 (unpinned, cannot be run here — "parity unpinned"):
 *   Tsit5 while the problem looks non-stiff; after every step (accepted or not) the AutoSwitch tests
 *   |eigen_est * dt_next / 3.5068| > 0.9, with eigen_est = ||k7-k6||/||u-g6|| under Tsit5 and ||J||_inf under
 *   Rosenbrock23; more than 10 successive positives switch to Rosenbrock23 (dt *= 2), more than 3 successive
 *   negatives switch back (dt /= 2).  The switch state survives set_u!/auto_dt_reset! (remesh) and is reset by
 *   reinit! (re-seed).  One PI controller with the order-5 betas (7/50, 2/25) serves both methods.
 * Rosenbrock23 (Shampine's ode23s as OrdinaryDiffEq writes it), d = 1/(2+sqrt 2), e32 = 6+sqrt 2, g = h d:
 *   W = I - g J;  k1 = W^-1 (f0 + g dT);  f1 = f(u0 + h/2 k1, t + h/2);  k2 = W^-1 (f1 - k1) + k1;  u = u0 + h k2;
 *   f2 = f(u, t + h);  k3 = W^-1 (f2 - e32 (k2 - f1) - 2 (k1 - f0) + h dT);  err = h/6 (k1 - 2 k2 + k3).
 * J = df/du is the exact Jacobian (the reference differentiates with ForwardDiff), written out by hand below for
 * the kernel-order RHS; dT = df/dt comes from the linear-in-time node wind.  x and y do not feed back, so W is
 * solved as a 3x3 system (adjugate) plus two substitutions.
 * ---------------------------------------------------------------------------------------- */
#define ROS_D 0.29289321881345254   /* 1/(2+sqrt 2) */
#define ROS_E32 7.414213562373095   /* 6+sqrt 2 */
#define ASW_STABILITY 3.5068        /* alg_stability_size(Tsit5()) */

/* f (3 components) plus the directional derivatives of f along ns seed directions (dL, dcx, dcy, du, dv) */
static void po_rhs_jvp(const po_model *M, int64_t idx, const double z[5], double u, double v,
                       int ns, const double seeds[][5], double f[3], double df[][3])
{
    const picles_phys *ph = &M->ph;
    const po_consts *k = &M->k;
    const double lne = z[0], cx = z[1], cy = z[2];
    const double pc = M->pc ? M->pc[idx] : 0.0;
    /* primal, kernel order (po_rhs_kernel) */
    double c2 = PO_FMA(cx, cx, cy * cy);
    double U2 = PO_FMA(u, u, v * v);
    double y = o_rsqrt(c2);
    double rc = ph->r_g * y;
    double ic2 = y * y;
    double minv = fmin(rc, 10.0);
    double wp = (0.5 * G0) * minv;
    double kp = (0.25 * G0) * (minv * minv);
    double rc2 = rc * rc;
    double qU2 = 0.25 * U2, invU2 = 1.0 / U2;
    double a2 = qU2 * rc2;
    double alpha2 = fmin(a2, 250000.0);
    double dotc = PO_FMA(u, cx, v * cy);
    double crsc = u * cy - v * cx;
    double sginv2 = fmin(rc2, 1e8);
    double hh = 0.5 * k->inv_rg;
    double ap = (hh * dotc) * sginv2;
    double ya = ap - 0.85;
    double neg2p = -2.0 * k->p;
    /* H_β, Δ_β and gH = dH/dya = 2p H (1 - H) (physics.h rhs3_jvp) */
    double hp, t, t1, t12, rHD, H, gH;
    if (k->p == 0.75) {
        double w = o_exp(-0.5 * fabs(ya));
        double w2 = w * w, s3 = w2 * w;
        double w4 = w2 * w2, w5 = w4 * w, w10 = w5 * w5, w20 = w10 * w10;
        t = w20 * w20;
        hp = 1.0 + s3;
        t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = 1.0 / (hp * t12);
        H = (t12 * rHD) * ((ya >= 0.0) ? 1.0 : s3);
        gH = (-neg2p) * (H * (1.0 - H));
    } else {
        double harg = neg2p * ya;
        double eH = o_exp((harg > 700.0) ? 700.0 : harg);
        hp = 1.0 + eH;
        double targ = -20.0 * fabs(ya);
        t = (targ <= -40.0) ? 0.0 : o_exp(targ);
        t1 = 1.0 + t;
        t12 = t1 * t1;
        rHD = 1.0 / (hp * t12);
        H = t12 * rHD;
        gH = (harg > 700.0) ? 0.0 : -((H * H) * (eH * neg2p));
    }
    double D = PO_FMA(-((5.0 * t) * hp), rHD, 1.0);
    const double r13 = 1.0 / (t12 * t1);
    const double rminv = 1.0 / minv;
    double aH = alpha2 * H;
    int n_is_2 = (k->n == 2.0);
    double E2 = o_exp(2.0 * lne);
    double k2 = kp * kp, k4 = k2 * k2;
    double Ek = E2 * k4;
    double ie2 = k->inv_eT * k->inv_eT, ieT4 = ie2 * ie2;
    double It = ph->input ? ph->C_e * aH : 0.0;
    double Dt = (ph->dissipation && n_is_2) ? Ek * ieT4 : 0.0;
    if (ph->dissipation && !n_is_2) Dt = o_exp(k->n * lne) * o_pow(kp * k->inv_eT, 2.0 * k->n);
    double Scg = ph->peak_shift ? (ph->C_alpha * D) * Ek : 0.0;
    int calm = (U2 == 0.0 || c2 == 0.0);
    double K2 = 2.0 * (k->inv_rg * k->inv_rg);
    double rU = rc2 * invU2;
    double cd = (K2 * crsc) * dotc;
    double s2 = calm ? 0.0 : cd * rU;
    double db2 = ph->dir_deadband * ph->dir_deadband;
    int dead = (db2 > 0.0 && crsc * crsc <= db2 * (U2 * c2));
    if (dead) s2 = 0.0;
    double Sd = ph->direction ? (ph->C_phi * aH) * s2 : 0.0;
    double wrS = (wp * ph->r_g) * Scg;
    double Sdm = Sd + (M->pc ? cx * pc : 0.0);
    {   /* the primal handed back is the RHS itself; the intermediates above (the association the kernel's rhs3_jvp uses, an
         * ulp away from po_rhs_kernel's in places) only feed the tangents */
        double dz[5];
        po_rhs_kernel(M, idx, z, u, v, dz);
        f[0] = dz[0]; f[1] = dz[1]; f[2] = dz[2];
    }
    /* tangents */
    for (int q = 0; q < ns; q++) {
        const double dL = seeds[q][0], dcx = seeds[q][1], dcy = seeds[q][2], du = seeds[q][3], dv = seeds[q][4];
        double dc2 = 2.0 * PO_FMA(cx, dcx, cy * dcy);
        double drc = -0.5 * ((rc * ic2) * dc2);
        double dminv = (rc <= 10.0) ? drc : 0.0;
        double dwp = (0.5 * G0) * dminv;
        double drc2 = 2.0 * (rc * drc);
        double dU2 = 2.0 * PO_FMA(u, du, v * dv);
        double dqU2 = 0.25 * dU2;
        double dinvU2 = -((invU2 * invU2) * dU2);
        double dalpha2 = (a2 <= 250000.0) ? PO_FMA(qU2, drc2, rc2 * dqU2) : 0.0;
        double ddot = PO_FMA(u, dcx, v * dcy) + PO_FMA(cx, du, cy * dv);
        double dcrs = (u * dcy - v * dcx) + (cy * du - cx * dv);
        double dsg = (rc2 <= 1e8) ? drc2 : 0.0;
        double dya = hh * PO_FMA(ddot, sginv2, dotc * dsg);
        double sgn = (ya < 0.0) ? 20.0 : -20.0;
        double dt_ = (t * sgn) * dya;
        double dH = gH * dya;
        double dD = -5.0 * ((dt_ * (1.0 - t)) * r13);
        double daH = PO_FMA(dalpha2, H, alpha2 * dH);
        double dlk = dminv * rminv;
        double dEk = (ph->dissipation && n_is_2) || ph->peak_shift ? Ek * PO_FMA(2.0, dL, 8.0 * dlk) : 0.0;
        double dIt = ph->input ? ph->C_e * daH : 0.0;
        double dDt = (ph->dissipation && n_is_2) ? dEk * ieT4 : 0.0;
        if (ph->dissipation && !n_is_2) dDt = Dt * PO_FMA(k->n, dL, (4.0 * k->n) * dlk);
        double dScg = ph->peak_shift ? ph->C_alpha * PO_FMA(dD, Ek, D * dEk) : 0.0;
        double ds2 = 0.0;
        if (!calm && !dead) {
            double dcd = K2 * PO_FMA(dcrs, dotc, crsc * ddot);
            double drU = PO_FMA(drc2, invU2, rc2 * dinvU2);
            ds2 = PO_FMA(dcd, rU, cd * drU);
        }
        double dSd = ph->direction ? ph->C_phi * PO_FMA(daH, s2, aH * ds2) : 0.0;
        if (M->pc) dSd = dSd + dcx * pc;
        double dwrS = ph->r_g * PO_FMA(dwp, Scg, wp * dScg);
        df[q][0] = PO_FMA(dwp, It - Dt, wp * (dIt - dDt)) + dwrS;
        df[q][1] = PO_FMA(dcy, Sdm, cy * dSd) - PO_FMA(dcx, wrS, cx * dwrS);
        df[q][2] = -(PO_FMA(dcx, Sdm, cx * dSd) + PO_FMA(dcy, wrS, cy * dwrS));
    }
}

/* the specialised physics of the kernels' FAST flavours: every switch on, n = 2, 2p = 3/2, no dead band (picles_hip.hip) */
static int po_is_fast(const po_model *M)
{
    const picles_phys *ph = &M->ph;
    return ph->propagation && ph->input && ph->dissipation && ph->peak_shift && ph->direction && M->k.n == 2.0 && M->k.p == 0.75 &&
           ph->dir_deadband * ph->dir_deadband == 0.0;
}

/* is the particle PLAIN (po_rhs_kernel)?  y <= ymax under a wind with 1e-290 <= U² and (U²/4) r_g² <= qU2r_max */
static int po_is_plain(const po_model *M, const double z[5], double u, double v)
{
    const double r_g = M->ph.r_g, rg2 = r_g * r_g, ymax = 10.0 / r_g;
    const double U2 = PO_FMA(u, u, v * v), qU2r = (0.25 * U2) * rg2;
    const double y = o_rsqrt(PO_FMA(z[1], z[1], z[2] * z[2]));
    return (U2 >= 1e-290 && qU2r <= 249999.0 / (ymax * ymax)) && (y <= ymax);
}

/* The Jacobian of the specialised-physics RHS for a plain particle, along the structure of the RHS (physics.h rhs3_jac_plain, same
 * operations in the same order): f = (F, c̄y Sd - c̄x S, -(c̄x Sd + c̄y S)) with F, S = ω_p r_g S_cg, Sd = S_dir functions of
 * (ln e, c² = |c̄|², c̄·u, c̄×u); nine partials and the chain rule.  J[r][c] = ∂f_r/∂u_c; dT = ∂f/∂t through the wind's slope (du, dv). */
static void po_rhs_jac_plain(const po_model *M, int64_t idx, const double z[5], double u, double v, double du, double dv,
                             int tvar, double J[3][3], double dT[3])
{
    const picles_phys *ph = &M->ph;
    const po_consts *k = &M->k;
    const double L = z[0], cx = z[1], cy = z[2];
    const double r_g = ph->r_g;
    const double g4 = 0.25 * G0, g42 = g4 * g4, K = g42 * g42;
    const double ie2 = k->inv_eT * k->inv_eT, inv_eT4 = ie2 * ie2;
    const double rg2 = r_g * r_g, rg4 = rg2 * rg2, rg8 = rg4 * rg4;
    const double Cw = (0.5 * G0) * r_g, Chrh = -0.25 * r_g;
    const double KeT4y = (K * inv_eT4) * rg8, KrCay = ((K * r_g) * ph->C_alpha) * rg8;
    const double Cs = (0.5 * ph->C_phi) * rg2;
    const double U2 = PO_FMA(u, u, v * v);
    const double qU2r = (0.25 * U2) * rg2;
    const double c2 = PO_FMA(cx, cx, cy * cy);
    const double y = o_rsqrt(c2);
    const double y2 = y * y;
    const double dotc = PO_FMA(u, cx, v * cy);
    const double crsc = u * cy - v * cx;
    const double wp = Cw * y;
    const double aph = (Chrh * dotc) * y2;
    const double m4 = y2 * y2;
    const double yh = aph + 0.425;
    const double w = o_exp(-fabs(yh));
    const double w2 = w * w, s3 = w2 * w;
    const double w5 = s3 * w2, w10 = w5 * w5, w20 = w10 * w10;
    const double t = w20 * w20;
    const double hp = 1.0 + s3, t1 = 1.0 + t;
    const double t12 = t1 * t1;
    const double rHD = 1.0 / (hp * t12);
    const double H = (t12 * rHD) * ((yh <= 0.0) ? 1.0 : s3);
    const double D = PO_FMA(-((5.0 * t) * hp), rHD, 1.0);
    const double DH = -3.0 * (H * (1.0 - H));
    const double iq = hp * rHD;
    const double DD = copysign(200.0 * ((t * (1.0 - t)) * (t1 * (iq * iq))), yh);
    const double yh_dot = Chrh * y2, yh_c2 = -(aph * y2);
    const double H_dot = DH * yh_dot, H_c2 = DH * yh_c2, D_dot = DD * yh_dot, D_c2 = DD * yh_c2;
    const double alpha2 = qU2r * y2;
    const double aH = alpha2 * H;
    const double aH_dot = alpha2 * H_dot;
    const double aH_c2 = alpha2 * PO_FMA(-y2, H, H_c2);
    const double Ek8 = o_exp_sat(L + L) * (m4 * m4);
    const double Dt = Ek8 * KeT4y;
    const double IDt = PO_FMA(ph->C_e, aH, -Dt);
    const double Q = wp * (Ek8 * KrCay);
    const double S = Q * D;
    const double wpDt = wp * Dt, wpCe = wp * ph->C_e;
    const double F_L = 2.0 * (S - wpDt);
    const double S_dot = Q * D_dot;
    const double F_dot = PO_FMA(wpCe, aH_dot, S_dot);
    const double S_c2 = PO_FMA(Q, D_c2, -4.5 * (y2 * S));
    const double F_c2 = PO_FMA(-0.5 * y2, wp * IDt, PO_FMA(4.0 * y2, wpDt, PO_FMA(wpCe, aH_c2, S_c2)));
    const double Ba = Cs * m4;
    const double B = Ba * H;
    const double Pcd = crsc * dotc;
    const double Sd = Pcd * B;
    const double PBa = Pcd * Ba;
    const double Sd_crs = dotc * B;
    const double Sd_dot = PO_FMA(PBa, H_dot, crsc * B);
    const double Sd_c2 = PBa * PO_FMA(-2.0 * y2, H, H_c2);
    const double ax = cx + cx, ay = cy + cy;
    const double F_x = PO_FMA(ax, F_c2, u * F_dot), F_y = PO_FMA(ay, F_c2, v * F_dot);
    const double S_x = PO_FMA(ax, S_c2, u * S_dot), S_y = PO_FMA(ay, S_c2, v * S_dot);
    double Sd_x = PO_FMA(ax, Sd_c2, PO_FMA(u, Sd_dot, -(v * Sd_crs)));
    const double Sd_y = PO_FMA(ay, Sd_c2, PO_FMA(v, Sd_dot, u * Sd_crs));
    double Sdm = Sd;
    if (M->pc) { Sdm = Sd + cx * M->pc[idx]; Sd_x = Sd_x + M->pc[idx]; }
    const double S_L = S + S;
    J[0][0] = F_L; J[0][1] = F_x; J[0][2] = F_y;
    J[1][0] = -(cx * S_L);
    J[1][1] = PO_FMA(cy, Sd_x, -PO_FMA(cx, S_x, S));
    J[1][2] = PO_FMA(cy, Sd_y, Sdm) - cx * S_y;
    J[2][0] = -(cy * S_L);
    J[2][1] = -(PO_FMA(cx, Sd_x, Sdm) + cy * S_x);
    J[2][2] = -PO_FMA(cx, Sd_y, PO_FMA(cy, S_y, S));
    dT[0] = dT[1] = dT[2] = 0.0;
    if (tvar) {
        const double dot_t = PO_FMA(cx, du, cy * dv), crs_t = PO_FMA(cy, du, -(cx * dv));
        const double U2_t = 2.0 * PO_FMA(u, du, v * dv);
        const double F_U2 = wpCe * ((0.25 * rg2) * (y2 * H));
        const double F_t = PO_FMA(F_dot, dot_t, F_U2 * U2_t);
        const double S_t = S_dot * dot_t;
        const double Sd_t = PO_FMA(Sd_dot, dot_t, Sd_crs * crs_t);
        dT[0] = F_t;
        dT[1] = PO_FMA(cy, Sd_t, -(cx * S_t));
        dT[2] = -PO_FMA(cx, Sd_t, cy * S_t);
    }
}

/* one attempted Rosenbrock23 step of size h from (u0, f0) at absolute time t; returns EEst^2 (kernel-order norm) and
 * the stiffness estimate ||J||_inf */
static double po_ros23_try(const po_model *M, int64_t idx, const double u0[5], const double f0[5],
                           double t, double h, double unew[5], double f2[5], po_pstats *st, double *eigen_est)
{
    const picles_ode *od = &M->od;
    const double ipx = M->ph.propagation ? (M->m11 ? M->m11[idx] : M->k.inv_dx) : 0.0;
    const double ipy = M->ph.propagation ? (M->m22 ? M->m22[idx] : M->k.inv_dy) : 0.0;
    double uw, vw, uw1, vw1;
    po_wind(M, idx, t, &uw, &vw);
    double dudt = 0.0, dvdt = 0.0;
    po_wind_dt(M, idx, t, &dudt, &dvdt);
    const int ns = M->wind_static ? 3 : 4;        /* static winds: dT = 0, its terms are not formed at all */
    double J[3][3], dT[3] = {0.0, 0.0, 0.0};      /* J[r][c] = d f_r / d u_c */
    if (M->order == 1 && po_is_fast(M) && po_is_plain(M, u0, uw, vw)) {
        /* kernel order, specialised physics, plain particle: the structured Jacobian */
        po_rhs_jac_plain(M, idx, u0, uw, vw, dudt, dvdt, !M->wind_static, J, dT);
    } else {
        const double seeds[4][5] = {{1, 0, 0, 0, 0}, {0, 1, 0, 0, 0}, {0, 0, 1, 0, 0}, {0, 0, 0, dudt, dvdt}};
        double fj[3], dfs[4][3] = {{0}};
        po_rhs_jvp(M, idx, u0, uw, vw, ns, seeds, fj, dfs);
        for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) J[r][c] = dfs[c][r];
        dT[0] = dfs[3][0]; dT[1] = dfs[3][1]; dT[2] = dfs[3][2];
    }
    st->rhs += ns;    /* counted like RHS evaluations (the reference's ForwardDiff pass costs about as much) */
    {   /* ||J||_inf over the 5 rows (rows x, y hold 1/dx, 1/dy) */
        double m = 0.0;
        for (int r = 0; r < 3; r++) {
            double rs = fabs(J[r][0]) + fabs(J[r][1]) + fabs(J[r][2]);
            m = fmax(m, rs);
        }
        m = fmax(m, fabs(ipx));
        m = fmax(m, fabs(ipy));
        *eigen_est = m;
    }
    const double g = h * ROS_D;
    /* W3 = I - g J3 and its inverse (adjugate / determinant) */
    double W[3][3], A[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) W[r][c] = ((r == c) ? 1.0 : 0.0) - g * J[r][c];
    A[0][0] = W[1][1] * W[2][2] - W[1][2] * W[2][1];
    A[0][1] = W[0][2] * W[2][1] - W[0][1] * W[2][2];
    A[0][2] = W[0][1] * W[1][2] - W[0][2] * W[1][1];
    A[1][0] = W[1][2] * W[2][0] - W[1][0] * W[2][2];
    A[1][1] = W[0][0] * W[2][2] - W[0][2] * W[2][0];
    A[1][2] = W[0][2] * W[1][0] - W[0][0] * W[1][2];
    A[2][0] = W[1][0] * W[2][1] - W[1][1] * W[2][0];
    A[2][1] = W[0][1] * W[2][0] - W[0][0] * W[2][1];
    A[2][2] = W[0][0] * W[1][1] - W[0][1] * W[1][0];
    double det = PO_FMA(W[0][0], A[0][0], PO_FMA(W[0][1], A[1][0], W[0][2] * A[2][0]));
    double idet = 1.0 / det;
#define WSOLVE(b, out) do { \
        double q0 = PO_FMA(A[0][0], (b)[0], PO_FMA(A[0][1], (b)[1], A[0][2] * (b)[2])) * idet; \
        double q1 = PO_FMA(A[1][0], (b)[0], PO_FMA(A[1][1], (b)[1], A[1][2] * (b)[2])) * idet; \
        double q2 = PO_FMA(A[2][0], (b)[0], PO_FMA(A[2][1], (b)[1], A[2][2] * (b)[2])) * idet; \
        (out)[0] = q0; (out)[1] = q1; (out)[2] = q2; \
        (out)[3] = PO_FMA(g * ipx, q1, (b)[3]); (out)[4] = PO_FMA(g * ipy, q2, (b)[4]); } while (0)
    double b[5], k1[5], k2[5], k3[5], f1[5], us[5];
    for (int i = 0; i < 3; i++) b[i] = M->wind_static ? f0[i] : PO_FMA(g, dT[i], f0[i]);
    b[3] = f0[3]; b[4] = f0[4];
    WSOLVE(b, k1);
    const double h2 = 0.5 * h;
    for (int i = 0; i < 5; i++) us[i] = PO_FMA(h2, k1[i], u0[i]);
    po_wind(M, idx, t + h2, &uw1, &vw1);
    po_rhs(M, idx, us, uw1, vw1, f1);
    for (int i = 0; i < 5; i++) b[i] = f1[i] - k1[i];
    WSOLVE(b, k2);
    for (int i = 0; i < 5; i++) k2[i] = k2[i] + k1[i];
    for (int i = 0; i < 5; i++) unew[i] = PO_FMA(h, k2[i], u0[i]);
    po_wind(M, idx, t + h, &uw1, &vw1);
    po_rhs(M, idx, unew, uw1, vw1, f2);
    st->rhs += 2;
    for (int i = 0; i < 5; i++) {
        b[i] = (f2[i] - ROS_E32 * (k2[i] - f1[i])) - 2.0 * (k1[i] - f0[i]);
        if (i < 3 && !M->wind_static) b[i] = b[i] + h * dT[i];
    }
    WSOLVE(b, k3);
#undef WSOLVE
    double at[5], sc[5];
    const double h6 = h * (1.0 / 6.0);
    for (int i = 0; i < 5; i++) {
        at[i] = h6 * ((k1[i] - 2.0 * k2[i]) + k3[i]);
        sc[i] = PO_FMA(fmax(fabs(u0[i]), fabs(unew[i])), od->reltol, od->abstol);
    }
    double p2 = sc[0] * sc[1], p3 = p2 * sc[2], p4 = p3 * sc[3], pp = p4 * sc[4];
    double q2 = sc[3] * sc[4], q1 = sc[2] * q2, q0 = sc[1] * q1;
    double n0 = at[0] * q0, n1 = (at[1] * sc[0]) * q1, n2 = (at[2] * p2) * q2;
    double n3 = (at[3] * p3) * sc[4], n4 = at[4] * p4;
    double S = n0 * n0;
    S = PO_FMA(n1, n1, S);
    S = PO_FMA(n2, n2, S);
    S = PO_FMA(n3, n3, S);
    S = PO_FMA(n4, n4, S);
    double rp = 1.0 / pp;
    return (S * 0.2) * (rp * rp);
}

/* step!(integrator, DT, true) for solver 2 (AutoTsit5(Rosenbrock23())): PI controller in log space as in the kernel
 * order; *asw carries the AutoSwitch state across model steps: bit 0 = Rosenbrock23 active, the rest = the signed
 * counter of successive stiffness-test outcomes, bit 30 of the counter part unused; *asw == INT32_MIN = "fresh". */
#define ASW_FRESH INT32_MIN
static void po_integrate_auto(const po_model *M, int64_t idx, double z[5], double *qold, double *dtn, int32_t *asw,
                              double t_start, double DT, po_pstats *st)
{
    const picles_ode *od = &M->od;
    double k1[5], kn[5], unew[5], uw, vw;
    double tr = 0.0;
    po_wind(M, idx, t_start, &uw, &vw);
    po_rhs(M, idx, z, uw, vw, k1);
    st->rhs++;
    int fresh = (*asw == ASW_FRESH);
    int stiff = fresh ? 0 : (*asw & 1);
    int count = fresh ? 0 : (*asw >> 1);
    double dt = *dtn;
    if (!(dt > 0.0)) {
        dt = po_initdt(M, idx, z, k1, t_start, st);
    }
    const double beta1 = 0.14, beta2 = 0.08;
    double eig[2] = {0.0, 1.0};      /* eigen_est² = eig[0] / eig[1] */
    int have_eig = 0;
    int64_t iter = 0;
    while (tr < DT) {
        iter++;
        if (iter > od->maxiters) { st->status |= PICLES_ST_MAXITERS; break; }
        /* choose_algorithm! (loopheader!): the test uses the estimate of the previous attempt and the proposed dt */
        if (fresh) {
            fresh = 0;                 /* first call: current = nonstiff, no test */
        } else if (have_eig) {
            /* |eigen_est dt / 3.5068| > 0.9  <=>  nu dt² > (0.9·3.5068)² nd  (a NaN on either side: not stiff) */
            int pos = eig[0] * (dt * dt) > ((0.9 * ASW_STABILITY) * (0.9 * ASW_STABILITY)) * eig[1];
            count = pos ? (count < 0 ? 1 : count + 1) : (count > 0 ? -1 : count - 1);
            if (!stiff && count > 10) { dt = dt * 2.0; stiff = 1; }
            else if (stiff && count < -3) { dt = dt * 0.5; stiff = 0; }
        }
        if (dt < od->dtmin) dt = od->dtmin;
        double rem = DT - tr;
        double h = (dt < rem) ? dt : rem;
        int last = !(dt < rem);
        double EE2;
        if (stiff) {
            double nj;
            EE2 = po_ros23_try(M, idx, z, k1, t_start + tr, h, unew, kn, st, &nj);
            eig[0] = nj * nj; eig[1] = 1.0;        /* ||J||_inf */
        } else {
            EE2 = po_dp5_try_e(M, idx, z, k1, t_start + tr, h, unew, kn, st, eig);
        }
        have_eig = 1;
        if (!M->order && !stiff) EE2 = EE2 * EE2;     /* the literal-order explicit pair returns EEst; Rosenbrock returns EEst² */
        if (!(EE2 == EE2)) { EE2 = INFINITY; st->status |= PICLES_ST_NONFINITE; }
        int accept = (EE2 <= 1.0) || (od->force_dtmin && h <= od->dtmin);
        double le = 0.5 * o_log_coarse(EE2);
        if (accept) {
            st->acc++;
            double qi = o_exp(PO_FMA(beta2, *qold, -(beta1 * le))) * CTRL_GAMMA;
            qi = fmax(fmin(qi, CTRL_QMAX), CTRL_QMIN);
            *qold = fmax(le, CTRL_LNQOLDINIT);
            dt = h * qi;
            for (int i = 0; i < 5; i++) { z[i] = unew[i]; k1[i] = kn[i]; }
            tr = last ? DT : tr + h;
            if (z[0] != z[0] || z[1] != z[1] || z[2] != z[2] || z[3] != z[3] || z[4] != z[4]) break;
        } else {
            st->rej++;
            double r = CTRL_GAMMA * o_exp(-(beta1 * le));
            r = fmax(r, CTRL_QMIN);
            dt = h * r;
            if (!od->force_dtmin && h <= od->dtmin) { st->status |= PICLES_ST_DTMIN; break; }
        }
    }
    *dtn = dt;
    *asw = (int32_t)((uint32_t)count << 1) | stiff;
}

/* step!(integrator, DT, true): integrate particle idx from clock to clock+DT. */
static void po_integrate(const po_model *M, int64_t idx, double z[5], double *qold, double *dtn,
                         double t_start, double DT, po_pstats *st)
{
    const picles_ode *od = &M->od;
    double k1[5], k7[5], unew[5], uw, vw;
    double tr = 0.0; /* time since t_start; absolute time = t_start + tr */
    po_wind(M, idx, t_start, &uw, &vw);
    po_rhs(M, idx, z, uw, vw, k1);
    st->rhs++;
    double dt = *dtn;
    if (!(dt > 0.0)) dt = po_initdt(M, idx, z, k1, t_start, st);
    /* beta2_default / beta1_default (OrdinaryDiffEq alg_utils.jl): DP5 4//100, 1//5 - 3beta2/4; Tsit5 2//25, 7//50 */
    const double beta1 = od->solver ? 0.14 : CTRL_BETA1, beta2 = od->solver ? 0.08 : CTRL_BETA2;
    int64_t iter = 0;
    while (tr < DT) {
        iter++;
        if (iter > od->maxiters) { st->status |= PICLES_ST_MAXITERS; break; }
        /* fix_dt_at_bounds!: dt >= max(eps(t), dtmin);  modify_dt_for_tstops!: dt <= tstop - t */
        if (dt < od->dtmin) dt = od->dtmin;
        double rem = DT - tr;
        double h = (dt < rem) ? dt : rem;
        int last = !(dt < rem);
        /* order 0: EEst; order 1: EEst² (the kernel order never takes the square root) */
        double EEst = po_dp5_try(M, idx, z, k1, t_start + tr, h, unew, k7, st);
        if (!(EEst == EEst)) { EEst = INFINITY; st->status |= PICLES_ST_NONFINITE; }
        int accept = (EEst <= 1.0) || (od->force_dtmin && h <= od->dtmin);
        if (M->order == 1) {
            /* kernel order: PI controller in log space, *qold holds ln(qold) */
            double le = 0.5 * o_log_coarse(EEst);   /* ~1e-9: the controller needs no more (pmath.h) */
            if (accept) {
                st->acc++;
                double qi = o_exp(PO_FMA(beta2, *qold, -(beta1 * le))) * CTRL_GAMMA;
                qi = fmax(fmin(qi, CTRL_QMAX), CTRL_QMIN);
                *qold = fmax(le, CTRL_LNQOLDINIT);
                dt = h * qi;
                for (int i = 0; i < 5; i++) { z[i] = unew[i]; k1[i] = k7[i]; }
                tr = last ? DT : tr + h;
                if (z[0] != z[0] || z[1] != z[1] || z[2] != z[2] || z[3] != z[3] || z[4] != z[4]) break;
            } else {
                st->rej++;
                double r = CTRL_GAMMA * o_exp(-(beta1 * le));
                r = fmax(r, CTRL_QMIN);
                dt = h * r;
                if (!od->force_dtmin && h <= od->dtmin) { st->status |= PICLES_ST_DTMIN; break; }
            }
            continue;
        }
        /* stepsize_controller!(PIController) */
        double q11 = 0.0, q;
        if (EEst == 0.0) {
            q = 1.0 / CTRL_QMAX;
        } else {
            q11 = o_pow(EEst, beta1);
            q = q11 / o_pow(*qold, beta2);
            double qg = q / CTRL_GAMMA;
            double lo = 1.0 / CTRL_QMAX, hi = 1.0 / CTRL_QMIN;
            q = (qg < hi) ? qg : hi;
            q = (q > lo) ? q : lo;
        }
        if (accept) {
            st->acc++;
            *qold = (EEst > CTRL_QOLDINIT) ? EEst : CTRL_QOLDINIT; /* step_accept_controller! */
            dt = h / q;
            for (int i = 0; i < 5; i++) { z[i] = unew[i]; k1[i] = k7[i]; }
            tr = last ? DT : tr + h;
            /* unstable_check: NaN in u stops the integration (retcode Unstable) */
            if (z[0] != z[0] || z[1] != z[1] || z[2] != z[2] || z[3] != z[3] || z[4] != z[4]) break;
        } else {
            st->rej++;
            double f = q11 / CTRL_GAMMA;          /* step_reject_controller! */
            double hi = 1.0 / CTRL_QMIN;
            dt = h / ((f < hi) ? f : hi);
            if (!od->force_dtmin && h <= od->dtmin) { st->status |= PICLES_ST_DTMIN; break; }
        }
    }
    *dtn = dt;
}

/* ------------------------------------------------------------------------------------------
 * ParticleInCell: get_absolute_i_and_w(z, i_node) (ParticleInCell.jl:58-71),
 * compute_weights_and_index_mininal (:149-157), construct_loop (:504-508),
 * push_to_grid! AbstractBoundary method (:341-376), wrap_index! (:444-454), test_domain (:464-466)
 * Indices here are 0-based; the drop / wrap tests are the reference's shifted by one.
 * ---------------------------------------------------------------------------------------- */
static inline void po_index_weight(double zp, int32_t i_node, int64_t idx[2], double w[2])
{
    double b = floor(zp);
    /* Julia's Int(floor(x)) throws beyond the Int64 range; a C conversion there is undefined (found with UBSan on the hostile
     * scenarios, where runaway particles reach 1e19 cells and more): saturate well inside the range, the wrap / drop rules
     * below treat the result like any other far-away cell */
    int64_t ib = (int64_t)((b > 9.0e18) ? 9.0e18 : (b < -9.0e18) ? -9.0e18 : b);
    double wc = rint((zp - b) * 1e6) / 1e6; /* round(·, digits=6) */
    idx[0] = ib + i_node;
    idx[1] = ib + i_node + 1;
    w[0] = 1.0 - wc;
    w[1] = wc;
}
static inline int64_t po_wrap(int64_t i, int64_t N)
{
    int64_t r = i % N;
    return (r < 0) ? r + N : r;
}
/* returns max |cell offset| of the 4 corners (the scatter "reach") */
static int po_particle_to_node(po_model *M, int32_t i, int32_t j, const double z[5])
{
    if (!(isfinite(z[3]) && isfinite(z[4]))) { M->cnt.dropped_nonfinite++; return 0; } /* Int(floor(NaN)) would throw in Julia */
    int64_t xi[2], yi[2];
    double xw[2], yw[2], c[3];
    po_index_weight(z[3], i, xi, xw);
    po_index_weight(z[4], j, yi, yw);
    po_particle_to_charge(M->order, z, c);
    static const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 0, 1, 1}; /* construct_loop order */
    for (int k = 0; k < 4; k++) {
        int64_t ii = xi[ox[k]], jj = yi[oy[k]];
        if (M->g.periodic_y == 2) {
            /* N_TripolarNorth (ParticleInCell.jl:353-361, TripolarNorthBoundary :409-428), 0-based: below the south
             * edge the corner is dropped; above the north fold it lands mirrored in x on row 2Ny-1-jj, charge unchanged */
            if (jj < 0) continue;
            if (jj >= M->Ny) {
                ii = M->Nx - 1 - po_wrap(ii + 1, M->Nx);
                jj = 2 * (int64_t)M->Ny - 1 - jj;
                if (jj < 0) continue;
            }
        } else if (!M->g.periodic_y && !(jj >= 0 && jj < M->Ny)) continue;
        if (!M->g.periodic_x && !(ii >= 0 && ii < M->Nx)) continue;
        ii = po_wrap(ii, M->Nx);
        jj = po_wrap(jj, M->Ny);
        double w = xw[ox[k]] * yw[oy[k]];
        int64_t n = ii + (int64_t)M->Nx * jj;
        M->state[n] += w * c[0];
        M->state[n + M->N] += w * c[1];
        M->state[n + 2 * M->N] += w * c[2];
    }
    int64_t r = 0, d;
    d = llabs(xi[0] - i); if (d > r) r = d;
    d = llabs(xi[1] - i); if (d > r) r = d;
    d = llabs(yi[0] - j); if (d > r) r = d;
    d = llabs(yi[1] - j); if (d > r) r = d;
    return (r > 0x7fffffff) ? 0x7fffffff : (int)r;
}

/* ------------------------------------------------------------------------------------------
 * advance!(PI, S, Failed, Grid, winds, DT, lne_max, wind_min², periodic, defaults)
 * mapping_2D.jl:118-243 — everything except the final ParticleToNode!, which the caller does
 * afterwards in ocean_points order (identical sums: advance! never reads State).
 * ---------------------------------------------------------------------------------------- */
static void po_advance_particle(po_model *M, int64_t idx, double DT, po_pstats *st)
{
    double z[5];
    for (int c = 0; c < 5; c++) z[c] = M->z[idx + c * M->N];
    double t_start = M->clock;
    int status = PICLES_ST_STEPPED;
    if (M->on[idx]) {
        po_pstats s = {0, 0, 0, 0};
        if (M->od.solver == 2) po_integrate_auto(M, idx, z, &M->qold[idx], &M->dtn[idx], &M->asw[idx], t_start, DT, &s);
        else po_integrate(M, idx, z, &M->qold[idx], &M->dtn[idx], t_start, DT, &s);
        st->rhs += s.rhs; st->acc += s.acc; st->rej += s.rej;
        status |= s.status;
    } else {
        double u, v;
        po_wind(M, idx, t_start + DT, &u, &v);
        if (u * u + v * v >= M->od.wind_min_squared) {       /* :172-185 */
            po_reseed(M, u, v, DT, z);
            M->dtn[idx] = -1.0;
            M->on[idx] = 1;
            status |= PICLES_ST_SWITCHED_ON;
        }
    }
    if (isnan(z[0]) || isnan(z[1]) || isnan(z[2])) {          /* :196-211 */
        double u, v;
        po_wind(M, idx, t_start + DT, &u, &v);
        po_reseed(M, u, v, DT, z);
        M->dtn[idx] = -1.0;
        status |= PICLES_ST_RESEED_NAN;
    } else if (isinf(z[0]) || isinf(z[1]) || isinf(z[2])) {   /* :213-222 */
        double u, v;
        po_wind(M, idx, t_start, &u, &v);
        po_reseed(M, u, v, DT, z);
        M->dtn[idx] = -1.0;
        status |= PICLES_ST_RESEED_INF;
    } else if (z[0] > M->od.log_energy_maximum) {             /* :224-235 */
        z[0] = M->od.log_energy_maximum;
        M->dtn[idx] = -1.0;
        status |= PICLES_ST_CLAMPED;
    }
    for (int c = 0; c < 5; c++) M->z[idx + c * M->N] = z[c];
    M->status[idx] = status;
}

/* NodeToParticle! (mapping_2D.jl:279-356) via remesh! (:250-269) */
static void po_remesh_particle(po_model *M, int64_t idx, double DT)
{
    double c[3] = {M->state[idx], M->state[idx + M->N], M->state[idx + 2 * M->N]};
    double u, v;
    po_wind(M, idx, M->clock, &u, &v);          /* winds at model.clock.time, before tick! */
    int bnd = M->bnd[idx];
    double z[5];
    if (!bnd && (c[0] >= M->md.minimal_state[0]) &&
        (c[1] * c[1] + c[2] * c[2] >= M->md.minimal_state[1])) {           /* A :306-312 */
        po_charge_to_particle(M->order, c, z);
        for (int k = 0; k < 5; k++) M->z[idx + k * M->N] = z[k];
        M->dtn[idx] = -1.0;
        M->on[idx] = 1;
    } else if (u * u + v * v >= M->od.wind_min_squared) {                   /* B :328-336, C :338-344 */
        po_reseed(M, u, v, DT, z);
        for (int k = 0; k < 5; k++) M->z[idx + k * M->N] = z[k];
        M->qold[idx] = PO_QOLD_RESET(M);   /* reinit! resets the controller */
        M->asw[idx] = INT32_MIN;           /* ... and the AutoSwitch state */
        M->dtn[idx] = -1.0;
        M->on[idx] = 1;
        __atomic_fetch_add(&M->cnt.reseeds, 1, __ATOMIC_RELAXED);
    } else {                                                                /* D :347-353 */
        M->on[idx] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * model: WaveGrowth2D constructor pieces (WaveGrowthModels2D.jl:194-345), mask classes
 * (mask_utils.jl:38-55), ocean_points (:256-270), check_boundary_point (core_2D.jl:360-366)
 * ---------------------------------------------------------------------------------------- */
static size_t po_rec_doubles(const po_model *M) { return (size_t)(M->nyl + 2 * M->R) * 6 * M->Nx; }

PO_EXPORT int32_t picles_oracle_create(const picles_grid *g, const picles_phys *p, const picles_ode *o,
                                       const picles_model *m, int32_t order, po_model **out)
{
    if (!g || !p || !o || !m || !out) return -1;
    if (g->Nx < 2 || g->Ny < 2) return -2;
    po_model *M = (po_model *)calloc(1, sizeof(po_model));
    M->g = *g; M->ph = *p; M->od = *o; M->md = *m;
    M->order = order;
    M->nthreads = 1;
    M->Nx = g->Nx; M->Ny = g->Ny;
    M->j0 = g->j_begin;
    M->nyl = (g->j_end > g->j_begin) ? g->j_end - g->j_begin : g->Ny;
    if (g->j_end <= g->j_begin) { M->j0 = 0; M->g.j_begin = 0; M->g.j_end = g->Ny; }
    M->single_slab = (M->j0 == 0 && M->nyl == M->Ny);
    M->R = 1;
    M->N = (int64_t)M->Nx * M->nyl;
    int64_t N = M->N;
    po_derive(p, g->dx, g->dy, &M->k);
    M->mask = (int8_t *)malloc(N);
    int any3 = 0;
    for (int j = 0; j < M->Ny; j++)
        for (int i = 0; i < M->Nx; i++) {
            int8_t mk;
            if (g->mask) mk = g->mask[i + (int64_t)M->Nx * j];
            else { /* make_boundaries(ones): grid-boundary ring on non-periodic axes */
                int ring = (!g->periodic_x && (i == 0 || i == M->Nx - 1)) ||
                           (!g->periodic_y && (j == 0 || j == M->Ny - 1));
                mk = ring ? 3 : 1;
            }
            if (mk == 3) any3 = 1;
            int jl = j - M->j0;
            if (jl >= 0 && jl < M->nyl) M->mask[i + (int64_t)M->Nx * jl] = mk;
        }
    M->g.mask = NULL;
    M->ngroups = (any3 && m->periodic_boundary) ? 2 : 1;
    M->state = (double *)calloc(3 * N, 8);
    M->movie = (double *)calloc(3 * N, 8);
    M->z = (double *)calloc(5 * N, 8);
    M->qold = (double *)calloc(N, 8);
    M->asw = (int32_t *)calloc(N, 4);
    M->dtn = (double *)calloc(N, 8);
    M->on = (uint8_t *)calloc(N, 1);
    M->bnd = (uint8_t *)calloc(N, 1);
    M->grp = (uint8_t *)calloc(N, 1);
    M->status = (int32_t *)calloc(N, 4);
    M->u0 = (double *)calloc(N, 8); M->v0 = (double *)calloc(N, 8);
    M->u1 = (double *)calloc(N, 8); M->v1 = (double *)calloc(N, 8);
    M->rec = (double *)calloc(po_rec_doubles(M), 8);
    M->wind_static = 1;
    /* ocean_points: findall(mask .== 1) [then findall(mask .== 3) if periodic_boundary], column-major */
    M->steplist = (int64_t *)malloc(N * 8);
    int64_t ns = 0;
    for (int64_t n = 0; n < N; n++) if (M->mask[n] == 1) { M->steplist[ns++] = n; M->grp[n] = 1; }
    if (m->periodic_boundary)
        for (int64_t n = 0; n < N; n++) if (M->mask[n] == 3) { M->steplist[ns++] = n; M->grp[n] = 2; }
    M->n_step = ns;
    for (int64_t n = 0; n < N; n++)
        M->bnd[n] = m->periodic_boundary ? (M->mask[n] == 2) : (M->mask[n] >= 2);
    *out = M;
    return 0;
}

static void po_polyline_free(po_model *M);
PO_EXPORT int32_t picles_oracle_destroy(po_model *M)
{
    if (!M) return 0;
    free(M->mask); free(M->state); free(M->movie); free(M->z); free(M->qold); free(M->asw); free(M->dtn); free(M->grp); free(M->rec);
    free(M->on); free(M->bnd); free(M->status); free(M->steplist);
    po_polyline_free(M);
    free(M->u0); free(M->v0); free(M->u1); free(M->v1); free(M->um); free(M->vm);
    free(M->m11); free(M->m22); free(M->pc);
    free(M);
    return 0;
}

/* per-node ProjetionKernel diagonal and PropagationCorrection coefficient (SphericalGrid.jl:207-240,
 * spherical_grid_corrections.jl:3-21); NULL restores the Cartesian constants */
PO_EXPORT int32_t picles_oracle_set_metric(po_model *M, const double *m11, const double *m22, const double *pc)
{
    free(M->m11); free(M->m22); free(M->pc);
    M->m11 = M->m22 = M->pc = NULL;
    if (m11 && m22 && pc) {
        M->m11 = (double *)malloc(M->N * 8); memcpy(M->m11, m11, M->N * 8);
        M->m22 = (double *)malloc(M->N * 8); memcpy(M->m22, m22, M->N * 8);
        M->pc = (double *)malloc(M->N * 8); memcpy(M->pc, pc, M->N * 8);
    }
    return 0;
}

PO_EXPORT int32_t picles_oracle_set_threads(po_model *M, int32_t n) { M->nthreads = n > 0 ? n : 1; return 0; }

static void po_polyline_free(po_model *M)
{
    for (int k = 0; k < PO_MAX_KNOTS; k++) { free(M->plu[k]); free(M->plv[k]); M->plu[k] = M->plv[k] = NULL; }
    for (int k = 0; k <= PO_MAX_KNOTS; k++) { free(M->pcu[k]); free(M->pcv[k]); M->pcu[k] = M->pcv[k] = NULL; }
    M->wind_nk = 0;
}

PO_EXPORT int32_t picles_oracle_set_winds3(po_model *M, const double *u0, const double *v0, double t0,
                                           const double *um, const double *vm,
                                           const double *u1, const double *v1, double t1)
{
    memcpy(M->u0, u0, M->N * 8);
    memcpy(M->v0, v0, M->N * 8);
    M->tw0 = t0;
    M->wind_knot = 0;
    po_polyline_free(M);
    free(M->um); free(M->vm);
    M->um = M->vm = NULL;
    if (u1 && v1 && t1 != t0) {
        memcpy(M->u1, u1, M->N * 8);
        memcpy(M->v1, v1, M->N * 8);
        M->tw1 = t1;
        M->wind_static = 0;
        if (um && vm) {
            M->um = (double *)malloc(M->N * 8); memcpy(M->um, um, M->N * 8);
            M->vm = (double *)malloc(M->N * 8); memcpy(M->vm, vm, M->N * 8);
        }
    } else {
        M->wind_static = 1;
        M->tw1 = t0;
    }
    return 0;
}
/* three levels, the middle one at a knot tk of a gridded wind, t0 < tk < t1 (the product's picles_set_winds_knot) */
PO_EXPORT int32_t picles_oracle_set_winds_knot(po_model *M, const double *u0, const double *v0, double t0,
                                               const double *uk, const double *vk, double tk,
                                               const double *u1, const double *v1, double t1)
{
    if (!uk || !vk || !u1 || !v1 || !(t0 < tk && tk < t1)) return -2;
    int32_t rc = picles_oracle_set_winds3(M, u0, v0, t0, uk, vk, u1, v1, t1);
    if (rc) return rc;
    M->wind_knot = 1;
    M->twk = tk;
    return 0;
}
PO_EXPORT int32_t picles_oracle_set_winds(po_model *M, const double *u0, const double *v0, double t0,
                                          const double *u1, const double *v1, double t1)
{
    return picles_oracle_set_winds3(M, u0, v0, t0, NULL, NULL, u1, v1, t1);
}
/* nlev >= 2 levels at strictly increasing times (the product's picles_set_winds_polyline): the reference's interpolant inside a step
 * that holds nlev - 2 time knots of the wind lattice */
PO_EXPORT int32_t picles_oracle_set_winds_polyline(po_model *M, int32_t nlev, const double *const *u, const double *const *v, const double *times)
{
    if (nlev < 2 || !u || !v || !times) return -2;
    for (int k = 1; k < nlev; k++) if (!(times[k - 1] < times[k])) return -2;
    if (nlev == 2) return picles_oracle_set_winds3(M, u[0], v[0], times[0], NULL, NULL, u[1], v[1], times[1]);
    if (nlev == 3) return picles_oracle_set_winds_knot(M, u[0], v[0], times[0], u[1], v[1], times[1], u[2], v[2], times[2]);
    const int nk = nlev - 2;
    if (nk > PO_MAX_KNOTS) return -2;
    int32_t rc = picles_oracle_set_winds_knot(M, u[0], v[0], times[0], u[1], v[1], times[1], u[nlev - 1], v[nlev - 1], times[nlev - 1]);
    if (rc) return rc;
    M->wind_nk = nk;
    for (int k = 0; k < nk; k++) {
        M->ptk[k] = times[k + 1];
        M->plu[k] = (double *)malloc(M->N * 8); memcpy(M->plu[k], u[k + 1], M->N * 8);
        M->plv[k] = (double *)malloc(M->N * 8); memcpy(M->plv[k], v[k + 1], M->N * 8);
    }
    /* what k_wind_poly lays down (picles_hip.hip): s of the knots, slopes per unit s of the segments, their jumps at the knots */
    const double idt = 1.0 / (M->tw1 - M->tw0);
    double ilen[PO_MAX_KNOTS + 1], sprev = 0.0;
    for (int k = 0; k <= nk; k++) {
        const double sn = (k < nk) ? (M->ptk[k] - M->tw0) * idt : 1.0;
        if (k < nk) M->psk[k] = sn;
        ilen[k] = 1.0 / (sn - sprev);
        sprev = sn;
    }
    for (int k = 0; k <= nk; k++) { M->pcu[k] = (double *)malloc(M->N * 8); M->pcv[k] = (double *)malloc(M->N * 8); }
    for (int64_t t = 0; t < M->N; t++) {
        for (int c = 0; c < 2; c++) {
            double prev = c ? M->v0[t] : M->u0[t], slp = 0.0;
            for (int j = 0; j <= nk; j++) {
                const double next = (j < nk) ? (c ? M->plv[j][t] : M->plu[j][t]) : (c ? M->v1[t] : M->u1[t]);
                const double sl = (next - prev) * ilen[j];
                (c ? M->pcv[j] : M->pcu[j])[t] = j ? sl - slp : sl;
                slp = sl;
                prev = next;
            }
        }
    }
    return 0;
}

/* init_particles! (run.jl:199-247) -> SeedParticle (core_2D.jl:434-488) -> InitParticleValues
 * (:247-288) -> init_z0_to_State! (initialize.jl:14-17) */
PO_EXPORT int32_t picles_oracle_seed(po_model *M, double t0)
{
    int64_t N = M->N;
    M->clock = t0;
    memset(M->rec, 0, po_rec_doubles(M) * 8);
    memset(M->state, 0, 3 * N * 8);
    memset(&M->cnt, 0, sizeof(M->cnt));
    for (int64_t n = 0; n < N; n++) {
        M->status[n] = 0;
        M->qold[n] = PO_QOLD_RESET(M);
        M->asw[n] = INT32_MIN;
        M->dtn[n] = M->od.dt0;
        for (int c = 0; c < 5; c++) M->z[n + c * N] = 0.0;
        M->on[n] = 0;
        if (M->mask[n] == 0) continue; /* land: dummy instance */
        double u, v, z[5];
        po_wind(M, n, 0.0 + (M->wind_static ? 0.0 : 0.0), &u, &v); /* winds at t = 0.0 (run.jl:213-215) */
        int on;
        if (M->md.init_type == 0) {
            if (sqrt(u * u + v * v) > sqrt(2.0)) {
                double s[3];
                po_windsea(u, v, M->od.timestep, s);
                z[0] = s[0]; z[1] = s[1]; z[2] = s[2];
                on = 1;
            } else {
                /* MinimalParticle(u,v,DT): MinimalWindsea with rand_sign -> +1 */
                double uu = (u == 0.0) ? 1.0 : u, vv = (v == 0.0) ? 1.0 : v;
                double am = sqrt(uu * uu + vv * vv), s[3];
                po_windsea(1.0 * uu / am, 1.0 * vv / am, M->od.timestep, s);
                z[0] = s[0]; z[1] = s[1]; z[2] = s[2];
                on = 0;
            }
        } else {
            z[0] = M->md.default_particle[0];
            z[1] = M->md.default_particle[1];
            z[2] = M->md.default_particle[2];
            on = 1;
        }
        z[3] = 0.0; z[4] = 0.0;
        for (int c = 0; c < 5; c++) M->z[n + c * N] = z[c];
        M->on[n] = (uint8_t)on;
        if (on) {
            double c3[3];
            po_particle_to_charge(M->order, z, c3);
            M->state[n] = c3[0];
            M->state[n + N] = c3[1];
            M->state[n + 2 * N] = c3[2];
        }
    }
    return 0;
}

/* time_step!_advance (TimeSteppers.jl:168-180) */
PO_EXPORT int32_t picles_oracle_advance(po_model *M, double DT)
{
    if (!M->single_slab) return -5; /* the sequential push needs the whole grid */
    uint64_t rhs = 0, acc = 0, rej = 0, adv = 0;
    int64_t ns = M->n_step;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(M->nthreads) reduction(+ : rhs, acc, rej, adv)
#endif
    for (int64_t s = 0; s < ns; s++) {
        int64_t idx = M->steplist[s];
        po_pstats st = {0, 0, 0, 0};
        int was_on = M->on[idx];
        po_advance_particle(M, idx, DT, &st);
        rhs += st.rhs; acc += st.acc; rej += st.rej; adv += was_on ? 1 : 0;
    }
    M->cnt.rhs_evals += rhs; M->cnt.steps_accepted += acc; M->cnt.steps_rejected += rej;
    M->cnt.particles_advanced += adv;
    /* ParticleToNode! in ocean_points order (the sequential sum order IS the definition) */
    int reach = 0;
    for (int64_t s = 0; s < ns; s++) {
        int64_t idx = M->steplist[s];
        int st = M->status[idx];
        if (st & (PICLES_ST_RESEED_NAN | PICLES_ST_RESEED_INF | PICLES_ST_SWITCHED_ON)) M->cnt.reseeds++;
        if (st & PICLES_ST_CLAMPED) M->cnt.clamps++;
        if (st & PICLES_ST_MAXITERS) M->cnt.maxiters_hits++;
        if (!M->on[idx]) continue;
        double z[5];
        for (int c = 0; c < 5; c++) z[c] = M->z[idx + c * M->N];
        int r = po_particle_to_node(M, (int32_t)(idx % M->Nx), (int32_t)(idx / M->Nx), z);
        if (r > reach) reach = r;
    }
    M->cnt.max_reach = reach;
    if (reach > M->cnt.max_reach_seen) M->cnt.max_reach_seen = reach;
    return 0;
}

/* ParticleToNode! alone, for the particles as they are (test hook for the boundary rules of the push) */
PO_EXPORT int32_t picles_oracle_scatter_only(po_model *M)
{
    if (!M->single_slab) return -5;
    for (int64_t s = 0; s < M->n_step; s++) {
        int64_t idx = M->steplist[s];
        if (!M->on[idx]) continue;
        double z[5];
        for (int c = 0; c < 5; c++) z[c] = M->z[idx + c * M->N];
        po_particle_to_node(M, (int32_t)(idx % M->Nx), (int32_t)(idx / M->Nx), z);
    }
    return 0;
}

/* time_step!_remesh (TimeSteppers.jl:182-193) */
PO_EXPORT int32_t picles_oracle_remesh(po_model *M, double DT)
{
    int64_t ns = M->n_step;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(M->nthreads)
#endif
    for (int64_t s = 0; s < ns; s++) po_remesh_particle(M, M->steplist[s], DT);
    return 0;
}

PO_EXPORT int32_t picles_oracle_zero_state(po_model *M) { memset(M->state, 0, 3 * M->N * 8); return 0; }
PO_EXPORT int32_t picles_oracle_tick(po_model *M, double dt) { M->clock += dt; return 0; }
PO_EXPORT double picles_oracle_clock(const po_model *M) { return M->clock; }

/* time_step! (TimeSteppers.jl:109-166) / movie_time_step! (:212-247); flags as picles_time_step */
PO_EXPORT int32_t picles_oracle_time_step(po_model *M, double dt, int32_t flags)
{
    if (flags & PICLES_STEP_ZERO_FIRST) picles_oracle_zero_state(M);
    picles_oracle_advance(M, dt);
    if (flags & PICLES_STEP_MOVIE) memcpy(M->movie, M->state, 3 * M->N * 8);
    picles_oracle_remesh(M, dt);
    if (flags & PICLES_STEP_MOVIE) picles_oracle_zero_state(M);
    M->clock += dt;
    return 0;
}

PO_EXPORT int32_t picles_oracle_get_state(po_model *M, double *s) { memcpy(s, M->state, 3 * M->N * 8); return 0; }
PO_EXPORT int32_t picles_oracle_set_state(po_model *M, const double *s) { memcpy(M->state, s, 3 * M->N * 8); return 0; }
PO_EXPORT int32_t picles_oracle_get_movie_state(po_model *M, double *s) { memcpy(s, M->movie, 3 * M->N * 8); return 0; }
PO_EXPORT int32_t picles_oracle_get_particles(po_model *M, double *z, uint8_t *on, uint8_t *bnd, int32_t *status)
{
    if (z) memcpy(z, M->z, 5 * M->N * 8);
    if (on) memcpy(on, M->on, M->N);
    if (bnd) memcpy(bnd, M->bnd, M->N);
    if (status) memcpy(status, M->status, M->N * 4);
    return 0;
}
PO_EXPORT int32_t picles_oracle_set_particles(po_model *M, const double *z, const uint8_t *on)
{
    if (z) memcpy(M->z, z, 5 * M->N * 8);
    if (on) memcpy(M->on, on, M->N);
    for (int64_t n = 0; n < M->N; n++) M->dtn[n] = -1.0;
    return 0;
}
PO_EXPORT int32_t picles_oracle_get_controller(po_model *M, double *qold, double *dtn)
{
    if (qold) memcpy(qold, M->qold, M->N * 8);
    if (dtn) memcpy(dtn, M->dtn, M->N * 8);
    return 0;
}
PO_EXPORT int32_t picles_oracle_get_counters(po_model *M, picles_counters *c) { *c = M->cnt; return 0; }
PO_EXPORT int32_t picles_oracle_get_mask(po_model *M, int8_t *mask) { memcpy(mask, M->mask, M->N); return 0; }
PO_EXPORT int64_t picles_oracle_n_stepped(po_model *M) { return M->n_step; }
PO_EXPORT double picles_oracle_e_T(po_model *M) { return M->k.e_T; }

/* ------------------------------------------------------------------------------------------
 * Slab stepping with a PULL scatter — CPU restatement of the algorithm the HIP kernels run
 * (DESIGN.md "k_scatter"): every particle leaves a scatter record (e, m_x, m_y, x, y, group);
 * node (i,j) then visits its (2R+1)² candidate sources in the reference's sequential order
 * (ocean list before grid-boundary list, each column-major; periodic images sorted by their
 * wrapped index) and adds (wx*wy)*charge of the corner that lands on it.  On a single slab this
 * must reproduce picles_oracle_advance's sequential push bit for bit (tests/test_oracle_pull.py);
 * with ghost record rows filled by the neighbours it is the multi-GPU step.
 * ---------------------------------------------------------------------------------------- */
static inline double *po_rec_row(po_model *M, int row) { return M->rec + (size_t)row * 6 * M->Nx; }

PO_EXPORT int32_t picles_oracle_set_halo_rows(po_model *M, int32_t r)
{
    if (r < 1) return -1;
    M->R = r;
    free(M->rec);
    M->rec = (double *)calloc(po_rec_doubles(M), 8);
    return 0;
}
PO_EXPORT int32_t picles_oracle_halo_rows(const po_model *M) { return M->R; }

PO_EXPORT int32_t picles_oracle_begin_step(po_model *M, double dt, int32_t flags)
{
    M->step_dt = dt;
    M->step_flags = flags;
    M->cnt.max_reach = 0;
    return 0;
}

static void po_write_record(po_model *M, int64_t idx, int *reach_out, int *overflow)
{
    int i = (int)(idx % M->Nx), jl = (int)(idx / M->Nx);
    double *rr = po_rec_row(M, jl + M->R);
    double z[5];
    for (int c = 0; c < 5; c++) z[c] = M->z[idx + c * M->N];
    double flag = 0.0;
    if (M->on[idx] && isfinite(z[3]) && isfinite(z[4])) {
        double c3[3];
        po_particle_to_charge(M->order, z, c3);
        rr[i] = c3[0]; rr[M->Nx + i] = c3[1]; rr[2 * M->Nx + i] = c3[2]; rr[3 * M->Nx + i] = z[3]; rr[4 * M->Nx + i] = z[4];
        flag = (double)M->grp[idx];
        int64_t xi[2], yi[2];
        double w[2];
        po_index_weight(z[3], 0, xi, w);
        po_index_weight(z[4], 0, yi, w);
        int64_t r64 = (xi[0] < 0) ? -xi[0] : xi[0] + 1, ry64 = (yi[0] < 0) ? -yi[0] : yi[0] + 1;
        int r = (r64 > 0x3fffffff) ? 0x3fffffff : (int)r64, ry = (ry64 > 0x3fffffff) ? 0x3fffffff : (int)ry64;
        if (ry > r) r = ry;
        if (r > *reach_out) *reach_out = r;
        if (!M->single_slab && r > M->R) (*overflow)++;
    }
    else if (M->on[idx]) M->cnt.dropped_nonfinite++;
    rr[5 * M->Nx + i] = flag;
}

PO_EXPORT int32_t picles_oracle_advance_rows(po_model *M, int32_t which)
{
    int R = M->R, r0 = 0, n0 = 0, r1 = 0, n1 = 0;
    int small = M->nyl <= 2 * R;
    if (which == PICLES_ROWS_ALL) { n0 = M->nyl; }
    else if (which == PICLES_ROWS_EDGE) { if (small) n0 = M->nyl; else { n0 = R; r1 = M->nyl - R; n1 = R; } }
    else if (which == PICLES_ROWS_INTERIOR) { if (small) return 0; r0 = R; n0 = M->nyl - 2 * R; }
    else return -2;
    uint64_t rhs = 0, acc = 0, rej = 0, adv = 0;
    int reach = 0, overflow = 0;
    for (int part = 0; part < 2; part++) {
        int ra = part ? r1 : r0, na = part ? n1 : n0;
        int64_t a = (int64_t)ra * M->Nx, b = (int64_t)(ra + na) * M->Nx;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(M->nthreads) reduction(+ : rhs, acc, rej, adv)
#endif
        for (int64_t idx = a; idx < b; idx++) {
            if (!M->grp[idx]) continue;
            po_pstats st = {0, 0, 0, 0};
            int was_on = M->on[idx];
            po_advance_particle(M, idx, M->step_dt, &st);
            rhs += st.rhs; acc += st.acc; rej += st.rej; adv += was_on ? 1 : 0;
        }
        for (int64_t idx = a; idx < b; idx++) {
            if (!M->grp[idx]) continue;
            int st = M->status[idx];
            if (st & (PICLES_ST_RESEED_NAN | PICLES_ST_RESEED_INF | PICLES_ST_SWITCHED_ON)) M->cnt.reseeds++;
            if (st & PICLES_ST_CLAMPED) M->cnt.clamps++;
            if (st & PICLES_ST_MAXITERS) M->cnt.maxiters_hits++;
            po_write_record(M, idx, &reach, &overflow);
        }
    }
    M->cnt.rhs_evals += rhs; M->cnt.steps_accepted += acc; M->cnt.steps_rejected += rej;
    M->cnt.particles_advanced += adv;
    M->cnt.halo_overflow += overflow;
    if (reach > M->cnt.max_reach) M->cnt.max_reach = reach;
    if (reach > M->cnt.max_reach_seen) M->cnt.max_reach_seen = reach;
    return 0;
}

/* candidate offsets d in [-R, R] of one axis for node coordinate c, in ascending (wrapped) source index —
 * the reference's sequential visiting order.  Non-periodic: out-of-range sources are dropped.  Periodic with
 * 2R+1 <= N: the rotation that starts at the smallest wrapped index.  Periodic with 2R+1 > N (a reach that
 * wraps around the axis): sources 0..N-1 in turn, each with all its aliasing offsets.  Returns the count. */
static int po_axis_candidates(int c, int N, int R, int periodic, int *d, int *src)
{
    int n = 0, W = 2 * R + 1;
    if (!periodic) {
        for (int k = -R; k <= R; k++)
            if (c + k >= 0 && c + k < N) { d[n] = k; src[n++] = c + k; }
    } else if (W <= N) {
        int sh = 0;
        if (c - R < 0) sh = R - c; else if (c + R >= N) sh = N - c + R;
        for (int q = 0; q < W; q++) {
            int k = (q + sh) % W - R;
            d[n] = k; src[n++] = (int)po_wrap(c + k, N);
        }
    } else {
        for (int sidx = 0; sidx < N; sidx++) {
            int k = sidx - c;
            int a = k + R, fl = a / N;
            if (a % N < 0) fl--;
            for (k -= fl * N; k <= R; k += N) { d[n] = k; src[n++] = sidx; }
        }
    }
    return n;
}

static void po_pull_node(po_model *M, int i, int jl, int R, int accum, double s[3])
{
    int j = jl + M->j0, Nx = M->Nx, Ny = M->Ny, W = 2 * R + 1;
    int64_t t = i + (int64_t)Nx * jl;
    s[0] = accum ? M->state[t] : 0.0;
    s[1] = accum ? M->state[t + M->N] : 0.0;
    s[2] = accum ? M->state[t + 2 * M->N] : 0.0;
    if (M->g.periodic_y == 2 && j >= Ny - R) {
        /* tripolar north fold: a node of the top band receives ordinary corners and corners folded back over the
         * seam (mirrored in x).  Sources are visited in ascending index; each replays its four corners in
         * construct_loop order through the boundary rule of the push. */
        static const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 0, 1, 1};
        for (int grp = 1; grp <= M->ngroups; grp++)
            for (int js = (j - R > 0 ? j - R : 0); js < Ny; js++) {
                const double *rr = po_rec_row(M, js - M->j0 + M->R);
                for (int is = 0; is < Nx; is++) {
                    if (rr[5 * Nx + is] != (double)grp) continue;
                    int64_t xi[2], yi[2];
                    double xw[2], yw[2];
                    po_index_weight(rr[3 * Nx + is], is, xi, xw);
                    po_index_weight(rr[4 * Nx + is], js, yi, yw);
                    for (int k = 0; k < 4; k++) {
                        int64_t ii = xi[ox[k]], jj = yi[oy[k]];
                        if (jj < 0) continue;
                        if (jj >= Ny) { ii = Nx - 1 - po_wrap(ii + 1, Nx); jj = 2 * (int64_t)Ny - 1 - jj; }
                        else ii = po_wrap(ii, Nx);
                        if (ii != i || jj != j) continue;
                        double w = xw[ox[k]] * yw[oy[k]];
                        s[0] += w * rr[is];
                        s[1] += w * rr[Nx + is];
                        s[2] += w * rr[2 * Nx + is];
                    }
                }
            }
        return;
    }
    int dxs[W], sxs[W], dys[W], sys[W];
    int nxc = po_axis_candidates(i, Nx, R, M->g.periodic_x, dxs, sxs);
    int nyc = po_axis_candidates(j, Ny, R, M->g.periodic_y == 1, dys, sys);
    /* sources in ascending index (row, then column); a source reachable through several aliasing offsets (2R+1 > N on
     * a periodic axis) is visited once, with all its offset pairs (at most one pair matches a corner of it) */
    for (int grp = 1; grp <= M->ngroups; grp++)
        for (int a0 = 0; a0 < nyc;) {
            int a1 = a0;
            while (a1 < nyc && sys[a1] == sys[a0]) a1++;
            for (int b0 = 0; b0 < nxc;) {
                int b1 = b0;
                while (b1 < nxc && sxs[b1] == sxs[b0]) b1++;
                const int ii = sxs[b0];
                for (int a = a0; a < a1; a++) {
                    int dj = dys[a];
                    int row = M->single_slab ? sys[a] + M->R : jl + dj + M->R;
                    const double *rr = po_rec_row(M, row);
                    if (rr[5 * Nx + ii] != (double)grp) continue;
                    int64_t xi[2], yi[2];
                    double xw[2], yw[2];
                    po_index_weight(rr[3 * Nx + ii], 0, xi, xw);
                    po_index_weight(rr[4 * Nx + ii], 0, yi, yw);
                    for (int b = b0; b < b1; b++) {
                        int di = dxs[b];
                        int ax = (int)(-di - xi[0]), ay = (int)(-dj - yi[0]);
                        if (ax < 0 || ax > 1 || ay < 0 || ay > 1) continue;
                        double w = xw[ax] * yw[ay];
                        s[0] += w * rr[ii];
                        s[1] += w * rr[Nx + ii];
                        s[2] += w * rr[2 * Nx + ii];
                    }
                }
                b0 = b1;
            }
            a0 = a1;
        }
}

/* pull-scatter own rows (+ optional remesh, MovieState, tick) */
PO_EXPORT int32_t picles_oracle_scatter_rows(po_model *M, int32_t do_remesh)
{
    int flags = M->step_flags;
    int movie = (flags & PICLES_STEP_MOVIE) != 0 && do_remesh;
    int accum = (flags & PICLES_STEP_ZERO_FIRST) ? 0 : 1;
    int R = M->single_slab ? (M->cnt.max_reach > 1 ? M->cnt.max_reach : 1) : M->R;
    if (R > 4096) return -6;   /* a runaway particle: the candidate tables of the pull (2R+1 entries on the stack) are not meant for it;
                                  the sequential push (picles_oracle_time_step) has no such limit */
    double *ns = (double *)malloc(3 * M->N * 8);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(M->nthreads)
#endif
    for (int64_t t = 0; t < M->N; t++) {
        double s[3];
        po_pull_node(M, (int)(t % M->Nx), (int)(t / M->Nx), R, accum, s);
        ns[t] = s[0]; ns[t + M->N] = s[1]; ns[t + 2 * M->N] = s[2];
    }
    memcpy(M->state, ns, 3 * M->N * 8);
    free(ns);
    if (do_remesh) {
        if (movie) memcpy(M->movie, M->state, 3 * M->N * 8);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(M->nthreads)
#endif
        for (int64_t t = 0; t < M->N; t++)
            if (M->grp[t]) po_remesh_particle(M, t, M->step_dt);
        if (movie) memset(M->state, 0, 3 * M->N * 8);
        M->clock += M->step_dt;
    }
    return 0;
}

PO_EXPORT int32_t picles_oracle_halo_ptr(po_model *M, int32_t side, int32_t send, double **ptr, size_t *bytes)
{
    size_t row_d = (size_t)6 * M->Nx;
    int row = send ? (side == 0 ? M->R : M->nyl) : (side == 0 ? 0 : M->nyl + M->R);
    *ptr = M->rec + (size_t)row * row_d;
    *bytes = (size_t)M->R * row_d * 8;
    return 0;
}

/* time_step! through the pull path (single slab) — must equal picles_oracle_time_step bitwise */
PO_EXPORT int32_t picles_oracle_time_step_pull(po_model *M, double dt, int32_t flags)
{
    picles_oracle_begin_step(M, dt, flags);
    picles_oracle_advance_rows(M, PICLES_ROWS_ALL);
    return picles_oracle_scatter_rows(M, 1);
}

/* ---- single-function entry points for unit tests ---------------------------------------- */
PO_EXPORT void picles_oracle_windsea(double U, double V, double T, double out[3]) { po_windsea(U, V, T, out); }
PO_EXPORT void picles_oracle_rhs(po_model *M, const double z[5], double u, double v, double dz[5]) { po_rhs(M, 0, z, u, v, dz); }
PO_EXPORT void picles_oracle_particle_to_charge(const double z[5], double c[3]) { po_particle_to_charge(0, z, c); }
PO_EXPORT void picles_oracle_charge_to_particle(const double c[3], double z[5]) { po_charge_to_particle(0, c, z); }
PO_EXPORT void picles_oracle_index_weight(double zp, int32_t i_node, int64_t idx[2], double w[2]) { po_index_weight(zp, i_node, idx, w); }
/* integrate node idx's wind with an explicit start state over DT; returns rhs/acc/rej in stats[3] */
PO_EXPORT int32_t picles_oracle_integrate(po_model *M, int64_t idx, double z[5], double *qold, double *dtn,
                                          double t_start, double DT, uint64_t stats[3])
{
    po_pstats st = {0, 0, 0, 0};
    po_integrate(M, idx, z, qold, dtn, t_start, DT, &st);
    stats[0] = st.rhs; stats[1] = st.acc; stats[2] = st.rej;
    return st.status;
}
PO_EXPORT int32_t picles_oracle_integrate_auto(po_model *M, int64_t idx, double z[5], double *qold, double *dtn, int32_t *asw,
                                               double t_start, double DT, uint64_t stats[3])
{
    po_pstats st = {0, 0, 0, 0};
    po_integrate_auto(M, idx, z, qold, dtn, asw, t_start, DT, &st);
    stats[0] = st.rhs; stats[1] = st.acc; stats[2] = st.rej;
    return st.status;
}
/* f and one directional derivative of the RHS (test hook for the hand-written Jacobian) */
PO_EXPORT void picles_oracle_rhs_jvp(po_model *M, const double z[5], double u, double v, const double seed[5],
                                     double f[3], double df[3])
{
    if (M->order == 1 && po_is_fast(M) && po_is_plain(M, z, u, v)) {
        /* the structured Jacobian of the kernel order (what po_ros23_try uses for this particle), applied to the direction */
        double J[3][3], dT[3], dz[5];
        po_rhs_jac_plain(M, 0, z, u, v, seed[3], seed[4], 1, J, dT);
        po_rhs_kernel(M, 0, z, u, v, dz);
        for (int r = 0; r < 3; r++) {
            f[r] = dz[r];
            df[r] = ((J[r][0] * seed[0] + J[r][1] * seed[1]) + J[r][2] * seed[2]) + dT[r];
        }
        return;
    }
    const double seeds[1][5] = {{seed[0], seed[1], seed[2], seed[3], seed[4]}};
    double d[1][3];
    po_rhs_jvp(M, 0, z, u, v, 1, seeds, f, d);
    df[0] = d[0][0]; df[1] = d[0][1]; df[2] = d[0][2];
}
PO_EXPORT int32_t picles_oracle_is_pmath(void)
{
#ifdef PO_PMATH
    return 1;
#else
    return 0;
#endif
}
PO_EXPORT int32_t picles_oracle_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
PO_EXPORT void picles_oracle_math(int32_t fn, int64_t n, const double *x, const double *y, double *out)
{
    for (int64_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = o_exp(x[i]); break;
        case 1: out[i] = o_log(x[i]); break;
        case 2: out[i] = o_pow(x[i], y[i]); break;
        case 3: out[i] = o_tanh(x[i]); break;
        case 4: out[i] = o_cosh(x[i]); break;
        case 5: out[i] = x[i] / y[i]; break;
        case 6: out[i] = sqrt(x[i]); break;
        case 7: out[i] = o_log_coarse(x[i]); break;
#ifdef PO_PMATH
        case 8: out[i] = pm_rsqrt(x[i]); break;
        case 9: out[i] = pm_div_1e6(x[i]); break;      /* the kernels' n / 1e6 (three operations), for the exhaustive comparison */
#endif
        default: out[i] = NAN;
        }
    }
}
