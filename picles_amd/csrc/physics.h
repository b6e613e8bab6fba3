/*
 * physics.h — per-particle arithmetic of the PiCLES 2D time step for gfx950 (device inline code).
 *
 * Everything one thread does to one particle lives here, in registers:
 *   seed_windsea      FetchRelations.get_initial_windsea            (FetchRelations.jl:314-359)
 *   rhs               particle_system(dz,z,params,t), Cartesian mesh (particle_waves_v5.jl:479-556)
 *   integrate_dp5     step!(integrator, DT, true) with DP5 + PI controller + Hairer initial dt
 *                     (call site mapping_2D.jl:152; OrdinaryDiffEq semantics: SURVEY Appendix C)
 *   particle_to_charge / charge_to_particle   core_2D.jl:69-78 / :121-128
 *   index_weight      get_absolute_i_and_w(z, i_node)                 (ParticleInCell.jl:58-71)
 *
 * The evaluation order ("kernel order") is chosen for the CDNA4 fp64 VALU:
 * one reciprocal 1/c_gp feeds k_p, ω_p, α, α_p and the direction term; |g| ≡ c_gp;
 * sin 2(θ_c-θ_w) = 2·cross·dot/(U c_gp)²; H_β through the logistic function; sech² through one
 * exp; every multiply-add that is fused is written as fma() and the TU is compiled with
 * -ffp-contract=off, so results are bit-identical to the CPU oracle built with the same
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
    double inv_dx, inv_dy;
    int propagation, input, dissipation, peak_shift, direction, n_is_2;
    /* ODE settings */
    double abstol, reltol, dt0, dtmin;
    long long maxiters;
    int force_dtmin;
    double lne_max, wind_min_sq;
    /* model */
    int init_type;
    double def_lne, def_cx, def_cy;
    double min_e, min_m2;
    /* winds */
    int wind_static;
    double tw0, inv_dtw;
};

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

/* PI controller (OrdinaryDiffEq defaults for DP5) */
#define PI_BETA1 0.17
#define PI_BETA2 0.04
#define PI_GAMMA 0.9
#define PI_QMIN 0.2
#define PI_QMAX 10.0
#define PI_QOLDINIT 1e-4

struct Vec5 {
    double lne, cx, cy, x, y;
};

struct Wind {
    double u0, v0, du, dv; /* level 0 and (level1 - level0) */
};

struct PStats {
    unsigned int rhs, acc, rej;
    int status;
};

PM_HD void wind_at(const KParams &P, const Wind &w, double t, double &u, double &v)
{
    if (P.wind_static) {
        u = w.u0;
        v = w.v0;
    } else {
        double s = (t - P.tw0) * P.inv_dtw;
        u = PM_FMA(w.du, s, w.u0);
        v = PM_FMA(w.dv, s, w.v0);
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

/* GetParticleEnergyMomentum */
PM_HD void particle_to_charge(double lne, double cx, double cy, double &e, double &mx, double &my)
{
    e = pm_exp(lne);
    double sp = __builtin_sqrt(cx * cx + cy * cy);
    mx = cx * e / (sp * sp) / 2.0;
    my = cy * e / (sp * sp) / 2.0;
}

/* GetVariablesAtVertex */
PM_HD void charge_to_particle(double e, double mx, double my, Vec5 &z)
{
    double ma = __builtin_sqrt(mx * mx + my * my);
    z.lne = pm_log(e);
    z.cx = mx * e / (2.0 * (ma * ma));
    z.cy = my * e / (2.0 * (ma * ma));
    z.x = 0.0;
    z.y = 0.0;
}

/* get_absolute_i_and_w: cell offset b = floor(z) relative to the birth node and the weight of
 * the upper node, round(z - b, digits = 6) */
PM_HD void index_weight(double zp, int &b, double &w_hi)
{
    double fb = __builtin_floor(zp);
    b = (int)fb;
    w_hi = __builtin_rint((zp - fb) * 1e6) / 1e6;
}

/* RHS in kernel order.  invU2 = 1/U² is hoisted by the caller when the wind is time-constant. */
PM_HD void rhs(const KParams &P, const Vec5 &z, double u, double v, Vec5 &d)
{
    double cx = z.cx, cy = z.cy;
    double c2 = PM_FMA(cx, cx, cy * cy);
    double cbar = __builtin_sqrt(c2);
    double U2 = PM_FMA(u, u, v * v);
    double U = __builtin_sqrt(U2);
    double cgp = cbar * P.inv_rg;
    double rc = 1.0 / cgp;
    double minv = (cgp >= 0.1) ? rc : 10.0;
    double wp = (0.5 * PK_G0) * minv;
    double kp = (0.25 * PK_G0) * (minv * minv);
    double a = (0.5 * U) * rc;
    double alpha = (a > 500.0) ? 500.0 : a;
    double gx = cx * P.inv_rg, gy = cy * P.inv_rg;
    double dot = PM_FMA(u, gx, v * gy);
    double crs = u * gy - v * gx;
    double rc2 = rc * rc;
    double sginv2 = (cgp >= 1e-4) ? rc2 : 1e8;
    double ap = (0.5 * dot) * sginv2;
    double ya = ap - 0.85;
    double H = 1.0 / (1.0 + pm_exp(P.neg2p * ya));
    double t = pm_exp(-20.0 * pm_fabs(ya));
    double t1 = 1.0 + t;
    double D = 1.0 - (5.0 * t) / (t1 * t1);

    double It = 0.0, Dt = 0.0, Scg = 0.0, Sd = 0.0, E2 = 0.0;
    if ((P.dissipation && P.n_is_2) || P.peak_shift) E2 = pm_exp(2.0 * z.lne);
    if (P.input) It = (P.C_e * H) * (alpha * alpha);
    if (P.dissipation) {
        double ke = kp * P.inv_eT;
        if (P.n_is_2) {
            double ke2 = ke * ke;
            Dt = E2 * (ke2 * ke2);
        } else {
            Dt = pm_exp(P.n * z.lne) * pm_pow(ke, 2.0 * P.n);
        }
    }
    if (P.peak_shift) {
        double k2 = kp * kp;
        Scg = ((P.C_alpha * D) * (k2 * k2)) * E2;
    }
    if (P.direction) {
        double s2;
        if (U == 0.0 || cgp == 0.0)
            s2 = 0.0;
        else
            s2 = ((2.0 * crs) * dot) * (rc2 * (1.0 / U2));
        Sd = (((alpha * alpha) * P.C_phi) * H) * s2;
    }
    double wrS = (wp * P.r_g) * Scg;
    d.lne = PM_FMA(wp, It - Dt, wrS);
    d.cx = PM_FMA(cy, Sd, -(cx * wrS));
    d.cy = -PM_FMA(cx, Sd, cy * wrS);
    if (P.propagation) {
        d.x = cx * P.inv_dx;
        d.y = cy * P.inv_dy;
    } else {
        d.x = 0.0;
        d.y = 0.0;
    }
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

#define V5_MAP2(out, A, B, EXPR)                   \
    {                                              \
        { double a = (A).lne, b = (B).lne; (out).lne = (EXPR); } \
        { double a = (A).cx,  b = (B).cx;  (out).cx  = (EXPR); } \
        { double a = (A).cy,  b = (B).cy;  (out).cy  = (EXPR); } \
        { double a = (A).x,   b = (B).x;   (out).x   = (EXPR); } \
        { double a = (A).y,   b = (B).y;   (out).y   = (EXPR); } \
    }

/* ode_determine_initdt (Hairer–Wanner), = auto_dt_reset! after every remesh */
PM_HD double init_dt(const KParams &P, const Wind &w, const Vec5 &u0, const Vec5 &f0, double t, PStats &st)
{
    Vec5 sk, q0, q1;
    V5_MAP2(sk, u0, u0, PM_FMA(pm_fabs(a), P.reltol, P.abstol));
    V5_MAP2(q0, u0, sk, a / b);
    V5_MAP2(q1, f0, sk, a / b);
    double d0 = rms5(q0.lne, q0.cx, q0.cy, q0.x, q0.y);
    double d1 = rms5(q1.lne, q1.cx, q1.cy, q1.x, q1.y);
    double dt0;
    if (d0 < 1e-5 || d1 < 1e-5) dt0 = 1e-6;
    else dt0 = 0.01 * (d0 / d1);
    if (dt0 < 10.0 * 2.220446049250313e-16) return 1e-6;
    Vec5 u1, f1;
    V5_MAP2(u1, f0, u0, PM_FMA(dt0, a, b));
    double uw, vw;
    wind_at(P, w, t + dt0, uw, vw);
    rhs(P, u1, uw, vw, f1);
    st.rhs++;
    Vec5 df;
    V5_MAP2(df, f1, f0, a - b);
    V5_MAP2(q1, df, sk, a / b);
    double d2 = rms5(q1.lne, q1.cx, q1.cy, q1.x, q1.y) / dt0;
    double m = (d1 > d2) ? d1 : d2;
    double dt1;
    if (m <= 1e-15) {
        double c = dt0 * 1e-3;
        dt1 = (1e-6 > c) ? 1e-6 : c;
    } else {
        double l10 = pm_log(m) * 0.43429448190325182765;
        dt1 = pm_exp(((2.0 + l10) * -0.2) * 2.30258509299404568402);
    }
    double h = 100.0 * dt0;
    if (dt1 < h) h = dt1;
    if (!(h == h)) h = 1e-6;
    return (P.dtmin > h) ? P.dtmin : h;
}

/* step!(integrator, DT, true): integrate z over [t_start, t_start+DT] */
PM_HD void integrate_dp5(const KParams &P, const Wind &w, Vec5 &z, double &qold, double &dtn,
                         double t_start, double DT, PStats &st)
{
    Vec5 k1, k2, k3, k4, k5, k6, k7, g, un;
    double uw, vw;
    double tr = 0.0;
    wind_at(P, w, t_start, uw, vw);
    rhs(P, z, uw, vw, k1);
    st.rhs++;
    double dt = dtn;
    if (!(dt > 0.0)) dt = init_dt(P, w, z, k1, t_start, st);
    long long iter = 0;
    while (tr < DT) {
        iter++;
        if (iter > P.maxiters) { st.status |= 2 /*PICLES_ST_MAXITERS*/; break; }
        if (dt < P.dtmin) dt = P.dtmin;
        double rem = DT - tr;
        bool last = !(dt < rem);
        double h = last ? rem : dt;
        double t = t_start + tr;
        {
            double a21h = h * DP_A21;
            V5_MAP2(g, k1, z, PM_FMA(a21h, a, b));
        }
        wind_at(P, w, PM_FMA(DP_C2, h, t), uw, vw);
        rhs(P, g, uw, vw, k2);
#define ST3(c) PM_FMA(h, PM_FMA(DP_A32, k2.c, DP_A31 * k1.c), z.c)
        g.lne = ST3(lne); g.cx = ST3(cx); g.cy = ST3(cy); g.x = ST3(x); g.y = ST3(y);
        wind_at(P, w, PM_FMA(DP_C3, h, t), uw, vw);
        rhs(P, g, uw, vw, k3);
#define ST4(c) PM_FMA(h, PM_FMA(DP_A43, k3.c, PM_FMA(DP_A42, k2.c, DP_A41 * k1.c)), z.c)
        g.lne = ST4(lne); g.cx = ST4(cx); g.cy = ST4(cy); g.x = ST4(x); g.y = ST4(y);
        wind_at(P, w, PM_FMA(DP_C4, h, t), uw, vw);
        rhs(P, g, uw, vw, k4);
#define ST5(c) PM_FMA(h, PM_FMA(DP_A54, k4.c, PM_FMA(DP_A53, k3.c, PM_FMA(DP_A52, k2.c, DP_A51 * k1.c))), z.c)
        g.lne = ST5(lne); g.cx = ST5(cx); g.cy = ST5(cy); g.x = ST5(x); g.y = ST5(y);
        wind_at(P, w, PM_FMA(DP_C5, h, t), uw, vw);
        rhs(P, g, uw, vw, k5);
#define ST6(c) PM_FMA(h, PM_FMA(DP_A65, k5.c, PM_FMA(DP_A64, k4.c, PM_FMA(DP_A63, k3.c, PM_FMA(DP_A62, k2.c, DP_A61 * k1.c)))), z.c)
        g.lne = ST6(lne); g.cx = ST6(cx); g.cy = ST6(cy); g.x = ST6(x); g.y = ST6(y);
        wind_at(P, w, t + h, uw, vw);
        rhs(P, g, uw, vw, k6);
#define ST7(c) PM_FMA(h, PM_FMA(DP_A76, k6.c, PM_FMA(DP_A75, k5.c, PM_FMA(DP_A74, k4.c, PM_FMA(DP_A73, k3.c, DP_A71 * k1.c)))), z.c)
        un.lne = ST7(lne); un.cx = ST7(cx); un.cy = ST7(cy); un.x = ST7(x); un.y = ST7(y);
        rhs(P, un, uw, vw, k7);
        st.rhs += 6;
#define ERRC(c) ((h * PM_FMA(DP_E7, k7.c, PM_FMA(DP_E6, k6.c, PM_FMA(DP_E5, k5.c, PM_FMA(DP_E4, k4.c, PM_FMA(DP_E3, k3.c, DP_E1 * k1.c)))))) / \
                 PM_FMA(pm_max(pm_fabs(z.c), pm_fabs(un.c)), P.reltol, P.abstol))
        double EEst = rms5(ERRC(lne), ERRC(cx), ERRC(cy), ERRC(x), ERRC(y));
#undef ST3
#undef ST4
#undef ST5
#undef ST6
#undef ST7
#undef ERRC
        if (!(EEst == EEst)) { EEst = pm_inf(); st.status |= 128 /*PICLES_ST_NONFINITE*/; }
        double q11 = 0.0, q;
        if (EEst == 0.0) {
            q = 1.0 / PI_QMAX;
        } else {
            q11 = pm_pow(EEst, PI_BETA1);
            q = q11 / pm_pow(qold, PI_BETA2);
            double qg = q / PI_GAMMA;
            const double lo = 1.0 / PI_QMAX, hi = 1.0 / PI_QMIN;
            q = (qg < hi) ? qg : hi;
            q = (q > lo) ? q : lo;
        }
        bool accept = (EEst <= 1.0) || (P.force_dtmin && h <= P.dtmin);
        if (accept) {
            st.acc++;
            qold = (EEst > PI_QOLDINIT) ? EEst : PI_QOLDINIT;
            dt = h / q;
            z = un;
            k1 = k7;
            tr = last ? DT : tr + h;
            if (z.lne != z.lne || z.cx != z.cx || z.cy != z.cy || z.x != z.x || z.y != z.y) break;
        } else {
            st.rej++;
            double f = q11 / PI_GAMMA;
            const double hi = 1.0 / PI_QMIN;
            dt = h / ((f < hi) ? f : hi);
            if (!P.force_dtmin && h <= P.dtmin) { st.status |= 64 /*PICLES_ST_DTMIN*/; break; }
        }
    }
    dtn = dt;
}

#endif /* PICLES_PHYSICS_H */
