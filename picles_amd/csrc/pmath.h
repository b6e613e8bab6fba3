/*
 * pmath.h — deterministic fp64 elementary functions shared by host and device.
 *
 * Why: the per-particle ODE advance is adaptive (accept/reject, PI step-size control) and,
 * with C_phi = 0.04, numerically stiff — a 1-ulp difference between glibc's and the device
 * library's exp/log/tanh is amplified to tolerance level.  To make "GPU == CPU oracle"
 * a BITWISE statement, every transcendental on the path is built here from IEEE-754
 * correctly rounded primitives only (+ - * / sqrt fma rint, bit moves), with every fused
 * multiply-add written explicitly (translation units are compiled with -ffp-contract=off),
 * so gcc on x86-64 and hipcc on gfx950 produce identical bits.
 *
 * Accuracy (measured in tests/test_pmath.py against glibc): exp < 2 ulp, log < 1 ulp;
 * pow(x,y) = exp(y log x) carries |y ln x| ulp (seeding only, compared at 1e-13).
 *
 * No reference counterpart: the reference calls Julia's libm-equivalents
 * (particle_waves_v5.jl:274-275,331-340; FetchRelations.jl:128-203).
 */
#ifndef PICLES_PMATH_H
#define PICLES_PMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD static inline
#endif

#define PM_FMA(a, b, c) __builtin_fma((a), (b), (c))

PM_HD uint64_t pm_bits(double x)
{
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
PM_HD double pm_from_bits(uint64_t u)
{
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
PM_HD double pm_inf(void) { return pm_from_bits(0x7ff0000000000000ULL); }
PM_HD double pm_nan(void) { return pm_from_bits(0x7ff8000000000000ULL); }
PM_HD int pm_isnan(double x) { return x != x; }
PM_HD int pm_isinf(double x) { return (pm_bits(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
PM_HD int pm_isfinite(double x) { return (pm_bits(x) & 0x7ff0000000000000ULL) != 0x7ff0000000000000ULL; }
PM_HD double pm_fabs(double x) { return __builtin_fabs(x); }   /* folds into a source modifier on the device */
PM_HD double pm_max(double a, double b) { return (a > b) ? a : b; }   /* not NaN-propagating on b */
PM_HD double pm_min(double a, double b) { return (a < b) ? a : b; }
/* IEEE minNum / maxNum: a NaN operand yields the other one.  One v_min_f64 / v_max_f64 on the device
 * (a compare + two selects otherwise); fmin / fmax on the host: identical results, NaNs included. */
PM_HD double pm_fmin(double a, double b) { return __builtin_fmin(a, b); }
PM_HD double pm_fmax(double a, double b) { return __builtin_fmax(a, b); }
/* the same minNum / maxNum with the instruction written out on the device.  The compiler precedes v_min / v_max with a
 * canonicalisation of every operand it cannot prove quiet (v_max x, x, x: loop-carried values, function results) — the
 * instruction quiets a signalling NaN by itself in the kernels' IEEE mode, so those are wasted issue slots (16 per RK attempt,
 * measured in the listing).  c: a wave-uniform bound (literal or kernel parameter). */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double pm_fmin_c(double a, double c)
{
    double r;
    __asm__("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
__device__ __forceinline__ double pm_fmax_c(double a, double c)
{
    double r;
    __asm__("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
}
__device__ __forceinline__ double pm_fmax_abs(double a, double b)      /* max(|a|, |b|) */
{
    double r;
    __asm__("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
PM_HD double pm_fmin_c(double a, double c) { return __builtin_fmin(a, c); }
PM_HD double pm_fmax_c(double a, double c) { return __builtin_fmax(a, c); }
PM_HD double pm_fmax_abs(double a, double b) { return __builtin_fmax(__builtin_fabs(a), __builtin_fabs(b)); }
#endif

/* A constant pinned to a scalar register pair AT ITS USE.  The compiler's habit with fp64 literals that are neither inline
 * constants nor "high dword only" is to park them in VGPR pairs for the whole kernel (loop-invariant) and, for a Horner step
 * p = fma(p, z, C), to copy C into the accumulator first (v_mov_b64 + v_fmac: two issue slots).  Inside the Runge-Kutta loop of the
 * fused step that was 14 vector registers of constants at a 128-register budget.  With the addend in scalar registers the step is one
 * v_fma_f64 (one scalar operand per instruction is allowed), the two s_mov that rebuild the pair each time go to the scalar unit,
 * which has slack.  (volatile: not hoisted out of the loop into the — full — scalar file.)  Identity on the host. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double pm_sc(double c)
{
    __asm__ volatile("" : "+s"(c));
    return c;
}
#else
PM_HD double pm_sc(double c) { return c; }
#endif

/* 2^k for -1022 <= k <= 1023 */
PM_HD double pm_pow2i(int k) { return pm_from_bits((uint64_t)(k + 1023) << 52); }

/* 1/b and a/b for PLAIN operands — b finite and normal with |b| <= 2^1021, a finite, the quotient neither overflowing nor
 * denormal — as the call sites below guarantee by construction.  On the device this is the compiler's own expansion of an IEEE
 * fp64 division (v_rcp_f64 seed, two Newton steps, one residual correction) without its range scaling (2 x v_div_scale) and its
 * special-case fix-up (v_div_fmas, v_div_fixup), which are the identity for plain operands: 7 / 8 issue slots instead of 11 / 12,
 * and the same correctly rounded quotient, i.e. the same bits as `/` on the host.  A NaN operand gives a NaN either way; the
 * one deviation inside the plain range is the sign of a zero quotient from a = -0 (+0 here) — the numerators at the call sites
 * are differences m - 1.0, which are never -0.
 * (tests/test_gpu_pmath.py compares the device results with the host's division bit by bit.) */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double pm_rcp_plain(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = PM_FMA(-b, y, 1.0);
    y = PM_FMA(y, e, y);
    e = PM_FMA(-b, y, 1.0);
    y = PM_FMA(y, e, y);
    e = PM_FMA(-b, y, 1.0);
    return PM_FMA(e, y, y);
}
__device__ __forceinline__ double pm_div_plain(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = PM_FMA(-b, y, 1.0);
    y = PM_FMA(y, e, y);
    e = PM_FMA(-b, y, 1.0);
    y = PM_FMA(y, e, y);
    double q = a * y;
    double r = PM_FMA(-b, q, a);
    return PM_FMA(r, y, q);
}
#else
PM_HD double pm_rcp_plain(double b) { return 1.0 / b; }
PM_HD double pm_div_plain(double a, double b) { return a / b; }
#endif

/* n / 1e6 for an integer-valued 0 <= n <= 1e6 (round(x, digits = 6) of a cell fraction): q = n·RN(1e-6) is within an ulp,
 * one residual and one correction make it the correctly rounded quotient (Markstein) — verified for every one of the 1 000 001
 * arguments against the division (tests/test_pmath.py::test_div_1e6_exhaustive).  3 issue slots instead of 12. */
PM_HD double pm_div_1e6(double n)
{
    double q = n * 1e-6;
    double r = PM_FMA(-1e6, q, n);
    return PM_FMA(r, 1e-6, q);
}

/* 2^(j/512), j = 0..511, correctly rounded, and the reduction constants (generated) */
#include "pm_exp_tab.h"

#if defined(__HIP_DEVICE_COMPILE__)
/* device: the table lives in LDS (per-lane index => ds_read_b64, no scalar path possible).
 * Every kernel that evaluates pm_exp calls pm_device_init() first. */
__device__ __constant__ const double PM_EXP_TAB_C[PM_EXP_N] = PM_EXP_TAB_INIT;
__device__ __forceinline__ double *pm_lds_tab(void)
{
    __shared__ double tab[PM_EXP_N];
    return tab;
}
__device__ __forceinline__ void pm_device_init(void)
{
    /* 256 threads per workgroup in every kernel that calls this (not blockDim.x: that is a load from the dispatch packet, and with
     * the trip count known both table loads are in flight together): two entries per thread */
    static_assert(PM_EXP_N == 512, "two table entries per thread of a 256-thread workgroup");
    const double t0 = PM_EXP_TAB_C[threadIdx.x], t1 = PM_EXP_TAB_C[threadIdx.x + 256];
    pm_lds_tab()[threadIdx.x] = t0;
    pm_lds_tab()[threadIdx.x + 256] = t1;
    __syncthreads();
}
#define PM_EXP_TAB(j) (pm_lds_tab()[(j)])
#else
static const double PM_EXP_TAB_H[PM_EXP_N] = PM_EXP_TAB_INIT;
#define PM_EXP_TAB(j) (PM_EXP_TAB_H[(j)])
#if defined(__HIPCC__)
__device__ __forceinline__ void pm_device_init(void) {}   /* host pass of hipcc: declaration only */
#endif
#endif

/* exp(x): table + short polynomial.  Out-of-range arguments are clamped and overflow / underflow
 * through the final ldexp; NaN propagates through the polynomial. */
/* core: x already inside [-746, 710] (or NaN).
 * x = (512 m + j) ln2/512 + r, |r| <= ln2/1024; exp(x) = 2^m * 2^(j/512) * (1 + r P3(r)); the truncation error
 * r^5/120 <= 1.2e-18 is far below the rounding of the final product. */
PM_HD double pm_exp_core(double x)
{
    /* k = rint(x 512/ln2) through the shifter 1.5·2^52: the sum is rounded to an integer by the addition itself (round to nearest
     * even, |x 512/ln2| < 2^20), its low 32 mantissa bits ARE k in two's complement, and the subtraction that recovers k as a
     * double is exact — two issue slots (fma, add) instead of three (mul, rndne, cvt) */
    const double kd = PM_FMA(x, PM_EXP_RN, 6755399441055744.0);
    const double k = kd - 6755399441055744.0;
    double r = PM_FMA(-k, PM_EXP_LHI, x);
    r = PM_FMA(-k, PM_EXP_LLO, r);
    int ki = (int)(uint32_t)pm_bits(kd);
    int j = ki & (PM_EXP_N - 1);
    int m = ki >> PM_EXP_SHIFT;
    double p = 4.1666666666666664e-02;                /* 1/4! */
    p = PM_FMA(p, r, 1.6666666666666666e-01);         /* 1/3! */
    p = PM_FMA(p, r, 0.5);
    p = PM_FMA(p, r, 1.0);
    double q = p * r;                                 /* expm1(r): small, so T + T q rounds once */
    double T = PM_EXP_TAB(j);
    return __builtin_ldexp(PM_FMA(T, q, T), m);
}

/* The clamps are the identity for |x| <= 700.  On the device they sit behind a wave-uniform test
 * (one compare instead of two compare+select pairs when every lane of the wave is in range — the
 * normal case); the result is the same bit pattern either way. */
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_WAVE_ALL(c) __all(c)
#define PM_RARE_PATH() asm volatile("")   /* keeps the rare path a real branch (no if-conversion into selects) */
#else
#define PM_WAVE_ALL(c) (c)
#define PM_RARE_PATH() ((void)0)
#endif

PM_HD double pm_exp(double x)
{
    if (!PM_WAVE_ALL(pm_fabs(x) <= 700.0)) {
        PM_RARE_PATH();
        x = (x > 710.0) ? 710.0 : x;
        x = (x < -746.0) ? -746.0 : x;
    }
    return pm_exp_core(x);
}

/* exp(min(x, 700)): never overflows (callers that multiply the result by a possible zero) */
PM_HD double pm_exp_finite(double x)
{
    if (!PM_WAVE_ALL(pm_fabs(x) <= 700.0)) {
        PM_RARE_PATH();
        x = (x > 700.0) ? 700.0 : x;
        x = (x < -746.0) ? -746.0 : x;
    }
    return pm_exp_core(x);
}

/* exp(x) for callers that guarantee |x| <= 700 or x NaN: same bits as pm_exp, no clamps
 * (a NaN reaches the table through a masked index and propagates through the polynomial) */
PM_HD double pm_exp_bounded(double x) { return pm_exp_core(x); }

/* log(x), branch-free main path (fdlibm style: x = 2^k (1+f), s = f/(2+f),
 * log(1+f) = f - hfsq + s (hfsq + R(s^2))), special cases selected at the end */
PM_HD double pm_log(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ux = pm_bits(x);
    int sub = ((ux >> 52) == 0);                      /* +0 or positive subnormal */
    double xs = sub ? x * 18014398509481984.0 : x;    /* 2^54 */
    int k = sub ? -54 : 0;
    ux = pm_bits(xs);
    uint32_t hx = (uint32_t)(ux >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    ux = ((uint64_t)hx << 32) | (ux & 0xffffffffULL);
    double m = pm_from_bits(ux);
    double f = m - 1.0;
    double hfsq = 0.5 * f * f;
    double s = pm_div_plain(f, 2.0 + f);      /* 2 + f in [1.70, 2.42] whatever x is (m is rebuilt from the mantissa bits) */
    double z = s * s;
    double w = z * z;
    double t1 = w * PM_FMA(w, PM_FMA(w, Lg6, Lg4), Lg2);
    double t2 = z * PM_FMA(w, PM_FMA(w, PM_FMA(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    double res = dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
    res = (x == 0.0) ? -pm_inf() : res;
    res = (x < 0.0) ? pm_nan() : res;
    res = (x == pm_inf()) ? x : res;
    res = (x != x) ? x : res;
    return res;
}

/* x^y for x > 0 (fetch-relation seeding and e_T only) */
/* 1/sqrt(x) for x > 0 (normal range), ~1 ulp, from +, *, fma and integer ops only — the same bits on
 * the host and on the device (the hardware v_rsq/v_rcp approximations are not reproducible on a CPU,
 * and sqrt followed by a division costs twice as many issue slots).  Bit-trick seed (3.4 %), four
 * Newton steps y += y (1/2 - x/2 y²).  x = 0 gives a huge finite value, never inf. */
PM_HD double pm_rsqrt(double x)
{
    uint32_t hi = 0x5fe6eb50u - ((uint32_t)(pm_bits(x) >> 32) >> 1);
    double y = pm_from_bits((uint64_t)hi << 32);
    const double hx = 0.5 * x;
    double e;
    e = PM_FMA(-hx, y * y, 0.5); y = PM_FMA(y, e, y);
    e = PM_FMA(-hx, y * y, 0.5); y = PM_FMA(y, e, y);
    e = PM_FMA(-hx, y * y, 0.5); y = PM_FMA(y, e, y);
    e = PM_FMA(-hx, y * y, 0.5); y = PM_FMA(y, e, y);
    return y;
}

/* ln(x) to ~1e-9 relative for the step-size controller (x = EEst² >= 0, possibly 0 or +inf):
 * no special cases — 0 and subnormals come out near -709·…, +inf near +710, which the controller
 * clamps exactly as it clamps -inf / +inf.  ln m = 2 atanh(s), s = (m-1)/(m+1), m in [√½, √2). */
PM_HD double pm_log_coarse(double x)
{
    uint64_t ux = pm_bits(x);
    uint32_t hx = (uint32_t)(ux >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    double m = pm_from_bits(((uint64_t)hx << 32) | (ux & 0xffffffffULL));
    double f = m - 1.0;
    double s = pm_div_plain(f, 2.0 + f);      /* 2 + f in [1.70, 2.42] */
    double z = s * s;
    double p = PM_FMA(z, PM_FMA(z, PM_FMA(z, z * pm_sc(2.0 / 9.0) + pm_sc(2.0 / 7.0), pm_sc(2.0 / 5.0)), pm_sc(2.0 / 3.0)), 2.0);
    return PM_FMA((double)k, pm_sc(6.93147180559945286227e-01), s * p);
}

PM_HD double pm_pow(double x, double y) { return pm_exp(y * pm_log(x)); }

/* tanh(x) = sign(x) (1-t)/(1+t), t = exp(-2|x|)   (absolute accuracy ~1e-16) */
PM_HD double pm_tanh(double x)
{
    if (x != x) return x;
    double t = pm_exp(-2.0 * pm_fabs(x));
    double r = (1.0 - t) / (1.0 + t);
    return (x < 0.0) ? -r : r;
}
/* cosh(x) = (e + 1/e)/2, e = exp(|x|) (overflows to inf like libm beyond ~710) */
PM_HD double pm_cosh(double x)
{
    if (x != x) return x;
    double e = pm_exp(pm_fabs(x));
    return 0.5 * (e + 1.0 / e);
}
/* logistic 1/(1+exp(-a)) : H_beta = 0.5 (1 + tanh(y)) = pm_logistic(2y) */
PM_HD double pm_logistic(double a) { return 1.0 / (1.0 + pm_exp(-a)); }
/* sech(x)^2 = 4t/(1+t)^2, t = exp(-2|x|) */
PM_HD double pm_sech2(double x)
{
    double t = pm_exp(-2.0 * pm_fabs(x));
    double d = 1.0 + t;
    return (4.0 * t) / (d * d);
}

#endif /* PICLES_PMATH_H */
