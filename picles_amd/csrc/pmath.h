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

/* The same with the two s_mov written out (PM_SCI(literal)): pm_sc pins the VALUE, but the compiler still creates the halves of
 * its operand where it likes — for a constant whose low dword is zero it shares a zero register and keeps the high dword as a loop
 * invariant, which under scalar register pressure is a spilled one (v_readlane + v_writelane per use, measured in the accept path
 * of the step-size controller).  Here both halves are created by the asm itself, at the use. */
#if defined(__HIP_DEVICE_COMPILE__)
template <unsigned long long BITS>
__device__ __forceinline__ double pm_sci_(void)
{
    unsigned lo, hi;
    __asm__ volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((unsigned)(BITS & 0xffffffffull)), "n"((unsigned)(BITS >> 32)));
    return pm_from_bits(((uint64_t)hi << 32) | lo);
}
#define PM_SCI(c) pm_sci_<__builtin_bit_cast(unsigned long long, (double)(c))>()
#else
#define PM_SCI(c) (c)
#endif

/* A constant kept in a VECTOR register pair: an instruction reads at most one scalar operand, so of the two constants of
 * fma(x, 512/ln2, shifter) one has to be a vector register; left to itself the compiler re-creates it at some of its uses (s_mov,
 * s_mov, v_mov_b64: one issue slot each time).  Behind an asm the value is opaque — an ordinary loop invariant of the code that
 * follows — but every pm_vc() is a value of its own: the Runge-Kutta loop creates ONE (PM_EXP_SHIFTER(), ahead of the loop) and hands
 * it to all its exponentials.  Identity on the host. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double pm_vc(double c)
{
    __asm__("" : "+v"(c));
    return c;
}
#else
PM_HD double pm_vc(double c) { return c; }
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
/* The LDS copy is BIASED: entry j holds 2^(j/512) with (j << 11) subtracted from its high dword, so that the exponent m and the
 * index j of ki = 512 m + j are put back by ONE integer instruction, hi + (ki << 11) = hi(2^(j/512)) + (m << 20) — the high dword of
 * 2^m 2^(j/512) (pm_exp_plain below; v_lshl_add_u32 in place of a shift and a v_ldexp_f64). */
__device__ __forceinline__ double pm_tab_bias(double t, unsigned j) { return pm_from_bits(pm_bits(t) - ((uint64_t)(j << 11) << 32)); }
__device__ __forceinline__ void pm_device_init(void)
{
    /* 256 threads per workgroup in every kernel that calls this (not blockDim.x: that is a load from the dispatch packet, and with
     * the trip count known both table loads are in flight together): two entries per thread */
    static_assert(PM_EXP_N == 512, "two table entries per thread of a 256-thread workgroup");
    const double t0 = PM_EXP_TAB_C[threadIdx.x], t1 = PM_EXP_TAB_C[threadIdx.x + 256];
    pm_lds_tab()[threadIdx.x] = pm_tab_bias(t0, threadIdx.x);
    pm_lds_tab()[threadIdx.x + 256] = pm_tab_bias(t1, threadIdx.x + 256);
    __syncthreads();
}
#define PM_EXP_TAB(j) (pm_from_bits(pm_bits(pm_lds_tab()[(j)]) + ((uint64_t)((unsigned)(j) << 11) << 32)))   /* the entry itself (rare paths) */
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
/* expm1(r) = r + r² (1/2 + r/6 + r²/24) for |r| <= ln2/1024, in Estrin's form: four issue slots like Horner's, but every
 * instruction has at most ONE constant that is not an inline operand — on the device it rides in a scalar register pair, where
 * Horner's first step fma(1/24, r, 1/6) wants one of its two in a vector register pair (kept for the whole kernel, or copied at
 * every use: measured, one v_mov_b64 per exponential) — and the dependent chain is one step shorter */
PM_HD double pm_expm1_poly(double r)
{
    const double r2 = r * r;
    const double a = PM_FMA(r, 1.6666666666666666e-01, 0.5);
    const double b = PM_FMA(r2, 4.1666666666666664e-02, a);
    return PM_FMA(r2, b, r);
}

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
    double q = pm_expm1_poly(r);                      /* expm1(r): small, so T + T q rounds once */
    double T = PM_EXP_TAB(j);
    return __builtin_ldexp(PM_FMA(T, q, T), m);
}

/* The clamps are the identity for |x| <= 700.  On the device they sit behind a wave-uniform test
 * (one compare instead of two compare+select pairs when every lane of the wave is in range — the
 * normal case); the result is the same bit pattern either way. */
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_WAVE_ALL(c) __all(c)
#define PM_RARE_PATH() asm volatile("; PM_RARE_PATH")   /* keeps the rare path a real branch (no if-conversion into selects); the comment marks it for scripts/isa_budget.py */
/* a value computed on a rare path, made opaque: the optimiser cannot merge the rare expression with the common one it replaces
 * (it would turn "compute, then overwrite on the rare path" into a select of the operands and a copy on the common path) */
#define PM_RARE_VALUE(x) asm volatile("" : "+v"(x))
#else
#define PM_WAVE_ALL(c) (c)
#define PM_RARE_PATH() ((void)0)
#define PM_RARE_VALUE(x) ((void)0)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
/* exp(x), x = -|a| if NEGABS else a, for |x| <= 700 (or a NaN): the bits of pm_exp_core(x) in fewer issue slots.
 *  - 2^m enters through the high dword of the (biased) table entry — one v_lshl_add_u32 — before the last fma instead of a shift
 *    and a v_ldexp_f64 after it: for a normal result (|x| <= 700) scaling by 2^m commutes with that rounding;
 *  - NEGABS: the sign and the absolute value ride on the operands of the two instructions that read x (source modifiers), so
 *    exp(-|a|) costs what exp(a) costs.
 * tests/test_gpu_pmath.py holds both forms against the host's pm_exp bit for bit. */
template <bool NEGABS>
__device__ __forceinline__ double pm_exp_plain(double a, double sh)   /* sh: the shifter in a vector register pair (PM_EXP_SHIFTER()) */
{
    const double x = NEGABS ? -__builtin_fabs(a) : a;
    /* kd = fma(x, 512/ln2, shifter), written out: the three-address form with the shifter read in place (the compiler's
     * two-address v_fmac copies it first) */
    double kd;
    if (NEGABS) __asm__("v_fma_f64 %0, -|%1|, %2, %3" : "=v"(kd) : "v"(a), "s"(PM_EXP_RN), "v"(sh));
    else __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(kd) : "v"(a), "s"(PM_EXP_RN), "v"(sh));
    const double k = kd - 6755399441055744.0;
    double r;
    if (NEGABS) __asm__("v_fma_f64 %0, -%1, %2, -|%3|" : "=v"(r) : "v"(k), "s"(PM_EXP_LHI), "v"(a));   /* |a| is never a value of its own */
    else r = PM_FMA(-k, PM_EXP_LHI, x);
    r = PM_FMA(-k, PM_EXP_LLO, r);
    const uint32_t ki = (uint32_t)pm_bits(kd);
    const double q = pm_expm1_poly(r);
    const uint64_t tb = pm_bits(pm_lds_tab()[ki & (PM_EXP_N - 1)]);
    uint32_t hi;
    __asm__("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(hi) : "v"(ki), "n"(20 - PM_EXP_SHIFT), "v"((uint32_t)(tb >> 32)));
    const double T = __hiloint2double((int)hi, (int)(uint32_t)tb);
    return PM_FMA(T, q, T);
}
#endif

/* The callers inside the Runge-Kutta loop create the shifter once (PM_EXP_SHIFTER(): an opaque, loop-invariant vector register
 * pair) and hand it to every exponential; the one-argument forms create their own.  The host ignores it. */
#define PM_EXP_SHIFTER() pm_vc(6755399441055744.0)

PM_HD double pm_exp_sh(double x, double sh)
{
    if (!PM_WAVE_ALL(pm_fabs(x) <= 700.0)) {
        PM_RARE_PATH();
        x = (x > 710.0) ? 710.0 : x;
        x = (x < -746.0) ? -746.0 : x;
        return pm_exp_core(x);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return pm_exp_plain<false>(x, sh);
#else
    return pm_exp_core(x);
#endif
}
PM_HD double pm_exp(double x) { return pm_exp_sh(x, PM_EXP_SHIFTER()); }

/* exp(min(x, 700)): never overflows (callers that multiply the result by a possible zero) */
PM_HD double pm_exp_finite(double x)
{
    if (!PM_WAVE_ALL(pm_fabs(x) <= 700.0)) {
        PM_RARE_PATH();
        x = (x > 700.0) ? 700.0 : x;
        x = (x < -746.0) ? -746.0 : x;
        return pm_exp_core(x);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return pm_exp_plain<false>(x, PM_EXP_SHIFTER());
#else
    return pm_exp_core(x);
#endif
}

/* exp(x) for callers that guarantee |x| <= 700 or x NaN: same bits as pm_exp, no clamps
 * (a NaN reaches the table through a masked index and propagates through the polynomial) */
#if defined(__HIP_DEVICE_COMPILE__)
PM_HD double pm_exp_bounded(double x) { return pm_exp_plain<false>(x, PM_EXP_SHIFTER()); }
#else
PM_HD double pm_exp_bounded(double x) { return pm_exp_core(x); }
#endif

/* exp(-|a|) for a caller that guarantees |a| <= 700 (or a NaN): the bits of pm_exp(-fabs(a)), no range test */
PM_HD double pm_exp_negabs_inrange(double a, double sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return pm_exp_plain<true>(a, sh);
#else
    return pm_exp_core(-pm_fabs(a));
#endif
}
/* exp(x) for a caller that guarantees |x| <= 700 (or a NaN), shifter handed in */
PM_HD double pm_exp_inrange(double x, double sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return pm_exp_plain<false>(x, sh);
#else
    return pm_exp_core(x);
#endif
}
/* SATURATING exponential exp(min(max(x, -700), 700)): in [1e-304, 1e304], never 0, denormal or inf; a NaN propagates.  For the
 * right-hand side, where e² beyond e^±700 has no meaning either way: the out-of-range side is two selects in front of the SAME
 * evaluation (no second copy of the exponential behind the range test, as pm_exp has for its ldexp form) */
PM_HD double pm_exp_sat(double x, double sh)
{
    if (!PM_WAVE_ALL(pm_fabs(x) <= 700.0)) {
        PM_RARE_PATH();
        x = (x > 700.0) ? 700.0 : x;
        x = (x < -700.0) ? -700.0 : x;
        PM_RARE_VALUE(x);
    }
    return pm_exp_inrange(x, sh);
}

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

/* fma(a, b, a) as ONE three-address v_fma_f64.  The seed of pm_rsqrt is (0, hi) with the zero a register shared by every seed
 * in the kernel: the compiler's two-address form (v_fmac) would first copy the pair — one more issue slot per reciprocal root. */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double pm_fma_aba(double a, double b)
{
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
PM_HD double pm_fma_aba(double a, double b) { return PM_FMA(a, b, a); }
#endif

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
    e = PM_FMA(-hx, y * y, 0.5); y = pm_fma_aba(y, e);     /* y + y e */
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
