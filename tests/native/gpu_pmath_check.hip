// Device-vs-host bit comparison of picles_amd/csrc/pmath.h (test program, built and run by tests/test_gpu_pmath.py).
// The kernels' bitwise parity with the CPU oracle rests on every pmath function producing the same bits on gfx950 and x86-64;
// the whole-step parity tests show that indirectly, this program shows it function by function, and in particular that the
// short division sequences (pm_rcp_plain / pm_div_plain: the compiler's IEEE expansion without range scaling and fix-up) equal
// the host's correctly rounded `/` over the operand ranges their call sites guarantee.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "pmath.h"

enum { F_RCP, F_DIV, F_DIV1E6, F_EXP, F_LOG, F_LOGC, F_RSQRT, F_EXPP, F_EXPS, F_EXPNI, F_EXPI, F_N };
static const char *NAMES[F_N] = {"pm_rcp_plain", "pm_div_plain", "pm_div_1e6", "pm_exp", "pm_log", "pm_log_coarse", "pm_rsqrt",
                                 "pm_exp(in range)", "pm_exp_sat", "pm_exp_negabs_inrange", "pm_exp_inrange"};

PM_HD double apply(int fn, double a, double b)
{
    switch (fn) {
    case F_RCP: return pm_rcp_plain(b);
    case F_DIV: return pm_div_plain(a, b);
    case F_DIV1E6: return pm_div_1e6(a);
    case F_EXP: return pm_exp(a);
    case F_LOG: return pm_log(a);
    case F_LOGC: return pm_log_coarse(a);
    // the plain-range forms of the exponential (device: pm_exp_plain — the 2^m through the biased table entry, the sign and the
    // absolute value as source modifiers); a wave takes them only when ALL its lanes are in range, so these operands come in
    // runs of 64: F_EXPP, F_EXPNI and F_EXPI all in range, F_EXPS (the saturating form) every second wave mixed with out-of-range operands
    case F_EXPP: return pm_exp(a);
    case F_EXPS: return pm_exp_sat(a, PM_EXP_SHIFTER());
    case F_EXPNI: return pm_exp_negabs_inrange(a, PM_EXP_SHIFTER());
    case F_EXPI: return pm_exp_inrange(a, PM_EXP_SHIFTER());
    default: return pm_rsqrt(a);
    }
}

__global__ void k_apply(int fn, int64_t n, const double *a, const double *b, double *out)
{
    pm_device_init();
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = apply(fn, a[i], b[i]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static double pow2(std::mt19937_64 &g, int lo, int hi)   // sign * m * 2^e, m in [1,2), e in [lo, hi]
{
    std::uniform_real_distribution<double> m(1.0, 2.0);
    std::uniform_int_distribution<int> e(lo, hi), s(0, 1);
    return (s(g) ? -1.0 : 1.0) * std::ldexp(m(g), e(g));
}

int main()
{
    const int64_t N = 1 << 22;
    std::mt19937_64 g(20261004);
    std::vector<double> a(N), b(N), out(N);
    double *da, *db, *dout;
    CK(hipMalloc(&da, N * 8)); CK(hipMalloc(&db, N * 8)); CK(hipMalloc(&dout, N * 8));
    int bad_total = 0;
    for (int fn = 0; fn < F_N; fn++) {
        std::uniform_real_distribution<double> u01(0.0, 1.0);
        for (int64_t i = 0; i < N; i++) {
            int kind = (int)(i & 3);
            switch (fn) {
            case F_RCP:
                // the RHS operand (1 + e^h)(1 + t)^2 in [1, 4e304]; the error norm's product of scales >= 1e-20; wide plain range
                if (kind == 0) b[i] = (1.0 + std::exp(700.0 * u01(g))) * (1.0 + 3.0 * u01(g));
                else if (kind == 1) b[i] = std::ldexp(1.0 + u01(g), -67 + (int)(400 * u01(g)));
                else b[i] = pow2(g, -1000, 1000);
                a[i] = 1.0;
                break;
            case F_DIV:
                if (kind == 0) { b[i] = 1.70 + 0.72 * u01(g); a[i] = b[i] - 2.0; }                 // the logarithms' f / (2 + f)
                else if (kind == 1) { b[i] = 1.70 + 0.72 * u01(g); a[i] = std::ldexp(u01(g) - 0.5, -(int)(52 * u01(g))); }
                else { b[i] = pow2(g, -400, 400); a[i] = pow2(g, -400, 400); }
                if (i < 64) a[i] = 0.0;      // +0 (a numerator of -0 would come out as +0: documented in pmath.h, never formed at the call sites)
                break;
            case F_DIV1E6: a[i] = (double)(i % 1000001); b[i] = 0.0; break;
            case F_EXP: a[i] = (kind == 0) ? -746.0 + 1456.0 * u01(g) : ((kind == 1) ? -40.0 * u01(g) : 20.0 * (u01(g) - 0.5)); b[i] = 0.0; break;
            case F_LOG: case F_LOGC: a[i] = (kind == 0) ? std::fabs(pow2(g, -1022, 1023)) : ((kind == 1) ? 1.0 + 0.6 * (u01(g) - 0.5) : std::exp(60.0 * (u01(g) - 0.5))); b[i] = 0.0; break;
            case F_EXPP: a[i] = (kind == 0) ? -700.0 + 1400.0 * u01(g) : ((kind == 1) ? -40.0 * u01(g) : ((kind == 2) ? 20.0 * (u01(g) - 0.5) : pow2(g, -60, 8))); b[i] = 0.0; break;
            case F_EXPS:
                if (((i >> 6) & 1) == 0) a[i] = (kind == 0) ? -700.0 + 1400.0 * u01(g) : ((kind == 1) ? 250.0 * (u01(g) - 0.5) : pow2(g, -60, 8));
                else a[i] = (kind == 0) ? -760.0 + 1520.0 * u01(g) : ((kind == 1) ? pow2(g, 9, 40) : 250.0 * (u01(g) - 0.5));
                if ((i & 1023) == 77) a[i] = (i & 1024) ? 0.0 : -0.0;
                b[i] = 0.0;
                break;
            case F_EXPI: a[i] = (kind == 0) ? -700.0 + 1400.0 * u01(g) : ((kind == 1) ? -40.0 * u01(g) : ((kind == 2) ? 20.0 * (u01(g) - 0.5) : pow2(g, -1022, 8))); b[i] = 0.0; break;
            case F_EXPNI: a[i] = (kind == 0) ? -700.0 + 1400.0 * u01(g) : ((kind == 1) ? 250.0 * (u01(g) - 0.5) : ((kind == 2) ? pow2(g, -1022, 8) : 2.0 * (u01(g) - 0.5))); b[i] = 0.0; break;
            case F_RSQRT: default: a[i] = (kind == 0) ? std::fabs(pow2(g, -1000, 1000)) : 1e-3 + 400.0 * u01(g); b[i] = 0.0; break;
            }
        }
        CK(hipMemcpy(da, a.data(), N * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), N * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_apply, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, fn, N, da, db, dout);
        CK(hipGetLastError());
        CK(hipMemcpy(out.data(), dout, N * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < N; i++) {
            double h = apply(fn, a[i], b[i]);
            // the short divisions are compared with the plain IEEE division as well (the host versions ARE that division)
            if (fn == F_RCP) h = 1.0 / b[i];
            if (fn == F_DIV) h = a[i] / b[i];
            if (fn == F_DIV1E6) h = a[i] / 1e6;
            uint64_t x, y;
            memcpy(&x, &h, 8); memcpy(&y, &out[i], 8);
            if (x != y) {
                if (bad < 5) fprintf(stderr, "%s(%a, %a): host %a device %a\n", NAMES[fn], a[i], b[i], h, out[i]);
                bad++;
            }
        }
        printf("%-14s %lld operands, %lld mismatches\n", NAMES[fn], (long long)N, (long long)bad);
        bad_total += bad != 0;
    }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return bad_total ? 1 : 0;
}
