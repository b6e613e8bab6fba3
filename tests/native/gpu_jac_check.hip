// Device-vs-host bit comparison of the structured Jacobian of the specialised RHS (physics.h rhs3_jac_plain) and of the RHS itself
// (rhs3) on plain particles: built and run by tests/test_gpu_pmath.py.  The whole-step parity tests show it through the Rosenbrock23
// attempts of the default solver; this program shows it entry by entry, with and without the per-node metric term and the wind's slope.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "physics.h"

struct In { double L, cx, cy, u, v, pc, du, dv; };
struct Out { double J[9], dT[3], f[3], y; };

template <bool METRIC, bool TV>
PM_HD void eval(const KParams &P, const In &a, Out &o)
{
    WindD W;
    W.sh = PM_EXP_SHIFTER();
    wind_derive(P, a.u, a.v, W);
    Vec3 dT = {0.0, 0.0, 0.0};
    o.y = rhs3_jac_plain<METRIC, TV>(P, a.L, a.cx, a.cy, W, a.pc, a.du, a.dv, o.J, dT);
    o.dT[0] = dT.lne; o.dT[1] = dT.cx; o.dT[2] = dT.cy;
    Vec3 f;
    rhs3<true, METRIC>(P, a.L, a.cx, a.cy, W, f, a.pc);
    o.f[0] = f.lne; o.f[1] = f.cx; o.f[2] = f.cy;
}

template <bool METRIC, bool TV>
__global__ void k_eval(KParams P, int64_t n, const In *in, Out *out)
{
    pm_device_init();
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) eval<METRIC, TV>(P, in[i], out[i]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main()
{
    KParams P;
    memset(&P, 0, sizeof(P));
    P.r_g = 0.85; P.inv_rg = 1.0 / P.r_g; P.C_alpha = -1.41; P.C_phi = 0.04; P.C_e = 2.16e-4;
    P.p = 0.75; P.n = 2.0; P.neg2p = -1.5; P.inv_eT = 1.0 / 0.6;
    P.propagation = P.input = P.dissipation = P.peak_shift = P.direction = P.n_is_2 = P.p_is_075 = P.fast_phys = 1;
    {
        const double k1 = 0.25 * PK_G0, k2 = k1 * k1, K = k2 * k2;
        const double i2 = P.inv_eT * P.inv_eT;
        P.inv_eT4 = i2 * i2; P.half_inv_rg = 0.5 * P.inv_rg; P.two_inv_rg2 = 2.0 * (P.inv_rg * P.inv_rg);
        P.KeT4 = K * P.inv_eT4; P.KrCa = (K * P.r_g) * P.C_alpha; P.Cdir = P.C_phi * P.two_inv_rg2;
        P.rg2 = P.r_g * P.r_g;
        const double rg4 = P.rg2 * P.rg2, rg8 = rg4 * rg4;
        P.Cw = (0.5 * PK_G0) * P.r_g; P.Chrh = -0.25 * P.r_g; P.ymax = 10.0 / P.r_g; P.sgmax = 1e8 / P.rg2;
        P.KeT4y = P.KeT4 * rg8; P.KrCay = P.KrCa * rg8; P.Cs = (0.5 * P.C_phi) * P.rg2; P.Cdir2 = 2.0 * P.C_phi;
        P.g4rg2 = k1 * P.rg2; P.qU2r_max = 249999.0 / (P.ymax * P.ymax);
    }
    const int64_t N = 1 << 20;
    std::mt19937_64 g(20261005);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    std::vector<In> in(N);
    for (auto &a : in) {
        // half of the states ordinary seas, half of them slow young seas just above the speed floor under winds from a breeze down to nothing
        const bool slow = (g() & 1) != 0;
        const double c = slow ? 0.086 + 0.3 * u01(g) : 0.2 + 12.0 * u01(g), th = 6.283185307179586 * u01(g);
        const double U = slow ? std::pow(10.0, -9.0 + 9.5 * u01(g)) : 0.05 + 30.0 * u01(g), tw = th + 2.0 * (u01(g) - 0.5) * ((g() & 3) ? 0.6 : 3.0);
        a.L = slow ? -20.0 + 12.0 * u01(g) : -18.0 + 21.0 * u01(g); a.cx = c * std::cos(th); a.cy = c * std::sin(th);
        a.u = U * std::cos(tw); a.v = U * std::sin(tw);
        a.pc = 1e-6 * (u01(g) - 0.5); a.du = 1e-3 * (u01(g) - 0.5); a.dv = 1e-3 * (u01(g) - 0.5);
    }
    In *din; Out *dout;
    CK(hipMalloc(&din, N * sizeof(In))); CK(hipMalloc(&dout, N * sizeof(Out)));
    CK(hipMemcpy(din, in.data(), N * sizeof(In), hipMemcpyHostToDevice));
    std::vector<Out> out(N);
    int bad_total = 0;
    for (int var = 0; var < 4; var++) {
        dim3 grid((unsigned)((N + 255) / 256)), block(256);
        switch (var) {
        case 0: hipLaunchKernelGGL((k_eval<false, false>), grid, block, 0, 0, P, N, din, dout); break;
        case 1: hipLaunchKernelGGL((k_eval<true, false>), grid, block, 0, 0, P, N, din, dout); break;
        case 2: hipLaunchKernelGGL((k_eval<false, true>), grid, block, 0, 0, P, N, din, dout); break;
        default: hipLaunchKernelGGL((k_eval<true, true>), grid, block, 0, 0, P, N, din, dout); break;
        }
        CK(hipGetLastError());
        CK(hipMemcpy(out.data(), dout, N * sizeof(Out), hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < N; i++) {
            Out h;
            memset(&h, 0, sizeof(h));
            switch (var) {
            case 0: eval<false, false>(P, in[i], h); break;
            case 1: eval<true, false>(P, in[i], h); break;
            case 2: eval<false, true>(P, in[i], h); break;
            default: eval<true, true>(P, in[i], h); break;
            }
            if (memcmp(&h, &out[i], sizeof(Out)) != 0) {
                if (bad < 3) {
                    const double *x = (const double *)&h, *y = (const double *)&out[i];
                    for (int k = 0; k < 16; k++) if (memcmp(&x[k], &y[k], 8)) fprintf(stderr, "variant %d particle %lld entry %d: host %a device %a\n", var, (long long)i, k, x[k], y[k]);
                }
                bad++;
            }
        }
        printf("jacobian metric=%d tvar=%d %lld states, %lld mismatches\n", var & 1, var >> 1, (long long)N, (long long)bad);
        bad_total += bad != 0;
    }
    (void)hipFree(din); (void)hipFree(dout);
    return bad_total ? 1 : 0;
}
