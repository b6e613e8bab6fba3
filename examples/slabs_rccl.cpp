/*
 * slabs_rccl.cpp — the 4096² BASELINE box cut into y-slabs over all GPUs of one node, driven through the C ABI and
 * RCCL alone (no Python, no torch): the native counterpart of picles_amd/parallel.py and of bench.py --gpus N.
 * One host thread per GPU; per model step and slab
 *
 *     picles_begin_fused_step                       (falls back to begin_step / advance_rows / scatter_remesh)
 *     picles_step_rows(EDGE, stream E)              the 2*halo_rows rows a neighbour needs
 *     ncclGroupStart; ncclSend/ncclRecv of the halo blocks, in place, on stream E; ncclGroupEnd
 *     picles_step_rows(INTERIOR, stream M)          overlaps the exchange
 *     stream M waits for stream E;  picles_end_fused_step
 *
 * The halo blocks are contiguous ranges of the library's own record memory (picles_halo_send_dev /
 * picles_halo_recv_dev), so RCCL reads and writes them without packing.
 *
 *   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I include examples/slabs_rccl.cpp -o /tmp/slabs_rccl -L picles_amd/csrc -lpicles_hip \
 *         -Wl,-rpath,$PWD/picles_amd/csrc -lrccl -lpthread && /tmp/slabs_rccl [n_gpus] [grid_n] [steps]
 *
 * STATUS: compiled in the build container (no GPU there); the single-GPU boxes of the test pool cannot run it.  The
 * same call sequence is what tests/test_gpu_slab_fuzz.py drives (with device-to-device copies in place of RCCL).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "picles_hip.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); exit(3); } } while (0)
#define PIC_OK(ctx, call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, picles_last_error(ctx)); exit(4); } } while (0)

struct Barrier {            /* C++17 has no std::barrier */
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        int gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }
    std::mutex m_; std::condition_variable cv_; int n_, count_ = 0, gen_ = 0;
};

int main(int argc, char **argv)
{
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    const int world = (argc > 1) ? atoi(argv[1]) : ndev;
    const int N = (argc > 2) ? atoi(argv[2]) : 4096;
    const int steps = (argc > 3) ? atoi(argv[3]) : 20;
    const int halo = 2;
    const double DT = 600.0, dx = 2000.0;
    if (world < 1 || world > ndev) { fprintf(stderr, "%d GPUs requested, %d present\n", world, ndev); return 1; }

    std::vector<ncclComm_t> comms(world);
    std::vector<int> devs(world);
    for (int r = 0; r < world; r++) devs[r] = r;
    /* PICLES_RCCL_SELF=1: with one GPU, still open a communicator and run the exchange against ourselves (exercises the
     * RCCL calls on the library's memory; the whole-grid context does not read its ghost rows) */
    const bool use_rccl = world > 1 || (getenv("PICLES_RCCL_SELF") && atoi(getenv("PICLES_RCCL_SELF")) == 1);
    if (use_rccl) NCCL_OK(ncclCommInitAll(comms.data(), world, devs.data()));

    /* bench06 physics (benchmark/bench06_homogenous_box_brenchmarlk.jl:47-126): C_phi = c_beta, gamma 0.88, DP5 */
    const double r_g = 0.85, c_D = 2e-3, c_beta = 4e-2, c_e = 1.3e-6, c_alpha = 11.8, r_w = 2.35, q = -0.25;
    const double C_e = r_w * c_beta * c_D / r_g;
    Barrier bar(world);
    std::vector<double> secs(world, 0.0);
    std::vector<unsigned long long> advanced(world, 0);

    auto worker = [&](int rank) {
        HIP_OK(hipSetDevice(rank));
        const int base = N / world, rem = N % world;
        const int j0 = rank * base + (rank < rem ? rank : rem), j1 = j0 + base + (rank < rem ? 1 : 0);
        picles_grid g = {N, N, dx, dx, 1, 1, nullptr, j0, j1};
        picles_phys ph = {r_g, -1.41, c_beta, C_e, 9.81, 0.88, q, c_beta, c_D, c_e, c_alpha, 1, 1, 1, 1, 1, 0, 0.0};
        picles_ode od = {1e-4, 1e-3, 10.0, 1.0, 1, /*DP5*/ 0, 10000, -13.589885017354083, std::log(27.0), 4.0, 1800.0};
        picles_model md = {1, 0, {0, 0, 0}, {1.253106339976604e-6, 1.2821164e-9}};
        picles_ctx *ctx = nullptr;
        int rc = picles_create(&g, &ph, &od, &md, rank, halo, &ctx);
        if (rc) { fprintf(stderr, "rank %d: picles_create rc=%d: %s\n", rank, rc, picles_last_error(nullptr)); exit(4); }
        const size_t n = (size_t)N * (j1 - j0);
        std::vector<double> u(n, 10.0), v(n, 10.0);
        PIC_OK(ctx, picles_set_winds(ctx, u.data(), v.data(), 0.0, nullptr, nullptr, 0.0));
        PIC_OK(ctx, picles_seed(ctx, 0.0));
        PIC_OK(ctx, picles_sync(ctx));
        hipStream_t sE, sM;
        hipEvent_t evE, evM;
        HIP_OK(hipStreamCreateWithFlags(&sE, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&sM, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&evE, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&evM, hipEventDisableTiming));
        const int prev = (rank + world - 1) % world, next = (rank + 1) % world;      /* periodic in y: a ring */

        auto step = [&]() {
            int fused = picles_begin_fused_step(ctx, DT);
            if (fused < 0) { fprintf(stderr, "begin_fused_step: %s\n", picles_last_error(ctx)); exit(4); }
            if (fused == 1) PIC_OK(ctx, picles_begin_step(ctx, DT, PICLES_STEP_ZERO_FIRST));
            /* the previous step's interior launch (stream M) wrote / read what the edge launch touches */
            HIP_OK(hipEventRecord(evM, sM));
            HIP_OK(hipStreamWaitEvent(sE, evM, 0));
            if (fused == 0) PIC_OK(ctx, picles_step_rows(ctx, PICLES_ROWS_EDGE, sE));
            else PIC_OK(ctx, picles_advance_rows(ctx, PICLES_ROWS_EDGE, sE));
            if (use_rccl) {
                void *s_lo, *s_hi, *r_lo, *r_hi;
                size_t b;
                PIC_OK(ctx, picles_halo_send_dev(ctx, 0, &s_lo, &b));
                PIC_OK(ctx, picles_halo_send_dev(ctx, 1, &s_hi, &b));
                PIC_OK(ctx, picles_halo_recv_dev(ctx, 0, &r_lo, &b));
                PIC_OK(ctx, picles_halo_recv_dev(ctx, 1, &r_hi, &b));
                NCCL_OK(ncclGroupStart());
                NCCL_OK(ncclSend(s_hi, b / 8, ncclDouble, next, comms[rank], sE));     /* my top rows -> below the next slab */
                NCCL_OK(ncclSend(s_lo, b / 8, ncclDouble, prev, comms[rank], sE));     /* my bottom rows -> above the previous slab */
                NCCL_OK(ncclRecv(r_lo, b / 8, ncclDouble, prev, comms[rank], sE));
                NCCL_OK(ncclRecv(r_hi, b / 8, ncclDouble, next, comms[rank], sE));
                NCCL_OK(ncclGroupEnd());
            }
            if (fused == 0) PIC_OK(ctx, picles_step_rows(ctx, PICLES_ROWS_INTERIOR, sM));
            else PIC_OK(ctx, picles_advance_rows(ctx, PICLES_ROWS_INTERIOR, sM));
            HIP_OK(hipEventRecord(evE, sE));
            HIP_OK(hipStreamWaitEvent(sM, evE, 0));                                     /* stream M waits for the halo */
            if (fused == 0) PIC_OK(ctx, picles_end_fused_step(ctx));
            else PIC_OK(ctx, picles_scatter_remesh(ctx, sM));
        };

        for (int k = 0; k < 3; k++) step();                                              /* warm-up (RCCL channels, clocks) */
        PIC_OK(ctx, picles_sync(ctx));
        PIC_OK(ctx, picles_reset_counters(ctx));
        bar.wait();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; k++) step();
        HIP_OK(hipStreamSynchronize(sE));
        HIP_OK(hipStreamSynchronize(sM));
        PIC_OK(ctx, picles_sync(ctx));
        bar.wait();
        secs[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        picles_counters c;
        PIC_OK(ctx, picles_get_counters(ctx, &c));
        advanced[rank] = c.particles_advanced;
        if (c.halo_overflow) fprintf(stderr, "rank %d: %llu particles travelled beyond %d halo rows\n", rank, (unsigned long long)c.halo_overflow, halo);
        picles_destroy(ctx);
    };

    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
    double tmax = 0;
    unsigned long long total = 0;
    for (int r = 0; r < world; r++) { tmax = std::max(tmax, secs[r]); total += advanced[r]; }
    printf("{\"n_gpus\": %d, \"grid\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"particle_steps_per_s\": %.4e}\n",
           world, N, steps, 1e3 * tmax / steps, (double)total / tmax);
    if (use_rccl) for (auto &c : comms) ncclCommDestroy(c);
    return 0;
}
