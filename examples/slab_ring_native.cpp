/*
 * slab_ring_native.cpp — the BASELINE box cut into y-slabs over the GPUs of one node, stepped by the library's NATIVE slab
 * ring (picles_slab_*): no Python, no torch, and no RCCL call in this file — the library binds RCCL itself (dlopen) and
 * issues the kernel launches and the ncclSend/ncclRecv groups of n model steps from ONE call per rank.
 * One host thread per GPU here; a multi-process host (MPI, Julia Distributed) does the same with one rank per process and
 * any means of handing the 128-byte id from rank 0 to the others.
 *
 *   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I include examples/slab_ring_native.cpp -o /tmp/slab_ring_native \
 *         -L picles_amd/csrc -lpicles_hip -Wl,-rpath,$PWD/picles_amd/csrc -lpthread
 *   /tmp/slab_ring_native [n_gpus] [grid_n] [steps]
 *
 * With one GPU the context is put in slab mode (picles_set_slab_mode), so the ring closes on itself and the pull of the edge
 * rows consumes ghost rows that RCCL delivered: the complete multi-GPU data path on the one device.  The program checks its own
 * result: the homogeneous periodic box must carry ONE energy value on every node of every slab.
 *
 * STATUS: built and run with one GPU by tests/test_gpu_more.py; more than one GPU per process has not been available to the
 * builder (single-GPU test boxes).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "picles_hip.h"

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)
#define PIC_OK(ctx, call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, picles_last_error(ctx)); exit(4); } } while (0)

struct Barrier {
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
    const int halo = 2, warm = 3;
    const double DT = 600.0, dx = 2000.0;
    if (world < 1 || world > ndev) { fprintf(stderr, "%d GPUs requested, %d present\n", world, ndev); return 1; }

    unsigned char id[PICLES_SLAB_ID_BYTES];
    if (picles_slab_unique_id(id)) { fprintf(stderr, "picles_slab_unique_id: %s\n", picles_last_error(nullptr)); return 3; }

    /* bench06 physics (benchmark/bench06_homogenous_box_brenchmarlk.jl:47-126): C_phi = c_beta, gamma 0.88, DP5 */
    const double r_g = 0.85, c_D = 2e-3, c_beta = 4e-2, c_e = 1.3e-6, c_alpha = 11.8, r_w = 2.35, q = -0.25;
    const double C_e = r_w * c_beta * c_D / r_g;
    Barrier bar(world);
    std::vector<double> secs(world, 0.0), emin(world, 0.0), emax(world, 0.0);
    std::vector<unsigned long long> advanced(world, 0);

    auto worker = [&](int rank) {
        HIP_OK(hipSetDevice(rank));
        const int base = N / world, rem = N % world;
        const int j0 = rank * base + std::min(rank, rem), j1 = j0 + base + (rank < rem ? 1 : 0);
        picles_grid g = {N, N, dx, dx, 1, 1, nullptr, j0, j1};
        picles_phys ph = {r_g, -1.41, c_beta, C_e, 9.81, 0.88, q, c_beta, c_D, c_e, c_alpha, 1, 1, 1, 1, 1, 0, 0.0};
        picles_ode od = {1e-4, 1e-3, 10.0, 1.0, 1, /*DP5*/ 0, 10000, -13.589885017354083, std::log(27.0), 4.0, 1800.0};
        picles_model md = {1, 0, {0, 0, 0}, {1.253106339976604e-6, 1.2821164e-9}};
        picles_ctx *ctx = nullptr;
        int rc = picles_create(&g, &ph, &od, &md, rank, halo, &ctx);
        if (rc) { fprintf(stderr, "rank %d: picles_create rc=%d: %s\n", rank, rc, picles_last_error(nullptr)); exit(4); }
        if (world == 1) PIC_OK(ctx, picles_set_slab_mode(ctx, 1));          /* ring of one: ghost rows are consumed */
        const size_t n = (size_t)N * (j1 - j0);
        std::vector<double> u(n, 10.0), v(n, 10.0);
        PIC_OK(ctx, picles_set_winds(ctx, u.data(), v.data(), 0.0, nullptr, nullptr, 0.0));
        PIC_OK(ctx, picles_slab_comm_init(ctx, id, rank, world));             /* blocks until every rank has joined */
        PIC_OK(ctx, picles_seed(ctx, 0.0));
        PIC_OK(ctx, picles_sync(ctx));
        PIC_OK(ctx, picles_slab_exchange(ctx));                               /* RCCL builds its channels on first use */
        PIC_OK(ctx, picles_seed(ctx, 0.0));
        PIC_OK(ctx, picles_slab_run_steps(ctx, DT, warm, PICLES_STEP_ZERO_FIRST));
        PIC_OK(ctx, picles_sync(ctx));
        PIC_OK(ctx, picles_reset_counters(ctx));
        bar.wait();
        auto t0 = std::chrono::steady_clock::now();
        PIC_OK(ctx, picles_slab_run_steps(ctx, DT, steps, PICLES_STEP_ZERO_FIRST));   /* ONE call: all steps */
        PIC_OK(ctx, picles_sync(ctx));
        bar.wait();
        secs[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        picles_counters c;
        PIC_OK(ctx, picles_get_counters(ctx, &c));
        advanced[rank] = c.particles_advanced;
        if (c.halo_overflow || c.dropped_nonfinite)
            fprintf(stderr, "rank %d: %llu particles beyond %d halo rows, %llu non-finite\n", rank,
                    (unsigned long long)c.halo_overflow, halo, (unsigned long long)c.dropped_nonfinite);
        std::vector<double> S(3 * n);
        PIC_OK(ctx, picles_get_state(ctx, S.data()));
        emin[rank] = *std::min_element(S.begin(), S.begin() + n);
        emax[rank] = *std::max_element(S.begin(), S.begin() + n);
        picles_destroy(ctx);
    };

    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(worker, r);
    for (auto &t : th) t.join();
    double tmax = 0, lo = emin[0], hi = emax[0];
    unsigned long long total = 0;
    for (int r = 0; r < world; r++) { tmax = std::max(tmax, secs[r]); total += advanced[r]; lo = std::min(lo, emin[r]); hi = std::max(hi, emax[r]); }
    const double spread = (hi - lo) / hi;
    printf("{\"n_gpus\": %d, \"grid\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"particle_steps_per_s\": %.4e, \"e\": %.17g, \"rel_spread\": %.3e}\n",
           world, N, steps, 1e3 * tmax / steps, (double)total / tmax, hi, spread);
    return (lo > 0 && spread < 1e-6) ? 0 : 5;
}
