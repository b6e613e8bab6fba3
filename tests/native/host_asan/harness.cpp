// harness.cpp — seeded random programs of C-ABI calls against the HOST side of libpicles_hip.so, built with AddressSanitizer on
// top of fake_hip.cpp (device memory = heap memory, kernels = no-ops).  Every buffer handed across the ABI is allocated with
// exactly the documented size (own rows: Nx * (j_end - j_begin)), so a copy that assumes any other size — whole grid instead of
// slab, old halo geometry after picles_set_halo_rows, a sample count that moved between two calls — is an ASan report.
// Scenarios mirror what the GPU fuzzers do around the kernels (tests/test_gpu_fuzz.py, _hostile, _sequences, _slab_fuzz,
// _native_ring, _loopback_ring): whole-grid and slab contexts, every getter / setter, split and fused phases, halo resizing,
// snapshot ring, timing samples, generic scatter, wind forms, the native ring through the thread loopback communicator.
// TEST INFRASTRUCTURE ONLY (tests/test_host_asan.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/picles_hip.h"

namespace {
struct Rng {
    std::mt19937_64 g;
    explicit Rng(uint64_t s) : g(s) {}
    int in(int lo, int hi) { return lo + (int)(g() % (uint64_t)(hi - lo + 1)); }      /* inclusive */
    bool coin() { return g() & 1; }
    double uni(double a, double b) { return a + (b - a) * (double)(g() >> 11) * (1.0 / 9007199254740992.0); }
};

/* exact-size heap buffers: ASan's red zones sit right behind the last element */
template <class T> struct Buf {
    T *p;
    size_t n;
    explicit Buf(size_t n_) : p((T *)malloc((n_ ? n_ : 1) * sizeof(T))), n(n_) { memset(p, 0, (n_ ? n_ : 1) * sizeof(T)); }
    ~Buf() { free(p); }
    Buf(const Buf &) = delete;
};

long g_calls = 0, g_errors_expected = 0;
void expect_ok(picles_ctx *c, int rc, const char *what)
{
    g_calls++;
    if (rc != 0) {
        fprintf(stderr, "harness: %s failed rc=%d: %s\n", what, rc, picles_last_error(c));
        exit(3);
    }
}
/* calls that may legitimately refuse (documented argument validation): must fail cleanly, never fault */
void may_refuse(int rc) { g_calls++; if (rc != 0) g_errors_expected++; }

void fill_structs(Rng &R, int Nx, int Ny, picles_grid &g, picles_phys &p, picles_ode &o, picles_model &m)
{
    memset(&g, 0, sizeof(g)); memset(&p, 0, sizeof(p)); memset(&o, 0, sizeof(o)); memset(&m, 0, sizeof(m));
    g.Nx = Nx; g.Ny = Ny; g.dx = 2000.0; g.dy = 1500.0;
    g.periodic_x = R.coin(); g.periodic_y = R.coin();
    p.r_g = 0.85; p.C_alpha = -1.41; p.C_phi = R.coin() ? 0.04 : 1.81e-5; p.C_e = 2.2117647058823533e-4; p.g = 9.81;
    p.gamma = 0.88; p.q = R.in(0, 5) ? -0.25 : -0.3; p.c_beta = 0.04; p.c_D = 2e-3; p.c_e = 1.3e-6; p.c_alpha = 11.8;
    p.propagation = p.input = p.dissipation = p.peak_shift = p.direction = 1;
    if (!R.in(0, 4)) p.direction = 0;
    o.abstol = 1e-4; o.reltol = 1e-3; o.dt0 = 1e-3; o.dtmin = 1e-4; o.force_dtmin = 1; o.solver = R.in(0, 2); o.maxiters = 10000;
    o.log_energy_minimum = -13.0; o.log_energy_maximum = 3.3; o.wind_min_squared = 4.0; o.timestep = 600.0;
    m.periodic_boundary = R.coin(); m.init_type = R.in(0, 3) ? 0 : 1;
    m.default_particle[0] = -11.0; m.default_particle[1] = 1e-3; m.default_particle[2] = 0.0;
    m.minimal_state[0] = 1.25e-6; m.minimal_state[1] = 1.28e-9;
}

void set_some_winds(Rng &R, picles_ctx *c, size_t N, int Nx, int ny, double t, double dt)
{
    Buf<double> u0(N), v0(N), um(N), vm(N), u1(N), v1(N);
    for (size_t k = 0; k < N; k++) { u0.p[k] = R.uni(-12, 12); v0.p[k] = R.uni(-12, 12); um.p[k] = u0.p[k] + 0.1; vm.p[k] = v0.p[k]; u1.p[k] = u0.p[k] + 0.3; v1.p[k] = v0.p[k] - 0.2; }
    switch (R.in(0, 3)) {
    case 0: expect_ok(c, picles_set_winds(c, u0.p, v0.p, t, nullptr, nullptr, t), "set_winds static"); break;
    case 1: expect_ok(c, picles_set_winds(c, u0.p, v0.p, t, u1.p, v1.p, t + dt), "set_winds 2"); break;
    case 2: expect_ok(c, picles_set_winds3(c, u0.p, v0.p, t, um.p, vm.p, u1.p, v1.p, t + dt), "set_winds3"); break;
    default: {
        int nx = R.in(2, 6), nyl = R.in(2, 5), nt = R.in(2, 5);
        Buf<double> gu((size_t)nx * nyl * nt), gv((size_t)nx * nyl * nt);
        for (size_t k = 0; k < gu.n; k++) { gu.p[k] = R.uni(-10, 10); gv.p[k] = R.uni(-10, 10); }
        expect_ok(c, picles_set_wind_grid(c, nx, nyl, nt, 0.0, 2000.0 * Nx / (nx - 1), 0.0, 1500.0 * 40 / (nyl - 1), 0.0, 3000.0, gu.p, gv.p, 0.0, 0.0), "set_wind_grid");
    }
    }
    Buf<double> a(N), b(N), d(N), e(N);
    expect_ok(c, picles_get_winds(c, a.p, R.coin() ? b.p : nullptr, d.p, e.p), "get_winds");
    may_refuse(picles_get_winds_mid(c, a.p, b.p));      /* returns 1 for two-level winds */
}

/* one context: rows [j0, j1) of an Nx x Ny grid */
void run_program(Rng &R, int Nx, int Ny, int j0, int j1, int halo, bool whole)
{
    picles_grid g; picles_phys p; picles_ode o; picles_model m;
    fill_structs(R, Nx, Ny, g, p, o, m);
    g.j_begin = j0; g.j_end = j1;
    Buf<int8_t> mask((size_t)Nx * Ny);
    if (R.coin()) {
        for (size_t k = 0; k < mask.n; k++) mask.p[k] = (int8_t)(R.in(0, 9) ? 1 : (R.coin() ? 0 : (R.coin() ? 2 : 3)));
        g.mask = mask.p;
    }
    picles_ctx *c = nullptr;
    int rc = picles_create(&g, &p, &o, &m, 0, halo, &c);
    g_calls++;
    if (rc != 0) { g_errors_expected++; (void)picles_last_error(nullptr); return; }      /* e.g. a slab with fewer rows than halo_rows */
    const int ny = j1 - j0;
    const size_t N = (size_t)Nx * ny;
    const double dt = 600.0;
    if (whole && !R.in(0, 5)) may_refuse(picles_set_slab_mode(c, 1));
    if (!R.in(0, 3)) {
        Buf<double> a(N), b(N), d(N);
        for (size_t k = 0; k < N; k++) { a.p[k] = 1e-3; b.p[k] = 2e-3; d.p[k] = 1e-9; }
        expect_ok(c, picles_set_metric(c, a.p, b.p, d.p), "set_metric");
        if (R.coin()) expect_ok(c, picles_set_metric(c, nullptr, nullptr, nullptr), "set_metric NULL");
    }
    set_some_winds(R, c, N, Nx, ny, 0.0, dt);
    expect_ok(c, picles_seed(c, 0.0), "seed");
    bool store = false, ring = false;
    const int nops = R.in(8, 30);
    for (int k = 0; k < nops; k++) {
        const bool single = picles_halo_rows(c) >= 1 && whole;
        switch (R.in(0, 21)) {
        case 0: if (single) may_refuse(picles_time_step(c, dt, R.in(0, 7))); break;      /* ATOMIC refuses slab mode / tripolar */
        case 1: if (single) may_refuse(picles_run_steps(c, dt, R.in(0, 4))); break;
        case 2: if (single) { may_refuse(picles_advance(c, dt, R.in(0, 1))); may_refuse(picles_remesh(c, dt)); expect_ok(c, picles_tick(c, dt), "tick"); } break;
        case 3: expect_ok(c, picles_zero_state(c), "zero_state"); break;
        case 4: { Buf<double> s(3 * N); expect_ok(c, picles_get_state(c, s.p), "get_state"); if (R.coin()) expect_ok(c, picles_set_state(c, s.p), "set_state"); } break;
        case 5: { Buf<double> s(3 * N); expect_ok(c, picles_get_movie_state(c, s.p), "get_movie_state"); } break;
        case 6: {
            Buf<double> z(5 * N); Buf<uint8_t> on(N), bd(N); Buf<int32_t> st(N);
            expect_ok(c, picles_get_particles(c, R.coin() ? z.p : nullptr, R.coin() ? on.p : nullptr, R.coin() ? bd.p : nullptr, R.coin() ? st.p : nullptr), "get_particles");
            if (R.coin()) expect_ok(c, picles_set_particles(c, z.p, R.coin() ? on.p : nullptr), "set_particles");
        } break;
        case 7: {
            picles_counters cn; expect_ok(c, picles_get_counters(c, &cn), "get_counters"); if (R.coin()) expect_ok(c, picles_reset_counters(c), "reset_counters");
            int n = picles_get_dispatch_order(c, nullptr, 0);       /* (the fake device runs no kernels: no order is ever complete) */
            if (n < 0) { fprintf(stderr, "dispatch_order rc=%d\n", n); exit(3); }
            int cap = R.in(0, 6);
            Buf<int32_t> ord((size_t)cap);
            if (picles_get_dispatch_order(c, cap ? ord.p : nullptr, cap) < 0) { fprintf(stderr, "dispatch_order(cap) failed\n"); exit(3); }
        } break;
        case 8: expect_ok(c, picles_enable_timing(c, R.coin()), "enable_timing"); break;
        case 9: {
            picles_timing t; expect_ok(c, picles_get_timing(c, &t), "get_timing");
            for (int kind = 0; kind < 3; kind++) {
                int n = picles_get_timing_samples(c, kind, nullptr, 0);
                if (n < 0) { fprintf(stderr, "timing_samples rc=%d\n", n); exit(3); }
                int cap = R.coin() ? n : R.in(0, n + 3);
                Buf<double> out((size_t)cap);
                int n2 = picles_get_timing_samples(c, kind, cap ? out.p : nullptr, cap);
                if (n2 != n) { fprintf(stderr, "timing sample count moved %d -> %d\n", n, n2); exit(3); }
            }
        } break;
        case 10: if (!store) { expect_ok(c, picles_store_init(c, R.in(1, 3)), "store_init"); store = true; } break;
        case 11: if (store) {
            may_refuse(picles_store_push(c));      /* ring full: refuses */
            if (R.coin() && picles_store_pending(c) > 0) { Buf<double> s(3 * N); double t; expect_ok(c, picles_store_pop(c, s.p, R.coin() ? &t : nullptr), "store_pop"); }
        } break;
        case 12: {      /* plain slab phases, halo blocks copied to ourselves */
            expect_ok(c, picles_begin_step(c, dt, R.in(0, 3)), "begin_step");
            expect_ok(c, picles_advance_rows(c, PICLES_ROWS_EDGE, nullptr), "advance_rows edge");
            for (int side = 0; side < 2; side++) {
                void *sp, *rp; size_t sb = 0, rb = 0;
                int r1 = picles_halo_send_dev(c, side, &sp, &sb), r2 = picles_halo_recv_dev(c, 1 - side, &rp, &rb);
                g_calls += 2;
                if (r1 || r2) { g_errors_expected++; continue; }      /* a whole-grid context given more halo rows than it has rows: refused */
                if (sb != rb || sb != (size_t)picles_halo_rows(c) * 6 * Nx * 8) { fprintf(stderr, "halo block size\n"); exit(3); }
                memmove(rp, sp, sb);      /* "device" memory is heap memory here: ASan checks both blocks lie inside the records */
            }
            expect_ok(c, picles_advance_rows(c, PICLES_ROWS_INTERIOR, nullptr), "advance_rows interior");
            expect_ok(c, picles_scatter_remesh(c, nullptr), "scatter_remesh");
        } break;
        case 13: {      /* fused slab phases */
            int f = picles_begin_fused_step(c, dt);
            g_calls++;
            if (f < 0) { fprintf(stderr, "begin_fused_step rc=%d %s\n", f, picles_last_error(c)); exit(3); }
            if (f == 0) {
                expect_ok(c, picles_step_rows(c, PICLES_ROWS_EDGE, nullptr), "step_rows edge");
                void *sp, *rp; size_t sb, rb;
                int r1 = picles_halo_send_dev(c, 1, &sp, &sb), r2 = picles_halo_recv_dev(c, 0, &rp, &rb);
                g_calls += 2;
                if (r1 == 0 && r2 == 0) memmove(rp, sp, sb); else g_errors_expected++;
                expect_ok(c, picles_step_rows(c, PICLES_ROWS_INTERIOR, nullptr), "step_rows interior");
                expect_ok(c, picles_end_fused_step(c), "end_fused_step");
            }
        } break;
        case 14: may_refuse(picles_set_halo_rows(c, R.in(0, 6))); break;      /* 0 and too many rows refuse */
        case 15: if (single) {
            int64_t np = R.in(0, 200);
            Buf<int32_t> ij(2 * (size_t)np); Buf<double> xy(2 * (size_t)np), ch(3 * (size_t)np);
            for (int64_t q = 0; q < np; q++) { ij.p[q] = R.in(-2, Nx + 1); ij.p[np + q] = R.in(-2, Ny + 1); xy.p[q] = R.uni(-3, 3); xy.p[np + q] = R.uni(-3, 3); }
            may_refuse(picles_scatter_particles(c, np, np ? ij.p : nullptr, np ? xy.p : nullptr, np ? ch.p : nullptr));
        } break;
        case 16: set_some_winds(R, c, N, Nx, ny, picles_clock(c), dt); break;
        case 17: expect_ok(c, picles_sync(c), "sync"); break;
        case 18: if (!ring) {      /* native ring of one through the loopback communicator */
            Buf<char> id(PICLES_SLAB_ID_BYTES);
            expect_ok(c, picles_slab_unique_id(id.p), "slab_unique_id");
            rc = picles_slab_comm_init(c, id.p, 0, 1);
            g_calls++;
            if (rc == 0) ring = true; else g_errors_expected++;
        } break;
        case 19: if (ring) {
            may_refuse(picles_slab_run_steps(c, dt, R.in(0, 3), R.in(0, 3)));
            if (R.coin()) expect_ok(c, picles_slab_exchange(c), "slab_exchange");
            void *e, *mm; expect_ok(c, picles_slab_streams(c, &e, &mm), "slab_streams");
        } break;
        case 20: if (ring && !R.in(0, 3)) { expect_ok(c, picles_slab_comm_destroy(c), "slab_comm_destroy"); ring = false; } break;
        default: if (single) may_refuse(picles_time_step(c, dt, PICLES_STEP_ZERO_FIRST)); break;
        }
    }
    expect_ok(c, picles_destroy(c), "destroy");
}

/* the native ring with `world` thread-ranks, one y-slab each (tests/test_gpu_loopback_ring.py without the kernels) */
void run_ring(uint64_t seed, int world)
{
    Rng R(seed);
    const int Nx = R.in(6, 30), halo = R.in(1, 3), rows = R.in(halo + 2, 10), Ny = rows * world + R.in(0, world - 1);
    picles_grid g0; picles_phys p; picles_ode o; picles_model m;
    fill_structs(R, Nx, Ny, g0, p, o, m);
    char id[PICLES_SLAB_ID_BYTES];
    if (picles_slab_unique_id(id) != 0) { fprintf(stderr, "unique id\n"); exit(3); }
    const int n_steps = R.in(1, 4), flags = R.coin() ? PICLES_STEP_ZERO_FIRST : R.in(0, 3);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            picles_grid g = g0;
            int base = Ny / world, rem = Ny % world;
            g.j_begin = r * base + (r < rem ? r : rem);
            g.j_end = g.j_begin + base + (r < rem ? 1 : 0);
            picles_ctx *c = nullptr;
            int rc = picles_create(&g, &p, &o, &m, 0, halo, &c);
            /* every rank must reach comm_init or none: creation fails for all ranks alike only through halo vs Ny; per-rank
             * row counts are >= halo + 2 here */
            if (rc != 0) { fprintf(stderr, "ring create rc=%d: %s\n", rc, picles_last_error(nullptr)); exit(3); }
            const size_t N = (size_t)Nx * (g.j_end - g.j_begin);
            Buf<double> u(N), v(N);
            for (size_t k = 0; k < N; k++) { u.p[k] = 9.0; v.p[k] = 4.0; }
            expect_ok(c, picles_set_winds(c, u.p, v.p, 0.0, nullptr, nullptr, 0.0), "ring set_winds");
            expect_ok(c, picles_seed(c, 0.0), "ring seed");
            expect_ok(c, picles_slab_comm_init(c, id, r, world), "ring comm_init");
            expect_ok(c, picles_slab_exchange(c), "ring exchange");
            expect_ok(c, picles_slab_run_steps(c, 600.0, n_steps, flags), "ring run_steps");
            Buf<double> s(3 * N);
            expect_ok(c, picles_get_state(c, s.p), "ring get_state");
            expect_ok(c, picles_set_halo_rows(c, halo + 1), "ring set_halo_rows");      /* collective by construction: same on every rank */
            expect_ok(c, picles_slab_run_steps(c, 600.0, 1, flags), "ring run_steps 2");
            picles_counters cn; expect_ok(c, picles_get_counters(c, &cn), "ring get_counters");
            expect_ok(c, picles_destroy(c), "ring destroy");
        });
    for (auto &t : th) t.join();
}
}   // namespace

int main(int argc, char **argv)
{
    const uint64_t first = argc > 1 ? strtoull(argv[1], nullptr, 10) : 0, count = argc > 2 ? strtoull(argv[2], nullptr, 10) : 200;
    if (picles_abi_version() != PICLES_ABI_VERSION) { fprintf(stderr, "ABI version\n"); return 2; }
    for (uint64_t seed = first; seed < first + count; seed++) {
        Rng R(0x9E3779B97F4A7C15ull * (seed + 1));
        const int Nx = R.in(4, 40), Ny = R.in(4, 40);
        const int kind = R.in(0, 9);
        if (kind <= 4) run_program(R, Nx, Ny, 0, Ny, R.in(1, 4), true);                 /* whole grid */
        else if (kind <= 7) {                                                           /* one slab of a larger grid */
            int j0 = R.in(0, Ny - 2), j1 = R.in(j0 + 1, Ny);
            run_program(R, Nx, Ny, j0, j1, R.in(1, 4), j0 == 0 && j1 == Ny);
        } else run_ring(seed, R.in(2, 3));
    }
    printf("host-asan harness: %llu scenarios from seed %llu, %ld ABI calls, %ld refused as documented, no sanitizer report\n",
           (unsigned long long)count, (unsigned long long)first, g_calls, g_errors_expected);
    return 0;
}
