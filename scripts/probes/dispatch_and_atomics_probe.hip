// Workgroup dispatch rate, an LDS table fill, chains of dependent loads and returned atomics on one / 64 addresses, as functions of the grid
// size (lab tool; hipcc -O3 --offload-arch=gfx950 scripts/probes/dispatch_and_atomics_probe.hip -o probe; results: profiles/r4_dispatch_and_atomics_probe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_empty(int *p) { if (p == (int *)1) *p = 0; }
__global__ void __launch_bounds__(256) k_lds(const double *src, int *p) { __shared__ double tab[512]; tab[threadIdx.x] = src[threadIdx.x]; tab[threadIdx.x + 256] = src[threadIdx.x + 256]; __syncthreads(); if (tab[(threadIdx.x * 7) & 511] == 12345.0) *p = 1; }
__global__ void __launch_bounds__(256) k_chain(const int *idx, int depth, int *p) { int v = blockIdx.x * 256 + threadIdx.x; for (int d = 0; d < depth; d++) v = idx[v]; if (v == -1) *p = 1; }
__global__ void __launch_bounds__(256) k_atomic(int *cnt, int *p) { if ((threadIdx.x) == 0) { int v = atomicAdd(cnt, 1); if (v == -1) *p = 1; } }
__global__ void __launch_bounds__(256) k_atomic64(int *cnt, int *p) { if ((threadIdx.x) == 0) { int v = atomicAdd(cnt + 16 * (blockIdx.x & 63), 1); if (v == -1) *p = 1; } }
#define T(name, launch) do { for (int w = 0; w < 3; w++) { launch; } hipEventRecord(e0); for (int r = 0; r < 20; r++) { launch; } hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-34s %8.1f us per launch\n", name, ms * 1000 / 20); } while (0)
int main() {
    const int N = 16384 * 256;
    int *idx, *p, *cnt; double *src;
    hipMalloc(&idx, N * 4); hipMalloc(&p, 4); hipMalloc(&cnt, 4096 * 4); hipMalloc(&src, 4096);
    int *h = (int *)malloc(N * 4); for (int i = 0; i < N; i++) h[i] = (int)(((long long)i * 2654435761ll + 12345) % N); hipMemcpy(idx, h, N * 4, hipMemcpyHostToDevice);
    hipMemset(cnt, 0, 4096 * 4); hipMemset(src, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int g : {2048, 8192, 16384, 65536}) {
        printf("grid %d x 256\n", g);
        T("empty", hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, 0, p));
        T("LDS table fill + barrier", hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 0, 0, src, p));
        if (g <= 16384) { T("chain of 4 dependent loads", hipLaunchKernelGGL(k_chain, dim3(g), dim3(256), 0, 0, idx, 4, p));
        T("chain of 8 dependent loads", hipLaunchKernelGGL(k_chain, dim3(g), dim3(256), 0, 0, idx, 8, p)); }
        T("one returning atomic, one address", hipLaunchKernelGGL(k_atomic, dim3(g), dim3(256), 0, 0, cnt, p));
        T("one returning atomic, 64 addresses", hipLaunchKernelGGL(k_atomic64, dim3(g), dim3(256), 0, 0, cnt, p));
    }
    return 0;
}
