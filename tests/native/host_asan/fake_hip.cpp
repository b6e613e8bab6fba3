// fake_hip.cpp — a HOST-ONLY stand-in for the HIP runtime, for ONE purpose: running the host side of libpicles_hip.so
// (the C ABI of include/picles_hip.h: allocation sizes, copies across the ABI, re-packing, rings, stores, event / stream
// lifetimes) under AddressSanitizer on a box without a GPU.  "Device" memory is ordinary heap memory, so every
// hipMemcpy / hipMemset the library issues is checked by ASan against the size of BOTH buffers; kernel launches are
// accepted and do nothing (the numbers that come back are meaningless and nothing here looks at them).  Streams and
// events are small heap objects: a use after destroy or a double destroy is reported.
// TEST INFRASTRUCTURE ONLY (tests/test_host_asan.py).  Never linked into, shipped with or loaded by the product.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
struct FakeStream { unsigned magic; };
struct FakeEvent { unsigned magic; bool recorded; };
constexpr unsigned SM = 0x57AEA11u, EM = 0xE7E47u;
thread_local hipError_t t_last = hipSuccess;
hipError_t fail(hipError_t e) { t_last = e; return e; }
bool ok_stream(hipStream_t s) { return s == nullptr || reinterpret_cast<FakeStream *>(s)->magic == SM; }   /* a freed stream: ASan reports the read */
}

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : fail(hipErrorInvalidDevice); }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int)
{
    switch (a) {
    case hipDeviceAttributeWarpSize: *v = 64; break;
    case hipDeviceAttributeMultiprocessorCount: *v = 256; break;
    case hipDeviceAttributeMaxThreadsPerBlock: *v = 1024; break;
    case hipDeviceAttributeMaxSharedMemoryPerBlock: *v = 65536; break;
    default: *v = 1024;
    }
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int)
{
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "fake gfx950");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950");
    p->warpSize = 64; p->multiProcessorCount = 256; p->maxThreadsPerBlock = 1024; p->sharedMemPerBlock = 65536;
    p->maxThreadsPerMultiProcessor = 2048; p->regsPerBlock = 65536; p->totalGlobalMem = (size_t)1 << 36;
    p->maxGridSize[0] = p->maxGridSize[1] = p->maxGridSize[2] = 0x7fffffff;
    p->maxThreadsDim[0] = p->maxThreadsDim[1] = p->maxThreadsDim[2] = 1024;
    return hipSuccess;
}
int hipGetStreamDeviceId(hipStream_t s) { (void)ok_stream(s); return 0; }

hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : fail(hipErrorOutOfMemory); }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : fail(hipErrorOutOfMemory); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st)
{
    if (!ok_stream(st)) return fail(hipErrorInvalidHandle);
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st)
{
    if (!ok_stream(st)) return fail(hipErrorInvalidHandle);
    memset(d, v, n);
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(new FakeStream{SM}); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s)
{
    FakeStream *f = reinterpret_cast<FakeStream *>(s);
    if (!f || f->magic != SM) return fail(hipErrorInvalidHandle);
    f->magic = 0;
    delete f;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { return ok_stream(s) ? hipSuccess : fail(hipErrorInvalidHandle); }
hipError_t hipStreamQuery(hipStream_t s) { return ok_stream(s) ? hipSuccess : fail(hipErrorInvalidHandle); }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new FakeEvent{EM, false}); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    FakeEvent *f = reinterpret_cast<FakeEvent *>(e);
    if (!f || f->magic != EM) return fail(hipErrorInvalidHandle);
    f->magic = 0;
    delete f;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    FakeEvent *f = reinterpret_cast<FakeEvent *>(e);
    if (!f || f->magic != EM || !ok_stream(s)) return fail(hipErrorInvalidHandle);
    f->recorded = true;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { return reinterpret_cast<FakeEvent *>(e)->magic == EM ? hipSuccess : fail(hipErrorInvalidHandle); }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
    if (reinterpret_cast<FakeEvent *>(a)->magic != EM || reinterpret_cast<FakeEvent *>(b)->magic != EM) return fail(hipErrorInvalidHandle);
    *ms = 0.125f;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    return (ok_stream(s) && reinterpret_cast<FakeEvent *>(e)->magic == EM) ? hipSuccess : fail(hipErrorInvalidHandle);
}

/* kernel launches: accepted, not executed */
hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t s)
{
    if (!ok_stream(s)) return fail(hipErrorInvalidHandle);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.x * block.y * block.z > 1024) return fail(hipErrorInvalidConfiguration);
    return hipSuccess;
}
struct CallCfg { dim3 g, b; size_t shm; hipStream_t s; };
static thread_local CallCfg t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t s) { t_cfg = {g, b, shm, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *shm, hipStream_t *s) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *s = t_cfg.s; return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
}
