// (test infrastructure) A stand-in HIP runtime that keeps "device" memory on the host and turns every kernel launch into a checked
// no-op, so that the HOST side of the library -- weight packing, offset / position tables, launch plans and grids, workspace growth, the
// block cache, calibration bookkeeping, the C ABI's argument handling -- can run under AddressSanitizer + UndefinedBehaviorSanitizer on a
// machine without a GPU (tests/test_engine_host_sanitizers.py builds csrc/ with `--cuda-host-only` and links it against this file instead
// of libamdhip64).  "Device" buffers are plain malloc blocks: every upload, fill and read-back of the engine is a memcpy / memset that
// ASan bounds-checks against the size the engine asked for.  Launches are validated for what the hardware would refuse (zero or
// oversized grids and blocks, more dynamic LDS than a CU has).  Nothing here is part of the product; the product links the real runtime.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
struct LaunchCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local LaunchCfg t_cfg;
std::atomic<long> g_launches{0}, g_allocs{0}, g_bad_launches{0};
thread_local hipError_t t_last = hipSuccess;
thread_local int t_device = 0;
hipError_t fail(hipError_t e) { t_last = e; return e; }
int g_dummy_handle;
}  // namespace

extern "C" {

// ---- what the compiler's launch sequence calls ---------------------------------------------------------------------------------
void** __hipRegisterFatBinary(const void*) { return reinterpret_cast<void**>(&g_dummy_handle); }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg = LaunchCfg{grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *stream = t_cfg.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t) {
    ++g_launches;
    const unsigned long long threads = 1ull * block.x * block.y * block.z;
    const bool ok = fn && args && grid.x && grid.y && grid.z && threads && threads <= 1024 && grid.y <= 65535 && grid.z <= 65535 &&
                    shmem <= 160 * 1024;
    if (!ok) {
        ++g_bad_launches;
        std::fprintf(stderr, "[hip stub] refused launch: grid %u x %u x %u, block %u x %u x %u, %zu bytes of LDS\n", grid.x, grid.y, grid.z,
                     block.x, block.y, block.z, shmem);
        return fail(hipErrorInvalidConfiguration);
    }
    return hipSuccess;
}
// negative control for the test: one byte past a block, on purpose -- the process must die with an AddressSanitizer report
void cv_stub_selftest_overflow(void) {
    volatile char* p = static_cast<volatile char*>(std::malloc(32));
    p[32] = 1;
    std::free(const_cast<char*>(p));
}
long cv_stub_launches(void) { return g_launches.load(); }
long cv_stub_bad_launches(void) { return g_bad_launches.load(); }
long cv_stub_allocations(void) { return g_allocs.load(); }

// ---- devices ---------------------------------------------------------------------------------------------------------------------
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return fail(hipErrorInvalidDevice); t_device = d; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t attr, int) {
    *v = attr == hipDeviceAttributeMultiprocessorCount ? 256 : attr == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 * 1024 : 0;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "host stand-in");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 64 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stand-in runtime error"; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// ---- memory: "device" = host, so every copy is bounds-checked by the sanitizer -----------------------------------------------------
hipError_t hipMalloc(void** p, size_t n) { ++g_allocs; *p = std::malloc(n ? n : 1); return *p ? hipSuccess : fail(hipErrorOutOfMemory); }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { ++g_allocs; *p = std::malloc(n ? n : 1); return *p ? hipSuccess : fail(hipErrorOutOfMemory); }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
// the virtual-memory calls behind CV_GUARD_ALLOC are not exercised on the host
hipError_t hipMemGetAllocationGranularity(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) { return fail(hipErrorNotSupported); }
hipError_t hipMemAddressReserve(void**, size_t, size_t, void*, unsigned long long) { return fail(hipErrorNotSupported); }
hipError_t hipMemAddressFree(void*, size_t) { return fail(hipErrorNotSupported); }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long) { return fail(hipErrorNotSupported); }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return fail(hipErrorNotSupported); }
hipError_t hipMemMap(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return fail(hipErrorNotSupported); }
hipError_t hipMemUnmap(void*, size_t) { return fail(hipErrorNotSupported); }
hipError_t hipMemSetAccess(void*, size_t, const hipMemAccessDesc*, size_t) { return fail(hipErrorNotSupported); }

// ---- streams, events, graphs --------------------------------------------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.125f; return hipSuccess; }
// no capture on the host: the engine falls back to eager launches (its documented behaviour when a capture cannot start)
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return fail(hipErrorStreamCaptureUnsupported); }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return fail(hipErrorStreamCaptureUnsupported); }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return fail(hipErrorNotSupported); }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return fail(hipErrorNotSupported); }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

}  // extern "C"
