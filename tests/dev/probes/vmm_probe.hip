// (test infrastructure, stand-alone) Do copies, fills and kernels behave on memory mapped with hipMemCreate / hipMemMap on this stack?
// Behind the caveat on CV_GUARD_ALLOC (profiles/r06_tuning.md section 8): under the guard allocator 51 of 104 op tests computed wrong
// numbers although nothing faulted.   build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void add_one(uint32_t* p, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1u;
}

static size_t mismatches(const std::vector<uint32_t>& got, uint32_t base, bool ramp) {
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != (ramp ? base + (uint32_t)i : base);
    return bad;
}

static int run(const char* what, uint32_t* d, size_t n) {
    std::vector<uint32_t> h(n), back(n);
    // 1. fill + copy back
    CK(hipMemset(d, 0x5A, n * 4));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(back.data(), d, n * 4, hipMemcpyDeviceToHost));
    const size_t bad_fill = mismatches(back, 0x5A5A5A5Au, false);
    // 2. host -> device -> host
    for (size_t i = 0; i < n; ++i) h[i] = 1000u + (uint32_t)i;
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(back.data(), d, n * 4, hipMemcpyDeviceToHost));
    const size_t bad_copy = mismatches(back, 1000u, true);
    // 3. a kernel in between
    hipLaunchKernelGGL(add_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(back.data(), d, n * 4, hipMemcpyDeviceToHost));
    const size_t bad_kernel = mismatches(back, 1001u, true);
    // 4. the async forms on a stream, pinned host memory
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t* pin = nullptr;
    CK(hipHostMalloc(&pin, n * 4, hipHostMallocDefault));
    for (size_t i = 0; i < n; ++i) pin[i] = 7u + (uint32_t)i;
    CK(hipMemcpyAsync(d, pin, n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(add_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, n);
    CK(hipStreamSynchronize(s));
    std::memset(pin, 0, n * 4);
    CK(hipMemcpyAsync(pin, d, n * 4, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    size_t bad_async = 0;
    for (size_t i = 0; i < n; ++i) bad_async += pin[i] != 8u + (uint32_t)i;
    (void)hipHostFree(pin);
    (void)hipStreamDestroy(s);
    std::printf("%-28s %zu words: fill %zu bad, copy %zu bad, kernel %zu bad, async %zu bad\n", what, n, bad_fill, bad_copy, bad_kernel, bad_async);
    return (bad_fill || bad_copy || bad_kernel || bad_async) ? 2 : 0;
}

int main() {
    const size_t n = (size_t)3 << 18;                       // 3 MB
    int rc = 0;
    uint32_t* plain = nullptr;
    CK(hipMalloc(&plain, n * 4));
    rc |= run("hipMalloc", plain, n);
    (void)hipFree(plain);

    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    std::printf("allocation granularity %zu bytes\n", gran);
    for (size_t guard_mb : {(size_t)0, (size_t)64}) {
        const size_t guard = (guard_mb << 20) / gran * gran, mapped = (n * 4 + gran - 1) / gran * gran;
        void* base = nullptr;
        hipMemGenericAllocationHandle_t h{};
        CK(hipMemAddressReserve(&base, mapped + 2 * guard, gran, nullptr, 0));
        CK(hipMemCreate(&h, mapped, &prop, 0));
        CK(hipMemMap((char*)base + guard, mapped, 0, h, 0));
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess((char*)base + guard, mapped, &acc, 1));
        char label[64];
        std::snprintf(label, sizeof label, "hipMemMap, %zu MB guards", guard_mb);
        rc |= run(label, (uint32_t*)((char*)base + guard), n);
        // the guard allocator's mode 1 places the buffer at the END of the mapping: an offset pointer into the mapped range
        if (mapped > n * 4 / 2) {
            std::snprintf(label, sizeof label, "  same, offset 256 B");
            rc |= run(label, (uint32_t*)((char*)base + guard + 256), n - 64);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap((char*)base + guard, mapped));
        CK(hipMemRelease(h));
        CK(hipMemAddressFree(base, mapped + 2 * guard));
    }
    std::printf(rc ? "mapped memory does NOT behave like hipMalloc'ed memory here\n" : "mapped memory behaves\n");
    return rc;
}
