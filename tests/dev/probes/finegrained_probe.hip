// Probe (developer tool): is fine-grained DEVICE memory readable by the host through the BAR on this box, and what does a read cost?
// hipcc --offload-arch=gfx950 -O2 -o finegrained_probe finegrained_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(unsigned* p, unsigned v) { if (threadIdx.x == 0) atomicMin(p, v); }
int main() {
    unsigned* p = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&p, 64, hipDeviceMallocFinegrained);
    printf("alloc: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipMemset(p, 0xff, 64);
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, p, 1234u);
    hipDeviceSynchronize();
    hipPointerAttribute_t a;
    e = hipPointerGetAttributes(&a, p);
    printf("attr: %s type %d host %p dev %p\n", hipGetErrorString(e), (int)a.type, a.hostPointer, a.devicePointer);
    fflush(stdout);
    volatile unsigned* hp = (volatile unsigned*)p;
    unsigned v = *hp;                                        // faults here if the pool is not host-accessible
    printf("host read: %u\n", v);
    auto t0 = std::chrono::steady_clock::now();
    unsigned acc = 0;
    for (int i = 0; i < 1000; ++i) acc += *hp;
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
    printf("host read cost: %.2f us (%u)\n", us, acc);
    unsigned h = 0;
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) { hipMemcpyAsync(&h, p, 4, hipMemcpyDeviceToHost, 0); hipStreamSynchronize(0); }
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
    printf("4-byte hipMemcpyAsync + sync: %.2f us\n", us);
    return 0;
}
