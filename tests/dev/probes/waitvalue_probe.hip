// Probe (developer tool): hipStreamWaitValue32 on this box -- does a stream released by a HOST store start its next kernel sooner than a
// kernel launched after the fact?   hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void stamp(unsigned long long* out) { if (threadIdx.x == 0) *out = wall_clock64(); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue: %d\n", can);
    if (!can) return 0;
    unsigned* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("signal alloc: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned long long* d_t; hipHostMalloc((void**)&d_t, 64, hipHostMallocDefault);
    volatile unsigned* hs = (volatile unsigned*)sig;
    *hs = 0;
    fflush(stdout);
    double lat_wait = 0, lat_launch = 0;
    const int N = 200;
    for (int i = 1; i <= N; ++i) {
        // (a) pre-enqueued kernel behind a wait, released by a host store
        *d_t = 0;
        e = hipStreamWaitValue32(st, sig, (unsigned)i, hipStreamWaitValueEq, 0xffffffffu);
        if (e != hipSuccess) { printf("wait: %s\n", hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, st, d_t);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        const double t0 = now_us();
        *hs = (unsigned)i;
        while (*(volatile unsigned long long*)d_t == 0) { if (now_us() - t0 > 2e6) { printf("TIMEOUT: the wait never released\n"); return 2; } }
        lat_wait += now_us() - t0;
        hipStreamSynchronize(st);
        // (b) kernel launched now
        *d_t = 0;
        const double t1 = now_us();
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, st, d_t);
        while (*(volatile unsigned long long*)d_t == 0) { }
        lat_launch += now_us() - t1;
        hipStreamSynchronize(st);
    }
    printf("host store -> kernel result visible: %.2f us;  hipLaunchKernel -> kernel result visible: %.2f us\n", lat_wait / N, lat_launch / N);
    return 0;
}
