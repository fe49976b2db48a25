// mfma_shape_ab.hip -- same-box A/B of the two gfx950 f16 MFMA shapes on the production tile of conv3x3_halo_kernel.
//
// the dominant kernel issues v_mfma_f32_16x16x32_f16; would v_mfma_f32_32x32x16_f16 at the SAME 64-channel x
// 64-pixel output per wave hold a higher clock / deliver more?  This stand-alone program reproduces what decides that -- the K loop's
// instruction mix at the production occupancy -- without the DMA, epilogue and address plumbing of the real kernel:
//   * workgroup = 4 waves, 68 KB of LDS (3 weight slots of 64 rows x 128 B + an 18 x 18 x 128 B halo) -> two workgroups per CU;
//   * per stage (one tap of one 32-channel block, split-f16: hi and lo chunk per 8-channel group) every wave reads 8 weight and
//     8 pixel fragments with ds_read_b128 from the same XOR-swizzled images the kernel uses, one s_barrier per stage, fragments of
//     stage s+1 read while stage s computes (register double buffering), three products hi.hi + lo.hi + hi.lo;
//   * 16x16x32: 4 x 4 accumulators of 4 registers, 48 MFMAs per stage;  32x32x16: 2 x 2 accumulators of 16 registers, 2 k-steps,
//     24 MFMAs per stage -- the same FLOPs, LDS bytes and registers.
// Random f16 operands (clock under load depends on the data), >= 2 s of back-to-back launches before the measured one, in-kernel
// clock = delta s_memtime / delta s_memrealtime x 100 MHz stamped around the loop (median over workgroups).
//
//   build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_ab.hip -o tools/mfma_shape_ab      run: tools/mfma_shape_ab [stages=1152]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int WSTAGE = 64 * 128;                // one weight slot
constexpr int HALO0 = 3 * WSTAGE;               // halo image behind the 3-slot ring
constexpr int HROWS = 18 * 18;
constexpr int LDS_BYTES = HALO0 + 44 * 1024;    // 68 KB as the production tile (halo padded to the DMA piece size)

template <bool M32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop_kernel(const half8* __restrict__ init, int stages,
                                                                                            float* __restrict__ out, unsigned long long* __restrict__ stamp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += 256) reinterpret_cast<half8*>(smem)[i] = init[(blockIdx.x * 7 + i) & 8191];
    __syncthreads();
    const int wrow0 = wave * 4;                 // first patch row of this wave (four rows of 16 pixels each)

    if constexpr (!M32) {
        const int q = lane >> 4, l15 = lane & 15, l7 = lane & 7;
        const int c0 = 2 * q + (q & 1), c1 = 2 * q + 1 - (q & 1);
        int aoff[2], boff[2][3];
        aoff[0] = l15 * 128 + ((c0 ^ l7) << 4);
        aoff[1] = l15 * 128 + ((c1 ^ l7) << 4);
        for (int kx = 0; kx < 3; ++kx) {
            const int col = l15 + kx, row = wrow0 * 18 + col;
            boff[0][kx] = HALO0 + row * 128 + ((c0 ^ (col & 7)) << 4);
            boff[1][kx] = HALO0 + row * 128 + ((c1 ^ (col & 7)) << 4);
        }
        f4 acc[4][4];
        for (int f = 0; f < 4; ++f)
            for (int g = 0; g < 4; ++g) acc[f][g] = f4{0.f, 0.f, 0.f, 0.f};
        struct Fr { half8 a[2][4], b[2][4]; } F[2];
        auto load = [&](Fr& X, int tap) __attribute__((always_inline)) {
            const int slot = (tap % 3) * WSTAGE, ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                X.a[0][f] = *reinterpret_cast<const half8*>(smem + slot + aoff[0] + f * 2048);
                X.a[1][f] = *reinterpret_cast<const half8*>(smem + slot + aoff[1] + f * 2048);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                X.b[0][g] = *reinterpret_cast<const half8*>(smem + boff[0][kx] + (g + ky) * (18 * 128));
                X.b[1][g] = *reinterpret_cast<const half8*>(smem + boff[1][kx] + (g + ky) * (18 * 128));
            }
        };
        auto mma = [&](const Fr& X) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X.a[0][f], X.b[0][g], acc[f][g], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X.a[1][f], X.b[0][g], acc[f][g], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(X.a[0][f], X.b[1][g], acc[f][g], 0, 0, 0);
        };
        load(F[0], 0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int s = 0; s < stages; s += 18) {
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                load(F[(j + 1) & 1], (j + 1) % 9);
                mma(F[j & 1]);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float sum = 0.f;
        for (int f = 0; f < 4; ++f)
            for (int g = 0; g < 4; ++g) sum += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
        out[blockIdx.x * 256 + tid] = sum;
        if (lane == 0) { stamp[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamp[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
    } else {
        const int h = lane >> 5, l31 = lane & 31, l7 = lane & 7;
        // k-step kk covers channel groups 2kk and 2kk+1; lane half h takes group g = 2kk + h: hi chunk 2g + (g&1), lo chunk 2g + 1 - (g&1)
        int aoff[2][2], boff[2][2][3];
        for (int kk = 0; kk < 2; ++kk) {
            const int g = 2 * kk + h, chi = 2 * g + (g & 1), clo = 2 * g + 1 - (g & 1);
            aoff[0][kk] = l31 * 128 + ((chi ^ l7) << 4);
            aoff[1][kk] = l31 * 128 + ((clo ^ l7) << 4);
            for (int kx = 0; kx < 3; ++kx) {
                const int col = (l31 & 15) + kx, row = (wrow0 + (l31 >> 4)) * 18 + col;
                boff[0][kk][kx] = HALO0 + row * 128 + ((chi ^ (col & 7)) << 4);
                boff[1][kk][kx] = HALO0 + row * 128 + ((clo ^ (col & 7)) << 4);
            }
        }
        f16v acc[2][2];
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        struct Fr { half8 a[2][2][2], b[2][2][2]; } F[2];      // [set][tile][k-step]
        auto load = [&](Fr& X, int tap) __attribute__((always_inline)) {
            const int slot = (tap % 3) * WSTAGE, ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    X.a[0][a][kk] = *reinterpret_cast<const half8*>(smem + slot + aoff[0][kk] + a * 4096);
                    X.a[1][a][kk] = *reinterpret_cast<const half8*>(smem + slot + aoff[1][kk] + a * 4096);
                }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    X.b[0][b][kk] = *reinterpret_cast<const half8*>(smem + boff[0][kk][kx] + (2 * b + ky) * (18 * 128));
                    X.b[1][b][kk] = *reinterpret_cast<const half8*>(smem + boff[1][kk][kx] + (2 * b + ky) * (18 * 128));
                }
        };
        auto mma = [&](const Fr& X) __attribute__((always_inline)) {
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                const int sa = pass == 1 ? 1 : 0, sb = pass == 2 ? 1 : 0;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X.a[sa][a][kk], X.b[sb][b][kk], acc[a][b], 0, 0, 0);
            }
        };
        load(F[0], 0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int s = 0; s < stages; s += 18) {
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                load(F[(j + 1) & 1], (j + 1) % 9);
                mma(F[j & 1]);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float sum = 0.f;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b)
                for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
        out[blockIdx.x * 256 + tid] = sum;
        if (lane == 0) { stamp[(blockIdx.x * 4 + wave) * 2] = t1 - t0; stamp[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool M32>
static int run(const char* name, const half8* init, int stages, int grid, float* out, unsigned long long* stamp) {
    auto kern = loop_kernel<M32>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const auto t_warm = std::chrono::steady_clock::now();
    int warm = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_warm).count() < 2.0) {      // >= 2 s under load first
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, init, stages, out, stamp);
        CK(hipDeviceSynchronize());
        warm += 20;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, init, stages, out, stamp);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<unsigned long long> h((size_t)grid * 8);
    CK(hipMemcpy(h.data(), stamp, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (size_t i = 0; i + 1 < h.size(); i += 2)
        if (h[i + 1]) { cyc.push_back((double)h[i]); clk.push_back((double)h[i] / (double)h[i + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    // issued FLOPs: per wave and stage 64 channels x 64 pixels x 32 k x 3 products x 2
    const double flop = (double)grid * 4 * (double)stages * 64.0 * 64.0 * 32.0 * 3.0 * 2.0;
    std::printf("{\"shape\": \"%s\", \"grid\": %d, \"stages\": %d, \"ms\": %.4f, \"issued_tflops\": %.1f, \"algorithmic_tflops_f16x3\": %.1f, "
                "\"cycles_per_stage_median\": %.1f, \"in_kernel_clock_ghz_median\": %.3f, \"warmup_launches\": %d}\n",
                name, grid, stages, ms, flop / (ms * 1e-3) / 1e12, flop / 3.0 / (ms * 1e-3) / 1e12,
                cyc[cyc.size() / 2] / stages, clk[clk.size() / 2], warm);
    return 0;
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? std::atoi(argv[1]) / 18 * 18 : 1152;      // 64 channel blocks x 18: a long 3x3 layer
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus * 2 * 4;                                           // four rounds of two resident workgroups per CU
    std::vector<_Float16> host(8192 * 8);
    unsigned rs = 12345u;
    for (auto& v : host) { rs = rs * 1664525u + 1013904223u; v = (_Float16)(((int)(rs >> 16) % 2001 - 1000) / 1000.0f); }
    half8* init; float* out; unsigned long long* stamp;
    CK(hipMalloc(&init, host.size() * 2)); CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&stamp, (size_t)grid * 64));
    CK(hipMemcpy(init, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {                               // A B A B: the order does not decide
        if (run<false>("16x16x32", init, stages, grid, out, stamp)) return 1;
        if (run<true>("32x32x16", init, stages, grid, out, stamp)) return 1;
    }
    return 0;
}
