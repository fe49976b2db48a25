// conv_device.h -- device helpers shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_halo.hip).
#pragma once
#include <type_traits>

#include "cv_kernels.h"

namespace cv {

#ifndef CV_STAMP
#define CV_STAMP 0        // diagnostic build: wave 0 of every workgroup stamps s_memtime / s_memrealtime around the K loop
#endif
#ifndef CV_IGEMM_PIPE
#define CV_IGEMM_PIPE 1   // conv_igemm.hip: register-rotating software pipeline (0 = the barrier -> DMA -> reads -> MFMA loop)
#endif
#ifndef CV_SCHED_HINTS
#define CV_SCHED_HINTS 1  // conv_halo.hip: interleave the next stage's DMA issue / LDS reads with the tail MFMAs
#endif
#ifndef CV_SETPRIO
#define CV_SETPRIO 1
#endif
constexpr bool kSetPrio = CV_SETPRIO != 0;
#ifndef CV_ABLATE
#define CV_ABLATE 0       // timing experiments only (wrong results): 1 = no steady-state DMA, 2 = no MFMA
#endif

template <typename T> struct FragT;
template <> struct FragT<half_t> { typedef half8 V; };
template <> struct FragT<float>  { typedef f4 V; };
template <> struct FragT<split_t> { typedef half8 V; };

__device__ __forceinline__ void mma16(f4& acc, const half8& a, const half8& b) {
#if CV_ABLATE == 2
    asm volatile("" ::"v"(a), "v"(b));                 // keep the fragment loads alive, issue no MFMA
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#endif
}
// f32: one 16-B chunk = 4 k-values per lane; lane group q owns chunk (sub*4+q), MFMA j contracts the
// j-th value of all four groups.  The k order inside a stage is permuted identically for A and B.
__device__ __forceinline__ void mma16(f4& acc, const f4& a, const f4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// the same with the address split into a wave-uniform base and a per-lane 32-bit byte offset: lets the compiler pick the
// SGPR-base form of the instruction, so the per-piece 64-bit address arithmetic runs on the scalar unit instead of taking vector
// issue slots that the MFMAs of both resident waves share
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase + voff),
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    // every wave: its own DMA of the stage about to be read has landed (all but N younger ones), its
    // LDS reads of the stage about to be overwritten have returned; then the workgroup rendezvous.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// N consecutive channels of one pixel -> 16-B stores.  ch0 = absolute index of the first channel in the buffer
// (selects the [hi,lo] / [lo,hi] chunk order of split-f16 groups; unused otherwise).
template <typename T, int N> struct OutVec;
template <int N> struct OutVec<half_t, N> {
    static __device__ __forceinline__ void store(half_t* dst, int, const float* v, float& bad) {
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            half8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) { h[j] = (half_t)v[i + j]; bad = __builtin_fmaf((float)h[j], 0.f, bad); }
            *reinterpret_cast<half8*>(dst + i) = h;
        }
    }
    // residual already in registers (kRawChunks 16-byte chunks per unit, fetched ahead of the epilogue)
    static constexpr int kRawChunks = N / 8;
    static __device__ __forceinline__ void fetch(const half_t* src, f4* raw) {
#pragma unroll
        for (int i = 0; i < N / 8; ++i) raw[i] = *reinterpret_cast<const f4*>(src + i * 8);
    }
    static __device__ __forceinline__ void add_raw(const f4* raw, int, float* v, float mul) {
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            const half8 h = __builtin_bit_cast(half8, raw[i / 8]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i + j] = __builtin_fmaf((float)h[j], mul, v[i + j]);
        }
    }
    static __device__ __forceinline__ void add(const half_t* src, int, float* v, float mul) {
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            const half8 h = *reinterpret_cast<const half8*>(src + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i + j] = __builtin_fmaf((float)h[j], mul, v[i + j]);
        }
    }
};
template <int N> struct OutVec<float, N> {
    static __device__ __forceinline__ void store(float* dst, int, const float* v, float& bad) {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            f4 o = {v[i], v[i + 1], v[i + 2], v[i + 3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) bad = __builtin_fmaf(o[j], 0.f, bad);
            *reinterpret_cast<f4*>(dst + i) = o;
        }
    }
    static constexpr int kRawChunks = N / 4;
    static __device__ __forceinline__ void fetch(const float* src, f4* raw) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) raw[i] = *reinterpret_cast<const f4*>(src + i * 4);
    }
    static __device__ __forceinline__ void add_raw(const f4* raw, int, float* v, float) {
#pragma unroll
        for (int i = 0; i < N; i += 4) { v[i] += raw[i / 4][0]; v[i + 1] += raw[i / 4][1]; v[i + 2] += raw[i / 4][2]; v[i + 3] += raw[i / 4][3]; }
    }
    static __device__ __forceinline__ void add(const float* src, int, float* v, float) {   // the f32 engine never rescales
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const f4 o = *reinterpret_cast<const f4*>(src + i);
            v[i] += o[0]; v[i + 1] += o[1]; v[i + 2] += o[2]; v[i + 3] += o[3];
        }
    }
};
template <int N> struct OutVec<split_t, N> {
    static __device__ __forceinline__ void store(split_t* dst, int ch0, const float* v, float& bad) {
        char* p = reinterpret_cast<char*>(dst);
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            const int par = ((ch0 + i) >> 3) & 1;
            // three vector instructions per PAIR of values for the hi/lo split (cv_kernels.h: split_pair) and the numeric guard on the
            // packed hi pair (hi * 0 is NaN for +-inf and NaN): ~2 instructions per stored value instead of ~4.5
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 hu, lu;
            h2 g2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned hp, lp;
                split_pair(v[i + j], v[i + j + 1], hp, lp);
                g2 = __builtin_elementwise_fma(__builtin_bit_cast(h2, hp), h2{(half_t)0.f, (half_t)0.f}, g2);
                hu[j / 2] = hp; lu[j / 2] = lp;
            }
            bad = __builtin_fmaf((float)g2[0] + (float)g2[1], 0.f, bad);
            const half8 hi = __builtin_bit_cast(half8, hu), lo = __builtin_bit_cast(half8, lu);
            *reinterpret_cast<half8*>(p + i * 4 + (par ? 16 : 0)) = hi;
            *reinterpret_cast<half8*>(p + i * 4 + (par ? 0 : 16)) = lo;
        }
    }
    // one 16-byte half of the 32-byte group starting at ch0: the hi chunk (want_hi) or the lo chunk.  Lets two lanes that
    // hold the same eight values share one store instruction (fused max-pool: both pixels of a pair hold the maximum).
    static __device__ __forceinline__ void store_half(split_t* dst, int ch0, const float* v, bool want_hi) {
        const int par = (ch0 >> 3) & 1;                  // even group: [hi, lo]; odd group: [lo, hi]
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 o;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            unsigned hp, lp;
            split_pair(v[j], v[j + 1], hp, lp);
            o[j / 2] = want_hi ? hp : lp;
        }
        *reinterpret_cast<u4*>(reinterpret_cast<char*>(dst) + ((want_hi ? par : par ^ 1) ? 16 : 0)) = o;
    }
    static constexpr int kRawChunks = N / 4;
    static __device__ __forceinline__ void fetch(const split_t* src, f4* raw) {
        const char* p = reinterpret_cast<const char*>(src);
#pragma unroll
        for (int i = 0; i < N / 4; ++i) raw[i] = *reinterpret_cast<const f4*>(p + i * 16);
    }
    static __device__ __forceinline__ void add_raw(const f4* raw, int, float* v, float mul) {   // hi + lo: chunk order does not matter
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            const half8 a = __builtin_bit_cast(half8, raw[i / 4]);
            const half8 b = __builtin_bit_cast(half8, raw[i / 4 + 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i + j] = __builtin_fmaf((float)a[j] + (float)b[j], mul, v[i + j]);
        }
    }
    static __device__ __forceinline__ void add(const split_t* src, int ch0, float* v, float mul) {
        const char* p = reinterpret_cast<const char*>(src);
#pragma unroll
        for (int i = 0; i < N; i += 8) {
            const half8 a = *reinterpret_cast<const half8*>(p + i * 4);
            const half8 b = *reinterpret_cast<const half8*>(p + i * 4 + 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i + j] = __builtin_fmaf((float)a[j] + (float)b[j], mul, v[i + j]);
        }
    }
};


// f16r engine (f16 kernels only): the residual trunk's f32 twin.  A unit = 8 consecutive channels = two 16-byte chunks.
__device__ __forceinline__ void trunk32_fetch(const float* src, f4* raw) {
    raw[0] = *reinterpret_cast<const f4*>(src);
    raw[1] = *reinterpret_cast<const f4*>(src + 4);
}
__device__ __forceinline__ void trunk32_add_raw(const f4* raw, float* v, float mul) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = __builtin_fmaf(raw[0][j], mul, v[j]); v[4 + j] = __builtin_fmaf(raw[1][j], mul, v[4 + j]); }
}
__device__ __forceinline__ void trunk32_store(float* dst, const float* v) {
    *reinterpret_cast<f4*>(dst) = f4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4*>(dst + 4) = f4{v[4], v[5], v[6], v[7]};
}

// A lane whose `bad` accumulator went NaN stored a non-finite value: record the launch's layer id (lowest id wins, so
// the host names the FIRST layer that left the representable range).
__device__ __forceinline__ void report_bad(const ConvParams& p, float bad) {
    if (bad != bad && p.flag) atomicMin(p.flag, p.layer_id);
}

}  // namespace cv
