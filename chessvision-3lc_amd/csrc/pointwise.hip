// pointwise.hip -- the HBM-bound kernels of the ChessVision hot path, hand-written for gfx950.
//
// Everything here moves 16 bytes per lane per access (8 f16 / 4 f32 channels of one NHWC pixel), is
// launched with >> 256 workgroups, and has no inter-block reuse, so there is nothing to tile: the bound is
// HBM bytes ((elements in + elements out) * dtype size).
//
// Reference ops replaced:  max_pool2d 2x2 (UNet Down), max_pool2d 3x3 s2 p1 (ResNet),
// upsample_bilinear2d(align_corners=True) (UNet Up, bilinear variant), conv2d 1x1 64->1 + bias (OutConv) fused
// with sigmoid/threshold (core.py:273, utils.py:101-112), conv 7x7 s2 p3 + BN + ReLU (ResNet stem),
// adaptive_avg_pool2d(1) + linear 512->13 (+ softmax, core.py:242), and the u8 HWC -> /255 -> NCHW input
// packing of core.py:215-216 / 236-237.
#include "pointwise.h"
#include "conv_device.h"   // OutVec (split-f16 store units)

#include <algorithm>
#include <cmath>

namespace cv {

// A "group" = the channels one lane moves per access: 8 (f16: 16 B; split-f16: 16 B hi + 16 B lo) or 4 (f32: 16 B).
template <typename T> struct Grp;
template <> struct Grp<half_t> {
    static constexpr int N = 8, BYTES = 16;
    static __device__ __forceinline__ void load(const char* p, int, float* v) {
        const half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    }
    static __device__ __forceinline__ void store(char* p, int, const float* v) {
        half8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
        *reinterpret_cast<half8*>(p) = h;
    }
    static __device__ __forceinline__ void store(char* p, int par, const float* v, float& bad) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bad = __builtin_fmaf((float)(half_t)v[j], 0.f, bad);
        store(p, par, v);
    }
};
template <> struct Grp<float> {
    static constexpr int N = 4, BYTES = 16;
    static __device__ __forceinline__ void load(const char* p, int, float* v) {
        const f4 h = *reinterpret_cast<const f4*>(p);
        v[0] = h[0]; v[1] = h[1]; v[2] = h[2]; v[3] = h[3];
    }
    static __device__ __forceinline__ void store(char* p, int, const float* v) {
        *reinterpret_cast<f4*>(p) = f4{v[0], v[1], v[2], v[3]};
    }
    static __device__ __forceinline__ void store(char* p, int par, const float* v, float& bad) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bad = __builtin_fmaf(v[j], 0.f, bad);
        store(p, par, v);
    }
};
template <> struct Grp<split_t> {              // [hi x8][lo x8] for even groups, [lo x8][hi x8] for odd ones
    static constexpr int N = 8, BYTES = 32;
    static __device__ __forceinline__ void load(const char* p, int parity, float* v) {
        const half8 hi = *reinterpret_cast<const half8*>(p + (parity ? 16 : 0));
        const half8 lo = *reinterpret_cast<const half8*>(p + (parity ? 0 : 16));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
    }
    static __device__ __forceinline__ void store(char* p, int parity, const float* v) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            unsigned hp, lp;
            split_pair(v[j], v[j + 1], hp, lp);             // cv_kernels.h: three vector instructions per pair
            hi[j / 2] = hp; lo[j / 2] = lp;
        }
        *reinterpret_cast<u4*>(p + (parity ? 16 : 0)) = hi;
        *reinterpret_cast<u4*>(p + (parity ? 0 : 16)) = lo;
    }
    static __device__ __forceinline__ void store(char* p, int parity, const float* v, float& bad) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bad = __builtin_fmaf((float)(half_t)v[j], 0.f, bad);   // inf / NaN -> NaN
        store(p, parity, v);
    }
};

// numeric guard (cv_kernels.h: ConvParams::flag): the lowest layer id that stored a non-finite value wins
__device__ __forceinline__ void report_bad(unsigned* flag, unsigned layer_id, float bad) {
    if (bad != bad && flag) atomicMin(flag, layer_id);
}

__device__ __forceinline__ size_t pix_index(const TensorRef& t, int n, int y, int x) {
    return (size_t)(n * (t.H + 2) + y + 1) * (t.W + 2) + (x + 1);
}
// byte address and hi/lo parity of group g (counted from the slice's first channel) of pixel `pix`
template <typename T> __device__ __forceinline__ char* grp_ptr(const TensorRef& t, size_t pix, int g, int* parity) {
    constexpr int ESZ = Grp<T>::BYTES / Grp<T>::N;
    *parity = (t.Coff / Grp<T>::N + g) & 1;
    return reinterpret_cast<char*>(t.base) + (pix * t.Cs + t.Coff) * ESZ + (size_t)g * Grp<T>::BYTES;
}

static inline unsigned grid_for(size_t work, int block = 256) {
    size_t g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g);
}

// decode a flat (pixel, group) work index of tensor `t`
struct PG { int n, y, x, g; bool live; };
template <typename T> __device__ __forceinline__ PG decode_pg(const TensorRef& t) {
    const int groups = t.C / Grp<T>::N;
    const size_t total = (size_t)t.N * t.H * t.W * groups;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    PG r;
    r.live = idx < total;
    r.g = (int)(idx % groups);
    size_t pix = idx / groups;
    r.x = (int)(pix % t.W); pix /= t.W;
    r.y = (int)(pix % t.H);
    r.n = (int)(pix / t.H);
    return r;
}

// ---- packing ----------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_nchw_f32_kernel(const float* __restrict__ src, int c, TensorRef dst, float mul, unsigned* flag) {
    constexpr int GN = Grp<T>::N;
    const PG p = decode_pg<T>(dst);
    if (!p.live) return;
    float v[GN];
#pragma unroll
    for (int j = 0; j < GN; ++j) {
        const int ch = p.g * GN + j;
        v[j] = ch < c ? src[((size_t)(p.n * c + ch) * dst.H + p.y) * dst.W + p.x] * mul : 0.f;
    }
    int par;
    char* d = grp_ptr<T>(dst, pix_index(dst, p.n, p.y, p.x), p.g, &par);
    float bad = 0.f;
    Grp<T>::store(d, par, v, bad);
    report_bad(flag, 0u, bad);                               // layer id 0 = the caller's input tensor
}

template <typename T>
__global__ void pack_hwc3_u8_kernel(const uint8_t* __restrict__ src, TensorRef dst, float mul) {
    constexpr int GN = Grp<T>::N;
    const size_t total = (size_t)dst.N * dst.H * dst.W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t pix = idx;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    const uint8_t* s = src + idx * 3;
    // /255 first (the reference's arithmetic, core.py:215), then the power-of-two range factor (exact)
    const float f[8] = {(float)s[0] / 255.f * mul, (float)s[1] / 255.f * mul, (float)s[2] / 255.f * mul, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8 / GN; ++g) {
        int par;
        char* d = grp_ptr<T>(dst, pix_index(dst, n, y, x), g, &par);
        Grp<T>::store(d, par, f + g * GN);
    }
}

template <typename T>
__global__ void unpack_nchw_f32_kernel(TensorRef src, float* __restrict__ dst, float mul) {
    constexpr int GN = Grp<T>::N;
    const size_t total = (size_t)src.N * src.C * src.H * src.W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t r = idx;
    const int x = (int)(r % src.W); r /= src.W;
    const int y = (int)(r % src.H); r /= src.H;
    const int c = (int)(r % src.C);
    const int n = (int)(r / src.C);
    int par;
    const char* p = grp_ptr<T>(src, pix_index(src, n, y, x), c / GN, &par);
    float v[GN];
    Grp<T>::load(p, par, v);
    float out = v[0];
#pragma unroll
    for (int j = 1; j < GN; ++j) out = (c % GN) == j ? v[j] : out;
    dst[idx] = out * mul;
}

// ---- range calibration: max |stored value| over the interior of a slice, as the bits of a non-negative float ------
// (unsigned order == float order there; NaN bit patterns sort above +inf, so "non-finite" reads as >= 0x7f800000)
template <typename T>
__global__ void absmax_kernel(TensorRef src, unsigned* __restrict__ out) {
    constexpr int GN = Grp<T>::N;
    const int groups = src.C / GN;
    const size_t total = (size_t)src.N * src.H * src.W * groups;
    unsigned m = 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        size_t pix = idx / groups;
        const int x = (int)(pix % src.W); pix /= src.W;
        const int y = (int)(pix % src.H);
        const int n = (int)(pix / src.H);
        int par;
        float v[GN];
        Grp<T>::load(grp_ptr<T>(src, pix_index(src, n, y, x), g, &par), par, v);
#pragma unroll
        for (int j = 0; j < GN; ++j) {
            const unsigned b = __float_as_uint(v[j]) & 0x7fffffffu;
            m = b > m ? b : m;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- load-time rounding-bias calibration (engine.cpp: Engine::measure_tap_sums): per convolution tap and input channel, the sum of
// the stored f16 input over the slice's images and every output position -- out[slice][tap][c] = sum_n sum_(oy,ox) x(n, oy*stride +
// ky - pad, ox*stride + kx - pad, c), the zero border read as it is.  One lane = one channel (consecutive lanes read consecutive
// halves); slices are summed on the host in index order, so the result is the same from run to run.
__global__ __launch_bounds__(64) void tap_sums_f16_kernel(TensorRef src, int Ho, int Wo, int stride, int k, int n_per_slice, double* __restrict__ out) {
    const int tap = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x, slice = blockIdx.z;
    if (c >= src.C) return;
    const int pad = (k - 1) / 2, ky = tap / k, kx = tap % k;
    const half_t* base = reinterpret_cast<const half_t*>(src.base) + src.Coff + c;
    const int n0 = slice * n_per_slice, n1 = min(src.N, n0 + n_per_slice);
    double total = 0.0;
    for (int n = n0; n < n1; ++n) {
        float part = 0.f;                                  // <= Ho * Wo terms of one image in f32, images in f64
        for (int oy = 0; oy < Ho; ++oy) {
            const size_t row = ((size_t)n * (src.H + 2) + (size_t)(oy * stride + ky - pad + 1)) * (src.W + 2);
            for (int ox = 0; ox < Wo; ++ox) part += (float)base[(row + (size_t)(ox * stride + kx - pad + 1)) * src.Cs];
        }
        total += (double)part;
    }
    out[((size_t)slice * (k * k) + tap) * src.C + c] = total;
}

// ---- pooling / upsampling --------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2x2_kernel(TensorRef src, TensorRef dst) {
    constexpr int GN = Grp<T>::N;
    const PG p = decode_pg<T>(dst);
    if (!p.live) return;
    float m[GN], v[GN];
    int par;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const char* s = grp_ptr<T>(src, pix_index(src, p.n, 2 * p.y + (t >> 1), 2 * p.x + (t & 1)), p.g, &par);
        Grp<T>::load(s, par, v);
#pragma unroll
        for (int j = 0; j < GN; ++j) m[j] = t == 0 ? v[j] : fmaxf(m[j], v[j]);
    }
    char* d = grp_ptr<T>(dst, pix_index(dst, p.n, p.y, p.x), p.g, &par);
    Grp<T>::store(d, par, m);
}

template <typename T>
__global__ void maxpool3x3s2_kernel(TensorRef src, TensorRef dst) {
    constexpr int GN = Grp<T>::N;
    const PG p = decode_pg<T>(dst);
    if (!p.live) return;
    float m[GN], v[GN];
#pragma unroll
    for (int j = 0; j < GN; ++j) m[j] = -INFINITY;
    int par;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * p.y - 1 + ky;
        if (iy < 0 || iy >= src.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * p.x - 1 + kx;
            if (ix < 0 || ix >= src.W) continue;
            const char* s = grp_ptr<T>(src, pix_index(src, p.n, iy, ix), p.g, &par);
            Grp<T>::load(s, par, v);
#pragma unroll
            for (int j = 0; j < GN; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    }
    char* d = grp_ptr<T>(dst, pix_index(dst, p.n, p.y, p.x), p.g, &par);
    Grp<T>::store(d, par, m);
}

// acc + w * (the low | high f16 half of `packed`), the f16 read by the FMA itself (v_fma_mix_f32: no separate conversion).  The compiler
// does not form this instruction from (float)h * w + acc here; only for kernels without MFMAs (cv_kernels.h: split_pair's note).
__device__ __forceinline__ float fma_mix_f16(unsigned packed, int high, float w, float acc) {
    float r;
#if defined(__HIP_DEVICE_COMPILE__)
    if (high) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w), "v"(acc));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(w), "v"(acc));
#else
    r = acc + w * (float)packed * 0.f + (float)high;      // host pass of the single-source compile: never executed
#endif
    return r;
}

// source pair and upper weight of one output index: torch's area_pixel_compute_source_index / compute_source_index_and_lambda with
// align_corners -- the product scale * index is ROUNDED before the integer part is taken off (no fused multiply-subtract: with
// contraction left to the compiler the two up-sample kernels below differed in the last bit of their weights)
__device__ __forceinline__ void bilinear_axis(float scale, int index, int in_size, int& i0, int& i1, float& l1) {
#pragma clang fp contract(off)
    const float f = scale * (float)index;
    i0 = (int)f;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = f - (float)i0;
}

// torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1); weights (1-l, l) in f32.
// Grid = (row segments, output rows, images): the row's source lines and vertical weights are wave-uniform and a thread finds its
// (column, channel group) with one 32-bit division (r03: the flat one-thread-per-element form decoded its index with three 64-bit
// divisions -- ~150 instructions around five 32-byte memory operations; a row-per-workgroup loop was slower still: fewer loads in flight).
template <typename T>
__global__ __launch_bounds__(256) void upsample_bilinear2x_kernel(TensorRef src, TensorRef dst, float mul, float sy, float sx, unsigned* flag, unsigned layer_id) {
    // sy, sx = (in - 1) / (out - 1) as float divisions done ONCE on the host (the same IEEE quotient torch's area_pixel_compute_scale
    // takes; per lane they were two software divisions, ~20 vector instructions)
    constexpr int GN = Grp<T>::N;
    const int y = blockIdx.y, n = blockIdx.z;
    const unsigned groups = (unsigned)(dst.C / GN);
    int y0, y1;
    float ly1;
    bilinear_axis(sy, y, src.H, y0, y1, ly1);
    const float ly0 = 1.f - ly1;
    const size_t row0 = pix_index(src, n, y0, 0), row1 = pix_index(src, n, y1, 0), orow = pix_index(dst, n, y, 0);
    float bad = 0.f;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (unsigned)dst.W * groups) {
        const unsigned x = idx / groups, g = idx - x * groups;
        int x0, x1;
        float lx1;
        bilinear_axis(sx, (int)x, src.W, x0, x1, lx1);
        const float lx0 = 1.f - lx1;
        if constexpr (__is_same(T, split_t)) {
            // split-f16 (round 5): the kernel is bound by vector issue, not by HBM (~200 instructions per 32 stored bytes: every tap
            // converted hi and lo to f32 and added them before the blend).  The taps stay f16: each of the eight products per value is one
            // mixed-precision FMA that reads the f16 half directly (v_fma_mix_f32), the four weights are formed once per lane, and the
            // numeric guard runs on the packed hi pairs.  Rounding: the weights ly*lx are rounded once instead of the nested form's
            // partial sums -- a last-bit difference (the parity bars are 1e-5 on the op, 1e-3 end to end).
            // 32-bit byte offsets from the (wave-uniform) tensor bases: tensors stay below 4 GiB (checked by the engine), and the per-tap
            // 64-bit address arithmetic was a third of this kernel's vector instructions
            const char* const sbase = reinterpret_cast<const char*>(src.base);
            const unsigned spix = (unsigned)src.Cs * 4u, goff = (unsigned)src.Coff * 4u + g * 32u;
            const unsigned o00 = ((unsigned)row0 + (unsigned)x0) * spix + goff, o10 = ((unsigned)row1 + (unsigned)x0) * spix + goff;
            const unsigned dxb = (unsigned)(x1 - x0) * spix;
            const char* const p00 = sbase + o00;
            const char* const p01 = sbase + (o00 + dxb);
            const char* const p10 = sbase + o10;
            const char* const p11 = sbase + (o10 + dxb);
            const half8 a00 = *reinterpret_cast<const half8*>(p00), b00 = *reinterpret_cast<const half8*>(p00 + 16);
            const half8 a01 = *reinterpret_cast<const half8*>(p01), b01 = *reinterpret_cast<const half8*>(p01 + 16);
            const half8 a10 = *reinterpret_cast<const half8*>(p10), b10 = *reinterpret_cast<const half8*>(p10 + 16);
            const half8 a11 = *reinterpret_cast<const half8*>(p11), b11 = *reinterpret_cast<const half8*>(p11 + 16);
            const float w00 = ly0 * lx0 * mul, w01 = ly0 * lx1 * mul, w10 = ly1 * lx0 * mul, w11 = ly1 * lx1 * mul;   // mul = 2^k: exact
            float o[8];
            typedef unsigned u4t __attribute__((ext_vector_type(4)));
            const u4t ta[4] = {__builtin_bit_cast(u4t, a00), __builtin_bit_cast(u4t, a01), __builtin_bit_cast(u4t, a10), __builtin_bit_cast(u4t, a11)};
            const u4t tb[4] = {__builtin_bit_cast(u4t, b00), __builtin_bit_cast(u4t, b01), __builtin_bit_cast(u4t, b10), __builtin_bit_cast(u4t, b11)};
            const float wt[4] = {w00, w01, w10, w11};
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
            // tap-major: the eight accumulators are independent, so no FMA waits for the one before it (value-major, the compiler padded
            // every dependent pair of these asm instructions with an s_nop)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fma_mix_f16(ta[t][j >> 1], j & 1, wt[t], o[j]);   // hi + lo of a tap: the chunk order
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fma_mix_f16(tb[t][j >> 1], j & 1, wt[t], o[j]);   // inside the group does not matter
            }
            const int opar = (dst.Coff / 8 + (int)g) & 1;
            char* const dp = reinterpret_cast<char*>(dst.base) + (((unsigned)orow + x) * ((unsigned)dst.Cs * 4u) + (unsigned)dst.Coff * 4u + g * 32u);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            u4 hi, lo;
            h2 g2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                unsigned hp, lp;
                split_pair(o[j], o[j + 1], hp, lp);                   // behind the FMAs that consume nothing else: no MFMA in this kernel
                g2 = __builtin_elementwise_fma(__builtin_bit_cast(h2, hp), h2{(half_t)0.f, (half_t)0.f}, g2);
                hi[j / 2] = hp; lo[j / 2] = lp;
            }
            bad = __builtin_fmaf((float)g2[0] + (float)g2[1], 0.f, bad);
            *reinterpret_cast<u4*>(dp + (opar ? 16 : 0)) = hi;
            *reinterpret_cast<u4*>(dp + (opar ? 0 : 16)) = lo;
        } else {
        float v00[GN], v01[GN], v10[GN], v11[GN], o[GN];
        int par;
        Grp<T>::load(grp_ptr<T>(src, row0 + x0, g, &par), par, v00);
        Grp<T>::load(grp_ptr<T>(src, row0 + x1, g, &par), par, v01);
        Grp<T>::load(grp_ptr<T>(src, row1 + x0, g, &par), par, v10);
        Grp<T>::load(grp_ptr<T>(src, row1 + x1, g, &par), par, v11);
#pragma unroll
        for (int j = 0; j < GN; ++j)
            o[j] = (ly0 * (lx0 * v00[j] + lx1 * v01[j]) + ly1 * (lx0 * v10[j] + lx1 * v11[j])) * mul;   // mul = 2^k: exact
        Grp<T>::store(grp_ptr<T>(dst, orow + x, g, &par), par, o, bad);
        }
    }
    report_bad(flag, layer_id, bad);
}

// split-f16, 2 x 2 OUTPUT pixels per lane (round 5).  With align_corners=True and an exact factor of two the output rows 2k - 1 and 2k
// blend the SAME two source rows (k - 1, k) -- likewise the columns -- so a lane that owns the block {2k - 1, 2k} x {2j - 1, 2j} loads
// four source pixels (8 x 16 bytes) for four outputs instead of 32 x 16 bytes: the one-output-per-lane kernel above moved its bytes
// at 0.44 of the HBM rate on tap loads through L1, not on arithmetic.  Each output keeps the exact operation order of that kernel
// (weights ly * lx * mul rounded once, taps accumulated 00, 01, 10, 11), so both kernels produce the same bits; a block whose rows or
// columns do NOT share their source pair (never for a factor of two; checked, not assumed: the indices come from the same float
// products torch forms) falls back to that order with its own loads.  Grid = (segments of (W_in + 1) x groups, H_in + 1, images).
__device__ __forceinline__ void upsample_split_emit(const TensorRef& dst, size_t opix, unsigned g, const unsigned (&t)[4][8], float w00, float w01,
                                                    float w10, float w11, float& bad) {
    const float wt[4] = {w00, w01, w10, w11};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fma_mix_f16(t[k][j >> 1], j & 1, wt[k], o[j]);          // first 16-byte half of the group
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fma_mix_f16(t[k][4 + (j >> 1)], j & 1, wt[k], o[j]);    // second half
    }
    const int opar = (dst.Coff / 8 + (int)g) & 1;
    char* const dp = reinterpret_cast<char*>(dst.base) + ((unsigned)opix * ((unsigned)dst.Cs * 4u) + (unsigned)dst.Coff * 4u + g * 32u);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    u4 hi, lo;
    h2 g2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        unsigned hp, lp;
        split_pair(o[j], o[j + 1], hp, lp);
        g2 = __builtin_elementwise_fma(__builtin_bit_cast(h2, hp), h2{(half_t)0.f, (half_t)0.f}, g2);
        hi[j / 2] = hp; lo[j / 2] = lp;
    }
    bad = __builtin_fmaf((float)g2[0] + (float)g2[1], 0.f, bad);
    *reinterpret_cast<u4*>(dp + (opar ? 16 : 0)) = hi;
    *reinterpret_cast<u4*>(dp + (opar ? 0 : 16)) = lo;
}

__global__ __launch_bounds__(256) void upsample_bilinear2x_split2x2_kernel(TensorRef src, TensorRef dst, float mul, float sy, float sx,
                                                                           unsigned* flag, unsigned layer_id) {
    const int k = blockIdx.y, n = blockIdx.z;
    const unsigned groups = (unsigned)(dst.C / 8);
    float bad = 0.f;
    // the block's two output rows (wave-uniform): source pair and weights of each, from torch's float products
    int ry[2] = {2 * k - 1, 2 * k}, y0[2], y1[2];
    float ly1[2];
    bool rv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        rv[r] = ry[r] >= 0 && ry[r] < dst.H;
        bilinear_axis(sy, rv[r] ? ry[r] : 0, src.H, y0[r], y1[r], ly1[r]);
    }
    const int rbase = rv[0] ? 0 : 1;                          // the row whose source pair the lane loads
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (unsigned)(src.W + 1) * groups) {
        const unsigned j = idx / groups, g = idx - j * groups;
        int cx[2] = {2 * (int)j - 1, 2 * (int)j}, x0[2], x1[2];
        float lx1[2];
        bool cv_[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            cv_[c] = cx[c] >= 0 && cx[c] < dst.W;
            bilinear_axis(sx, cv_[c] ? cx[c] : 0, src.W, x0[c], x1[c], lx1[c]);
        }
        const int cbase = cv_[0] ? 0 : 1;
        const char* const sbase = reinterpret_cast<const char*>(src.base);
        const unsigned spix = (unsigned)src.Cs * 4u, goff = (unsigned)src.Coff * 4u + g * 32u;
        auto load4 = [&](int yy0, int yy1, int xx0, int xx1, unsigned (&t)[4][8]) {
            const unsigned r0 = (unsigned)pix_index(src, n, yy0, 0), r1 = (unsigned)pix_index(src, n, yy1, 0);
            const unsigned off[4] = {(r0 + (unsigned)xx0) * spix + goff, (r0 + (unsigned)xx1) * spix + goff, (r1 + (unsigned)xx0) * spix + goff,
                                     (r1 + (unsigned)xx1) * spix + goff};
            typedef unsigned u4t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u4t a = *reinterpret_cast<const u4t*>(sbase + off[q]), b = *reinterpret_cast<const u4t*>(sbase + off[q] + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) { t[q][i] = a[i]; t[q][4 + i] = b[i]; }
            }
        };
        unsigned t[4][8];
        load4(y0[rbase], y1[rbase], x0[cbase], x1[cbase], t);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (!rv[r]) continue;
            const bool rows_shared = y0[r] == y0[rbase] && y1[r] == y1[rbase];
            const float ly0 = 1.f - ly1[r];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!cv_[c]) continue;
                const float lx0 = 1.f - lx1[c];
                const float w00 = ly0 * lx0 * mul, w01 = ly0 * lx1[c] * mul, w10 = ly1[r] * lx0 * mul, w11 = ly1[r] * lx1[c] * mul;   // mul = 2^k: exact
                const size_t opix = pix_index(dst, n, ry[r], cx[c]);
                if (rows_shared && x0[c] == x0[cbase] && x1[c] == x1[cbase]) {
                    upsample_split_emit(dst, opix, g, t, w00, w01, w10, w11, bad);
                } else {                                      // not reached for a factor of two; kept so that the kernel is exact for any geometry
                    unsigned u[4][8];
                    load4(y0[r], y1[r], x0[c], x1[c], u);
                    upsample_split_emit(dst, opix, g, u, w00, w01, w10, w11, bad);
                }
            }
        }
    }
    report_bad(flag, layer_id, bad);
}

// ---- OutConv 1x1 (C -> 1) + bias, fused sigmoid/threshold mask ------------------------------------
// C/N lanes share one pixel (one group each); xor-shuffle reduce inside the lane group.
template <typename T>
__global__ void outc_1x1_kernel(TensorRef src, const float* __restrict__ w, const float* __restrict__ bias, float mul,
                                float* __restrict__ logits, uint8_t* __restrict__ mask, float thr, unsigned* flag,
                                unsigned layer_id) {
    constexpr int GN = Grp<T>::N;
    const int lpp = src.C / GN;                               // power of two, <= 64
    const size_t npix = (size_t)src.N * src.H * src.W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int part = (int)(idx % lpp);
    size_t pix = idx / lpp;
    const bool live = pix < npix;
    if (!live) pix = npix - 1;
    size_t r = pix;
    const int x = (int)(r % src.W); r /= src.W;
    const int y = (int)(r % src.H);
    const int n = (int)(r / src.H);
    float v[GN];
    int par;
    Grp<T>::load(grp_ptr<T>(src, pix_index(src, n, y, x), part, &par), par, v);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < GN; ++j) acc += v[j] * w[part * GN + j];
    for (int m = 1; m < lpp; m <<= 1) acc += __shfl_xor(acc, m);
    if (live && part == 0) {
        const float l = acc * mul + bias[0];
        logits[pix] = l;
        if (mask) mask[pix] = (1.f / (1.f + __expf(-l))) > thr ? 255 : 0;
        report_bad(flag, layer_id, l * 0.f);
    }
}

// ---- ResNet stem: conv 7x7 s2 p3, 1 -> 64, + BN affine + ReLU -------------------------------------
// One workgroup per 64x64 square.  The zero-bordered plane sits in LDS; each lane owns one output pixel at
// a time, holds its 7x7 patch in VGPRs and runs the 64x49 filter bank from the scalar cache (wave-uniform
// weights -> s_load + v_fma with an SGPR operand), i.e. a pure VALU f32 kernel.
template <typename T, typename XT>
__global__ __launch_bounds__(256) void stem7x7_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, TensorRef dst) {
    constexpr int IN = 64, P = 3, LD = IN + 2 * P;            // 70
    constexpr int GN = Grp<T>::N;
    __shared__ float plane[LD * LD];
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < LD * LD; i += 256) plane[i] = 0.f;
    __syncthreads();
    const XT* xs = x + (size_t)n * IN * IN;
    for (int i = tid; i < IN * IN; i += 256) {
        float v = (float)xs[i];
        if (sizeof(XT) == 1) v = v / 255.f;
        plane[(i / IN + P) * LD + (i % IN) + P] = v;
    }
    __syncthreads();
    constexpr int OUT = 32;
    for (int it = 0; it < OUT * OUT / 256; ++it) {
        const int pid = it * 256 + tid;
        const int oy = pid / OUT, ox = pid % OUT;
        float patch[49];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) patch[ky * 7 + kx] = plane[(2 * oy + ky) * LD + 2 * ox + kx];
        const size_t opix = pix_index(dst, n, oy, ox);
        // 8 output channels per trip: 392 wave-uniform weights stream through SGPRs, results leave as
        // 16-B stores.  Not unrolled on purpose: keeps the scalar loads inside the loop (no LICM hoist).
#pragma unroll 1
        for (int cb = 0; cb < 64; cb += 8) {
            const float* wc = w + cb * 49;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 49; ++k) a = __builtin_fmaf(patch[k], wc[c * 49 + k], a);
                a = a * scale[cb + c] + shift[cb + c];
                v[c] = a > 0.f ? a : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; i += GN) {
                int par;
                char* d = grp_ptr<T>(dst, opix, (cb + i) / GN, &par);
                Grp<T>::store(d, par, v + i);
            }
        }
    }
}

// ---- ResNet stem on the matrix cores, fused with the 3x3/s2/p1 max-pool (f16 and split-f16 engines) ----------
// conv 7x7 s2 p3 (1 -> 64) + BN + ReLU + max_pool2d(3, 2, 1):  (n,1,64,64) -> 64 ch @ 16x16, one pass, nothing but the
// pooled result leaves the CU.  GEMM view per square: D[ch][pix] = sum_k W[ch][k] * patch[pix][k], K = 7 rows x 8
// (kx padded 7 -> 8, ky padded 7 -> 8) = 64 = two 16x16x32 MFMA k-steps; lane group q of k-step ks owns filter row
// ky = 4*ks + q, whose 8 taps are 4 consecutive LDS dwords of the zero-bordered input plane -> the B fragment is four
// ds_read_b32, no packing.  One 16-lane pixel fragment = one pooled output row; the 9 pool-window positions are
// computed one after the other (2.25x recompute of the tiny conv instead of staging 32x32x64 outputs in LDS) and
// max-reduced in registers.  Out-of-range window positions contribute 0, which equals -inf padding after ReLU.
// SPLIT = split-f16 ARITHMETIC (hi + lo planes, three products); it is what a split_t tensor implies, and the f16r engine asks
// for it with T = half_t: its stem computes at f32 grade and leaves the pooled result twice, as the f16 tensor layer1.0.conv1
// consumes and as the f32 twin (dst.base32) the residual trunk starts from.
template <typename T, typename XT, bool SPLIT = sizeof(T) == 4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_pool_mfma_kernel(const XT* __restrict__ x, int n,
                                                             const half8* __restrict__ wpk,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float in_mul, TensorRef dst,
                                                             unsigned* flag, unsigned layer_id) {
    constexpr int LD = 98, ROWS = 72, PAD = 5;              // LD/2 = 49 dwords: odd rows land on the other bank half
    constexpr int NPL = SPLIT ? 2 : 1;                      // hi (+ lo) plane
    // two sets of planes: square n+1 is fetched into registers before square n is computed and written to the other set after it,
    // so the global-load latency and the conversion pass hide behind the MFMA phase and one barrier per square is left (r03:
    // 42 % of the wave cycles were waits with a single set)
    __shared__ __attribute__((aligned(16))) half_t plane[2][NPL][ROWS * LD];
    __shared__ __attribute__((aligned(16))) char stage[4][16 * 272];      // one pooled row per wave on its way out (see the stores below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;

    half8 ah[2][4], al[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            ah[ks][f] = wpk[((0 * 2 + ks) * 4 + f) * 64 + lane];
            al[ks][f] = SPLIT ? wpk[((1 * 2 + ks) * 4 + f) * 64 + lane] : ah[ks][f];
        }
    float sc[16], sh[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sc[i] = scale[q * 16 + i]; sh[i] = shift[q * 16 + i]; }

    for (int i = tid; i < 2 * NPL * ROWS * LD; i += 256) (&plane[0][0][0])[i] = (half_t)0.f;

    XT pre[16];                                              // this thread's 16 pixels of the NEXT square
    auto fetch = [&](int sq) __attribute__((always_inline)) {
        const XT* xs = x + (size_t)sq * 4096;
#pragma unroll
        for (int k = 0; k < 16; ++k) pre[k] = xs[tid + 256 * k];
    };
    auto deposit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = tid + 256 * k;
            float v = (float)pre[k];
            if (sizeof(XT) == 1) v = v / 255.f;
            v *= in_mul;                                     // power-of-two range factor of the input plane (exact)
            const half_t hi = (half_t)v;
            const int at = ((i >> 6) + PAD) * LD + (i & 63) + PAD;
            plane[buf][0][at] = hi;
            if (SPLIT) plane[buf][NPL - 1][at] = (half_t)(v - (float)hi);
        }
    };
    float bad = 0.f;
    int buf = 0;
    if ((int)blockIdx.x < n) fetch(blockIdx.x);
    __syncthreads();                                         // the zero fill is complete
    if ((int)blockIdx.x < n) deposit(0);
    __syncthreads();
    for (int sq = blockIdx.x; sq < n; sq += gridDim.x, buf ^= 1) {
        const int nxt = sq + gridDim.x;
        if (nxt < n) fetch(nxt);                             // in flight during this square's MFMA phase
        const unsigned* p32h = reinterpret_cast<const unsigned*>(&plane[buf][0][0]);
        const unsigned* p32l = reinterpret_cast<const unsigned*>(&plane[buf][NPL - 1][0]);
        // Each wave owns four consecutive pooled rows = conv-output rows 8w-1 .. 8w+7.  A conv row is computed ONCE, as its
        // even-column and odd-column fragments (lane l15 <-> columns 2*l15 and 2*l15+1); the third pool-window column
        // (2*l15-1) is the odd fragment shifted by one lane (DPP row_shr:1, zero fill = the window's left padding), and the
        // conv row shared by two vertically adjacent windows is carried in registers: 9 x 2 fragment computations per four
        // pooled rows instead of the 4 x 9 of a window-by-window walk (r01).  Out-of-range rows / columns contribute 0, which
        // equals -inf padding after ReLU.
        auto conv_row_max = [&](int cy, float* hm) __attribute__((always_inline)) {   // max over the three window columns: 16 channels per lane
            float ev[16], od[16];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                f4 acc[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int row = 2 * cy + 2 + ks * 4 + q;                   // plane row of filter row ky = 4*ks+q
                    const int dw = (row * LD + 2 * (2 * l15 + par) + 2) >> 1;  // conv column 2*l15 + par
                    union { unsigned u[4]; half8 h; } bh, bl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { bh.u[j] = p32h[dw + j]; bl.u[j] = SPLIT ? p32l[dw + j] : 0u; }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks][f], bh.h, acc[f], 0, 0, 0);
                        if (SPLIT) {
                            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks][f], bh.h, acc[f], 0, 0, 0);
                            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks][f], bl.h, acc[f], 0, 0, 0);
                        }
                    }
                }
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        // BN affine only: max_pool(relu(x)) == relu(max_pool(x)), and the zero that stands for the pool's padding is
                        // harmless inside a maximum that is clamped at zero afterwards -- one ReLU per pooled value instead of one per
                        // conv value (2.25x fewer).  Two values per v_pk_fma_f32.
                        const f2 a = {acc[f][r], acc[f][r + 1]}, m = {sc[f * 4 + r], sc[f * 4 + r + 1]}, b = {sh[f * 4 + r], sh[f * 4 + r + 1]};
                        const f2 v = __builtin_elementwise_fma(a, m, b);
                        if (par == 0) { ev[f * 4 + r] = v[0]; ev[f * 4 + r + 1] = v[1]; } else { od[f * 4 + r] = v[0]; od[f * 4 + r + 1] = v[1]; }
                    }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(od[i]), 0x111, 0xf, 0xf, true));   // od of lane - 1
                hm[i] = __builtin_fmaxf(__builtin_fmaxf(ev[i], od[i]), left);
            }
        };
        float carry[16];
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) carry[i] = 0.f;                       // conv row -1: padding
        } else {
            conv_row_max(8 * wave - 1, carry);
        }
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int py = 4 * wave + j;
            float mid[16], low[16], best[16];
            conv_row_max(2 * py, mid);
            conv_row_max(2 * py + 1, low);
#pragma unroll
            for (int i = 0; i < 16; ++i) { best[i] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(carry[i], mid[i]), low[i]), 0.f); carry[i] = low[i]; }
            // A pooled row of the square = 16 pixels x 64 channels = ONE contiguous run in memory (2-4 KB; the launcher checks that the
            // tensor holds exactly the stem's 64 channels).  The MFMA leaves a lane with 16 channels of one pixel, i.e. 64 scattered
            // 16-byte pieces per store instruction; through the wave's LDS corner the run leaves as 1 KB per instruction in full lines
            // (r05_tuning.md section 15).  Wave-private staging: DS operations of a wave execute in order.  The split-f16 instantiation keeps
            // its direct stores: there the staging costs 10-12 spilled registers and the kernel is bound by its matrix work anyway.
            const size_t opix = pix_index(dst, sq, py, l15);
            if constexpr (!__is_same(T, half_t)) {
#pragma unroll
                for (int i = 0; i < 16; i += Grp<T>::N) {
                    int par;
                    char* d = grp_ptr<T>(dst, opix, (q * 16 + i) / Grp<T>::N, &par);
                    Grp<T>::store(d, par, best + i, bad);
                }
            } else {
                constexpr int ESZ = Grp<T>::BYTES / Grp<T>::N, PB = 64 * ESZ, SRA = PB + 16;
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                char* const stg = &stage[wave][0];
#pragma unroll
                for (int i = 0; i < 16; i += Grp<T>::N)
                    Grp<T>::store(stg + l15 * SRA + (q * 16 + i) * ESZ, ((q * 16 + i) / Grp<T>::N) & 1, best + i, bad);
                wave_lds_sync();
                char* const grow = reinterpret_cast<char*>(dst.base) + pix_index(dst, sq, py, 0) * (size_t)PB;
#pragma unroll
                for (int t = 0; t < 16 * PB / 1024; ++t) {
                    const int o = t * 1024 + lane * 16;
                    *reinterpret_cast<u4*>(grow + o) = *reinterpret_cast<const u4*>(stg + (o / PB) * SRA + (o % PB));
                }
                wave_lds_sync();
                if constexpr (__is_same(T, half_t) && SPLIT) {
                    if (dst.base32) {
#pragma unroll
                        for (int i = 0; i < 16; i += 4)
                            *reinterpret_cast<f4*>(stg + l15 * 272 + (q * 16 + i) * 4) = f4{best[i], best[i + 1], best[i + 2], best[i + 3]};
                        wave_lds_sync();
                        char* const grow32 = reinterpret_cast<char*>(dst.base32) + pix_index(dst, sq, py, 0) * (size_t)256;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int o = t * 1024 + lane * 16;
                            *reinterpret_cast<u4*>(grow32 + o) = *reinterpret_cast<const u4*>(stg + (o / 256) * 272 + (o % 256));
                        }
                        wave_lds_sync();
                    }
                }
            }
        }
        if (nxt < n) deposit(buf ^ 1);                       // the other set: nobody reads it during this square
        __syncthreads();                                     // next square's planes visible; every read of this square's set done
    }
    report_bad(flag, layer_id, bad);
}


// ---- UNet first layer on the matrix cores, straight from the caller's image ---------------------------------------
// inc.double_conv.0: conv 3x3 p1 (3 -> 64) + BN + ReLU, fused with the input packing (u8 HWC / 255 or f32 NCHW -> internal
// layout).  The layer is 0.2 % of the network's MACs but writes 16.8 MB per image: it is bound by that store stream, and the
// generic implicit-GEMM kernel (three 128-byte K stages over an 8-channel padded copy of the image, one 128-pixel tile per
// workgroup) spent its time in per-workgroup prologues instead.  Here K = 3 x 3 x 3 = 27 -> 32 = ONE 16x16x32 MFMA k-step
// (x3 products for split-f16); a workgroup owns an 8-row band of one image, keeps the 10 x 258 x 3 input patch in LDS as
// f32, and every wave walks 2 rows x 16 fragments: eight ds_read_b32 build the B fragment (lane group q holds k = 8q..8q+7,
// k = (ky*3 + kx)*3 + c), 4 x (1 | 3) MFMAs give the lane 16 consecutive channels of its pixel, and the wave-private LDS
// transpose of conv_igemm.hip turns them into 2 KB-contiguous stores.  The 8-channel padded input tensor and its packing
// kernel disappear.
template <typename T, typename XT>
__global__ __launch_bounds__(256) void inc0_mfma_kernel(const XT* __restrict__ x, const half8* __restrict__ wpk,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float in_mul, TensorRef dst, unsigned* flag, unsigned layer_id) {
    constexpr bool SPLIT = sizeof(T) == 4;
    constexpr int W = 256, BAND = 8, PR = BAND + 2, PW = 260;      // patch rows / row pitch (floats): cols -1..256 + pad
    constexpr int SROW = 272;                                        // staging row pitch (bytes), as conv_igemm.hip
    __shared__ __attribute__((aligned(16))) float patch[3 * PR * PW + 4];
    __shared__ __attribute__((aligned(16))) char stage_mem[4 * 16 * SROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;
    const int n = blockIdx.x / (W / BAND), band = blockIdx.x % (W / BAND);
    const int y0 = band * BAND;

    // ---- input patch: rows y0-1 .. y0+8, all 256 columns, 3 channels; zero outside the image -------------------------
    float bad = 0.f;
    if (tid < 4) patch[3 * PR * PW + tid] = 0.f;                      // the zero slot the padded k indices read
    for (int i = tid; i < 3 * PR; i += 256) { patch[i * PW] = 0.f; patch[i * PW + W + 1] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < PR; ++r) {
            const int gy = y0 - 1 + r;
            float v = 0.f;
            if (gy >= 0 && gy < W) {
                if (sizeof(XT) == 1) v = (float)x[((size_t)(n * W + gy) * W + tid) * 3 + c] / 255.f;
                else v = (float)x[((size_t)(n * 3 + c) * W + gy) * W + tid];
            }
            bad = __builtin_fmaf(v, 0.f, bad);
            patch[(c * PR + r) * PW + tid + 1] = v * in_mul;
        }
    report_bad(flag, 0u, bad);                                       // layer id 0 = the caller's input tensor

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    half8 ah[4], al[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        ah[f] = wpk[(0 * 4 + f) * 64 + lane];
        al[f] = SPLIT ? wpk[(1 * 4 + f) * 64 + lane] : ah[f];
    }
    float sc[16], sh[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sc[i] = scale[q * 16 + i]; sh[i] = shift[q * 16 + i]; }
    int koff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = q * 8 + j;
        const int tap = k / 3, c = k - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        koff[j] = k < 27 ? (c * PR + ky) * PW + kx : 3 * PR * PW;   // (float index relative to the pixel's patch origin) | zero slot
    }
    __syncthreads();

    char* const stg = stage_mem + wave * (16 * SROW);
    float out_bad = 0.f;
    for (int it = 0; it < 2 * (W / 16); ++it) {
        const int r = wave * 2 + it / (W / 16), cb = it % (W / 16);   // row of the band, 16-pixel block of the row
        const int pbase = r * PW + cb * 16 + l15;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = patch[(q * 8 + j < 27 ? pbase : 0) + koff[j]];
        half8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { bh[j] = (half_t)v[j]; bl[j] = (half_t)(v[j] - (float)bh[j]); }
        f4 acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bh, f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (SPLIT) {
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[f], bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[f], bl, acc[f], 0, 0, 0);
            }
        }
        // BN + ReLU, then the wave-private transpose: lane (pixel l15, group q) holds 16 consecutive channels
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f4 t;
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_fmaxf(acc[f][k] * sc[f * 4 + k] + sh[f * 4 + k], 0.f);
            *reinterpret_cast<f4*>(stg + l15 * SROW + (q * 16 + f * 4) * 4) = t;
        }
        wave_lds_sync();
        const size_t row_pix = pix_index(dst, n, y0 + r, cb * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {                                  // 16 px x 8 groups of 8 channels = 128 units
            const int unit = lane + 64 * i, px = unit >> 3, g = unit & 7;
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                const f4 t = *reinterpret_cast<const f4*>(stg + px * SROW + (g * 8 + j) * 4);
                w[j] = t[0]; w[j + 1] = t[1]; w[j + 2] = t[2]; w[j + 3] = t[3];
            }
            int par;
            char* d = grp_ptr<T>(dst, row_pix + px, g, &par);
            Grp<T>::store(d, par, w, out_bad);
        }
        wave_lds_sync();
    }
    report_bad(flag, layer_id, out_bad);
}

// ---- head: adaptive_avg_pool2d(1) + Linear(512 -> 13) (+ softmax) ----------------------------------
// Persistent waves, one square at a time.  Lane l always owns channels 8l .. 8l+7, so its 13 x 8 fc weights are loaded ONCE
// into registers and reused for every square the wave walks (r01 re-read them from L2 for each square: 104 vector loads per
// wave and square, which -- not the 2 KB of activations -- set the kernel's time).  Per square: 32-byte loads of the lane's
// channels at each pixel, 104 FMAs, then the 13 partial sums are reduced across the 64 lanes by a halving butterfly (each step
// exchanges half of the live values with the partner lane: 7 + 4 + 2 + 1 shuffles, then 2 for the last four lanes) instead of
// 13 full butterflies (78 shuffles); soft-max runs across the 13 lanes that end up holding one logit each.
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(TensorRef src, const float* __restrict__ w,
                                                   const float* __restrict__ b, float mul, float* __restrict__ out,
                                                   int softmax, unsigned* flag, unsigned layer_id) {
    constexpr int NC = 13;
    constexpr int GN = Grp<T>::N, GPL = 8 / GN;              // groups per lane: 1 (f16, split-f16) or 2 (f32)
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float wr[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const f4 t = *reinterpret_cast<const f4*>(w + k * 512 + lane * 8 + j);
            wr[k][j] = t[0]; wr[k][j + 1] = t[1]; wr[k][j + 2] = t[2]; wr[k][j + 3] = t[3];
        }
    // which logit this lane holds after the halving butterfly, and its bias
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
    const int slot = b4 * 4 + b3 * 2 + b2;
    const bool holds = b5 ? slot < 6 : slot < 7;
    const int cls = b5 * 7 + slot;
    const float bias = holds ? b[cls] : 0.f;
    const float inv = mul / (float)(src.H * src.W);           // mul = 2^exp of the stored tensor (exact)
    float bad = 0.f;
    for (int n = wave0; n < src.N; n += nwaves) {
        float s8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s8[j] = 0.f;
        for (int y = 0; y < src.H; ++y)
            for (int x = 0; x < src.W; ++x) {
                const size_t pix = pix_index(src, n, y, x);
#pragma unroll
                for (int g = 0; g < GPL; ++g) {
                    int par;
                    float v[GN];
                    Grp<T>::load(grp_ptr<T>(src, pix, lane * GPL + g, &par), par, v);
#pragma unroll
                    for (int j = 0; j < GN; ++j) s8[g * GN + j] += v[j];
                }
            }
        float acc[NC + 1];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a = __builtin_fmaf(s8[j] * inv, wr[k][j], a);
            acc[k] = a;
        }
        acc[NC] = 0.f;
        // halving butterfly: 14 -> 7 -> 4 -> 2 -> 1 live values, the upper lane of each pair keeps the upper half
        float v7[8];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float r = __shfl_xor(b5 ? acc[i] : acc[7 + i], 32);
            v7[i] = (b5 ? acc[7 + i] : acc[i]) + r;
        }
        v7[7] = 0.f;
        float v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = __shfl_xor(b4 ? v7[i] : v7[4 + i], 16);
            v4[i] = (b4 ? v7[4 + i] : v7[i]) + r;
        }
        float v2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float r = __shfl_xor(b3 ? v4[i] : v4[2 + i], 8);
            v2[i] = (b3 ? v4[2 + i] : v4[i]) + r;
        }
        float v = (b2 ? v2[1] : v2[0]) + __shfl_xor(b2 ? v2[0] : v2[1], 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        v += bias;
        if (holds) bad = __builtin_fmaf(v, 0.f, bad);
        if (softmax) {
            float mx = holds ? v : -INFINITY;
#pragma unroll
            for (int m = 32; m >= 4; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
            const float e = holds ? __expf(v - mx) : 0.f;
            float sum = e;
#pragma unroll
            for (int m = 32; m >= 4; m >>= 1) sum += __shfl_xor(sum, m);
            v = e / sum;
        }
        if (holds && (lane & 3) == 0) out[(size_t)n * NC + cls] = v;
    }
    report_bad(flag, layer_id, bad);
}

__global__ void softmax13_kernel(const float* __restrict__ logits, int n, float* __restrict__ probs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[13];
    float mx = -3.4e38f;
#pragma unroll
    for (int j = 0; j < 13; ++j) { v[j] = logits[(size_t)i * 13 + j]; mx = fmaxf(mx, v[j]); }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 13; ++j) { v[j] = __expf(v[j] - mx); s += v[j]; }
    const float r = 1.f / s;
#pragma unroll
    for (int j = 0; j < 13; ++j) probs[(size_t)i * 13 + j] = v[j] * r;
}

// ---- MFMA lane-map probes (same fragment addressing as conv_igemm.hip) ---------------------------
__global__ void mfma_probe_f16_kernel(const half_t* a, const half_t* b, float* d) {
    const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
    const half8 fa = *reinterpret_cast<const half8*>(a + r * 32 + q * 8);
    const half8 fb = *reinterpret_cast<const half8*>(b + r * 32 + q * 8);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[(q * 4 + i) * 16 + r] = acc[i];      // D[row = A row][col = B row]
}
__global__ void mfma_probe_f32_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
    const f4 fa = *reinterpret_cast<const f4*>(a + r * 16 + q * 4);
    const f4 fb = *reinterpret_cast<const f4*>(b + r * 16 + q * 4);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) d[(q * 4 + i) * 16 + r] = acc[i];
}

// ---- host wrappers ----------------------------------------------------------------------------------
#define CV_DISPATCH(KERN, WORK, ...)                                                                         \
    do {                                                                                                     \
        const dim3 g_(grid_for(WORK)), b_(256);                                                              \
        if (dt == kF16) hipLaunchKernelGGL(KERN<half_t>, g_, b_, 0, s, __VA_ARGS__);                         \
        else if (dt == kSplit) hipLaunchKernelGGL(KERN<split_t>, g_, b_, 0, s, __VA_ARGS__);                 \
        else hipLaunchKernelGGL(KERN<float>, g_, b_, 0, s, __VA_ARGS__);                                     \
        return hipGetLastError();                                                                            \
    } while (0)

static inline float pow2f(int e) { return ldexpf(1.f, e); }

hipError_t pack_nchw_f32(int dt, const float* src, int c, const TensorRef& dst, unsigned* flag, hipStream_t s) {
    if (dst.C % dtype_group(dt) || dst.Coff % dtype_group(dt)) return hipErrorInvalidValue;
    CV_DISPATCH(pack_nchw_f32_kernel, (size_t)dst.N * dst.H * dst.W * (dst.C / dtype_group(dt)), src, c, dst, pow2f(-dst.exp), flag);
}
hipError_t pack_hwc3_u8(int dt, const uint8_t* src, const TensorRef& dst, hipStream_t s) {
    if (dst.C != 8 || dst.Coff % 8) return hipErrorInvalidValue;
    CV_DISPATCH(pack_hwc3_u8_kernel, (size_t)dst.N * dst.H * dst.W, src, dst, pow2f(-dst.exp));
}
hipError_t unpack_nchw_f32(int dt, const TensorRef& src, float* dst, hipStream_t s) {
    if (src.Coff % dtype_group(dt)) return hipErrorInvalidValue;
    CV_DISPATCH(unpack_nchw_f32_kernel, (size_t)src.N * src.C * src.H * src.W, src, dst, pow2f(src.exp));
}
hipError_t absmax(int dt, const TensorRef& src, unsigned* out, hipStream_t s) {
    if (src.C % dtype_group(dt) || src.Coff % dtype_group(dt)) return hipErrorInvalidValue;
    const size_t work = (size_t)src.N * src.H * src.W * (src.C / dtype_group(dt));
    const dim3 g_((unsigned)std::min<size_t>(grid_for(work), 4096)), b_(256);
    if (dt == kF16) hipLaunchKernelGGL(absmax_kernel<half_t>, g_, b_, 0, s, src, out);
    else if (dt == kSplit) hipLaunchKernelGGL(absmax_kernel<split_t>, g_, b_, 0, s, src, out);
    else hipLaunchKernelGGL(absmax_kernel<float>, g_, b_, 0, s, src, out);
    return hipGetLastError();
}
hipError_t tap_sums_f16(const TensorRef& src, int Ho, int Wo, int stride, int k, int slices, double* out, hipStream_t s) {
    if (slices < 1 || (k != 1 && k != 3) || (Ho - 1) * stride + k - (k - 1) / 2 > src.H + 1 || (Wo - 1) * stride + k - (k - 1) / 2 > src.W + 1)
        return hipErrorInvalidValue;
    const int per = (src.N + slices - 1) / slices;
    hipLaunchKernelGGL(tap_sums_f16_kernel, dim3((unsigned)(k * k), (unsigned)((src.C + 63) / 64), (unsigned)slices), dim3(64), 0, s, src, Ho, Wo,
                       stride, k, per, out);
    return hipGetLastError();
}
hipError_t maxpool2x2(int dt, const TensorRef& src, const TensorRef& dst, hipStream_t s) {
    CV_DISPATCH(maxpool2x2_kernel, (size_t)dst.N * dst.H * dst.W * (dst.C / dtype_group(dt)), src, dst);
}
hipError_t maxpool3x3s2(int dt, const TensorRef& src, const TensorRef& dst, hipStream_t s) {
    CV_DISPATCH(maxpool3x3s2_kernel, (size_t)dst.N * dst.H * dst.W * (dst.C / dtype_group(dt)), src, dst);
}
hipError_t upsample_bilinear2x(int dt, const TensorRef& src, const TensorRef& dst, unsigned* flag, unsigned layer_id,
                               hipStream_t s) {
    const dim3 g_((unsigned)((dst.W * (dst.C / dtype_group(dt)) + 255) / 256), (unsigned)dst.H, (unsigned)dst.N), b_(256);
    const float mul = pow2f(src.exp - dst.exp);
    const float sy = dst.H > 1 ? (float)(src.H - 1) / (float)(dst.H - 1) : 0.f;
    const float sx = dst.W > 1 ? (float)(src.W - 1) / (float)(dst.W - 1) : 0.f;
    if (dt == kF16) hipLaunchKernelGGL(upsample_bilinear2x_kernel<half_t>, g_, b_, 0, s, src, dst, mul, sy, sx, flag, layer_id);
    else if (dt == kSplit) {
        static const bool block2x2 = [] { const char* v = std::getenv("CV_UPSAMPLE_2X2"); return !(v && v[0] == '0'); }();
        if (block2x2 && dst.H == 2 * src.H && dst.W == 2 * src.W) {
            const dim3 g2((unsigned)(((src.W + 1) * (dst.C / 8) + 255) / 256), (unsigned)(src.H + 1), (unsigned)dst.N);
            hipLaunchKernelGGL(upsample_bilinear2x_split2x2_kernel, g2, b_, 0, s, src, dst, mul, sy, sx, flag, layer_id);
        } else hipLaunchKernelGGL(upsample_bilinear2x_kernel<split_t>, g_, b_, 0, s, src, dst, mul, sy, sx, flag, layer_id);
    }
    else hipLaunchKernelGGL(upsample_bilinear2x_kernel<float>, g_, b_, 0, s, src, dst, mul, sy, sx, flag, layer_id);
    return hipGetLastError();
}
hipError_t outc_1x1(int dt, const TensorRef& src, const float* w, const float* bias, float* logits,
                    uint8_t* mask, float threshold, unsigned* flag, unsigned layer_id, hipStream_t s) {
    const int lpp = src.C / dtype_group(dt);
    if (lpp < 1 || lpp > 64 || (lpp & (lpp - 1))) return hipErrorInvalidValue;
    CV_DISPATCH(outc_1x1_kernel, (size_t)src.N * src.H * src.W * lpp, src, w, bias, pow2f(src.exp), logits, mask, threshold,
                flag, layer_id);
}
hipError_t stem7x7(int dt, const void* x, bool x_is_u8, int n, const float* w, const float* scale,
                   const float* shift, const TensorRef& dst, hipStream_t s) {
    if (dst.C != 64 || dst.H != 32 || dst.W != 32 || dst.Coff != 0) return hipErrorInvalidValue;
    const dim3 g((unsigned)n), b(256);
#define CV_STEM(T)                                                                                                   \
    do {                                                                                                             \
        if (x_is_u8) hipLaunchKernelGGL((stem7x7_kernel<T, uint8_t>), g, b, 0, s, (const uint8_t*)x, w, scale, shift, dst); \
        else hipLaunchKernelGGL((stem7x7_kernel<T, float>), g, b, 0, s, (const float*)x, w, scale, shift, dst);      \
    } while (0)
    if (dt == kF16) CV_STEM(half_t); else if (dt == kSplit) CV_STEM(split_t); else CV_STEM(float);
#undef CV_STEM
    return hipGetLastError();
}
hipError_t stem_pool_mfma(int dt, const void* x, bool x_is_u8, int n, const void* wpk, const float* scale,
                          const float* shift, int in_exp, const TensorRef& dst, unsigned* flag, unsigned layer_id,
                          hipStream_t s) {
    if (dt == kF32 || dst.C != 64 || dst.Cs != 64 || dst.H != 16 || dst.W != 16 || dst.Coff != 0) return hipErrorInvalidValue;   // rows of the tensor are contiguous runs
    const dim3 g((unsigned)(n < 2048 ? n : 2048)), b(256);
    const half8* w = reinterpret_cast<const half8*>(wpk);
    const float im = pow2f(-in_exp);
    if (dt == kF16 && dst.base32) {                      // f16r engine: split-f16 arithmetic, f16 tensor + f32 twin out
        if (x_is_u8) hipLaunchKernelGGL((stem_pool_mfma_kernel<half_t, uint8_t, true>), g, b, 0, s, (const uint8_t*)x, n, w, scale, shift, im, dst, flag, layer_id);
        else hipLaunchKernelGGL((stem_pool_mfma_kernel<half_t, float, true>), g, b, 0, s, (const float*)x, n, w, scale, shift, im, dst, flag, layer_id);
    } else if (dt == kF16) {
        if (x_is_u8) hipLaunchKernelGGL((stem_pool_mfma_kernel<half_t, uint8_t>), g, b, 0, s, (const uint8_t*)x, n, w, scale, shift, im, dst, flag, layer_id);
        else hipLaunchKernelGGL((stem_pool_mfma_kernel<half_t, float>), g, b, 0, s, (const float*)x, n, w, scale, shift, im, dst, flag, layer_id);
    } else {
        if (x_is_u8) hipLaunchKernelGGL((stem_pool_mfma_kernel<split_t, uint8_t>), g, b, 0, s, (const uint8_t*)x, n, w, scale, shift, im, dst, flag, layer_id);
        else hipLaunchKernelGGL((stem_pool_mfma_kernel<split_t, float>), g, b, 0, s, (const float*)x, n, w, scale, shift, im, dst, flag, layer_id);
    }
    return hipGetLastError();
}
hipError_t inc0_mfma(int dt, const void* x, bool x_is_u8, int n, const void* wpk, const float* scale, const float* shift,
                     int in_exp, const TensorRef& dst, unsigned* flag, unsigned layer_id, hipStream_t s) {
    if (dt == kF32 || dst.C != 64 || dst.H != 256 || dst.W != 256 || dst.Coff != 0 || dst.Cs != 64) return hipErrorInvalidValue;
    const dim3 g((unsigned)(n * 32)), b(256);
    const half8* w = reinterpret_cast<const half8*>(wpk);
    const float im = pow2f(-in_exp);
    if (dt == kF16) {
        if (x_is_u8) hipLaunchKernelGGL((inc0_mfma_kernel<half_t, uint8_t>), g, b, 0, s, (const uint8_t*)x, w, scale, shift, im, dst, flag, layer_id);
        else hipLaunchKernelGGL((inc0_mfma_kernel<half_t, float>), g, b, 0, s, (const float*)x, w, scale, shift, im, dst, flag, layer_id);
    } else {
        if (x_is_u8) hipLaunchKernelGGL((inc0_mfma_kernel<split_t, uint8_t>), g, b, 0, s, (const uint8_t*)x, w, scale, shift, im, dst, flag, layer_id);
        else hipLaunchKernelGGL((inc0_mfma_kernel<split_t, float>), g, b, 0, s, (const float*)x, w, scale, shift, im, dst, flag, layer_id);
    }
    return hipGetLastError();
}
hipError_t head_avgpool_fc(int dt, const TensorRef& src, const float* w, const float* b, float* out,
                           int softmax, unsigned* flag, unsigned layer_id, hipStream_t s) {
    if (src.C != 512 || src.Coff != 0 || src.Cs != 512) return hipErrorInvalidValue;      // the kernel keeps lane l on channels 8l..8l+7
    const int blocks = (src.N + 3) / 4;
    const dim3 g((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)), blk(256);
    const float mul = pow2f(src.exp);
    if (dt == kF16) hipLaunchKernelGGL(head_kernel<half_t>, g, blk, 0, s, src, w, b, mul, out, softmax, flag, layer_id);
    else if (dt == kSplit) hipLaunchKernelGGL(head_kernel<split_t>, g, blk, 0, s, src, w, b, mul, out, softmax, flag, layer_id);
    else hipLaunchKernelGGL(head_kernel<float>, g, blk, 0, s, src, w, b, mul, out, softmax, flag, layer_id);
    return hipGetLastError();
}
hipError_t softmax13(const float* logits, int n, float* probs, hipStream_t s) {
    hipLaunchKernelGGL(softmax13_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, s, logits, n, probs);
    return hipGetLastError();
}
hipError_t mfma_probe_f16(const half_t* a, const half_t* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_f16_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}
hipError_t mfma_probe_f32(const float* a, const float* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_f32_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}

// ---- f16r engine: ResNet stage-boundary shortcut (conv 1x1 / stride 2 + BN) from f32 twin to f32 twin ---------------------------
// reference: timm BasicBlock.downsample = [Conv2d(C, 2C, 1, stride 2, bias=False), BatchNorm2d] (notebooks/model-summary.ipynb).
// The shortcut IS the residual trunk at a stage boundary, so it runs at f32 grade: both operands are split into f16 hi + lo (the
// pixel values on the fly from the f32 twin, the weights at load time, rows normalised) and every k-step is three f16 MFMAs
// hi.hi + lo.hi + hi.lo with f32 accumulation -- the arithmetic of the f16x3 engine.  Round 4 ran these three layers on the f32-input
// MFMA through the generic implicit-GEMM kernel: 0.28 + 0.20 + 0.16 ms per 16384 squares for 1.1 % of the network's MACs (K = Cin is
// two to eight stages: all prologue); here a wave owns 16 output pixels x 128 channels, reads each pixel's Cin floats once and
// streams the packed weight fragments from L2; HBM-bound (Cin floats in, 2 Cin floats out per output pixel).
template <int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void shortcut1x1s2_kernel(const float* __restrict__ x, int n, int H, int W, const half8* __restrict__ wpk,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ y, unsigned* flag, unsigned layer_id) {
    constexpr int COUT = 2 * CIN, KS = CIN / 32, CG = COUT / 128;
    const int Ho = H / 2, Wo = W / 2, M = n * Ho * Wo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, l15 = lane & 15;
    const int task = blockIdx.x * 4 + wave;                      // (group of 16 output pixels, group of 128 output channels)
    const int cg = task % CG, pg = task / CG;
    if (pg * 16 >= M) return;
    int m = pg * 16 + l15;
    const bool live = m < M;
    if (!live) m = M - 1;
    const int ox = m % Wo, oy = (m / Wo) % Ho, img = m / (Wo * Ho);
    const float* const xp = x + ((size_t)(img * (H + 2) + 2 * oy + 1) * (W + 2) + 2 * ox + 1) * CIN + q * 8;
    f4 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = f4{0.f, 0.f, 0.f, 0.f};
    // Two k-steps at a time, every load of the batch issued before its first MFMA (4 pixel + 32 weight-fragment loads = 144
    // registers in flight): left to itself the compiler interleaves load / wait / MFMA one fragment at a time and a wave spends its
    // life in ~32 dependent L2 round trips (measured: 0.38 ms per 16384 squares for layer2.0, slower than the generic kernel).
#pragma unroll
    for (int kb = 0; kb < KS; kb += 2) {
        f4 xv[2][2];
        half8 wa[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xv[u][0] = *reinterpret_cast<const f4*>(xp + (kb + u) * 32);
            xv[u][1] = *reinterpret_cast<const f4*>(xp + (kb + u) * 32 + 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const half8* const wp = wpk + ((size_t)(cg * KS + kb + u) * 16) * 64 + lane;      // [cg][ks][fragment 8][hi | lo][lane]
#pragma unroll
            for (int i = 0; i < 16; ++i) wa[u][i] = wp[i * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // B fragment: k = 8 q + j of this k-step = eight consecutive input channels of pixel l15, split into f16 hi + lo.  Plain C on
            // purpose: the inline-asm pair conversion (cv_kernels.h: split_pair) next to independent MFMAs gave run-to-run different
            // results here -- the compiler cannot see the hazard between an in-flight MFMA's source registers and an asm's output
            half8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = j < 4 ? xv[u][0][j] : xv[u][1][j - 4];
                bh[j] = (half_t)v;
                bl[j] = (half_t)(v - (float)bh[j]);
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][2 * f], bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][2 * f + 1], bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[u][2 * f], bl, acc[f], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // lane (q, l15): channels 128 cg + 32 q + 4 f + r of pixel l15 (the weight rows were packed in that order): 128 contiguous bytes
    const int c0 = cg * 128 + q * 32;
    float* const yp = y + ((size_t)(img * (Ho + 2) + oy + 1) * (Wo + 2) + ox + 1) * COUT + c0;
    float bad = 0.f;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const f4 sc = *reinterpret_cast<const f4*>(scale + c0 + f * 4), sh = *reinterpret_cast<const f4*>(shift + c0 + f * 4);
        f4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] = acc[f][r] * sc[r] + sh[r]; bad = __builtin_fmaf(o[r], 0.f, bad); }
        if (live) *reinterpret_cast<f4*>(yp + f * 4) = o;
    }
    if (live && bad != bad && flag) atomicMin(flag, layer_id);
}

// Second form (round 5, late): the weights of the workgroup's 128-channel group live in LDS (KS x 16 KB, staged once), eight PERSISTENT
// waves walk the pixel groups and read their weight fragments from LDS -- in the first form every wave streamed those 32-128 KB from L2
// for 16 pixels of work (2.1 GB of L2 traffic per launch against 0.2-0.8 GB of HBM bytes: that, not the stride-2 read, was what the
// launch waited for).  The next pixel group's floats are fetched while the current one is multiplied (CIN <= 128: 16-32 registers).
// Same arithmetic, same operation order per output as the first form: same bits.
// SPLIT (round 6: the headline engine's shortcuts): x and y are split-f16 tensors instead of f32 twins.  A channel group is 32 bytes
// either way, so every address is the same; the two 16-byte chunks a lane fetches per k-step ARE its hi and lo operand (no conversion),
// and the staged epilogue writes each group's hi and lo chunk from the lane pair that shares the group -- still one contiguous KB per
// store instruction.  Products and their order (w_hi x_hi, w_lo x_hi, w_hi x_lo per 32-channel k-step) are those of conv_igemm_kernel
// on split_t: bit-identical to the generic launch it replaces.
// CONVT (round 6, SPLIT only): the same GEMM shape -- K = Cin, 2 Cin rows -- is a k2 / s2 transposed convolution whose rows are (dy, dx, co),
// 4 x Cin/2 of them: UNet up3.up (256 -> 128) and up4.up (128 -> 64).  Differences to the shortcut: the input pixels are dense (stride
// 1: 16 consecutive pixels are 8-16 KB of contiguous memory), and a staged row group leaves as a PIXEL SHUFFLE -- rows of class (dy, dx)
// go to output pixel (2 y + dy, 2 x + dx), channel slice [yCoff, yCoff + cout) of a buffer with yCs channels per pixel (the concatenated
// skip | up tensor).  The generic 256 x 256 tile moves these two layers' bytes at 3.2 / 4.0 TB/s: a 4- or 8-stage K loop, then 256 KB of
// stores per tile with nothing else resident on the CU; here the weights stay in LDS and eight persistent waves alternate loads, MFMAs
// and full-line stores.  Products and their order are the generic kernel's: bit-identical.
template <int CIN, int NW, bool PREFETCH, bool STAGE, bool SPLIT = false, bool CONVT = false>
__global__ __launch_bounds__(64 * NW) void shortcut1x1s2_lds_kernel(
    const float* __restrict__ x, int n, int H, int W, const half8* __restrict__ wpk, const float* __restrict__ scale,
    const float* __restrict__ shift, float* __restrict__ y, unsigned* flag, unsigned layer_id, int yCs = 0, int yCoff = 0, int cout = 0) {
    static_assert(!CONVT || (SPLIT && STAGE), "the transposed-convolution form exists for split-f16 tensors, staged stores");
    constexpr int COUT = 2 * CIN, KS = CIN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem_sc[];
    half8* const wl = reinterpret_cast<half8*>(smem_sc);
    const int cg = blockIdx.y;
    for (int i = threadIdx.x; i < KS * 16 * 64; i += 64 * NW) wl[i] = wpk[(size_t)cg * KS * 16 * 64 + i];
    // the group's BN scale / shift next to the weights: read from LDS in the epilogue.  As global loads they sat BEHIND the previous
    // iteration's stores in the wave's in-order memory counter, so every iteration waited for its predecessor's stores to retire
    // (7.7 us per iteration for 0.4 us of MFMAs)
    float* const sl = reinterpret_cast<float*>(smem_sc + (size_t)KS * 16 * 1024);
    if (threadIdx.x < 128) { sl[threadIdx.x] = scale[cg * 128 + threadIdx.x]; sl[128 + threadIdx.x] = shift[cg * 128 + threadIdx.x]; }
    __syncthreads();
    const int Ho = CONVT ? H : H / 2, Wo = CONVT ? W : W / 2, M = n * Ho * Wo, ngroups = (M + 15) / 16;   // GEMM pixels = output pixels of the shortcut | INPUT pixels of the transposed convolution
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, l15 = lane & 15;
    const int c0 = cg * 128 + q * 32;
    auto src_of = [&](int pg) {
        int m = pg * 16 + l15;
        m = m < M ? m : M - 1;
        const int ox = m % Wo, oy = (m / Wo) % Ho, img = m / (Wo * Ho);
        if constexpr (CONVT) return x + ((size_t)(img * (H + 2) + oy + 1) * (W + 2) + ox + 1) * CIN + q * 8;
        else return x + ((size_t)(img * (H + 2) + 2 * oy + 1) * (W + 2) + 2 * ox + 1) * CIN + q * 8;
    };
    auto fetch = [&](const float* xp, f4 (&xv)[KS][2]) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            xv[k][0] = *reinterpret_cast<const f4*>(xp + k * 32);
            xv[k][1] = *reinterpret_cast<const f4*>(xp + k * 32 + 4);
        }
    };
    float bad = 0.f;
    const int stride = gridDim.x * NW;
    int pg = blockIdx.x * NW + wave;
    f4 xn[KS][2];
    if (pg < ngroups) fetch(src_of(pg), xn);
    for (; pg < ngroups; pg += stride) {
        f4 xv[KS][2];
#pragma unroll
        for (int k = 0; k < KS; ++k) { xv[k][0] = xn[k][0]; xv[k][1] = xn[k][1]; }
        if (PREFETCH) { if (pg + stride < ngroups) fetch(src_of(pg + stride), xn); }
        f4 acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            half8 bh, bl;                                    // plain C conversion: see the first form
            if constexpr (SPLIT) {                           // group k*4 + q: even groups are stored [hi, lo], odd ones [lo, hi]
                const half8 a = __builtin_bit_cast(half8, xv[k][0]), b = __builtin_bit_cast(half8, xv[k][1]);
                bh = (q & 1) ? b : a;
                bl = (q & 1) ? a : b;
            } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = j < 4 ? xv[k][0][j] : xv[k][1][j - 4];
                bh[j] = (half_t)v;
                bl[j] = (half_t)(v - (float)bh[j]);
            }
            }
            // weight fragments four channel fragments at a time (32 registers live; left alone the compiler hoists every LDS read of
            // every k-step to the top of the iteration and spills 200-400 registers)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                half8 wa[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) wa[i] = wl[(k * 16 + h * 8 + i) * 64 + lane];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    acc[h * 4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2 * f], bh, acc[h * 4 + f], 0, 0, 0);
                    acc[h * 4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2 * f + 1], bh, acc[h * 4 + f], 0, 0, 0);
                    acc[h * 4 + f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2 * f], bl, acc[h * 4 + f], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int m = pg * 16 + l15;
        const bool live = m < M;
        const int mm = live ? m : M - 1;
        const int ox = mm % Wo, oy = (mm / Wo) % Ho, img = mm / (Wo * Ho);
        // shortcut: the output pixel; transposed convolution: output pixel (2 oy, 2 ox) of the 2H x 2W plane, class (dy, dx) adds dy rows + dx
        const size_t opix = CONVT ? (size_t)(img * (2 * Ho + 2) + 2 * oy + 1) * (2 * Wo + 2) + 2 * ox + 1
                                  : (size_t)(img * (Ho + 2) + oy + 1) * (Wo + 2) + ox + 1;
        if constexpr (STAGE) {
            // the accumulators leave each lane with 32 channels of ONE pixel: eight store instructions of 64 scattered 16-byte pieces.  Park
            // the wave's 16 x 128 tile in its own LDS corner (rows padded to 528 B) and read it back two whole pixel rows per instruction:
            // every store instruction then writes 1 KB in full lines.  Wave-private: DS operations of a wave execute in order.
            // SR pixel rows per pass (16 = the whole tile; Cin = 256 keeps 128 KB of weights in LDS and stages 4 rows at a time).
            constexpr int SROW = 528, SR = CIN <= 128 ? 16 : 4;
            char* const stg = smem_sc + (size_t)KS * 16 * 1024 + 1024 + (size_t)wave * (SR * SROW + 64);
            unsigned* const pixw = reinterpret_cast<unsigned*>(stg + SR * SROW);
            f4 o[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                const f4 sc = *reinterpret_cast<const f4*>(sl + q * 32 + f * 4), sh = *reinterpret_cast<const f4*>(sl + 128 + q * 32 + f * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[f][r] = acc[f][r] * sc[r] + sh[r]; bad = __builtin_fmaf(o[f][r], 0.f, bad); }
            }
#pragma unroll
            for (int ps = 0; ps < 16 / SR; ++ps) {
                if (SR == 16 || (l15 / SR) == ps) {
#pragma unroll
                    for (int f = 0; f < 8; ++f) *reinterpret_cast<f4*>(stg + (l15 % SR) * SROW + (q * 32 + f * 4) * 4) = o[f];
                    if (q == 0) pixw[l15 % SR] = live ? (unsigned)opix : 0xffffffffu;
                }
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < SR / 2; ++j) {
                    const int px = 2 * j + (lane >> 5), part = lane & 31;
                    const unsigned op = pixw[px];
                    if constexpr (SPLIT) {
                        // lanes 2g and 2g + 1 share channel group g of the pixel: one writes the group's first 16-byte chunk, the other the second
                        const int gl = part >> 1, par = gl & 1;                  // (cg * 16 + gl) & 1: a 128-channel group starts on an even group
                        const f4 v0 = *reinterpret_cast<const f4*>(stg + px * SROW + gl * 32), v1 = *reinterpret_cast<const f4*>(stg + px * SROW + gl * 32 + 16);
                        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if ((part & 1) == 0) {                                   // range guard on the f16 hi halves, once per group
#pragma unroll
                            for (int i = 0; i < 8; ++i) bad = __builtin_fmaf((float)(half_t)v[i], 0.f, bad);
                        }
                        if constexpr (CONVT) {
                            const int row = cg * 128 + gl * 8, cls = row / cout, co = row - cls * cout, ch0 = yCoff + co;
                            if (op != 0xffffffffu)
                                OutVec<split_t, 8>::store_half(reinterpret_cast<split_t*>(y) + ((size_t)op + (size_t)((cls >> 1) * (2 * Wo + 2) + (cls & 1))) * yCs + ch0,
                                                               ch0, v, (part & 1) == ((ch0 >> 3) & 1));
                        } else {
                        if (op != 0xffffffffu)
                            OutVec<split_t, 8>::store_half(reinterpret_cast<split_t*>(y) + (size_t)op * COUT + cg * 128 + gl * 8, gl * 8, v, (part & 1) == par);
                        }
                    } else {
                    const f4 v = *reinterpret_cast<const f4*>(stg + px * SROW + part * 16);
                    if (op != 0xffffffffu) *reinterpret_cast<f4*>(y + (size_t)op * COUT + cg * 128 + part * 4) = v;
                    }
                }
                wave_lds_sync();
            }
        } else {
        float* const yp = y + opix * COUT + c0;
        float ov[32];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const f4 sc = *reinterpret_cast<const f4*>(sl + q * 32 + f * 4), sh = *reinterpret_cast<const f4*>(sl + 128 + q * 32 + f * 4);
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = acc[f][r] * sc[r] + sh[r]; bad = __builtin_fmaf(o[r], 0.f, bad); ov[f * 4 + r] = o[r]; }
            if (!SPLIT && live) *reinterpret_cast<f4*>(yp + f * 4) = o;
        }
        if (SPLIT && live) OutVec<split_t, 32>::store(reinterpret_cast<split_t*>(yp), c0, ov, bad);
        }
        if (!PREFETCH) { if (pg + stride < ngroups) fetch(src_of(pg + stride), xn); }
    }
    if (bad != bad && flag) atomicMin(flag, layer_id);
}

hipError_t shortcut1x1s2(const TensorRef& x32, const void* wpk, const float* scale, const float* shift, const TensorRef& y32,
                         unsigned* flag, unsigned layer_id, hipStream_t s, bool split) {
    if ((!split && (!x32.f32_only || !y32.f32_only)) || (split && (x32.f32_only || y32.f32_only)) || x32.Coff || y32.Coff || x32.Cs != x32.C || y32.Cs != y32.C || y32.C != 2 * x32.C ||
        x32.H != 2 * y32.H || x32.W != 2 * y32.W || x32.N != y32.N)
        return hipErrorInvalidValue;
    const int cin = x32.C, M = x32.N * y32.H * y32.W;
    const int tasks = ((M + 15) / 16) * (2 * cin / 128);
    const dim3 grid((unsigned)((tasks + 3) / 4)), block(256);
    const float* xb = reinterpret_cast<const float*>(x32.base);
    float* yb = reinterpret_cast<float*>(y32.base);
    const half8* w = reinterpret_cast<const half8*>(wpk);
    static const bool lds_form = [] { const char* v = std::getenv("CV_SHORTCUT_LDS"); return !(v && v[0] == '0'); }();
    if (lds_form) {
        static const int per_cu = [] { const char* v = std::getenv("CV_SHORTCUT_WGS"); return v && *v ? std::atoi(v) : 1; }();
        static const int nw_knob = [] { const char* v = std::getenv("CV_SHORTCUT_NW"); return v && *v ? std::atoi(v) : 8; }();
        static const int pf_knob = [] { const char* v = std::getenv("CV_SHORTCUT_PF"); return v && *v ? std::atoi(v) : 1; }();
        const int nw = (nw_knob == 4 && cin == 64) ? 4 : 8;
        const int cgs = 2 * cin / 128, groups = (M + 15) / 16;
        int wgx = (256 * per_cu + cgs - 1) / cgs;                       // persistent: about per_cu workgroups per CU over all channel groups
        if (wgx > (groups + nw - 1) / nw) wgx = (groups + nw - 1) / nw;
        if (wgx < 1) wgx = 1;
        const dim3 g2((unsigned)wgx, (unsigned)cgs), b2((unsigned)(64 * nw));
        static const int stage_knob = [] { const char* v = std::getenv("CV_SHORTCUT_STAGE"); return v && *v ? std::atoi(v) : 1; }();
        const bool stage = stage_knob != 0;
        const size_t lds = (size_t)(cin / 32) * 16 * 1024 + 1024 + (stage ? (size_t)nw * ((cin <= 128 ? 16 : 4) * 528 + 64) : 0);   // weights + scale / shift (+ staging)
#define CV_SC_LAUNCH1(CIN_, NW_, PF_, ST_, SP_)                                                                                           \
        do {                                                                                                                              \
            static bool set_ = false;                                                                                                     \
            if (!set_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shortcut1x1s2_lds_kernel<CIN_, NW_, PF_, ST_, SP_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_ = true; } \
            hipLaunchKernelGGL((shortcut1x1s2_lds_kernel<CIN_, NW_, PF_, ST_, SP_>), g2, b2, lds, s, xb, x32.N, x32.H, x32.W, w, scale, shift, yb, flag, layer_id); \
        } while (0)
#define CV_SC_LAUNCH(CIN_, NW_, PF_, ST_) do { if (split) CV_SC_LAUNCH1(CIN_, NW_, PF_, ST_, true); else CV_SC_LAUNCH1(CIN_, NW_, PF_, ST_, false); } while (0)
        if (cin == 64 && nw == 4) { if (stage) CV_SC_LAUNCH(64, 4, true, true); else CV_SC_LAUNCH(64, 4, true, false); }
        else if (cin == 64) { if (stage) CV_SC_LAUNCH(64, 8, true, true); else CV_SC_LAUNCH(64, 8, true, false); }
        else if (cin == 128) { if (stage) CV_SC_LAUNCH(128, 8, true, true); else CV_SC_LAUNCH(128, 8, true, false); }
        else if (cin == 256 && pf_knob) { if (stage) CV_SC_LAUNCH(256, 8, true, true); else CV_SC_LAUNCH(256, 8, true, false); }
        else if (cin == 256) { if (stage) CV_SC_LAUNCH(256, 8, false, true); else CV_SC_LAUNCH(256, 8, false, false); }
        else return hipErrorInvalidValue;
#undef CV_SC_LAUNCH
#undef CV_SC_LAUNCH1
        return hipGetLastError();
    }
    if (split) return hipErrorInvalidValue;               // the first form exists for the f32 twins only
    if (cin == 64) hipLaunchKernelGGL(shortcut1x1s2_kernel<64>, grid, block, 0, s, xb, x32.N, x32.H, x32.W, w, scale, shift, yb, flag, layer_id);
    else if (cin == 128) hipLaunchKernelGGL(shortcut1x1s2_kernel<128>, grid, block, 0, s, xb, x32.N, x32.H, x32.W, w, scale, shift, yb, flag, layer_id);
    else if (cin == 256) hipLaunchKernelGGL(shortcut1x1s2_kernel<256>, grid, block, 0, s, xb, x32.N, x32.H, x32.W, w, scale, shift, yb, flag, layer_id);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// k2 / s2 transposed convolution (+ bias) between split-f16 tensors on the LDS-resident-weights kernel above (UNet up3.up, up4.up)
hipError_t convt2x2_lds(const TensorRef& x, const void* wpk, const float* scale, const float* shift, const TensorRef& y, unsigned* flag,
                        unsigned layer_id, hipStream_t s) {
    const int cin = x.C, cout = y.C;
    if ((cin != 128 && cin != 256) || 2 * cout != cin || x.Coff || x.Cs != cin || x.f32_only || y.f32_only || y.H != 2 * x.H || y.W != 2 * x.W ||
        y.N != x.N || y.Coff % 8 || y.Cs % 8 || cout % 8)
        return hipErrorInvalidValue;
    const long long M = (long long)x.N * x.H * x.W;
    if (M <= 0 || (long long)y.N * (y.H + 2) * (y.W + 2) >= (1ll << 32)) return hipErrorInvalidValue;
    const int nw = 8, cgs = 2 * cin / 128;
    const int groups = (int)((M + 15) / 16);
    static const int per_cu = [] { const char* v = std::getenv("CV_CONVT_WGS"); return v && *v ? std::atoi(v) : 1; }();
    int wgx = (256 * per_cu + cgs - 1) / cgs;                           // persistent: about per_cu workgroups per CU over all row groups
    if (wgx > (groups + nw - 1) / nw) wgx = (groups + nw - 1) / nw;
    if (wgx < 1) wgx = 1;
    const dim3 grid((unsigned)wgx, (unsigned)cgs), block((unsigned)(64 * nw));
    const size_t lds = (size_t)(cin / 32) * 16 * 1024 + 1024 + (size_t)nw * ((cin <= 128 ? 16 : 4) * 528 + 64);
    const float* xb = reinterpret_cast<const float*>(x.base);
    float* yb = reinterpret_cast<float*>(y.base);
    const half8* w = reinterpret_cast<const half8*>(wpk);
#define CV_CT_LAUNCH(CIN_)                                                                                                                 \
    do {                                                                                                                                  \
        static bool set_ = false;                                                                                                         \
        if (!set_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shortcut1x1s2_lds_kernel<CIN_, 8, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set_ = true; } \
        hipLaunchKernelGGL((shortcut1x1s2_lds_kernel<CIN_, 8, true, true, true, true>), grid, block, lds, s, xb, x.N, x.H, x.W, w, scale, shift, yb, flag, layer_id, y.Cs, y.Coff, cout); \
    } while (0)
    if (cin == 128) CV_CT_LAUNCH(128); else CV_CT_LAUNCH(256);
#undef CV_CT_LAUNCH
    return hipGetLastError();
}

}  // namespace cv
