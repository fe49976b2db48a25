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

namespace cv {

template <typename T> struct V16;
template <> struct V16<half_t> { static constexpr int N = 8; typedef half8 V; };
template <> struct V16<float>  { static constexpr int N = 4; typedef f4 V; };

__device__ __forceinline__ size_t pix_index(const TensorRef& t, int n, int y, int x) {
    return (size_t)(n * (t.H + 2) + y + 1) * (t.W + 2) + (x + 1);
}
template <typename T> __device__ __forceinline__ T* elem_ptr(const TensorRef& t, size_t pix, int c) {
    return reinterpret_cast<T*>(t.base) + pix * t.Cs + t.Coff + c;
}

static inline unsigned grid_for(size_t work, int block = 256) {
    size_t g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g);
}

// ---- packing ----------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_nchw_f32_kernel(const float* __restrict__ src, int c, TensorRef dst) {
    constexpr int VN = V16<T>::N;
    const int groups = dst.C / VN;
    const size_t total = (size_t)dst.N * dst.H * dst.W * groups;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % groups);
    size_t pix = idx / groups;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    typename V16<T>::V v;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
        const int ch = g * VN + j;
        const float f = ch < c ? src[((size_t)(n * c + ch) * dst.H + y) * dst.W + x] : 0.f;
        v[j] = (T)f;
    }
    *reinterpret_cast<typename V16<T>::V*>(elem_ptr<T>(dst, pix_index(dst, n, y, x), g * VN)) = v;
}

template <typename T>
__global__ void pack_hwc3_u8_kernel(const uint8_t* __restrict__ src, TensorRef dst) {
    const size_t total = (size_t)dst.N * dst.H * dst.W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t pix = idx;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    const uint8_t* s = src + idx * 3;
    float f[8] = {(float)s[0] / 255.f, (float)s[1] / 255.f, (float)s[2] / 255.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    T* d = elem_ptr<T>(dst, pix_index(dst, n, y, x), 0);
    constexpr int VN = V16<T>::N;
#pragma unroll
    for (int i = 0; i < 8; i += VN) {
        typename V16<T>::V v;
#pragma unroll
        for (int j = 0; j < VN; ++j) v[j] = (T)f[i + j];
        *reinterpret_cast<typename V16<T>::V*>(d + i) = v;
    }
}

template <typename T>
__global__ void unpack_nchw_f32_kernel(TensorRef src, float* __restrict__ dst) {
    const size_t total = (size_t)src.N * src.C * src.H * src.W;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t r = idx;
    const int x = (int)(r % src.W); r /= src.W;
    const int y = (int)(r % src.H); r /= src.H;
    const int c = (int)(r % src.C);
    const int n = (int)(r / src.C);
    dst[idx] = (float)*elem_ptr<T>(src, pix_index(src, n, y, x), c);
}

// ---- pooling / upsampling --------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2x2_kernel(TensorRef src, TensorRef dst) {
    constexpr int VN = V16<T>::N;
    typedef typename V16<T>::V V;
    const int groups = dst.C / VN;
    const size_t total = (size_t)dst.N * dst.H * dst.W * groups;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % groups);
    size_t pix = idx / groups;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    const size_t p00 = pix_index(src, n, 2 * y, 2 * x);
    const size_t rowp = (size_t)(src.W + 2);
    const V a = *reinterpret_cast<const V*>(elem_ptr<T>(src, p00, g * VN));
    const V b = *reinterpret_cast<const V*>(elem_ptr<T>(src, p00 + 1, g * VN));
    const V c = *reinterpret_cast<const V*>(elem_ptr<T>(src, p00 + rowp, g * VN));
    const V d = *reinterpret_cast<const V*>(elem_ptr<T>(src, p00 + rowp + 1, g * VN));
    const V m = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
    *reinterpret_cast<V*>(elem_ptr<T>(dst, pix_index(dst, n, y, x), g * VN)) = m;
}

template <typename T>
__global__ void maxpool3x3s2_kernel(TensorRef src, TensorRef dst) {
    constexpr int VN = V16<T>::N;
    typedef typename V16<T>::V V;
    const int groups = dst.C / VN;
    const size_t total = (size_t)dst.N * dst.H * dst.W * groups;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % groups);
    size_t pix = idx / groups;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    V m;
#pragma unroll
    for (int j = 0; j < VN; ++j) m[j] = (T)(-65504.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * y - 1 + ky;
        if (iy < 0 || iy >= src.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * x - 1 + kx;
            if (ix < 0 || ix >= src.W) continue;
            const V v = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, iy, ix), g * VN));
            m = __builtin_elementwise_max(m, v);
        }
    }
    *reinterpret_cast<V*>(elem_ptr<T>(dst, pix_index(dst, n, y, x), g * VN)) = m;
}

// torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1); weights (1-l, l) in f32
template <typename T>
__global__ void upsample_bilinear2x_kernel(TensorRef src, TensorRef dst) {
    constexpr int VN = V16<T>::N;
    typedef typename V16<T>::V V;
    const int groups = dst.C / VN;
    const size_t total = (size_t)dst.N * dst.H * dst.W * groups;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % groups);
    size_t pix = idx / groups;
    const int x = (int)(pix % dst.W); pix /= dst.W;
    const int y = (int)(pix % dst.H);
    const int n = (int)(pix / dst.H);
    const float sy = dst.H > 1 ? (float)(src.H - 1) / (float)(dst.H - 1) : 0.f;
    const float sx = dst.W > 1 ? (float)(src.W - 1) / (float)(dst.W - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < src.H - 1 ? 1 : 0), x1 = x0 + (x0 < src.W - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const V v00 = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, y0, x0), g * VN));
    const V v01 = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, y0, x1), g * VN));
    const V v10 = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, y1, x0), g * VN));
    const V v11 = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, y1, x1), g * VN));
    V o;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
        const float r = ly0 * (lx0 * (float)v00[j] + lx1 * (float)v01[j]) +
                        ly1 * (lx0 * (float)v10[j] + lx1 * (float)v11[j]);
        o[j] = (T)r;
    }
    *reinterpret_cast<V*>(elem_ptr<T>(dst, pix_index(dst, n, y, x), g * VN)) = o;
}

// ---- OutConv 1x1 (C -> 1) + bias, fused sigmoid/threshold mask ------------------------------------
// LPP = C*sizeof(T)/16 lanes share one pixel (each 16 B of its channels); xor-shuffle reduce inside the group.
template <typename T>
__global__ void outc_1x1_kernel(TensorRef src, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ logits, uint8_t* __restrict__ mask, float thr) {
    constexpr int VN = V16<T>::N;
    typedef typename V16<T>::V V;
    const int lpp = src.C / VN;                               // power of two, <= 64
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
    const V v = *reinterpret_cast<const V*>(elem_ptr<T>(src, pix_index(src, n, y, x), part * VN));
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < VN; ++j) acc += (float)v[j] * w[part * VN + j];
    for (int m = 1; m < lpp; m <<= 1) acc += __shfl_xor(acc, m);
    if (live && part == 0) {
        const float l = acc + bias[0];
        logits[pix] = l;
        if (mask) mask[pix] = (1.f / (1.f + __expf(-l))) > thr ? 255 : 0;
    }
}

// ---- ResNet stem: conv 7x7 s2 p3, 1 -> 64, + BN affine + ReLU -------------------------------------
// One workgroup per 64x64 square.  The zero-bordered plane sits in LDS; each lane owns one output pixel at
// a time, holds its 7x7 patch in VGPRs and runs the 64x49 filter bank from the scalar cache (wave-uniform
// weights -> s_load + v_fma with an SGPR operand), i.e. a pure VALU f32 kernel: K = 49 is too thin and the
// input too small for the MFMA path to pay.
template <typename T, typename XT>
__global__ __launch_bounds__(256) void stem7x7_kernel(const XT* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, TensorRef dst) {
    constexpr int IN = 64, P = 3, LD = IN + 2 * P;            // 70
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
        T* d = elem_ptr<T>(dst, pix_index(dst, n, oy, ox), 0);
        constexpr int VN = V16<T>::N;
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
            for (int i = 0; i < 8; i += VN) {
                typename V16<T>::V o;
#pragma unroll
                for (int j = 0; j < VN; ++j) o[j] = (T)v[i + j];
                *reinterpret_cast<typename V16<T>::V*>(d + cb + i) = o;
            }
        }
    }
}

// ---- head: adaptive_avg_pool2d(1) + Linear(C -> 13) (+ softmax) ------------------------------------
// One wave per square; lane owns C/64 consecutive channels; 13 wave-wide xor-butterfly reductions.
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(TensorRef src, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out,
                                                   int softmax) {
    constexpr int NC = 13;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= src.N) return;                                    // whole wave exits together
    const int cpl = src.C / 64;                                // channels per lane (8 for C = 512)
    float acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = 0.f;
    const float inv = 1.f / (float)(src.H * src.W);
    for (int cc = 0; cc < cpl; ++cc) {
        const int c = lane * cpl + cc;
        float s = 0.f;
        for (int y = 0; y < src.H; ++y)
            for (int x = 0; x < src.W; ++x) s += (float)*elem_ptr<T>(src, pix_index(src, n, y, x), c);
        s *= inv;
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] = __builtin_fmaf(s, w[j * src.C + c], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m);
        acc[j] += b[j];
    }
    if (softmax) {
        float mx = acc[0];
#pragma unroll
        for (int j = 1; j < NC; ++j) mx = fmaxf(mx, acc[j]);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) { acc[j] = __expf(acc[j] - mx); sum += acc[j]; }
        const float r = 1.f / sum;
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] *= r;
    }
    if (lane < NC) {
        float v = acc[0];
#pragma unroll
        for (int j = 1; j < NC; ++j) v = lane == j ? acc[j] : v;
        out[(size_t)n * NC + lane] = v;
    }
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
#define CV_LAUNCH(KERN, WORK, ...)                                                           \
    do {                                                                                     \
        hipLaunchKernelGGL(KERN, dim3(grid_for(WORK)), dim3(256), 0, s, __VA_ARGS__);        \
        return hipGetLastError();                                                            \
    } while (0)

hipError_t pack_nchw_f32(bool f16, const float* src, int c, const TensorRef& dst, hipStream_t s) {
    const size_t work = (size_t)dst.N * dst.H * dst.W * (dst.C / (f16 ? 8 : 4));
    if (f16) CV_LAUNCH(pack_nchw_f32_kernel<half_t>, work, src, c, dst);
    CV_LAUNCH(pack_nchw_f32_kernel<float>, work, src, c, dst);
}
hipError_t pack_hwc3_u8(bool f16, const uint8_t* src, const TensorRef& dst, hipStream_t s) {
    if (dst.C != 8) return hipErrorInvalidValue;
    const size_t work = (size_t)dst.N * dst.H * dst.W;
    if (f16) CV_LAUNCH(pack_hwc3_u8_kernel<half_t>, work, src, dst);
    CV_LAUNCH(pack_hwc3_u8_kernel<float>, work, src, dst);
}
hipError_t unpack_nchw_f32(bool f16, const TensorRef& src, float* dst, hipStream_t s) {
    const size_t work = (size_t)src.N * src.C * src.H * src.W;
    if (f16) CV_LAUNCH(unpack_nchw_f32_kernel<half_t>, work, src, dst);
    CV_LAUNCH(unpack_nchw_f32_kernel<float>, work, src, dst);
}
hipError_t maxpool2x2(bool f16, const TensorRef& src, const TensorRef& dst, hipStream_t s) {
    const size_t work = (size_t)dst.N * dst.H * dst.W * (dst.C / (f16 ? 8 : 4));
    if (f16) CV_LAUNCH(maxpool2x2_kernel<half_t>, work, src, dst);
    CV_LAUNCH(maxpool2x2_kernel<float>, work, src, dst);
}
hipError_t maxpool3x3s2(bool f16, const TensorRef& src, const TensorRef& dst, hipStream_t s) {
    const size_t work = (size_t)dst.N * dst.H * dst.W * (dst.C / (f16 ? 8 : 4));
    if (f16) CV_LAUNCH(maxpool3x3s2_kernel<half_t>, work, src, dst);
    CV_LAUNCH(maxpool3x3s2_kernel<float>, work, src, dst);
}
hipError_t upsample_bilinear2x(bool f16, const TensorRef& src, const TensorRef& dst, hipStream_t s) {
    const size_t work = (size_t)dst.N * dst.H * dst.W * (dst.C / (f16 ? 8 : 4));
    if (f16) CV_LAUNCH(upsample_bilinear2x_kernel<half_t>, work, src, dst);
    CV_LAUNCH(upsample_bilinear2x_kernel<float>, work, src, dst);
}
hipError_t outc_1x1(bool f16, const TensorRef& src, const float* w, const float* bias, float* logits,
                    uint8_t* mask, float threshold, hipStream_t s) {
    const int lpp = src.C / (f16 ? 8 : 4);
    if (lpp < 1 || lpp > 64 || (lpp & (lpp - 1))) return hipErrorInvalidValue;
    const size_t work = (size_t)src.N * src.H * src.W * lpp;
    if (f16) CV_LAUNCH(outc_1x1_kernel<half_t>, work, src, w, bias, logits, mask, threshold);
    CV_LAUNCH(outc_1x1_kernel<float>, work, src, w, bias, logits, mask, threshold);
}
hipError_t stem7x7(bool f16, const void* x, bool x_is_u8, int n, const float* w, const float* scale,
                   const float* shift, const TensorRef& dst, hipStream_t s) {
    if (dst.C != 64 || dst.H != 32 || dst.W != 32 || dst.Coff != 0) return hipErrorInvalidValue;
    const dim3 g((unsigned)n), b(256);
    if (f16) {
        if (x_is_u8) hipLaunchKernelGGL((stem7x7_kernel<half_t, uint8_t>), g, b, 0, s, (const uint8_t*)x, w, scale, shift, dst);
        else         hipLaunchKernelGGL((stem7x7_kernel<half_t, float>), g, b, 0, s, (const float*)x, w, scale, shift, dst);
    } else {
        if (x_is_u8) hipLaunchKernelGGL((stem7x7_kernel<float, uint8_t>), g, b, 0, s, (const uint8_t*)x, w, scale, shift, dst);
        else         hipLaunchKernelGGL((stem7x7_kernel<float, float>), g, b, 0, s, (const float*)x, w, scale, shift, dst);
    }
    return hipGetLastError();
}
hipError_t head_avgpool_fc(bool f16, const TensorRef& src, const float* w, const float* b, float* out,
                           int softmax, hipStream_t s) {
    if (src.C % 64) return hipErrorInvalidValue;
    const dim3 g((unsigned)((src.N + 3) / 4)), blk(256);
    if (f16) hipLaunchKernelGGL(head_kernel<half_t>, g, blk, 0, s, src, w, b, out, softmax);
    else     hipLaunchKernelGGL(head_kernel<float>, g, blk, 0, s, src, w, b, out, softmax);
    return hipGetLastError();
}
hipError_t softmax13(const float* logits, int n, float* probs, hipStream_t s) {
    CV_LAUNCH(softmax13_kernel, (size_t)n, logits, n, probs);
}
hipError_t mfma_probe_f16(const half_t* a, const half_t* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_f16_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}
hipError_t mfma_probe_f32(const float* a, const float* b, float* d, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_f32_kernel, dim3(1), dim3(64), 0, s, a, b, d);
    return hipGetLastError();
}

}  // namespace cv
