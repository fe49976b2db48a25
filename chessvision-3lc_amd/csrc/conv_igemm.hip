// conv_igemm.hip -- implicit-GEMM convolution on MFMA for gfx950 (CDNA4), hand-written.
//
// Serves every dense contraction on the ChessVision hot path:
//   UNet  DoubleConv 3x3 p1 (x18), ConvTranspose2d k2 s2 (x4, as a 1-tap GEMM + pixel-shuffle store)
//   ResNet-18 3x3 s1/s2 (x16), 1x1 s2 downsample (x3)
// with BatchNorm(eval) affine, residual add and ReLU fused into the epilogue, so no activation is ever
// re-read for normalisation (reference: conv2d -> batch_norm -> relu are three torch kernels).
//
// GEMM orientation:  D[ch][pix] = sum_k Wt[ch][k] * X[pix][k]      (k = tap-major, channel-minor)
//   MFMA A operand = weights (rows = output channels), B operand = activations (cols = pixels).
//   The 16x16 C/D fragment then gives each lane 4 consecutive rows of one pixel; the host packs weight
//   rows so that a lane's rows across its FC fragments are 4*FC *consecutive channels*, hence every
//   lane stores 32 B (f16) / 64 B (f32) contiguous NHWC bytes and 4 lanes cover a 128-B line.
//
// Staging:  one "stage" = 128 bytes of K per row (64 f16 / 32 f32).  Both operands are brought HBM/L2 ->
//   LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip) into an
//   NS-deep ring; one s_barrier per stage, counted s_waitcnt vmcnt(N) so NS-2 stages stay in flight
//   across the barrier.  Rows are 128 B = 8 x 16-B chunks, chunk c of row r stored at position c^(r&7)
//   (XOR swizzle applied on the *source* address for activations, pre-applied by the host for weights)
//   so the ds_read_b128 fragment reads are bank-conflict free.
//   The activation gather needs no bounds checks: tensors carry a zero border (cv_kernels.h).
#include "cv_kernels.h"
#include "conv_igemm.h"
#include "conv_device.h"

namespace cv {

// CT x PT = channel x pixel tile of the workgroup (NW waves as WGC x NW/WGC); every wave owns a 64-channel
// slab (FC = 4 fragments: the row permutation the host packs for) and PT*WGC/NW pixels.  NW = 8 puts two waves
// of one workgroup on every SIMD.  Two K loops (see the main-loop comment): a register-rotating software pipeline for
// long K on the f16-MFMA dtypes, a plain barrier -> prefetch -> reads -> MFMAs loop otherwise.
// SEP: every stage's 8 K chunks are one contiguous 128-byte line of a pixel (all layers with Cin a multiple of the
// line), so the gather offset is kbase[stage] + 16 * chunk: the per-stage base comes through the scalar cache and no
// LDS is spent on the offset table (lets two 80 KB workgroups share a CU).
// POS: GEMM rows ordered [output position][image]; the K loop walks only the stages whose tap reads a real pixel at the tile's
// position (ConvParams::ptab; the 3x3 layers on ResNet-18's 2x2 / 4x4 / 8x8 maps at throughput batch sizes).
template <typename T, int CT, int PT, int WGC, int NS, int NW, bool SEP, bool POS = false>
__global__ __launch_bounds__(64 * NW) void conv_igemm_kernel(const ConvParams p) {
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
#include "conv_igemm_body.h"
}

// TWO independent layers of the same tile configuration in ONE launch (round 5; single boards: a ResNet-18 stage's 1x1 / stride-2
// shortcut convolution beside the stage's first 3x3 convolution -- both read the block input, neither fills the chip, and a
// dependent launch costs 5-6 us whatever it does).  Workgroups [0, a_n) run layer `a`, [a_pad, a_pad + b_n) layer `b`; a_pad = a_n
// rounded up to 8 so that both halves keep the XCD-aware tile walk of the body (workgroup index mod 8 = XCD); the padding returns
// at once.  Each half is exactly the launch conv_igemm_kernel would have been: same tiles, same arithmetic, same bits.
template <typename TA, typename TB, int CT, int PT, int WGC, int NS, int NW, bool SEP>
__global__ __launch_bounds__(64 * NW) void conv_igemm_pair_kernel(const ConvParams pa, const ConvParams pb, const unsigned a_n,
                                                                   const unsigned a_pad, const unsigned b_n) {
    // two copies of the body, each on its own kernel argument (choosing `p` at run time -- a reference to pa or pb -- makes the compiler
    // park a copy of the 448-byte parameter block in scratch) and each in its own arithmetic: TA != TB is the fp16 classifier, whose
    // shortcut convolutions run on the f32-input MFMA (resnet.cpp)
    constexpr bool POS = false;
    if (blockIdx.x < a_pad) {
        if (blockIdx.x >= a_n) return;
        typedef TA T;
        const ConvParams& p = pa;
        const unsigned nwg = a_n, bid = blockIdx.x;
#include "conv_igemm_body.h"
    } else {
        typedef TB T;
        const ConvParams& p = pb;
        const unsigned nwg = b_n, bid = blockIdx.x - a_pad;
#include "conv_igemm_body.h"
    }
}

// ---- split-K second pass: sum the splits in order, then the same epilogue as above -----------------------------------
// One lane = 8 consecutive channels of one output pixel (one 16 / 32-byte store unit).  Deterministic: the summation order is the
// split index, whatever order the first pass's workgroups finished in.  POOL: one lane = the same 8 channels of a 2 x 2 pixel
// block, which also yields the block's max_pool2d(2) value (UNet encoder: the pooled copy next to the skip tensor, as the halo
// kernel's fused epilogue writes it) -- a split launch then needs no stand-alone pooling kernel behind it.
template <typename T, bool POOL>
__global__ __launch_bounds__(64) void conv_splitk_reduce_kernel(const ConvParams p) {
    constexpr int UN = 8;
    constexpr int NPX = POOL ? 4 : 1;
    const int upp = p.rows / UN;                          // units per pixel (rows is a multiple of 16)
    const long long idx = (long long)blockIdx.x * 64 + threadIdx.x;
    const long long nblk = POOL ? (long long)p.M / 4 : (long long)p.M;
    if (idx >= nblk * upp) return;
    const int blk = (int)(idx / upp), cu = (int)(idx - (long long)blk * upp);
    const int row = cu * UN;
    const int HoWo = p.Ho * p.Wo;
    int n, oy, ox;
    if constexpr (POOL) {
        const int qw = p.Wo >> 1, qhw = (p.Ho >> 1) * qw;
        n = blk / qhw;
        const int rem = blk - n * qhw;
        oy = 2 * (rem / qw);
        ox = 2 * (rem - (rem / qw) * qw);
    } else {
        n = blk / HoWo;
        const int rem = blk - n * HoWo;
        oy = rem / p.Wo;
        ox = rem - oy * p.Wo;
    }
    const f4 sa = *reinterpret_cast<const f4*>(p.scale + row), sb = *reinterpret_cast<const f4*>(p.scale + row + 4);
    const f4 ha = *reinterpret_cast<const f4*>(p.shift + row), hb = *reinterpret_cast<const f4*>(p.shift + row + 4);
    T* const ybase = reinterpret_cast<T*>(p.y);
    const T* const rbase = reinterpret_cast<const T*>(p.res);
    float bad = 0.f;
    float best[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) best[j] = -3.0e38f;
#pragma unroll
    for (int px = 0; px < NPX; ++px) {
        const int yy = oy + (px >> 1), xx = ox + (px & 1);
        const int pix = (n * p.Ho + yy) * p.Wo + xx;
        float w[UN];
        {
            // four running sums over the splits ks = 0, 1, 2, 3 (mod 4), combined as (s0 + s1) + (s2 + s3): a FIXED order, and four
            // independent load streams in flight per lane (one dependent chain of ksplit loads is latency-bound: 35 us at 36 splits)
            const float* src = p.partial + (size_t)pix * p.prow + row;
            const size_t sstride = (size_t)p.M * p.prow;
            f4 a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = f4{0.f, 0.f, 0.f, 0.f}; b[j] = f4{0.f, 0.f, 0.f, 0.f}; }
            int ks = 0;
            for (; ks + 4 <= p.ksplit; ks += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] += *reinterpret_cast<const f4*>(src + (ks + j) * sstride);
                    b[j] += *reinterpret_cast<const f4*>(src + (ks + j) * sstride + 4);
                }
            }
            for (int j = 0; ks < p.ksplit; ++ks, ++j) {
                a[j] += *reinterpret_cast<const f4*>(src + ks * sstride);
                b[j] += *reinterpret_cast<const f4*>(src + ks * sstride + 4);
            }
            const f4 sa_ = (a[0] + a[1]) + (a[2] + a[3]), sb_ = (b[0] + b[1]) + (b[2] + b[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = sa_[j] * sa[j] + ha[j]; w[4 + j] = sb_[j] * sb[j] + hb[j]; }
        }
        unsigned ob = p.shuffle ? (unsigned)((n * p.yHp + 2 * yy + 1) * p.yWp + 2 * xx + 1) : (unsigned)((n * p.yHp + yy + 1) * p.yWp + xx + 1);
        int co = row;
        if (p.shuffle) {                                     // rows are (dy, dx, co): k2 s2 transposed conv
            const int grp = row / p.Cout;
            co = row - grp * p.Cout;
            ob += (unsigned)((grp >> 1) * p.yWp + (grp & 1));
        }
        if (rbase) {
            bool done = false;
            if constexpr (__is_same(T, half_t)) {
                if (p.res_f32) {
                    f4 raw[2];
                    trunk32_fetch(reinterpret_cast<const float*>(p.res) + (size_t)ob * p.rCs + p.rCoff + co, raw);
                    trunk32_add_raw(raw, w, p.res_mul);
                    done = true;
                }
            }
            if (!done) OutVec<T, UN>::add(rbase + (size_t)ob * p.rCs + p.rCoff + co, p.rCoff + co, w, p.res_mul);
        }
        if (p.relu) {
#pragma unroll
            for (int j = 0; j < UN; ++j) w[j] = w[j] > 0.f ? w[j] : 0.f;
        }
        OutVec<T, UN>::store(ybase + (size_t)ob * p.yCs + p.yCoff + co, p.yCoff + co, w, bad);
        if constexpr (__is_same(T, half_t)) {
            if (p.y32) trunk32_store(reinterpret_cast<float*>(p.y32) + (size_t)ob * p.yCs + p.yCoff + co, w);
        }
        if constexpr (POOL) {
#pragma unroll
            for (int j = 0; j < UN; ++j) best[j] = best[j] > w[j] ? best[j] : w[j];
        }
    }
    if constexpr (POOL) {
        const unsigned qb = (unsigned)((n * p.pHp + (oy >> 1) + 1) * p.pWp + (ox >> 1) + 1);
        OutVec<T, UN>::store(reinterpret_cast<T*>(p.pool_y) + (size_t)qb * p.pCs + p.pCoff + row, p.pCoff + row, best, bad);
    }
    report_bad(p, bad);
}

// ---- host-side launch -------------------------------------------------------------------------------
template <typename T, int CT, int PT, int WGC, int NS, int NW, bool SEP>
static hipError_t launch_one(const ConvParams& p, hipStream_t stream) {
    const int stages = p.ksplit > 1 ? p.kper : p.nStages;    // the offset table in LDS covers one split only
    const size_t lds = (size_t)NS * (CT + PT) * 128 + (SEP ? 0 : (((size_t)stages * 8 * 4 + 15) & ~(size_t)15));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int nPt = (p.M + PT - 1) / PT;
    const int splits = p.ksplit > 1 ? p.ksplit : 1;
    auto kern = conv_igemm_kernel<T, CT, PT, WGC, NS, NW, SEP>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nPt * p.nCt * splits)), dim3(64 * NW), lds, stream, p);
    return hipGetLastError();
}

// position-major launch (ConvParams::ptab): grid = 8 XCDs x output positions x the XCD's share of the (image tile, channel tile) columns
template <typename T, int CT, int PT, int WGC, int NS, int NW>
static hipError_t launch_pos(const ConvParams& p, hipStream_t stream) {
    const size_t lds = (size_t)NS * (CT + PT) * 128;
    if (lds > 160 * 1024 || !p.kbase || !p.ptab || !p.pcount || !p.porder || p.ksplit > 1 || p.head_w || p.posN <= 0) return hipErrorInvalidValue;
    if (p.nPtPer != (p.posN + PT - 1) / PT || p.M != p.posN * p.Ho * p.Wo) return hipErrorInvalidValue;
    auto kern = conv_igemm_kernel<T, CT, PT, WGC, NS, NW, true, true>;
    const unsigned col_groups = ((unsigned)(p.nPtPer * p.nCt) + 7u) >> 3;          // columns (image tile, channel tile) per XCD
    hipLaunchKernelGGL(kern, dim3(8u * col_groups * (unsigned)(p.Ho * p.Wo)), dim3(64 * NW), lds, stream, p);
    return hipGetLastError();
}

// two layers in one launch (conv_igemm_pair_kernel): same tile configuration, both with table-free gather offsets (SEP)
template <typename TA, typename TB, int CT, int PT, int WGC, int NS, int NW>
static hipError_t launch_pair(const ConvParams& a, const ConvParams& b, hipStream_t stream) {
    const size_t lds = (size_t)NS * (CT + PT) * 128;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto blocks = [](const ConvParams& p) { return (unsigned)(((p.M + PT - 1) / PT) * p.nCt * (p.ksplit > 1 ? p.ksplit : 1)); };
    const unsigned a_n = blocks(a), b_n = blocks(b), a_pad = (a_n + 7u) & ~7u;
    auto kern = conv_igemm_pair_kernel<TA, TB, CT, PT, WGC, NS, NW, true>;
    hipLaunchKernelGGL(kern, dim3(a_pad + b_n), dim3(64 * NW), lds, stream, a, b, a_n, a_pad, b_n);
    return hipGetLastError();
}

// second pass of a split-K launch (either conv kernel): sums p.partial over the splits and runs the layer's epilogue
hipError_t conv_splitk_reduce_launch(int dt, const ConvParams& p, hipStream_t stream) {
    if (p.ksplit <= 1 || !p.partial) return hipErrorInvalidValue;
    const bool pool = p.pool_y != nullptr;                 // fused 2x2 max-pool: even output extent, no pixel shuffle (checked by the engine)
    if (pool && (p.shuffle || (p.Ho & 1) || (p.Wo & 1))) return hipErrorInvalidValue;
    const long long units = (long long)(pool ? p.M / 4 : p.M) * (p.rows / 8);
    const dim3 grid((unsigned)((units + 63) / 64)), block(64);
#define CV_REDUCE(T) \
    do { if (pool) hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, true>), grid, block, 0, stream, p); \
         else hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, false>), grid, block, 0, stream, p); } while (0)
    if (dt == kF16) CV_REDUCE(half_t); else if (dt == kSplit) CV_REDUCE(split_t); else CV_REDUCE(float);
#undef CV_REDUCE
    return hipGetLastError();
}

template <typename T, int CT, int PT, int WGC, int NS, int NW>
static hipError_t prepare_one() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, CT, PT, WGC, NS, NW, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, CT, PT, WGC, NS, NW, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_pair_kernel<T, T, CT, PT, WGC, NS, NW, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess || !__is_same(T, half_t)) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_pair_kernel<float, half_t, CT, PT, WGC, NS, NW, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <typename T, int CT, int PT, int WGC, int NS, int NW>
static hipError_t prepare_pos() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<T, CT, PT, WGC, NS, NW, true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// (the 4-wave 128x256 tile and the ring-3 64x256 tile of round 1 lost every sweep and are no longer instantiated)
#define CV_FOR_EACH_CFG(X, T)              \
    X(T, 64, 128, 1, 3, 4, kCfg64x128)     \
    X(T, 128, 128, 2, 3, 4, kCfg128x128)   \
    X(T, 64, 256, 1, 2, 4, kCfg64x256)     \
    X(T, 64, 128, 1, 2, 4, kCfg64x128)     \
    X(T, 128, 128, 2, 2, 4, kCfg128x128)   \
    X(T, 128, 256, 2, 3, 8, kCfg128x256w8) \
    X(T, 256, 256, 4, 2, 8, kCfg256x256w8)

// the tiles that take position-major launches: the two 8-wave tiles (what the deep ResNet-18 stages run at throughput sizes)
#define CV_FOR_EACH_POS_CFG(X, T)          \
    X(T, 128, 128, 2, 2, 4, kCfg128x128)   \
    X(T, 128, 128, 2, 3, 4, kCfg128x128)   \
    X(T, 128, 256, 2, 3, 8, kCfg128x256w8) \
    X(T, 256, 256, 4, 2, 8, kCfg256x256w8)

hipError_t conv_igemm_prepare() {
    hipError_t e;
#define X(T, CT, PT, WGC, NS, NW, ID) \
    if ((e = prepare_one<T, CT, PT, WGC, NS, NW>()) != hipSuccess) return e;
    CV_FOR_EACH_CFG(X, half_t)
    CV_FOR_EACH_CFG(X, float)
    CV_FOR_EACH_CFG(X, split_t)
#undef X
#define X(T, CT, PT, WGC, NS, NW, ID) \
    if ((e = prepare_pos<T, CT, PT, WGC, NS, NW>()) != hipSuccess) return e;
    CV_FOR_EACH_POS_CFG(X, half_t)
    CV_FOR_EACH_POS_CFG(X, float)
    CV_FOR_EACH_POS_CFG(X, split_t)
#undef X
    return hipSuccess;
}

bool conv_cfg_has_pos(int cfg) { return cfg == kCfg128x128 || cfg == kCfg128x256w8 || cfg == kCfg256x256w8; }

hipError_t conv_igemm_pos_launch(int cfg, int ns, int dt, const ConvParams& p, hipStream_t stream) {
#define X(T, CT, PT, WGC, NS, NW, ID) \
    if (cfg == ID && ns == NS) return launch_pos<T, CT, PT, WGC, NS, NW>(p, stream);
    if (dt == kF16) { CV_FOR_EACH_POS_CFG(X, half_t) } else if (dt == kSplit) { CV_FOR_EACH_POS_CFG(X, split_t) } else { CV_FOR_EACH_POS_CFG(X, float) }
#undef X
    return hipErrorInvalidValue;
}

hipError_t conv_igemm_launch(int cfg, int ns, int dt, const ConvParams& p, hipStream_t stream) {
#define X(T, CT, PT, WGC, NS, NW, ID)                                                        \
    if (cfg == ID && ns == NS)                                                                \
        return p.kbase ? launch_one<T, CT, PT, WGC, NS, NW, true>(p, stream) : launch_one<T, CT, PT, WGC, NS, NW, false>(p, stream);
    if (dt == kF16) { CV_FOR_EACH_CFG(X, half_t) } else if (dt == kSplit) { CV_FOR_EACH_CFG(X, split_t) } else { CV_FOR_EACH_CFG(X, float) }
#undef X
    return hipErrorInvalidValue;
}

// `a` and `b` in one launch; both must carry kbase (table-free offsets) and suit the same (cfg, ns) -- the engine checks.  Arithmetic:
// the same for both, or f32 (`a`) beside f16 (`b`).
hipError_t conv_igemm_pair_launch(int cfg, int ns, int dt_a, int dt_b, const ConvParams& a, const ConvParams& b, hipStream_t stream) {
    if (!a.kbase || !b.kbase) return hipErrorInvalidValue;
    const bool mixed = dt_a == kF32 && dt_b == kF16;
    if (dt_a != dt_b && !mixed) return hipErrorInvalidValue;
#define X(T, CT, PT, WGC, NS, NW, ID) \
    if (cfg == ID && ns == NS) return launch_pair<T, T, CT, PT, WGC, NS, NW>(a, b, stream);
#define XM(T, CT, PT, WGC, NS, NW, ID) \
    if (cfg == ID && ns == NS) return launch_pair<float, half_t, CT, PT, WGC, NS, NW>(a, b, stream);
    if (mixed) { CV_FOR_EACH_CFG(XM, half_t) }
    else if (dt_a == kF16) { CV_FOR_EACH_CFG(X, half_t) } else if (dt_a == kSplit) { CV_FOR_EACH_CFG(X, split_t) } else { CV_FOR_EACH_CFG(X, float) }
#undef X
#undef XM
    return hipErrorInvalidValue;
}

bool conv_cfg_has_ns(int cfg, int ns) {
    if (cfg == kCfg128x256w8) return ns == 3;
    if (cfg == kCfg256x256w8 || cfg == kCfg64x256) return ns == 2;
    if (cfg == kCfg128x256) return false;
    return ns == 2 || ns == 3;
}
int conv_cfg_ct(int cfg) {
    if (cfg == kCfg256x256w8) return 256;
    return (cfg == kCfg64x256 || cfg == kCfg64x128) ? 64 : 128;
}
int conv_cfg_pt(int cfg) {
    return (cfg == kCfg64x128 || cfg == kCfg128x128) ? 128 : 256;
}

}  // namespace cv
