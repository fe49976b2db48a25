// conv_halo.hip -- 3x3 / stride-1 convolution with the activation operand kept as a haloed patch in LDS.
//
// Why a second conv kernel: in conv_igemm.hip every K stage DMAs its own activation tile, so a 3x3 layer moves the
// same input pixels L2 -> LDS nine times (once per tap).  The r01 ablation (profiles/r01_tuning.md, step 12) showed that
// byte stream, not the matrix pipe, is what the stage time follows.  Here a workgroup owns a TH x 16 pixel patch of one
// image and, per 128-byte input-channel block, DMAs the (TH+2) x 18 halo ONCE; the nine taps are nine shifted
// fragment reads of that LDS image.  Only the weight tile still streams per stage.  L2 -> LDS bytes per MFMA drop from
// 128 (128x256 tile) / 213 (64x256) to 54 / 66.
//
// Same GEMM orientation, weight packing ([ctTile][stage][row][swizzled chunk], K = [channel block][tap][channel]),
// fragment maps, epilogue (BN affine, residual, ReLU, LDS-staged coalesced stores, optional fused 1x1 head) and
// precisions as conv_igemm.hip.  Requirements checked by the host: k = 3, stride 1, one channel block = one 128-byte
// line (separable offsets), output width a multiple of 16, output height a multiple of TH.
//
// Synchronisation: one s_barrier per stage (= channel block x tap).  The weight ring is NSW = 3 deep; the halo is
// double buffered and the next block's halo is issued at tap 0.  DMA completes in issue order, so the counted
// s_waitcnt before stage s = (cb, tap) may leave in flight exactly what was issued after W(s): W(s+1), plus the next
// halo when tap is 1 or 2.
#include "cv_kernels.h"
#include "conv_igemm.h"
#include "conv_device.h"

#include <cstdlib>

namespace cv {

// TPS = taps per stage (1, or 3 = one filter row: fewer, fatter stages for the 64-row tile); NSW = weight ring depth.
template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW>
__global__ __launch_bounds__(64 * NW) void conv3x3_halo_kernel(const ConvParams p) {
    static_assert((TPS == 1 || TPS == 3) && (NSW == 2 || NSW == 3), "stage shape");
    constexpr int SPC = 9 / TPS;                        // stages per channel block
    constexpr int WGP = NW / WGC;                       // wave groups along the patch rows
    static_assert(CT / WGC == 64, "every wave owns a 64-channel slab");
    constexpr int FC = 4;
    constexpr int FP = TH / WGP;                        // patch rows (= 16-pixel fragments) per wave
    static_assert(FP >= 1 && FP <= 4 && TH % WGP == 0, "wave tile");
    constexpr int WTAP = CT * 128;                      // weight bytes of one tap
    constexpr int WSTAGE = TPS * WTAP;
    constexpr int HR = 18 * (TH + 2);                   // halo rows (one pixel = one 128-byte LDS row)
    constexpr int H = ((HR + 7) / 8 + NW - 1) / NW;     // halo DMA wave-instructions per wave (8 rows each)
    constexpr int HBYTES = H * NW * 1024;
    constexpr int LW = TPS * CT / (8 * NW);
    static_assert(LW >= 1 && LW + H <= 63, "vmcnt range");
    constexpr bool kSplit16 = sizeof(T) == 4 && !__is_same(T, float);
    typedef typename FragT<T>::V V;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem + NSW * WSTAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned nwg = gridDim.x, bid = blockIdx.x;   // XCD-aware remap, as in conv_igemm.hip
    const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int ctTile = lid % p.nCt;
    int pt = lid / p.nCt;
    const int tilesX = p.Wo / 16, tilesY = p.Ho / TH;
    const int tx = pt % tilesX; pt /= tilesX;
    const int ty = pt % tilesY;
    const int n = pt / tilesY;
    const int nCb = p.nStages / 9, nS = nCb * SPC;       // p.nStages counts (channel block, tap) pairs

    // DMA sources of this lane's halo rows: row r of the halo <-> padded input pixel (ty*TH + r/18, tx*16 + r%18)
    unsigned hoff[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        const int rr = r < HR ? r : HR - 1;             // rows of the padded tail re-read the last real one
        const int hy = rr / 18, hx = rr - hy * 18;
        hoff[i] = (unsigned)((n * p.xHp + ty * TH + hy) * p.xWp + tx * 16 + hx) * (unsigned)p.xCs * (unsigned)sizeof(T) +
                  (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
    }
    const char* const xsrc = p.x + p.xCoffBytes;
    const char* const wsrc = p.w + (size_t)ctTile * p.nStages * WTAP + wave * 1024 + lane * 16;

    auto issue_w = [&](int s, int slot) {
        char* sW = smem + slot * WSTAGE;
#pragma unroll
        for (int i = 0; i < LW; ++i) glds16(wsrc + (size_t)s * WSTAGE + i * (NW * 1024), sW + (i * NW + wave) * 1024);   // stage s = TPS consecutive taps
    };
    auto issue_halo = [&](int cb, int hb) {
        char* sH = halo + hb * HBYTES;
#pragma unroll
        for (int i = 0; i < H; ++i) glds16(xsrc + hoff[i] + cb * 128, sH + (i * NW + wave) * 1024);
    };

    const int wci = wave / WGP, wpi = wave % WGP;
    const int q = lane >> 4, l15 = lane & 15;
    const int rowW = (wci * 64 + l15) * 128;
    const int wrow0 = wpi * FP;                          // first patch row of this wave
    int hr0[FP];                                         // halo row of (patch row, pixel l15) for tap (0,0)
#pragma unroll
    for (int g = 0; g < FP; ++g) hr0[g] = (wrow0 + g) * 18 + l15;
    const int l7 = lane & 7;

    f4 acc[FC][FP];
#pragma unroll
    for (int f = 0; f < FC; ++f)
#pragma unroll
        for (int g = 0; g < FP; ++g) acc[f][g] = f4{0.f, 0.f, 0.f, 0.f};

    auto compute_tap = [&](auto tap_tag, const char* sW, const char* sH) {
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr int TOFF = (TAP / 3) * 18 + (TAP % 3);
        // The halo addresses of all 9 taps x FP rows x {hi, lo} are loop invariant; left alone, LICM keeps ~70 of them
        // live across the channel-block loop and the 128-row split-f16 tile spills.  Re-deriving them per tap costs a
        // handful of VALU ops, so make the row base opaque here.
        int hrb[FP];
#pragma unroll
        for (int g = 0; g < FP; ++g) { hrb[g] = hr0[g]; asm volatile("" : "+v"(hrb[g])); }
        if constexpr (kSplit16) {
            const int chi = 2 * q + (q & 1), clo = 2 * q + 1 - (q & 1);
            V ah[FC], al[FC], bh[FP], bl[FP];
#pragma unroll
            for (int f = 0; f < FC; ++f) ah[f] = *reinterpret_cast<const V*>(sW + rowW + f * 2048 + ((chi ^ l7) << 4));
#pragma unroll
            for (int g = 0; g < FP; ++g) {
                const int hr = hrb[g] + TOFF;
                bh[g] = *reinterpret_cast<const V*>(sH + hr * 128 + ((chi ^ (hr & 7)) << 4));
            }
#pragma unroll
            for (int f = 0; f < FC; ++f) al[f] = *reinterpret_cast<const V*>(sW + rowW + f * 2048 + ((clo ^ l7) << 4));
#pragma unroll
            for (int g = 0; g < FP; ++g) {
                const int hr = hrb[g] + TOFF;
                bl[g] = *reinterpret_cast<const V*>(sH + hr * 128 + ((clo ^ (hr & 7)) << 4));
            }
            if (kSetPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int g = 0; g < FP; ++g) mma16(acc[f][g], ah[f], bh[g]);
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int g = 0; g < FP; ++g) {
                    mma16(acc[f][g], al[f], bh[g]);
                    mma16(acc[f][g], ah[f], bl[g]);
                }
            if (kSetPrio) __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int c = sub * 4 + q;
                V a[FC], b[FP];
#pragma unroll
                for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const V*>(sW + rowW + f * 2048 + ((c ^ l7) << 4));
#pragma unroll
                for (int g = 0; g < FP; ++g) {
                    const int hr = hrb[g] + TOFF;
                    b[g] = *reinterpret_cast<const V*>(sH + hr * 128 + ((c ^ (hr & 7)) << 4));
                }
#pragma unroll
                for (int f = 0; f < FC; ++f)
#pragma unroll
                    for (int g = 0; g < FP; ++g) mma16(acc[f][g], a[f], b[g]);
            }
        }
    };
    // stage J of a channel block covers taps J*TPS .. J*TPS + TPS - 1
    auto compute = [&](auto j_tag, int wslot, int hb) {
        constexpr int J = decltype(j_tag)::value;
        const char* sW = smem + wslot * WSTAGE;
        const char* sH = halo + hb * HBYTES;
        compute_tap(std::integral_constant<int, J * TPS>{}, sW, sH);
        if constexpr (TPS == 3) {
            compute_tap(std::integral_constant<int, J * TPS + 1>{}, sW + WTAP, sH);
            compute_tap(std::integral_constant<int, J * TPS + 2>{}, sW + 2 * WTAP, sH);
        }
    };

    // ---- main loop: channel blocks x stages ----------------------------------------------------------------
    issue_halo(0, 0);
    issue_w(0, 0);
    if (NSW == 3 && nS > 1) issue_w(1, 1);
    int slotC = 0, slotI = NSW - 1, s = 0;
    for (int cb = 0; cb < nCb; ++cb) {
        const int hb = cb & 1;
        const bool more_cb = cb + 1 < nCb;
        auto stage = [&](auto j_tag) {
            constexpr int J = decltype(j_tag)::value;
            // in flight after W(s): W(s+1) (ring 3 only) and the next halo if it was issued one (or two) stages ago
            const bool w_next = NSW == 3 && s + 1 < nS;
            const bool halo_young = (J == 1 || (NSW == 3 && J == 2)) && more_cb;
            if (w_next && halo_young) wait_vm_barrier<LW + H>();
            else if (w_next) wait_vm_barrier<LW>();
            else if (halo_young) wait_vm_barrier<H>();
            else wait_vm_barrier<0>();
            if (s + NSW - 1 < nS) issue_w(s + NSW - 1, slotI);
            if (J == 0 && more_cb) issue_halo(cb + 1, hb ^ 1);
            compute(j_tag, slotC, hb);
            slotC = slotC == NSW - 1 ? 0 : slotC + 1;
            slotI = slotI == NSW - 1 ? 0 : slotI + 1;
            ++s;
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{});
        stage(std::integral_constant<int, 2>{});
        if constexpr (SPC == 9) {
            stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{});
            stage(std::integral_constant<int, 5>{}); stage(std::integral_constant<int, 6>{});
            stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
        }
    }

    // ---- epilogue (see conv_igemm.hip for the rationale of the staged store) ------------------------------
    constexpr int NV = 4 * FC;
    const int row0 = ctTile * CT + wci * 64 + q * NV;
    float sc[NV], sh[NV];
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const f4 a = *reinterpret_cast<const f4*>(p.scale + row0 + i);
        const f4 b = *reinterpret_cast<const f4*>(p.shift + row0 + i);
        sc[i] = a[0]; sc[i + 1] = a[1]; sc[i + 2] = a[2]; sc[i + 3] = a[3];
        sh[i] = b[0]; sh[i + 1] = b[1]; sh[i + 2] = b[2]; sh[i + 3] = b[3];
    }
    T* const ybase = reinterpret_cast<T*>(p.y);
    const T* const rbase = reinterpret_cast<const T*>(p.res);
    const int oy0 = ty * TH + wrow0, ox = tx * 16 + l15;

    if (p.head_w) {                                      // fused OutConv, as in conv_igemm.hip
        float hw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) hw[i] = p.head_w[row0 + i];
#pragma unroll
        for (int g = 0; g < FP; ++g) {
            float part = 0.f;
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[f][g][r] * sc[f * 4 + r] + sh[f * 4 + r];
                    v = (p.relu && v < 0.f) ? 0.f : v;
                    part = __builtin_fmaf(v, hw[f * 4 + r], part);
                }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (q == 0) {
                const size_t pix = ((size_t)n * p.Ho + oy0 + g) * p.Wo + ox;
                const float l = part + p.head_b[0];
                p.head_logits[pix] = l;
                if (p.head_mask) p.head_mask[pix] = (1.f / (1.f + __expf(-l))) > p.head_thr ? 255 : 0;
            }
        }
        return;
    }

    constexpr int UN = __is_same(T, float) ? 4 : 8;
    constexpr int UPP = 64 / UN;
    constexpr int UPL = 16 * UPP / 64;
    constexpr int SROW = 272;
    static_assert(NW * 16 * SROW <= NSW * WSTAGE + 2 * HBYTES, "staging must fit in LDS");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    char* const stg = smem + wave * (16 * SROW);
    const int slab0 = ctTile * CT + wci * 64;
#pragma unroll
    for (int g = 0; g < FP; ++g) {
        const unsigned obase = (unsigned)((n * p.yHp + oy0 + g + 1) * p.yWp + ox + 1);
#pragma unroll
        for (int f = 0; f < FC; ++f) {
            f4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[f][g][r] * sc[f * 4 + r] + sh[f * 4 + r];
            *reinterpret_cast<f4*>(stg + l15 * SROW + (q * NV + f * 4) * 4) = t;
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int unit = lane + 64 * i;
            const int px = unit / UPP, cu = unit % UPP;
            float w[UN];
#pragma unroll
            for (int j = 0; j < UN; j += 4) {
                const f4 t = *reinterpret_cast<const f4*>(stg + px * SROW + (cu * UN + j) * 4);
                w[j] = t[0]; w[j + 1] = t[1]; w[j + 2] = t[2]; w[j + 3] = t[3];
            }
            const unsigned ob = __shfl(obase, px);
            const int co = slab0 + cu * UN;
            if (co < p.rows) {
                if (rbase) OutVec<T, UN>::add(rbase + (size_t)ob * p.rCs + p.rCoff + co, p.rCoff + co, w);
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < UN; ++j) w[j] = w[j] > 0.f ? w[j] : 0.f;
                }
                OutVec<T, UN>::store(ybase + (size_t)ob * p.yCs + p.yCoff + co, p.yCoff + co, w);
            }
        }
        asm volatile("" ::: "memory");
    }
}

// ---- host-side launch -------------------------------------------------------------------------------------
template <int CT, int TH, int NW, int TPS, int NSW>
static constexpr size_t halo_lds() {
    constexpr int HR = 18 * (TH + 2);
    constexpr int H = ((HR + 7) / 8 + NW - 1) / NW;
    return (size_t)NSW * TPS * CT * 128 + (size_t)2 * H * NW * 1024;
}

template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW>
static hipError_t launch_halo(const ConvParams& p, int n_images, hipStream_t stream) {
    const int tiles = n_images * (p.Ho / TH) * (p.Wo / 16);
    auto kern = conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW>;
    const size_t lds = halo_lds<CT, TH, NW, TPS, NSW>();
    static_assert(halo_lds<CT, TH, NW, TPS, NSW>() <= 160 * 1024, "LDS budget");
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * p.nCt)), dim3(64 * NW), lds, stream, p);
    return hipGetLastError();
}

template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW>
static hipError_t prepare_halo() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// configurations: 64 channels x 16x16 patch (8 waves as 1 x 8, two patch rows each), one filter ROW per stage, ring 2;
//                 128 channels x 16x16 patch (8 waves as 2 x 4, four patch rows each), one tap per stage, ring 3
#define CV_FOR_EACH_HALO(X, T)   \
    X(T, 64, 16, 1, 8, 3, 2)     \
    X(T, 128, 16, 2, 8, 1, 3)

hipError_t conv_halo_prepare() {
    hipError_t e;
#define X(T, CT, TH, WGC, NW, TPS, NSW) \
    if ((e = prepare_halo<T, CT, TH, WGC, NW, TPS, NSW>()) != hipSuccess) return e;
    CV_FOR_EACH_HALO(X, half_t)
    CV_FOR_EACH_HALO(X, float)
    CV_FOR_EACH_HALO(X, split_t)
#undef X
    return hipSuccess;
}

// The 64-row tile is instantiated and tested (CV_HALO64=1) but measured slower than conv_igemm's 64x256 tile with
// two workgroups per CU (r01_tuning.md step 14): its 24-MFMA stages are barrier-bound.  Default: 128-row tile only.
bool conv_halo_supported(int ct, int Ho, int Wo) {
    static const bool allow64 = [] { const char* v = std::getenv("CV_HALO64"); return v && v[0] == '1'; }();
    return (ct == 128 || (ct == 64 && allow64)) && Ho % 16 == 0 && Wo % 16 == 0;
}

hipError_t conv_halo_launch(int ct, int dt, const ConvParams& p, int n_images, hipStream_t stream) {
#define X(T, CT, TH, WGC, NW, TPS, NSW) \
    if (ct == CT) return launch_halo<T, CT, TH, WGC, NW, TPS, NSW>(p, n_images, stream);
    if (dt == kF16) { CV_FOR_EACH_HALO(X, half_t) } else if (dt == kSplit) { CV_FOR_EACH_HALO(X, split_t) } else { CV_FOR_EACH_HALO(X, float) }
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cv
