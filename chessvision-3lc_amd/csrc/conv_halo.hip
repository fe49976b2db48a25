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
// K loop: one stage = (channel block, tap), one s_barrier per stage, software pipelined through registers: the
// fragments of stage s+1 are read while stage s computes (see the comment at the main loop).  The weight ring is 3
// deep, the halo double buffered; DMA completes in issue order, so the counted s_waitcnt at the barrier of stage s
// may leave in flight exactly what was issued after W(s+1).
#include "cv_kernels.h"
#include "conv_igemm.h"
#include "conv_device.h"

#include <cstdlib>

namespace cv {

// TPS = taps per weight stage and NSW = weight ring depth (1 and 3 in every shipped configuration: the pipelined loop is
// written for them; a filter-row variant was measured and dropped, r01_tuning.md).
// IMG = 0: one TH x 16 patch of one (large) image per workgroup.  IMG = 8: feature maps of 8 x 8 (ResNet layer2): the
// 16 x 16 pixel tile is four whole images (2 x 2), whose zero-bordered 10 x 10 PHWC planes are contiguous in memory, so
// the "halo" is simply 400 consecutive pixels and a fragment's 16 lanes read row y of two neighbouring images.
// PERSIST: the grid is one workgroup per CU slot and every workgroup walks tiles lid, lid + grid, ...; the first DMAs of the
// next tile (its halo and three weight stages) are issued before the current tile's epilogue, so their latency, the
// store drain and the workgroup relaunch disappear behind it (r01_tuning.md step 22).
// DBH: halo double buffered (the next channel block's halo lands while the current one is computed).  DBH = false keeps ONE
// halo buffer: at every channel-block boundary the workgroup drains, DMAs the next halo and waits for it -- a bubble that the
// second workgroup resident on the CU fills; what it buys is LDS: a 64-channel tile over a 16 x 16 patch (four patch rows per
// wave = twice the MFMAs per weight byte and per fragment read of the 8 x 16 tile) then fits twice per CU.
// FUSE0 (single halo buffer only): the halo is not DMA'd but PRODUCED in LDS by the network's first layer (3 -> 64 channels, K =
// 27 -> one MFMA k-step) from a 20 x 20 x 3 patch of the caller's image -- UNet inc.double_conv.0 fused into inc.double_conv.3.
// The 64-channel full-resolution tensor between the two convs (1.07 GB per 64 images, written once and read 1.27x) never
// exists; the recompute is the 18^2/16^2 halo overlap of a layer that holds 0.2 % of the network's MACs.
#ifndef CV_HALO_TH8_SINGLE
#define CV_HALO_TH8_SINGLE 0      // experiment (with -DCV_HALO_TH64=8): 8 x 16 patch with ONE halo buffer = 47 KB -> THREE workgroups per CU
#endif
constexpr int halo_waves_per_eu(int ct, int th, bool dbh, int chain = 0) { return chain == 1 ? 1 : (CV_HALO_TH8_SINGLE && ct == 64 && th == 8 && !dbh) ? 3 : 2; }
// CHAIN (single halo buffer, f16, 64 -> 64 channels, 16 x 16 maps = one patch per image): the "channel blocks" of the K loop are
// FOUR CONVOLUTIONS in a row (ResNet-18 layer1 = two BasicBlocks).  At every block boundary the epilogue of convolution c -- BN,
// (+ f32 residual), ReLU -- rounds to f16 and writes the result over the halo buffer IN PLACE (every wave has drained its reads of
// it by then; the zero border stays), which is then the resident input of convolution c + 1; the weight stream simply continues.
// The three intermediate tensors never travel to HBM as f16; the f32 trunk twin does (written after convolution 1, read back by
// the same lanes after convolution 3): round 5, profiles/r05_tuning.md.  CHAIN = 2: that form, two workgroups per CU.  CHAIN = 1: ONE
// workgroup per CU with 512 registers per lane -- the f32 residual of the whole tile (64 registers) is fetched at kernel start and stays in
// registers through both blocks, so the first block's output never leaves the chip and no epilogue waits for a load.
template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW, int IMG, bool PERSIST, bool DBH = true, bool FUSE0 = false, int CHAIN = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(halo_waves_per_eu(CT, TH, DBH, CHAIN), halo_waves_per_eu(CT, TH, DBH, CHAIN)))) void conv3x3_halo_kernel(const ConvParams p) {
    static_assert(!FUSE0 || (!DBH && !PERSIST && IMG == 0 && CT == 64 && NW == 4 && __is_same(T, split_t)), "fused producer: split-f16 64-channel single-halo tile");
    static_assert(!CHAIN || (!DBH && !PERSIST && !FUSE0 && IMG == 0 && CT == 64 && TH == 16 && NW == 4 && __is_same(T, half_t)), "chained convolutions: f16 64-channel single-halo tile over whole 16 x 16 images");
    static_assert(TPS == 1 && (NSW == 3 || NSW == 4), "stage shape");
    constexpr int SPC = 9 / TPS;                        // stages per channel block
    constexpr int WGP = NW / WGC;                       // wave groups along the patch rows
    static_assert(CT / WGC == 64, "every wave owns a 64-channel slab");
    constexpr int FC = 4;
    constexpr int FP = TH / WGP;                        // patch rows (= 16-pixel fragments) per wave
    static_assert(FP >= 1 && FP <= 4 && TH % WGP == 0, "wave tile");
    constexpr int WTAP = CT * 128;                      // weight bytes of one tap
    constexpr int WSTAGE = TPS * WTAP;
    static_assert(IMG == 0 || (IMG == 8 && TH == 16), "packed-image mode: 2 x 2 images of 8 x 8");
    constexpr int HLW = IMG ? IMG + 2 : 18;             // pixels per halo line
    constexpr int HR = IMG ? 4 * HLW * HLW : 18 * (TH + 2);   // halo rows (one pixel = one 128-byte LDS row)
    // DMA roles.  With 8 waves, waves w and w + 4 share a SIMD: if both issued their DMA pieces right after the barrier
    // (each piece costs its wave ~60-100 issue cycles) the matrix pipe of that SIMD would sit idle meanwhile.  So waves
    // 0..3 move the weight stages and waves 4..7 the halo (two pieces per stage over taps 0..5): on every SIMD one wave
    // issues DMA while the other already feeds MFMAs.  4-wave workgroups (one wave per SIMD) keep symmetric duties.
    constexpr bool kRoles = NW == 8;
    constexpr int NWI = kRoles ? NW / 2 : NW;           // waves sharing one kind of DMA
    constexpr int H = ((HR + 7) / 8 + NWI - 1) / NWI;   // halo DMA wave-instructions per issuing wave (8 rows each)
    constexpr int HPS = kRoles ? (H + 5) / 6 : H;       // ... of which per stage (roles: spread over taps 0..5)
    constexpr int HBYTES = H * NWI * 1024;
    constexpr int LW = TPS * CT / (8 * NWI);
    static_assert(LW >= 1 && LW + H <= 63 && (!kRoles || 6 * HPS >= H), "vmcnt range / halo spread");
    constexpr bool kSplit16 = sizeof(T) == 4 && !__is_same(T, float);
    typedef typename FragT<T>::V V;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem + NSW * WSTAGE;
#if CV_STAMP
    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned nwg = gridDim.x, bid = blockIdx.x;   // XCD-aware remap, as in conv_igemm.hip
    const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tilesX = IMG ? 1 : p.Wo / 16, tilesY = IMG ? 1 : p.Ho / TH;
    const int nImg = p.M / (p.Ho * p.Wo);
    // Split-K (single boards: a 16 x 16 feature map is ONE patch, so a deep layer has Cout/64 tiles and a K loop of hundreds of stages):
    // the channel blocks of a tile's K loop are dealt to p.ksplit workgroups, p.kper blocks each; every workgroup leaves its raw f32
    // accumulators in p.partial[split][pixel][channel] and conv_splitk_reduce_kernel (conv_igemm.hip) sums them in split order and runs
    // the epilogue.  Plain launches only (not persistent, no fused producer, no packed images).
    const int kSplits = (!PERSIST && !FUSE0 && IMG == 0 && p.ksplit > 1) ? p.ksplit : 1;
    const int nCbAll = p.nStages / 9;                   // p.nStages counts (channel block, tap) pairs
    int nCb = nCbAll, nS = nCb * SPC, cbBase = 0, ksCur = 0;
    const unsigned nTiles = (unsigned)((IMG ? (nImg + 3) / 4 : nImg * tilesX * tilesY) * p.nCt);
    // the tile the DMA side works on (in a persistent workgroup that is already the NEXT tile during an epilogue)
    int ctTile, tx, ty, n;
    unsigned hbase;
    const char* wsrc;                                   // wave-uniform: this wave's first weight piece of the tile (stage 0)
    // 4-wave tiles have registers to spare: the byte offsets of this lane's halo rows are computed once per tile instead of
    // once per channel block (a divide by 18 and a swizzle per piece: ~14 vector instructions x 11 pieces in one stage)
    constexpr bool kHoistHalo = NW == 4;
    unsigned hoff[kHoistHalo ? H : 1];

    // DMA sources of this lane's halo rows: row r of the halo <-> padded input pixel (ty*TH + r/18, tx*16 + r%18).
    // Needed once per channel block only, so they are re-derived at each use (the lane id is made opaque to keep LICM
    // from parking H more VGPRs across the whole K loop).
    const unsigned xpix = (unsigned)p.xCs * (unsigned)sizeof(T);
    const char* const xsrc = p.x + p.xCoffBytes;
    const bool is_w = !kRoles || wave < NWI;            // this wave moves weight stages / halo pieces
    const bool is_h = !kRoles || wave >= NWI;
    const int wi = kRoles && wave >= NWI ? wave - NWI : wave;     // index among the waves of its role
    const unsigned lane16 = (unsigned)lane * 16u;
    // byte offset (from xsrc, channel block 0) of the halo row this lane moves in piece i
    auto halo_row_offset = [&](int i) __attribute__((always_inline)) -> unsigned {
        const int r = (i * NWI + wi) * 8 + (lane >> 3);
        const int rr = r < HR ? r : HR - 1;             // rows of the padded tail re-read the last real one
        if constexpr (IMG == 0) {
            const int hy = rr / 18, hx = rr - hy * 18;
            return (hbase + (unsigned)(hy * p.xWp + hx)) * xpix + (unsigned)(((lane & 7) ^ (hx & 7)) * 16);
        } else {                                         // 400 consecutive pixels; images past the batch re-read the last one
            const int lim = (nImg - n) * (HLW * HLW) - 1;
            const int rc = rr < lim ? rr : lim;
            const int hx = (rr % (HLW * HLW)) % HLW;
            return (unsigned)(n * (HLW * HLW) + rc) * xpix + (unsigned)(((lane & 7) ^ (hx & 7)) * 16);
        }
    };
    auto decode = [&](unsigned tile) __attribute__((always_inline)) {
        int pt;
        if (kSplits > 1) {
            // the patches of one (channel tile, K split) are neighbours: they stream the same weight slab through one L2
            const int nPatch = nImg * tilesX * tilesY;
            pt = tile % nPatch;
            const int rest = tile / nPatch;
            ctTile = rest % p.nCt;
            ksCur = rest / p.nCt;
            cbBase = ksCur * p.kper;
            nCb = nCbAll - cbBase < p.kper ? nCbAll - cbBase : p.kper;
            nS = nCb * SPC;
        } else {
            ctTile = tile % p.nCt;
            pt = tile / p.nCt;
        }
        tx = pt % tilesX; pt /= tilesX;
        ty = pt % tilesY;
        n = IMG ? 4 * (pt / tilesY) : pt / tilesY;      // (first) image of the tile
        hbase = (unsigned)((n * p.xHp + ty * TH) * p.xWp + tx * 16);
        wsrc = p.w + ((size_t)ctTile * p.nStages + (size_t)cbBase * 9) * WTAP + wi * 1024;
        if constexpr (kHoistHalo) {
#pragma unroll
            for (int i = 0; i < H; ++i) hoff[i] = halo_row_offset(i);
        }
    };

    const int q = lane >> 4, l15 = lane & 15;
    auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {
        char* sW = smem + slot * WSTAGE;
#pragma unroll
        for (int i = 0; i < LW; ++i) glds16s(wsrc + (size_t)s * WSTAGE + i * (NWI * 1024), lane16, sW + (i * NWI + wi) * 1024);   // stage s = TPS consecutive taps
    };
    auto issue_halo = [&](int cb, int hb, auto i0_tag, auto n_tag) __attribute__((always_inline)) {          // pieces [I0, I0 + N) of this wave's H
        constexpr int I0 = decltype(i0_tag)::value, N = decltype(n_tag)::value;
        char* sH = halo + hb * HBYTES;
        const char* const src = xsrc + (cbBase + cb) * 128;   // wave-uniform base of this channel block
        if constexpr (kHoistHalo) {
#pragma unroll
            for (int i = I0; i < I0 + N && i < H; ++i) glds16s(src, hoff[i], sH + (i * NWI + wi) * 1024);
        } else {
            // 8-wave tiles sit at the register limit: the offsets are re-derived at each use (needed once per channel block; the
            // lane id is made opaque to keep LICM from parking H more VGPRs across the whole K loop)
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int i = I0; i < I0 + N && i < H; ++i) {
                const int r = (i * NWI + wi) * 8 + (ln >> 3);
                const int rr = r < HR ? r : HR - 1;     // rows of the padded tail re-read the last real one
                unsigned off;
                if constexpr (IMG == 0) {
                    const int hy = rr / 18, hx = rr - hy * 18;
                    off = (hbase + (unsigned)(hy * p.xWp + hx)) * xpix + (unsigned)(((ln & 7) ^ (hx & 7)) * 16);
                } else {                                 // 400 consecutive pixels; images past the batch re-read the last one
                    const int lim = (nImg - n) * (HLW * HLW) - 1;
                    const int rc = rr < lim ? rr : lim;
                    const int hx = (rr % (HLW * HLW)) % HLW;
                    off = (unsigned)(n * (HLW * HLW) + rc) * xpix + (unsigned)(((ln & 7) ^ (hx & 7)) * 16);
                }
                glds16s(src, off, sH + (i * NWI + wi) * 1024);
            }
        }
    };

    // ---- fused producer (FUSE0) --------------------------------------------------------------------------------------
    constexpr int PW0 = 20;                                        // patch pitch (floats): image rows / cols ty*16-2 .. +17
    // [3][20][20] image values + one zero slot (4864 B), each ALREADY split: low half = f16(v), high half = f16(v - hi).  The split
    // is done once per patch element here instead of once per use in the fragment builds (every element is read ~7 times per
    // channel block): a B fragment is then eight ds_read_b32 and eight v_perm_b32, no conversions.
    unsigned* const patch0 = reinterpret_cast<unsigned*>(halo + HBYTES);
    half8* const w0lds = reinterpret_cast<half8*>(halo + HBYTES + 4864);      // first-layer fragments of channel block 1 (4 KB)
    float* const sc0lds = reinterpret_cast<float*>(halo + HBYTES + 4864 + 4096);   // scale[64], shift[64]
    auto load_patch0 = [&]() __attribute__((always_inline)) {
        // everything the two productions need that is not lane-private goes to LDS once, in one round of global latency: the
        // image patch, the weight fragments of channel block 1 (block 0's are used right away, from registers) and the BN constants
        w0lds[tid] = reinterpret_cast<const half8*>(p.f0_w)[2 * 2 * 64 + tid];
        if (tid < 64) { sc0lds[tid] = p.f0_scale[tid]; sc0lds[64 + tid] = p.f0_shift[tid]; }
        float bad0 = 0.f;
        for (int idx = tid; idx < 3 * 20 * PW0 + 1; idx += 64 * NW) {
            const int c = idx / (20 * PW0), rem = idx - c * (20 * PW0), r = rem / PW0, col = rem - r * PW0;
            const int gy = ty * TH - 2 + r, gx = tx * 16 - 2 + col;
            float v = 0.f;
            if (idx < 3 * 20 * PW0 && gy >= 0 && gy < p.xHp - 2 && gx >= 0 && gx < p.xWp - 2) {
                const int Hh = p.xHp - 2, Ww = p.xWp - 2;
                if (p.f0_u8) v = (float)reinterpret_cast<const uint8_t*>(p.f0_x)[((size_t)(n * Hh + gy) * Ww + gx) * 3 + c] / 255.f;
                else v = reinterpret_cast<const float*>(p.f0_x)[((size_t)(n * 3 + c) * Hh + gy) * Ww + gx];
            }
            bad0 = __builtin_fmaf(v, 0.f, bad0);
            v *= p.f0_in_mul;
            const half_t vh = (half_t)v;
            const half_t vl = (half_t)(v - (float)vh);
            patch0[idx] = (unsigned)__builtin_bit_cast(unsigned short, vh) | ((unsigned)__builtin_bit_cast(unsigned short, vl) << 16);
        }
        if (bad0 != bad0 && p.flag) atomicMin(p.flag, 0u);         // layer id 0 = the caller's input tensor
    };
    // halo(cb) <- relu(bn(conv3x3(patch))) for channels 32*cb .. 32*cb+31 of the 18 x 18 halo pixels, zero outside the image (the
    // zero padding of THIS layer), written in the layout the tap views read: pixel row of 128 B, 16-byte chunks swizzled by the
    // pixel column, [hi, lo] / [lo, hi] by group parity.  Lane (pixel l15, group q) ends with the 8 channels of group q.
    auto produce0 = [&](int cb) __attribute__((always_inline)) {
        half8 ah0, al0, ah1, al1;
        if (cb == 0) {                                   // wave-uniform
            const half8* w0 = reinterpret_cast<const half8*>(p.f0_w) + lane;
            ah0 = w0[0 * 64]; al0 = w0[1 * 64]; ah1 = w0[2 * 64]; al1 = w0[3 * 64];
        } else {
            ah0 = w0lds[0 * 64 + lane]; al0 = w0lds[1 * 64 + lane]; ah1 = w0lds[2 * 64 + lane]; al1 = w0lds[3 * 64 + lane];
        }
        float sc0[8], sh0[8];
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            const f4 a = *reinterpret_cast<const f4*>(sc0lds + cb * 32 + q * 8 + j);
            const f4 b = *reinterpret_cast<const f4*>(sc0lds + 64 + cb * 32 + q * 8 + j);
            sc0[j] = a[0]; sc0[j + 1] = a[1]; sc0[j + 2] = a[2]; sc0[j + 3] = a[3];
            sh0[j] = b[0]; sh0[j + 1] = b[1]; sh0[j + 2] = b[2]; sh0[j + 3] = b[3];
        }
        int ko[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = q * 8 + j, tap = k / 3, c = k - tap * 3, ky = tap / 3, kx = tap - ky * 3;
            ko[j] = k < 27 ? (c * 20 + ky) * PW0 + kx : 3 * 20 * PW0;      // relative to the halo pixel's patch origin | the zero slot
        }
        // halo pixel of (fragment fr, lane): hp = 16 fr + l15, walked in steps of NW fragments = 64 pixels = 3 halo lines + 10 --
        // additions instead of a divide by 18 per fragment.  A wave owns fragments wave, wave + NW, ... (5 or 6 of the 21); they are
        // produced THREE at a time, phase by phase (addresses | 24 patch reads | 18 MFMAs in six independent chains | BN, ReLU, hi/lo
        // split | stores): one fragment at a time every phase waited out the latency of the one before it (LDS read -> dependent
        // MFMAs -> conversions), on an issue port the partner workgroup's MFMA stream already half fills (in-kernel stamps, r03: the
        // two productions took 38 % of the tile's life).
        constexpr int NF = (HR + 15) / 16, U = 3;
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        int hp = wave * 16 + l15;
        int hy = hp / 18, hx = hp - hy * 18;
        for (int fr0 = wave; fr0 < NF; fr0 += U * NW) {
            int hyu[U], hxu[U], hpu[U];
            unsigned w8[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                hpu[u] = hp;
                const bool tail = hp >= HR;              // padded tail of the last fragment / slots past the last fragment: nothing is stored
                hyu[u] = tail ? 17 : hy; hxu[u] = tail ? 17 : hx;
                const int pb = hyu[u] * PW0 + hxu[u];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[u][j] = patch0[(q * 8 + j < 27 ? pb : 0) + ko[j]];
                hp += 16 * NW;
                hx += (16 * NW) % 18; hy += (16 * NW) / 18;
                if (hx >= 18) { hx -= 18; hy += 1; }
            }
            f4 a0[U], a1[U];
            half8 bh[U], bl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                u4 uh, ul;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    uh[m] = __builtin_amdgcn_perm(w8[u][2 * m + 1], w8[u][2 * m], 0x05040100u);     // the two hi halves
                    ul[m] = __builtin_amdgcn_perm(w8[u][2 * m + 1], w8[u][2 * m], 0x07060302u);     // the two lo halves
                }
                bh[u] = __builtin_bit_cast(half8, uh); bl[u] = __builtin_bit_cast(half8, ul);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh[u], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                a1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh[u], f4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh[u], a0[u], 0, 0, 0);
                a1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh[u], a1[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl[u], a0[u], 0, 0, 0);
                a1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl[u], a1[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int iy = ty * TH - 1 + hyu[u], ix = tx * 16 - 1 + hxu[u];
                // unsigned compares: one test per coordinate, no short-circuit branches
                const bool inside = ((unsigned)iy < (unsigned)(p.xHp - 2)) & ((unsigned)ix < (unsigned)(p.xWp - 2));
                u4 hi8, lo8;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float v0 = (j < 4 ? a0[u][j] : a1[u][j - 4]) * sc0[j] + sh0[j];
                    float v1 = (j < 4 ? a0[u][j + 1] : a1[u][j - 3]) * sc0[j + 1] + sh0[j + 1];
                    v0 = inside ? __builtin_fmaxf(v0, 0.f) : 0.f;
                    v1 = inside ? __builtin_fmaxf(v1, 0.f) : 0.f;
                    unsigned hp, lp;
                    split_pair(v0, v1, hp, lp);
                    hi8[j / 2] = hp; lo8[j / 2] = lp;
                }
                if (hpu[u] < HR) {
                    char* row = halo + hpu[u] * 128;
                    *reinterpret_cast<u4*>(row + (((2 * q + (q & 1)) ^ (hxu[u] & 7)) << 4)) = hi8;
                    *reinterpret_cast<u4*>(row + (((2 * q + 1 - (q & 1)) ^ (hxu[u] & 7)) << 4)) = lo8;
                }
            }
        }
    };

    const int wci = wave / WGP, wpi = wave % WGP;
    const int rowW = (wci * 64 + l15) * 128;
    const int wrow0 = wpi * FP;                          // first patch row of this wave
    const int l7 = lane & 7;

    f4 acc[FC][FP];

    // Fragment sets of one stage.  Every precision reads two 16-byte chunks per lane and operand: split-f16 the hi and
    // the lo chunk of its channel group (products hi.hi, lo.hi, hi.lo), f16 / f32 the two k-halves of the 128-byte line
    // (products set0.set0, set1.set1).
    struct Frags { V a[2][FC]; V b[2][FP]; };
    const int c0 = kSplit16 ? 2 * q + (q & 1) : q;
    const int c1 = kSplit16 ? 2 * q + 1 - (q & 1) : 4 + q;

    // LDS read addresses = one VGPR per (chunk set[, filter column]) + an immediate: the weight row of this lane, and
    // the halo pixel (patch row 0 of this wave, column l15 + kx).  The halo image is XOR-swizzled by the pixel's COLUMN
    // (hx & 7; all 16 lanes of a fragment read share the halo line, so this is as conflict-free as a row key), which
    // makes the swizzle independent of the filter row and of the patch row: both are immediates.  The ring slot of
    // stage s is s % 3 = tap % 3 (nine taps per channel block), an immediate as well.  (r01: the address arithmetic of
    // the previous layout cost ~35 VALU issues per stage and wave -- the SIMD's vector issue port is shared with the
    // MFMAs of both resident waves.)
    int aoff[2], boff[2][3];
    aoff[0] = rowW + ((c0 ^ l7) << 4);
    aoff[1] = rowW + ((c1 ^ l7) << 4);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        // halo row of (first patch row of this wave, pixel l15) for filter column kx, and its swizzle key (the column)
        const int col = IMG ? (l15 & 7) + kx : l15 + kx;
        const int row = IMG ? ((wrow0 >> 3) * 2 + (l15 >> 3)) * (HLW * HLW) + (wrow0 & 7) * HLW + col : wrow0 * 18 + col;
        boff[0][kx] = NSW * WSTAGE + row * 128 + ((c0 ^ (col & 7)) << 4);
        boff[1][kx] = NSW * WSTAGE + row * 128 + ((c1 ^ (col & 7)) << 4);
    }
    // Weight ring slot of a stage: NSW == 3 divides the nine taps of a channel block, so slot = tap % 3 is an immediate;
    // NSW == 4 (one more stage of L2 -> LDS latency hidden) tracks the slot at run time: `wslot` = byte offset of the slot
    // of the stage being computed, two VALU adds per stage.
    int wslot = 0;
    auto next_slot = [&](int off) __attribute__((always_inline)) { return off + WSTAGE == NSW * WSTAGE ? 0 : off + WSTAGE; };
    auto load_a = [&](Frags& F, auto slot_tag, int roff) __attribute__((always_inline)) {        // weights: published by the barrier of the previous stage
        constexpr int SLOT = decltype(slot_tag)::value;
        const int so = NSW == 3 ? SLOT * WSTAGE : roff;
        const int a0 = aoff[0] + so, a1 = aoff[1] + so;
#pragma unroll
        for (int f = 0; f < FC; ++f) F.a[0][f] = *reinterpret_cast<const V*>(smem + a0 + f * 2048);
#pragma unroll
        for (int f = 0; f < FC; ++f) F.a[1][f] = *reinterpret_cast<const V*>(smem + a1 + f * 2048);
    };
    auto load_b = [&](Frags& F, auto tap_tag, auto hb_tag) __attribute__((always_inline)) {      // pixels: nine shifted views of the resident halo
        constexpr int TAP = decltype(tap_tag)::value;
        constexpr int KY = TAP / 3, KX = TAP % 3;
        constexpr int BUF = decltype(hb_tag)::value * HBYTES;
#pragma unroll
        for (int g = 0; g < FP; ++g) F.b[0][g] = *reinterpret_cast<const V*>(smem + boff[0][KX] + (BUF + (g + KY) * (HLW * 128)));
#pragma unroll
        for (int g = 0; g < FP; ++g) F.b[1][g] = *reinterpret_cast<const V*>(smem + boff[1][KX] + (BUF + (g + KY) * (HLW * 128)));
    };
    // first third (split) / half of the stage's MFMAs: needs set 0 only
    auto mma_head = [&](const Frags& F) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < FC; ++f)
#pragma unroll
            for (int g = 0; g < FP; ++g) mma16(acc[f][g], F.a[0][f], F.b[0][g]);
    };
    auto mma_tail = [&](const Frags& F) __attribute__((always_inline)) {               // passes over all accumulators: dependent MFMAs stay FC*FP apart
        if constexpr (kSplit16) {
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int g = 0; g < FP; ++g) mma16(acc[f][g], F.a[1][f], F.b[0][g]);
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int g = 0; g < FP; ++g) mma16(acc[f][g], F.a[0][f], F.b[1][g]);
        } else {
#pragma unroll
            for (int f = 0; f < FC; ++f)
#pragma unroll
                for (int g = 0; g < FP; ++g) mma16(acc[f][g], F.a[1][f], F.b[1][g]);
        }
    };
    constexpr int kMfmaPerMma = __is_same(T, float) ? 4 : 1;
    constexpr int NTAIL = FC * FP * (kSplit16 ? 2 : 1) * kMfmaPerMma;         // MFMA instructions of the tail
    constexpr int MPL = (NTAIL - LW) / (2 * FC) > 2 ? 2 : ((NTAIL - LW) / (2 * FC) > 0 ? (NTAIL - LW) / (2 * FC) : 1);

    // ---- main loop: channel blocks x taps, register double-buffered ----------------------------------------
    // Stage s = (channel block, tap).  Its fragments were read from LDS during stage s-1, so the MFMA pipe never waits
    // for the barrier-then-read sequence:
    //     MFMA head(s)  |  wait W(s+1) landed + barrier  |  DMA W(s+3) (+ next halo at tap 0)  |  ds_read frags(s+1)
    //     |  MFMA tail(s)                      <- covers the LDS latency of the reads just issued
    // The barrier of stage s publishes W(s+1) and frees the slot of W(s) (every wave's reads of it completed before
    // its own lgkmcnt(0) at that barrier).  DMA completes in issue order, so the counted wait may leave in flight
    // what was issued after W(s+1): W(s+2), and the next halo when it was issued one or two stages ago.
#if CV_STAMP
    unsigned long long st_wait = 0, st_head = 0, st_tail = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- CHAIN: epilogue of convolution c, straight from the accumulator layout -------------------------------------
    // lane (q, l15) holds channels 16 q .. 16 q + 15 of the pixels (patch row wrow0 + g, column l15), g = 0 .. FP-1: 64 contiguous
    // bytes of an f32 plane / 32 of an f16 plane / two 16-byte chunks of the pixel's halo row per g.
    constexpr int kChainStores = CHAIN == 2 ? 4 * FP : 0;       // f32 stores of convolution 1's epilogue (vmcnt bookkeeping)
    f4 rtrunk[CHAIN == 1 ? FP : 1][4];                          // CHAIN 1: the f32 trunk of this lane's outputs (input twin, then block 0's output)
    auto chain_load_trunk = [&]() __attribute__((always_inline)) {
        if constexpr (CHAIN == 1) {
            const float* const rb = reinterpret_cast<const float*>(p.ch_res0) + q * 16;
            const unsigned pix0 = (unsigned)((n * p.yHp + wrow0 + 1) * p.yWp + l15 + 1);
#pragma unroll
            for (int g = 0; g < FP; ++g)
#pragma unroll
                for (int f = 0; f < 4; ++f) rtrunk[g][f] = *reinterpret_cast<const f4*>(rb + (size_t)(pix0 + (unsigned)(g * p.yWp)) * 64 + f * 4);
        }
    };
    // CHAIN 2, convolution 1 (block 0's output): the residual epilogue in the UNIT layout of the kernel's ordinary epilogue -- the
    // accumulators are staged through LDS (the halo buffer is dead here and is the staging area; the weight ring stays live) and read
    // back so that a lane holds 8 consecutive channels of a pixel and 8 consecutive lanes one pixel: the f32 residual arrives in one
    // burst of full lines, the f32 output leaves in full lines, and a unit's f16 values are exactly one 16-byte chunk of the pixel's
    // halo row, written once every wave has finished with the staging area.  (Straight from the accumulator layout the same traffic was
    // 64 requests of 16 bytes per wave instruction: r05_tuning.md step 2.)
#ifndef CV_CHAIN_MIDSTAGE_BUILD
#define CV_CHAIN_MIDSTAGE_BUILD 0      // 1: also build the staged form of convolution 1's epilogue (CV_CHAIN_MIDSTAGE=1 selects it).  Measured (r05_tuning.md step 8): its mere presence costs the kernel 35 spilled loop invariants and 10 %; enabled it wins half of that back
#endif
    auto chain_epilogue_res_mid = [&]() __attribute__((always_inline)) {
        if constexpr (CHAIN == 2 && CV_CHAIN_MIDSTAGE_BUILD) {
            constexpr int UN = 8, UPP = 64 / UN, UPL = 16 * UPP / 64, SROW = 272, RG = 2;
            static_assert(NW * RG * 16 * SROW <= HBYTES && FP % RG == 0, "staging fits in the halo buffer");
            float sc[16], sh[16];
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const f4 a = *reinterpret_cast<const f4*>(p.ch_scale[1] + q * 16 + i);
                const f4 b = *reinterpret_cast<const f4*>(p.ch_shift[1] + q * 16 + i);
                sc[i] = a[0]; sc[i + 1] = a[1]; sc[i + 2] = a[2]; sc[i + 3] = a[3];
                sh[i] = b[0]; sh[i + 1] = b[1]; sh[i + 2] = b[2]; sh[i + 3] = b[3];
            }
            const float* const rb = reinterpret_cast<const float*>(p.ch_res0);
            float* const ob32 = reinterpret_cast<float*>(p.ch_y32_mid);
            const float rm = p.ch_res_mul[0];
            // padded-plane pixel of (patch row wrow0 + g, column px) of image n
            auto pixel = [&](int g, int px) __attribute__((always_inline)) { return (unsigned)((n * p.yHp + wrow0 + g + 1) * p.yWp + px + 1); };
            // the residual of one row pair at a time (the whole tile's 64 registers spill here): the second pair's burst is issued before
            // the first pair is worked on
            f4 rraw[2][RG][UPL][2];
            auto fetch_pair = [&](int g0, int slot) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int i = 0; i < UPL; ++i) {
                        const int unit = lane + 64 * i, px = unit / UPP, co = (unit % UPP) * UN;
                        trunk32_fetch(rb + ((size_t)pixel(g0 + r, px) * 64 + co), rraw[slot][r][i]);
                    }
            };
            fetch_pair(0, 0);
            char* const stg = halo + wave * (RG * 16 * SROW);
            half8 hold[FP][UPL];
            float bad = 0.f;
#pragma unroll
            for (int g0 = 0; g0 < FP; g0 += RG) {
                if (g0 + RG < FP) fetch_pair(g0 + RG, ((g0 / RG) + 1) & 1);
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int f = 0; f < FC; ++f) {
                        f4 t;
#pragma unroll
                        for (int k = 0; k < 4; ++k) t[k] = acc[f][g0 + r][k] * sc[f * 4 + k] + sh[f * 4 + k];
                        *reinterpret_cast<f4*>(stg + r * (16 * SROW) + l15 * SROW + (q * 16 + f * 4) * 4) = t;
                    }
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < UPL; ++i) {
                    const int unit = lane + 64 * i, px = unit / UPP, cu = unit % UPP;
#pragma unroll
                    for (int r = 0; r < RG; ++r) {
                        float w[UN];
#pragma unroll
                        for (int j = 0; j < UN; j += 4) {
                            const f4 t = *reinterpret_cast<const f4*>(stg + r * (16 * SROW) + px * SROW + (cu * UN + j) * 4);
                            w[j] = t[0]; w[j + 1] = t[1]; w[j + 2] = t[2]; w[j + 3] = t[3];
                        }
                        trunk32_add_raw(rraw[(g0 / RG) & 1][r][i], w, rm);
#pragma unroll
                        for (int j = 0; j < UN; ++j) w[j] = __builtin_fmaxf(w[j], 0.f);
                        trunk32_store(ob32 + ((size_t)pixel(g0 + r, px) * 64 + cu * UN), w);
                        half8 h;
#pragma unroll
                        for (int j = 0; j < UN; ++j) { h[j] = (half_t)w[j]; bad = __builtin_fmaf((float)h[j], 0.f, bad); }
                        hold[g0 + r][i] = h;
                    }
                }
                wave_lds_sync();
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read its staged rows back: the buffer becomes the halo again
#pragma unroll
            for (int g = 0; g < FP; ++g)
#pragma unroll
                for (int i = 0; i < UPL; ++i) {
                    const int unit = lane + 64 * i, px = unit / UPP, cu = unit % UPP, hx = px + 1;
                    *reinterpret_cast<half8*>(halo + ((wrow0 + g + 1) * 18 + hx) * 128 + ((cu ^ (hx & 7)) << 4)) = hold[g][i];
                }
            // the staging area covered halo rows 0 .. 271, border pixels included: the zero border (= the next convolution's padding) is
            // restored -- 68 border pixels x 8 chunks of 16 bytes
            typedef unsigned u4z __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = tid + 256 * k, bp = idx >> 3, ch = idx & 7;
                if (bp < 68) {
                    const int j = bp - 36;
                    const int hy = bp < 18 ? 0 : bp < 36 ? 17 : 1 + (j >> 1);
                    const int hxz = bp < 18 ? bp : bp < 36 ? bp - 18 : ((j & 1) ? 17 : 0);
                    *reinterpret_cast<u4z*>(halo + (hy * 18 + hxz) * 128 + ch * 16) = u4z{0u, 0u, 0u, 0u};
                }
            }
            if (bad != bad && p.flag) atomicMin(p.flag, p.ch_layer_id[1]);
        }
    };
    auto chain_epilogue = [&](int c) __attribute__((always_inline)) {
        if constexpr (CHAIN) {
            if (CHAIN == 2 && CV_CHAIN_MIDSTAGE_BUILD && c == 1 && (p.chain & 16)) { chain_epilogue_res_mid(); return; }
            const bool has_res = (c & 1) != 0, last = c == 3;   // wave-uniform
            const float* const scp = p.ch_scale[c] + q * 16;
            const float* const shp = p.ch_shift[c] + q * 16;
            float sc[16], sh[16];
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const f4 a = *reinterpret_cast<const f4*>(scp + i);
                const f4 b = *reinterpret_cast<const f4*>(shp + i);
                sc[i] = a[0]; sc[i + 1] = a[1]; sc[i + 2] = a[2]; sc[i + 3] = a[3];
                sh[i] = b[0]; sh[i + 1] = b[1]; sh[i + 2] = b[2]; sh[i + 3] = b[3];
            }
            // padded-plane pixel of (patch row wrow0, column l15) of image n; planes are yHp x yWp pixels of 64 channels
            const unsigned pix0 = (unsigned)((n * p.yHp + wrow0 + 1) * p.yWp + l15 + 1);
            // residual: two patch rows in flight (the whole tile's 64 registers do not fit beside the accumulators and the next
            // stage's weight fragments)
            f4 r[2][4];
            const float* const rb = reinterpret_cast<const float*>(last ? p.ch_y32_mid : p.ch_res0) + q * 16;
            auto fetch_res = [&](int g) __attribute__((always_inline)) {
#pragma unroll
                for (int f = 0; f < 4; ++f) r[g & 1][f] = *reinterpret_cast<const f4*>(rb + (size_t)(pix0 + (unsigned)(g * p.yWp)) * 64 + f * 4);
            };
            if (CHAIN == 2 && has_res) fetch_res(0);
            const float rm = p.ch_res_mul[c >> 1];
            float* const o32 = reinterpret_cast<float*>(last ? p.y32 : p.ch_y32_mid) + q * 16;
            half_t* const o16 = reinterpret_cast<half_t*>(p.y) + q * 16;
            const int hx = l15 + 1;
            float bad = 0.f;
#pragma unroll
            for (int g = 0; g < FP; ++g) {
                if (CHAIN == 2 && has_res && g + 1 < FP) fetch_res(g + 1);
                float v[16];
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float t = acc[f][g][k] * sc[f * 4 + k] + sh[f * 4 + k];
                        if constexpr (CHAIN == 2) {
                            if (has_res) t = __builtin_fmaf(r[g & 1][f][k], rm, t);
                        } else {
                            if (has_res) t = __builtin_fmaf(rtrunk[g][f][k], rm, t);
                        }
                        v[f * 4 + k] = __builtin_fmaxf(t, 0.f);
                        if constexpr (CHAIN == 1) {
                            if (has_res) rtrunk[g][f][k] = v[f * 4 + k];   // block 0's output (stored units of ITS exponent) is block 1's residual
                        }
                    }
                half8 h0, h1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    h0[j] = (half_t)v[j]; h1[j] = (half_t)v[8 + j];
                    bad = __builtin_fmaf((float)h0[j], 0.f, bad); bad = __builtin_fmaf((float)h1[j], 0.f, bad);
                }
                const size_t pix = (size_t)(pix0 + (unsigned)(g * p.yWp));
                if (has_res && (CHAIN == 2 || last)) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) *reinterpret_cast<f4*>(o32 + pix * 64 + f * 4) = f4{v[f * 4], v[f * 4 + 1], v[f * 4 + 2], v[f * 4 + 3]};
                }
                if (last) {
                    *reinterpret_cast<half8*>(o16 + pix * 64) = h0;
                    *reinterpret_cast<half8*>(o16 + pix * 64 + 8) = h1;
                } else {
                    char* const row = halo + ((wrow0 + g + 1) * 18 + hx) * 128;
                    *reinterpret_cast<half8*>(row + (((2 * q) ^ (hx & 7)) << 4)) = h0;
                    *reinterpret_cast<half8*>(row + (((2 * q + 1) ^ (hx & 7)) << 4)) = h1;
                }
            }
            if (bad != bad && p.flag) atomicMin(p.flag, p.ch_layer_id[c]);
        }
    };

    Frags F0, F1;
    int s = 0;
    auto issue_prologue_halo = [&]() __attribute__((always_inline)) {
        if (is_h && !FUSE0) issue_halo(0, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
    };
    auto issue_prologue_w = [&]() __attribute__((always_inline)) {
        if (is_w) {
            issue_w(0, 0); issue_w(1, 1); issue_w(2, 2);
            if (NSW == 4) issue_w(3 < nS ? 3 : nS - 1, 3);
        }
    };
    auto issue_prologue = [&]() __attribute__((always_inline)) {                        // first DMAs of the tile `decode` was last called for
        issue_prologue_halo();
        issue_prologue_w();
    };
    auto stage = [&](Frags& cur, Frags& nxt, auto j_tag, auto hb_tag, int cb) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;
        constexpr int HB = decltype(hb_tag)::value;
        const bool more_cb = cb + 1 < nCb;
        // Single halo buffer: the fragments of tap 8 are read during tap 7's head, and every wave drains those reads
        // (lgkmcnt(0)) before tap 7's barrier -- past that barrier nobody reads the halo any more.  The next
        // block's halo is therefore issued in tap 7's DMA slot, flies during the rest of taps 7 and 8, and is waited for at
        // the end of tap 8, where the first fragments of the new block are read (they cannot be prefetched earlier).
        constexpr bool kRefill = !DBH && J == 8;
        constexpr bool kRefillIssue = !DBH && J == 7;
#if CV_STAMP
        const unsigned long long st_0 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_sched_barrier(0);              // head MFMAs (they wait for this stage's reads) stay behind the previous tail
        // the next tap's pixel fragments come from the resident halo: no need to wait for the barrier to read them
#if CV_ABLATE != 3
        if constexpr (!kRefill)
            load_b(nxt, std::integral_constant<int, (J + 1) % 9>{}, std::integral_constant<int, (DBH && J == 8 ? HB ^ 1 : HB)>{});
#endif
        mma_head(cur);
#if CV_SCHED_HINTS
        // an MFMA first: its wait for the weight fragments (read during the previous tail) is emitted as lgkmcnt(0),
        // which must not find a fresh read in the queue
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerMma, 0);
#pragma unroll
        for (int i = 0; i < 2 * FP; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // one ds_read
            __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerMma, 0);
        }
#endif
        // vmcnt at the barrier of stage s: W(s+1) must have landed, W(s+2) (issued one stage ago; a weight stage is
        // issued at every barrier, clamped at the end) may fly.  Symmetric duties: plus the next halo when it was issued
        // one or two stages ago.  Roles: the halo waves drain theirs at tap 7 (its last pieces leave at tap 5; the
        // next block's pixel reads start in the head of tap 8).
        // the next halo is younger than W(s+1): double buffered, issued at tap 0; single buffer, issued at tap 7
        const bool halo_young = !kRoles && !FUSE0 && !CHAIN && more_cb && (DBH ? (J >= 1 && J <= NSW - 1) : J == 8);
        // CHAIN: the sixteen f32 stores of convolution 1's epilogue are younger than the weight stage issued in its tap 8 and older than
        // the one issued in tap 0 of convolution 2: the first two barriers of convolution 2 may leave them in flight as well
        const bool chain_young = CHAIN == 2 && J <= 1 && cb == 2;
#if CV_STAMP
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_sched_barrier(0);              // the head MFMAs stay in front of the rendezvous, the tail behind it
        constexpr int FLY = (NSW - 2) * LW;             // weight pieces of the stages after W(s+1), still allowed in flight
        static_assert(FLY + H <= 63, "vmcnt range");
        if (kRoles && !is_w) {
            if (J == 7) wait_vm_barrier<0>();
            else wait_vm_barrier<63>();
        } else if (halo_young) wait_vm_barrier<FLY + H>();
        else if (chain_young) wait_vm_barrier<FLY + kChainStores>();
        else wait_vm_barrier<FLY>();
        __builtin_amdgcn_sched_barrier(0);
#if CV_STAMP
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        // One scheduling region from here to the next barrier: no branches around the DMA / LDS reads (the clamped
        // weight stage and the reads past the last stage are harmless repeats), and the hints below deal the DMA
        // issues and LDS reads out between the tail MFMAs instead of leaving the matrix pipe idle while both waves
        // of a SIMD issue them back to back (measured r01: that phase alone was 29 % of the stage).
#if CV_ABLATE != 1 && CV_ABLATE != 3
        if constexpr (kRoles) {
            // un-interleaved on purpose: the other wave of this SIMD covers the matrix pipe meanwhile
            if (is_w) issue_w(s + NSW < nS ? s + NSW : nS - 1, NSW == 3 ? J % 3 : wslot / WSTAGE);
            else if (DBH && J < 6 && more_cb) issue_halo(cb + 1, HB ^ 1, std::integral_constant<int, J * HPS>{}, std::integral_constant<int, HPS>{});
        } else {
            issue_w(s + NSW < nS ? s + NSW : nS - 1, NSW == 3 ? J % 3 : wslot / WSTAGE);
            if (kRefillIssue && !FUSE0 && !CHAIN && more_cb) issue_halo(cb + 1, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
            if (DBH && J == 0 && more_cb) issue_halo(cb + 1, HB ^ 1, std::integral_constant<int, 0>{}, std::integral_constant<int, H>{});
        }
#endif
#if CV_ABLATE != 3
        wslot = next_slot(wslot);                         // now the slot of stage s + 1
        // CHAIN 2, tap 8 of convolution 1: its epilogue needs these registers for the residual burst -- the weight fragments are read after it
        const bool late_a = CHAIN == 2 && CV_CHAIN_MIDSTAGE_BUILD && J == 8 && cb == 1 && (p.chain & 16);
        if (!late_a) load_a(nxt, std::integral_constant<int, (J + 1) % 3>{}, wslot);
#endif
#if CV_STAMP
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
#endif
        mma_tail(cur);
#if CV_SCHED_HINTS
        if constexpr (!kRoles) {
#pragma unroll
            for (int i = 0; i < LW + (kRefillIssue && !FUSE0 && !CHAIN ? H : 0); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // one LDS-DMA (VMEM read)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * FC; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // one ds_read
            __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
        }
#endif
        if constexpr (kRefill) {
            if (more_cb) {                               // wave-uniform
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FUSE0) {
                    // this wave passed tap 8's barrier, so every wave passed tap 7's, before which all reads of the old halo
                    // were drained: the buffer is free.  Produce the next block's halo, then publish it.
                    produce0(cb + 1);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else if constexpr (CHAIN) {
                    // same argument: nobody reads the halo any more.  Convolution cb's output replaces it, then the accumulators restart.
                    chain_epilogue(cb);
#pragma unroll
                    for (int f = 0; f < FC; ++f)
#pragma unroll
                        for (int g = 0; g < FP; ++g) acc[f][g] = f4{0.f, 0.f, 0.f, 0.f};
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (late_a) load_a(nxt, std::integral_constant<int, (J + 1) % 3>{}, wslot);
                } else {
                    // everybody's pieces of the next halo have landed (only the weight stage issued after them may still fly)
                    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LW) : "memory");
                }
                load_b(nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if CV_STAMP
        const unsigned long long st_d = __builtin_amdgcn_s_memtime();
        st_head += st_a - st_0; st_wait += st_b - st_a; st_tail += st_d - st_b; (void)st_c;
#endif
        ++s;
    };
    auto run_cb = [&](auto pb_tag, int cb) __attribute__((always_inline)) {             // PB = parity of the channel block = halo buffer = parity of its tap 0
        constexpr int PB = decltype(pb_tag)::value;
        typedef std::integral_constant<int, DBH ? PB : 0> HBt;
        Frags& E = PB ? F1 : F0;                         // fragments of even taps
        Frags& O = PB ? F0 : F1;
        stage(E, O, std::integral_constant<int, 0>{}, HBt{}, cb); stage(O, E, std::integral_constant<int, 1>{}, HBt{}, cb);
        stage(E, O, std::integral_constant<int, 2>{}, HBt{}, cb); stage(O, E, std::integral_constant<int, 3>{}, HBt{}, cb);
        stage(E, O, std::integral_constant<int, 4>{}, HBt{}, cb); stage(O, E, std::integral_constant<int, 5>{}, HBt{}, cb);
        stage(E, O, std::integral_constant<int, 6>{}, HBt{}, cb); stage(O, E, std::integral_constant<int, 7>{}, HBt{}, cb);
        stage(E, O, std::integral_constant<int, 8>{}, HBt{}, cb);
    };
    unsigned tile = lid;
    decode(tile);
    issue_prologue();
    chain_load_trunk();                                  // CHAIN 1: behind the halo / weight DMAs (the compiler waits for it where it is first used)
    if constexpr (FUSE0) {                               // the first weight stages fly while the first halo is produced
        load_patch0();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        produce0(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    for (bool first = true;; first = false) {
#pragma unroll
    for (int f = 0; f < FC; ++f)
#pragma unroll
        for (int g = 0; g < FP; ++g) acc[f][g] = f4{0.f, 0.f, 0.f, 0.f};
    if (first) {
        if (is_w) wait_vm_barrier<(NSW - 1) * LW>();    // halo(0) and W(0) landed; the later weight stages may still fly
        else wait_vm_barrier<0>();
    } else {
        wait_vm_barrier<0>();                            // issued a whole epilogue ago; also drains that epilogue's stores
    }
    wslot = 0;
    load_a(F0, std::integral_constant<int, 0>{}, 0);
    load_b(F0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    s = 0;
    for (int cb = 0; cb < nCb; cb += 2) {
        run_cb(std::integral_constant<int, 0>{}, cb);
        if (cb + 1 < nCb) run_cb(std::integral_constant<int, 1>{}, cb + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped repeats of the last weight stage are still landing

#if CV_STAMP
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    if (p.stamp && lane == 0) {
        unsigned long long* o = p.stamp + ((size_t)bid * NW + wave) * 8;
        o[0] = st_t1 - st_t0; o[1] = st_r1 - st_r0; o[2] = st_wait; o[3] = (unsigned long long)nS;
        o[4] = st_head; o[5] = st_t0 - st_k0; o[6] = st_tail;
    }
#endif
    if constexpr (CHAIN == 1) {                          // convolution 3 from the trunk registers: + block 0's output, ReLU -> f32 twin and f16 copy
        chain_epilogue(3);
        return;
    }
    // CHAIN == 2: convolution 3 leaves through the ordinary epilogue below -- the ring and the halo are dead now, so its LDS-staged
    // full-line stores and its residual burst apply; the host points scale / shift / res / res_mul at convolution 3's constants and at
    // block 0's f32 output (written by this workgroup's convolution-1 epilogue, complete long ago: 18 stages of counted waits lie between)
    // the epilogue works on the tile just finished; the DMA side moves on to the next one
    const int eCt = ctTile, eTx = tx, eTy = ty, eN = n;
    const unsigned nxt_tile = tile + nwg;
    const bool more_tiles = PERSIST && nxt_tile < nTiles;
    // ---- epilogue (see conv_igemm.hip for the rationale of the staged store) ------------------------------
    constexpr int NV = 4 * FC;
    const int row0 = eCt * CT + wci * 64 + q * NV;
    float sc[NV], sh[NV];
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const f4 a = *reinterpret_cast<const f4*>(p.scale + row0 + i);
        const f4 b = *reinterpret_cast<const f4*>(p.shift + row0 + i);
        sc[i] = a[0]; sc[i + 1] = a[1]; sc[i + 2] = a[2]; sc[i + 3] = a[3];
        sh[i] = b[0]; sh[i + 1] = b[1]; sh[i + 2] = b[2]; sh[i + 3] = b[3];
    }
    if (kSplits > 1) {
        // raw accumulators -> partial[split][pixel][channel]; a lane's 16 channels are 64 contiguous bytes
        float* const part = p.partial + (size_t)ksCur * p.M * p.prow + row0;
#pragma unroll
        for (int g = 0; g < FP; ++g) {
            const size_t pix = ((size_t)eN * p.Ho + eTy * TH + wrow0 + g) * p.Wo + eTx * 16 + l15;
#pragma unroll
            for (int f = 0; f < FC; ++f) *reinterpret_cast<f4*>(part + pix * p.prow + f * 4) = acc[f][g];
        }
        return;
    }
    T* const ybase = reinterpret_cast<T*>(p.y);
    const T* const rbase = reinterpret_cast<const T*>(p.res);
    // output pixel of (patch row wrow0 + g, lane l15): image, row origin, column
    const int oimg = IMG ? eN + (wrow0 >> 3) * 2 + (l15 >> 3) : eN;
    const int oy0 = IMG ? (wrow0 & 7) : eTy * TH + wrow0, ox = IMG ? (l15 & 7) : eTx * 16 + l15;

    if (!PERSIST && p.head_w) {                          // fused OutConv, as in conv_igemm.hip (64-channel tile only)
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
                const size_t pix = ((size_t)oimg * p.Ho + oy0 + g) * p.Wo + ox;
                const float l = part + p.head_b[0];
                p.head_logits[pix] = l;
                if (p.head_mask) p.head_mask[pix] = (1.f / (1.f + __expf(-l))) > p.head_thr ? 255 : 0;
                report_bad(p, l * 0.f);
            }
        }
#if CV_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.stamp && lane == 0) p.stamp[((size_t)bid * NW + wave) * 8 + 7] = __builtin_amdgcn_s_memtime() - st_t1;
#endif
        return;
    }

    constexpr int UN = __is_same(T, float) ? 4 : 8;
    constexpr int UPP = 64 / UN;
    constexpr int UPL = 16 * UPP / 64;
    constexpr int SROW = 272;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done reading the ring and the halo
#if CV_STAMP
    const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
    if (more_tiles) {                                    // next tile: halo(0) -> buffer 0, W(0..2) -> ring; lands during this epilogue
        decode(nxt_tile);
        if constexpr (DBH) issue_prologue();
        else issue_prologue_halo();                      // single halo buffer: the ring is this epilogue's staging area, the weights follow it
    }
    // Patch rows are staged RG at a time in wave-private LDS and read back so that consecutive lanes hold consecutive
    // bytes; output addresses are derived from the pixel index.  A persistent workgroup stages where the next tile's DMAs in
    // flight do not write, one row per wave: double-buffered halo -> halo buffer 1 (halo(0) and the first weight stages fly);
    // single halo buffer -> the weight ring (only the next halo flies; the weight stages are issued behind the staging).
    constexpr int RG = PERSIST ? 1 : 2;
    static_assert(FP % 2 == 0 && NW * RG * 16 * SROW <= (PERSIST ? (DBH ? HBYTES : NSW * WSTAGE) : NSW * WSTAGE + (DBH ? 2 : 1) * HBYTES), "staging must fit in LDS");
    char* const stg = (PERSIST && DBH ? halo + HBYTES : smem) + wave * (RG * 16 * SROW);
    const int slab0 = eCt * CT + wci * 64;
    T* const pbase = reinterpret_cast<T*>(p.pool_y);
    const bool relu_early = p.relu && !rbase;
    // output pixel index (in padded-plane pixels) of (patch row oy0 + g, pixel px of the 16-lane row)
    auto out_pixel = [&](int g, int px, bool* live) __attribute__((always_inline)) -> unsigned {
        if constexpr (IMG == 0) {
            *live = true;
            return (unsigned)((eN * p.yHp + oy0 + g + 1) * p.yWp + eTx * 16 + px + 1);
        } else {
            const int img = eN + (wrow0 >> 3) * 2 + (px >> 3);
            *live = img < nImg;
            return (unsigned)((img * p.yHp + oy0 + g + 1) * p.yWp + (px & 7) + 1);
        }
    };
    float hold[UPL][UN];                                 // RG == 1 only
    // Residual (ResNet shortcut): every unit this lane will add is fetched NOW, in one burst, so that the loads' latency passes
    // behind the staging of the first rows instead of being paid unit by unit inside the store loop (in-kernel stamps, r02: the
    // epilogue of a residual layer took 26 k cycles against 11 k without one -- as long as the 18-stage K loop itself).
    constexpr bool kHalf = __is_same(T, half_t);         // f16 kernels can take the shortcut from the trunk's f32 twin (two chunks per unit)
    constexpr int RC = kHalf ? 2 : OutVec<T, UN>::kRawChunks;
    constexpr bool kPrefetchRes = !PERSIST && NW == 4;   // the 4-wave tile has the registers (FP x UPL units x RC chunks)
    const bool res32 = kHalf && p.res_f32;               // wave-uniform
    const float* const rbase32 = reinterpret_cast<const float*>(p.res);
    float* const ybase32 = kHalf ? reinterpret_cast<float*>(p.y32) : nullptr;
    f4 rraw[kPrefetchRes ? FP : 1][UPL][RC];
    if constexpr (kPrefetchRes) {
        if (rbase) {
#pragma unroll
            for (int g = 0; g < FP; ++g)
#pragma unroll
                for (int i = 0; i < UPL; ++i) {
                    const int unit = lane + 64 * i, px = unit / UPP, co = slab0 + (unit % UPP) * UN;
                    bool live;
                    const unsigned ob = out_pixel(g, px, &live);
                    if (co < p.rows && live) {
                        if (res32) trunk32_fetch(rbase32 + (ob * (unsigned)p.rCs + (unsigned)(p.rCoff + co)), rraw[g][i]);
                        else OutVec<T, UN>::fetch(rbase + (ob * (unsigned)p.rCs + (unsigned)(p.rCoff + co)), rraw[g][i]);
                    } else {
#pragma unroll
                        for (int c = 0; c < RC; ++c) rraw[g][i][c] = f4{0.f, 0.f, 0.f, 0.f};
                    }
                }
        }
    }
    float bad = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < FP; g0 += RG) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int f = 0; f < FC; ++f) {
                f4 t;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[k] = acc[f][g0 + r][k] * sc[f * 4 + k] + sh[f * 4 + k];
                    // ReLU on the FMA result (known canonical: one v_max); after the read-back the compiler has to
                    // canonicalise first.  With a residual the ReLU follows the add below.
                    if (relu_early) t[k] = __builtin_fmaxf(t[k], 0.f);
                }
                *reinterpret_cast<f4*>(stg + r * (16 * SROW) + l15 * SROW + (q * NV + f * 4) * 4) = t;
            }
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int unit = lane + 64 * i;
            const int px = unit / UPP, cu = unit % UPP;
            const int co = slab0 + cu * UN;
            float w[RG][UN];
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int j = 0; j < UN; j += 4) {
                    const f4 t = *reinterpret_cast<const f4*>(stg + r * (16 * SROW) + px * SROW + (cu * UN + j) * 4);
                    w[r][j] = t[0]; w[r][j + 1] = t[1]; w[r][j + 2] = t[2]; w[r][j + 3] = t[3];
                }
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                bool live;
                const unsigned ob = out_pixel(g0 + r, px, &live);
                if (co < p.rows && live) {
                    if (rbase) {                         // tensors stay below 4 GiB (checked on the host): 32-bit element offsets
                        if (res32) {
                            if constexpr (kHalf) {
                                if constexpr (kPrefetchRes) trunk32_add_raw(rraw[g0 + r][i], w[r], p.res_mul);
                                else {
                                    f4 raw[2];
                                    trunk32_fetch(rbase32 + (ob * (unsigned)p.rCs + (unsigned)(p.rCoff + co)), raw);
                                    trunk32_add_raw(raw, w[r], p.res_mul);
                                }
                            }
                        } else if constexpr (kPrefetchRes) OutVec<T, UN>::add_raw(rraw[g0 + r][i], p.rCoff + co, w[r], p.res_mul);
                        else OutVec<T, UN>::add(rbase + (ob * (unsigned)p.rCs + (unsigned)(p.rCoff + co)), p.rCoff + co, w[r], p.res_mul);
                        if (p.relu) {
#pragma unroll
                            for (int j = 0; j < UN; ++j) w[r][j] = __builtin_fmaxf(w[r][j], 0.f);
                        }
                    }
                    OutVec<T, UN>::store(ybase + (ob * (unsigned)p.yCs + (unsigned)(p.yCoff + co)), p.yCoff + co, w[r], bad);
                    if constexpr (kHalf) {
                        if (ybase32) trunk32_store(ybase32 + (ob * (unsigned)p.yCs + (unsigned)(p.yCoff + co)), w[r]);
                    }
                }
            }
            if (IMG == 0 && pbase) {                     // wave-uniform: fused 2x2 max-pool of the (post-ReLU) row pair
                // The horizontal partner pixel is read from the staged row itself (values there are already ReLU'd:
                // pooled layers carry no residual).  Lane shuffles would be eight dependent LDS round trips per unit
                // and doubled this epilogue (r01_tuning.md step 25).
                float m[UN];
#pragma unroll
                for (int j = 0; j < UN; j += 4) {
                    f4 best = *reinterpret_cast<const f4*>(stg + (px ^ 1) * SROW + (cu * UN + j) * 4);
#pragma unroll
                    for (int r = 1; r < RG; ++r) {
                        const f4 t = *reinterpret_cast<const f4*>(stg + r * (16 * SROW) + (px ^ 1) * SROW + (cu * UN + j) * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) best[k] = best[k] > t[k] ? best[k] : t[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = w[0][j + k];
#pragma unroll
                        for (int r = 1; r < RG; ++r) v = v > w[r][j + k] ? v : w[r][j + k];
                        m[j + k] = v > best[k] ? v : best[k];
                    }
                }
                bool emit = true;
                if constexpr (RG == 1) {                 // one row per pass: the upper row of a pair waits in registers
                    if ((g0 & 1) == 0) {
                        emit = false;
#pragma unroll
                        for (int j = 0; j < UN; ++j) hold[i][j] = m[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < UN; ++j) m[j] = m[j] > hold[i][j] ? m[j] : hold[i][j];
                    }
                }
                const unsigned qb = (unsigned)((eN * p.pHp + ((oy0 + g0) >> 1) + 1) * p.pWp + ((eTx * 16 + px) >> 1) + 1);
                if constexpr (kSplit16) {                // both pixels of a pair hold the maximum: one stores hi, the other lo
                    if (emit && co < p.rows)
                        OutVec<T, UN>::store_half(pbase + (qb * (unsigned)p.pCs + (unsigned)(p.pCoff + co)), p.pCoff + co, m, (px & 1) == 0);
                } else {
                    if (emit && (px & 1) == 0 && co < p.rows)
                        OutVec<T, UN>::store(pbase + (qb * (unsigned)p.pCs + (unsigned)(p.pCoff + co)), p.pCoff + co, m, bad);
                }
            }
        }
        wave_lds_sync();
    }
#if CV_STAMP
    const unsigned long long st_e1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.stamp && lane == 0) {
        unsigned long long* o = p.stamp + ((size_t)bid * NW + wave) * 8;
        o[7] = __builtin_amdgcn_s_memtime() - st_t1;     // epilogue until the stores have drained
        o[4] = st_e0 - st_t1;                            // ... until every wave has left the K loop
        o[6] = st_e1 - st_t1;                            // ... until the last store is issued
    }
#endif
    report_bad(p, bad);
    if (!more_tiles) break;
    if constexpr (PERSIST && !DBH) {                     // every wave has read its staged rows back: the ring is free for the next tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        issue_prologue_w();
    }
    tile = nxt_tile;
    }
}

// ---- host-side launch -------------------------------------------------------------------------------------
// the 64-channel tile over a 16 x 16 patch keeps a single halo buffer (see the kernel: DBH)
template <int CT, int TH> static constexpr bool halo_double() { return !(CT == 64 && (TH == 16 || (TH == 8 && CV_HALO_TH8_SINGLE))); }

template <int CT, int TH, int NW, int TPS, int NSW, int IMG>
static constexpr size_t halo_lds() {
    constexpr int HR = IMG ? 4 * (IMG + 2) * (IMG + 2) : 18 * (TH + 2);
    constexpr int NWI = NW == 8 ? NW / 2 : NW;          // as in the kernel: waves per DMA role
    constexpr int H = ((HR + 7) / 8 + NWI - 1) / NWI;
    return (size_t)NSW * TPS * CT * 128 + (size_t)(halo_double<CT, TH>() ? 2 : 1) * H * NWI * 1024;
}

static int g_halo_cus = 256;                            // CUs of the device (set by conv_halo_prepare)

// 8-wave tiles (one 136-152 KB workgroup per CU) run persistent; CV_HALO_PERSIST=0 launches one workgroup per tile
template <int NW> static bool halo_persistent() {
    static const bool on = [] { const char* v = std::getenv("CV_HALO_PERSIST"); return !(v && v[0] == '0'); }();
    return NW == 8 && on;
}

constexpr size_t kFuse0Lds = 4864 + 4096 + 512;         // [3][20][20] f32 input patch + zero slot | block-1 weight fragments | BN constants
template <typename T, int CT, int TH> static constexpr bool halo_can_fuse0() { return CT == 64 && TH == 16 && __is_same(T, split_t); }

template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW, int IMG>
static hipError_t launch_halo(const ConvParams& p, int n_images, hipStream_t stream) {
    const bool split = p.ksplit > 1;
    if (split && (IMG != 0 || p.f0_x || !p.partial || p.kper < 1)) return hipErrorInvalidValue;
    const int tiles = (IMG ? (n_images + 3) / 4 : n_images * (p.Ho / TH) * (p.Wo / 16)) * p.nCt * (split ? p.ksplit : 1);
    const size_t lds = halo_lds<CT, TH, NW, TPS, NSW, IMG>();
    if (p.f0_x) {
        if constexpr (halo_can_fuse0<T, CT, TH>() && IMG == 0) {
            auto kern = conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, false, false, true>;
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds + kFuse0Lds, stream, p);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    static_assert(halo_lds<CT, TH, NW, TPS, NSW, IMG>() <= 160 * 1024, "LDS budget");
    // persistent only where it pays (same-box A/B, r01_tuning.md step 22): several tiles per CU and a K loop short enough
    // for the hidden prologue to matter; long loops lose ~1 % to the single-row staging
    static const int max_k = [] { const char* v = std::getenv("CV_HALO_PERSIST_MAXK"); return v && *v ? std::atoi(v) : 72; }();
    bool launched = false;
    if constexpr (CT == 64 && NW == 4 && IMG == 0 && !halo_double<CT, TH>()) {
        // the production tile, persistent (two resident workgroups per CU walk the tiles; the next tile's halo flies during the
        // epilogue): CV_HALO_PERSIST64=1, launches of short K with many tiles only
        static const int on64 = [] { const char* v = std::getenv("CV_HALO_PERSIST64"); return v && *v ? std::atoi(v) : 0; }();
        static const int max_k64 = [] { const char* v = std::getenv("CV_HALO_PERSIST64_MAXK"); return v && *v ? std::atoi(v) : 36; }();
        if (on64 && !split && !p.head_w && !p.res && tiles >= 8 * g_halo_cus && p.nStages <= max_k64) {
            auto kern = conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, true, false>;
            hipLaunchKernelGGL(kern, dim3((unsigned)(2 * g_halo_cus)), dim3(64 * NW), lds, stream, p);
            launched = true;
        }
    }
    if constexpr (NW == 8) {
        if (halo_persistent<NW>() && !split && tiles >= 4 * g_halo_cus && p.nStages <= max_k) {
            auto kern = conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, true, true>;
            const int grid = tiles < g_halo_cus ? tiles : g_halo_cus;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds, stream, p);
            launched = true;
        }
    }
    if (!launched) {
        auto kern = conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, false, halo_double<CT, TH>()>;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * NW), lds, stream, p);
    }
    return hipGetLastError();
}

template <typename T, int CT, int TH, int WGC, int NW, int TPS, int NSW, int IMG>
static hipError_t prepare_halo() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, false, halo_double<CT, TH>()>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (halo_can_fuse0<T, CT, TH>() && IMG == 0) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if constexpr (CT == 64 && NW == 4 && IMG == 0 && !halo_double<CT, TH>()) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, true, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if constexpr (NW == 8) {                             // persistent variants of the 8-wave (double-buffered) tiles
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<T, CT, TH, WGC, NW, TPS, NSW, IMG, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    return e;
}

// configurations: 64 channels x 16x16 patch (4 waves, four patch rows each), one tap per stage, ring 3, ONE halo buffer: 68 KB,
//                 so two workgroups share a CU; each fills the other's barrier stalls and its halo refill at the channel-block
//                 boundaries (r02_tuning.md: +8-16 % over the double-buffered 8x16 patch of round 1, which did half the MFMAs per
//                 weight byte and per fragment read; -DCV_HALO_TH64=8 builds that tile).  THE production tile: layers of any
//                 width run it as Cout/64 channel tiles (engine.cpp: choose_ct), with the fused first-layer producer for
//                 UNet inc (FUSE0) and over four packed 8x8 images for ResNet-18 layer2;
//                 128 channels x 16x16 patch (8 waves as 2 x 4, four patch rows each, halo double buffered, persistent for short
//                 K) and its packed-image form: round 1's tile for Cout >= 128, now an A/B option (CV_CT64_MAXROWS=64,
//                 CV_HALO_IMG8_64=0) -- one 136 KB workgroup per CU loses to two 68 KB ones at every width (r02_tuning.md step 14).
// (Measured and dropped: an 8-wave 64 x 16x16 variant, -10 % (r01_tuning.md step 17); the 16x16 patch with a double-buffered halo
//  and therefore one workgroup per CU, -8..-17 %; a 4-deep weight ring, +-1 % (r02_tuning.md).)
#ifndef CV_HALO_NSW64
#define CV_HALO_NSW64 3
#endif
#ifndef CV_HALO_NSW128
#define CV_HALO_NSW128 3
#endif
#ifndef CV_HALO_TH64
#define CV_HALO_TH64 16                       // patch rows of the 64-channel tile: 16 (single halo buffer, 68 KB) | 8 (double buffered, 72 KB)
#endif
#define CV_FOR_EACH_HALO_MAIN(X, T)           \
    X(T, 64, CV_HALO_TH64, 1, 4, 1, CV_HALO_NSW64, 0)    \
    X(T, 128, 16, 2, 8, 1, CV_HALO_NSW128, 0) \
    X(T, 128, 16, 2, 8, 1, 3, 8)              \
    X(T, 64, 16, 1, 4, 1, 3, 8)
// round 4: the 64-channel tile over an 8 x 16 patch (double-buffered halo, 72 KB, two workgroups per CU -- round 1's tile) next to the
// production 16 x 16 one, for launches whose 16 x 16 patches number fewer than the chip can hold (single boards: 128 tiles on 256
// CUs): twice the workgroups for the same work
#if CV_HALO_TH64 == 16
#define CV_FOR_EACH_HALO(X, T) CV_FOR_EACH_HALO_MAIN(X, T) X(T, 64, 8, 1, 4, 1, 3, 0)
#else
#define CV_FOR_EACH_HALO(X, T) CV_FOR_EACH_HALO_MAIN(X, T)
#endif

hipError_t conv_halo_prepare() {
    hipError_t e;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        g_halo_cus = cus;
#define X(T, CT, TH, WGC, NW, TPS, NSW, IMG) \
    if ((e = prepare_halo<T, CT, TH, WGC, NW, TPS, NSW, IMG>()) != hipSuccess) return e;
    CV_FOR_EACH_HALO(X, half_t)
    CV_FOR_EACH_HALO(X, float)
    CV_FOR_EACH_HALO(X, split_t)
#undef X
#if CV_HALO_TH64 == 16 && CV_HALO_NSW64 == 3
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<half_t, 64, 16, 1, 4, 1, 3, 0, false, false, false, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<half_t, 64, 16, 1, 4, 1, 3, 0, false, false, false, 2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
#endif
    return hipSuccess;
}

// The 64-row tile (4 waves, two workgroups per CU) beats conv_igemm's 64x256 tile on the 64-channel layers (r01_tuning.md step
// 16, r02_tuning.md); CV_HALO64=0 switches it off for A/B runs, CV_HALO_IMG8=0 the packed-image mode.
bool conv_halo_can_fuse_first_layer(int ct, int dt) { return ct == 64 && CV_HALO_TH64 == 16 && dt == kSplit; }

bool conv_halo_supported(int ct, int Ho, int Wo) {
    static const bool allow64 = [] { const char* v = std::getenv("CV_HALO64"); return !(v && v[0] == '0'); }();
    static const bool allow_img8 = [] { const char* v = std::getenv("CV_HALO_IMG8"); return !(v && v[0] == '0'); }();
    static const bool img8_64 = [] { const char* v = std::getenv("CV_HALO_IMG8_64"); return !(v && v[0] == '0'); }();
    if (ct == 128 && Ho == 8 && Wo == 8) return allow_img8;
    if (ct == 64 && Ho == 8 && Wo == 8) return allow_img8 && img8_64 && CV_HALO_TH64 == 16;
    return (ct == 128 || (ct == 64 && allow64)) && Ho % 16 == 0 && Wo % 16 == 0;
}

bool conv_halo_has_th8(int ct) { return ct == 64 && CV_HALO_TH64 == 16; }

// four chained 64 -> 64 convolutions on whole 16 x 16 images (ConvParams::chain): one workgroup per image, 69.6 KB of LDS
bool conv_halo_has_chain() { return CV_HALO_TH64 == 16 && CV_HALO_NSW64 == 3; }
hipError_t conv_halo_chain_launch(const ConvParams& p, int n_images, hipStream_t stream) {
#if CV_HALO_TH64 == 16 && CV_HALO_NSW64 == 3
    if (!p.chain || p.nStages != 36 || p.Ho != 16 || p.Wo != 16 || p.yCs != 64 || p.xCs != 64 || p.ksplit > 1 || n_images < 1) return hipErrorInvalidValue;
    const size_t lds = halo_lds<64, 16, 4, 1, 3, 0>();
    if ((p.chain & 3) == 1) {                            // one workgroup per CU (its 512 registers per lane see to that)
        auto kern = conv3x3_halo_kernel<half_t, 64, 16, 1, 4, 1, 3, 0, false, false, false, 1>;
        hipLaunchKernelGGL(kern, dim3((unsigned)n_images), dim3(256), lds, stream, p);
    } else {
        auto kern = conv3x3_halo_kernel<half_t, 64, 16, 1, 4, 1, 3, 0, false, false, false, 2>;
        hipLaunchKernelGGL(kern, dim3((unsigned)n_images), dim3(256), lds, stream, p);
    }
    return hipGetLastError();
#else
    return hipErrorInvalidValue;
#endif
}

hipError_t conv_halo_launch(int ct, int dt, const ConvParams& p, int n_images, hipStream_t stream, int th) {
    const int img = (p.Ho == 8 && p.Wo == 8) ? 8 : 0;
    if (img || !(th == 8 && conv_halo_has_th8(ct))) th = ct == 64 ? CV_HALO_TH64 : 16;
#define X(T, CT, TH, WGC, NW, TPS, NSW, IMG) \
    if (ct == CT && img == IMG && (IMG != 0 || th == TH)) return launch_halo<T, CT, TH, WGC, NW, TPS, NSW, IMG>(p, n_images, stream);
    if (dt == kF16) { CV_FOR_EACH_HALO(X, half_t) } else if (dt == kSplit) { CV_FOR_EACH_HALO(X, split_t) } else { CV_FOR_EACH_HALO(X, float) }
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cv
