// conv_igemm_body.h -- the body of conv_igemm_kernel / conv_igemm_pair_kernel (conv_igemm.hip), included TEXTUALLY inside both.
// Expects in scope: the template parameters T, CT, PT, WGC, NS, NW, SEP and the constant POS (position-major rows, ConvParams::ptab);
// `p` (the layer's ConvParams); `nwg` / `bid` (unsigned: the number of workgroups that run this layer and this workgroup's index
// among them).  A function taking ConvParams by reference or by
// value instead changes the register allocation of the production kernels (152 -> 156, 228 -> 232, 254 -> 255 VGPRs, measured on the
// ISA): the textual form keeps conv_igemm_kernel byte-identical to what it was before the pair kernel existed.
    constexpr int WGP = NW / WGC;
    constexpr int WCT = CT / WGC, WPT = PT / WGP;
    constexpr int FC = WCT / 16, FP = WPT / 16;
    static_assert(FC == kConvFC, "host weight packing assumes 64-channel wave slabs");
    constexpr int STAGE = (CT + PT) * 128;
    constexpr int LW = CT / (8 * NW), LX = PT / (8 * NW);   // DMA wave-instructions per wave per stage
    constexpr int L = LW + LX;
    typedef typename FragT<T>::V V;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* koffs = reinterpret_cast<int*>(smem + NS * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Remap so each XCD walks a
    // contiguous range of tiles: vertically adjacent pixel tiles, which read the same input rows, then share
    // one L2 instead of each pulling their own copy over the fabric.  Bijective for any grid size; affects
    // speed only.
    const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    // split-K launches: the pixel tiles of one (channel tile, K split) are neighbours -- they stream the SAME weight slab (the large
    // operand of the deep layers these launches serve) and, after the XCD remap above, through the same L2
    int ctTile, ptTile, sBase = 0, nS = p.nStages;
    if (!POS && p.ksplit > 1) {
        const int nPt = (p.M + PT - 1) / PT;
        ptTile = lid % nPt;
        const int t = lid / nPt;
        ctTile = t % p.nCt;
        const int ks = t / p.nCt;
        sBase = ks * p.kper;
        nS = min(p.kper, p.nStages - sBase);
    } else {
        ctTile = lid % p.nCt;
        ptTile = lid / p.nCt;
    }
    const int HoWo = p.Ho * p.Wo;
    // POS: workgroups differ in K length (a corner position of a 4x4 map walks 4 taps, an interior one 9) and a CU holds ONE of these
    // tiles at a time, a launch being 2-4 rounds deep -- so the walk is longest-first inside every XCD: with the columns (image tile,
    // channel tile) dealt round-robin over the XCDs, XCD x (workgroups x, x + 8, ...: dispatched in that order) runs all columns of its
    // heaviest position class first, the corners last, and the tail of the launch is a short tile instead of a long one (measured
    // before this order: layer3 0.87 of the unsplit time with 0.69 of the stages; f32, which is MFMA-bound: 0.98).  Workgroups that run
    // side by side on an XCD are then the same-class positions of the same images: they share input lines (L2) and the weight stages
    // of their position, and they stay in step because their K loops are equally long.
    int pos = 0, tImg = 0;
    if constexpr (POS) {
        static_assert(SEP, "position-major launches use the table-free gather offsets");
        const unsigned nCol = (unsigned)(p.nPtPer * p.nCt), nColX = (nCol + 7) >> 3;    // columns per XCD (grid = 8 * nColX * HoWo)
        const unsigned j = bid >> 3, cl = j % nColX, col = (bid & 7) + 8 * cl;
        if (col >= nCol) return;                                                          // padding of the last column group
        ctTile = col / p.nPtPer;
        tImg = col - ctTile * p.nPtPer;
        pos = __builtin_amdgcn_readfirstlane(p.porder[j / nColX]);
        ptTile = tImg;
        nS = p.pcount[pos];
    }
    const int posY = POS ? pos / p.Wo : 0, posX = POS ? pos - posY * p.Wo : 0;

    if constexpr (!SEP)
        for (int i = tid; i < nS * 8; i += 64 * NW) koffs[i] = p.koff[sBase * 8 + i];

    // DMA source of this lane's activation rows: row r of the pixel tile <-> output pixel ptTile*PT + r
    unsigned xoff[LX];
#pragma unroll
    for (int i = 0; i < LX; ++i) {
        int pix = ptTile * PT + (i * NW + wave) * 8 + (lane >> 3);
        int n, oy, ox;
        if constexpr (POS) {
            n = pix < p.posN ? pix : p.posN - 1;
            oy = posY; ox = posX;
        } else {
            pix = pix < p.M ? pix : p.M - 1;
            n = pix / HoWo;
            const int rem = pix - n * HoWo;
            oy = rem / p.Wo;
            ox = rem - oy * p.Wo;
        }
        xoff[i] = (unsigned)((n * p.xHp + oy * p.stride) * p.xWp + ox * p.stride) * (unsigned)p.xCs *
                  (unsigned)sizeof(T);
    }
    const int myChunk = (lane & 7) ^ (lane >> 3);      // logical K chunk this lane fetches (swizzled)
    const char* wsrc = p.w + ((size_t)ctTile * p.nStages + sBase) * (CT * 128) + wave * 1024 + lane * 16;
    __syncthreads();                                    // koffs visible

    auto issue = [&](int s, int buf) {
        char* sW = smem + buf * STAGE;
        char* sX = sW + CT * 128;
        int ws = s, ko;
        if constexpr (POS) {
            // {weight stage, gather base} of this position's s-th live stage: one 8-byte load through the scalar cache
            typedef int i2_t __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(4))) i2_t* c2ptr_t;
            const i2_t e = reinterpret_cast<c2ptr_t>(reinterpret_cast<uintptr_t>(p.ptab))[pos * p.nStages + s];
            ws = e[0];
            ko = e[1] + myChunk * 16;
        }
        const char* gw = wsrc + (size_t)ws * (CT * 128);
#pragma unroll
        for (int i = 0; i < LW; ++i) glds16(gw + i * (NW * 1024), sW + (i * NW + wave) * 1024);
        if constexpr (POS) {
        } else if constexpr (SEP) {
            // constant address space => s_load through the scalar cache (a VGPR load here would make the compiler
            // drain vmcnt, i.e. the whole DMA ring, every stage)
            typedef const __attribute__((address_space(4))) int* cptr_t;
            ko = reinterpret_cast<cptr_t>(reinterpret_cast<uintptr_t>(p.kbase))[sBase + s] + myChunk * 16;
        } else {
            ko = koffs[s * 8 + myChunk];
        }
#pragma unroll
        for (int i = 0; i < LX; ++i) glds16(p.x + xoff[i] + ko, sX + (i * NW + wave) * 1024);
    };

    const int wci = wave / WGP, wpi = wave % WGP;
    const int q = lane >> 4, l15 = lane & 15;
    const int rowW = (wci * WCT + l15) * 128;
    const int rowX = CT * 128 + (wpi * WPT + l15) * 128;

    f4 acc[FC][FP];
#pragma unroll
    for (int f = 0; f < FC; ++f)
#pragma unroll
        for (int g = 0; g < FP; ++g) acc[f][g] = f4{0.f, 0.f, 0.f, 0.f};

    constexpr bool kSplit16 = sizeof(T) == 4 && !__is_same(T, float);
    static_assert(NS == 2 || NS == 3, "ring depth 2 or 3");
    // Two K loops, chosen per launch (wave-uniform): the register-rotating pipeline for the long loops of the f16-MFMA
    // dtypes, the plain barrier -> DMA -> reads -> MFMA loop for short K (1-tap GEMMs, the 3-channel first layer:
    // the pipeline's longer prologue costs 4-9 % there) and for f32 (32-cycle MFMAs already cover the bubble; the
    // per-fragment refill order serialises dependent f32 MFMAs: -15 % on layer2, r01_tuning.md step 18).
    // ---- main loop: register-rotating software pipeline -----------------------------------------------------
    // The fragments of stage s are in registers when the stage starts (read from LDS during stage s-1), and the MFMAs
    // run channel-fragment-major: block f = all products of weight fragment f.  A fragment's registers are refilled
    // with stage s+1 as soon as its last MFMA has been issued -- weight fragment f right after block f, pixel fragment
    // g inside the last block -- so nothing is double buffered (the 256x256 tile has no registers for that) and
    // every LDS read has at least a block of MFMAs between issue and use.  The one barrier of the stage sits behind
    // block 0: it publishes stage s+1 (needed from block 0's refill on) and frees the ring slot of stage s, which
    // nobody reads during stage s, for the DMA of stage s+NS.  The matrix pipe therefore always has queued work on
    // both sides of the rendezvous.  (Before r01 step 18 every wave did barrier -> DMA issue -> reads -> MFMAs.)
    // Per stage and operand two 16-byte chunks per lane: split-f16 the hi and lo chunk of its channel group (products
    // hi.hi, lo.hi, hi.lo), f16 / f32 the two k-halves of the line (products set0.set0, set1.set1).
    constexpr int FPH = FP > 4 ? 4 : FP;               // pixel fragments per pass (FP = 8: two passes, bounds VGPR use)
    constexpr int NP = FP / FPH;
    struct Frags { V a[2][FC]; V b[2][FPH]; } F;
    const int c0 = kSplit16 ? 2 * q + (q & 1) : q;
    const int c1 = kSplit16 ? 2 * q + 1 - (q & 1) : 4 + q;
    const int swz0 = (c0 ^ (lane & 7)) << 4, swz1 = (c1 ^ (lane & 7)) << 4;
    auto load_a = [&](int f, int so) {
        F.a[0][f] = *reinterpret_cast<const V*>(smem + (so + rowW + swz0) + f * 2048);
        F.a[1][f] = *reinterpret_cast<const V*>(smem + (so + rowW + swz1) + f * 2048);
    };
    auto load_b = [&](int g, int gsrc, int so) {        // register slot g <- pixel fragment gsrc of the stage in ring offset so
        F.b[0][g] = *reinterpret_cast<const V*>(smem + (so + rowX + swz0) + gsrc * 2048);
        F.b[1][g] = *reinterpret_cast<const V*>(smem + (so + rowX + swz1) + gsrc * 2048);
    };
    auto mma_fg = [&](int f, int g, int gacc) {
        mma16(acc[f][gacc], F.a[0][f], F.b[0][g]);
        if constexpr (kSplit16) {
            mma16(acc[f][gacc], F.a[1][f], F.b[0][g]);
            mma16(acc[f][gacc], F.a[0][f], F.b[1][g]);
        } else {
            mma16(acc[f][gacc], F.a[1][f], F.b[1][g]);
        }
    };
    auto block = [&](int f, int ps) {
#pragma unroll
        for (int g = 0; g < FPH; ++g) mma_fg(f, g, ps * FPH + g);
    };
    constexpr bool kPipeType = CV_IGEMM_PIPE != 0 && !__is_same(T, float);
    if (kPipeType && nS >= 8) {
    issue(0, 0);
    if (nS > 1) issue(1, 1);
    if (NS == 3 && nS > 2) issue(2, 2);
    if (NS == 3 && nS > 2) wait_vm_barrier<2 * L>();
    else if (nS > 1) wait_vm_barrier<L>();
    else wait_vm_barrier<0>();
#pragma unroll
    for (int f = 0; f < FC; ++f) load_a(f, 0);
#pragma unroll
    for (int g = 0; g < FPH; ++g) load_b(g, g, 0);
    int slotS = 0, slotN = nS > 1 ? 1 : 0;             // ring slots of stage s and stage s+1
    // one stage; MORE = a next stage exists (its data is published by this stage's barrier and refills the fragments)
    auto stage = [&](auto more_tag, int s) {
        constexpr bool MORE = decltype(more_tag)::value;
        const int soS = slotS * STAGE, soN = slotN * STAGE;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const bool lastp = ps == NP - 1;
            block(0, ps);
            if (lastp && MORE) {
                __builtin_amdgcn_sched_barrier(0);
                if (NS == 3 && s + 2 < nS) wait_vm_barrier<L>(); else wait_vm_barrier<0>();
                __builtin_amdgcn_sched_barrier(0);
                if (s + NS < nS) issue(s + NS, slotS);
                load_a(0, soN);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int f = 1; f < FC - 1; ++f) {
                block(f, ps);
                if (lastp && MORE) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_a(f, soN);
                }
            }
#pragma unroll
            for (int g = 0; g < FPH; ++g) {
                mma_fg(FC - 1, g, ps * FPH + g);
                if (!lastp) {                            // next pass of this stage: its pixel fragments, same ring slot
                    __builtin_amdgcn_sched_barrier(0);  // refill in place: no hoisting over the fragment's last use
                    load_b(g, (ps + 1) * FPH + g, soS);
                } else if (MORE) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_b(g, g, soN);
                }
            }
            if (lastp && MORE) {
                load_a(FC - 1, soN);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s + 1 < nS; ++s) {
        stage(std::true_type{}, s);
        slotS = slotN;
        slotN = slotN == NS - 1 ? 0 : slotN + 1;
    }
    stage(std::false_type{}, nS - 1);
    } else {
        // plain loop: barrier -> prefetch of a later stage -> fragment reads -> MFMAs
        auto compute = [&](int so) {
#pragma unroll
            for (int f = 0; f < FC; ++f) load_a(f, so);
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
#pragma unroll
                for (int g = 0; g < FPH; ++g) load_b(g, ps * FPH + g, so);
                if (kSetPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int f = 0; f < FC; ++f)             // first the products that need only the first chunk of each row
#pragma unroll
                    for (int g = 0; g < FPH; ++g) mma16(acc[f][ps * FPH + g], F.a[0][f], F.b[0][g]);
#pragma unroll
                for (int f = 0; f < FC; ++f)
#pragma unroll
                    for (int g = 0; g < FPH; ++g) {
                        if constexpr (kSplit16) {
                            mma16(acc[f][ps * FPH + g], F.a[1][f], F.b[0][g]);
                            mma16(acc[f][ps * FPH + g], F.a[0][f], F.b[1][g]);
                        } else {
                            mma16(acc[f][ps * FPH + g], F.a[1][f], F.b[1][g]);
                        }
                    }
                if (kSetPrio) __builtin_amdgcn_s_setprio(0);
            }
        };
        if constexpr (NS == 2) {
            issue(0, 0);
            int buf = 0;
            for (int t = 0; t < nS; ++t) {
                wait_vm_barrier<0>();
                if (CV_ABLATE != 1 && t + 1 < nS) issue(t + 1, buf ^ 1);
                compute(buf * STAGE);
                buf ^= 1;
            }
        } else {
            issue(0, 0);
            if (nS > 1) issue(1, 1);
            int bufC = 0, bufI = 2;
            for (int t = 0; t < nS; ++t) {
                if (t + 1 < nS) wait_vm_barrier<L>(); else wait_vm_barrier<0>();
                if (CV_ABLATE != 1 && t + 2 < nS) issue(t + 2, bufI);
                compute(bufC * STAGE);
                bufC = bufC == 2 ? 0 : bufC + 1;
                bufI = bufI == 2 ? 0 : bufI + 1;
            }
        }
    }

    // ---- epilogue: BN affine (+ residual) (+ ReLU), convert, 16-B NHWC stores -------------------------
    constexpr int NV = 4 * FC;                          // consecutive channels held by this lane
    const int row0 = ctTile * CT + wci * WCT + q * NV;  // first GEMM row (== channel, by host permutation)
    if (!POS && p.ksplit > 1) {                         // (position-major launches are never split, nor do they feed the fused head)
        // split-K: raw accumulators -> partial[split][pixel][channel]; a lane's 16 channels are 64 contiguous bytes, the four
        // lane groups of a pixel cover 256.  conv_splitk_reduce_kernel finishes the layer.
        float* const part = p.partial + (size_t)(sBase / p.kper) * p.M * p.prow + row0;
#pragma unroll
        for (int g = 0; g < FP; ++g) {
            const int pix = ptTile * PT + wpi * WPT + g * 16 + l15;
            if (pix < p.M) {
#pragma unroll
                for (int f = 0; f < FC; ++f) *reinterpret_cast<f4*>(part + (size_t)pix * p.prow + f * 4) = acc[f][g];
            }
        }
        return;
    }
    float sc[NV], sh[NV];
#pragma unroll
    for (int i = 0; i < NV; i += 4) {
        const f4 a = *reinterpret_cast<const f4*>(p.scale + row0 + i);
        const f4 b = *reinterpret_cast<const f4*>(p.shift + row0 + i);
        sc[i] = a[0]; sc[i + 1] = a[1]; sc[i + 2] = a[2]; sc[i + 3] = a[3];
        sh[i] = b[0]; sh[i + 1] = b[1]; sh[i + 2] = b[2]; sh[i + 3] = b[3];
    }
    int co0 = row0, dy = 0, dx = 0;
    if (p.shuffle) {                                     // rows are (dy, dx, co): k2 s2 transposed conv
        const int g = row0 / p.Cout;
        co0 = row0 - g * p.Cout;
        dy = g >> 1;
        dx = g & 1;
    }
    T* const ybase = reinterpret_cast<T*>(p.y);
    const T* const rbase = reinterpret_cast<const T*>(p.res);

    if (!POS && p.head_w) {
        // fused OutConv (UNet outc): 4 lanes (q = 0..3) hold a pixel's 64 channels; nothing is stored but the logit
        float hw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) hw[i] = p.head_w[row0 + i];
#pragma unroll
        for (int g = 0; g < FP; ++g) {
            const int pix = ptTile * PT + wpi * WPT + g * 16 + l15;
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
            if (pix < p.M && q == 0) {
                const float l = part + p.head_b[0];
                p.head_logits[pix] = l;
                if (p.head_mask) p.head_mask[pix] = (1.f / (1.f + __expf(-l))) > p.head_thr ? 255 : 0;
                report_bad(p, l * 0.f);
            }
        }
        return;
    }

    // Staged store: the MFMA leaves each lane with 16 channels of one pixel, i.e. 64 scattered 16-byte pieces per
    // store instruction.  Each wave instead parks one 16-pixel x 64-channel fragment (f32, rows padded to 272 B so
    // the b128 writes spread over all banks) in its own corner of the now idle LDS ring and reads it back so that
    // consecutive lanes hold consecutive bytes: a quarter-wave then writes (and reads the residual of) one pixel's
    // whole channel slab as full 128-byte lines.  Wave-private staging: DS ops of a wave execute in order, so only
    // compiler barriers separate the write and read phases.
    constexpr int UN = __is_same(T, float) ? 4 : 8;     // channels per store unit (16 B; 32 B for split-f16)
    constexpr int UPP = 64 / UN;                         // units per pixel of the wave's 64-channel slab
    constexpr int UPL = 16 * UPP / 64;                   // units per lane per 16-pixel fragment
    constexpr int SROW = 272;
    static_assert(NW * 16 * SROW <= NS * STAGE, "staging must fit in the ring");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading the ring
    char* const stg = smem + wave * (16 * SROW);
    const int slab0 = ctTile * CT + wci * WCT;
    (void)co0; (void)dy; (void)dx;
    float bad = 0.f;
#pragma unroll
    for (int g = 0; g < FP; ++g) {
        const int pix = ptTile * PT + wpi * WPT + g * 16 + l15;
        int n, oy, ox, plive;
        if constexpr (POS) {
            plive = pix < p.posN ? 1 : 0;
            n = plive ? pix : p.posN - 1;
            oy = posY; ox = posX;
        } else {
            plive = pix < p.M ? 1 : 0;
            const int cp = plive ? pix : p.M - 1;
            n = cp / HoWo;
            const int rem = cp - n * HoWo;
            oy = rem / p.Wo;
            ox = rem - oy * p.Wo;
        }
        const unsigned obase = p.shuffle ? (unsigned)((n * p.yHp + 2 * oy + 1) * p.yWp + 2 * ox + 1)
                                         : (unsigned)((n * p.yHp + oy + 1) * p.yWp + ox + 1);
#pragma unroll
        for (int f = 0; f < FC; ++f) {
            f4 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = acc[f][g][r] * sc[f * 4 + r] + sh[f * 4 + r];
            *reinterpret_cast<f4*>(stg + l15 * SROW + (q * NV + f * 4) * 4) = t;
        }
        wave_lds_sync();
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
            unsigned ob = __shfl(obase, px);
            const int lv = __shfl(plive, px);
            const int row = slab0 + cu * UN;
            int co = row;
            if (p.shuffle) {                             // rows are (dy, dx, co): k2 s2 transposed conv
                const int grp = row / p.Cout;
                co = row - grp * p.Cout;
                ob += (unsigned)((grp >> 1) * p.yWp + (grp & 1));
            }
            if (lv && row < p.rows) {
                if (rbase) {
                    bool done = false;
                    if constexpr (__is_same(T, half_t)) {
                        if (p.res_f32) {                 // wave-uniform: the shortcut comes from the trunk's f32 twin
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
            }
        }
        wave_lds_sync();
    }
    report_bad(p, bad);
