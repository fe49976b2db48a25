// resnet.cpp -- timm ResNet-18 (in_chans=1, num_classes=13) plan: packing + forward schedule.
//
// Module tree per notebooks/model-summary.ipynb (reference) / SURVEY.md Appendix B:
//   conv1 7x7 s2 p3 -> bn1 -> ReLU -> maxpool 3x3 s2 p1 -> layer1..4 (2 BasicBlocks each; first block of
//   layer2-4 has stride 2 and a [conv1x1 s2, BN] shortcut) -> global avg pool -> fc 512 -> 13.
// BasicBlock = conv1-bn1-ReLU-conv2-bn2-(+shortcut)-ReLU: both BNs, the residual add and both ReLUs are conv
// epilogues here; the squares of many boards are batched so the 2x2 / 4x4 stages still form large GEMMs.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "engine.h"
#include "models.h"
#include "pointwise.h"

namespace cv {

Status build_conv_bn_public(Engine& e, ConvLayer& L, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw, int layer_dt = -1);
Status need_public(const ParamMap& pm, const std::string& key, std::vector<int64_t> shape, const float** out);
Status bn_fold_public(const ParamMap& pm, const std::string& prefix, int c, std::vector<float>& scale, std::vector<float>& shift);
Status reject_unknown_keys_public(const ParamMap& pm, const std::vector<std::string>& known, const char* model);

static Status resnet_reserve(Engine& e, int n);
static Status layer1_chain(Engine& e, int n, hipStream_t s);
static Status resnet_chunk(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s);

// form of the chained layer1 launch (f16r): 2 (default) = two workgroups per CU, block 0's f32 output written and read back; 1 = one
// workgroup per CU with 512 registers, the f32 trunk stays in registers -- no round trip, but nothing covers the workgroup's barriers,
// prologue and epilogues: 1.81 against 1.62 ms per 16384 squares (CV_CHAIN_WG=1; r05_tuning.md step 6)
static int chain_form() {
    static const int form = [] { const char* v = std::getenv("CV_CHAIN_WG"); return v && v[0] == '1' ? 1 : 2; }();
    return form;
}

// split-f16 engine: squares from which the dedicated shortcut kernel replaces the generic launch (CV_SHORTCUT_SPLIT_MIN; below it the
// generic launch shares conv1's launch, Engine::PendingConv)
static int fast_sc_min_squares() {
    static const int v = [] { const char* e = std::getenv("CV_SHORTCUT_SPLIT_MIN"); return e && *e ? std::atoi(e) : 1024; }();
    return v;
}

static void bn_keys(std::vector<std::string>& out, const std::string& p) {
    for (const char* leaf : {".weight", ".bias", ".running_mean", ".running_var"}) out.push_back(p + leaf);
}

// fold the stem's output exponent into the device copies of its epilogue constants (ConvLayer::set_exps for the stem)
static Status stem_set_exp(Engine::ResNet& R, int dt, int out_exp, hipStream_t s) {
    if (R.stem_scale.ptr && out_exp == R.stem_out_exp) return Status();
    if (capture_flag()) return fail(1, "stem constants re-folded during graph capture");
    if (R.stem_scale.ptr) CV_HIP(hipStreamSynchronize(s));
    std::vector<float> sc(64), sh(64);
    const int in_exp = dt == kF32 ? 0 : kInputExp;
    for (int i = 0; i < 64; ++i) { sc[i] = std::ldexp(R.h_stem_scale[i], in_exp - out_exp); sh[i] = std::ldexp(R.h_stem_shift[i], -out_exp); }
    CV_TRY(R.stem_scale.upload(sc.data(), 64 * sizeof(float)));
    CV_TRY(R.stem_shift.upload(sh.data(), 64 * sizeof(float)));
    R.stem_out_exp = out_exp;
    return Status();
}

Status resnet_load(Engine& e, const ParamMap& pm) {
    auto m = std::make_unique<Engine::ResNet>();
    Engine::ResNet& R = *m;
    const int dt = e.dt;
    CV_TRY(e.guard_init());
    // the f32 engine materialises the 32x32x64 stem output (34 x 34 x 64 x 4 B per square with its border): keep that
    // tensor under the 4 GiB the 32-bit DMA offsets address
    R.max_cap = e.resnet_chunk;
    if (dt == kF32)
        while ((uint64_t)R.max_cap * 34 * 34 * 64 * 4 >= (1ull << 32) && R.max_cap > 1) R.max_cap /= 2;
    const int S = R.max_cap;
    std::vector<std::string> known;

    {   // stem: conv1 (64,1,7,7) + bn1
        const float* w;
        CV_TRY(need_public(pm, "conv1.weight", {64, 1, 7, 7}, &w));
        std::vector<float> sc, sh;
        CV_TRY(bn_fold_public(pm, "bn1", 64, sc, sh));
        known.push_back("conv1.weight"); bn_keys(known, "bn1");
        std::vector<float> wn(w, w + 64 * 49);
        R.h_stem_scale.assign(64, 0.f); R.h_stem_shift.assign(64, 0.f);
        for (int ch = 0; ch < 64; ++ch) {
            int ex = 0;
            float mx = 0.f;
            for (int k = 0; k < 49; ++k) {
                if (!std::isfinite(w[ch * 49 + k])) return fail(1, "conv1: non-finite weight in the state dict");
                mx = std::max(mx, std::fabs(w[ch * 49 + k]));
            }
            if (!std::isfinite(sc[ch]) || !std::isfinite(sh[ch])) return fail(1, "bn1: non-finite BatchNorm scale/shift (running_var + eps <= 0?)");
            if (dt != kF32 && mx > 0.f) {                   // filter normalisation, as ConvLayer rows (engine.cpp: finish_layer)
                (void)std::frexp(mx, &ex);
                for (int k = 0; k < 49; ++k) wn[ch * 49 + k] = std::ldexp(wn[ch * 49 + k], -ex);
            }
            R.h_stem_scale[ch] = std::ldexp(sc[ch], ex);
            R.h_stem_shift[ch] = sh[ch];
        }
        CV_TRY(R.stem_w.upload(wn.data(), 64 * 49 * sizeof(float)));
        if (dt != kF32) {
            // MFMA A-operand image of the 64x(7x7) filter bank: k-slot (ks, q, j) = filter tap (ky = 4*ks + q, kx = j),
            // MFMA row i of fragment f = channel 16*(i/4) + 4*f + i%4 (the conv epilogue's lane-contiguous order)
            std::vector<_Float16> pk((size_t)2 * 2 * 4 * 64 * 8);
            for (int hl = 0; hl < 2; ++hl)
                for (int ks = 0; ks < 2; ++ks)
                    for (int f = 0; f < 4; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int i = lane & 15, q = lane >> 4, ky = ks * 4 + q;
                            const int ch = 16 * (i / 4) + 4 * f + (i % 4);
                            for (int j = 0; j < 8; ++j) {
                                const float v = (ky < 7 && j < 7) ? wn[ch * 49 + ky * 7 + j] : 0.f;
                                const _Float16 hi = (_Float16)v;
                                pk[((((size_t)hl * 2 + ks) * 4 + f) * 64 + lane) * 8 + j] = hl ? (_Float16)(v - (float)hi) : hi;
                            }
                        }
            CV_TRY(R.stem_wpk.upload(pk.data(), pk.size() * sizeof(_Float16)));
        }
        R.stem_id = e.register_layer("conv1");
        CV_TRY(stem_set_exp(R, dt, 0, nullptr));
    }
    if (dt == kF32) { R.stem_out.shape(32, 32, 64, dt); R.acts.push_back(&R.stem_out); }   // other engines fuse stem + pool
    R.pool_out.shape(16, 16, 64, dt);
    R.pool_out.want32 = e.trunk32;
    R.acts.push_back(&R.pool_out);

    const int widths[4] = {64, 128, 256, 512};
    const int res[4] = {16, 8, 4, 2};
    int cin = 64;
    int64_t macs = 49LL * 64 * 32 * 32;
    for (int l = 0; l < 4; ++l) {
        for (int bi = 0; bi < 2; ++bi) {
            Engine::ResNet::Block& B = R.blocks[l * 2 + bi];
            const std::string p = "layer" + std::to_string(l + 1) + "." + std::to_string(bi);
            const int w = widths[l];
            const int stride = (bi == 0 && l > 0) ? 2 : 1;
            const int64_t px = (int64_t)S * res[l] * res[l];
            known.push_back(p + ".conv1.weight"); bn_keys(known, p + ".bn1");
            known.push_back(p + ".conv2.weight"); bn_keys(known, p + ".bn2");
            CV_TRY(build_conv_bn_public(e, B.conv1, pm, p + ".conv1", p + ".bn1", w, cin, 3, stride, cin, px, res[l]));
            CV_TRY(build_conv_bn_public(e, B.conv2, pm, p + ".conv2", p + ".bn2", w, w, 3, 1, w, px, res[l]));
            B.has_down = (stride != 1 || cin != w);
            if (B.has_down) {
                known.push_back(p + ".downsample.0.weight"); bn_keys(known, p + ".downsample.1");
                // f16r: the shortcut convolution IS the trunk at the stage boundaries -- f16 products there put their rounding
                // straight onto the skip path (CPU emulation of the arithmetic, 4096 squares: worst soft-max error 9.7e-4 with f16
                // shortcuts, 6.5e-4 with exact ones).  It holds 1.1 % of the network's MACs: run it on the f32 MFMA, f32 twin in,
                // f32 twin out; no f16 copy of the shortcut tensor exists.
                CV_TRY(build_conv_bn_public(e, B.down, pm, p + ".downsample.0", p + ".downsample.1", w, cin, 1, stride, cin, px, res[l],
                                            e.trunk32 ? (int)kF32 : -1));
                B.sc.shape(res[l], res[l], w, dt);
                B.sc.want32 = B.sc.only32 = e.trunk32;
                R.acts.push_back(&B.sc);
                // measured (profiles/r05_tuning.md sections 4 and 15): the first form of the dedicated kernel (weight fragments streamed from L2 by
                // every wave) 0.31 / 0.21 / 0.15 ms per 16384 squares against 0.27 / 0.19 / 0.16 for `down` on the f32-input MFMA; the second
                // form (weights resident in LDS, persistent waves, full-line stores) 0.145 / 0.085 / 0.084 -- ON by default;
                // CV_SHORTCUT_FAST=0 switches it off (the generic launch then rides with conv1 at single-board sizes, Engine::PendingConv)
                static const bool fast_on = [] { const char* v = std::getenv("CV_SHORTCUT_FAST"); return !(v && v[0] == '0'); }();
                // round 6: the headline engine (split-f16 tensors) takes the same kernel in its SPLIT form at throughput batch sizes -- there it
                // computes exactly what the generic launch computes (same products, same order: bit-identical), so the choice may follow the batch
                // (layer4's shortcut, 256 -> 512 on 2 x 2 maps, stays on the generic tile there: 0.071 against 0.090 ms per 16384 squares)
                if (((e.trunk32 && dt == kF16) || dt == kSplit) && fast_on && w == 2 * cin && (cin == 64 || cin == 128 || (cin == 256 && dt != kSplit))) {
                    // split-f16 image of the 1x1 weights for shortcut1x1s2: rows normalised to [0.5, 1) (exponent into the scale), hi / lo halves,
                    // [channel group of 128][k-step of 32][fragment 8][hi | lo][lane 64][8]; MFMA row i of fragment f = channel 32 (i/4) + 4 f + i%4
                    const float* wd;
                    CV_TRY(need_public(pm, p + ".downsample.0.weight", {w, cin, 1, 1}, &wd));
                    std::vector<float> sc, sh;
                    CV_TRY(bn_fold_public(pm, p + ".downsample.1", w, sc, sh));
                    std::vector<int> rex((size_t)w, 0);
                    B.h_sc_scale.assign((size_t)w, 0.f); B.h_sc_shift.assign((size_t)w, 0.f);
                    for (int ch = 0; ch < w; ++ch) {
                        float mx = 0.f;
                        for (int k = 0; k < cin; ++k) {
                            if (!std::isfinite(wd[(size_t)ch * cin + k])) return fail(1, p + ".downsample.0: non-finite weight in the state dict");
                            mx = std::max(mx, std::fabs(wd[(size_t)ch * cin + k]));
                        }
                        if (mx > 0.f) (void)std::frexp(mx, &rex[ch]);
                        B.h_sc_scale[ch] = std::ldexp(sc[ch], rex[ch]);
                        B.h_sc_shift[ch] = sh[ch];
                    }
                    const int KS = cin / 32, CG = w / 128;
                    std::vector<_Float16> pk((size_t)CG * KS * 16 * 64 * 8);
                    for (int cg = 0; cg < CG; ++cg)
                        for (int ks = 0; ks < KS; ++ks)
                            for (int f = 0; f < 8; ++f)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int i = lane & 15, q = lane >> 4;
                                    const int ch = cg * 128 + (i / 4) * 32 + f * 4 + (i % 4);
                                    for (int j = 0; j < 8; ++j) {
                                        const float v = std::ldexp(wd[(size_t)ch * cin + ks * 32 + q * 8 + j], -rex[ch]);
                                        const _Float16 hi = (_Float16)v;
                                        const size_t at = ((((size_t)(cg * KS + ks) * 8 + f) * 2) * 64 + lane) * 8 + j;
                                        pk[at] = hi;
                                        pk[at + 64 * 8] = (_Float16)(v - (float)hi);
                                    }
                                }
                    CV_TRY(B.sc_wpk.upload(pk.data(), pk.size() * sizeof(_Float16)));
                    CV_TRY(B.sc_scale.alloc((size_t)w * sizeof(float), false));
                    CV_TRY(B.sc_shift.alloc((size_t)w * sizeof(float), false));
                    B.sc_id = B.down.layer_id;
                    B.fast_sc = true;
                }
                macs += (int64_t)cin * w * res[l] * res[l];
            }
            B.mid.shape(res[l], res[l], w, dt);
            B.out.shape(res[l], res[l], w, dt);
            B.out.want32 = e.trunk32;                        // f16r: the residual trunk never rounds to f16 (Activation::buf32)
            R.acts.push_back(&B.mid); R.acts.push_back(&B.out);
            macs += ((int64_t)cin * 9 * w + (int64_t)w * 9 * w) * res[l] * res[l];
            cin = w;
        }
    }
    {
        const float *w, *b;
        CV_TRY(need_public(pm, "fc.weight", {13, 512}, &w));
        CV_TRY(need_public(pm, "fc.bias", {13}, &b));
        known.push_back("fc.weight"); known.push_back("fc.bias");
        for (int i = 0; i < 13 * 512; ++i)
            if (!std::isfinite(w[i])) return fail(1, "fc: non-finite weight in the state dict");
        for (int i = 0; i < 13; ++i)
            if (!std::isfinite(b[i])) return fail(1, "fc: non-finite bias in the state dict");
        CV_TRY(R.fc_w.upload(w, 13 * 512 * sizeof(float)));
        CV_TRY(R.fc_b.upload(b, 13 * sizeof(float)));
        R.head_id = e.register_layer("fc");
        macs += 13 * 512;
    }
    CV_TRY(reject_unknown_keys_public(pm, known, "resnet18(in_chans=1, num_classes=13)"));
    R.macs = macs;
    {   // f16r: layer1 as one chained launch -- the four layers' weight stages (9 taps x 64 rows x 128 B each) back to back
        static const bool on = [] { const char* v = std::getenv("CV_RESNET_CHAIN"); return !(v && v[0] == '0'); }();
        ConvLayer* L[4] = {&R.blocks[0].conv1, &R.blocks[0].conv2, &R.blocks[1].conv1, &R.blocks[1].conv2};
        bool fits = on && e.trunk32 && dt == kF16 && conv_halo_has_chain();
        const size_t one = (size_t)9 * 64 * 128;
        for (ConvLayer* l : L) fits = fits && l->dt == kF16 && l->ct == 64 && l->rows == 64 && l->nStages == 9 && l->nCt == 1 && l->w.bytes >= one && l->halo_ok;
        if (fits) {
            CV_TRY(R.chain_w.alloc(4 * one, false));
            for (int i = 0; i < 4; ++i) CV_HIP(sync_memcpy((char*)R.chain_w.ptr + i * one, L[i]->w.ptr, one, hipMemcpyDeviceToDevice));
            R.chain_ok = true;
        }
    }
    e.resnet = std::move(m);

    // range calibration on 128 squares: noise, flat grey levels, gradients and checkers (what board crops look like)
    // (the rounding-bias pass of the fp16 layers, ConvLayer::want_round_err, reads 128 more: 64 of noise and 64 smooth ones --
    // bilinear blow-ups of a random 4 x 4 grid, what a board crop looks like away from piece edges -- so that the channel means it
    // measures are not those of one texture)
    Status st = resnet_reserve(e, 128);
    if (st.ok() && dt != kF32 && calibration_enabled()) {
        constexpr int kCal = 128, kBias = 256;
        std::vector<float> host((size_t)kBias * 4096);
        uint32_t rs = 0x2545F491u;
        auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return (rs >> 24) & 0xffu; };
        for (int q = 0; q < kCal; ++q)
            for (int y = 0; y < 64; ++y)
                for (int x = 0; x < 64; ++x) {
                    float v;
                    if (q < 48) v = (float)rnd();
                    else if (q < 80) v = (float)((q - 48) * 8 + 3);                                   // flat levels 3..251
                    else if (q < 104) v = (float)(((x * (q - 79)) + y * 3) & 255);                      // ramps
                    else v = (((x >> (q & 3)) + (y >> ((q >> 2) & 3))) & 1) ? 235.f : (float)(20 + (q - 104) * 6);   // checkers
                    host[((size_t)q * 64 + y) * 64 + x] = v / 255.f;
                }
        for (int q = kCal; q < kBias; ++q) {
            float grid[5][5];
            for (auto& row : grid) for (float& g : row) g = (float)rnd();
            for (int y = 0; y < 64; ++y)
                for (int x = 0; x < 64; ++x) {
                    float v;
                    if (q < kCal + 64) v = (float)rnd();
                    else {                                                                               // bilinear, cell = 16 pixels
                        const int gy = y >> 4, gx = x >> 4;
                        const float fy = (float)(y & 15) / 16.f, fx = (float)(x & 15) / 16.f;
                        const float top = grid[gy][gx] * (1.f - fx) + grid[gy][gx + 1] * fx, bot = grid[gy + 1][gx] * (1.f - fx) + grid[gy + 1][gx + 1] * fx;
                        v = std::floor(top * (1.f - fy) + bot * fy + 0.5f);
                    }
                    host[((size_t)q * 64 + y) * 64 + x] = v / 255.f;
                }
        }
        DeviceBuffer xin, lout;
        st = xin.upload(host.data(), host.size() * sizeof(float));
        if (st.ok()) st = lout.alloc((size_t)kBias * 13 * sizeof(float), false);
        if (st.ok())
            st = e.calibrate(e.resnet->acts, [&]() -> Status {                 // the statistics accumulate over the chunks
                for (int off = 0; off < 128; off += e.resnet->cap) {
                    const int c = std::min(e.resnet->cap, 128 - off);
                    CV_TRY(resnet_chunk(e, (const float*)xin.ptr + (size_t)off * 4096, false, c, (float*)lout.ptr + (size_t)off * 13, false, nullptr));
                }
                return Status();
            }, nullptr, "ResNet-18");
        if (st.ok()) {                                                          // exponents are final: rounding-bias pass of the f16 layers
            std::vector<ConvLayer*> layers;
            for (auto& B : e.resnet->blocks) { layers.push_back(&B.conv1); layers.push_back(&B.conv2); if (B.has_down) layers.push_back(&B.down); }
            st = e.calibrate_rounding_bias(layers, [&]() -> Status {
                for (int off = 0; off < kBias; off += e.resnet->cap) {
                    const int c = std::min(e.resnet->cap, kBias - off);
                    CV_TRY(resnet_chunk(e, (const float*)xin.ptr + (size_t)off * 4096, false, c, (float*)lout.ptr + (size_t)off * 13, false, nullptr));
                }
                return Status();
            }, nullptr);
        }
        if (st.ok()) {
            hipError_t he = device_synchronize();
            if (he != hipSuccess) st = hip_fail(he, "ResNet-18 calibration");
        }
    }
    if (!st.ok()) e.resnet.reset();
    return st;
}

static Status resnet_reserve(Engine& e, int n) {
    Engine::ResNet& R = *e.resnet;
    const int want = std::min(R.max_cap, std::max(n, 1));
    if (want <= R.cap) return Status();
    CV_HIP(device_synchronize());
    e.graph_invalidate();                            // captured launches hold the old buffers
    CV_TRY(Activation::reserve_all(R.acts, want));
    R.cap = want;
    const int S = want;
    R.taps.clear();
    if (e.dt == kF32) R.taps["act1"] = R.stem_out.ref(S);
    R.taps["maxpool"] = R.pool_out.ref(S);
    for (int l = 0; l < 4; ++l) {
        for (int bi = 0; bi < 2; ++bi) {
            Engine::ResNet::Block& B = R.blocks[l * 2 + bi];
            const std::string p = "layer" + std::to_string(l + 1) + "." + std::to_string(bi);
            // chained layer1 (f16r): conv1's output of both blocks and the f16 copy of the first block's output live in LDS only --
            // no tap for the former, the latter is read from its f32 twin
            const bool chained = R.chain_ok && l == 0;
            if (!chained) R.taps[p + ".act1"] = B.mid.ref(S);
            if (chained && bi == 0) {                        // block 0's output: its f32 twin (form 2) or nowhere outside the kernel (form 1)
                if (chain_form() == 2) R.taps[p] = B.out.ref32(S);
            } else R.taps[p] = B.out.ref(S);
            if (B.has_down) R.taps[p + ".downsample"] = B.sc.only32 ? B.sc.ref32(S) : B.sc.ref(S);
        }
        R.taps["layer" + std::to_string(l + 1)] = R.blocks[l * 2 + 1].out.ref(S);
    }
    return Status();
}

int64_t resnet_macs(Engine& e) { return e.resnet ? e.resnet->macs : 0; }

Status resnet_activation(Engine& e, const std::string& name, TensorRef* out) {
    if (!e.resnet) return fail(3, "ResNet-18 not loaded");
    auto it = e.resnet->taps.find(name);
    if (it == e.resnet->taps.end()) {
        // the fp16 classifier runs layer1 as ONE launch: conv1's outputs (and, in the one-workgroup form, block 0's output) exist in LDS only
        if (e.resnet->chain_ok && (name == "layer1.0.act1" || name == "layer1.1.act1" || name == "layer1.0"))
            return fail(1, "ResNet activation '" + name + "' is not materialised by precision f16r: layer1 runs as one chained launch and "
                           "keeps it in LDS -- set CV_RESNET_CHAIN=0 before creating the engine to run layer1 as four launches");
        return fail(1, "unknown ResNet activation '" + name + "'");
    }
    *out = it->second;
    out->exp = static_cast<Activation*>(out->owner)->exp;
    out->N = e.resnet->last_n;
    return Status();
}

static Status resnet_chunk(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s) {
    Engine::ResNet& R = *e.resnet;
    R.last_n = n;
    e.ws_slot = 1;
    const int dt = e.dt;
    const double esz = dtype_size(dt), in_b = x_u8 ? 1.0 : 4.0;
    auto begin = [&](const char* name, double macs, double bytes) { if (e.profiling) e.prof_begin(name, false, macs, s, bytes); };
    auto end = [&](const char* name, hipError_t err) -> Status {
        if (e.profiling) e.prof_end(s);
        if (err != hipSuccess) return hip_fail(err, name);
        return Status();
    };
    if (dt == kF32) {
        begin("stem7x7", 49.0 * 64 * 1024 * n, (double)n * (4096 * in_b + 1024 * 64 * esz));
        CV_TRY(end("stem7x7", stem7x7(dt, x, x_u8, n, (const float*)R.stem_w.ptr, (const float*)R.stem_scale.ptr,
                                       (const float*)R.stem_shift.ptr, R.stem_out.ref(n), s)));
        begin("maxpool3x3s2", 0, (double)n * (1024 + 256) * 64 * esz);
        CV_TRY(end("maxpool3x3s2", maxpool3x3s2(dt, R.stem_out.ref(n), R.pool_out.ref(n), s)));
    } else {
        CV_TRY(stem_set_exp(R, dt, R.pool_out.exp, s));
        begin("stem7x7+maxpool (mfma)", 49.0 * 64 * 1024 * n, (double)n * (4096 * in_b + 256 * 64 * (esz + (R.pool_out.want32 ? 4.0 : 0.0))));
        CV_TRY(end("stem_pool_mfma", stem_pool_mfma(dt, x, x_u8, n, R.stem_wpk.ptr, (const float*)R.stem_scale.ptr,
                                                     (const float*)R.stem_shift.ptr, kInputExp, R.pool_out.ref(n),
                                                     e.guard_ptr(), R.stem_id, s)));
        if (e.calibrating) CV_TRY(e.measure(R.pool_out.ref(n), s));
    }
    TensorRef cur = R.pool_out.ref(n);
    int first_block = 0;
    if (R.chain_ok && !e.calibrating) {                // the calibration passes run layer by layer (every tensor is measured)
        CV_TRY(layer1_chain(e, n, s));
        cur = R.blocks[1].out.ref(n);
        first_block = 2;
    }
    for (int i = first_block; i < 8; ++i) {
        Engine::ResNet::Block& B = R.blocks[i];
        TensorRef shortcut = cur;
        bool conv1_done = false;
        // the shortcut convolution and conv1 both read the block input and do not depend on each other: at single-board sizes neither
        // fills the chip, and they go out as ONE launch (Engine::PendingConv; three launches of ~7-10 us fewer per forward)
        auto down_beside_conv1 = [&](const TensorRef& down_in, const TensorRef& down_out) -> Status {
            Engine::PendingConv pa, pb;
            e.defer = &pa; e.defer_first = true;
            Status st = e.run_conv(B.down, down_in, down_out, nullptr, false, s);
            e.defer = &pb; e.defer_first = false;
            if (st.ok()) st = e.run_conv(B.conv1, cur, B.mid.ref(n), nullptr, true, s);
            e.defer = nullptr;
            CV_TRY(st);
            conv1_done = true;
            return e.flush_pending(pa, pb, s);
        };
        if (B.has_down) {
            if (B.sc.only32) {                                 // f16r: f32-grade convolution from the trunk's twin to the shortcut's
                TensorRef in32 = cur;
                in32.base = cur.base32; in32.base32 = nullptr; in32.f32_only = 1;
                // every launch size takes the same kernel: results across batch shapes agree to 1e-4 (only f32 summation
                // orders differ), and the generic f32-MFMA shortcut differs from this one by up to 1e-3 -- a switch at 1024 squares made the
                // 8-rank test's single-photo recomputations disagree with the FENs of its 256-board jobs
                if (B.fast_sc && !e.calibrating) {
                    const TensorRef out32 = B.sc.ref32(n);
                    if (B.sc_in_exp != in32.exp || B.sc_out_exp != out32.exp) {          // fold the tensor exponents (ConvLayer::set_exps)
                        if (capture_flag()) return fail(1, "shortcut constants re-folded during graph capture");
                        CV_HIP(hipStreamSynchronize(s));
                        std::vector<float> sc(B.h_sc_scale.size()), sh(sc.size());
                        for (size_t k = 0; k < sc.size(); ++k) {
                            sc[k] = std::ldexp(B.h_sc_scale[k], in32.exp - out32.exp);
                            sh[k] = std::ldexp(B.h_sc_shift[k], -out32.exp);
                            if (!std::isfinite(sc[k]) || !std::isfinite(sh[k])) return fail(1, "shortcut: range factors leave the f32 range");
                        }
                        CV_HIP(sync_memcpy(B.sc_scale.ptr, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
                        CV_HIP(sync_memcpy(B.sc_shift.ptr, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
                        B.sc_in_exp = in32.exp; B.sc_out_exp = out32.exp;
                    }
                    const double opx = (double)n * out32.H * out32.W;
                    if (e.profiling) {
                        e.prof_begin(B.down.name, true, opx * in32.C * out32.C, s, opx * (in32.C + out32.C) * 4.0 + (double)in32.C * out32.C * 4.0);
                        e.prof.back().kernel = "shortcut1x1s2_kernel<" + std::to_string(in32.C) + ">";
                    }
                    const hipError_t err = shortcut1x1s2(in32, B.sc_wpk.ptr, (const float*)B.sc_scale.ptr, (const float*)B.sc_shift.ptr, out32,
                                                         e.guard_ptr(), B.sc_id, s);
                    if (e.profiling) e.prof_end(s);
                    if (err != hipSuccess) return hip_fail(err, "shortcut1x1s2");
                } else {
                    CV_TRY(down_beside_conv1(in32, B.sc.ref32(n)));
                }
            } else if (B.fast_sc && dt == kSplit && !e.calibrating && n >= fast_sc_min_squares()) {
                // split-f16 tensors in, split-f16 tensors out; below the threshold the generic launch rides with conv1 (one launch fewer)
                const TensorRef out = B.sc.ref(n);
                if (B.sc_in_exp != cur.exp || B.sc_out_exp != out.exp) {
                    if (capture_flag()) return fail(1, "shortcut constants re-folded during graph capture");
                    CV_HIP(hipStreamSynchronize(s));
                    std::vector<float> sc(B.h_sc_scale.size()), sh(sc.size());
                    for (size_t k = 0; k < sc.size(); ++k) {
                        sc[k] = std::ldexp(B.h_sc_scale[k], cur.exp - out.exp);
                        sh[k] = std::ldexp(B.h_sc_shift[k], -out.exp);
                        if (!std::isfinite(sc[k]) || !std::isfinite(sh[k])) return fail(1, "shortcut: range factors leave the f32 range");
                    }
                    CV_HIP(sync_memcpy(B.sc_scale.ptr, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
                    CV_HIP(sync_memcpy(B.sc_shift.ptr, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
                    B.sc_in_exp = cur.exp; B.sc_out_exp = out.exp;
                }
                const double opx = (double)n * out.H * out.W;
                if (e.profiling) {
                    e.prof_begin(B.down.name, true, opx * cur.C * out.C, s, opx * (cur.C + out.C) * 4.0 + (double)cur.C * out.C * 4.0);
                    e.prof.back().kernel = "shortcut1x1s2_kernel<" + std::to_string(cur.C) + ",split>";
                }
                const hipError_t err = shortcut1x1s2(cur, B.sc_wpk.ptr, (const float*)B.sc_scale.ptr, (const float*)B.sc_shift.ptr, out,
                                                     e.guard_ptr(), B.sc_id, s, /*split=*/true);
                if (e.profiling) e.prof_end(s);
                if (err != hipSuccess) return hip_fail(err, "shortcut1x1s2 (split)");
            } else {
                CV_TRY(down_beside_conv1(cur, B.sc.ref(n)));
            }
            shortcut = B.sc.ref(n);
        }
        if (!conv1_done) CV_TRY(e.run_conv(B.conv1, cur, B.mid.ref(n), nullptr, true, s));
        CV_TRY(e.run_conv(B.conv2, B.mid.ref(n), B.out.ref(n), &shortcut, true, s));
        cur = B.out.ref(n);
    }
    int head_dt = dt;
    if (cur.base32) { cur.base = cur.base32; head_dt = kF32; }   // f16r: pool the trunk's f32 twin
    begin("head_avgpool_fc", 13.0 * 512 * n, (double)n * (cur.H * cur.W * 512 * dtype_size(head_dt) + 13 * 4));
    CV_TRY(end("head_avgpool_fc", head_avgpool_fc(head_dt, cur, (const float*)R.fc_w.ptr, (const float*)R.fc_b.ptr, out,
                                                   softmax ? 1 : 0, e.guard_ptr(), R.head_id, s)));
    return Status();
}

// layer1.0 and layer1.1 in one launch (f16r engine): pool_out (f16 copy + f32 twin) -> blocks[1].out (f16 copy + f32 twin), the first
// block's f32 output twin as the only intermediate in memory.  Same arithmetic as the four run_conv launches: f16 products, f32
// accumulation, BN affine and residual in f32, one rounding to f16 per tensor.
static Status layer1_chain(Engine& e, int n, hipStream_t s) {
    Engine::ResNet& R = *e.resnet;
    Engine::ResNet::Block &B0 = R.blocks[0], &B1 = R.blocks[1];
    const TensorRef x = R.pool_out.ref(n), y0 = B0.out.ref(n), y1 = B1.out.ref(n);
    if (!x.base || !x.base32 || !y0.base32 || !y1.base || !y1.base32 || x.Cs != 64 || y1.Cs != 64 || x.H != 16 || x.W != 16)
        return fail(3, "layer1 chain: trunk tensors are not in the f16r layout");
    // tensor exponents exactly as the layer-by-layer path folds them
    CV_TRY(B0.conv1.set_exps(x.exp, B0.mid.exp, s));
    CV_TRY(B0.conv2.set_exps(B0.mid.exp, B0.out.exp, s));
    CV_TRY(B1.conv1.set_exps(B0.out.exp, B1.mid.exp, s));
    CV_TRY(B1.conv2.set_exps(B1.mid.exp, B1.out.exp, s));
    ConvLayer* L[4] = {&B0.conv1, &B0.conv2, &B1.conv1, &B1.conv2};
    ConvParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = reinterpret_cast<const char*>(x.base);
    p.w = reinterpret_cast<const char*>(R.chain_w.ptr);
    p.y = reinterpret_cast<char*>(y1.base);
    p.y32 = reinterpret_cast<char*>(y1.base32);
    p.M = n * 256; p.Ho = 16; p.Wo = 16;
    p.xHp = 18; p.xWp = 18; p.stride = 1; p.xCs = 64; p.xCoffBytes = 0;
    p.yHp = 18; p.yWp = 18; p.yCs = 64; p.yCoff = 0;
    p.Cout = 64; p.rows = 64; p.nStages = 36; p.nCt = 1; p.relu = 1;
    p.flag = e.guard_ptr();
    p.layer_id = B1.conv2.layer_id;
    // bit 4: convolution 1's residual epilogue staged through the (dead) halo buffer in the unit layout (CV_CHAIN_MIDSTAGE; two-workgroup form)
    static const int mid_staged = [] { const char* v = std::getenv("CV_CHAIN_MIDSTAGE"); return v && v[0] == '1' ? 16 : 0; }();
    p.chain = chain_form() | (chain_form() == 2 ? mid_staged : 0);
    for (int i = 0; i < 4; ++i) {
        p.ch_scale[i] = reinterpret_cast<const float*>(L[i]->scale.ptr);
        p.ch_shift[i] = reinterpret_cast<const float*>(L[i]->shift.ptr);
        p.ch_layer_id[i] = L[i]->layer_id;
    }
    p.ch_res0 = reinterpret_cast<const char*>(x.base32);
    p.ch_y32_mid = reinterpret_cast<char*>(y0.base32);
    p.ch_res_mul[0] = std::ldexp(1.f, x.exp - B0.out.exp);
    p.ch_res_mul[1] = std::ldexp(1.f, B0.out.exp - B1.out.exp);
    // the last convolution's epilogue is the kernel's ordinary one (two-workgroup form): its constants in the ordinary fields
    p.scale = p.ch_scale[3]; p.shift = p.ch_shift[3];
    p.res = reinterpret_cast<const char*>(y0.base32); p.res_f32 = 1; p.res_mul = p.ch_res_mul[1]; p.rCs = 64; p.rCoff = 0;
    if (e.profiling) {
        // compulsory bytes: f16 input + its f32 twin, the first block's f32 output written and read back, f32 + f16 output, weights
        const double px = (double)n * 256 * 64;
        e.prof_begin("layer1 (4 convs, chained)", true, 4.0 * 9 * 64 * 64 * 256 * (double)n, s, px * (2 + 4 + 4 + 4 + 4 + 2) + 4.0 * 9 * 64 * 64 * 2);
        e.prof.back().kernel = "conv3x3_halo_kernel<half_t,64,16x16,CHAIN4>";
    }
    const hipError_t err = conv_halo_chain_launch(p, n, s);
    if (e.profiling) e.prof_end(s);
    if (err != hipSuccess) return hip_fail(err, "layer1 chain launch");
    return Status();
}

Status resnet_forward(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s) {
    if (!e.resnet) return fail(3, "ResNet-18 weights not loaded (call cv_load_resnet18 first)");
    if (n < 0 || (n > 0 && (!x || !out))) return fail(1, "cv_resnet18_forward: null tensor or negative batch");
    if (n == 0) return Status();
    CV_TRY(e.order_forward(1, s));
    CV_TRY(resnet_reserve(e, n));
    if (n <= e.resnet->cap && n <= 512) {                // one small chunk (a board's 64 squares): the launch sequence replays as a hipGraph
        Engine::GraphKey key;
        key.model = 1; key.n = n; key.flags = (x_u8 ? 1 : 0) | (softmax ? 2 : 0); key.x = x; key.out = out;
        e.resnet->last_n = n;                            // a graph replay skips resnet_chunk's host side (see unet_forward)
        e.ws_slot = 1;
        return e.run_graphed(key, s, [&](hipStream_t st) { return resnet_chunk(e, x, x_u8, n, out, softmax, st); });
    }
    const size_t in_stride = (size_t)64 * 64 * (x_u8 ? 1 : 4);
    for (int off = 0; off < n; off += e.resnet->cap) {
        const int c = std::min(e.resnet->cap, n - off);
        CV_TRY(resnet_chunk(e, (const char*)x + (size_t)off * in_stride, x_u8, c, out + (size_t)off * 13, softmax, s));
    }
    return Status();
}

}  // namespace cv
