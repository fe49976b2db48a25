// unet.cpp -- UNet(3 -> 1) plan: packing from the reference state dict + the forward schedule.
//
// Follows the module tree the reference constructs at chessvision/core.py:88 (layout: SURVEY.md Appendix A):
//   inc -> down1..4 (MaxPool2d(2) + DoubleConv) -> up1..4 (ConvTranspose2d k2 s2 | bilinear x2, cat([skip, up]),
//   DoubleConv) -> outc (1x1).
// MI355X-first differences from the torch graph: BN+ReLU live in the conv epilogues; `torch.cat` does not exist
// (the encoder's second conv and the up-sampler write the two channel halves of one buffer); F.pad is a
// no-op at 256x256 and is not materialised; the batch is processed in chunks that keep every layer's grid
// at >= one full wave of workgroups while bounding the live working set.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "engine.h"
#include "models.h"
#include "pointwise.h"

namespace cv {


static Status need(const ParamMap& pm, const std::string& key, std::vector<int64_t> shape, const float** out) {
    auto it = pm.find(key);
    if (it == pm.end()) return fail(1, "state dict is missing key '" + key + "'");
    if (it->second.shape != shape) {
        std::string got, want;
        for (auto d : it->second.shape) got += std::to_string(d) + ",";
        for (auto d : shape) want += std::to_string(d) + ",";
        return fail(1, "state dict key '" + key + "' has shape (" + got + ") expected (" + want + ")");
    }
    *out = it->second.data;
    return Status();
}
Status need_public(const ParamMap& pm, const std::string& key, std::vector<int64_t> shape, const float** out) {
    return need(pm, key, std::move(shape), out);
}

// BatchNorm2d(eval): y = (x - mean) / sqrt(var + eps) * gamma + beta  ->  scale, shift
static Status bn_fold(const ParamMap& pm, const std::string& prefix, int c, std::vector<float>& scale,
                      std::vector<float>& shift) {
    const float *g, *b, *m, *v;
    CV_TRY(need(pm, prefix + ".weight", {c}, &g));
    CV_TRY(need(pm, prefix + ".bias", {c}, &b));
    CV_TRY(need(pm, prefix + ".running_mean", {c}, &m));
    CV_TRY(need(pm, prefix + ".running_var", {c}, &v));
    scale.resize(c); shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const float s = g[i] / std::sqrt(v[i] + 1e-5f);
        scale[i] = s;
        shift[i] = b[i] - m[i] * s;
    }
    return Status();
}
Status bn_fold_public(const ParamMap& pm, const std::string& prefix, int c, std::vector<float>& scale, std::vector<float>& shift) {
    return bn_fold(pm, prefix, c, scale, shift);
}

static Status build_conv_bn(Engine& e, ConvLayer& L, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw = 0, int layer_dt = -1) {
    const float* w;
    CV_TRY(need(pm, conv_key + ".weight", {cout, cin, k, k}, &w));
    std::vector<float> sc, sh;
    CV_TRY(bn_fold(pm, bn_key, cout, sc, sh));
    CV_TRY(L.build_conv(conv_key, layer_dt < 0 ? e.dt : layer_dt, w, cout, cin, k, stride, sc.data(), sh.data(), cinPad, pixels, out_hw));
    L.layer_id = e.register_layer(conv_key);
    return Status();
}
Status build_conv_bn_public(Engine& e, ConvLayer& L, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw, int layer_dt) {
    return build_conv_bn(e, L, pm, conv_key, bn_key, cout, cin, k, stride, cinPad, pixels, out_hw, layer_dt);
}

// every key the architecture defines, so that a checkpoint of a DIFFERENT architecture (extra / renamed keys: the
// reference's UNet is an un-vendored submodule on an `experimental` branch) is rejected instead of half-loaded
static Status reject_unknown_keys(const ParamMap& pm, const std::vector<std::string>& known, const char* model) {
    for (const auto& kv : pm) {
        bool ok = false;
        for (const auto& k : known)
            if (kv.first == k) { ok = true; break; }
        if (!ok) return fail(1, std::string("state dict has key '") + kv.first + "' that " + model + " does not define");
    }
    return Status();
}
Status reject_unknown_keys_public(const ParamMap& pm, const std::vector<std::string>& known, const char* model) {
    return reject_unknown_keys(pm, known, model);
}
static void bn_keys(std::vector<std::string>& out, const std::string& p) {
    for (const char* leaf : {".weight", ".bias", ".running_mean", ".running_var"}) out.push_back(p + leaf);
}
static void double_conv_keys(std::vector<std::string>& out, const std::string& p) {
    out.push_back(p + "0.weight"); bn_keys(out, p + "1");
    out.push_back(p + "3.weight"); bn_keys(out, p + "4");
}

static Status unet_reserve(Engine& e, int n);
static Status unet_chunk(Engine& e, const void* x, bool x_u8, int n, float* logits, uint8_t* mask, float thr, hipStream_t s);

// Two deterministic calibration images, NCHW f32 in [0,1]: uniform noise, and a structured frame (dark noise, a bright
// checkered quadrilateral, saturated white and black blocks) -- the extremes of what a photo can put into the network.
static void calibration_images(std::vector<float>& x) {
    x.assign((size_t)2 * 3 * 65536, 0.f);
    uint32_t st = 0x9E3779B9u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 24) & 0xffu; };
    for (size_t i = 0; i < (size_t)3 * 65536; ++i) x[i] = (float)rnd() / 255.f;
    float* im = x.data() + (size_t)3 * 65536;
    for (int y = 0; y < 256; ++y)
        for (int xx = 0; xx < 256; ++xx) {
            const bool board = xx > 40 + y / 16 && xx < 215 - y / 20 && y > 35 && y < 220;
            for (int c = 0; c < 3; ++c) {
                float v = (float)(rnd() % 40) / 255.f;
                if (board) v = ((((xx - 40) / 22 + (y - 35) / 23) & 1) ? 230.f : 165.f) / 255.f;
                if (xx < 32 && y < 32) v = 1.f;
                if (xx >= 224 && y >= 224) v = 0.f;
                im[((size_t)c * 256 + y) * 256 + xx] = v;
            }
        }
}

Status unet_load(Engine& e, const ParamMap& pm) {
    auto m = std::make_unique<Engine::UNet>();
    Engine::UNet& U = *m;
    const int dt = e.dt;
    CV_TRY(e.guard_init());
    U.bilinear = pm.find("up1.up.weight") == pm.end();
    {   // CV_FUSE_HEAD=0 keeps OutConv a separate kernel (then the up4 output tensor can be read back)
        const char* v = std::getenv("CV_FUSE_HEAD");
        U.fuse_head = !(v && v[0] == '0');
    }
    U.max_cap = e.unet_chunk;
    const int S = U.max_cap;                          // tile choices are made for full chunks
    const int f = U.bilinear ? 2 : 1;
    U.c5 = 1024 / f;
    const int enc_c[5] = {64, 128, 256, 512, U.c5};
    const int res[5] = {256, 128, 64, 32, 16};
    auto px = [&](int level) { return (int64_t)S * res[level] * res[level]; };

    std::vector<std::string> known;
    double_conv_keys(known, "inc.double_conv.");
    // encoder
    CV_TRY(build_conv_bn(e, U.inc0, pm, "inc.double_conv.0", "inc.double_conv.1", 64, 3, 3, 1, 8, px(0), 256));
    CV_TRY(build_conv_bn(e, U.inc1, pm, "inc.double_conv.3", "inc.double_conv.4", 64, 64, 3, 1, 64, px(0), 256));
    if (dt != kF32) {
        // the same layer for the fused first-layer kernel (pointwise.hip: inc0_mfma): rows normalised exactly as finish_layer
        // does (the epilogue constants of U.inc0 carry the row exponents), k = (ky*3 + kx)*3 + c, padded 27 -> 32
        const char* v = std::getenv("CV_INC0");
        U.fused_inc0 = !(v && v[0] == '0');
        const float* w;
        CV_TRY(need(pm, "inc.double_conv.0.weight", {64, 3, 3, 3}, &w));
        std::vector<_Float16> pk((size_t)2 * 4 * 64 * 8);
        for (int hl = 0; hl < 2; ++hl)
            for (int f = 0; f < 4; ++f)
                for (int lane = 0; lane < 64; ++lane) {
                    const int i = lane & 15, q = lane >> 4;
                    const int ch = 16 * (i / 4) + 4 * f + (i % 4);
                    float mx = 0.f;
                    for (int t = 0; t < 27; ++t) mx = std::max(mx, std::fabs(w[ch * 27 + t]));
                    int ex = 0;
                    if (mx > 0.f) (void)std::frexp(mx, &ex);
                    for (int j = 0; j < 8; ++j) {
                        const int k = q * 8 + j, tap = k / 3, c = k % 3;
                        const float val = k < 27 ? std::ldexp(w[(ch * 3 + c) * 9 + tap], -ex) : 0.f;
                        const _Float16 hi = (_Float16)val;
                        pk[(((size_t)hl * 4 + f) * 64 + lane) * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
                    }
                }
        CV_TRY(U.inc0_wpk.upload(pk.data(), pk.size() * sizeof(_Float16)));
        if (dt == kSplit) {
            // ... and for the producer inside inc.double_conv.3's kernel: two 16-row fragments per 32-channel block, lane
            // (i, q) of fragment f <-> channel 32*cb + 8*(i/4) + 4*f + i%4, so that a lane ends with the 8 channels of group q
            const char* v2 = std::getenv("CV_FUSE_INC");
            U.fused_inc = U.fused_inc0 && !(v2 && v2[0] == '0');
            std::vector<_Float16> pk2((size_t)2 * 2 * 2 * 64 * 8);
            for (int cb = 0; cb < 2; ++cb)
                for (int f = 0; f < 2; ++f)
                    for (int hl = 0; hl < 2; ++hl)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int i = lane & 15, q = lane >> 4;
                            const int ch = 32 * cb + 8 * (i / 4) + 4 * f + (i % 4);
                            float mx = 0.f;
                            for (int t = 0; t < 27; ++t) mx = std::max(mx, std::fabs(w[ch * 27 + t]));
                            int ex = 0;
                            if (mx > 0.f) (void)std::frexp(mx, &ex);
                            for (int j = 0; j < 8; ++j) {
                                const int k = q * 8 + j, tap = k / 3, c = k % 3;
                                const float val = k < 27 ? std::ldexp(w[(ch * 3 + c) * 9 + tap], -ex) : 0.f;
                                const _Float16 hi = (_Float16)val;
                                pk2[((((size_t)cb * 2 + f) * 2 + hl) * 64 + lane) * 8 + j] = hl ? (_Float16)(val - (float)hi) : hi;
                            }
                        }
            CV_TRY(U.inc0_wpk2.upload(pk2.data(), pk2.size() * sizeof(_Float16)));
        }
    }
    for (int i = 0; i < 4; ++i) {
        const std::string p = "down" + std::to_string(i + 1) + ".maxpool_conv.1.double_conv.";
        double_conv_keys(known, p);
        CV_TRY(build_conv_bn(e, U.d[i][0], pm, p + "0", p + "1", enc_c[i + 1], enc_c[i], 3, 1, enc_c[i], px(i + 1), res[i + 1]));
        CV_TRY(build_conv_bn(e, U.d[i][1], pm, p + "3", p + "4", enc_c[i + 1], enc_c[i + 1], 3, 1, enc_c[i + 1], px(i + 1), res[i + 1]));
    }
    // decoder: up_i consumes the deeper tensor (channels deep_c) and skip level (3 - i)
    //   transposed: up: deep_c -> deep_c/2 ; conv: cat(skip, up) = deep_c -> out_c -> out_c
    //   bilinear  : up keeps deep_c (== skip channels) ; conv: 2*deep_c -> mid = deep_c -> out_c
    int deep_c = U.c5;
    int up_c[4];                                      // channels of the up-sampled half of cat[lvl], by decoder stage
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;                       // skip level, output resolution res[lvl]
        const int skip_c = enc_c[lvl];
        const std::string p = "up" + std::to_string(i + 1);
        int cat_c, mid_c, out_c;
        if (!U.bilinear) {
            const float *w, *b;
            CV_TRY(need(pm, p + ".up.weight", {deep_c, deep_c / 2, 2, 2}, &w));
            CV_TRY(need(pm, p + ".up.bias", {deep_c / 2}, &b));
            known.push_back(p + ".up.weight"); known.push_back(p + ".up.bias");
            CV_TRY(U.upT[i].build_convT(p + ".up", dt, w, deep_c, deep_c / 2, b, px(lvl + 1)));
            U.upT[i].layer_id = e.register_layer(p + ".up");
            static const bool fast_on = [] { const char* v = std::getenv("CV_CONVT_FAST"); return !(v && v[0] == '0'); }();
            static const bool fast_256 = [] { const char* v = std::getenv("CV_CONVT_FAST_256"); return v && v[0] == '1'; }();
            // measured (r06_tuning.md section 7): up4.up (K = 128) 0.40 -> 0.36 ms per 64 boards; up3.up (K = 256: 128 KB of weights leave
            // room for four staged pixel rows per wave only) 0.255 -> 0.29 ms -- it stays on the generic tile unless CV_CONVT_FAST_256=1
            if (dt == kSplit && fast_on && (deep_c == 128 || (deep_c == 256 && fast_256))) {
                // split-f16 image of the (dy, dx, co) x ci matrix for convt2x2_lds: rows normalised to [0.5, 1) as ConvLayer rows are
                // (finish_layer), hi / lo halves, [row group of 128][k-step of 32][fragment 8][hi | lo][lane 64][8]; MFMA row i of
                // fragment f = row 32 (i/4) + 4 f + i%4 of the group
                const int cin = deep_c, cout = deep_c / 2, rows = 4 * cout;
                Engine::UNet::FastUp& F = U.fast_up[i];
                std::vector<int> rex((size_t)rows, 0);
                F.h_scale.assign((size_t)rows, 0.f); F.h_shift.assign((size_t)rows, 0.f);
                auto wat = [&](int r, int ci) { const int d = r / cout, co = r - d * cout; return w[((size_t)ci * cout + co) * 4 + d]; };
                for (int r = 0; r < rows; ++r) {
                    float mx = 0.f;
                    for (int ci = 0; ci < cin; ++ci) mx = std::max(mx, std::fabs(wat(r, ci)));
                    if (mx > 0.f) (void)std::frexp(mx, &rex[r]);
                    F.h_scale[r] = std::ldexp(1.f, rex[r]);
                    F.h_shift[r] = b[r % cout];
                }
                const int KS = cin / 32, CG = rows / 128;
                std::vector<_Float16> pk((size_t)CG * KS * 16 * 64 * 8);
                for (int cg = 0; cg < CG; ++cg)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int f = 0; f < 8; ++f)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int ii = lane & 15, q = lane >> 4;
                                const int r = cg * 128 + (ii / 4) * 32 + f * 4 + (ii % 4);
                                for (int j = 0; j < 8; ++j) {
                                    const float v = std::ldexp(wat(r, ks * 32 + q * 8 + j), -rex[r]);
                                    const _Float16 hi = (_Float16)v;
                                    const size_t at = ((((size_t)(cg * KS + ks) * 8 + f) * 2) * 64 + lane) * 8 + j;
                                    pk[at] = hi;
                                    pk[at + 64 * 8] = (_Float16)(v - (float)hi);
                                }
                            }
                CV_TRY(F.wpk.upload(pk.data(), pk.size() * sizeof(_Float16)));
                CV_TRY(F.scale.alloc((size_t)rows * sizeof(float), false));
                CV_TRY(F.shift.alloc((size_t)rows * sizeof(float), false));
                F.on = true;
            }
            cat_c = skip_c + deep_c / 2;
            out_c = skip_c;
            mid_c = out_c;
        } else {
            U.up_id[i] = e.register_layer(p + ".up");
            cat_c = skip_c + deep_c;
            mid_c = cat_c / 2;
            out_c = (i == 3) ? 64 : skip_c / 2;
        }
        up_c[i] = cat_c - skip_c;
        const std::string c = p + ".conv.double_conv.";
        double_conv_keys(known, c);
        U.u[i][0].keep_host_weights = dt != kF32;     // consumer of cat([skip, up]): the halves may end up with different exponents
        CV_TRY(build_conv_bn(e, U.u[i][0], pm, c + "0", c + "1", mid_c, cat_c, 3, 1, cat_c, px(lvl), res[lvl]));
        CV_TRY(build_conv_bn(e, U.u[i][1], pm, c + "3", c + "4", out_c, mid_c, 3, 1, mid_c, px(lvl), res[lvl]));
        deep_c = out_c;
    }
    {
        const float *w, *b;
        CV_TRY(need(pm, "outc.conv.weight", {1, 64, 1, 1}, &w));
        CV_TRY(need(pm, "outc.conv.bias", {1}, &b));
        known.push_back("outc.conv.weight"); known.push_back("outc.conv.bias");
        for (int i = 0; i < 64; ++i)
            if (!std::isfinite(w[i])) return fail(1, "outc.conv: non-finite weight in the state dict");
        if (!std::isfinite(b[0])) return fail(1, "outc.conv: non-finite bias in the state dict");
        CV_TRY(U.outc_w.upload(w, 64 * sizeof(float)));
        CV_TRY(U.outc_b.upload(b, sizeof(float)));
        U.outc_id = e.register_layer("outc.conv");
    }
    CV_TRY(reject_unknown_keys(pm, known, U.bilinear ? "UNet(3,1,bilinear=True)" : "UNet(3,1)"));

    // activation shapes (dedicated buffers: borders are zeroed once and stay zero); memory comes with unet_reserve
    U.in8.shape(256, 256, 8, dt);
    U.in8.fixed_exp = true;
    U.in8.exp = dt == kF32 ? 0 : kInputExp;
    U.a_inc0.shape(256, 256, 64, dt);
    U.acts = {&U.a_inc0};
    if (!U.fused_inc0) U.acts.push_back(&U.in8);     // the packed 8-channel copy of the input exists on the generic path only
    for (int lvl = 0; lvl < 4; ++lvl) {
        const int skip_c = enc_c[lvl];
        U.cat[lvl].shape(res[lvl], res[lvl], skip_c + up_c[3 - lvl], dt);
        U.cat[lvl].split_c = dt == kF32 ? 0 : skip_c;   // [skip | up-sampled]: one exponent per half
        U.pool[lvl].shape(res[lvl + 1], res[lvl + 1], skip_c, dt);
        U.pool[lvl].tie = &U.cat[lvl];               // max-pool of the skip half: same scale
        U.dmid[lvl].shape(res[lvl + 1], res[lvl + 1], enc_c[lvl + 1], dt);
        U.acts.push_back(&U.cat[lvl]); U.acts.push_back(&U.pool[lvl]); U.acts.push_back(&U.dmid[lvl]);
    }
    U.bott.shape(16, 16, U.c5, dt);
    U.acts.push_back(&U.bott);
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;
        U.umid[i].shape(res[lvl], res[lvl], U.u[i][0].cout, dt);
        U.uout[i].shape(res[lvl], res[lvl], U.u[i][1].cout, dt);
        U.acts.push_back(&U.umid[i]);
        if (!(i == 3 && U.fuse_head)) U.acts.push_back(&U.uout[i]);   // with the fused head the up4 output never reaches memory
    }

    // algorithmic multiply-accumulates per image (conv / conv-transpose / outc), true channel counts
    int64_t macs = 0;
    auto add = [&](const ConvLayer& L, int64_t out_pixels) { macs += L.macs_per_out_pixel() * out_pixels; };
    add(U.inc0, 65536); add(U.inc1, 65536);
    for (int i = 0; i < 4; ++i) { add(U.d[i][0], (int64_t)res[i + 1] * res[i + 1]); add(U.d[i][1], (int64_t)res[i + 1] * res[i + 1]); }
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;
        if (!U.bilinear) add(U.upT[i], (int64_t)res[lvl + 1] * res[lvl + 1]);
        add(U.u[i][0], (int64_t)res[lvl] * res[lvl]); add(U.u[i][1], (int64_t)res[lvl] * res[lvl]);
    }
    macs += 64LL * 65536;
    U.macs = macs;

    e.unet = std::move(m);
    // range calibration of the f16-based engines on two images (Activation, Engine::calibrate)
    Status st = unet_reserve(e, 2);
    if (st.ok() && dt != kF32 && calibration_enabled()) {
        std::vector<float> host;
        calibration_images(host);
        DeviceBuffer xin, lout;
        st = xin.upload(host.data(), host.size() * sizeof(float));
        if (st.ok()) st = lout.alloc((size_t)2 * 65536 * sizeof(float), false);
        if (st.ok())
            st = e.calibrate(e.unet->acts, [&]() -> Status {                   // the statistics accumulate over the chunks
                for (int off = 0; off < 2; off += e.unet->cap) {
                    const int c = std::min(e.unet->cap, 2 - off);
                    CV_TRY(unet_chunk(e, (const float*)xin.ptr + (size_t)off * 3 * 65536, false, c, (float*)lout.ptr + (size_t)off * 65536,
                                      nullptr, 0.5f, nullptr));
                }
                return Status();
            }, nullptr, "UNet");
        if (st.ok()) {                                                          // f16 engines: rounding-bias pass on the same two images
            Engine::UNet& U = *e.unet;
            std::vector<ConvLayer*> layers{&U.inc0, &U.inc1};
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) { layers.push_back(&U.d[i][j]); layers.push_back(&U.u[i][j]); }
            st = e.calibrate_rounding_bias(layers, [&]() -> Status {
                for (int off = 0; off < 2; off += e.unet->cap) {
                    const int c = std::min(e.unet->cap, 2 - off);
                    CV_TRY(unet_chunk(e, (const float*)xin.ptr + (size_t)off * 3 * 65536, false, c, (float*)lout.ptr + (size_t)off * 65536,
                                      nullptr, 0.5f, nullptr));
                }
                return Status();
            }, nullptr);
        }
        if (st.ok()) {
            hipError_t he = device_synchronize();
            if (he != hipSuccess) st = hip_fail(he, "UNet calibration");
        }
    }
    if (!st.ok()) e.unet.reset();
    return st;
}

// (re)allocate the workspace for min(n, chunk) images; exponents survive, module taps are rebuilt
static Status unet_reserve(Engine& e, int n) {
    Engine::UNet& U = *e.unet;
    const int want = std::min(U.max_cap, std::max(n, 1));
    if (want <= U.cap) return Status();
    CV_HIP(device_synchronize());                  // nothing may still read the buffers about to be replaced
    e.graph_invalidate();                            // captured launches hold the old buffers
    CV_TRY(Activation::reserve_all(U.acts, want));
    U.cap = want;
    const int S = want;
    const int enc_c[5] = {64, 128, 256, 512, U.c5};
    U.taps.clear();
    // module-name taps for cv_get_activation (names follow the reference state-dict prefixes)
    if (!U.fused_inc0) U.taps["input"] = U.in8.ref(S, 0, 8);
    if (!U.fused_inc) U.taps["inc.double_conv.2"] = U.a_inc0.ref(S);     // fused: the tensor exists only inside inc.double_conv.3's kernel
    U.taps["inc.double_conv.5"] = U.cat[0].ref(S, 0, 64);
    U.taps["inc"] = U.taps["inc.double_conv.5"];
    for (int i = 0; i < 4; ++i) {
        const std::string p = "down" + std::to_string(i + 1);
        U.taps[p + ".maxpool_conv.0"] = U.pool[i].ref(S);
        U.taps[p + ".maxpool_conv.1.double_conv.2"] = U.dmid[i].ref(S);
        TensorRef out = (i < 3) ? U.cat[i + 1].ref(S, 0, enc_c[i + 1]) : U.bott.ref(S);
        U.taps[p + ".maxpool_conv.1.double_conv.5"] = out;
        U.taps[p] = out;
    }
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;
        const std::string p = "up" + std::to_string(i + 1);
        U.taps[p + ".up"] = U.cat[lvl].ref(S, enc_c[lvl], U.cat[lvl].C - enc_c[lvl]);
        U.taps[p + ".conv.double_conv.2"] = U.umid[i].ref(S);
        if (!(i == 3 && U.fuse_head)) {
            U.taps[p + ".conv.double_conv.5"] = U.uout[i].ref(S);
            U.taps[p] = U.uout[i].ref(S);
        }
    }
    return Status();
}

int64_t unet_macs(Engine& e) { return e.unet ? e.unet->macs : 0; }

Status unet_activation(Engine& e, const std::string& name, TensorRef* out) {
    if (!e.unet) return fail(3, "UNet not loaded");
    auto it = e.unet->taps.find(name);
    if (it == e.unet->taps.end()) return fail(1, "unknown UNet activation '" + name + "'");
    *out = it->second;
    out->exp = static_cast<Activation*>(out->owner)->exp_of(out->Coff);
    out->N = e.unet->last_n;
    return Status();
}

static Status unet_chunk(Engine& e, const void* x, bool x_u8, int n, float* logits, uint8_t* mask, float thr,
                         hipStream_t s) {
    Engine::UNet& U = *e.unet;
    U.last_n = n;
    e.ws_slot = 0;
    const int dt = e.dt;
    const int enc_c[5] = {64, 128, 256, 512, U.c5};
    auto timed = [&](const char* name, hipError_t err) -> Status {
        if (e.profiling) e.prof_end(s);
        if (err != hipSuccess) return hip_fail(err, name);
        return Status();
    };
    auto begin = [&](const char* name, double bytes = 0) { if (e.profiling) e.prof_begin(name, false, 0, s, bytes); };
    const double esz = dtype_size(dt);

    const TensorRef pool0 = U.pool[0].ref(n);
    bool inc_done = false;
    if (U.fused_inc && !e.calibrating) {
        // inc.double_conv.0 produced inside inc.double_conv.3's kernel: the tensor between them never reaches memory.  (Calibration
        // runs the layers apart so that the intermediate's exponent is measured; launches the halo tile does not take fall back.)
        CV_TRY(U.inc0.set_exps(kInputExp, U.a_inc0.exp, s));
        const Engine::Fuse0 f0{x, x_u8, U.inc0_wpk2.ptr, (const float*)U.inc0.scale.ptr, (const float*)U.inc0.shift.ptr,
                               std::ldexp(1.f, -kInputExp), 27.0 * 64};
        Status st = e.run_conv(U.inc1, U.a_inc0.ref(n), U.cat[0].ref(n, 0, 64), nullptr, true, s, nullptr, &pool0, &f0);
        if (st.ok()) inc_done = true;
        else if (st.code != Engine::kNotFused) return st;
    }
    if (inc_done) {
    } else if (U.fused_inc0) {
        // first layer straight from the caller's image: no packed copy of the input, no separate packing kernel
        CV_TRY(U.inc0.set_exps(kInputExp, U.a_inc0.exp, s));
        if (e.profiling) e.prof_begin(U.inc0.name, true, 27.0 * 64 * 65536 * n, s, (double)n * 65536 * ((x_u8 ? 3 : 12) + 64 * esz) + 27 * 64 * esz);
        CV_TRY(timed("inc0_mfma", inc0_mfma(dt, x, x_u8, n, U.inc0_wpk.ptr, (const float*)U.inc0.scale.ptr, (const float*)U.inc0.shift.ptr,
                                            kInputExp, U.a_inc0.ref(n), e.guard_ptr(), U.inc0.layer_id, s)));
        if (e.calibrating) CV_TRY(e.measure(U.a_inc0.ref(n), s));
    } else {
        begin("pack_input", (double)n * 65536 * ((x_u8 ? 3 : 12) + 8 * esz));
        if (x_u8) CV_TRY(timed("pack_hwc3_u8", pack_hwc3_u8(dt, (const uint8_t*)x, U.in8.ref(n, 0, 8), s)));
        else      CV_TRY(timed("pack_nchw_f32", pack_nchw_f32(dt, (const float*)x, 3, U.in8.ref(n, 0, 8), e.guard_ptr(), s)));
        CV_TRY(e.run_conv(U.inc0, U.in8.ref(n, 0, 8), U.a_inc0.ref(n), nullptr, true, s));
    }
    // every encoder level's second conv also emits its 2x2 max-pool (fused into the epilogue where the halo kernel runs)
    if (!inc_done) CV_TRY(e.run_conv(U.inc1, U.a_inc0.ref(n), U.cat[0].ref(n, 0, 64), nullptr, true, s, nullptr, &pool0));
    for (int i = 0; i < 4; ++i) {
        CV_TRY(e.run_conv(U.d[i][0], U.pool[i].ref(n), U.dmid[i].ref(n), nullptr, true, s));
        TensorRef out = (i < 3) ? U.cat[i + 1].ref(n, 0, enc_c[i + 1]) : U.bott.ref(n);
        const TensorRef pool_next = (i < 3) ? U.pool[i + 1].ref(n) : TensorRef();
        CV_TRY(e.run_conv(U.d[i][1], U.dmid[i].ref(n), out, nullptr, true, s, nullptr, i < 3 ? &pool_next : nullptr));
    }
    TensorRef deep = U.bott.ref(n);
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;
        TensorRef up = U.cat[lvl].ref(n, enc_c[lvl], U.cat[lvl].C - enc_c[lvl]);
        static const int fast_min = [] { const char* v = std::getenv("CV_CONVT_FAST_MIN"); return v && *v ? std::atoi(v) : 8; }();
        if (!U.bilinear && U.fast_up[i].on && !e.calibrating && n >= fast_min) {
            Engine::UNet::FastUp& F = U.fast_up[i];
            if (F.in_exp != deep.exp || F.out_exp != up.exp) {          // fold the tensor exponents (ConvLayer::set_exps)
                if (capture_flag()) return fail(1, "transposed-conv constants re-folded during graph capture");
                CV_HIP(hipStreamSynchronize(s));
                std::vector<float> sc(F.h_scale.size()), sh(sc.size());
                for (size_t k = 0; k < sc.size(); ++k) {
                    sc[k] = std::ldexp(F.h_scale[k], deep.exp - up.exp);
                    sh[k] = std::ldexp(F.h_shift[k], -up.exp);
                    if (!std::isfinite(sc[k]) || !std::isfinite(sh[k])) return fail(1, U.upT[i].name + ": range factors leave the f32 range");
                }
                CV_HIP(sync_memcpy(F.scale.ptr, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
                CV_HIP(sync_memcpy(F.shift.ptr, sh.data(), sh.size() * sizeof(float), hipMemcpyHostToDevice));
                F.in_exp = deep.exp; F.out_exp = up.exp;
            }
            if (e.profiling) {
                const double ipx = (double)n * deep.H * deep.W;
                e.prof_begin(U.upT[i].name, true, ipx * (double)U.upT[i].macs_per_out_pixel(), s,
                             ipx * deep.C * esz + 4.0 * ipx * up.C * esz + (double)deep.C * up.C * 4 * esz);
                e.prof.back().kernel = "convt2x2_lds_kernel<" + std::to_string(deep.C) + ",split>";
            }
            const hipError_t err = convt2x2_lds(deep, F.wpk.ptr, (const float*)F.scale.ptr, (const float*)F.shift.ptr, up, e.guard_ptr(),
                                                U.upT[i].layer_id, s);
            if (e.profiling) e.prof_end(s);
            if (err != hipSuccess) return hip_fail(err, "convt2x2_lds");
        } else if (!U.bilinear) {
            CV_TRY(e.run_conv(U.upT[i], deep, up, nullptr, false, s));
        } else {
            begin("upsample_bilinear2x", (double)n * deep.H * deep.W * deep.C * esz * 5.0);     // 1 read + 4 written elements
            CV_TRY(timed("upsample_bilinear2x", upsample_bilinear2x(dt, deep, up, e.guard_ptr(), U.up_id[i], s)));
            if (e.calibrating) CV_TRY(e.measure(up, s));
        }
        CV_TRY(e.run_conv(U.u[i][0], U.cat[lvl].ref(n), U.umid[i].ref(n), nullptr, true, s));
        if (i == 3 && U.fuse_head) {
            // OutConv (1x1, 64 -> 1, + bias) and the sigmoid/threshold mask ride in the epilogue of the last 3x3 conv:
            // the 64-channel full-resolution tensor is never written or re-read.
            const Engine::Head head{(const float*)U.outc_w.ptr, (const float*)U.outc_b.ptr, logits, mask, thr};
            CV_TRY(e.run_conv(U.u[i][1], U.umid[i].ref(n), U.uout[i].ref(n), nullptr, true, s, &head));
            return Status();
        }
        CV_TRY(e.run_conv(U.u[i][1], U.umid[i].ref(n), U.uout[i].ref(n), nullptr, true, s));
        deep = U.uout[i].ref(n);
    }
    begin("outc_1x1", (double)n * 65536 * (64 * esz + 4 + (mask ? 1 : 0)));
    CV_TRY(timed("outc_1x1", outc_1x1(dt, deep, (const float*)U.outc_w.ptr, (const float*)U.outc_b.ptr, logits,
                                      mask, thr, e.guard_ptr(), U.outc_id, s)));
    return Status();
}

Status unet_forward(Engine& e, const void* x, bool x_u8, int batch, float* logits, uint8_t* mask, float thr,
                    hipStream_t s) {
    if (!e.unet) return fail(3, "UNet weights not loaded (call cv_load_unet first)");
    if (batch < 0 || (batch > 0 && (!x || !logits))) return fail(1, "cv_unet_forward: null tensor or negative batch");
    if (batch == 0) return Status();
    CV_TRY(e.order_forward(0, s));
    CV_TRY(unet_reserve(e, batch));
    if (batch <= e.unet->cap && batch <= 8) {            // one small chunk: the launch sequence replays as a hipGraph
        Engine::GraphKey key;
        key.model = 0; key.n = batch; key.flags = x_u8 ? 1 : 0; key.x = x; key.out = logits; key.mask = mask;
        std::memcpy(&key.thr_bits, &thr, sizeof(float));
        e.unet->last_n = batch;                          // a graph replay skips unet_chunk's host side: what cv_get_activation reports
        e.ws_slot = 0;                                   // must not depend on whether this call replayed
        return e.run_graphed(key, s, [&](hipStream_t st) { return unet_chunk(e, x, x_u8, batch, logits, mask, thr, st); });
    }
    const size_t in_stride = (size_t)3 * 256 * 256 * (x_u8 ? 1 : 4);
    for (int off = 0; off < batch; off += e.unet->cap) {
        const int n = std::min(e.unet->cap, batch - off);
        CV_TRY(unet_chunk(e, (const char*)x + (size_t)off * in_stride, x_u8, n, logits + (size_t)off * 65536,
                          mask ? mask + (size_t)off * 65536 : nullptr, thr, s));
    }
    return Status();
}

}  // namespace cv
