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

static Status build_conv_bn(ConvLayer& L, int dt, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw = 0) {
    const float* w;
    CV_TRY(need(pm, conv_key + ".weight", {cout, cin, k, k}, &w));
    std::vector<float> sc, sh;
    CV_TRY(bn_fold(pm, bn_key, cout, sc, sh));
    return L.build_conv(conv_key, dt, w, cout, cin, k, stride, sc.data(), sh.data(), cinPad, pixels, out_hw);
}
Status build_conv_bn_public(ConvLayer& L, int dt, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw) {
    return build_conv_bn(L, dt, pm, conv_key, bn_key, cout, cin, k, stride, cinPad, pixels, out_hw);
}

Status unet_load(Engine& e, const ParamMap& pm) {
    auto m = std::make_unique<Engine::UNet>();
    Engine::UNet& U = *m;
    const int dt = e.dt;
    U.bilinear = pm.find("up1.up.weight") == pm.end();
    {   // CV_FUSE_HEAD=0 keeps OutConv a separate kernel (then the up4 output tensor can be read back)
        const char* v = std::getenv("CV_FUSE_HEAD");
        U.fuse_head = !(v && v[0] == '0');
    }
    U.cap = e.unet_chunk;
    const int S = U.cap;
    const int f = U.bilinear ? 2 : 1;
    U.c5 = 1024 / f;
    const int enc_c[5] = {64, 128, 256, 512, U.c5};
    const int res[5] = {256, 128, 64, 32, 16};
    auto px = [&](int level) { return (int64_t)S * res[level] * res[level]; };

    // encoder
    CV_TRY(build_conv_bn(U.inc0, dt, pm, "inc.double_conv.0", "inc.double_conv.1", 64, 3, 3, 1, 8, px(0), 256));
    CV_TRY(build_conv_bn(U.inc1, dt, pm, "inc.double_conv.3", "inc.double_conv.4", 64, 64, 3, 1, 64, px(0), 256));
    for (int i = 0; i < 4; ++i) {
        const std::string p = "down" + std::to_string(i + 1) + ".maxpool_conv.1.double_conv.";
        CV_TRY(build_conv_bn(U.d[i][0], dt, pm, p + "0", p + "1", enc_c[i + 1], enc_c[i], 3, 1, enc_c[i], px(i + 1), res[i + 1]));
        CV_TRY(build_conv_bn(U.d[i][1], dt, pm, p + "3", p + "4", enc_c[i + 1], enc_c[i + 1], 3, 1, enc_c[i + 1], px(i + 1), res[i + 1]));
    }
    // decoder: up_i consumes the deeper tensor (channels deep_c) and skip level (3 - i)
    //   transposed: up: deep_c -> deep_c/2 ; conv: cat(skip, up) = deep_c -> out_c -> out_c
    //   bilinear  : up keeps deep_c (== skip channels) ; conv: 2*deep_c -> mid = deep_c -> out_c
    int deep_c = U.c5;
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;                       // skip level, output resolution res[lvl]
        const int skip_c = enc_c[lvl];
        const std::string p = "up" + std::to_string(i + 1);
        int cat_c, mid_c, out_c;
        if (!U.bilinear) {
            const float *w, *b;
            CV_TRY(need(pm, p + ".up.weight", {deep_c, deep_c / 2, 2, 2}, &w));
            CV_TRY(need(pm, p + ".up.bias", {deep_c / 2}, &b));
            CV_TRY(U.upT[i].build_convT(p + ".up", dt, w, deep_c, deep_c / 2, b, px(lvl + 1)));
            cat_c = skip_c + deep_c / 2;
            out_c = skip_c;
            mid_c = out_c;
        } else {
            cat_c = skip_c + deep_c;
            mid_c = cat_c / 2;
            out_c = (i == 3) ? 64 : skip_c / 2;
        }
        const std::string c = p + ".conv.double_conv.";
        CV_TRY(build_conv_bn(U.u[i][0], dt, pm, c + "0", c + "1", mid_c, cat_c, 3, 1, cat_c, px(lvl), res[lvl]));
        CV_TRY(build_conv_bn(U.u[i][1], dt, pm, c + "3", c + "4", out_c, mid_c, 3, 1, mid_c, px(lvl), res[lvl]));
        deep_c = out_c;
    }
    {
        const float *w, *b;
        CV_TRY(need(pm, "outc.conv.weight", {1, 64, 1, 1}, &w));
        CV_TRY(need(pm, "outc.conv.bias", {1}, &b));
        CV_TRY(U.outc_w.upload(w, 64 * sizeof(float)));
        CV_TRY(U.outc_b.upload(b, sizeof(float)));
    }

    // activations (dedicated buffers: borders are zeroed once and stay zero)
    CV_TRY(U.in8.create(S, 256, 256, 8, dt));
    CV_TRY(U.a_inc0.create(S, 256, 256, 64, dt));
    for (int lvl = 0; lvl < 4; ++lvl) {
        const int skip_c = enc_c[lvl];
        const int up_c = U.u[3 - lvl][0].cin - skip_c;
        CV_TRY(U.cat[lvl].create(S, res[lvl], res[lvl], skip_c + up_c, dt));
        CV_TRY(U.pool[lvl].create(S, res[lvl + 1], res[lvl + 1], skip_c, dt));
        if (lvl < 3) CV_TRY(U.dmid[lvl].create(S, res[lvl + 1], res[lvl + 1], enc_c[lvl + 1], dt));
    }
    CV_TRY(U.dmid[3].create(S, 16, 16, U.c5, dt));
    CV_TRY(U.bott.create(S, 16, 16, U.c5, dt));
    for (int i = 0; i < 4; ++i) {
        const int lvl = 3 - i;
        CV_TRY(U.umid[i].create(S, res[lvl], res[lvl], U.u[i][0].cout, dt));
        CV_TRY(U.uout[i].create(S, res[lvl], res[lvl], U.u[i][1].cout, dt));
    }

    // module-name taps for cv_get_activation (names follow the reference state-dict prefixes)
    U.taps["input"] = U.in8.ref(S, 0, 8);
    U.taps["inc.double_conv.2"] = U.a_inc0.ref(S);
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
        if (!(i == 3 && U.fuse_head)) {                  // with the fused head the up4 output never reaches memory
            U.taps[p + ".conv.double_conv.5"] = U.uout[i].ref(S);
            U.taps[p] = U.uout[i].ref(S);
        }
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
    return Status();
}

int64_t unet_macs(Engine& e) { return e.unet ? e.unet->macs : 0; }

Status unet_activation(Engine& e, const std::string& name, TensorRef* out) {
    if (!e.unet) return fail(3, "UNet not loaded");
    auto it = e.unet->taps.find(name);
    if (it == e.unet->taps.end()) return fail(1, "unknown UNet activation '" + name + "'");
    *out = it->second;
    out->N = e.unet->last_n;
    return Status();
}

static Status unet_chunk(Engine& e, const void* x, bool x_u8, int n, float* logits, uint8_t* mask, float thr,
                         hipStream_t s) {
    Engine::UNet& U = *e.unet;
    U.last_n = n;
    const int dt = e.dt;
    const int enc_c[5] = {64, 128, 256, 512, U.c5};
    auto timed = [&](const char* name, hipError_t err) -> Status {
        if (e.profiling) e.prof_end(s);
        if (err != hipSuccess) return hip_fail(err, name);
        return Status();
    };
    auto begin = [&](const char* name) { if (e.profiling) e.prof_begin(name, false, 0, s); };

    begin("pack_input");
    if (x_u8) CV_TRY(timed("pack_hwc3_u8", pack_hwc3_u8(dt, (const uint8_t*)x, U.in8.ref(n, 0, 8), s)));
    else      CV_TRY(timed("pack_nchw_f32", pack_nchw_f32(dt, (const float*)x, 3, U.in8.ref(n, 0, 8), s)));

    CV_TRY(e.run_conv(U.inc0, U.in8.ref(n, 0, 8), U.a_inc0.ref(n), nullptr, true, s));
    // every encoder level's second conv also emits its 2x2 max-pool (fused into the epilogue where the halo kernel runs)
    const TensorRef pool0 = U.pool[0].ref(n);
    CV_TRY(e.run_conv(U.inc1, U.a_inc0.ref(n), U.cat[0].ref(n, 0, 64), nullptr, true, s, nullptr, &pool0));
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
        if (!U.bilinear) {
            CV_TRY(e.run_conv(U.upT[i], deep, up, nullptr, false, s));
        } else {
            begin("upsample_bilinear2x");
            CV_TRY(timed("upsample_bilinear2x", upsample_bilinear2x(dt, deep, up, s)));
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
    begin("outc_1x1");
    CV_TRY(timed("outc_1x1", outc_1x1(dt, deep, (const float*)U.outc_w.ptr, (const float*)U.outc_b.ptr, logits,
                                      mask, thr, s)));
    return Status();
}

Status unet_forward(Engine& e, const void* x, bool x_u8, int batch, float* logits, uint8_t* mask, float thr,
                    hipStream_t s) {
    if (!e.unet) return fail(3, "UNet weights not loaded (call cv_load_unet first)");
    if (batch < 0 || (batch > 0 && (!x || !logits))) return fail(1, "cv_unet_forward: null tensor or negative batch");
    const size_t in_stride = (size_t)3 * 256 * 256 * (x_u8 ? 1 : 4);
    for (int off = 0; off < batch; off += e.unet->cap) {
        const int n = std::min(e.unet->cap, batch - off);
        CV_TRY(unet_chunk(e, (const char*)x + (size_t)off * in_stride, x_u8, n, logits + (size_t)off * 65536,
                          mask ? mask + (size_t)off * 65536 : nullptr, thr, s));
    }
    return Status();
}

}  // namespace cv
