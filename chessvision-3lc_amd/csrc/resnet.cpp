// resnet.cpp -- timm ResNet-18 (in_chans=1, num_classes=13) plan: packing + forward schedule.
//
// Module tree per notebooks/model-summary.ipynb (reference) / SURVEY.md Appendix B:
//   conv1 7x7 s2 p3 -> bn1 -> ReLU -> maxpool 3x3 s2 p1 -> layer1..4 (2 BasicBlocks each; first block of
//   layer2-4 has stride 2 and a [conv1x1 s2, BN] shortcut) -> global avg pool -> fc 512 -> 13.
// BasicBlock = conv1-bn1-ReLU-conv2-bn2-(+shortcut)-ReLU: both BNs, the residual add and both ReLUs are conv
// epilogues here; the squares of many boards are batched so the 2x2 / 4x4 stages still form large GEMMs.
#include <cmath>

#include "engine.h"
#include "models.h"
#include "pointwise.h"

namespace cv {

Status build_conv_bn_public(ConvLayer& L, int dt, const ParamMap& pm, const std::string& conv_key,
                            const std::string& bn_key, int cout, int cin, int k, int stride, int cinPad,
                            int64_t pixels, int out_hw);


static Status need2(const ParamMap& pm, const std::string& key, std::vector<int64_t> shape, const float** out) {
    auto it = pm.find(key);
    if (it == pm.end()) return fail(1, "state dict is missing key '" + key + "'");
    if (it->second.shape != shape) return fail(1, "state dict key '" + key + "' has an unexpected shape");
    *out = it->second.data;
    return Status();
}

Status resnet_load(Engine& e, const ParamMap& pm) {
    auto m = std::make_unique<Engine::ResNet>();
    Engine::ResNet& R = *m;
    const int dt = e.dt;
    // the f32 engine materialises the 32x32x64 stem output (34 x 34 x 64 x 4 B per square with its border): keep that
    // tensor under the 4 GiB the 32-bit DMA offsets address
    R.cap = e.resnet_chunk;
    if (dt == kF32)
        while ((uint64_t)R.cap * 34 * 34 * 64 * 4 >= (1ull << 32) && R.cap > 1) R.cap /= 2;
    const int S = R.cap;

    {   // stem: conv1 (64,1,7,7) + bn1
        const float *w, *g, *b, *mu, *var;
        CV_TRY(need2(pm, "conv1.weight", {64, 1, 7, 7}, &w));
        CV_TRY(need2(pm, "bn1.weight", {64}, &g));
        CV_TRY(need2(pm, "bn1.bias", {64}, &b));
        CV_TRY(need2(pm, "bn1.running_mean", {64}, &mu));
        CV_TRY(need2(pm, "bn1.running_var", {64}, &var));
        std::vector<float> sc(64), sh(64);
        for (int i = 0; i < 64; ++i) { sc[i] = g[i] / std::sqrt(var[i] + 1e-5f); sh[i] = b[i] - mu[i] * sc[i]; }
        CV_TRY(R.stem_w.upload(w, 64 * 49 * sizeof(float)));
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
                                const float v = (ky < 7 && j < 7) ? w[ch * 49 + ky * 7 + j] : 0.f;
                                const _Float16 hi = (_Float16)v;
                                pk[((((size_t)hl * 2 + ks) * 4 + f) * 64 + lane) * 8 + j] = hl ? (_Float16)(v - (float)hi) : hi;
                            }
                        }
            CV_TRY(R.stem_wpk.upload(pk.data(), pk.size() * sizeof(_Float16)));
        }
        CV_TRY(R.stem_scale.upload(sc.data(), 64 * sizeof(float)));
        CV_TRY(R.stem_shift.upload(sh.data(), 64 * sizeof(float)));
    }
    if (dt == kF32) CV_TRY(R.stem_out.create(S, 32, 32, 64, dt));   // other engines fuse stem + pool
    CV_TRY(R.pool_out.create(S, 16, 16, 64, dt));
    if (dt == kF32) R.taps["act1"] = R.stem_out.ref(S);
    R.taps["maxpool"] = R.pool_out.ref(S);

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
            CV_TRY(build_conv_bn_public(B.conv1, dt, pm, p + ".conv1", p + ".bn1", w, cin, 3, stride, cin, px, res[l]));
            CV_TRY(build_conv_bn_public(B.conv2, dt, pm, p + ".conv2", p + ".bn2", w, w, 3, 1, w, px, res[l]));
            B.has_down = (stride != 1 || cin != w);
            if (B.has_down) {
                CV_TRY(build_conv_bn_public(B.down, dt, pm, p + ".downsample.0", p + ".downsample.1", w, cin, 1, stride, cin, px, res[l]));
                CV_TRY(B.sc.create(S, res[l], res[l], w, dt));
                macs += (int64_t)cin * w * res[l] * res[l];
            }
            CV_TRY(B.mid.create(S, res[l], res[l], w, dt));
            CV_TRY(B.out.create(S, res[l], res[l], w, dt));
            macs += ((int64_t)cin * 9 * w + (int64_t)w * 9 * w) * res[l] * res[l];
            R.taps[p + ".act1"] = B.mid.ref(S);
            R.taps[p] = B.out.ref(S);
            if (B.has_down) R.taps[p + ".downsample"] = B.sc.ref(S);
            cin = w;
        }
        R.taps["layer" + std::to_string(l + 1)] = R.blocks[l * 2 + 1].out.ref(S);
    }
    {
        const float *w, *b;
        CV_TRY(need2(pm, "fc.weight", {13, 512}, &w));
        CV_TRY(need2(pm, "fc.bias", {13}, &b));
        CV_TRY(R.fc_w.upload(w, 13 * 512 * sizeof(float)));
        CV_TRY(R.fc_b.upload(b, 13 * sizeof(float)));
        macs += 13 * 512;
    }
    R.macs = macs;
    e.resnet = std::move(m);
    return Status();
}

int64_t resnet_macs(Engine& e) { return e.resnet ? e.resnet->macs : 0; }

Status resnet_activation(Engine& e, const std::string& name, TensorRef* out) {
    if (!e.resnet) return fail(3, "ResNet-18 not loaded");
    auto it = e.resnet->taps.find(name);
    if (it == e.resnet->taps.end()) return fail(1, "unknown ResNet activation '" + name + "'");
    *out = it->second;
    out->N = e.resnet->last_n;
    return Status();
}

static Status resnet_chunk(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s) {
    Engine::ResNet& R = *e.resnet;
    R.last_n = n;
    const int dt = e.dt;
    auto begin = [&](const char* name, double macs) { if (e.profiling) e.prof_begin(name, false, macs, s); };
    auto end = [&](const char* name, hipError_t err) -> Status {
        if (e.profiling) e.prof_end(s);
        if (err != hipSuccess) return hip_fail(err, name);
        return Status();
    };
    if (dt == kF32) {
        begin("stem7x7", 49.0 * 64 * 1024 * n);
        CV_TRY(end("stem7x7", stem7x7(dt, x, x_u8, n, (const float*)R.stem_w.ptr, (const float*)R.stem_scale.ptr,
                                       (const float*)R.stem_shift.ptr, R.stem_out.ref(n), s)));
        begin("maxpool3x3s2", 0);
        CV_TRY(end("maxpool3x3s2", maxpool3x3s2(dt, R.stem_out.ref(n), R.pool_out.ref(n), s)));
    } else {
        begin("stem7x7+maxpool (mfma)", 49.0 * 64 * 1024 * n);
        CV_TRY(end("stem_pool_mfma", stem_pool_mfma(dt, x, x_u8, n, R.stem_wpk.ptr, (const float*)R.stem_scale.ptr,
                                                     (const float*)R.stem_shift.ptr, R.pool_out.ref(n), s)));
    }
    TensorRef cur = R.pool_out.ref(n);
    for (int i = 0; i < 8; ++i) {
        Engine::ResNet::Block& B = R.blocks[i];
        TensorRef shortcut = cur;
        if (B.has_down) {
            CV_TRY(e.run_conv(B.down, cur, B.sc.ref(n), nullptr, false, s));
            shortcut = B.sc.ref(n);
        }
        CV_TRY(e.run_conv(B.conv1, cur, B.mid.ref(n), nullptr, true, s));
        CV_TRY(e.run_conv(B.conv2, B.mid.ref(n), B.out.ref(n), &shortcut, true, s));
        cur = B.out.ref(n);
    }
    begin("head_avgpool_fc", 13.0 * 512 * n);
    CV_TRY(end("head_avgpool_fc", head_avgpool_fc(dt, cur, (const float*)R.fc_w.ptr, (const float*)R.fc_b.ptr, out,
                                                   softmax ? 1 : 0, s)));
    return Status();
}

Status resnet_forward(Engine& e, const void* x, bool x_u8, int n, float* out, bool softmax, hipStream_t s) {
    if (!e.resnet) return fail(3, "ResNet-18 weights not loaded (call cv_load_resnet18 first)");
    if (n < 0 || (n > 0 && (!x || !out))) return fail(1, "cv_resnet18_forward: null tensor or negative batch");
    const size_t in_stride = (size_t)64 * 64 * (x_u8 ? 1 : 4);
    for (int off = 0; off < n; off += e.resnet->cap) {
        const int c = std::min(e.resnet->cap, n - off);
        CV_TRY(resnet_chunk(e, (const char*)x + (size_t)off * in_stride, x_u8, c, out + (size_t)off * 13, softmax, s));
    }
    return Status();
}

}  // namespace cv
