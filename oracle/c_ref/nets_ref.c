/* nets_ref.c -- the two COMPOSED networks of the ChessVision hot path in plain C, float64 throughout.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  An implementation of the whole forward passes that shares no code with
 * oracle/unet_ref.py / oracle/resnet_ref.py (torch nn.Modules) nor with the HIP product: it walks a flat state dict by key name
 * and composes loops.  What it pins that the per-op cross-check (ops_ref.c) cannot: the ORDER OF COMPOSITION -- concatenation as
 * [skip, up-sampled] (channel order of the decoder's first conv), where the padding sits, which tensor a residual add takes,
 * stem -> pool order, stride placement, BN before / after the add.  Every activation is held in double and every sum is a
 * double sum, so the result is the mathematical function of the checkpoint to ~1e-12; fp32 implementations must land within
 * their own rounding (torch CPU ~1e-5, the f16x3 engine ~1e-4 on logits).
 *
 * Sources restated (reference = /root/reference; none of its files can be executed here):
 *   UNet(n_channels=3, n_classes=1, bilinear)   ctor at chessvision/core.py:88, scripts/train/train_unet.py:461-465; module tree
 *       of the un-vendored Pytorch-UNet submodule (.gitmodules:1-4) = the upstream milesial layout, SURVEY.md Appendix A:
 *       inc=DoubleConv(3,64); down1..4=MaxPool2d(2)+DoubleConv (128,256,512,1024/f); up1..4=Up(1024,512/f) ... Up(128,64);
 *       Up.forward(x1, x2): x1 = up(x1); pad x1 to x2's size; x = cat([x2, x1], dim=1); DoubleConv(x);  outc = Conv2d(64,1,1)
 *       (f = 2 and DoubleConv(in, out, mid=in/2) when bilinear, Upsample(scale 2, bilinear, align_corners=True))
 *   timm resnet18(num_classes=13, in_chans=1)   built at chessvision/utils.py:32-39; module order and shapes pinned by
 *       notebooks/model-summary.ipynb:31-124,142-236: conv1 7x7/2 p3 -> bn1 -> act1 -> maxpool 3x3/2 p1 -> layer1..4 (two
 *       BasicBlocks each: conv1-bn1-act1-conv2-bn2, + shortcut (downsample = conv1x1/stride + BN at layer2-4.0), act2)
 *       -> global average pool -> fc(512, 13)
 * Built with OpenMP when available (make), single-threaded otherwise: same results either way (no reduction is parallelised).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const char* name;
    const float* data;
    int ndim;
    long long shape[4];
} ref_param_t;

typedef struct {
    double* v;
    int C, H, W;
} tens_t;

typedef struct {
    const ref_param_t* p;
    int n;
    char* err;
    int errcap;
    int failed;
} dict_t;

static void fail(dict_t* d, const char* what, const char* key) {
    if (!d->failed && d->err && d->errcap > 0) snprintf(d->err, (size_t)d->errcap, "%s '%s'", what, key);
    d->failed = 1;
}

static const float* get(dict_t* d, const char* key, int ndim, long long s0, long long s1, long long s2, long long s3) {
    const long long want[4] = {s0, s1, s2, s3};
    for (int i = 0; i < d->n; ++i) {
        if (strcmp(d->p[i].name, key) != 0) continue;
        if (d->p[i].ndim != ndim) { fail(d, "rank mismatch for", key); return NULL; }
        for (int k = 0; k < ndim; ++k)
            if (d->p[i].shape[k] != want[k]) { fail(d, "shape mismatch for", key); return NULL; }
        return d->p[i].data;
    }
    fail(d, "missing key", key);
    return NULL;
}

static tens_t talloc(int C, int H, int W) {
    tens_t t;
    t.C = C; t.H = H; t.W = W;
    t.v = (double*)calloc((size_t)C * H * W, sizeof(double));
    return t;
}
static void tfree(tens_t* t) { free(t->v); t->v = NULL; }

/* y = conv2d(x, w[, bias]); w: (Cout, Cin, K, K) float */
static tens_t conv2d(const tens_t x, const float* w, const float* bias, int Cout, int K, int stride, int pad) {
    const int Ho = (x.H + 2 * pad - K) / stride + 1, Wo = (x.W + 2 * pad - K) / stride + 1;
    tens_t y = talloc(Cout, Ho, Wo);
    if (!w) return y;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        double* out = y.v + (size_t)co * Ho * Wo;
        if (bias)
            for (int i = 0; i < Ho * Wo; ++i) out[i] = (double)bias[co];
        for (int ci = 0; ci < x.C; ++ci) {
            const double* in = x.v + (size_t)ci * x.H * x.W;
            for (int ky = 0; ky < K; ++ky)
                for (int kx = 0; kx < K; ++kx) {
                    const double wv = (double)w[(((size_t)co * x.C + ci) * K + ky) * K + kx];
                    for (int oy = 0; oy < Ho; ++oy) {
                        const int iy = oy * stride - pad + ky;
                        if (iy < 0 || iy >= x.H) continue;
                        for (int ox = 0; ox < Wo; ++ox) {
                            const int ix = ox * stride - pad + kx;
                            if (ix < 0 || ix >= x.W) continue;
                            out[oy * Wo + ox] += wv * in[iy * x.W + ix];
                        }
                    }
                }
        }
    }
    return y;
}

/* BatchNorm2d in eval mode, in place: (x - running_mean) / sqrt(running_var + eps) * weight + bias, eps = 1e-5 */
static void batchnorm(dict_t* d, tens_t t, const char* prefix) {
    char key[256];
    snprintf(key, sizeof key, "%s.weight", prefix);
    const float* g = get(d, key, 1, t.C, 0, 0, 0);
    snprintf(key, sizeof key, "%s.bias", prefix);
    const float* b = get(d, key, 1, t.C, 0, 0, 0);
    snprintf(key, sizeof key, "%s.running_mean", prefix);
    const float* m = get(d, key, 1, t.C, 0, 0, 0);
    snprintf(key, sizeof key, "%s.running_var", prefix);
    const float* v = get(d, key, 1, t.C, 0, 0, 0);
    if (!g || !b || !m || !v) return;
    for (int c = 0; c < t.C; ++c) {
        const double inv = 1.0 / sqrt((double)v[c] + 1e-5);
        double* p = t.v + (size_t)c * t.H * t.W;
        for (int i = 0; i < t.H * t.W; ++i) p[i] = (p[i] - (double)m[c]) * inv * (double)g[c] + (double)b[c];
    }
}

static void relu(tens_t t) {
    const size_t n = (size_t)t.C * t.H * t.W;
    for (size_t i = 0; i < n; ++i)
        if (t.v[i] < 0.0) t.v[i] = 0.0;
}

static tens_t maxpool(const tens_t x, int K, int stride, int pad) {
    const int Ho = (x.H + 2 * pad - K) / stride + 1, Wo = (x.W + 2 * pad - K) / stride + 1;
    tens_t y = talloc(x.C, Ho, Wo);
    for (int c = 0; c < x.C; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                double best = -INFINITY;
                for (int ky = 0; ky < K; ++ky)
                    for (int kx = 0; kx < K; ++kx) {
                        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                        if (iy < 0 || iy >= x.H || ix < 0 || ix >= x.W) continue;      /* padding counts as -inf */
                        const double v = x.v[((size_t)c * x.H + iy) * x.W + ix];
                        if (v > best) best = v;
                    }
                y.v[((size_t)c * Ho + oy) * Wo + ox] = best;
            }
    return y;
}

/* ConvTranspose2d(kernel 2, stride 2): w (Cin, Cout, 2, 2), out(co, 2y+dy, 2x+dx) = b[co] + sum_ci x(ci, y, x) w(ci, co, dy, dx) */
static tens_t conv_transpose2(const tens_t x, const float* w, const float* bias, int Cout) {
    tens_t y = talloc(Cout, 2 * x.H, 2 * x.W);
    if (!w || !bias) return y;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co)
        for (int oy = 0; oy < 2 * x.H; ++oy)
            for (int ox = 0; ox < 2 * x.W; ++ox) {
                double acc = (double)bias[co];
                for (int ci = 0; ci < x.C; ++ci)
                    acc += x.v[((size_t)ci * x.H + oy / 2) * x.W + ox / 2] * (double)w[(((size_t)ci * Cout + co) * 2 + oy % 2) * 2 + ox % 2];
                y.v[((size_t)co * 2 * x.H + oy) * 2 * x.W + ox] = acc;
            }
    return y;
}

/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True): source = dest * (in - 1) / (out - 1) */
static tens_t upsample2(const tens_t x) {
    const int Ho = 2 * x.H, Wo = 2 * x.W;
    tens_t y = talloc(x.C, Ho, Wo);
    for (int c = 0; c < x.C; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const double sy = Ho > 1 ? (double)oy * (double)(x.H - 1) / (double)(Ho - 1) : 0.0;
                const double sx = Wo > 1 ? (double)ox * (double)(x.W - 1) / (double)(Wo - 1) : 0.0;
                int y0 = (int)floor(sy), x0 = (int)floor(sx);
                if (y0 > x.H - 1) y0 = x.H - 1;
                if (x0 > x.W - 1) x0 = x.W - 1;
                const int y1 = y0 + 1 < x.H ? y0 + 1 : y0, x1 = x0 + 1 < x.W ? x0 + 1 : x0;
                const double fy = sy - y0, fx = sx - x0;
                const double* p = x.v + (size_t)c * x.H * x.W;
                y.v[((size_t)c * Ho + oy) * Wo + ox] = (1 - fy) * ((1 - fx) * p[y0 * x.W + x0] + fx * p[y0 * x.W + x1]) +
                                                       fy * ((1 - fx) * p[y1 * x.W + x0] + fx * p[y1 * x.W + x1]);
            }
    return y;
}

/* DoubleConv: [conv3x3 (no bias) -> BN -> ReLU] x 2; keys <prefix>0.weight, <prefix>1.*, <prefix>3.weight, <prefix>4.* */
static tens_t double_conv(dict_t* d, const tens_t x, const char* prefix, int mid, int out) {
    char key[256], bn[256];
    snprintf(key, sizeof key, "%s0.weight", prefix);
    tens_t a = conv2d(x, get(d, key, 4, mid, x.C, 3, 3), NULL, mid, 3, 1, 1);
    snprintf(bn, sizeof bn, "%s1", prefix);
    batchnorm(d, a, bn);
    relu(a);
    snprintf(key, sizeof key, "%s3.weight", prefix);
    tens_t b = conv2d(a, get(d, key, 4, out, mid, 3, 3), NULL, out, 3, 1, 1);
    snprintf(bn, sizeof bn, "%s4", prefix);
    batchnorm(d, b, bn);
    relu(b);
    tfree(&a);
    return b;
}

/* Up.forward(x1 = deeper tensor, x2 = skip): up-sample x1, pad it to x2's size, concatenate [x2, x1], DoubleConv */
static tens_t up_block(dict_t* d, const tens_t x1, const tens_t x2, const char* name, int bilinear, int out) {
    char key[256], key2[256];
    tens_t u;
    int mid;
    if (bilinear) {
        u = upsample2(x1);
        mid = (x1.C + x2.C) / 2;
    } else {
        snprintf(key, sizeof key, "%s.up.weight", name);
        snprintf(key2, sizeof key2, "%s.up.bias", name);
        u = conv_transpose2(x1, get(d, key, 4, x1.C, x1.C / 2, 2, 2), get(d, key2, 1, x1.C / 2, 0, 0, 0), x1.C / 2);
        mid = out;
    }
    /* F.pad(x1, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2]) then torch.cat([x2, x1], dim=1) */
    const int dY = x2.H - u.H, dX = x2.W - u.W;
    tens_t cat = talloc(x2.C + u.C, x2.H, x2.W);
    memcpy(cat.v, x2.v, sizeof(double) * (size_t)x2.C * x2.H * x2.W);
    for (int c = 0; c < u.C; ++c)
        for (int y = 0; y < u.H; ++y)
            for (int x = 0; x < u.W; ++x) {
                const int ty = y + dY / 2, tx = x + dX / 2;
                if (ty < 0 || ty >= x2.H || tx < 0 || tx >= x2.W) continue;
                cat.v[((size_t)(x2.C + c) * x2.H + ty) * x2.W + tx] = u.v[((size_t)c * u.H + y) * u.W + x];
            }
    tfree(&u);
    snprintf(key, sizeof key, "%s.conv.double_conv.", name);
    tens_t y = double_conv(d, cat, key, mid, out);
    tfree(&cat);
    return y;
}

/* x: (N, 3, H, W) float32 in [0, 1] (H, W multiples of 16) -> logits (N, 1, H, W) as DOUBLE.  Returns 0, or 1 with a message. */
int ref_unet_forward(const ref_param_t* params, int n_params, int bilinear, const float* x, int N, int H, int W, double* logits,
                     char* err, int errcap) {
    dict_t d = {params, n_params, err, errcap, 0};
    const int f = bilinear ? 2 : 1;
    for (int n = 0; n < N && !d.failed; ++n) {
        tens_t in = talloc(3, H, W);
        for (size_t i = 0; i < (size_t)3 * H * W; ++i) in.v[i] = (double)x[(size_t)n * 3 * H * W + i];
        tens_t x1 = double_conv(&d, in, "inc.double_conv.", 64, 64);
        tens_t p1 = maxpool(x1, 2, 2, 0);
        tens_t x2 = double_conv(&d, p1, "down1.maxpool_conv.1.double_conv.", 128, 128);
        tens_t p2 = maxpool(x2, 2, 2, 0);
        tens_t x3 = double_conv(&d, p2, "down2.maxpool_conv.1.double_conv.", 256, 256);
        tens_t p3 = maxpool(x3, 2, 2, 0);
        tens_t x4 = double_conv(&d, p3, "down3.maxpool_conv.1.double_conv.", 512, 512);
        tens_t p4 = maxpool(x4, 2, 2, 0);
        tens_t x5 = double_conv(&d, p4, "down4.maxpool_conv.1.double_conv.", 1024 / f, 1024 / f);
        tens_t u1 = up_block(&d, x5, x4, "up1", bilinear, 512 / f);
        tens_t u2 = up_block(&d, u1, x3, "up2", bilinear, 256 / f);
        tens_t u3 = up_block(&d, u2, x2, "up3", bilinear, 128 / f);
        tens_t u4 = up_block(&d, u3, x1, "up4", bilinear, 64);
        tens_t o = conv2d(u4, get(&d, "outc.conv.weight", 4, 1, 64, 1, 1), get(&d, "outc.conv.bias", 1, 1, 0, 0, 0), 1, 1, 1, 0);
        if (!d.failed) memcpy(logits + (size_t)n * H * W, o.v, sizeof(double) * (size_t)H * W);
        tens_t* all[] = {&in, &x1, &p1, &x2, &p2, &x3, &p3, &x4, &p4, &x5, &u1, &u2, &u3, &u4, &o};
        for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) tfree(all[i]);
    }
    return d.failed;
}

/* timm BasicBlock: conv1 3x3/stride - bn1 - ReLU - conv2 3x3 - bn2; shortcut = x or downsample(conv1x1/stride - BN); add; ReLU */
static tens_t basic_block(dict_t* d, const tens_t x, const char* name, int planes, int stride) {
    char key[256], bn[256];
    snprintf(key, sizeof key, "%s.conv1.weight", name);
    tens_t a = conv2d(x, get(d, key, 4, planes, x.C, 3, 3), NULL, planes, 3, stride, 1);
    snprintf(bn, sizeof bn, "%s.bn1", name);
    batchnorm(d, a, bn);
    relu(a);
    snprintf(key, sizeof key, "%s.conv2.weight", name);
    tens_t b = conv2d(a, get(d, key, 4, planes, planes, 3, 3), NULL, planes, 3, 1, 1);
    snprintf(bn, sizeof bn, "%s.bn2", name);
    batchnorm(d, b, bn);
    tfree(&a);
    if (stride != 1 || x.C != planes) {
        snprintf(key, sizeof key, "%s.downsample.0.weight", name);
        tens_t s = conv2d(x, get(d, key, 4, planes, x.C, 1, 1), NULL, planes, 1, stride, 0);
        snprintf(bn, sizeof bn, "%s.downsample.1", name);
        batchnorm(d, s, bn);
        for (size_t i = 0; i < (size_t)b.C * b.H * b.W; ++i) b.v[i] += s.v[i];
        tfree(&s);
    } else {
        for (size_t i = 0; i < (size_t)b.C * b.H * b.W; ++i) b.v[i] += x.v[i];
    }
    relu(b);
    return b;
}

/* x: (N, 1, H, W) float32 in [0, 1] -> logits (N, 13) as DOUBLE (no soft-max; core.py:241-242 applies it afterwards) */
int ref_resnet18_forward(const ref_param_t* params, int n_params, const float* x, int N, int H, int W, double* logits, char* err,
                         int errcap) {
    dict_t d = {params, n_params, err, errcap, 0};
    const int planes[4] = {64, 128, 256, 512};
    for (int n = 0; n < N && !d.failed; ++n) {
        tens_t in = talloc(1, H, W);
        for (size_t i = 0; i < (size_t)H * W; ++i) in.v[i] = (double)x[(size_t)n * H * W + i];
        tens_t s = conv2d(in, get(&d, "conv1.weight", 4, 64, 1, 7, 7), NULL, 64, 7, 2, 3);
        batchnorm(&d, s, "bn1");
        relu(s);
        tens_t cur = maxpool(s, 3, 2, 1);
        tfree(&in); tfree(&s);
        for (int l = 0; l < 4; ++l)
            for (int b = 0; b < 2; ++b) {
                char name[64];
                snprintf(name, sizeof name, "layer%d.%d", l + 1, b);
                tens_t nxt = basic_block(&d, cur, name, planes[l], (b == 0 && l > 0) ? 2 : 1);
                tfree(&cur);
                cur = nxt;
            }
        const float* fw = get(&d, "fc.weight", 2, 13, 512, 0, 0);
        const float* fb = get(&d, "fc.bias", 1, 13, 0, 0, 0);
        if (fw && fb && !d.failed) {
            double pooled[512];
            for (int c = 0; c < 512; ++c) {
                double acc = 0.0;
                for (int i = 0; i < cur.H * cur.W; ++i) acc += cur.v[(size_t)c * cur.H * cur.W + i];
                pooled[c] = acc / (double)(cur.H * cur.W);
            }
            for (int o = 0; o < 13; ++o) {
                double acc = (double)fb[o];
                for (int c = 0; c < 512; ++c) acc += pooled[c] * (double)fw[(size_t)o * 512 + c];
                logits[(size_t)n * 13 + o] = acc;
            }
        }
        tfree(&cur);
    }
    return d.failed;
}
