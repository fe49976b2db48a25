/* ops_ref.c -- plain-C restatement of every primitive op on the ChessVision CNN hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Independent of torch: used by tests/test_oracle_ops.py to
 * cross-check the torch-CPU arithmetic the oracle modules are composed of, so the oracle is pinned by two
 * implementations that share no code.  All tensors are contiguous NCHW float32; accumulation is double.
 *
 * Ops (reference call sites):
 *   conv2d            UNet DoubleConv / OutConv, ResNet conv1 / blocks / downsample   (torch.nn.Conv2d)
 *   conv_transpose2d  UNet Up.up, kernel 2 stride 2                                   (torch.nn.ConvTranspose2d)
 *   batchnorm_eval    y = (x - mean) / sqrt(var + eps) * gamma + beta                 (torch.nn.BatchNorm2d.eval())
 *   relu, maxpool2d, upsample_bilinear2x (align_corners=True), global_avgpool, linear,
 *   softmax (chessvision/core.py:242), sigmoid (core.py:273)
 */
#include <math.h>
#include <stddef.h>

#define IDX4(n, c, y, x, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (y)) * (W) + (x))

void ref_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout,
                int K, int stride, int pad) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < K; ++ky) {
                            const int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                const int ix = ox * stride - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[IDX4(n, ci, iy, ix, Cin, H, W)] *
                                       (double)w[(((size_t)co * Cin + ci) * K + ky) * K + kx];
                            }
                        }
                    y[IDX4(n, co, oy, ox, Cout, Ho, Wo)] = (float)acc;
                }
}

/* weight layout (Cin, Cout, 2, 2); out(n, co, 2y+dy, 2x+dx) = b[co] + sum_ci in(n,ci,y,x) * w(ci,co,dy,dx) */
void ref_conv_transpose2x2(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                           int Cout) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    const int iy = oy / 2, ix = ox / 2, dy = oy % 2, dx = ox % 2;
                    double acc = bias ? bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        acc += (double)x[IDX4(n, ci, iy, ix, Cin, H, W)] *
                               (double)w[(((size_t)ci * Cout + co) * 2 + dy) * 2 + dx];
                    y[IDX4(n, co, oy, ox, Cout, Ho, Wo)] = (float)acc;
                }
}

void ref_batchnorm_eval(const float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                        float eps, float* y, int N, int C, int HW) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const double s = (double)gamma[c] / sqrt((double)var[c] + (double)eps);
            for (int i = 0; i < HW; ++i) {
                const size_t k = ((size_t)n * C + c) * HW + i;
                y[k] = (float)(((double)x[k] - (double)mean[c]) * s + (double)beta[c]);
            }
        }
}

void ref_relu(const float* x, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f;
}

void ref_maxpool2d(const float* x, float* y, int N, int C, int H, int W, int K, int stride, int pad) {
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    float m = -INFINITY;
                    for (int ky = 0; ky < K; ++ky) {
                        const int iy = oy * stride - pad + ky;
                        if (iy < 0 || iy >= H) continue;
                        for (int kx = 0; kx < K; ++kx) {
                            const int ix = ox * stride - pad + kx;
                            if (ix < 0 || ix >= W) continue;
                            const float v = x[IDX4(n, c, iy, ix, C, H, W)];
                            if (v > m) m = v;
                        }
                    }
                    y[IDX4(n, c, oy, ox, C, Ho, Wo)] = m;
                }
}

/* scale 2, bilinear, align_corners=True: src = dst * (in - 1) / (out - 1) */
void ref_upsample_bilinear2x(const float* x, float* y, int N, int C, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    const double fy = Ho > 1 ? (double)oy * (H - 1) / (Ho - 1) : 0.0;
                    const double fx = Wo > 1 ? (double)ox * (W - 1) / (Wo - 1) : 0.0;
                    int y0 = (int)fy, x0 = (int)fx;
                    if (y0 > H - 1) y0 = H - 1;
                    if (x0 > W - 1) x0 = W - 1;
                    const int y1 = y0 < H - 1 ? y0 + 1 : y0, x1 = x0 < W - 1 ? x0 + 1 : x0;
                    const double ly = fy - y0, lx = fx - x0;
                    const double v = (1 - ly) * ((1 - lx) * x[IDX4(n, c, y0, x0, C, H, W)] + lx * x[IDX4(n, c, y0, x1, C, H, W)]) +
                                     ly * ((1 - lx) * x[IDX4(n, c, y1, x0, C, H, W)] + lx * x[IDX4(n, c, y1, x1, C, H, W)]);
                    y[IDX4(n, c, oy, ox, C, Ho, Wo)] = (float)v;
                }
}

void ref_global_avgpool(const float* x, float* y, int N, int C, int HW) {
    for (int i = 0; i < N * C; ++i) {
        double s = 0.0;
        for (int k = 0; k < HW; ++k) s += x[(size_t)i * HW + k];
        y[i] = (float)(s / HW);
    }
}

void ref_linear(const float* x, const float* w, const float* b, float* y, int N, int In, int Out) {
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < Out; ++o) {
            double s = b ? b[o] : 0.0;
            for (int i = 0; i < In; ++i) s += (double)x[(size_t)n * In + i] * (double)w[(size_t)o * In + i];
            y[(size_t)n * Out + o] = (float)s;
        }
}

void ref_softmax_rows(const float* x, float* y, int N, int C) {
    for (int n = 0; n < N; ++n) {
        double m = -INFINITY, s = 0.0;
        for (int c = 0; c < C; ++c) if (x[(size_t)n * C + c] > m) m = x[(size_t)n * C + c];
        for (int c = 0; c < C; ++c) s += exp((double)x[(size_t)n * C + c] - m);
        for (int c = 0; c < C; ++c) y[(size_t)n * C + c] = (float)(exp((double)x[(size_t)n * C + c] - m) / s);
    }
}

void ref_sigmoid(const float* x, float* y, size_t n) {
    for (size_t i = 0; i < n; ++i) y[i] = (float)(1.0 / (1.0 + exp(-(double)x[i])));
}
