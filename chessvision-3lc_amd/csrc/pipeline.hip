// pipeline.hip -- device versions of the classical stages either side of the CNNs.
//
//   resize_area_u8     cv2.resize(image, (256,256), INTER_AREA)                          reference core.py:212
//   extract_squares_u8 cv2.warpPerspective(image, M, (512,512)) -> cvtColor(BGR2GRAY) -> flip(.., 1) ->
//                      ChessVision.extract_squares                                        reference utils.py:131-132,
//                                                                                         core.py:298-300, 419-439
// Both are HBM-bound byte kernels (one lane per output pixel, coalesced u8 stores); they exist so the batched pipeline
// keeps the image on the device between the two CNNs instead of making two host round trips per board.  Arithmetic
// mirrors chessvision/classical.py (the host restatement the tests use as checker): integer box mean for integer shrink
// factors; double-precision homography + 1/32-pixel snapped bilinear taps + round-half-even; the 14-bit fixed-point
// gray conversion of OpenCV.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cv {

// ---- INTER_AREA resize -------------------------------------------------------------------------------------
__global__ void resize_area_u8_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int c,
                                      uint8_t* __restrict__ dst, int oh, int ow) {
#pragma clang fp contract(off)                          // keep mul/add unfused: matches the numpy checker bit for bit
    const size_t total = (size_t)n * oh * ow * c;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t r = idx;
    const int ch = (int)(r % c); r /= c;
    const int ox = (int)(r % ow); r /= ow;
    const int oy = (int)(r % oh);
    const int img = (int)(r / oh);
    const uint8_t* s = src + (size_t)img * h * w * c;
    if (h % oh == 0 && w % ow == 0) {                      // exact box mean, round half up: (sum + n/2) / n
        const int fy = h / oh, fx = w / ow;
        unsigned acc = 0;
        for (int y = 0; y < fy; ++y)
            for (int x = 0; x < fx; ++x) acc += s[((size_t)(oy * fy + y) * w + ox * fx + x) * c + ch];
        dst[idx] = (uint8_t)((acc + (fy * fx) / 2) / (fy * fx));
        return;
    }
    // fractional shrink: coverage-weighted mean (rows then columns accumulate in double); enlarging: bilinear
    const double sy = (double)h / oh, sx = (double)w / ow;
    double acc = 0.0;
    if (oh <= h && ow <= w) {
        const double y_lo = oy * sy, y_hi = (oy + 1) * sy, x_lo = ox * sx, x_hi = (ox + 1) * sx;
        const int y0 = (int)floor(y_lo), y1 = min(h, (int)ceil(y_hi));
        const int x0 = (int)floor(x_lo), x1 = min(w, (int)ceil(x_hi));
        for (int y = y0; y < y1; ++y) {
            const double wy = fmax(0.0, fmin(y_hi, (double)(y + 1)) - fmax(y_lo, (double)y)) / sy;
            double row = 0.0;
            for (int x = x0; x < x1; ++x) {
                const double wx = fmax(0.0, fmin(x_hi, (double)(x + 1)) - fmax(x_lo, (double)x)) / sx;
                row += wx * (double)s[((size_t)y * w + x) * c + ch];
            }
            acc += wy * row;
        }
    } else {
        const double fy = (oy + 0.5) * sy - 0.5, fx = (ox + 0.5) * sx - 0.5;
        const int iy = (int)floor(fy), ix = (int)floor(fx);
        const double ty = fy - iy, tx = fx - ix;
        const int ya = min(max(iy, 0), h - 1), yb = min(max(iy + 1, 0), h - 1);
        const int xa = min(max(ix, 0), w - 1), xb = min(max(ix + 1, 0), w - 1);
        acc = (1 - ty) * ((1 - tx) * s[((size_t)ya * w + xa) * c + ch] + tx * s[((size_t)ya * w + xb) * c + ch]) +
              ty * ((1 - tx) * s[((size_t)yb * w + xa) * c + ch] + tx * s[((size_t)yb * w + xb) * c + ch]);
    }
    const double v = rint(acc);
    dst[idx] = (uint8_t)(v < 0.0 ? 0.0 : v > 255.0 ? 255.0 : v);
}

// ---- warp + gray + flip + 64-way split --------------------------------------------------------------------
// inv: per board the 3x3 map from board pixels (x, y, 1) to source pixels, row-major doubles (host side inverts the
// reference's getPerspectiveTransform matrix).  One lane per pixel of the 512x512 board; writes the classifier's
// (64 squares, 64, 64) u8 layout directly, and optionally the flipped gray board itself.
__global__ void extract_squares_u8_kernel(const uint8_t* __restrict__ images, int n, int h, int w,
                                          const double* __restrict__ inv, uint8_t* __restrict__ squares,
                                          uint8_t* __restrict__ boards) {
#pragma clang fp contract(off)
    constexpr int B = 512;
    const size_t total = (size_t)n * B * B;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int bx = (int)(idx % B);                       // pixel of the FLIPPED board (what the classifier sees)
    const int by = (int)((idx / B) % B);
    const int img = (int)(idx / ((size_t)B * B));
    const double* m = inv + (size_t)img * 9;
    const double xs = (double)(B - 1 - bx), ys = (double)by;       // undo cv2.flip(board, 1)
    // cv2.warpPerspective, INTER_LINEAR, BORDER_CONSTANT(0), in OpenCV's fixed-point form (imgwarp.cpp): source coordinates in 1/32
    // pixel (INTER_BITS = 5; X = round(X0 * (32 / W0)), ties to even), integer bilinear weights (32-a)(32-b)*32 ... a*b*32 that sum to
    // 2^15 (INTER_REMAP_COEF_BITS), pixel = (sum + 2^14) >> 15: round half UP.  (Rounds 1-2 blended in double and rounded ties to even:
    // one grey level apart on ~0.5 % of the pixels.)
    const double den = m[6] * xs + m[7] * ys + m[8];
    const double scale = den != 0.0 ? 32.0 / den : 0.0;
    double fxs = (m[0] * xs + m[1] * ys + m[2]) * scale;
    double fys = (m[3] * xs + m[4] * ys + m[5]) * scale;
    fxs = fxs < -2147483648.0 ? -2147483648.0 : fxs > 2147483647.0 ? 2147483647.0 : fxs;
    fys = fys < -2147483648.0 ? -2147483648.0 : fys > 2147483647.0 ? 2147483647.0 : fys;
    const long long xi = (long long)rint(fxs), yi = (long long)rint(fys);
    const long long x0 = xi >> 5, y0 = yi >> 5;
    const int ax = (int)(xi & 31), ay = (int)(yi & 31);
    const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
    const bool inside = x0 >= -1 && x0 < w && y0 >= -1 && y0 < h;
    const uint8_t* s = images + (size_t)img * h * w * 3;
    int bgr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 0;
        if (inside) {
            auto tap = [&](long long yy, long long xx) -> int {
                return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (int)s[((size_t)yy * w + xx) * 3 + c] : 0;
            };
            acc = w00 * tap(y0, x0) + w01 * tap(y0, x0 + 1) + w10 * tap(y0 + 1, x0) + w11 * tap(y0 + 1, x0 + 1);
        }
        bgr[c] = (acc + (1 << 14)) >> 15;
    }
    // OpenCV 4.x 8-bit BGR2GRAY: 15 fractional bits (BY15 / GY15 / RY15, gray_shift = 15); 3.x used 1868 / 9617 / 4899 >> 14
    const uint8_t gray = (uint8_t)((bgr[0] * 3735 + bgr[1] * 19235 + bgr[2] * 9798 + (1 << 14)) >> 15);
    if (boards) boards[idx] = gray;
    const int sq = (by >> 6) * 8 + (bx >> 6);            // a8..h8, a7.. order (reference core.py:436-439)
    squares[((size_t)img * 64 + sq) * 4096 + (size_t)(by & 63) * 64 + (bx & 63)] = gray;
}

hipError_t resize_area_u8(const uint8_t* src, int n, int h, int w, int c, uint8_t* dst, int oh, int ow, hipStream_t s) {
    const size_t total = (size_t)n * oh * ow * c;
    hipLaunchKernelGGL(resize_area_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, c, dst, oh, ow);
    return hipGetLastError();
}
hipError_t extract_squares_u8(const uint8_t* images, int n, int h, int w, const double* inv, uint8_t* squares,
                              uint8_t* boards, hipStream_t s) {
    const size_t total = (size_t)n * 512 * 512;
    hipLaunchKernelGGL(extract_squares_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, images, n, h, w, inv,
                       squares, boards);
    return hipGetLastError();
}

}  // namespace cv
