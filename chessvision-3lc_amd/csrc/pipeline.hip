// pipeline.hip -- device versions of the classical stages either side of the CNNs.
//
//   resize_area_u8     cv2.resize(image, (256,256), INTER_AREA)                          reference core.py:212
//   extract_squares_u8 cv2.warpPerspective(image, M, (512,512)) -> cvtColor(BGR2GRAY) -> flip(.., 1) ->
//                      ChessVision.extract_squares                                        reference utils.py:131-132,
//                                                                                         core.py:298-300, 419-439
// Both are HBM-bound byte kernels; they exist so that the image stays on the device between the two CNNs.  Byte work is
// bit-exact against chessvision/classical.py (the host restatement) AND oracle/classical_ref.py (the independent one): integer
// box mean with round-half-up for integer shrink factors; the warp follows OpenCV's WarpPerspectiveInvoker operation by
// operation (block-start + in-block-column association of the double-precision homography, 1/32-pixel coordinates rounded
// half-even, int16-saturated integer pixel, integer bilinear weights summing to 2^15, round half UP), then the 15-bit fixed-point
// gray conversion of OpenCV 4.x.
//
// Vectorisation (round 4): a lane owns FOUR horizontally adjacent output pixels and writes them with one 32-bit store per
// output tensor; the warp fetches a tap pair (6 bytes of BGR BGR) with one unaligned 64-bit load per source row instead of six
// byte loads; workgroups walk 64 x 16 output tiles = one square-aligned patch whose source footprint (a ~64 x 16 pixel
// parallelogram, 3-4 KB) stays in the CU's vector cache across the tile's rows.  The 2x2 box mean of the 512 -> 256 resize reads
// two 24-byte row segments per lane and writes 12 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdlib>
#include <vector>

namespace cv {

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// ---- INTER_AREA resize -------------------------------------------------------------------------------------
// fast path of the pipeline: 3 channels, shrink factor exactly 2 in both directions, output width a multiple of 4.
// One lane = 4 output pixels: sums 2 rows x 8 source pixels (24 bytes per row, 8-byte aligned), (sum + 2) >> 2.
__global__ __launch_bounds__(256) void resize_area_2x2c3_kernel(const uint8_t* __restrict__ src, int n, int h, int w,
                                                                uint8_t* __restrict__ dst) {
    const int oh = h >> 1, ow = w >> 1, qw = ow >> 2;                 // qw = 4-pixel groups per output row
    const size_t total = (size_t)n * oh * qw;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int gx = (int)(idx % qw);
    const size_t r = idx / qw;
    const int oy = (int)(r % oh);
    const size_t img = r / oh;
    const uint8_t* s0 = src + ((img * h + (size_t)2 * oy) * w + (size_t)8 * gx) * 3;
    const uint8_t* s1 = s0 + (size_t)w * 3;
    uint32_t a[6], b[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32x2 va = *reinterpret_cast<const u32x2*>(s0 + 8 * i);
        const u32x2 vb = *reinterpret_cast<const u32x2*>(s1 + 8 * i);
        a[2 * i] = va[0]; a[2 * i + 1] = va[1]; b[2 * i] = vb[0]; b[2 * i + 1] = vb[1];
    }
    auto byte_of = [](const uint32_t* v, int k) -> uint32_t { return (v[k >> 2] >> ((k & 3) * 8)) & 0xffu; };
    uint32_t out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 12; ++k) {                                      // output byte k = pixel k/3, channel k%3
        const int p = k / 3, c = k % 3;
        const uint32_t sum = byte_of(a, 6 * p + c) + byte_of(a, 6 * p + 3 + c) + byte_of(b, 6 * p + c) + byte_of(b, 6 * p + 3 + c);
        out[k >> 2] |= ((sum + 2u) >> 2) << ((k & 3) * 8);
    }
    uint8_t* d = dst + ((img * oh + oy) * (size_t)ow + (size_t)4 * gx) * 3;
    *reinterpret_cast<u32x3*>(d) = u32x3{out[0], out[1], out[2]};        // 12 bytes, 4-byte aligned
}

// Fractional shrink = OpenCV's ResizeArea_Invoker with float32 work type, operation by operation: per contributing source row a
// horizontal pass buf = buf + S * alpha over the column table's entries in order, then sum = beta * buf for the first source row of
// a destination row and sum = sum + beta * buf for the others; saturate_cast<uchar>(sum) rounds half to even.  The tables
// (computeResizeAreaTab: weights computed in double, stored as float) come from the host (resize_area_tables below).
// One lane = one destination pixel, all channels (<= 4).
struct AreaTabs { const int* xofs; const int* xsi; const float* xa; const int* yofs; const int* ysi; const float* ya; };

__global__ __launch_bounds__(256) void resize_area_tab_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int c,
                                                              uint8_t* __restrict__ dst, int oh, int ow, AreaTabs t) {
#pragma clang fp contract(off)
    const size_t total = (size_t)n * oh * ow;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dx = (int)(idx % ow);
    const int dy = (int)((idx / ow) % oh);
    const size_t img = idx / ((size_t)ow * oh);
    const uint8_t* s = src + img * (size_t)h * w * c;
    const int k0 = t.xofs[dx], k1 = t.xofs[dx + 1], j0 = t.yofs[dy], j1 = t.yofs[dy + 1];
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = j0; j < j1; ++j) {
        const uint8_t* row = s + (size_t)t.ysi[j] * w * c;
        const float beta = t.ya[j];
        float buf[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = k0; k < k1; ++k) {
            const float alpha = t.xa[k];
            const uint8_t* px = row + (size_t)t.xsi[k] * c;
            for (int ch = 0; ch < c; ++ch) buf[ch] = buf[ch] + (float)px[ch] * alpha;
        }
        for (int ch = 0; ch < c; ++ch) sum[ch] = j == j0 ? beta * buf[ch] : sum[ch] + beta * buf[ch];
    }
    uint8_t* d = dst + idx * c;
    for (int ch = 0; ch < c; ++ch) {
        const float v = rintf(sum[ch]);
        d[ch] = (uint8_t)(v < 0.f ? 0.f : v > 255.f ? 255.f : v);
    }
}

// inv_x / inv_y: (scale, inv_scale) of each axis as the HOST computed them (inv_scale = dsize / ssize, scale = 1 / inv_scale)
__global__ void resize_area_u8_kernel(const uint8_t* __restrict__ src, int n, int h, int w, int c,
                                      uint8_t* __restrict__ dst, int oh, int ow, double2 inv_x, double2 inv_y) {
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
    // (a fractional SHRINK never reaches this kernel: resize_area_tab_kernel reproduces OpenCV's float32 table form; the branch below
    // stays for callers without tables)
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
        // Enlarging in either direction: OpenCV has no area algorithm there and runs its 8-bit bilinear resizer with the AREA
        // coefficient rule (resize.cpp): s = floor(d * scale), f = (float)((d + 1) - (s + 1) * inv_scale) reduced to [0, 1), both
        // clamped at the last source pixel, 11-bit coefficients round((1 - f) * 2048), round(f * 2048); horizontal pass in int32,
        // vertical pass (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 (VResizeLinear<uchar>).  Integers: exact.
        auto axis = [](int d, double scale, double inv_scale, int ssize, int& s0, int& s1, int& c0, int& c1) {
            int sx = (int)floor(d * scale);
            float f = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
            f = f <= 0.f ? 0.f : f - floorf(f);
            if (sx >= ssize - 1) { f = 0.f; sx = ssize - 1; }
            s0 = sx; s1 = sx + 1 < ssize ? sx + 1 : ssize - 1;
            c0 = (int)rintf((1.f - f) * 2048.f); c1 = (int)rintf(f * 2048.f);
        };
        int x0, x1, a0, a1, y0, y1, b0, b1;
        axis(ox, inv_x.x, inv_x.y, w, x0, x1, a0, a1);
        axis(oy, inv_y.x, inv_y.y, h, y0, y1, b0, b1);
        const int r0 = (int)s[((size_t)y0 * w + x0) * c + ch] * a0 + (int)s[((size_t)y0 * w + x1) * c + ch] * a1;
        const int r1 = (int)s[((size_t)y1 * w + x0) * c + ch] * a0 + (int)s[((size_t)y1 * w + x1) * c + ch] * a1;
        dst[idx] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
        return;
    }
    const double v = rint(acc);
    dst[idx] = (uint8_t)(v < 0.0 ? 0.0 : v > 255.0 ? 255.0 : v);
}

// ---- warp + gray + flip + 64-way split --------------------------------------------------------------------
// inv: per board the 3x3 map from board pixels (x, y, 1) to source pixels, row-major doubles = cv::invert of the
// getPerspectiveTransform matrix, computed on the host in OpenCV's order of operations (csrc/homography.cpp).
// Workgroup = 64 x 16 pixels of the FLIPPED board (what the classifier sees) = a quarter of one square's rows; wave w owns rows
// 4w .. 4w+3, lane l the pixels 4*(l%16) .. +3 of row l/16.  Writes the classifier's (64 squares, 64, 64) u8 layout directly,
// and optionally the flipped gray board itself.
// `one`: a single board's matrix travelling as a kernel ARGUMENT (cv_process_image: the 72-byte host-to-device copy in front of this
// launch cost ~10 us of stream time); inv == nullptr selects it.
struct WarpMatrix { double m[9]; };
__global__ __launch_bounds__(256) void extract_squares_u8_kernel(const uint8_t* __restrict__ images, int n, int h, int w,
                                                                 const double* __restrict__ inv, uint8_t* __restrict__ squares,
                                                                 uint8_t* __restrict__ boards, const WarpMatrix one) {
#pragma clang fp contract(off)
    constexpr int B = 512;
    const int img = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx0 = blockIdx.x * 64 + (lane & 15) * 4;                  // first of this lane's 4 pixels (flipped board)
    const int by = blockIdx.y * 16 + wave * 4 + (lane >> 4);
    double m0, m1, m2, m3, m4, m5, m6, m7, m8;
    if (inv) {
        const double* m = inv + (size_t)img * 9;
        m0 = m[0]; m1 = m[1]; m2 = m[2]; m3 = m[3]; m4 = m[4]; m5 = m[5]; m6 = m[6]; m7 = m[7]; m8 = m[8];
    } else {
        m0 = one.m[0]; m1 = one.m[1]; m2 = one.m[2]; m3 = one.m[3]; m4 = one.m[4]; m5 = one.m[5]; m6 = one.m[6]; m7 = one.m[7]; m8 = one.m[8];
    }
    // cv2.flip(board, 1) undone: flipped pixel bx is warp-space column xs = 511 - bx.  The lane's four columns xs = xs3 .. xs3+3
    // (xs3 = 508 - bx0, a multiple of 4) lie in ONE 64-column block of OpenCV's walk (WarpPerspectiveInvoker: BLOCK_SZ = 32, bh0 =
    // min(16, h), bw0 = min(32 * 32 / bh0, w) = 64 for the 512 x 512 board; the 128 x 32 blocks are warpAffine's), whose start column
    // enters the arithmetic:
    //   X0 = M0*blk + M1*y + M2;  W = W0 + M6*x1;  W = W ? 32/W : 0;  X = round((X0 + M0*x1) * W)   (x1 = column inside the block)
    const int xs3 = B - 4 - bx0;
    const double blk = (double)(xs3 & ~63), ys = (double)by;
    const double X0 = (m0 * blk + m1 * ys) + m2;
    const double Y0 = (m3 * blk + m4 * ys) + m5;
    const double W0 = (m6 * blk + m7 * ys) + m8;
    const uint8_t* s = images + (size_t)img * h * w * 3;
    const unsigned row_bytes = (unsigned)w * 3u;
    uint32_t packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                       // flipped pixel bx0 + j  <->  column xs3 + 3 - j
        const double x1 = (double)((xs3 & 63) + 3 - j);
        double W = W0 + m6 * x1;
        W = W != 0.0 ? 32.0 / W : 0.0;
        const double fxs = (X0 + m0 * x1) * W;
        const double fys = (Y0 + m3 * x1) * W;
        // saturate_cast<int>(clamp(f, INT_MIN, INT_MAX)): nearest-even, then the conversion itself saturates at both ends
        const int xi = (int)rint(fxs), yi = (int)rint(fys);
        const int x0 = xi >> 5, y0 = yi >> 5;                           // (the CV_16SC2 map saturates to int16: only far outside any image)
        const unsigned wx1 = (unsigned)(xi & 31), wy1 = (unsigned)(yi & 31), wx0 = 32u - wx1, wy0 = 32u - wy1;
        // Separable form of OpenCV's integer blend: with h_r = wx0 * p(r, x0) + wx1 * p(r, x0 + 1),
        //   (w00 p00 + w01 p01 + w10 p10 + w11 p11 + 2^14) >> 15  ==  (wy0 h_0 + wy1 h_1 + 512) >> 10     (w_ij = 32 wy_i wx_j: exact)
        unsigned bb, gg, rr;
        if ((unsigned)x0 < (unsigned)(w - 2) && (unsigned)y0 < (unsigned)(h - 1)) {
            // interior: both taps of a row are 6 consecutive bytes B0 G0 R0 B1 G1 R1; one unaligned 64-bit load per row (stays inside
            // the row), the horizontal blends as byte dot products (v_dot4_u32_u8) straight on the loaded words
            const unsigned off = (unsigned)(y0 * w + x0) * 3u;
            const uint64_t t = *reinterpret_cast<const u64_unaligned*>(s + off);
            const uint64_t u = *reinterpret_cast<const u64_unaligned*>(s + off + row_bytes);
            const unsigned tlo = (unsigned)t, thi = (unsigned)(t >> 32), ulo = (unsigned)u, uhi = (unsigned)(u >> 32);
            const unsigned WB = wx0 | (wx1 << 24), WG0 = wx0 << 8, WR0 = wx0 << 16, WG1 = wx1, WR1 = wx1 << 8;
            const unsigned hB0 = __builtin_amdgcn_udot4(tlo, WB, 0u, false);
            const unsigned hG0 = __builtin_amdgcn_udot4(thi, WG1, __builtin_amdgcn_udot4(tlo, WG0, 0u, false), false);
            const unsigned hR0 = __builtin_amdgcn_udot4(thi, WR1, __builtin_amdgcn_udot4(tlo, WR0, 0u, false), false);
            const unsigned hB1 = __builtin_amdgcn_udot4(ulo, WB, 0u, false);
            const unsigned hG1 = __builtin_amdgcn_udot4(uhi, WG1, __builtin_amdgcn_udot4(ulo, WG0, 0u, false), false);
            const unsigned hR1 = __builtin_amdgcn_udot4(uhi, WR1, __builtin_amdgcn_udot4(ulo, WR0, 0u, false), false);
            bb = (wy0 * hB0 + (wy1 * hB1 + 512u)) >> 10;
            gg = (wy0 * hG0 + (wy1 * hG1 + 512u)) >> 10;
            rr = (wy0 * hR0 + (wy1 * hR1 + 512u)) >> 10;
        } else {
            const int cx = x0 < -32768 ? -32768 : x0 > 32767 ? 32767 : x0;            // the map is CV_16SC2
            const int cy = y0 < -32768 ? -32768 : y0 > 32767 ? 32767 : y0;
            bb = gg = rr = 0u;
            if (cx >= -1 && cy >= -1 && cx < w && cy < h) {             // frame: BORDER_CONSTANT(0) per tap
                unsigned v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    auto tap = [&](int yy, int xx) -> unsigned {
                        return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (unsigned)s[((size_t)yy * w + xx) * 3 + c] : 0u;
                    };
                    v[c] = (wy0 * (wx0 * tap(cy, cx) + wx1 * tap(cy, cx + 1)) + (wy1 * (wx0 * tap(cy + 1, cx) + wx1 * tap(cy + 1, cx + 1)) + 512u)) >> 10;
                }
                bb = v[0]; gg = v[1]; rr = v[2];
            }
        }
        // OpenCV 4.x 8-bit BGR2GRAY: 15 fractional bits (BY15 / GY15 / RY15, gray_shift = 15); 3.x used 1868 / 9617 / 4899 >> 14
        const uint32_t gray = (bb * 3735u + gg * 19235u + (rr * 9798u + (1u << 14))) >> 15;
        packed |= gray << (8 * j);
    }
    if (boards) *reinterpret_cast<uint32_t*>(boards + ((size_t)img * B + by) * B + bx0) = packed;
    const int sq = (by >> 6) * 8 + (bx0 >> 6);            // a8..h8, a7.. order (reference core.py:436-439)
    *reinterpret_cast<uint32_t*>(squares + ((size_t)img * 64 + sq) * 4096 + (size_t)(by & 63) * 64 + (bx0 & 63)) = packed;
}

// CV_WARP=float: the float-coordinate reading of cv2.warpPerspective (chessvision/classical.py: warp_mode) -- same outputs, same
// layout; one lane per pixel of the flipped board.  Every operation is a separately rounded float32 operation (no contraction), in the
// order the host form and the oracle use: the three agree byte for byte.
__global__ __launch_bounds__(256) void extract_squares_u8_float_kernel(const uint8_t* __restrict__ images, int n, int h, int w,
                                                                       const double* __restrict__ inv, uint8_t* __restrict__ squares,
                                                                       uint8_t* __restrict__ boards, const WarpMatrix one) {
#pragma clang fp contract(off)
    constexpr int B = 512;
    const int img = blockIdx.z;
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = (float)(inv ? inv[(size_t)img * 9 + i] : one.m[i]);
    const float fx = (float)(B - 1 - bx), fy = (float)by;               // cv2.flip(board, 1) undone
    const float den = (fx * m[6] + fy * m[7]) + m[8];
    const float sx = ((fx * m[0] + fy * m[1]) + m[2]) / den;
    const float sy = ((fx * m[3] + fy * m[4]) + m[5]) / den;
    uint32_t gray = 0;
    if (isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f) {
        const float flx = floorf(sx), fly = floorf(sy);
        const int ix = (int)flx, iy = (int)fly;
        const float al = sx - flx, be = sy - fly;
        const uint8_t* s = images + (size_t)img * h * w * 3;
        unsigned v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            auto tap = [&](int yy, int xx) -> float {
                return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? (float)s[((size_t)yy * w + xx) * 3 + c] : 0.f;
            };
            const float p00 = tap(iy, ix), p01 = tap(iy, ix + 1), p10 = tap(iy + 1, ix), p11 = tap(iy + 1, ix + 1);
            const float top = p00 + al * (p01 - p00);
            const float bot = p10 + al * (p11 - p10);
            const float r = rintf(top + be * (bot - top));
            v[c] = (unsigned)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
        }
        gray = (v[0] * 3735u + v[1] * 19235u + (v[2] * 9798u + (1u << 14))) >> 15;
    } else {
        gray = (1u << 14) >> 15;                                          // border value 0 in every channel
    }
    if (boards) boards[((size_t)img * B + by) * B + bx] = (uint8_t)gray;
    const int sq = (by >> 6) * 8 + (bx >> 6);
    squares[((size_t)img * 64 + sq) * 4096 + (size_t)(by & 63) * 64 + (bx & 63)] = (uint8_t)gray;
}

static bool warp_float_mode() {
    static const bool on = [] { const char* v = std::getenv("CV_WARP"); return v && (v[0] == 'f' || v[0] == 'F') && (v[1] == 'l' || v[1] == 'L'); }();
    return on;
}

// host side of the fractional shrink: OpenCV's computeResizeAreaTab for one axis (source indices in PIXELS, float weights, CSR offsets)
void resize_area_table(int ssize, int dsize, std::vector<int>& ofs, std::vector<int>& si, std::vector<float>& alpha) {
    const double scale = (double)ssize / dsize;
    ofs.assign(1, 0); si.clear(); alpha.clear();
    for (int d = 0; d < dsize; ++d) {
        const double fs1 = d * scale, fs2 = fs1 + scale;
        const double cell = scale < ssize - fs1 ? scale : ssize - fs1;
        int s1 = (int)std::ceil(fs1), s2 = (int)std::floor(fs2);
        s2 = s2 < ssize - 1 ? s2 : ssize - 1;
        s1 = s1 < s2 ? s1 : s2;
        if (s1 - fs1 > 1e-3) { si.push_back(s1 - 1); alpha.push_back((float)((s1 - fs1) / cell)); }
        for (int sx = s1; sx < s2; ++sx) { si.push_back(sx); alpha.push_back((float)(1.0 / cell)); }
        if (fs2 - s2 > 1e-3) {
            double a = fs2 - s2 < 1.0 ? fs2 - s2 : 1.0;
            a = a < cell ? a : cell;
            si.push_back(s2); alpha.push_back((float)(a / cell));
        }
        ofs.push_back((int)si.size());
    }
}

// tabs: device pointers of the two tables (built by the caller with resize_area_table and uploaded once per geometry)
hipError_t resize_area_u8_tab(const uint8_t* src, int n, int h, int w, int c, uint8_t* dst, int oh, int ow, const int* xofs,
                              const int* xsi, const float* xa, const int* yofs, const int* ysi, const float* ya, hipStream_t s) {
    if (c < 1 || c > 4) return hipErrorInvalidValue;
    const size_t total = (size_t)n * oh * ow;
    const AreaTabs t{xofs, xsi, xa, yofs, ysi, ya};
    hipLaunchKernelGGL(resize_area_tab_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, c, dst, oh, ow, t);
    return hipGetLastError();
}

hipError_t resize_area_u8(const uint8_t* src, int n, int h, int w, int c, uint8_t* dst, int oh, int ow, hipStream_t s) {
    if (c == 3 && h == 2 * oh && w == 2 * ow && ow % 4 == 0 && ((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 3) == 0) {
        const size_t total = (size_t)n * oh * (ow / 4);
        hipLaunchKernelGGL(resize_area_2x2c3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, dst);
        return hipGetLastError();
    }
    const size_t total = (size_t)n * oh * ow * c;
    const double isx = (double)ow / (double)w, isy = (double)oh / (double)h;
    hipLaunchKernelGGL(resize_area_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, c, dst, oh, ow,
                       double2{1.0 / isx, isx}, double2{1.0 / isy, isy});
    return hipGetLastError();
}
hipError_t extract_squares_u8(const uint8_t* images, int n, int h, int w, const double* inv, uint8_t* squares,
                              uint8_t* boards, hipStream_t s) {
    WarpMatrix none;
    for (int i = 0; i < 9; ++i) none.m[i] = 0.0;
    // grid.z = boards: jobs of the pipeline carry <= a few hundred boards (65535 is the limit)
    for (int off = 0; off < n; off += 32768) {
        const int cnt = n - off < 32768 ? n - off : 32768;
        if (warp_float_mode()) {
            hipLaunchKernelGGL(extract_squares_u8_float_kernel, dim3(8, 128, (unsigned)cnt), dim3(256), 0, s,
                               images + (size_t)off * h * w * 3, cnt, h, w, inv + (size_t)off * 9, squares + (size_t)off * 64 * 4096,
                               boards ? boards + (size_t)off * 512 * 512 : nullptr, none);
            continue;
        }
        hipLaunchKernelGGL(extract_squares_u8_kernel, dim3(8, 32, (unsigned)cnt), dim3(256), 0, s,
                           images + (size_t)off * h * w * 3, cnt, h, w, inv + (size_t)off * 9, squares + (size_t)off * 64 * 4096,
                           boards ? boards + (size_t)off * 512 * 512 : nullptr, none);
    }
    return hipGetLastError();
}

// ONE board whose matrix (host memory, 9 doubles) rides in the kernel arguments: no device copy of it exists
hipError_t extract_squares_u8_one(const uint8_t* image, int h, int w, const double* inv_host, uint8_t* squares, uint8_t* board, hipStream_t s) {
    WarpMatrix one;
    for (int i = 0; i < 9; ++i) one.m[i] = inv_host[i];
    if (warp_float_mode())
        hipLaunchKernelGGL(extract_squares_u8_float_kernel, dim3(8, 128, 1), dim3(256), 0, s, image, 1, h, w, (const double*)nullptr, squares, board, one);
    else
        hipLaunchKernelGGL(extract_squares_u8_kernel, dim3(8, 32, 1), dim3(256), 0, s, image, 1, h, w, (const double*)nullptr, squares, board, one);
    return hipGetLastError();
}

}  // namespace cv
