// homography.cpp -- quadrangle -> the 3x3 map the device warp consumes, in OpenCV's order of operations (host side).
//
// Reference: utils.extract_perspective (chessvision/utils.py:115-132) calls cv2.getPerspectiveTransform(approx, dest) and
// cv2.warpPerspective(image, coeffs, out_size); the latter inverts the matrix before walking the destination.  The last bit of
// that inverse decides which way a source coordinate that is an exact .5 tie in 1/32 pixels rounds, so both steps follow the
// arithmetic OpenCV 4.x publishes, operation by operation (IEEE double, no fused multiply-add):
//   getPerspectiveTransform (imgproc/src/imgwarp.cpp): rows i / i+4 of an 8x8 system, the -x*u products formed in FLOAT
//       (Point2f operands), solve(A, B, X, DECOMP_LU) = LUImpl of core/src/matrix_decomp.cpp for this size: partial pivoting
//       on the first largest magnitude, alpha = A[j][i] * (-1 / A[i][i]), back substitution s -= A[i][k] * x[k]; x[i] = s / A[i][i]
//   invert, 3x3 double (core/src/lapack.cpp): cofactors times the reciprocal of det3 (expanded along the first row)
// oracle/classical_ref.py holds an independent scalar-Python restatement of the same two procedures; tests require the two to
// agree bit for bit.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "engine.h"

namespace cv {

#pragma clang fp contract(off)

// m: 9 doubles (row-major, m[8] = 1), all zero when the four points are degenerate (OpenCV leaves the solution untouched)
void perspective_transform_cv(const float src[8], const float dst[8], double m[9]) {
#pragma clang fp contract(off)
    double a[8][8], b[8];
    for (int i = 0; i < 4; ++i) {
        const float x = src[2 * i], y = src[2 * i + 1], u = dst[2 * i], v = dst[2 * i + 1];
        a[i][0] = a[i + 4][3] = x;
        a[i][1] = a[i + 4][4] = y;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        volatile float xu = -x * u, yu = -y * u, xv = -x * v, yv = -y * v;   // float products, as OpenCV's Point2f arithmetic
        a[i][6] = xu; a[i][7] = yu; a[i + 4][6] = xv; a[i + 4][7] = yv;
        b[i] = u; b[i + 4] = v;
    }
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < 8; ++i) {
        int k = i;
        for (int j = i + 1; j < 8; ++j)
            if (std::fabs(a[j][i]) > std::fabs(a[k][i])) k = j;
        if (std::fabs(a[k][i]) < eps) { std::memset(m, 0, 9 * sizeof(double)); return; }
        if (k != i) {
            for (int j = i; j < 8; ++j) { const double t = a[i][j]; a[i][j] = a[k][j]; a[k][j] = t; }
            const double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1 / a[i][i];
        for (int j = i + 1; j < 8; ++j) {
            const double alpha = a[j][i] * d;
            for (int c = i + 1; c < 8; ++c) a[j][c] += alpha * a[i][c];
            b[j] += alpha * b[i];
        }
    }
    for (int i = 7; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < 8; ++c) s -= a[i][c] * b[c];
        b[i] = s / a[i][i];
    }
    for (int i = 0; i < 8; ++i) m[i] = b[i];
    m[8] = 1.0;
}

// inv = m^-1 as cv::invert computes it for a 3x3 double matrix; zeros when det == 0
void invert3_cv(const double m[9], double inv[9]) {
#pragma clang fp contract(off)
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0.0) { std::memset(inv, 0, 9 * sizeof(double)); return; }
    const double r = 1.0 / det;
    inv[0] = (e * i - f * h) * r; inv[1] = (c * h - b * i) * r; inv[2] = (b * f - c * e) * r;
    inv[3] = (f * g - d * i) * r; inv[4] = (a * i - c * g) * r; inv[5] = (c * d - a * f) * r;
    inv[6] = (d * h - e * g) * r; inv[7] = (b * g - a * h) * r; inv[8] = (a * e - b * d) * r;
}

// quads: n x 4 x (x, y) float32 in source pixels, vertex order TR, TL, BL, BR (what _rotate_quadrangle leaves) -> per board
// the forward matrix (nullable) and its inverse (board pixel -> source pixel), for the destination corners (0,0), (w,0), (w,h), (0,h)
void board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse) {
    const float dst[8] = {0.f, 0.f, (float)out_w, 0.f, (float)out_w, (float)out_h, 0.f, (float)out_h};
    for (int k = 0; k < n; ++k) {
        double m[9];
        perspective_transform_cv(quads + (size_t)k * 8, dst, m);
        if (forward) std::memcpy(forward + (size_t)k * 9, m, sizeof(m));
        if (inverse) invert3_cv(m, inverse + (size_t)k * 9);
    }
}

}  // namespace cv
