// contour.cpp -- binary mask -> board quadrangle on the host, in C++.
//
// Replaces, for the batched pipeline, the chain the reference runs through OpenCV per image
// (chessvision/core.py:357-411): findContours(RETR_CCOMP) -> [area / bounding-box filter when more than one contour]
// -> approxPolyDP(0.1 * perimeter, closed) -> first 4-vertex result -> _rotate_quadrangle.
// Same algorithm as chessvision/classical.py (Suzuki-Abe border following over 8-connected components and their
// holes, Douglas-Peucker for closed curves with OpenCV's start-point strategy); tests/test_contour_cpp.py checks the
// two implementations agree exactly on the reference's mask fixtures and on random polygons.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

namespace cv {

namespace {

struct Pt { int x, y; };
typedef std::vector<Pt> Contour;

// 8-neighbourhood, clockwise from west, (dy, dx) with y down
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
const int kDx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

inline int nbr_index(int dy, int dx) {
    for (int i = 0; i < 8; ++i)
        if (kDy[i] == dy && kDx[i] == dx) return i;
    return 0;
}

struct Image {
    const uint8_t* p; int h, w;
    bool at(int y, int x) const { return y >= 0 && y < h && x >= 0 && x < w && p[(size_t)y * w + x] != 0; }
};

// Suzuki-Abe Algorithm 1, step 3: follow one border from `start`, `prev` = the background pixel scanned before it
Contour trace_border(const Image& f, int si, int sj, int pi, int pj) {
    Contour pts;
    const int k0 = nbr_index(pi - si, pj - sj);
    int fi = -1, fj = -1;
    for (int s = 0; s < 8; ++s) {
        const int k = (k0 + s) & 7;
        if (f.at(si + kDy[k], sj + kDx[k])) { fi = si + kDy[k]; fj = sj + kDx[k]; break; }
    }
    if (fi < 0) { pts.push_back({sj, si}); return pts; }
    int i2 = fi, j2 = fj, i3 = si, j3 = sj;
    for (;;) {
        const int k = nbr_index(i2 - i3, j2 - j3);
        int i4 = i3, j4 = j3;
        for (int s = 1; s <= 8; ++s) {
            const int kk = ((k - s) % 8 + 8) & 7;
            if (f.at(i3 + kDy[kk], j3 + kDx[kk])) { i4 = i3 + kDy[kk]; j4 = j3 + kDx[kk]; break; }
        }
        pts.push_back({j3, i3});
        if (i4 == si && j4 == sj && i3 == fi && j3 == fj) break;
        i2 = i3; j2 = j3; i3 = i4; j3 = j4;
    }
    return pts;
}

struct Comp { int y, x; int bw, bh; };             // first raster pixel and bounding-box extent of a component

// connected components by flood fill; returns the first raster pixel of every component (label order = raster order)
std::vector<Comp> component_starts(const std::vector<uint8_t>& on, int h, int w, bool eight,
                                   bool skip_frame_touching) {
    std::vector<int> lab((size_t)h * w, 0);
    std::vector<Comp> starts;
    std::vector<int> stack;
    int next = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            if (!on[(size_t)y * w + x] || lab[(size_t)y * w + x]) continue;
            ++next;
            bool touches = false;
            int x0 = x, x1 = x, y0 = y, y1 = y;
            stack.clear();
            stack.push_back(y * w + x);
            lab[(size_t)y * w + x] = next;
            while (!stack.empty()) {
                const int p = stack.back(); stack.pop_back();
                const int cy = p / w, cx = p % w;
                if (cy == 0 || cx == 0 || cy == h - 1 || cx == w - 1) touches = true;
                x0 = std::min(x0, cx); x1 = std::max(x1, cx); y0 = std::min(y0, cy); y1 = std::max(y1, cy);
                for (int k = 0; k < 8; ++k) {
                    if (!eight && (k & 1)) continue;                    // odd entries are the diagonals
                    const int ny = cy + kDy[k], nx = cx + kDx[k];
                    if (ny < 0 || nx < 0 || ny >= h || nx >= w) continue;
                    const size_t q = (size_t)ny * w + nx;
                    if (on[q] && !lab[q]) { lab[q] = next; stack.push_back((int)q); }
                }
            }
            if (!(skip_frame_touching && touches)) starts.push_back({y, x, x1 - x0 + 1, y1 - y0 + 1});
        }
    return starts;
}

// All borders (outer, then holes).  `*total` receives the number of contours the image has; when it exceeds one the
// reference filters by area share >= 0.35, and since a border polygon's area is below its bounding-box area, borders
// whose box is smaller than that can never pass: they are counted but not traced (noisy masks have thousands).
std::vector<Contour> find_contours(const uint8_t* mask, int h, int w, size_t* total) {
    const Image f{mask, h, w};
    std::vector<uint8_t> fg((size_t)h * w), bg((size_t)h * w);
    for (size_t i = 0; i < fg.size(); ++i) { fg[i] = mask[i] != 0; bg[i] = !fg[i]; }
    const std::vector<Comp> outer = component_starts(fg, h, w, true, false);
    const std::vector<Comp> holes = component_starts(bg, h, w, false, true);
    *total = outer.size() + holes.size();
    const bool prune = *total > 1;
    const double need = 0.35 * (double)h * w;
    std::vector<Contour> out;
    for (auto& s : outer)
        if (!prune || (double)s.bw * s.bh >= need) out.push_back(trace_border(f, s.y, s.x, s.y, s.x - 1));
    for (auto& s : holes)       // a hole border runs on the foreground pixels around the hole: box is 2 wider/taller
        if (!prune || (double)(s.bw + 2) * (s.bh + 2) >= need) out.push_back(trace_border(f, s.y, s.x - 1, s.y, s.x));
    return out;
}

double contour_area(const Contour& c) {
    if (c.size() < 3) return 0.0;
    double a = 0.0, b = 0.0;
    for (size_t i = 0; i < c.size(); ++i) {
        const Pt& p = c[i]; const Pt& n = c[(i + 1) % c.size()];
        a += (double)p.x * n.y; b += (double)p.y * n.x;
    }
    return std::fabs(a - b) * 0.5;
}

double arc_length(const Contour& c) {
    double s = 0.0;
    for (size_t i = 0; i < c.size(); ++i) {
        const Pt& p = c[i]; const Pt& n = c[(i + 1) % c.size()];
        s += std::sqrt((double)(n.x - p.x) * (n.x - p.x) + (double)(n.y - p.y) * (n.y - p.y));
    }
    return s;
}

Contour approx_poly_dp(const Contour& src, double epsilon) {
    const int count = (int)src.size();
    Contour dst;
    if (count == 0) return dst;
    const double eps2 = epsilon * epsilon;
    int pos = 0, right_start = 0;
    bool le_eps = false;
    for (int it = 0; it < 3; ++it) {
        pos = (pos + right_start) % count;
        long long best = -1; int bj = 0;
        for (int j = 1; j < count; ++j) {
            const Pt& p = src[(pos + j) % count];
            const long long d = (long long)(p.x - src[pos].x) * (p.x - src[pos].x) + (long long)(p.y - src[pos].y) * (p.y - src[pos].y);
            if (d > best) { best = d; bj = j; }
        }
        if (count == 1) { best = 0; bj = 0; }
        right_start = bj;
        le_eps = (double)best <= eps2;
    }
    if (le_eps) { dst.push_back(src[pos]); return dst; }
    const int a = pos % count, b = (right_start + pos) % count;
    std::vector<std::pair<int, int>> stack;
    stack.push_back({b, a});
    stack.push_back({a, b});
    while (!stack.empty()) {
        const int s = stack.back().first, e = stack.back().second;
        stack.pop_back();
        const Pt start = src[s], end = src[e];
        bool le = true; int split = s;
        if ((s + 1) % count != e) {
            const double dx = end.x - start.x, dy = end.y - start.y;
            double md = -1.0;
            const int last = e > s ? e : e + count;
            for (int i = s + 1; i < last; ++i) {
                const Pt& p = src[i % count];
                const double d = std::fabs((double)(p.y - start.y) * dx - (double)(p.x - start.x) * dy);
                if (d > md) { md = d; split = i % count; }
            }
            le = md * md <= eps2 * (dx * dx + dy * dy);
        }
        if (le) dst.push_back(start);
        else { stack.push_back({split, e}); stack.push_back({s, split}); }
    }
    size_t i = 0;
    while (dst.size() > 2 && i < dst.size()) {
        const size_t n = dst.size();
        const Pt start = dst[(i + n - 1) % n], cur = dst[i], end = dst[(i + 1) % n];
        const double dx = end.x - start.x, dy = end.y - start.y;
        const double dist = std::fabs((double)(cur.x - start.x) * dy - (double)(cur.y - start.y) * dx);
        const double inner = (double)(cur.x - start.x) * (end.x - cur.x) + (double)(cur.y - start.y) * (end.y - cur.y);
        if (dist * dist <= 0.5 * eps2 * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) dst.erase(dst.begin() + i);
        else ++i;
    }
    return dst;
}

}  // namespace

// quad: 4 x (x, y) in mask pixels after the reference's rotation rule; returns true when a quadrangle was found
bool find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8]) {
    size_t total = 0;
    std::vector<Contour> contours = find_contours(mask, h, w, &total);
    if (total > 1) {                                   // reference core.py:362-366, 381-404
        std::vector<Contour> kept;
        const double area = (double)h * w;
        for (auto& c : contours) {
            const double share = contour_area(c) / area;
            if (share < 0.35 || share > 1.0) continue;
            int x0 = c[0].x, x1 = c[0].x, y0 = c[0].y, y1 = c[0].y;
            for (auto& p : c) { x0 = std::min(x0, p.x); x1 = std::max(x1, p.x); y0 = std::min(y0, p.y); y1 = std::max(y1, p.y); }
            const double bw = x1 - x0 + 1, bh = y1 - y0 + 1;
            const double r = (bw == 0 || bh == 0) ? -1.0 : std::min(bw, bh) / std::max(bw, bh);
            if (r >= 0.6) kept.push_back(c);
        }
        contours.swap(kept);
    }
    for (auto& c : contours) {
        Contour q = approx_poly_dp(c, 0.1 * arc_length(c));
        if (q.size() != 4) continue;
        int order[4] = {0, 1, 2, 3};
        if (q[0].x < q[2].x) { order[0] = 3; order[1] = 0; order[2] = 1; order[3] = 2; }   // core.py:406-411
        for (int i = 0; i < 4; ++i) { quad[2 * i] = q[order[i]].x; quad[2 * i + 1] = q[order[i]].y; }
        return true;
    }
    return false;
}

}  // namespace cv
