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

// Connected components on RUNS instead of pixels (round 4: the flood fill over 65536 pixels -- twice, foreground and background --
// cost 0.7 ms per mask and was the largest host stage of process_image).  A row is packed to one bit per pixel (8 mask bytes per
// 64-bit multiply), its runs of set bits are read off the transitions with count-trailing-zeros, and a run joins the runs of the
// previous row it touches (8-connectivity: overlap after widening by one pixel; 4-connectivity: plain overlap) through a
// union-find.  Board masks have one to three runs per row, so the whole labelling touches a few hundred runs.
// Result = what the flood fill returned: per component its first pixel in raster order and its bounding box, components in raster
// order of that pixel; `skip_frame_touching` drops components with a pixel on the image frame (background that is not a hole).
struct RowBits {
    int w = 0, words = 0;
    std::vector<uint64_t> bits;                     // h x words, bit x of row y = pixel (y, x) is foreground
    void build(const uint8_t* mask, int h, int w_) {
        w = w_; words = (w + 63) / 64;
        bits.assign((size_t)h * words, 0);
        for (int y = 0; y < h; ++y) {
            const uint8_t* row = mask + (size_t)y * w;
            uint64_t* out = bits.data() + (size_t)y * words;
            int x = 0;
            for (; x + 8 <= w; x += 8) {
                uint64_t v;
                __builtin_memcpy(&v, row + x, 8);
                // high bit of every byte := byte != 0, then gather the eight high bits into one byte
                const uint64_t nz = (v | ((v & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL)) & 0x8080808080808080ULL;
                const uint64_t packed = ((nz >> 7) * 0x0102040810204080ULL) >> 56;
                out[x >> 6] |= packed << (x & 63);
            }
            for (; x < w; ++x)
                if (row[x]) out[x >> 6] |= 1ULL << (x & 63);
        }
    }
};

struct Run { int x0, x1; int parent; };             // [x0, x1) on its row; parent: union-find over run indices

struct RunComp { int fy, fx, x0, x1, y0, y1; bool touches; };

inline int find_root(std::vector<Run>& runs, int i) {
    while (runs[i].parent != i) { runs[i].parent = runs[runs[i].parent].parent; i = runs[i].parent; }
    return i;
}

std::vector<Comp> component_starts(const RowBits& rb, int h, bool invert, bool eight, bool skip_frame_touching) {
    const int w = rb.w, words = rb.words;
    std::vector<Run> runs;
    std::vector<RunComp> comp;                      // indexed like runs; valid at roots
    std::vector<uint64_t> row((size_t)words);
    int prev_begin = 0, prev_end = 0;               // runs of the previous row: [prev_begin, prev_end)
    const uint64_t tail = (w & 63) ? ((1ULL << (w & 63)) - 1) : ~0ULL;
    for (int y = 0; y < h; ++y) {
        const uint64_t* src = rb.bits.data() + (size_t)y * words;
        for (int k = 0; k < words; ++k) row[k] = invert ? ~src[k] : src[k];
        row[words - 1] &= tail;
        const int cur_begin = (int)runs.size();
        // runs of this row: walk the 0->1 and 1->0 transitions
        uint64_t carry = 0;                          // last bit of the previous word
        int open = -1;
        for (int k = 0; k < words; ++k) {
            uint64_t t = row[k] ^ ((row[k] << 1) | carry);
            carry = row[k] >> 63;
            while (t) {
                const int x = k * 64 + __builtin_ctzll(t);
                t &= t - 1;
                if (open < 0) open = x;
                else { runs.push_back({open, x, (int)runs.size()}); open = -1; }
            }
        }
        if (open >= 0) runs.push_back({open, w, (int)runs.size()});
        const int cur_end = (int)runs.size();
        comp.resize(runs.size());
        int p = prev_begin;
        for (int r = cur_begin; r < cur_end; ++r) {
            const int x0 = runs[r].x0, x1 = runs[r].x1;
            comp[r] = {y, x0, x0, x1 - 1, y, y, y == 0 || y == h - 1 || x0 == 0 || x1 == w};
            const int lo = eight ? x0 - 1 : x0, hi = eight ? x1 + 1 : x1;          // a previous run [p0, p1) touches iff p0 < hi && p1 > lo
            while (p < prev_end && runs[p].x1 <= lo) ++p;
            for (int q = p; q < prev_end && runs[q].x0 < hi; ++q) {
                int a = find_root(runs, q), b = find_root(runs, r);
                if (a == b) continue;
                // keep the root whose first pixel comes first in raster order
                const bool a_first = comp[a].fy < comp[b].fy || (comp[a].fy == comp[b].fy && comp[a].fx < comp[b].fx);
                const int root = a_first ? a : b, child = a_first ? b : a;
                runs[child].parent = root;
                comp[root].x0 = std::min(comp[root].x0, comp[child].x0); comp[root].x1 = std::max(comp[root].x1, comp[child].x1);
                comp[root].y0 = std::min(comp[root].y0, comp[child].y0); comp[root].y1 = std::max(comp[root].y1, comp[child].y1);
                comp[root].touches = comp[root].touches || comp[child].touches;
            }
        }
        prev_begin = cur_begin; prev_end = cur_end;
    }
    std::vector<Comp> starts;
    for (int r = 0; r < (int)runs.size(); ++r) {     // runs are in raster order, and a root is the raster-first run of its component
        if (runs[r].parent != r) continue;
        const RunComp& c = comp[r];
        if (skip_frame_touching && c.touches) continue;
        starts.push_back({c.fy, c.fx, c.x1 - c.x0 + 1, c.y1 - c.y0 + 1});
    }
    return starts;
}

// All borders (outer, then holes).  `*total` receives the number of contours the image has; when it exceeds one the
// reference filters by area share >= 0.35, and since a border polygon's area is below its bounding-box area, borders
// whose box is smaller than that can never pass: they are counted but not traced (noisy masks have thousands).
std::vector<Contour> find_contours(const uint8_t* mask, int h, int w, size_t* total) {
    const Image f{mask, h, w};
    RowBits rb;
    rb.build(mask, h, w);
    const std::vector<Comp> outer = component_starts(rb, h, false, true, false);
    const std::vector<Comp> holes = component_starts(rb, h, true, false, true);
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
