// contour.cpp -- binary mask -> board quadrangle on the host, in C++.
//
// Replaces, for the batched pipeline, the chain the reference runs through OpenCV per image
// (chessvision/core.py:357-411): findContours(RETR_CCOMP) -> [area / bounding-box filter when more than one contour]
// -> approxPolyDP(0.1 * perimeter, closed) -> first 4-vertex result -> _rotate_quadrangle.
// Every OpenCV call of that chain is restated in OpenCV's own arithmetic and order (round 5): border following in OpenCV's
// start pixel / direction, CHAIN_APPROX_TC89_KCOS (the Teh-Chin dominant-point pass findContours applies to the traced
// chain, core.py:360), RETR_CCOMP's output order (newest outer border first, each followed by its holes), arcLength with
// float segment lengths, approxPolyDP including its in-place clean-up pass.  Same results as chessvision/classical.py
// (numpy) and as the independent oracle oracle/c_ref/contours_ref.c (raster-scan relabelling); tests/test_contour_cpp.py,
// tests/test_contour_parity.py.
#include <algorithm>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

// the float / double roundings of the restated OpenCV arithmetic are part of the result: no fused multiply-add anywhere here
#pragma clang fp contract(off)

namespace cv {

namespace {

struct Pt { int x, y; };
typedef std::vector<Pt> Contour;

// 8-neighbourhood, clockwise from west, (dy, dx) with y down
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
const int kDx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};

inline int nbr_index(int dy, int dx) {
    static const int8_t kIndex[3][3] = {{1, 2, 3}, {0, 0, 4}, {7, 6, 5}};    // [dy + 1][dx + 1]; the centre never occurs
    return kIndex[dy + 1][dx + 1];
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
#if defined(__SSE2__)
            // 16 pixels per step: byte != 0 -> one bit each (pcmpeqb against zero, pmovmskb, inverted)
            for (; x + 16 <= w; x += 16) {
                const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + x));
                const uint64_t z = (uint64_t)(uint16_t)~_mm_movemask_epi8(_mm_cmpeq_epi8(v, _mm_setzero_si128()));
                out[x >> 6] |= z << (x & 63);
            }
#endif
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

// `lookup` (optional) receives what is needed to ask "which component owns pixel (y, x)" afterwards
struct RunIndex {
    std::vector<Run> runs;
    std::vector<int> row_begin;                     // runs of row y: [row_begin[y], row_begin[y + 1])
    std::vector<int> comp_of_root;                  // root run -> index into the returned starts (-1: skipped)
    int component_at(int y, int x) {
        for (int r = row_begin[y]; r < row_begin[y + 1]; ++r)
            if (runs[r].x0 <= x && x < runs[r].x1) return comp_of_root[find_root(runs, r)];
        return -1;
    }
};

std::vector<Comp> component_starts(const RowBits& rb, int h, bool invert, bool eight, bool skip_frame_touching,
                                   RunIndex* lookup = nullptr) {
    const int w = rb.w, words = rb.words;
    std::vector<Run> runs;
    std::vector<int> row_begin((size_t)h + 1, 0);
    std::vector<RunComp> comp;                      // indexed like runs; valid at roots
    std::vector<uint64_t> row((size_t)words);
    int prev_begin = 0, prev_end = 0;               // runs of the previous row: [prev_begin, prev_end)
    const uint64_t tail = (w & 63) ? ((1ULL << (w & 63)) - 1) : ~0ULL;
    for (int y = 0; y < h; ++y) {
        const uint64_t* src = rb.bits.data() + (size_t)y * words;
        for (int k = 0; k < words; ++k) row[k] = invert ? ~src[k] : src[k];
        row[words - 1] &= tail;
        const int cur_begin = (int)runs.size();
        row_begin[y] = cur_begin;
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
    row_begin[h] = (int)runs.size();
    std::vector<Comp> starts;
    std::vector<int> comp_of_root(lookup ? runs.size() : 0, -1);
    for (int r = 0; r < (int)runs.size(); ++r) {     // runs are in raster order, and a root is the raster-first run of its component
        if (runs[r].parent != r) continue;
        const RunComp& c = comp[r];
        if (skip_frame_touching && c.touches) continue;
        if (lookup) comp_of_root[r] = (int)starts.size();
        starts.push_back({c.fy, c.fx, c.x1 - c.x0 + 1, c.y1 - c.y0 + 1});
    }
    if (lookup) { lookup->runs.swap(runs); lookup->row_begin.swap(row_begin); lookup->comp_of_root.swap(comp_of_root); }
    return starts;
}

// CHAIN_APPROX_TC89_KCOS as cv2.findContours applies it to the traced chain (OpenCV imgproc/src/contours.cpp,
// icvApproximateChainTC89; reference call site core.py:360).  `c` = every border pixel in tracing order (the chain's points).
// pass 0 keeps the points where the chain code changes; pass 1 gives each its Teh-Chin region of support k (grow while the
// chord p[i-k]p[i+k] lengthens and the distance-to-chord / chord ratio rises) and its k-cosine -- the cosine of the angle at
// the point + 1.1, rounded to FLOAT and compared through its bit pattern, largest over j = k .. 1 while it grows; pass 2 is the
// non-maximum suppression over k/2 neighbours (a suppressed point's measure drops to 0 for the points after it); pass 3
// removes 1-support points that do not beat both chain neighbours.
inline int32_t float_bits(float f) { int32_t i; __builtin_memcpy(&i, &f, 4); return i; }

Contour tc89_kcos(const Contour& c) {
    const int len = (int)c.size();
    if (len <= 1) return c;                                  // isolated pixel: chain of length 0 -> its origin
    static const int kAbsDiff[15] = {1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1};
    static const int kCode[3][3] = {{3, 2, 1}, {4, -1, 0}, {5, 6, 7}};        // [dy + 1][dx + 1] -> OpenCV direction code
    std::vector<int> s((size_t)len), k((size_t)len, 0), next((size_t)len, -1);
    auto code_out = [&](int i) { const Pt& a = c[i]; const Pt& b = c[i + 1 < len ? i + 1 : 0]; return kCode[b.y - a.y + 1][b.x - a.x + 1]; };
    int head = -1, tail = -1, prev_code = code_out(len - 1);
    for (int i = 0; i < len; ++i) {
        const int code = code_out(i);
        s[i] = kAbsDiff[code - prev_code + 7];
        if (s[i] != 0) { if (tail < 0) head = i; else next[tail] = i; tail = i; }
        prev_code = code;
    }
    if (head < 0) return c;                                  // cannot happen for a closed chain (OpenCV asserts)
    for (int i = head; i >= 0; i = next[i]) {
        const Pt p0 = c[i];
        int kk, l = 0, d_num = 0;
        for (kk = 1;; ++kk) {
            if (kk > len) return c;                          // OpenCV asserts k <= len
            const Pt& a = c[i - kk < 0 ? i - kk + len : i - kk];
            const Pt& b = c[i + kk >= len ? i + kk - len : i + kk];
            const int dx = b.x - a.x, dy = b.y - a.y;
            const int lk = dx * dx + dy * dy;
            const int dk_num = (p0.x - a.x) * dy - (p0.y - a.y) * dx;
            const int32_t d = float_bits((float)(((double)d_num) * lk - ((double)dk_num) * l));
            if (kk > 1 && (l >= lk || ((d_num > 0 && d <= 0) || (d_num < 0 && d >= 0)))) break;
            d_num = dk_num;
            l = lk;
        }
        k[i] = --kk;
        int sv = 0;
        for (int j = kk; j > 0; --j) {
            const Pt& a = c[i - j < 0 ? i - j + len : i - j];
            const Pt& b = c[i + j >= len ? i + j - len : i + j];
            const int dx1 = a.x - p0.x, dy1 = a.y - p0.y, dx2 = b.x - p0.x, dy2 = b.y - p0.y;
            if ((dx1 | dy1) == 0 || (dx2 | dy2) == 0) break;
            double num = dx1 * dx2 + dy1 * dy2;
            num = (float)(num / std::sqrt(((double)dx1 * dx1 + (double)dy1 * dy1) * ((double)dx2 * dx2 + (double)dy2 * dy2)));
            const int32_t sk = float_bits((float)(num + 1.1));
            if (j < kk && sk <= sv) break;
            sv = sk;
        }
        s[i] = sv;
    }
    auto prune = [&](auto&& drop) {                         // walk the kept list, unlink the points `drop` names
        int prev = -1;
        for (int i = head; i >= 0;) {
            const int nx = next[i];
            if (drop(i)) { if (prev < 0) head = nx; else next[prev] = nx; s[i] = 0; }
            else prev = i;
            i = nx;
        }
    };
    prune([&](int i) {
        const int k2 = k[i] >> 1;
        for (int j = 1; j <= k2; ++j) {
            if (s[i - j < 0 ? i - j + len : i - j] > s[i]) return true;
            if (s[i + j >= len ? i + j - len : i + j] > s[i]) return true;
        }
        return false;
    });
    prune([&](int i) {
        if (k[i] != 1) return false;
        return s[i] <= s[i == 0 ? len - 1 : i - 1] || s[i] <= s[i + 1 == len ? 0 : i + 1];
    });
    Contour out;
    for (int i = head; i >= 0; i = next[i]) out.push_back(c[i]);
    return out;
}

// What cv2.findContours(mask, RETR_CCOMP, method) returns, in its order: OpenCV links every new contour in front of its parent's
// children and walks the two-level tree in pre-order, i.e. outer borders from the LAST found (raster order of their first pixel)
// to the first, each followed by the borders of its holes, last found first.  `*total` receives the number of contours the image
// has; when it exceeds one the reference filters by area share >= 0.35, and since a border polygon's area is below its
// bounding-box area, borders whose box is smaller than that can never pass: with `prune` they are counted but not traced (noisy
// masks have thousands).
struct Border { Contour pts; bool hole; };

std::vector<Border> find_contours(const uint8_t* mask, int h, int w, size_t* total, bool tc89, bool prune_small) {
    const Image f{mask, h, w};
    RowBits rb;
    rb.build(mask, h, w);
    RunIndex fg;
    const std::vector<Comp> outer = component_starts(rb, h, false, true, false, &fg);
    // a hole needs a row with foreground on both sides of it: while no row has two foreground runs every background run reaches the
    // frame and the second labelling (clean board masks: a third of this function's time) has nothing to find
    bool may_have_holes = false;
    for (int y = 0; y < h && !may_have_holes; ++y) may_have_holes = fg.row_begin[y + 1] - fg.row_begin[y] > 1;
    const std::vector<Comp> holes = may_have_holes ? component_starts(rb, h, true, false, true) : std::vector<Comp>();
    *total = outer.size() + holes.size();
    const bool prune = prune_small && *total > 1;
    const double need = 0.35 * (double)h * w;
    // a hole border runs on the foreground pixels around the hole (box 2 wider / taller) and belongs to the component that owns
    // the pixel left of the hole's first pixel
    std::vector<std::vector<int>> holes_of(outer.size());
    for (int q = 0; q < (int)holes.size(); ++q) {
        if (prune && (double)(holes[q].bw + 2) * (holes[q].bh + 2) < need) continue;
        const int owner = fg.component_at(holes[q].y, holes[q].x - 1);
        if (owner >= 0) holes_of[owner].push_back(q);
    }
    std::vector<Border> out;
    for (int a = (int)outer.size() - 1; a >= 0; --a) {
        const Comp& s = outer[a];
        if (!prune || (double)s.bw * s.bh >= need) {
            Contour c = trace_border(f, s.y, s.x, s.y, s.x - 1);
            out.push_back({tc89 ? tc89_kcos(c) : c, false});
        }
        for (int q = (int)holes_of[a].size() - 1; q >= 0; --q) {
            const Comp& hs = holes[holes_of[a][q]];
            Contour c = trace_border(f, hs.y, hs.x - 1, hs.y, hs.x);
            out.push_back({tc89 ? tc89_kcos(c) : c, true});
        }
    }
    return out;
}

// cv2.contourArea: shoelace, exact in double for pixel coordinates
double contour_area(const Contour& c) {
    if (c.empty()) return 0.0;
    double a00 = 0.0;
    Pt prev = c.back();
    for (const Pt& p : c) { a00 += (double)prev.x * p.y - (double)prev.y * p.x; prev = p; }
    return std::fabs(a00 * 0.5);
}

// cv2.arcLength(closed): OpenCV converts the points to Point2f and takes every segment's length in FLOAT (std::sqrt(float)),
// summing in double, closing segment first
double arc_length(const Contour& c) {
    if (c.size() <= 1) return 0.0;
    double perimeter = 0.0;
    float px = (float)c.back().x, py = (float)c.back().y;
    for (const Pt& p : c) {
        const float x = (float)p.x, y = (float)p.y, dx = x - px, dy = y - py;
        perimeter += std::sqrt(dx * dx + dy * dy);
        px = x; py = y;
    }
    return perimeter;
}

// cv2.approxPolyDP(closed = true) for integer points, OpenCV imgproc/src/approx.cpp approxPolyDP_<int>: three farthest-point hops
// choose the first split, a stack of (start, end) slices drives Douglas-Peucker (a slice's start point is emitted when nothing in
// it is farther than epsilon from its chord), then ONE clean-up pass over the result in place: a vertex within sqrt(0.5) epsilon of
// the chord of its neighbours (chord not axis-parallel, vertex between them) is dropped and its successor kept unexamined.
Contour approx_poly_dp(const Contour& src, double epsilon) {
    const int count = (int)src.size();
    Contour dst;
    if (count == 0) return dst;
    const double eps = epsilon * epsilon;
    auto advance = [count](int& pos) { if (++pos >= count) pos = 0; };
    int pos = 0, right_start = 0;
    bool le_eps = false;
    Pt start_pt{-1000000, -1000000};
    for (int it = 0; it < 3; ++it) {
        double max_dist = 0;
        pos = (pos + right_start) % count;
        start_pt = src[pos]; advance(pos);
        for (int j = 1; j < count; ++j) {
            const Pt pt = src[pos]; advance(pos);
            const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_start = j; }
        }
        le_eps = max_dist <= eps;
    }
    std::vector<std::pair<int, int>> stack;
    if (!le_eps) {
        const int a = pos % count, b = (right_start + a) % count;
        stack.push_back({b, a});
        stack.push_back({a, b});
    } else dst.push_back(start_pt);
    int split = right_start;
    while (!stack.empty()) {
        const int s = stack.back().first, e = stack.back().second;
        stack.pop_back();
        const Pt end_pt = src[e];
        pos = s;
        start_pt = src[pos]; advance(pos);
        if (pos != e) {
            double max_dist = 0;
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            while (pos != e) {
                const Pt pt = src[pos]; advance(pos);
                const double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; split = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else le_eps = true;
        if (le_eps) dst.push_back(start_pt);
        else { stack.push_back({split, e}); stack.push_back({s, split}); }
    }
    const int cnt = (int)dst.size();
    int new_count = cnt, rpos = cnt - 1, wpos;
    auto read = [&](Pt& p) { p = dst[rpos]; if (++rpos >= cnt) rpos = 0; };
    Pt pt, end_pt;
    read(start_pt);
    wpos = rpos;
    read(pt);
    for (int i = 0; i < cnt && new_count > 2; ++i) {
        read(end_pt);
        const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        const double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        const double inner = (double)((pt.x - start_pt.x) * (end_pt.x - pt.x) + (pt.y - start_pt.y) * (end_pt.y - pt.y));
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
            --new_count;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            read(pt);
            ++i;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    dst.resize((size_t)new_count);
    return dst;
}

}  // namespace

// quad: 4 x (x, y) in mask pixels after the reference's rotation rule; returns true when a quadrangle was found
bool find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8]) {
    size_t total = 0;
    std::vector<Border> contours = find_contours(mask, h, w, &total, true, true);
    const double area = (double)h * w;
    for (auto& b : contours) {
        const Contour& c = b.pts;
        if (total > 1) {                               // reference core.py:362-366, 381-404
            const double share = contour_area(c) / area;
            if (share < 0.35 || share > 1.0) continue;
            int x0 = c[0].x, x1 = c[0].x, y0 = c[0].y, y1 = c[0].y;
            for (auto& p : c) { x0 = std::min(x0, p.x); x1 = std::max(x1, p.x); y0 = std::min(y0, p.y); y1 = std::max(y1, p.y); }
            const double bw = x1 - x0 + 1, bh = y1 - y0 + 1;                 // cv2.boundingRect
            const double r = (bw == 0 || bh == 0) ? -1.0 : std::min(bw, bh) / std::max(bw, bh);
            if (r < 0.6) continue;
        }
        Contour q = approx_poly_dp(c, 0.1 * arc_length(c));
        if (q.size() != 4) continue;
        int order[4] = {0, 1, 2, 3};
        if (q[0].x < q[2].x) { order[0] = 3; order[1] = 0; order[2] = 1; order[3] = 2; }   // core.py:406-411
        for (int i = 0; i < 4; ++i) { quad[2 * i] = q[order[i]].x; quad[2 * i + 1] = q[order[i]].y; }
        return true;
    }
    return false;
}

// cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_NONE | CHAIN_APPROX_TC89_KCOS)[0] flattened (parity tests and callers that want
// the contours themselves).  Returns the number of contours, or -1 when a capacity is too small.
long find_contours_flat(const uint8_t* mask, int h, int w, bool tc89, int32_t* xy, long cap_pts, int32_t* counts, int32_t* holes,
                        long cap_contours) {
    size_t total = 0;
    std::vector<Border> contours = find_contours(mask, h, w, &total, tc89, false);
    if ((long)contours.size() > cap_contours) return -1;
    long used = 0;
    for (size_t q = 0; q < contours.size(); ++q) {
        const Contour& c = contours[q].pts;
        if (used + (long)c.size() > cap_pts) return -1;
        for (const Pt& p : c) { xy[2 * used] = p.x; xy[2 * used + 1] = p.y; ++used; }
        counts[q] = (int32_t)c.size();
        holes[q] = contours[q].hole ? 1 : 0;
    }
    return (long)contours.size();
}

}  // namespace cv
