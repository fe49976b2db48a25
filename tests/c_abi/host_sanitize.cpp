// (test infrastructure) The host-side C++ of the library -- mask -> contours -> quadrangle (contour.cpp), quadrangle -> homographies
// (homography.cpp), probabilities -> labels / FEN / pawn rule (position.cpp) -- driven over a few thousand generated inputs in a binary
// built with AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_sanitizers.py compiles the three units from csrc/ together with
// this file; GPU sanitizers are not available on the pool, the host code is what CAN be checked this way).  The inputs aim at the edges:
// 1 x 1 and single-row masks, odd sizes, shapes touching the frame, holes inside holes, one-pixel lines, noise at every density, capacities
// one short of what a call needs, collinear and repeated quadrangle corners, ties and NaNs in the probability tables.  What this binary
// proves is that none of it reads or writes out of bounds, overflows a signed integer or shifts out of range -- and, built with
// -DWITH_ORACLE, that on every generated mask the contours (both approximation methods) and the quadrangle equal those of the
// independent plain-C oracle (compiled in with the same sanitizers).  Homographies and positions are sanity-checked only (their value
// parity: test_classical.py, test_classical_ref.py, test_core_api.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace cv {
void decode_positions(const float* probs, int n_boards, int flip, char* fen, char* original_fen, int8_t* labels, int32_t* fixes, int32_t* n_fixes);
bool find_quadrangle(const uint8_t* mask, int h, int w, int32_t quad[8]);
long find_contours_flat(const uint8_t* mask, int h, int w, bool tc89, int32_t* xy, long cap_pts, int32_t* counts, int32_t* holes, long cap_contours);
void board_homographies(const float* quads, int n, int out_w, int out_h, double* forward, double* inverse);
}  // namespace cv

#ifdef WITH_ORACLE
// the independent plain-C restatement of the same OpenCV chain (oracle/c_ref/contours_ref.c, raster-scan relabelling instead of border
// following from start pixels), compiled into this binary with the same sanitizers: every generated mask is also a differential test
extern "C" {
int ref_find_contours(const uint8_t* mask, int h, int w, int method, int* xy, int cap_pts, int* counts, int* holes, int cap_contours);
int ref_find_quadrangle(const uint8_t* mask, int h, int w, int* quad);
}
#endif

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    int below(int n) { return (int)(next() % (uint64_t)n); }
    float unit() { return (float)((next() >> 11) * (1.0 / 9007199254740992.0)); }
};

void fill_polygon(std::vector<uint8_t>& m, int h, int w, const int* xs, const int* ys, int n, uint8_t v) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            bool in = false;
            for (int i = 0, j = n - 1; i < n; j = i++)
                if (((ys[i] > y) != (ys[j] > y)) && (x < (double)(xs[j] - xs[i]) * (y - ys[i]) / (double)(ys[j] - ys[i]) + xs[i])) in = !in;
            if (in) m[(size_t)y * w + x] = v;
        }
}

// one generated mask per (kind, seed); kinds cover what the doc comment lists
std::vector<uint8_t> make_mask(int kind, int h, int w, Rng& r) {
    std::vector<uint8_t> m((size_t)h * w, 0);
    switch (kind) {
    case 0: break;                                                       // empty
    case 1: std::fill(m.begin(), m.end(), 255); break;                   // full: the one contour is the frame
    case 2: for (auto& v : m) v = r.below(2) ? 255 : 0; break;           // salt and pepper
    case 3: { const int d = 1 + r.below(20); for (auto& v : m) v = r.below(20) < d ? 1 + r.below(255) : 0; break; }   // any non-zero value counts
    case 4: for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) m[(size_t)y * w + x] = ((x ^ y) & 1) ? 255 : 0; break;   // checkerboard
    case 5: {                                                            // a quadrilateral, possibly clipped by the frame
        int xs[4], ys[4];
        for (int i = 0; i < 4; ++i) { xs[i] = r.below(w + 8) - 4; ys[i] = r.below(h + 8) - 4; }
        fill_polygon(m, h, w, xs, ys, 4, 255);
        break;
    }
    case 6: {                                                            // rings: holes inside outers inside holes
        const int cx = w / 2, cy = h / 2;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) { const int d = std::max(std::abs(x - cx), std::abs(y - cy)); m[(size_t)y * w + x] = (d / (1 + r.s % 3)) & 1 ? 255 : 0; }
        break;
    }
    case 7: {                                                            // one-pixel lines and isolated pixels
        for (int k = 0; k < 6; ++k) {
            int x = r.below(w), y = r.below(h);
            const int dx = r.below(3) - 1, dy = r.below(3) - 1, len = r.below(std::max(h, w));
            for (int i = 0; i < len && x >= 0 && x < w && y >= 0 && y < h; ++i, x += dx, y += dy) m[(size_t)y * w + x] = 255;
        }
        break;
    }
    case 8: {                                                            // a large convex blob with pits on its edge (many dominant points)
        int xs[4] = {w / 8, w - w / 8, w - w / 6, w / 6}, ys[4] = {h / 8, h / 7, h - h / 8, h - h / 7};
        fill_polygon(m, h, w, xs, ys, 4, 255);
        for (int k = 0; k < 40; ++k) m[(size_t)r.below(h) * w + r.below(w)] ^= 255;
        break;
    }
    default: {                                                           // frame-hugging: first / last rows and columns set
        for (int x = 0; x < w; ++x) { m[x] = 255; m[(size_t)(h - 1) * w + x] = 255; }
        for (int y = 0; y < h; ++y) { m[(size_t)y * w] = 255; m[(size_t)y * w + w - 1] = 255; }
        if (h > 4 && w > 4) m[(size_t)(h / 2) * w + w / 2] = 255;
    }
    }
    return m;
}

long g_masks = 0, g_quads = 0, g_contours = 0, g_short = 0, g_diff = 0;

int check_mask(const std::vector<uint8_t>& m, int h, int w) {
    ++g_masks;
    int32_t quad[8] = {0};
    const bool found = cv::find_quadrangle(m.data(), h, w, quad);
#ifdef WITH_ORACLE
    {
        int rq[8] = {0};
        const int rf = ref_find_quadrangle(m.data(), h, w, rq);
        if (rf < 0 || (rf == 1) != found || (found && std::memcmp(rq, quad, sizeof(rq)) != 0)) {
            std::fprintf(stderr, "quadrangle differs from the oracle's on a %d x %d mask (found %d vs %d)\n", h, w, (int)found, rf);
            return 1;
        }
        ++g_diff;
    }
#endif
    if (found) {
        ++g_quads;
        for (int i = 0; i < 4; ++i)
            if (quad[2 * i] < 0 || quad[2 * i] >= w || quad[2 * i + 1] < 0 || quad[2 * i + 1] >= h) { std::fprintf(stderr, "quadrangle vertex outside a %d x %d mask\n", h, w); return 1; }
    }
    // generous capacities first (every border pixel can appear up to four times on a traced chain), then exactly enough, then one short
    const long cap_pts = 4L * h * w + 16, cap_cnt = (long)h * w + 4;
    std::vector<int32_t> xy((size_t)2 * cap_pts), counts((size_t)cap_cnt), holes((size_t)cap_cnt);
    for (int tc89 = 0; tc89 < 2; ++tc89) {
        const long n = cv::find_contours_flat(m.data(), h, w, tc89 != 0, xy.data(), cap_pts, counts.data(), holes.data(), cap_cnt);
        if (n < 0) { std::fprintf(stderr, "capacity 4*h*w not enough for a %d x %d mask\n", h, w); return 1; }
        g_contours += n;
        long pts = 0;
        for (long q = 0; q < n; ++q) {
            if (counts[q] <= 0 || (holes[q] != 0 && holes[q] != 1)) { std::fprintf(stderr, "bad contour record\n"); return 1; }
            pts += counts[q];
        }
#ifdef WITH_ORACLE
        {
            std::vector<int> rxy((size_t)2 * cap_pts), rc((size_t)cap_cnt), rh((size_t)cap_cnt);
            const int rn = ref_find_contours(m.data(), h, w, tc89, rxy.data(), (int)cap_pts, rc.data(), rh.data(), (int)cap_cnt);
            bool same = rn == n;
            for (long q = 0; same && q < n; ++q) same = rc[q] == counts[q] && rh[q] == holes[q];
            same = same && std::memcmp(rxy.data(), xy.data(), (size_t)2 * pts * sizeof(int)) == 0;
            if (!same) { std::fprintf(stderr, "contours (method %d) differ from the oracle's on a %d x %d mask: %ld vs %d\n", tc89, h, w, n, rn); return 1; }
            ++g_diff;
        }
#endif
        for (long i = 0; i < pts; ++i)
            if (xy[2 * i] < 0 || xy[2 * i] >= w || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= h) { std::fprintf(stderr, "contour point outside the mask\n"); return 1; }
        if (n > 0) {
            std::vector<int32_t> xy2((size_t)2 * pts), c2((size_t)n), h2((size_t)n);           // exact fit: ASan sees any write past the end
            if (cv::find_contours_flat(m.data(), h, w, tc89 != 0, xy2.data(), pts, c2.data(), h2.data(), n) != n) { std::fprintf(stderr, "exact capacities refused\n"); return 1; }
            if (std::memcmp(xy2.data(), xy.data(), (size_t)2 * pts * sizeof(int32_t)) != 0) { std::fprintf(stderr, "two calls, two answers\n"); return 1; }
            std::vector<int32_t> xy3((size_t)2 * std::max(1L, pts - 1));
            if (cv::find_contours_flat(m.data(), h, w, tc89 != 0, xy3.data(), pts - 1, c2.data(), h2.data(), n) != -1) { std::fprintf(stderr, "one point short accepted\n"); return 1; }
            if (n > 1) {
                std::vector<int32_t> c3((size_t)(n - 1)), h3((size_t)(n - 1));
                if (cv::find_contours_flat(m.data(), h, w, tc89 != 0, xy2.data(), pts, c3.data(), h3.data(), n - 1) != -1) { std::fprintf(stderr, "one contour short accepted\n"); return 1; }
            }
            g_short += 2;
        }
    }
    return 0;
}

int run_masks() {
    static const int sizes[][2] = {{1, 1}, {1, 2}, {2, 1}, {1, 17}, {17, 1}, {2, 2}, {3, 3}, {3, 64}, {64, 3}, {5, 7}, {16, 16}, {31, 33}, {37, 53}, {64, 64}, {65, 63}, {128, 96}, {256, 256}};
    for (auto& sz : sizes) {
        const int h = sz[0], w = sz[1];
        const int reps = (h * w <= 4096) ? 24 : (h * w <= 16384 ? 6 : 2);
        for (int kind = 0; kind < 10; ++kind)
            for (int rep = 0; rep < reps; ++rep) {
                Rng r((uint64_t)(h * 1000003 + w * 1009 + kind * 31 + rep));
                if (check_mask(make_mask(kind, h, w, r), h, w)) return 1;
            }
    }
    return 0;
}

int run_homographies() {
    Rng r(99);
    long degenerate = 0, finite = 0;
    for (int iter = 0; iter < 4000; ++iter) {
        const int n = 1 + r.below(5);
        std::vector<float> quads((size_t)n * 8);
        for (int b = 0; b < n; ++b) {
            float* q = &quads[(size_t)b * 8];
            const int kind = r.below(8);
            for (int i = 0; i < 8; ++i) q[i] = r.unit() * 512.f;
            if (kind == 0) for (int i = 2; i < 8; ++i) q[i] = q[i & 1];                         // four times the same point
            if (kind == 1) for (int i = 0; i < 4; ++i) { q[2 * i] = 10.f * i; q[2 * i + 1] = 20.f * i; }   // collinear
            if (kind == 2) { q[4] = q[0]; q[5] = q[1]; }                                        // two equal corners
            if (kind == 3) for (int i = 0; i < 8; ++i) q[i] = (float)(r.below(3) - 1) * 1e30f;   // huge
            if (kind == 4) q[r.below(8)] = std::numeric_limits<float>::quiet_NaN();
            if (kind == 5) q[r.below(8)] = std::numeric_limits<float>::infinity();
        }
        std::vector<double> fwd((size_t)n * 9), inv((size_t)n * 9);
        const int ow = 1 + r.below(1024), oh = 1 + r.below(1024);
        cv::board_homographies(quads.data(), n, ow, oh, fwd.data(), inv.data());
        cv::board_homographies(quads.data(), n, ow, oh, nullptr, inv.data());                  // the pipeline's form: inverse only
        cv::board_homographies(quads.data(), n, ow, oh, fwd.data(), nullptr);
        for (int b = 0; b < n; ++b) {
            bool zero = true, fin = true;
            for (int i = 0; i < 9; ++i) { zero = zero && fwd[(size_t)b * 9 + i] == 0.0; fin = fin && std::isfinite(fwd[(size_t)b * 9 + i]); }
            degenerate += zero; finite += fin && !zero;
        }
    }
    std::printf("homographies: %ld regular, %ld degenerate (all-zero) maps\n", finite, degenerate);
    return 0;
}

int run_positions() {
    Rng r(7);
    long boards = 0, fixes_total = 0;
    for (int iter = 0; iter < 600; ++iter) {
        const int n = 1 + r.below(9), flip = r.below(2);
        std::vector<float> probs((size_t)n * 64 * 13);
        const int kind = r.below(7);
        for (auto& p : probs) p = r.unit();
        if (kind == 1) std::fill(probs.begin(), probs.end(), 1.f / 13.f);                       // thirteen-way ties everywhere
        if (kind == 2) for (size_t i = 0; i < probs.size(); i += 13) { for (int k = 0; k < 13; ++k) probs[i + k] = 0.f; probs[i + (r.below(2) ? 3 : 9)] = 1.f; }   // pawns everywhere: the rule fires on 16 squares
        if (kind == 3) for (auto& p : probs) if (r.below(50) == 0) p = std::numeric_limits<float>::quiet_NaN();
        if (kind == 4) for (auto& p : probs) p = r.below(40) == 0 ? std::numeric_limits<float>::infinity() : p;
        if (kind == 5) for (size_t i = 0; i < probs.size(); i += 13) { for (int k = 0; k < 13; ++k) probs[i + k] = 0.f; probs[i + 12] = 1.f; }   // empty boards: "8/8/8/8/8/8/8/8"
        if (kind == 6) for (auto& p : probs) p = -p;                                             // not probabilities at all
        std::vector<char> fen((size_t)n * 72, '#'), orig((size_t)n * 72, '#');
        std::vector<int8_t> labels((size_t)n * 64, -1);
        std::vector<int32_t> fixes((size_t)n * 16 * 4, -1);
        int32_t n_fixes = -1;
        cv::decode_positions(probs.data(), n, flip, fen.data(), orig.data(), labels.data(), fixes.data(), &n_fixes);
        if (n_fixes < 0 || n_fixes > n * 16) { std::fprintf(stderr, "n_fixes %d for %d boards\n", n_fixes, n); return 1; }
        for (int b = 0; b < n; ++b) {
            const size_t len = strnlen(&fen[(size_t)b * 72], 72), len0 = strnlen(&orig[(size_t)b * 72], 72);
            if (len == 0 || len > 71 || len0 == 0 || len0 > 71) { std::fprintf(stderr, "FEN of length %zu / %zu\n", len, len0); return 1; }
            for (int i = 0; i < 64; ++i)
                if (labels[(size_t)b * 64 + i] < 0 || labels[(size_t)b * 64 + i] > 12) { std::fprintf(stderr, "label out of range\n"); return 1; }
        }
        for (int f = 0; f < n_fixes; ++f) {
            const int32_t* x = &fixes[(size_t)f * 4];
            if (x[0] < 0 || x[0] >= n || x[1] < 0 || x[1] >= 64 || x[2] < 0 || x[2] > 12 || x[3] < 0 || x[3] > 12) { std::fprintf(stderr, "bad fix record\n"); return 1; }
        }
        boards += n; fixes_total += n_fixes;
    }
    std::printf("positions: %ld boards decoded, %ld pawn-rule fixes\n", boards, fixes_total);
    return 0;
}

}  // namespace

int main() {
    if (run_masks()) return 1;
    std::printf("masks: %ld masks, %ld quadrangles, %ld contours, %ld under-capacity calls refused\n", g_masks, g_quads, g_contours, g_short);
    std::printf("oracle: %ld results compared with oracle/c_ref/contours_ref.c, all equal\n", g_diff);
    if (run_homographies()) return 1;
    if (run_positions()) return 1;
    std::printf("host sanitizers: ok\n");
    return 0;
}
