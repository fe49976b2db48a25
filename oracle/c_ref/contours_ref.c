/* contours_ref.c -- the mask -> quadrangle chain of the reference, restated literally in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Shares no code with the product (csrc/contour.cpp labels components on
 * bit-packed runs and traces from component starts; chessvision/classical.py uses scipy labels): here the image is scanned in
 * raster order and relabelled in place, exactly as the published algorithm does, so a misreading on the product's side shows.
 *
 * Reference call sites (chessvision/core.py):
 *   357-379  cv2.findContours(mask, RETR_CCOMP, CHAIN_APPROX_TC89_KCOS) -> [filter when > 1 contour] ->
 *            for each: cv2.arcLength(closed) -> cv2.approxPolyDP(0.1 * arclen, closed) -> first with 4 vertices -> rotate
 *   381-404  _filter_contours: cv2.contourArea / (h*w) in [0.35, 1.0], utils.ratio(h, w) of cv2.boundingRect >= 0.6
 *   406-411  _rotate_quadrangle
 *
 * The arithmetic lives in opencv-python 4.11.0.86 (uv.lock:2663-2664), absent from /root/reference; restated from the
 * algorithms it publishes (imgproc/src/contours.cpp, approx.cpp, shapedescr.cpp of the 4.x line):
 *   * findContours = Suzuki & Abe 1985, Algorithm 1, on the image padded by one zero pixel (the C++ wrapper's copyMakeBorder):
 *     raster scan, an outer border starts at f(i,j) == 1 with f(i,j-1) == 0, a hole border at f(i,j) >= 1 with f(i,j+1) == 0,
 *     followed pixels are relabelled NBD / -NBD (the minus sign marks "the right neighbour was an examined 0-pixel"), parents by
 *     the paper's LNBD table.  Border following in OpenCV's direction codes (0 = east, counter-clockwise on screen: NE, N, NW, W, SW, S, SE):
 *     first neighbour searched CLOCKWISE from west (outer) / east (hole), then counter-clockwise from the previous pixel, one
 *     chain code per step, the chain closes on the step back into the start pixel.
 *   * RETR_CCOMP output order: every new contour is linked in FRONT of its parent's child list (cvInsertNodeIntoTree) and the
 *     C++ wrapper walks the tree in pre-order: outer borders from the last found to the first, each followed by its holes, last
 *     found first.
 *   * CHAIN_APPROX_TC89_KCOS = icvApproximateChainTC89: pass 0 drops points of zero 1-curvature, pass 1 finds each point's
 *     region of support (Teh-Chin) and its k-cosine in FLOAT compared through its bit pattern, pass 2 non-maximum suppression
 *     over half the support, pass 3 drops 1-support points that do not beat both neighbours.
 *   * arcLength: every segment sqrtf(dx*dx + dy*dy) in FLOAT, summed in double, starting with the closing segment.
 *   * contourArea: shoelace in double; boundingRect: max - min + 1.
 *   * approxPolyDP (closed): three farthest-point hops, stack-driven Douglas-Peucker, the final in-place clean-up pass.
 * PARITY UNPINNED against OpenCV itself (not installable here); pinned on the reference's 631 label masks <-> coordinates.json.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int x, y; } pt_t;

static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct {
    int is_hole, parent;            /* parent: index of the enclosing border in discovery order, -1 = the frame */
    pt_t origin;                    /* in image coordinates (padding removed) */
    int n_codes, cap;
    signed char* codes;
} border_t;

static void push_code(border_t* b, int code) {
    if (b->n_codes == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 64;
        b->codes = (signed char*)realloc(b->codes, (size_t)b->cap);
    }
    b->codes[b->n_codes++] = (signed char)code;
}

/* Algorithm 1 step 3: follow the border that starts at (i, j); F is the padded label image */
static void follow_border(int* F, int stride, int i, int j, int is_hole, int nbd, border_t* b) {
    const int from = is_hole ? 0 : 4;        /* (i2, j2): east of the start for a hole border, west for an outer border */
    int s = from, found = 0;
    for (int n = 0; n < 7; ++n) {            /* 3.1: clockwise around (i, j), starting next to (i2, j2) */
        s = (s - 1) & 7;
        if (F[(i + DY[s]) * stride + j + DX[s]] != 0) { found = 1; break; }
    }
    if (!found) { F[i * stride + j] = -nbd; return; }           /* isolated pixel: chain of length 0 */
    const int i1 = i + DY[s], j1 = j + DX[s];
    int i3 = i, j3 = j;                      /* 3.2 */
    for (;;) {
        int t = s, east_zero = 0;            /* 3.3: counter-clockwise around (i3, j3), starting after (i2, j2) = direction s */
        for (int n = 0; n < 8; ++n) {
            t = (t + 1) & 7;
            if (F[(i3 + DY[t]) * stride + j3 + DX[t]] != 0) break;
            if (t == 0) east_zero = 1;
        }
        int* cell = &F[i3 * stride + j3];    /* 3.4 */
        if (east_zero) *cell = -nbd;
        else if (*cell == 1) *cell = nbd;
        push_code(b, t);
        const int i4 = i3 + DY[t], j4 = j3 + DX[t];
        if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;  /* 3.5 */
        i3 = i4; j3 = j4;
        s = (t + 4) & 7;
    }
}

typedef struct { border_t* v; int n, cap; } borders_t;

static border_t* new_border(borders_t* bs) {
    if (bs->n == bs->cap) {
        bs->cap = bs->cap ? bs->cap * 2 : 16;
        bs->v = (border_t*)realloc(bs->v, sizeof(border_t) * (size_t)bs->cap);
    }
    border_t* b = &bs->v[bs->n++];
    memset(b, 0, sizeof(*b));
    return b;
}

static void free_borders(borders_t* bs) {
    for (int k = 0; k < bs->n; ++k) free(bs->v[k].codes);
    free(bs->v);
}

/* the raster scan of Algorithm 1; borders come out in discovery order with the paper's parent table applied */
static void scan_borders(const uint8_t* mask, int h, int w, borders_t* bs) {
    const int stride = w + 2;
    int* F = (int*)calloc((size_t)(h + 2) * stride, sizeof(int));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) F[(y + 1) * stride + x + 1] = mask[(size_t)y * w + x] != 0;
    int nbd = 1;
    for (int i = 1; i <= h; ++i) {
        int lnbd = 1;
        for (int j = 1; j <= w; ++j) {
            const int v = F[i * stride + j];
            if (v == 0) continue;
            int start = 0, is_hole = 0;
            if (v == 1 && F[i * stride + j - 1] == 0) start = 1;                       /* (a) outer border */
            else if (v >= 1 && F[i * stride + j + 1] == 0) {                             /* (b) hole border */
                start = 1; is_hole = 1;
                if (v > 1) lnbd = v;
            }
            if (start) {
                ++nbd;
                /* (2) parent: B' = border number lnbd (1 = the frame, a hole border) */
                const int prev_is_hole = lnbd == 1 ? 1 : bs->v[lnbd - 2].is_hole;
                const int prev_parent = lnbd == 1 ? -1 : bs->v[lnbd - 2].parent;
                const int parent = (prev_is_hole == is_hole) ? prev_parent : (lnbd == 1 ? -1 : lnbd - 2);
                border_t* b = new_border(bs);
                b->is_hole = is_hole; b->parent = parent;
                b->origin.x = j - 1; b->origin.y = i - 1;
                follow_border(F, stride, i, j, is_hole, nbd, b);
            }
            const int now = F[i * stride + j];                                          /* (4) */
            if (now != 1) lnbd = now < 0 ? -now : now;
        }
    }
    free(F);
}

/* RETR_CCOMP order: outer borders newest first, each followed by its holes newest first */
static int ccomp_order(const borders_t* bs, int* order) {
    int n = 0;
    for (int a = bs->n - 1; a >= 0; --a) {
        if (bs->v[a].is_hole) continue;
        order[n++] = a;
        for (int c = bs->n - 1; c > a; --c)
            if (bs->v[c].is_hole && bs->v[c].parent == a) order[n++] = c;
    }
    return n;                                 /* == bs->n unless a hole had no outer parent (cannot happen) */
}

/* every point of the chain (CHAIN_APPROX_NONE) */
static int chain_points(const border_t* b, pt_t* out) {
    if (b->n_codes == 0) { out[0] = b->origin; return 1; }
    pt_t p = b->origin;
    for (int k = 0; k < b->n_codes; ++k) {
        out[k] = p;
        p.x += DX[(int)b->codes[k]]; p.y += DY[(int)b->codes[k]];
    }
    return b->n_codes;
}

/* ---- CHAIN_APPROX_TC89_KCOS ------------------------------------------------------------------------------------------ */
typedef struct { pt_t pt; int k, s, next; } ptinfo_t;       /* next: index of the next kept point, -1 = end of the list */

static int32_t float_bits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

static int tc89_kcos(const border_t* b, pt_t* out) {
    static const int abs_diff[15] = {1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1};
    const int len = b->n_codes;
    if (len == 0) { out[0] = b->origin; return 1; }
    ptinfo_t* a = (ptinfo_t*)calloc((size_t)len, sizeof(ptinfo_t));
    int head = -1, tail = -1;
    /* pass 0: all points of the digital curve; those of zero 1-curvature leave the list */
    {
        pt_t p = b->origin;
        int prev_code = b->codes[len - 1];
        for (int i = 0; i < len; ++i) {
            const int code = b->codes[i];
            const int s = abs_diff[code - prev_code + 7];
            a[i].pt = p; a[i].s = s; a[i].next = -1;
            if (s != 0) {
                if (tail < 0) head = i; else a[tail].next = i;
                tail = i;
            }
            p.x += DX[code]; p.y += DY[code];
            prev_code = code;
        }
    }
    if (head < 0) { free(a); return -1; }     /* OpenCV asserts here; a closed chain always turns somewhere */
    /* pass 1: region of support and k-cosine */
    for (int cur = head; cur >= 0; cur = a[cur].next) {
        const int i = cur;
        const pt_t p0 = a[i].pt;
        int k, l = 0, d_num = 0;
        for (k = 1;; ++k) {
            if (k > len) { free(a); return -1; }
            int i1 = i - k; if (i1 < 0) i1 += len;
            int i2 = i + k; if (i2 >= len) i2 -= len;
            const int dx = a[i2].pt.x - a[i1].pt.x, dy = a[i2].pt.y - a[i1].pt.y;
            const int lk = dx * dx + dy * dy;
            const int dk_num = (p0.x - a[i1].pt.x) * dy - (p0.y - a[i1].pt.y) * dx;
            const float d = (float)(((double)d_num) * lk - ((double)dk_num) * l);
            const int32_t di = float_bits(d);
            if (k > 1 && (l >= lk || ((d_num > 0 && di <= 0) || (d_num < 0 && di >= 0)))) break;
            d_num = dk_num;
            l = lk;
        }
        a[cur].k = --k;
        int s = 0;
        for (int j = k; j > 0; --j) {
            int i1 = i - j; if (i1 < 0) i1 += len;
            int i2 = i + j; if (i2 >= len) i2 -= len;
            const int dx1 = a[i1].pt.x - p0.x, dy1 = a[i1].pt.y - p0.y;
            const int dx2 = a[i2].pt.x - p0.x, dy2 = a[i2].pt.y - p0.y;
            if ((dx1 | dy1) == 0 || (dx2 | dy2) == 0) break;
            double num = dx1 * dx2 + dy1 * dy2;
            num = (float)(num / sqrt(((double)dx1 * dx1 + (double)dy1 * dy1) * ((double)dx2 * dx2 + (double)dy2 * dy2)));
            const float sk = (float)(num + 1.1);
            const int32_t ski = float_bits(sk);
            if (j < k && ski <= s) break;
            s = ski;
        }
        a[cur].s = s;
    }
    /* pass 2: non-maxima suppression over half the region of support */
    {
        int prev = -1;
        for (int cur = head; cur >= 0;) {
            const int k2 = a[cur].k >> 1, s = a[cur].s, i = cur;
            int j;
            for (j = 1; j <= k2; ++j) {
                int i2 = i - j; if (i2 < 0) i2 += len;
                if (a[i2].s > s) break;
                i2 = i + j; if (i2 >= len) i2 -= len;
                if (a[i2].s > s) break;
            }
            const int nxt = a[cur].next;
            if (j <= k2) {
                if (prev < 0) head = nxt; else a[prev].next = nxt;
                a[cur].s = 0;
            } else prev = cur;
            cur = nxt;
        }
    }
    if (head < 0) { free(a); return -1; }
    /* pass 3: points with a 1-point support that do not dominate both neighbours */
    {
        int prev = -1;
        for (int cur = head; cur >= 0;) {
            const int nxt = a[cur].next;
            int drop = 0;
            if (a[cur].k == 1) {
                const int s = a[cur].s, i = cur;
                int i1 = i - 1; if (i1 < 0) i1 += len;
                int i2 = i + 1; if (i2 >= len) i2 -= len;
                if (s <= a[i1].s || s <= a[i2].s) drop = 1;
            }
            if (drop) {
                if (prev < 0) head = nxt; else a[prev].next = nxt;
                a[cur].s = 0;
            } else prev = cur;
            cur = nxt;
        }
    }
    if (head < 0) { free(a); return -1; }
    int n = 0;
    for (int cur = head; cur >= 0; cur = a[cur].next) out[n++] = a[cur].pt;
    free(a);
    return n;
}

/* ---- shape descriptors ------------------------------------------------------------------------------------------------ */
double ref_arc_length_closed(const int* xy, int n) {
    if (n <= 1) return 0.0;
    double perimeter = 0.0;
    float px = (float)xy[2 * (n - 1)], py = (float)xy[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        const float x = (float)xy[2 * i], y = (float)xy[2 * i + 1];
        const float dx = x - px, dy = y - py;
        perimeter += sqrtf(dx * dx + dy * dy);
        px = x; py = y;
    }
    return perimeter;
}

double ref_contour_area(const int* xy, int n) {
    if (n == 0) return 0.0;
    double a00 = 0.0;
    float px = (float)xy[2 * (n - 1)], py = (float)xy[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        const float x = (float)xy[2 * i], y = (float)xy[2 * i + 1];
        a00 += (double)px * y - (double)py * x;
        px = x; py = y;
    }
    return fabs(a00 * 0.5);
}

/* approxPolyDP_<int>, closed curve.  dst must hold n points; returns the number written */
int ref_approx_poly_dp_closed(const int* xy, int count, double eps, int* dst_xy) {
    if (count == 0) return 0;
    const pt_t* src = (const pt_t*)xy;
    pt_t* dst = (pt_t*)dst_xy;
    typedef struct { int start, end; } range_t;
    size_t stacksz = (size_t)count + 8, top = 0;
    range_t* stack = (range_t*)malloc(sizeof(range_t) * stacksz);
    range_t slice = {0, 0}, right_slice = {0, 0};
    pt_t start_pt = {-1000000, -1000000}, end_pt = {0, 0}, pt = {0, 0};
    int pos = 0, new_count = 0, le_eps = 0;
#define READ_PT(p, pos_) do { (p) = src[pos_]; if (++(pos_) >= count) (pos_) = 0; } while (0)
#define PUSH(sl) do { if (top >= stacksz) { stacksz = stacksz * 3 / 2; stack = (range_t*)realloc(stack, sizeof(range_t) * stacksz); } \
                      stack[top++] = (sl); } while (0)
    eps *= eps;
    /* 1. approximately the two farthest points of the contour */
    right_slice.start = 0;
    for (int i = 0; i < 3; ++i) {
        double max_dist = 0;
        pos = (pos + right_slice.start) % count;
        READ_PT(start_pt, pos);
        for (int j = 1; j < count; ++j) {
            READ_PT(pt, pos);
            const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_slice.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    /* 2. the stack */
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        PUSH(right_slice);
        PUSH(slice);
    } else dst[new_count++] = start_pt;
    /* 3. the recursion */
    while (top > 0) {
        slice = stack[--top];
        end_pt = src[slice.end];
        pos = slice.start;
        READ_PT(start_pt, pos);
        if (pos != slice.end) {
            double max_dist = 0;
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                READ_PT(pt, pos);
                const double dist = fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right_slice.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = 1;
            start_pt = src[slice.start];
        }
        if (le_eps) dst[new_count++] = start_pt;
        else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            PUSH(right_slice);
            PUSH(slice);
        }
    }
#undef READ_PT
#undef PUSH
    free(stack);
    /* last stage: drop points on [almost] straight lines, in place */
    {
        const int cnt = new_count;
        int rpos = cnt - 1, wpos, i;
#define READ_DST(p) do { (p) = dst[rpos]; if (++rpos >= cnt) rpos = 0; } while (0)
        READ_DST(start_pt);
        wpos = rpos;
        READ_DST(pt);
        for (i = 0; i < cnt && new_count > 2; ++i) {
            READ_DST(end_pt);
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            const double dist = fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
            const double inner = (double)((pt.x - start_pt.x) * (end_pt.x - pt.x) + (pt.y - start_pt.y) * (end_pt.y - pt.y));
            if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
                new_count--;
                dst[wpos] = start_pt = end_pt;
                if (++wpos >= cnt) wpos = 0;
                READ_DST(pt);
                i++;
                continue;
            }
            dst[wpos] = start_pt = pt;
            if (++wpos >= cnt) wpos = 0;
            pt = end_pt;
        }
#undef READ_DST
    }
    return new_count;
}

/* ---- exported drivers ------------------------------------------------------------------------------------------------- */
/* method: 0 = CHAIN_APPROX_NONE, 1 = CHAIN_APPROX_TC89_KCOS.  Contours in cv2.findContours(RETR_CCOMP) order; points of all
 * contours back to back in xy (capacity cap_pts points), per contour its point count and hole flag (capacity cap_contours).
 * Returns the number of contours, -1 when a capacity is too small, -2 when the chain approximation failed. */
int ref_find_contours(const uint8_t* mask, int h, int w, int method, int* xy, int cap_pts, int* counts, int* holes, int cap_contours) {
    borders_t bs = {0, 0, 0};
    scan_borders(mask, h, w, &bs);
    int rc = bs.n;
    if (bs.n > cap_contours) rc = -1;
    int* order = (int*)malloc(sizeof(int) * (size_t)(bs.n + 1));
    const int n = ccomp_order(&bs, order);
    if (n != bs.n) rc = -2;
    int used = 0;
    for (int q = 0; q < n && rc >= 0; ++q) {
        const border_t* b = &bs.v[order[q]];
        const int need = b->n_codes ? b->n_codes : 1;
        if (used + need > cap_pts) { rc = -1; break; }
        const int got = method == 1 ? tc89_kcos(b, (pt_t*)(xy + 2 * used)) : chain_points(b, (pt_t*)(xy + 2 * used));
        if (got < 0) { rc = -2; break; }
        counts[q] = got; holes[q] = b->is_hole;
        used += got;
    }
    free(order);
    free_borders(&bs);
    return rc;
}

/* chessvision/core.py:357-379 + 381-411.  quad = 4 x (x, y); returns 1 when a quadrangle was found, 0 when not, < 0 on failure */
int ref_find_quadrangle(const uint8_t* mask, int h, int w, int* quad) {
    borders_t bs = {0, 0, 0};
    scan_borders(mask, h, w, &bs);
    int* order = (int*)malloc(sizeof(int) * (size_t)(bs.n + 1));
    const int n = ccomp_order(&bs, order);
    int rc = 0;
    const double mask_area = (double)h * (double)w;
    for (int q = 0; q < n && rc == 0; ++q) {
        const border_t* b = &bs.v[order[q]];
        const int cap = b->n_codes ? b->n_codes : 1;
        pt_t* pts = (pt_t*)malloc(sizeof(pt_t) * (size_t)cap);
        const int cnt = tc89_kcos(b, pts);
        if (cnt < 0) { free(pts); rc = -2; break; }
        int keep = 1;
        if (n > 1) {                                                           /* core.py:362-366 */
            const double area = ref_contour_area((const int*)pts, cnt) / mask_area;
            if (area < 0.35 || area > 1.0) keep = 0;
            if (keep) {
                int x0 = pts[0].x, x1 = pts[0].x, y0 = pts[0].y, y1 = pts[0].y;
                for (int k = 1; k < cnt; ++k) {
                    if (pts[k].x < x0) x0 = pts[k].x;
                    if (pts[k].x > x1) x1 = pts[k].x;
                    if (pts[k].y < y0) y0 = pts[k].y;
                    if (pts[k].y > y1) y1 = pts[k].y;
                }
                const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;                  /* boundingRect */
                const double r = (bw == 0 || bh == 0) ? -1.0 : (double)(bw < bh ? bw : bh) / (double)(bw < bh ? bh : bw);
                if (r < 0.6) keep = 0;
            }
        }
        if (keep) {
            const double arclen = ref_arc_length_closed((const int*)pts, cnt);
            pt_t* ap = (pt_t*)malloc(sizeof(pt_t) * (size_t)cnt);
            const int m = ref_approx_poly_dp_closed((const int*)pts, cnt, 0.1 * arclen, (int*)ap);
            if (m == 4) {
                int o[4] = {0, 1, 2, 3};
                if (ap[0].x < ap[2].x) { o[0] = 3; o[1] = 0; o[2] = 1; o[3] = 2; }   /* core.py:406-411 */
                for (int k = 0; k < 4; ++k) { quad[2 * k] = ap[o[k]].x; quad[2 * k + 1] = ap[o[k]].y; }
                rc = 1;
            }
            free(ap);
        }
        free(pts);
    }
    free(order);
    free_borders(&bs);
    return rc;
}
