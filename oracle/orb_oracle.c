/*
 * orb_oracle.c -- CPU restatement of the ORB extract + match hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see orb_oracle.h).  PARITY UNPINNED (see
 * orb_constants.h).  Build with -ffp-contract=off: every float step below is one IEEE
 * single-precision operation, in the order written.
 *
 * The reference tree holds no line of this arithmetic; the call site that enters it is
 * /root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:594, the parameters
 * come from :193-206.  Each function names the upstream routine it restates.
 */
#include "orb_oracle.h"
#include "orb_constants.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- OpenCV rounding helpers ------------------------------------------------------- */
static int cv_round_f(float v) { return (int)lrintf(v); } /* half to even (SSE cvtss2si) */
static int cv_round_d(double v) { return (int)lrint(v); }
static int cv_floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}
static int cv_ceil_f(float v)
{
    int i = (int)v;
    return i + (i < v);
}
static short sat_short_from_float(float v)
{
    int i = cv_round_f(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

void orc_default_params(orc_params *p)
{
    p->n_features = ORC_DEFAULT_NFEATURES;
    p->scale_factor = ORC_DEFAULT_SCALE;
    p->n_levels = ORC_DEFAULT_NLEVELS;
    p->ini_th_fast = ORC_DEFAULT_INI_TH;
    p->min_th_fast = ORC_DEFAULT_MIN_TH;
    p->lapping_x0 = ORC_DEFAULT_LAPPING_X0;
    p->lapping_x1 = ORC_DEFAULT_LAPPING_X1;
    p->steer_fma = 0;
}

/* ORBextractor::ORBextractor: scale tables, per-level quotas, umax */
int orc_geometry_init(orc_geometry *g, const orc_params *p, int width, int height)
{
    if (p->n_levels < 1 || p->n_levels > ORC_MAX_LEVELS) return -1;
    memset(g, 0, sizeof(*g));
    g->n_levels = p->n_levels;
    const double scale_factor = (double)p->scale_factor; /* member is double, ctor arg float */
    g->scale[0] = 1.0f;
    for (int i = 1; i < p->n_levels; i++) g->scale[i] = (float)(g->scale[i - 1] * scale_factor);
    for (int i = 0; i < p->n_levels; i++) g->inv_scale[i] = 1.0f / g->scale[i];

    const float factor = (float)(1.0f / scale_factor);
    float n_desired = p->n_features * (1 - factor) /
                      (1 - (float)pow((double)factor, (double)p->n_levels));
    int sum = 0;
    for (int l = 0; l < p->n_levels - 1; l++) {
        g->quota[l] = cv_round_f(n_desired);
        sum += g->quota[l];
        n_desired *= factor;
    }
    g->quota[p->n_levels - 1] = p->n_features - sum > 0 ? p->n_features - sum : 0;

    /* umax */
    int v, v0;
    const int vmax = cv_floor_f(ORC_HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    const int vmin = cv_ceil_f(ORC_HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = ORC_HALF_PATCH_SIZE * ORC_HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) g->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = ORC_HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (g->umax[v0] == g->umax[v0 + 1]) ++v0;
        g->umax[v] = v0;
        ++v0;
    }

    /* ComputePyramid: Size(cvRound((float)cols*scale), cvRound((float)rows*scale)) */
    for (int l = 0; l < p->n_levels; l++) {
        g->w[l] = cv_round_f((float)width * g->inv_scale[l]);
        g->h[l] = cv_round_f((float)height * g->inv_scale[l]);
        /* the cell grid needs >= one 35-px cell between the 16-px borders, and the
         * quadtree needs round(width/height) >= 1 root */
        const int bw = g->w[l] - 2 * (ORC_EDGE_THRESHOLD - 3);
        const int bh = g->h[l] - 2 * (ORC_EDGE_THRESHOLD - 3);
        if (bw < ORC_CELL_W || bh < ORC_CELL_W) return -2;
        if ((int)roundf((float)bw / (float)bh) < 1) return -3;
    }
    return 0;
}

/* ---- K0: cv::cvtColor(..., RGB2GRAY / BGR2GRAY), 8U, OpenCV 4.5 15-bit fixed point -- */
void orc_gray(const uint8_t *src, int w, int h, int channels, int stride, int rgb,
              uint8_t *dst)
{
    /* coefficient for byte 0 and byte 2 swap with the declared channel order */
    const int c0 = rgb ? ORC_GRAY_RY : ORC_GRAY_BY;
    const int c1 = ORC_GRAY_GY;
    const int c2 = rgb ? ORC_GRAY_BY : ORC_GRAY_RY;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * stride;
        for (int x = 0; x < w; x++, s += channels)
            dst[(size_t)y * w + x] =
                (uint8_t)((s[0] * c0 + s[1] * c1 + s[2] * c2 + (1 << (ORC_GRAY_SHIFT - 1))) >>
                          ORC_GRAY_SHIFT);
    }
}

/* ---- K1: cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), CV_8UC1 generic path -------
 * HResizeLinear<uchar,int,short,2048> then
 * VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,VResizeLinearVec_32s8u> */
void orc_resize_linear(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *row0 = (int *)malloc(sizeof(int) * dw);
    int *row1 = (int *)malloc(sizeof(int) * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_from_float((1.f - fx) * ORC_RESIZE_COEF_SCALE);
        ialpha[2 * dx + 1] = sat_short_from_float(fx * ORC_RESIZE_COEF_SCALE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        const short b0 = sat_short_from_float((1.f - fy) * ORC_RESIZE_COEF_SCALE);
        const short b1 = sat_short_from_float(fy * ORC_RESIZE_COEF_SCALE);
        int sy0 = sy, sy1 = sy + 1; /* clip(sy - ksize2 + 1 + k, 0, ssize.height) */
        sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
        const uint8_t *s0 = src + (size_t)sy0 * sw, *s1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = s0[sx] * ialpha[2 * dx] + s0[sx + 1] * ialpha[2 * dx + 1];
                row1[dx] = s1[sx] * ialpha[2 * dx] + s1[sx + 1] * ialpha[2 * dx + 1];
            } else {
                row0[dx] = s0[sx] * ORC_RESIZE_COEF_SCALE;
                row1[dx] = s1[sx] * ORC_RESIZE_COEF_SCALE;
            }
        }
        uint8_t *d = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; dx++)
            d[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(ialpha);
    free(row0);
    free(row1);
}

/* ORBextractor::ComputePyramid.  The 19-px BORDER_REFLECT_101 frame upstream adds around
 * every level is never read by detection, orientation or sampling (keypoints stay >= 19 px
 * inside, max rotated tap radius is 18); it only feeds GaussianBlur at the rim, where it
 * equals BORDER_REFLECT_101 of the interior -- orc_blur reflects instead of storing it. */
void orc_pyramid(const uint8_t *src, int stride, const orc_geometry *g, uint8_t **levels)
{
    for (int y = 0; y < g->h[0]; y++)
        memcpy(levels[0] + (size_t)y * g->w[0], src + (size_t)y * stride, (size_t)g->w[0]);
    for (int l = 1; l < g->n_levels; l++)
        orc_resize_linear(levels[l - 1], g->w[l - 1], g->h[l - 1], levels[l], g->w[l], g->h[l]);
}

/* ---- K2: cv::FAST_t<16> ------------------------------------------------------------- */
static const int k_ring[16][2] = ORC_FAST_RING;

/* cv::cornerScore<16>: largest threshold for which the pixel stays a corner */
static int fast_corner_score(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
        int m = a < d[k] ? a : d[k];
        a0 = a0 > m ? a0 : m;
        m = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > m ? a0 : m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
        int m = b > d[k] ? b : d[k];
        b0 = b0 < m ? b0 : m;
        m = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < m ? b0 : m;
    }
    return -b0 - 1;
}

/* corner test of FAST_t<16>: > 8 contiguous ring pixels darker than v-t or brighter than v+t */
static int fast_is_corner(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = 25;
    const int v = ptr[0];
    int count = 0, vt = v - threshold;
    for (int k = 0; k < N; k++) {
        if (ptr[pixel[k]] < vt) {
            if (++count > K) return 1;
        } else
            count = 0;
    }
    count = 0;
    vt = v + threshold;
    for (int k = 0; k < N; k++) {
        if (ptr[pixel[k]] > vt) {
            if (++count > K) return 1;
        } else
            count = 0;
    }
    return 0;
}

static void fast_make_offsets(int pixel[25], int pitch)
{
    for (int k = 0; k < 16; k++) pixel[k] = k_ring[k][0] + k_ring[k][1] * pitch;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

void orc_fast_score_map(const uint8_t *img, int w, int h, int threshold, uint8_t *out)
{
    int pixel[25];
    fast_make_offsets(pixel, w);
    memset(out, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = img + (size_t)y * w + x;
            if (fast_is_corner(p, pixel, threshold))
                out[(size_t)y * w + x] = (uint8_t)fast_corner_score(p, pixel, threshold);
        }
}

int orc_fast_cell(const uint8_t *img, int pitch, int x0, int y0, int x1, int y1,
                  int threshold, orc_point *out, int max_out)
{
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw < 7 || ch < 7) return 0;
    int pixel[25];
    fast_make_offsets(pixel, pitch);
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    uint8_t *score = (uint8_t *)calloc((size_t)cw * ch, 1);
    /* rows 3 .. ch-4 and columns 3 .. cw-4 are evaluated; everything else stays 0, which
     * is what the 3-row ring buffer of FAST_t holds for them */
    for (int i = 3; i < ch - 3; i++)
        for (int j = 3; j < cw - 3; j++) {
            const uint8_t *p = img + (size_t)(y0 + i) * pitch + (x0 + j);
            if (fast_is_corner(p, pixel, threshold))
                score[i * cw + j] = (uint8_t)fast_corner_score(p, pixel, threshold);
        }
    int n = 0;
    for (int i = 3; i < ch - 3; i++)
        for (int j = 3; j < cw - 3; j++) {
            const int s = score[i * cw + j]; /* 0 for non-corners: never > a neighbour */
            const uint8_t *r0 = score + (i - 1) * cw + j, *r1 = score + i * cw + j,
                          *r2 = score + (i + 1) * cw + j;
            if (s > r1[1] && s > r1[-1] && s > r0[-1] && s > r0[0] && s > r0[1] &&
                s > r2[-1] && s > r2[0] && s > r2[1]) {
                if (n >= max_out) { free(score); return -1; }
                out[n].x = j;
                out[n].y = i;
                out[n].response = s;
                n++;
            }
        }
    free(score);
    return n;
}

/* ---- K3: ORBextractor::ComputeKeyPointsOctTree, the cell loop ------------------------ */
int orc_candidates(const uint8_t *img, int w, int h, int ini_th, int min_th,
                   orc_point *out, int max_out)
{
    const int min_bx = ORC_EDGE_THRESHOLD - 3, min_by = min_bx;
    const int max_bx = w - ORC_EDGE_THRESHOLD + 3, max_by = h - ORC_EDGE_THRESHOLD + 3;
    const float width = (float)(max_bx - min_bx), height = (float)(max_by - min_by);
    const float W = ORC_CELL_W;
    const int n_cols = (int)(width / W), n_rows = (int)(height / W);
    const int w_cell = (int)ceilf(width / n_cols), h_cell = (int)ceilf(height / n_rows);
    int n = 0;
    for (int i = 0; i < n_rows; i++) {
        const int ini_y = min_by + i * h_cell;
        int max_y = ini_y + h_cell + 6;
        if (ini_y >= max_by - 3) continue;
        if (max_y > max_by) max_y = max_by;
        for (int j = 0; j < n_cols; j++) {
            const int ini_x = min_bx + j * w_cell;
            int max_x = ini_x + w_cell + 6;
            if (ini_x >= max_bx - 6) continue;
            if (max_x > max_bx) max_x = max_bx;
            int c = orc_fast_cell(img, w, ini_x, ini_y, max_x, max_y, ini_th, out + n, max_out - n);
            if (c < 0) return -1;
            if (c == 0) {
                c = orc_fast_cell(img, w, ini_x, ini_y, max_x, max_y, min_th, out + n, max_out - n);
                if (c < 0) return -1;
            }
            for (int k = 0; k < c; k++) {
                out[n + k].x += j * w_cell;
                out[n + k].y += i * h_cell;
            }
            n += c;
        }
    }
    return n;
}

/* ---- libstdc++ std::sort (bits/stl_algo.h, bits/stl_heap.h; GCC 11) restated ---------
 * for ORB-SLAM3's compareNodes: by size, then by UL.x, NOT a total order, so the
 * permutation of equal elements is whatever introsort leaves -- hence the restatement.
 * tests/test_oracle_units.py pins it against this container's real std::sort. */
static int node_less(const orc_sort_item *a, const orc_sort_item *b)
{
    if (a->size < b->size) return 1;
    if (a->size > b->size) return 0;
    return a->ulx < b->ulx;
}
static void item_swap(orc_sort_item *a, orc_sort_item *b)
{
    orc_sort_item t = *a;
    *a = *b;
    *b = t;
}
static void ss_push_heap(orc_sort_item *first, int hole, int top, orc_sort_item value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && node_less(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void ss_adjust_heap(orc_sort_item *first, int hole, int len, orc_sort_item value)
{
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (node_less(first + second, first + (second - 1))) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    ss_push_heap(first, hole, top, value);
}
int orc_std_sort_heap_calls = 0; /* how often the depth limit fell back to heapsort (tests) */
static void ss_heap_sort(orc_sort_item *first, int n) /* __partial_sort(first, last, last) */
{
    orc_std_sort_heap_calls++;
    if (n >= 2) { /* __make_heap */
        int parent = (n - 2) / 2;
        for (;;) {
            orc_sort_item v = first[parent];
            ss_adjust_heap(first, parent, n, v);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = n; /* __sort_heap */
    while (last > 1) {
        --last;
        orc_sort_item v = first[last]; /* __pop_heap(first, last, last) */
        first[last] = first[0];
        ss_adjust_heap(first, 0, last, v);
    }
}
static void ss_unguarded_linear_insert(orc_sort_item *last)
{
    orc_sort_item val = *last;
    orc_sort_item *next = last - 1;
    while (node_less(&val, next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}
static void ss_insertion_sort(orc_sort_item *first, orc_sort_item *last)
{
    if (first == last) return;
    for (orc_sort_item *i = first + 1; i != last; ++i) {
        if (node_less(i, first)) {
            orc_sort_item val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(*first));
            *first = val;
        } else
            ss_unguarded_linear_insert(i);
    }
}
static void ss_introsort_loop(orc_sort_item *first, orc_sort_item *last, int depth_limit)
{
    while (last - first > 16) {
        if (depth_limit == 0) {
            ss_heap_sort(first, (int)(last - first));
            return;
        }
        --depth_limit;
        /* __unguarded_partition_pivot */
        orc_sort_item *mid = first + (last - first) / 2;
        orc_sort_item *a = first + 1, *b = mid, *c = last - 1;
        if (node_less(a, b)) { /* __move_median_to_first(first, a, b, c) */
            if (node_less(b, c)) item_swap(first, b);
            else if (node_less(a, c)) item_swap(first, c);
            else item_swap(first, a);
        } else if (node_less(a, c)) item_swap(first, a);
        else if (node_less(b, c)) item_swap(first, c);
        else item_swap(first, b);
        orc_sort_item *lo = first + 1, *hi = last; /* __unguarded_partition(first+1,last,first) */
        for (;;) {
            while (node_less(lo, first)) ++lo;
            --hi;
            while (node_less(first, hi)) --hi;
            if (!(lo < hi)) break;
            item_swap(lo, hi);
            ++lo;
        }
        ss_introsort_loop(lo, last, depth_limit);
        last = lo;
    }
}
void orc_std_sort(orc_sort_item *a, int n)
{
    if (n <= 0) return;
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) lg++; /* std::__lg */
    ss_introsort_loop(a, a + n, lg * 2);
    if (n > 16) { /* __final_insertion_sort */
        ss_insertion_sort(a, a + 16);
        for (orc_sort_item *i = a + 16; i != a + n; ++i) ss_unguarded_linear_insert(i);
    } else
        ss_insertion_sort(a, a + n);
}

/* ---- K4: ORBextractor::DistributeOctTree with a literal std::list ------------------- */
typedef struct qnode {
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    int *keys; /* indices into cand[], in insertion order (vKeys) */
    int n_keys;
    int no_more;
    struct qnode *prev, *next;
} qnode;

typedef struct {
    qnode *head, *tail;
    int size;
} qlist;

static void ql_push_front(qlist *l, qnode *n)
{
    n->prev = NULL;
    n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n;
    l->size++;
}
static void ql_push_back(qlist *l, qnode *n)
{
    n->next = NULL;
    n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n;
    l->size++;
}
static qnode *ql_erase(qlist *l, qnode *n) /* returns the following node */
{
    qnode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    free(n->keys);
    free(n);
    return nx;
}

/* ExtractorNode::DivideNode; children with no keys are returned with n_keys == 0 */
static void divide_node(const qnode *p, const orc_point *cand, qnode *c[4])
{
    const int half_x = (int)ceilf((float)(p->urx - p->ulx) / 2);
    const int half_y = (int)ceilf((float)(p->bry - p->uly) / 2);
    for (int k = 0; k < 4; k++) {
        c[k] = (qnode *)calloc(1, sizeof(qnode));
        c[k]->keys = (int *)malloc(sizeof(int) * (size_t)(p->n_keys > 0 ? p->n_keys : 1));
    }
    qnode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ulx = p->ulx;            n1->uly = p->uly;
    n1->urx = p->ulx + half_x;   n1->ury = p->uly;
    n1->blx = p->ulx;            n1->bly = p->uly + half_y;
    n1->brx = p->ulx + half_x;   n1->bry = p->uly + half_y;

    n2->ulx = n1->urx; n2->uly = n1->ury;
    n2->urx = p->urx;  n2->ury = p->ury;
    n2->blx = n1->brx; n2->bly = n1->bry;
    n2->brx = p->urx;  n2->bry = p->uly + half_y;

    n3->ulx = n1->blx; n3->uly = n1->bly;
    n3->urx = n1->brx; n3->ury = n1->bry;
    n3->blx = p->blx;  n3->bly = p->bly;
    n3->brx = n1->brx; n3->bry = p->bly;

    n4->ulx = n3->urx; n4->uly = n3->ury;
    n4->urx = n2->brx; n4->ury = n2->bry;
    n4->blx = n3->brx; n4->bly = n3->bry;
    n4->brx = p->brx;  n4->bry = p->bry;

    for (int i = 0; i < p->n_keys; i++) {
        const orc_point *kp = &cand[p->keys[i]];
        qnode *dst;
        if (kp->x < n1->urx) dst = kp->y < n1->bry ? n1 : n3;
        else dst = kp->y < n1->bry ? n2 : n4;
        dst->keys[dst->n_keys++] = p->keys[i];
    }
    for (int k = 0; k < 4; k++)
        if (c[k]->n_keys == 1) c[k]->no_more = 1;
}

typedef struct {
    int size;
    qnode *node;
} size_and_node;

int orc_distribute(const orc_point *cand, int n_cand, int min_x, int max_x, int min_y,
                   int max_y, int N, orc_point *out, int max_out)
{
    const int n_ini = (int)roundf((float)(max_x - min_x) / (float)(max_y - min_y));
    if (n_ini < 1) return -1;
    const float hX = (float)(max_x - min_x) / n_ini;

    qlist L = {0};
    qnode **ini = (qnode **)malloc(sizeof(qnode *) * (size_t)n_ini);
    for (int i = 0; i < n_ini; i++) {
        qnode *ni = (qnode *)calloc(1, sizeof(qnode));
        ni->ulx = (int)(hX * (float)i);       ni->uly = 0;
        ni->urx = (int)(hX * (float)(i + 1)); ni->ury = 0;
        ni->blx = ni->ulx; ni->bly = max_y - min_y;
        ni->brx = ni->urx; ni->bry = max_y - min_y;
        ni->keys = (int *)malloc(sizeof(int) * (size_t)(n_cand > 0 ? n_cand : 1));
        ql_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n_cand; i++) {
        const int r = (int)((float)cand[i].x / hX);
        if (r < 0 || r >= n_ini) { /* upstream would index out of range; cannot happen for
                                      x in [3, width-3) but never write wild */
            free(ini);
            return -2;
        }
        ini[r]->keys[ini[r]->n_keys++] = i;
    }
    free(ini);

    for (qnode *lit = L.head; lit;) {
        if (lit->n_keys == 1) { lit->no_more = 1; lit = lit->next; }
        else if (lit->n_keys == 0) lit = ql_erase(&L, lit);
        else lit = lit->next;
    }

    int finish = 0;
    size_and_node *vsz = (size_and_node *)malloc(sizeof(size_and_node) * (size_t)(4 * (n_cand + 4)));
    size_and_node *vprev = (size_and_node *)malloc(sizeof(size_and_node) * (size_t)(4 * (n_cand + 4)));
    orc_sort_item *items = (orc_sort_item *)malloc(sizeof(orc_sort_item) * (size_t)(4 * (n_cand + 4)));
    int n_vsz = 0;

    while (!finish) {
        int prev_size = L.size;
        int n_to_expand = 0;
        n_vsz = 0;
        qnode *lit = L.head;
        while (lit) {
            if (lit->no_more) { lit = lit->next; continue; }
            qnode *c[4];
            divide_node(lit, cand, c);
            for (int k = 0; k < 4; k++) {
                if (c[k]->n_keys > 0) {
                    ql_push_front(&L, c[k]);
                    if (c[k]->n_keys > 1) {
                        n_to_expand++;
                        vsz[n_vsz].size = c[k]->n_keys;
                        vsz[n_vsz].node = c[k];
                        n_vsz++;
                    }
                } else {
                    free(c[k]->keys);
                    free(c[k]);
                }
            }
            lit = ql_erase(&L, lit);
        }

        if (L.size >= N || L.size == prev_size) {
            finish = 1;
        } else if (L.size + n_to_expand * 3 > N) {
            while (!finish) {
                prev_size = L.size;
                const int n_prev = n_vsz;
                memcpy(vprev, vsz, sizeof(size_and_node) * (size_t)n_prev);
                n_vsz = 0;
                /* sort(vPrev.begin(), vPrev.end(), compareNodes) */
                for (int j = 0; j < n_prev; j++) {
                    items[j].size = vprev[j].size;
                    items[j].ulx = vprev[j].node->ulx;
                    items[j].id = j;
                }
                orc_std_sort(items, n_prev);
                for (int j = n_prev - 1; j >= 0; j--) {
                    qnode *nd = vprev[items[j].id].node;
                    qnode *c[4];
                    divide_node(nd, cand, c);
                    for (int k = 0; k < 4; k++) {
                        if (c[k]->n_keys > 0) {
                            ql_push_front(&L, c[k]);
                            if (c[k]->n_keys > 1) {
                                vsz[n_vsz].size = c[k]->n_keys;
                                vsz[n_vsz].node = c[k];
                                n_vsz++;
                            }
                        } else {
                            free(c[k]->keys);
                            free(c[k]);
                        }
                    }
                    ql_erase(&L, nd);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prev_size) finish = 1;
            }
        }
    }

    /* retain the best point of each node, list order; first maximum wins */
    int n_out = 0;
    for (qnode *lit = L.head; lit; lit = lit->next) {
        int best = lit->keys[0];
        int max_resp = cand[best].response;
        for (int k = 1; k < lit->n_keys; k++)
            if (cand[lit->keys[k]].response > max_resp) {
                best = lit->keys[k];
                max_resp = cand[best].response;
            }
        if (n_out >= max_out) { n_out = -3; break; }
        out[n_out++] = cand[best];
    }
    for (qnode *lit = L.head; lit;) lit = ql_erase(&L, lit);
    free(vsz);
    free(vprev);
    free(items);
    return n_out;
}

/* ---- K5: cv::fastAtan2 (scalar atan_f32) and IC_Angle ------------------------------- */
float orc_fast_atan2(float y, float x)
{
    const float p1 = ORC_ATAN2_P1, p3 = ORC_ATAN2_P3, p5 = ORC_ATAN2_P5, p7 = ORC_ATAN2_P7;
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

float orc_ic_angle(const uint8_t *img, int pitch, int x, int y, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * pitch + x;
    for (int u = -ORC_HALF_PATCH_SIZE; u <= ORC_HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= ORC_HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * pitch], val_minus = center[u - v * pitch];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---- K6a: cv::GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101), 8U fixed-point path ------- */
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}
void orc_blur(const uint8_t *src, int w, int h, uint8_t *dst)
{
    static const int k[7] = ORC_GAUSS_TAPS;
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int i = -3; i <= 3; i++) acc += k[i + 3] * s[reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = (uint16_t)acc; /* ufixedpoint16, 8 fractional bits */
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int j = -3; j <= 3; j++)
                acc += (uint32_t)k[j + 3] * tmp[(size_t)reflect101(y + j, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)((acc + (1u << (ORC_GAUSS_SHIFT - 1))) >> ORC_GAUSS_SHIFT);
        }
    free(tmp);
}

/* ---- K6b: computeOrbDescriptor ------------------------------------------------------ */
static const int k_pattern[1024] = ORC_BIT_PATTERN_31;

static int tap_row(int px, int py, float a, float b, int fma)
{
    return cv_round_f(fma ? fmaf((float)px, b, (float)py * a) : (float)px * b + (float)py * a);
}
static int tap_col(int px, int py, float a, float b, int fma)
{
    return cv_round_f(fma ? fmaf((float)px, a, -((float)py * b)) : (float)px * a - (float)py * b);
}

void orc_descriptor(const uint8_t *img, int pitch, int x, int y, float angle_deg, uint8_t desc[32])
{
    orc_descriptor_ex(img, pitch, x, y, angle_deg, 0, desc);
}

void orc_descriptor_ex(const uint8_t *img, int pitch, int x, int y, float angle_deg, int steer_fma,
                       uint8_t desc[32])
{
    const float factor_pi = (float)(3.14159265358979323846 / 180.f);
    const float angle = angle_deg * factor_pi;
    /* upstream: (float)cos(angle), (float)sin(angle) on a float argument = libm cosf/sinf.
     * The reference image is ubuntu:22.04 (dockerfile:1) = glibc 2.35, the libm of this
     * container, so the literal call is the restatement.  (glibc's result is its FMA ifunc
     * variant on any current x86 host; the device restates that algorithm and
     * tests/test_float_steps.py pins the restatement against this libm over every float
     * in [2^-15, 120).) */
    const float a = cosf(angle), b = sinf(angle);
    const uint8_t *center = img + (size_t)y * pitch + x;
    const int *pat = k_pattern;
    for (int i = 0; i < 32; i++, pat += 32) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
            const int x0 = pat[4 * bit], y0 = pat[4 * bit + 1];
            const int x1 = pat[4 * bit + 2], y1 = pat[4 * bit + 3];
            const int t0 = center[tap_row(x0, y0, a, b, steer_fma) * pitch + tap_col(x0, y0, a, b, steer_fma)];
            const int t1 = center[tap_row(x1, y1, a, b, steer_fma) * pitch + tap_col(x1, y1, a, b, steer_fma)];
            val |= (t0 < t1) << bit;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---- ORBextractor::operator() ------------------------------------------------------- */
int orc_extract(const uint8_t *gray, int w, int h, int stride, const orc_params *p,
                orc_keypoint *kps, uint8_t *desc, int max_kp, int *level_counts)
{
    orc_geometry g;
    if (orc_geometry_init(&g, p, w, h) != 0) return -1;
    uint8_t *levels[ORC_MAX_LEVELS] = {0};
    for (int l = 0; l < g.n_levels; l++) levels[l] = (uint8_t *)malloc((size_t)g.w[l] * g.h[l]);
    orc_pyramid(gray, stride, &g, levels);

    orc_point *lvl_pts[ORC_MAX_LEVELS] = {0};
    float *lvl_ang[ORC_MAX_LEVELS] = {0};
    int lvl_n[ORC_MAX_LEVELS] = {0};
    int total = 0, rc = 0;
    const int min_b = ORC_EDGE_THRESHOLD - 3;

    for (int l = 0; l < g.n_levels && rc == 0; l++) {
        const int lw = g.w[l], lh = g.h[l];
        const int max_cand = (lw * lh) / 2 + 16;
        orc_point *cand = (orc_point *)malloc(sizeof(orc_point) * (size_t)max_cand);
        const int nc = orc_candidates(levels[l], lw, lh, p->ini_th_fast, p->min_th_fast, cand, max_cand);
        if (nc < 0) { rc = -2; free(cand); break; }
        /* DistributeOctTree stops at the first pass that reaches N nodes, and a pass may quadruple the node count
         * (wide level, small quota: 7 root nodes -> 28 leaves for N = 15): up to 4 N + 4 n_ini nodes, never more than
         * there are candidates */
        const int cap = nc + 8;
        lvl_pts[l] = (orc_point *)malloc(sizeof(orc_point) * (size_t)(cap > 0 ? cap : 1));
        int nk = 0;
        if (nc > 0) {
            nk = orc_distribute(cand, nc, min_b, lw - ORC_EDGE_THRESHOLD + 3, min_b,
                                lh - ORC_EDGE_THRESHOLD + 3, g.quota[l], lvl_pts[l], cap);
            if (nk < 0) { rc = -3; free(cand); break; }
        }
        free(cand);
        lvl_ang[l] = (float *)malloc(sizeof(float) * (size_t)(nk > 0 ? nk : 1));
        for (int i = 0; i < nk; i++) {
            lvl_pts[l][i].x += min_b;
            lvl_pts[l][i].y += min_b;
            lvl_ang[l][i] = orc_ic_angle(levels[l], lw, lvl_pts[l][i].x, lvl_pts[l][i].y, g.umax);
        }
        lvl_n[l] = nk;
        total += nk;
        if (level_counts) level_counts[l] = nk;
    }

    if (rc == 0 && total > max_kp) rc = -4;
    if (rc == 0) {
        int mono = 0, stereo = total - 1;
        for (int l = 0; l < g.n_levels; l++) {
            if (lvl_n[l] == 0) continue;
            uint8_t *blurred = (uint8_t *)malloc((size_t)g.w[l] * g.h[l]);
            orc_blur(levels[l], g.w[l], g.h[l], blurred);
            const float scale = g.scale[l];
            const int scaled_patch = (int)(ORC_PATCH_SIZE * scale);
            for (int i = 0; i < lvl_n[l]; i++) {
                orc_keypoint kp;
                kp.x = (float)lvl_pts[l][i].x;
                kp.y = (float)lvl_pts[l][i].y;
                kp.octave = l;
                kp.size = (float)scaled_patch;
                kp.angle = lvl_ang[l][i];
                kp.response = (float)lvl_pts[l][i].response;
                uint8_t d[32];
                orc_descriptor_ex(blurred, g.w[l], lvl_pts[l][i].x, lvl_pts[l][i].y, kp.angle, p->steer_fma, d);
                if (l != 0) { kp.x *= scale; kp.y *= scale; }
                int slot;
                if (kp.x >= (float)p->lapping_x0 && kp.x <= (float)p->lapping_x1) slot = stereo--;
                else slot = mono++;
                kps[slot] = kp;
                memcpy(desc + (size_t)slot * 32, d, 32);
            }
            free(blurred);
        }
    }
    for (int l = 0; l < g.n_levels; l++) {
        free(levels[l]);
        free(lvl_pts[l]);
        free(lvl_ang[l]);
    }
    return rc == 0 ? total : rc;
}

/* ---- K7: ORBmatcher::DescriptorDistance + the build's all-pairs rule (A.6) ----------- */
static int descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    const uint32_t *pa = (const uint32_t *)a, *pb = (const uint32_t *)b;
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void orc_match(const uint8_t *q, int nq, const uint8_t *t, int nt, int th, int ratio_num,
               int ratio_den, int exclude_self, int32_t *idx, uint16_t *d1, uint16_t *d2)
{
    for (int i = 0; i < nq; i++) {
        int best = 0xFFFF, second = 0xFFFF, best_j = -1;
        for (int j = 0; j < nt; j++) {
            if (exclude_self && j == i) continue;
            const int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < best) { second = best; best = d; best_j = j; }
            else if (d < second) second = d;
        }
        d1[i] = (uint16_t)best;
        d2[i] = (uint16_t)second;
        /* accept iff d1 <= TH and d1 < ratio*d2, integer form d1*den < d2*num */
        idx[i] = (best_j >= 0 && (th < 0 || (best <= th && best * ratio_den < second * ratio_num))) ? best_j : -1;
    }
}

/* ---- downstream of the path: pose-only optimisation (ORB-SLAM3 Optimizer::PoseOptimization,
 * g2o EdgeSE3ProjectXYZOnlyPose) restated as damped Gauss-Newton in double precision.  Used by
 * tests to show the north-star clause "downstream PnP pose within 1e-4 rel": fed with the HIP
 * path's keypoints / matches it must return what it returns for the oracle's.  4 rounds x 10
 * iterations, Huber kernel sqrt(5.991) (dropped in the last two rounds), outliers (chi2 > 5.991)
 * re-classified after every round; pose update T <- exp([w, v]) * T. */
static void mat3_mul(const double a[9], const double b[9], double c[9])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static void se3_exp(const double d[6], double R[9], double t[3]) /* d = [omega, upsilon] */
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double A, B, Cc;
    if (th < 1e-8) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; Cc = 1.0 / 6.0 - th2 / 120.0; }
    else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; Cc = (1.0 - A) / th2; }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    mat3_mul(W, W, W2);
    double V[9];
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + A * W[i] + B * W2[i];
        V[i] = I + B * W[i] + Cc * W2[i];
    }
    for (int i = 0; i < 3; i++) t[i] = V[3 * i] * d[3] + V[3 * i + 1] * d[4] + V[3 * i + 2] * d[5];
}
static int chol6_solve(double H[36], double b[6]) /* H x = b, H SPD; x returned in b */
{
    for (int j = 0; j < 6; j++) {
        double s = H[7 * j];
        for (int k = 0; k < j; k++) s -= H[6 * j + k] * H[6 * j + k];
        if (s <= 0) return -1;
        H[7 * j] = sqrt(s);
        for (int i = j + 1; i < 6; i++) {
            double v = H[6 * i + j];
            for (int k = 0; k < j; k++) v -= H[6 * i + k] * H[6 * j + k];
            H[6 * i + j] = v / H[7 * j];
        }
    }
    for (int i = 0; i < 6; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= H[6 * i + k] * b[k];
        b[i] = v / H[7 * i];
    }
    for (int i = 5; i >= 0; i--) {
        double v = b[i];
        for (int k = i + 1; k < 6; k++) v -= H[6 * k + i] * b[k];
        b[i] = v / H[7 * i];
    }
    return 0;
}

/* R (row-major 3x3), t: in = initial Tcw, out = optimised.  obs in pixels, inv_sigma2 per
 * observation.  inlier[i] receives 1/0.  Returns the number of inliers, or < 0. */
int orc_pnp_pose_only(int n, const double *pts3d, const double *obs, const double *inv_sigma2, double fx, double fy,
                      double cx, double cy, double R[9], double t[3], uint8_t *inlier)
{
    const double chi2_th = 5.991, delta = sqrt(5.991);
    {
        /* the rotation next to R (rows by Gram-Schmidt, the third as a cross product): the steps below only multiply by exact
         * exponentials and never remove a deviation from orthonormality the caller's prediction brings (vo_oracle._orthonormal) */
        const double n0 = sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]);
        const double r0[3] = {R[0] / n0, R[1] / n0, R[2] / n0};
        const double d01 = R[3] * r0[0] + R[4] * r0[1] + R[5] * r0[2];
        double r1[3] = {R[3] - d01 * r0[0], R[4] - d01 * r0[1], R[5] - d01 * r0[2]};
        const double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
        for (int k = 0; k < 3; k++) r1[k] /= n1;
        R[0] = r0[0]; R[1] = r0[1]; R[2] = r0[2];
        R[3] = r1[0]; R[4] = r1[1]; R[5] = r1[2];
        R[6] = r0[1] * r1[2] - r0[2] * r1[1]; R[7] = r0[2] * r1[0] - r0[0] * r1[2]; R[8] = r0[0] * r1[1] - r0[1] * r1[0];
    }
    if (n < 3) return -1;
    for (int i = 0; i < n; i++) inlier[i] = 1;
    int n_in = n;
    for (int round = 0; round < 4; round++) {
        const int robust = round < 2;
        double lambda = 1e-6;
        for (int it = 0; it < 10; it++) {
            double H[36] = {0}, b[6] = {0};
            for (int i = 0; i < n; i++) {
                if (!inlier[i]) continue;
                const double *P = pts3d + 3 * i;
                const double x = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
                const double y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
                const double z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
                if (z <= 0) continue;
                const double iz = 1.0 / z, iz2 = iz * iz;
                const double ex = obs[2 * i] - (fx * x * iz + cx), ey = obs[2 * i + 1] - (fy * y * iz + cy);
                const double w0 = inv_sigma2[i];
                const double e2 = w0 * (ex * ex + ey * ey);
                const double w = (robust && e2 > delta * delta) ? w0 * delta / sqrt(e2) : w0;
                const double J0[6] = {x * y * iz2 * fx, -(1 + x * x * iz2) * fx, y * iz * fx, -iz * fx, 0, x * iz2 * fx};
                const double J1[6] = {(1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, 0, -iz * fy, y * iz2 * fy};
                for (int a = 0; a < 6; a++) {
                    b[a] -= w * (J0[a] * ex + J1[a] * ey);
                    for (int c = 0; c < 6; c++) H[6 * a + c] += w * (J0[a] * J0[c] + J1[a] * J1[c]);
                }
            }
            for (int a = 0; a < 6; a++) H[7 * a] += lambda * (1.0 + H[7 * a]);
            if (chol6_solve(H, b) != 0) return -2;
            double dR[9], dt[3], Rn[9];
            se3_exp(b, dR, dt);
            mat3_mul(dR, R, Rn);
            const double tn[3] = {dR[0] * t[0] + dR[1] * t[1] + dR[2] * t[2] + dt[0],
                                  dR[3] * t[0] + dR[4] * t[1] + dR[5] * t[2] + dt[1],
                                  dR[6] * t[0] + dR[7] * t[1] + dR[8] * t[2] + dt[2]};
            memcpy(R, Rn, sizeof(Rn));
            memcpy(t, tn, sizeof(tn));
            /* the round ends once a step has moved nothing (same rule as the product's sst_pose_only) */
            double step = 0;
            for (int a = 0; a < 6; a++) step = fmax(step, fabs(b[a]));
            if (step < 1e-10) break;
        }
        n_in = 0;
        for (int i = 0; i < n; i++) {
            const double *P = pts3d + 3 * i;
            const double x = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
            const double y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
            const double z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
            double chi2 = 1e30;
            if (z > 0) {
                const double ex = obs[2 * i] - (fx * x / z + cx), ey = obs[2 * i + 1] - (fy * y / z + cy);
                chi2 = inv_sigma2[i] * (ex * ex + ey * ey);
            }
            inlier[i] = chi2 <= chi2_th;
            n_in += inlier[i];
        }
        if (n_in < 3) return -3;
    }
    return n_in;
}
