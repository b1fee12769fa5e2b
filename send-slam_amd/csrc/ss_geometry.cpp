/*
 * ss_geometry.cpp -- host-side geometry of an extraction context: pyramid level sizes,
 * scale tables, per-level quotas, umax, the cell grid, quadtree roots, resize coefficient
 * tables and the HBM layout (ss_layout.h).
 *
 * Mirrors what ORB_SLAM3::ORBextractor's constructor and ComputePyramid /
 * ComputeKeyPointsOctTree derive from (nfeatures, scaleFactor, nlevels) and the image size;
 * parameters come from the reference's YAML literals
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:193-206).
 * Host float steps use the same types and order as upstream (DESIGN.md "float steps").
 */
#include "ss_geometry.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "ss_constants.h"

namespace {

inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(float v)
{
    int i = (int)v;
    return i - (i > v);
}
inline int cv_ceil(float v)
{
    int i = (int)v;
    return i + (i < v);
}
inline int16_t sat_short(float v)
{
    int i = cv_round(v);
    return (int16_t)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

/* cv::resize INTER_LINEAR coefficient tables (imgproc/resize.cpp, fixed-point 8U path) */
void build_axis_table(int dn, int sn, bool is_x, std::vector<ss_rtab> &out, int pad_to)
{
    const double inv_scale = (double)dn / sn;
    const double scale = 1. / inv_scale;
    const size_t base = out.size();
    for (int d = 0; d < dn; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        ss_rtab e;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; } /* also covers dx >= xmax: S[sx]*ONE */
            e.s0 = (uint16_t)s;
            e.s1 = (uint16_t)(s + 1 < sn ? s + 1 : sn - 1);
        } else {
            /* rows: weights keep their value, indices are clipped */
            int s0 = s, s1 = s + 1;
            s0 = s0 < 0 ? 0 : (s0 < sn ? s0 : sn - 1);
            s1 = s1 < 0 ? 0 : (s1 < sn ? s1 : sn - 1);
            e.s0 = (uint16_t)s0;
            e.s1 = (uint16_t)s1;
        }
        e.a0 = sat_short((1.f - f) * SS_RESIZE_COEF_SCALE);
        e.a1 = sat_short(f * SS_RESIZE_COEF_SCALE);
        out.push_back(e);
    }
    while ((int)(out.size() - base) < pad_to) out.push_back(out.back());
}

} // namespace

int ss_build_geometry(const ss_orb_params &p, int width, int height, ss_geom *g,
                      ss_host_tables *tabs, std::string *err)
{
    if (p.n_levels < 1 || p.n_levels > SS_MAX_LEVELS_ || p.n_features < 1 ||
        !(p.scale_factor > 1.0f) || p.ini_th_fast < 0 || p.min_th_fast < 0 ||
        p.ini_th_fast > 255 || p.min_th_fast > p.ini_th_fast) {
        *err = "invalid ORB parameters";
        return SS_ERR_INVALID_ARG;
    }
    if (width < 1 || height < 1 || width > 4095 || height > 4095) {
        *err = "image size out of range (1..4095)";
        return SS_ERR_INVALID_ARG;
    }
    memset(g, 0, sizeof(*g));
    g->n_levels = p.n_levels;
    g->w = width;
    g->h = height;
    g->ini_th = p.ini_th_fast;
    g->min_th = p.min_th_fast;
    g->lap_x0 = p.lapping_x0;
    g->lap_x1 = p.lapping_x1;
    g->n_features = p.n_features;

    /* ORBextractor ctor: mvScaleFactor (float, multiplied by the double member), quotas */
    float scale[SS_MAX_LEVELS_], inv_scale[SS_MAX_LEVELS_];
    const double scale_factor = (double)p.scale_factor;
    scale[0] = 1.0f;
    for (int i = 1; i < p.n_levels; i++) scale[i] = (float)(scale[i - 1] * scale_factor);
    for (int i = 0; i < p.n_levels; i++) inv_scale[i] = 1.0f / scale[i];
    const float factor = (float)(1.0f / scale_factor);
    float n_desired = p.n_features * (1 - factor) /
                      (1 - (float)pow((double)factor, (double)p.n_levels));
    int sum_features = 0;
    for (int l = 0; l < p.n_levels - 1; l++) {
        g->lv[l].quota = cv_round(n_desired);
        sum_features += g->lv[l].quota;
        n_desired *= factor;
    }
    g->lv[p.n_levels - 1].quota = p.n_features - sum_features > 0 ? p.n_features - sum_features : 0;

    /* umax */
    {
        int v, v0;
        const int vmax = cv_floor(SS_HALF_PATCH * sqrtf(2.f) / 2 + 1);
        const int vmin = cv_ceil(SS_HALF_PATCH * sqrtf(2.f) / 2);
        const double hp2 = SS_HALF_PATCH * SS_HALF_PATCH;
        for (v = 0; v <= vmax; ++v) g->umax[v] = cv_round(sqrt(hp2 - v * v));
        for (v = SS_HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (g->umax[v0] == g->umax[v0 + 1]) ++v0;
            g->umax[v] = v0;
            ++v0;
        }
        /* the per-lane form of the same table: left half = u in [-d, -1], right half = u in [0, d] */
        for (int lane = 0; lane < 64; lane++) {
            const int row = lane & 31, half = lane >> 5;
            int count = 0;
            g->ic_u0[lane] = 0;
            for (int k = 0; k < 4; k++) {
                static const int8_t pattern[1024] = {SS_BIT_PATTERN_31_VALUES};
                uint32_t word = 0;
                for (int b = 0; b < 4; b++) word |= (uint32_t)(uint8_t)pattern[4 * (lane + 64 * k) + b] << (8 * b);
                g->pat4[lane][k] = word;
            }
            if (row <= 2 * SS_HALF_PATCH) {
                const int d = g->umax[row < SS_HALF_PATCH ? SS_HALF_PATCH - row : row - SS_HALF_PATCH];
                count = half ? d + 1 : d;
                g->ic_u0[lane] = half ? 0 : -d;
            }
            for (int k = 0; k < 4; k++) {
                uint32_t m = 0;
                for (int b = 0; b < 4; b++)
                    if (4 * k + b < count) m |= 0xFFu << (8 * b);
                g->ic_mask[lane][k] = m;
            }
        }
    }

    tabs->rtab.clear();
    tabs->tiles2.clear();
    tabs->cinfo.clear();
    tabs->tilecell.clear();
    tabs->tile_recs.clear();
    tabs->cell_units.clear();
    uint32_t off = 0;
    int bucket_base = 0, chunk_base = 0;
    int cell_base = 0, cand_base = 0, sel_base = 0, node_base = 0, item_base = 0, tile2_base = 0;
    for (int l = 0; l < p.n_levels; l++) {
        ss_level &L = g->lv[l];
        L.w = cv_round((float)width * inv_scale[l]);
        L.h = cv_round((float)height * inv_scale[l]);
        L.scale = scale[l];
        L.scaled_patch = (int)(SS_PATCH_SIZE * scale[l]);
        L.pitch = align_up(L.w, 64);
        L.off = off;
        off += (uint32_t)align_up(L.pitch * L.h, 256);

        const int min_b = SS_MIN_BORDER;
        const int max_bx = L.w - SS_EDGE_THRESHOLD + 3, max_by = L.h - SS_EDGE_THRESHOLD + 3;
        const float fw = (float)(max_bx - min_b), fh = (float)(max_by - min_b);
        if (fw < SS_CELL_W || fh < SS_CELL_W) {
            *err = "image too small: level " + std::to_string(l) + " is " + std::to_string(L.w) +
                   "x" + std::to_string(L.h) + ", needs a 35-px cell inside 16-px borders";
            return SS_ERR_TOO_SMALL;
        }
        const float W = SS_CELL_W;
        L.n_cols = (int)(fw / W);
        L.n_rows = (int)(fh / W);
        L.w_cell = (int)ceilf(fw / L.n_cols);
        L.h_cell = (int)ceilf(fh / L.n_rows);
        L.cell_base = cell_base;
        cell_base += L.n_cols * L.n_rows;

        /* per-column / per-row window info: cell c evaluates [ini+3, max-3) where
         * max = min(ini + cell + 6, maxBorder); cells starting at or past maxBorder - 6 (columns)
         * / - 3 (rows) are skipped (ComputeKeyPointsOctTree's two `continue`s) */
        for (int axis = 0; axis < 2; axis++) {
            const int n = axis == 0 ? L.w : L.h;
            const int n_cells = axis == 0 ? L.n_cols : L.n_rows;
            const int cell = axis == 0 ? L.w_cell : L.h_cell;
            const int max_b = axis == 0 ? max_bx : max_by;
            const int skip_from = axis == 0 ? max_bx - 6 : max_by - 3;
            const size_t base = tabs->cinfo.size();
            (axis == 0 ? L.xinfo_off : L.yinfo_off) = (int)base;
            tabs->cinfo.resize(base + (size_t)align_up(n, 4), 0);
            for (int k = 0; k < n_cells; k++) {
                const int ini = min_b + k * cell;
                if (ini >= skip_from) continue;
                int mx = ini + cell + 6;
                if (mx > max_b) mx = max_b;
                const int lo = ini + 3, hi = mx - 3; /* [lo, hi) */
                for (int p = lo; p < hi; p++)
                    tabs->cinfo[base + p] = (uint16_t)(SS_CI_VALID | (p == lo ? SS_CI_LOW : 0) |
                                                       (p == hi - 1 ? SS_CI_HIGH : 0) | (uint16_t)k);
            }
        }

        /* quadtree roots: nIni = round(width / height), hX = width / nIni */
        L.n_ini = (int)roundf((float)(max_bx - min_b) / (float)(max_by - min_b));
        if (L.n_ini < 1) {
            *err = "unsupported aspect ratio: level narrower than half its height";
            return SS_ERR_TOO_SMALL;
        }
        L.hx = (float)(max_bx - min_b) / L.n_ini;

        /* capacities: NMS survivors are never 8-adjacent inside one cell */
        const int bw = max_bx - min_b, bh = max_by - min_b;
        L.cand_base = cand_base;
        L.cand_cap = align_up(((bw + L.n_cols) * (bh + L.n_rows)) / 4 + L.n_cols * L.n_rows + 64, 64);
        cand_base += L.cand_cap;
        /* a cell evaluates at most w_cell x h_cell pixels and its survivors are never 8-adjacent */
        L.bucket_cap = ((L.w_cell + 1) / 2) * ((L.h_cell + 1) / 2);
        L.bucket_base = bucket_base;
        bucket_base += L.bucket_cap * L.n_cols * L.n_rows;
        L.chunk_base = chunk_base;
        chunk_base += (L.n_cols * L.n_rows + 63) / 64;
        const int most = (L.quota + 3 > 4 * L.n_ini ? L.quota + 3 : 4 * L.n_ini);
        L.sel_base = sel_base;
        L.sel_cap = align_up(most + 5, 8);
        sel_base += L.sel_cap;
        L.node_base = node_base;
        L.node_cap = 24 * (most + 4) + 64;
        node_base += L.node_cap;
        L.item_base = item_base;
        L.item_cap = align_up(most + 16, 8);
        item_base += L.item_cap;

        L.tiles_x = (L.w + SS_TILE_W - 1) / SS_TILE_W;

        L.tile2_base = tile2_base;
        L.tiles2_y = (L.h + SS_TILE_H2 - 1) / SS_TILE_H2;
        tile2_base += L.tiles_x * L.tiles2_y;
        for (int ty = 0; ty < L.tiles2_y; ty++)
            for (int tx = 0; tx < L.tiles_x; tx++)
                tabs->tiles2.push_back(((uint32_t)l << 20) | ((uint32_t)ty << 8) | (uint32_t)tx);

        /* which cell windows a 64x32 tile meets, and over which (tile, sub-list) pairs a cell is spread */
        {
            const uint16_t *xin = tabs->cinfo.data() + L.xinfo_off, *yin = tabs->cinfo.data() + L.yinfo_off;
            auto first_cell = [](const uint16_t *info, int from, int to) {
                for (int p = from; p < to; p++)
                    if (info[p] & SS_CI_VALID) return (int)(info[p] & SS_CI_CELL);
                return 0;
            };
            for (int ty = 0; ty < L.tiles2_y; ty++)
                for (int tx = 0; tx < L.tiles_x; tx++) {
                    const int col0 = first_cell(xin, tx * SS_TILE_W, std::min((tx + 1) * SS_TILE_W, L.w));
                    const int row0 = first_cell(yin, ty * SS_TILE_H2, std::min((ty + 1) * SS_TILE_H2, L.h));
                    tabs->tilecell.push_back((uint32_t)col0 | ((uint32_t)row0 << 16));
                    const uint32_t rec[SS_TILE_REC_WORDS] = {(uint32_t)l, (uint32_t)(tx * SS_TILE_W), (uint32_t)(ty * SS_TILE_H2),
                                                             (uint32_t)L.w, (uint32_t)L.h, (uint32_t)L.pitch, L.off,
                                                             (uint32_t)L.xinfo_off, (uint32_t)L.yinfo_off,
                                                             (uint32_t)col0 | ((uint32_t)row0 << 16)};
                    tabs->tile_recs.insert(tabs->tile_recs.end(), rec, rec + SS_TILE_REC_WORDS);
                }
            const size_t ubase = tabs->cell_units.size();
            tabs->cell_units.resize(ubase + (size_t)L.n_cols * L.n_rows * SS_CELL_UNITS, 0xFFFFFFFFu);
            std::vector<int> x_lo(L.n_cols, 1 << 30), x_hi(L.n_cols, -1), y_lo(L.n_rows, 1 << 30), y_hi(L.n_rows, -1);
            for (int p = 0; p < L.w; p++)
                if (xin[p] & SS_CI_VALID) {
                    const int c = xin[p] & SS_CI_CELL;
                    x_lo[c] = std::min(x_lo[c], p);
                    x_hi[c] = std::max(x_hi[c], p);
                }
            for (int p = 0; p < L.h; p++)
                if (yin[p] & SS_CI_VALID) {
                    const int c = yin[p] & SS_CI_CELL;
                    y_lo[c] = std::min(y_lo[c], p);
                    y_hi[c] = std::max(y_hi[c], p);
                }
            for (int ci = 0; ci < L.n_rows; ci++)
                for (int cj = 0; cj < L.n_cols; cj++) {
                    if (x_hi[cj] < 0 || y_hi[ci] < 0) continue; /* a cell the grid loop skips */
                    int n_units = 0;
                    for (int ty = y_lo[ci] / SS_TILE_H2; ty <= y_hi[ci] / SS_TILE_H2; ty++)
                        for (int tx = x_lo[cj] / SS_TILE_W; tx <= x_hi[cj] / SS_TILE_W; tx++) {
                            const int tile = L.tile2_base + ty * L.tiles_x + tx;
                            const uint32_t tc = tabs->tilecell[tile];
                            const int kc = cj - (int)(tc & 0xFFFFu), kr = ci - (int)(tc >> 16);
                            if (kc < 0 || kc >= 3 || kr < 0 || kr >= SS_TS_ROWS || n_units >= SS_CELL_UNITS) {
                                *err = "unsupported cell / tile geometry at level " + std::to_string(l);
                                return SS_ERR_INVALID_ARG;
                            }
                            tabs->cell_units[ubase + ((size_t)ci * L.n_cols + cj) * SS_CELL_UNITS + n_units++] =
                                (uint32_t)tile | ((uint32_t)(kr * 3 + kc) << 24);
                        }
                }
        }

        if (l > 0) {
            const ss_level &P = g->lv[l - 1];
            L.xtab_off = (int)tabs->rtab.size();
            build_axis_table(L.w, P.w, true, tabs->rtab, align_up(L.w, 256));
            L.ytab_off = (int)tabs->rtab.size();
            build_axis_table(L.h, P.h, false, tabs->rtab, L.h);
        }
    }
    /* the resize kernels read a whole 64-column tile's taps whatever the level's width: the last table must not end
     * the allocation short of that */
    for (int i = 0; i < 64 && !tabs->rtab.empty(); i++) tabs->rtab.push_back(tabs->rtab.back());
    g->block_bytes = off;
    g->n_cells = cell_base;
    g->cand_total = cand_base;
    g->bucket_total = bucket_base;
    g->chunks_total = chunk_base;
    g->sel_total = sel_base;
    g->node_total = node_base;
    g->item_total = item_base;
    g->tiles2_total = tile2_base;
    g->kcap = align_up(sel_base, 64);
    return SS_OK;
}
