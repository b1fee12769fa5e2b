/* Host-only sweep of ss_build_geometry (send-slam_amd/csrc/ss_geometry.cpp): every image size either
 * builds or is rejected as too small -- never "unsupported cell / tile geometry" -- and the tables the
 * kernels index blindly hold what they assume: a tile meets at most 3 x 2 cell windows, every cell is
 * covered by its (tile, sub-list) units exactly once, bucket capacities bound the window areas. */
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>

#include "../../send-slam_amd/csrc/ss_geometry.h"

int main(int argc, char **argv)
{
    const int step = argc > 1 ? atoi(argv[1]) : 7;
    long built = 0, small = 0;
    const float scales[3] = {1.2f, 1.5f, 2.0f};
    for (int si = 0; si < 3; si++)
        for (int w = 60; w <= 4095; w += (w < 400 ? 1 : step * 9))
            for (int h = 60; h <= 2300; h += (h < 300 ? (w < 400 ? 1 : 3) : step * 11)) {
                ss_orb_params p;
                p.n_features = 1000; p.scale_factor = scales[si]; p.n_levels = si == 0 ? 8 : 4;
                p.ini_th_fast = 20; p.min_th_fast = 7; p.lapping_x0 = 0; p.lapping_x1 = 1000; p.max_batch = 1;
                ss_geom g;
                ss_host_tables t;
                std::string err;
                const int rc = ss_build_geometry(p, w, h, &g, &t, &err);
                if (rc == SS_ERR_TOO_SMALL) { small++; continue; }
                if (rc != SS_OK) { printf("FAIL %dx%d scale %.1f: %d %s\n", w, h, scales[si], rc, err.c_str()); return 1; }
                built++;
                for (int l = 0; l < g.n_levels; l++) {
                    const ss_level &L = g.lv[l];
                    if (L.w_cell < 35 || L.h_cell < 35) { printf("FAIL cell < 35 at %dx%d\n", w, h); return 1; }
                    if (L.bucket_cap < ((L.w_cell + 1) / 2) * ((L.h_cell + 1) / 2)) { printf("FAIL bucket cap\n"); return 1; }
                    /* every valid (x, y) belongs to exactly one unit of its cell, and the unit's sub-list index is right */
                    const uint16_t *xin = t.cinfo.data() + L.xinfo_off, *yin = t.cinfo.data() + L.yinfo_off;
                    for (int y = 0; y < L.h; y++) {
                        if (!(yin[y] & SS_CI_VALID)) continue;
                        for (int x = 0; x < L.w; x++) {
                            if (!(xin[x] & SS_CI_VALID)) continue;
                            const int cj = xin[x] & SS_CI_CELL, ci = yin[y] & SS_CI_CELL;
                            const int tile = L.tile2_base + (y / SS_TILE_H2) * L.tiles_x + x / SS_TILE_W;
                            const uint32_t tc = t.tilecell[tile];
                            const int k = (ci - (int)(tc >> 16)) * 3 + (cj - (int)(tc & 0xFFFF));
                            if (k < 0 || k >= SS_TS_CELLS) { printf("FAIL k=%d at %dx%d l%d (%d,%d)\n", k, w, h, l, x, y); return 1; }
                            const uint32_t want = (uint32_t)tile | ((uint32_t)k << 24);
                            const uint32_t *u = t.cell_units.data() + ((size_t)L.cell_base + (size_t)ci * L.n_cols + cj) * SS_CELL_UNITS;
                            int hits = 0;
                            for (int q = 0; q < SS_CELL_UNITS && u[q] != 0xFFFFFFFFu; q++) hits += u[q] == want;
                            if (hits != 1) { printf("FAIL unit coverage at %dx%d l%d (%d,%d): %d\n", w, h, l, x, y, hits); return 1; }
                            x += 5; /* sample */
                        }
                        y += 3;
                    }
                }
            }
    printf("built=%ld too_small=%ld\n", built, small);
    return built > 1000 ? 0 : 1;
}
