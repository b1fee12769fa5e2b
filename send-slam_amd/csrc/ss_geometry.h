/* ss_geometry.h -- host-side geometry / layout builder (see ss_geometry.cpp). */
#ifndef SS_GEOMETRY_H
#define SS_GEOMETRY_H

#include <string>
#include <vector>

#include "../../include/sendslam_orb.h"
#include "ss_layout.h"

struct ss_host_tables {
    std::vector<ss_rtab> rtab;   /* resize coefficient tables, all levels */
    std::vector<uint32_t> tiles2; /* 64x32 tiles: level << 20 | tile_y << 8 | tile_x, all levels */
    std::vector<uint16_t> cinfo; /* cell-window info per column / row, all levels */
    std::vector<uint32_t> tile_recs;  /* SS_TILE_REC_WORDS words per 64x32 tile, tiles2 order */
    std::vector<uint32_t> tilecell;   /* per 64x32 tile: first cell column | first cell row << 16 */
    std::vector<uint32_t> cell_units; /* per cell: SS_CELL_UNITS x (tile2 index | sub-list << 24), ~0 = end */
};

int ss_build_geometry(const ss_orb_params &p, int width, int height, ss_geom *g,
                      ss_host_tables *tabs, std::string *err);

#endif
