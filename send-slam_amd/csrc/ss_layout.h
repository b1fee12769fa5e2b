/*
 * ss_layout.h -- HBM layout of one extraction context, shared by host code and kernels.
 *
 * All per-frame buffers are [batch slot][...] so one launch covers a whole batch of frames
 * (one camera's frame batch per GPU; BASELINE.json north_star).  Per frame:
 *
 *   pyr / blur / score   three identical "pyramid blocks": level l at byte offset lv[l].off,
 *                        row pitch lv[l].pitch (multiple of 64 B -> every row starts on a
 *                        64-B line, dword/dwordx4 row loads are aligned), no border stored
 *                        (DESIGN.md "no stored border").
 *   tsurv / thdr         NMS survivors as the FAST kernel leaves them: per 64x32 tile, up to
 *                        SS_TS_CELLS sub-lists (one per cell window the tile meets) packed back to
 *                        back, and their count words (survivors low half, those >= iniTh high half)
 *   cell_cnt             one count word per grid cell, all levels (cell_base + row*n_cols + col)
 *   bucket               the survivors of each cell, unordered (bucket_base + cell * bucket_cap),
 *                        gathered from the tile sub-lists
 *   cand                 packed candidates, level l at cand_base, upstream order
 *   sel                  per-level quadtree survivors, level l at sel_base, list order
 *   qt_* scratch         quadtree ping-pong record buffers, node table, sort items
 *   kps / desc           final keypoints and descriptors, kcap rows
 *
 * A candidate / record is one u32: x[0:12) | y[12:24) | response[24:32), x and y relative
 * to the (16,16) border origin for candidates, level coordinates for `sel`.
 */
#ifndef SS_LAYOUT_H
#define SS_LAYOUT_H

#include <stdint.h>

#define SS_MAX_LEVELS_ 16
#define SS_TILE_W 64
#ifndef SS_TILE_H2
#define SS_TILE_H2 32 /* tile height of the FAST / NMS / blur kernel (32 or 64; 8 * SS_TILE_H2 threads per block) */
#endif
/* NMS survivors of one 64x32 tile: a tile meets at most 3 x 2 cell windows (cells are >= 35 px) and
 * survivors of ONE window are never 8-adjacent, so <= (32 + 1) * (16 + 1) of them */
#define SS_TS_ROWS (SS_TILE_H2 > 32 ? 3 : 2) /* cell rows a tile can meet (cells are >= 35 px) */
#define SS_TS_CAP (SS_TILE_H2 > 32 ? 1152 : 576)
#define SS_TS_CELLS (3 * SS_TS_ROWS) /* sub-lists per tile: (cell row - first row) * 3 + (cell col - first col) */
#define SS_TS_HDR (SS_TILE_H2 > 32 ? 12 : 8) /* header words per tile (SS_TS_CELLS count words, padded) */
#define SS_TILE_REC_WORDS 16 /* per-tile record of the FAST kernel: level, x0, y0, w, h, pitch, off, xinfo_off, yinfo_off, first cells */
#define SS_CELL_UNITS 12 /* (tile, sub-list) pairs one cell can be spread over */

#define SS_PACK(x, y, r) ((uint32_t)(x) | ((uint32_t)(y) << 12) | ((uint32_t)(r) << 24))
#define SS_PX(p) ((int)((p) & 0xFFFu))
#define SS_PY(p) ((int)(((p) >> 12) & 0xFFFu))
#define SS_PR(p) ((int)((p) >> 24))

typedef struct {
    int32_t w, h, pitch;
    uint32_t off; /* byte offset inside a pyramid block */
    /* cell grid of ComputeKeyPointsOctTree */
    int32_t n_cols, n_rows, w_cell, h_cell;
    int32_t cell_base;
    int32_t quota;
    int32_t cand_base, cand_cap;
    int32_t bucket_base, bucket_cap; /* NMS survivors per cell: entries per cell, first entry of the level */
    int32_t chunk_base;              /* K3b blocks: 64 cells each */
    int32_t sel_base, sel_cap;
    int32_t node_base, node_cap; /* quadtree nodes */
    int32_t item_base, item_cap; /* quadtree sort items / expandable lists */
    /* 64x32 tiles of the FAST / blur kernel */
    int32_t tiles_x;
    int32_t tile2_base, tiles2_y;
    /* resize tables for building THIS level from level-1 (entries of 8 bytes) */
    int32_t xtab_off, ytab_off;
    /* cell-window tables (u16 per column / row of the level): SS_CI_* bits | cell index */
    int32_t xinfo_off, yinfo_off;
    /* quadtree roots */
    int32_t n_ini;
    float hx;
    float scale;
    int32_t scaled_patch;
} ss_level;

typedef struct {
    int32_t n_levels;
    int32_t w, h;
    int32_t ini_th, min_th;
    int32_t lap_x0, lap_x1;
    int32_t n_features;
    int32_t kcap;           /* keypoint rows per frame */
    uint32_t block_bytes;   /* one pyramid block */
    int32_t n_cells;        /* all levels */
    int32_t cand_total;     /* u32 per frame */
    int32_t bucket_total;   /* u32 per frame */
    int32_t chunks_total;
    int32_t sel_total;
    int32_t node_total;
    int32_t item_total;
    int32_t tiles2_total;
    int32_t umax[16];
    /* IC_Angle work split of k_orient_describe: lane = (row lane & 31, half lane >> 5) of the disc takes the pixels
     * u0 .. u0 + 15 of its row, of which the bytes set in ic_mask[lane] are inside the disc (ss_geometry.cpp) */
    int32_t ic_pad[2];       /* keeps ic_mask 16-byte aligned (loaded as dwordx4) */
    uint32_t ic_mask[64][4];
    int32_t ic_u0[64];
    /* rBRIEF pattern re-laid per lane: lane L owns test pairs L, 64 + L, 128 + L, 192 + L (x0 y0 x1 y1 as int8 each):
     * one dwordx4 load per lane instead of four dword loads */
    uint32_t pat4[64][4];
    ss_level lv[SS_MAX_LEVELS_];
} ss_geom;

/* cell-window info of one column (or row): is it inside an evaluated FAST window, is it the
 * first / last pixel of that window (no NMS neighbour on that side), which cell */
#define SS_CI_VALID 0x2000u
#define SS_CI_LOW 0x4000u
#define SS_CI_HIGH 0x8000u
#define SS_CI_CELL 0x03FFu

/* resize table entries */
typedef struct {
    uint16_t s0, s1; /* source index and clamped neighbour */
    int16_t a0, a1;  /* 11-bit fixed-point weights */
} ss_rtab;

/* quadtree node: rectangle (UL.x, UR.x, UL.y, BL.y), record segment, flags */
typedef struct {
    uint16_t x0, x1, y0, y1;
    int32_t beg;
    int32_t cnt;  /* > 0 */
    int32_t flags; /* bit0 alive, bit1 no_more, bit2 buffer index of the segment */
} ss_qnode;

/* per-(frame, level) status words written by the kernels */
typedef struct {
    int32_t n_cand;
    int32_t n_sel;
    int32_t error; /* 0 ok; SS_ERR_OVERFLOW-style codes */
    int32_t pad;
} ss_level_state;

#endif
