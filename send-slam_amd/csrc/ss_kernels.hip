/*
 * ss_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the ORB hot path.
 *
 * Stage map; the reference's implementation of every stage is the
 * unvendored ORB-SLAM3 library entered at
 * /root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:594:
 *   K0  k_ingest / k_ingest_gray16   cvtColor RGB/BGR -> gray (or pitched copy) into pyramid level 0
 *   K1  k_resize_lds / k_resize      ORBextractor::ComputePyramid: cv::resize INTER_LINEAR, level by level
 *   K2 + K3a + K6a  k_fast_score     cv::FAST-9-16 corner response R-1 (threshold-free), the NMS cv::FAST runs
 *                       inside each 35-px cell window, and GaussianBlur 7x7 sigma 2 (8-bit fixed-point
 *                       path) -- one staged tile feeds all three; survivors go to per-tile sub-lists
 *   K3b k_bucket_gather, k_cells_emit  survivors regrouped per cell, iniTh -> minTh fallback, ordered
 *                       compaction into the candidate list (rank of every survivor inside its cell)
 *   K4  k_quadtree      ORBextractor::DistributeOctTree, one 4-wave workgroup per (frame, level)
 *   --  k_slots         ORBextractor::operator() output order (lapping-area rule)
 *   K5/K6b k_orient_describe   IC_Angle + fastAtan2, steered rBRIEF (4 x __ballot -> 256 bits)
 *   K7  k_match_mfma_x / k_match_mfma / k_match / k_match_stream / k_match_merge*   Hamming best / second best + ratio test
 *
 * Integer / byte work throughout.  Two contractions on the path run on the matrix cores, both exact: the Hamming
 * distance of K7 as a +-1 dot product (FP4 operands in k_match_mfma_x, i8 in k_match_mfma) and
 * the 7 x 7 Gaussian of K6a as two band-matrix products of 16 x 16 x 32 int8 tiles inside k_fast_score (FT_BLUR_MFMA;
 * the vector-pipe form is a compile-time switch).  Nothing else is reshaped into a GEMM.  Every kernel takes the batch slot in blockIdx.y or .z so one launch covers a
 * batch of frames.  Level 0 of the pyramid is read IN PLACE from the caller's buffer when that is a
 * 1-channel image with 16-byte aligned rows (lvl0 != NULL below); otherwise k_ingest writes it into the
 * pyramid block first.
 * Float steps are single IEEE operations (-ffp-contract=off, ss_float_steps.h).
 */
#include <algorithm>
#include <cstddef>
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "ss_constants.h"
#include "ss_float_steps.h"
#include "ss_kernels.h"
#include "ss_layout.h"

#define WAVE 64
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt()
{
    return ((uint64_t)1 << lane_id()) - 1;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int min3(int a, int b, int c) { return imin(imin(a, b), c); }
__device__ __forceinline__ int max3(int a, int b, int c) { return imax(imax(a, b), c); }

/* XCD-aware block index (cdna guide T1): workgroups are dealt round-robin over the 8 XCDs, each
 * with its own L2, so blockIdx b and b+1 never share an L2.  This bijective remap gives every XCD
 * one CONTIGUOUS range of logical indices: horizontally adjacent tiles / cells (consecutive
 * logical index) then run on the same XCD and share the 64-B sectors their halos overlap on.
 * Placement only changes speed, never results. */
__device__ __forceinline__ int xcd_remap(int bid, int n)
{
    const int q = n >> 3, r = n & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

/* sum over the 64 lanes, returned uniform.  DPP row operations (no LDS round trips as __shfl_xor's ds_bpermute):
 * two quad permutes and two mirrors leave every lane of a 16-lane row with the row's sum, two row broadcasts carry
 * the rows' sums to lane 63 */
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); /* row_half_mirror */
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); /* row_mirror */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); /* row_bcast15 -> rows 1, 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); /* row_bcast31 -> rows 2, 3 */
    return __builtin_amdgcn_readlane(v, 63);
}

/* inclusive prefix sum over the 64 lanes: four row shifts inside the 16-lane rows, two row broadcasts across them */
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); /* row_bcast15 -> rows 1, 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); /* row_bcast31 -> rows 2, 3 */
    return v;
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

/* ------------------------------------------------------------------------------------ */
/* K0: ingest.  channels == 1: pitched copy; 3/4: fixed-point gray, c0/c1/c2 = weights of   */
/* byte 0/1/2 (host swaps RY/BY with the calibration's rgb flag).  4 pixels per thread.    */
/* ------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_ingest(const uint8_t *__restrict__ src, int channels,
                                                int64_t row_stride, int64_t frame_stride,
                                                int c0, int c1, int c2, uint8_t *__restrict__ pyr,
                                                const ss_geom *__restrict__ g)
{
    const int w = g->lv[0].w, h = g->lv[0].h, pitch = g->lv[0].pitch;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h || x4 >= w) return;
    const uint8_t *s = src + (int64_t)blockIdx.z * frame_stride + (int64_t)y * row_stride;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = imin(x4 + i, w - 1);
        uint32_t v;
        if (channels == 1) {
            v = s[x];
        } else {
            const uint8_t *p = s + (int64_t)x * channels;
            v = (uint32_t)(p[0] * c0 + p[1] * c1 + p[2] * c2 + (1 << (SS_GRAY_SHIFT - 1))) >> SS_GRAY_SHIFT;
        }
        out |= (v & 0xFFu) << (8 * i);
    }
    *(uint32_t *)(pyr + (size_t)blockIdx.z * g->block_bytes + g->lv[0].off + (size_t)y * pitch + x4) = out;
}

/* 1-channel input whose rows are 16-byte aligned: 16 pixels per thread, dwordx4 in, dwordx4 out */
__global__ __launch_bounds__(256) void k_ingest_gray16(const uint8_t *__restrict__ src, int64_t row_stride, int64_t frame_stride,
                                                       uint8_t *__restrict__ pyr, const ss_geom *__restrict__ g)
{
    const int w = g->lv[0].w, h = g->lv[0].h, pitch = g->lv[0].pitch;
    const int x16 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 16;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h || x16 >= w) return;
    const uint8_t *sp = src + (int64_t)blockIdx.z * frame_stride + (int64_t)y * row_stride + x16;
    uint8_t *dp = pyr + (size_t)blockIdx.z * g->block_bytes + g->lv[0].off + (size_t)y * pitch + x16;
    if (x16 + 16 <= w) {
        *(uint4 *)dp = *(const uint4 *)sp;
    } else { /* last, partial group of the row: the pitch covers it, the source row may not */
        uint32_t out[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) out[i >> 2] |= (uint32_t)sp[imin(i, w - 1 - x16)] << (8 * (i & 3));
        *(uint4 *)dp = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

/* ------------------------------------------------------------------------------------ */
/* K1: one pyramid step.  Host tables hold OpenCV's fixed-point taps (ss_geometry.cpp).   */
/* ------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_resize(uint8_t *__restrict__ pyr, const ss_geom *__restrict__ g,
                                                const ss_rtab *__restrict__ rtab, int level,
                                                const uint8_t *__restrict__ lvl0, int lvl0_pitch, int64_t lvl0_fs)
{
    const ss_level &D = g->lv[level];
    const ss_level &S = g->lv[level - 1];
    const int dx4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dy >= D.h || dx4 >= D.w) return;
    uint8_t *base = pyr + (size_t)blockIdx.z * g->block_bytes;
    const bool inplace = level == 1 && lvl0 != nullptr; /* level 0 lives in the caller's buffer */
    const uint8_t *src = inplace ? lvl0 + (int64_t)blockIdx.z * lvl0_fs : base + S.off;
    const int spitch = inplace ? lvl0_pitch : S.pitch;
    const ss_rtab ty = rtab[D.ytab_off + dy];
    const uint8_t *r0 = src + (size_t)ty.s0 * spitch, *r1 = src + (size_t)ty.s1 * spitch;
    const int b0 = ty.a0, b1 = ty.a1;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const ss_rtab tx = rtab[D.xtab_off + dx4 + i]; /* table padded past w */
        const int h0 = r0[tx.s0] * tx.a0 + r0[tx.s1] * tx.a1;
        const int h1 = r1[tx.s0] * tx.a0 + r1[tx.s1] * tx.a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out |= ((uint32_t)v & 0xFFu) << (8 * i);
    }
    *(uint32_t *)(base + D.off + (size_t)dy * D.pitch + dx4) = out;
}

/* LDS-staged, separable form of the same step for pyramid scale factors <= 1.4 (the source window of a 64x64
 * destination tile then fits 112 B x 80 rows, from a 16-byte aligned column on).  cv::resize's 8-bit linear path is two passes with a rounding in between:
 * h = S[s0] * a0 + S[s1] * a1 per source row, then ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2 >> 2.  Every
 * source row serves two destination rows, so the horizontal sums are formed ONCE per source row (78 rows for 64
 * destination rows at 1.2) and kept in LDS as h >> 4 (16 bits):
 *   pass 1  thread = four destination columns x every 16th source row.  The <= 8 source bytes of the four columns are
 *           12 staged bytes funnelled to 8 (two v_alignbyte); one v_perm_b32 per column picks its two taps' bytes into
 *           16-bit halves and one v_dot2_u32_u16 multiplies them by (a0, a1): two instructions per sum, no byte loads,
 *           no per-tap address.  The taps are carried times 16, so h >> 4 is bytes 1-2 of the product sum and one
 *           v_perm_b32 packs two of them for the store.  s1 = s0 + 1 wherever a1 != 0 (ss_geometry.cpp build_axis_table).
 *   pass 2  thread = four destination columns x four destination rows: two 8-byte LDS reads per row, products on the
 *           24-bit multiplier with SDWA half-word operands.
 * 15 VALU lane-operations per pixel against 26 for the one-pass form that formed every h twice. */
#define RS_TILE_H 64
#define RS_ROWS 80 /* 64 * 1.2 + 2 rounded up */
#define RS_WORDS 28 /* 112-byte source rows from a 16-byte aligned column on: 64 * 1.2 + 2 bytes + 15 of alignment */

__global__ __launch_bounds__(256) void k_resize_lds(uint8_t *__restrict__ pyr, const ss_geom *__restrict__ g,
                                                    const ss_rtab *__restrict__ rtab, int level,
                                                    const uint8_t *__restrict__ lvl0, int lvl0_pitch, int64_t lvl0_fs)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[RS_ROWS + 1][RS_WORDS]; /* + 1: a funnel may read one dword past the last row's window */
    __shared__ __attribute__((aligned(8))) uint16_t hbuf[RS_ROWS][SS_TILE_W];
    __shared__ ss_rtab yt[RS_TILE_H];
    const ss_level &D = g->lv[level];
    const ss_level &S = g->lv[level - 1];
    const int tiles_x = (D.w + SS_TILE_W - 1) / SS_TILE_W;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int x0 = (tile % tiles_x) * SS_TILE_W, y0 = (tile / tiles_x) * RS_TILE_H;
    uint8_t *base = pyr + (size_t)blockIdx.y * g->block_bytes;
    const bool inplace = level == 1 && lvl0 != nullptr; /* level 0 lives in the caller's buffer */
    const uint8_t *src = inplace ? lvl0 + (int64_t)blockIdx.y * lvl0_fs : base + S.off;
    const int spitch = inplace ? lvl0_pitch : S.pitch;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    /* this thread's four columns: taps straight to registers (x table is padded past w); row taps of the tile to LDS */
    ss_rtab rx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) rx[i] = rtab[D.xtab_off + x0 + 4 * tx + i];
    if (threadIdx.x < RS_TILE_H) yt[threadIdx.x] = rtab[D.ytab_off + imin(y0 + (int)threadIdx.x, D.h - 1)];
    const int gx0 = (int)rtab[D.xtab_off + x0].s0 & ~15;     /* first source byte, 16-byte aligned (rows are: levels have a
                                                              * 64-byte pitch, the caller's level 0 a 16-byte one) */
    const int gy0 = (int)rtab[D.ytab_off + imin(y0, D.h - 1)].s0;
    const int gy1 = (int)rtab[D.ytab_off + imin(y0 + RS_TILE_H - 1, D.h - 1)].s1; /* last source row used */
    {
        /* 16 bytes per load; all of a thread's loads first, then the LDS stores: one memory latency instead of one per round */
        constexpr int VEC = RS_WORDS / 4, ROUNDS = (RS_ROWS * VEC + 255) / 256;
        uint4 v[ROUNDS];
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = (int)threadIdx.x + 256 * it;
            const int r = idx / VEC, c = idx - r * VEC;
            const int gy = gy0 + r, gx = gx0 + 16 * c;
            /* rows and pitches are far below 2^24 and a level below 2^32 bytes: v_mad_u32_u24 instead of a 64-bit multiply */
            v[it] = (idx < RS_ROWS * VEC && gy <= gy1 && gx < spitch) ? *(const uint4 *)(src + (__umul24((uint32_t)gy, (uint32_t)spitch) + (uint32_t)gx)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = (int)threadIdx.x + 256 * it;
            if (idx < RS_ROWS * VEC) ((uint4 *)&lds[0][0])[idx] = v[it];
        }
    }
    /* pass 1 constants: dword b of the window row holds the first column's s0; the funnel {B:A} = 8 bytes from byte
     * o0 = s0[0] - gx0 on; column i reads funnel bytes k, k + 1 (k = s0[i] - s0[0] <= 6) as two 16-bit halves */
    const int o0 = (int)rx[0].s0 - gx0, bword = o0 >> 2;
    const uint32_t fsh = (uint32_t)o0 & 3u;
    uint32_t sel[4], taps[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t k = (uint32_t)((int)rx[i].s0 - (int)rx[0].s0);
        sel[i] = 0x0C000C00u | k | ((k + 1) << 16);
        taps[i] = ((uint32_t)(uint16_t)rx[i].a0 << 4) | ((uint32_t)(uint16_t)rx[i].a1 << 20); /* 16 x the 11-bit weights: <= 2^15 */
    }
    __syncthreads();
    const int n_rows = gy1 - gy0 + 1;
#pragma unroll
    for (int rr = 0; rr < RS_ROWS / 16; rr++) {
        const int r = ty + 16 * rr;
        if (r >= n_rows) break;
        const uint32_t *w = &lds[r][bword];
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
        const uint32_t fa = __builtin_amdgcn_alignbyte(d1, d0, fsh), fb = __builtin_amdgcn_alignbyte(d2, d1, fsh);
        uint32_t hs[4]; /* 16 h < 2^24: h >> 4 is its bytes 1 and 2, and one v_perm_b32 packs two of them */
#pragma unroll
        for (int i = 0; i < 4; i++) hs[i] = dot2_u16(__builtin_amdgcn_perm(fb, fa, sel[i]), taps[i], 0u);
        *(uint2 *)&hbuf[r][4 * tx] = make_uint2(__builtin_amdgcn_perm(hs[1], hs[0], 0x06050201u), __builtin_amdgcn_perm(hs[3], hs[2], 0x06050201u));
    }
    __syncthreads();
    const int dx4 = x0 + 4 * tx;
    if (dx4 >= D.w) return;
#pragma unroll
    for (int rr = 0; rr < RS_TILE_H / 16; rr++) {
        const int ly = ty + 16 * rr, dy = y0 + ly;
        if (dy >= D.h) break;
        const ss_rtab ry = yt[ly];
        const uint2 h0 = *(const uint2 *)&hbuf[ry.s0 - gy0][4 * tx], h1 = *(const uint2 *)&hbuf[ry.s1 - gy0][4 * tx];
        const int b0 = ry.a0, b1 = ry.a1;
        const uint32_t p0[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
        const uint32_t p1[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            /* every factor is below 2^12 (taps) or 2^16 (h >> 4): the 24-bit multiplier gives the 32-bit products */
            const int v = ((__mul24(b0, (int)p0[i]) >> 16) + (__mul24(b1, (int)p1[i]) >> 16) + 2) >> 2;
            out |= ((uint32_t)v & 0xFFu) << (8 * i);
        }
        *(uint32_t *)(base + D.off + (__umul24((uint32_t)dy, (uint32_t)D.pitch) + (uint32_t)dx4)) = out;
    }
}

/* Two pyramid steps in one launch: a block builds a 48x64 tile of level l+1 AND, first, the part of level l it is
 * resized from -- 64 columns x <= 80 rows, computed from level l-1 into LDS by the same two passes as k_resize_lds, never
 * read back from memory.  Level l is still written (every block stores the columns / rows from its own window origin up to
 * its right / lower neighbour's: each pixel by exactly one block), level l+1 reads level l's ROUNDED bytes from LDS, so
 * both levels are bit for bit what two launches produce.  The windows overlap by 1-2 pixels: 14 % more pixels of level l
 * are computed, and staging level l for the second step is gone; seven pyramid launches become four.
 * The host launches this only for level pairs whose windows it has checked against the buffer sizes below, tile by tile
 * (ssk_resize_pair_fits). */
#define RP_TILE_W 48
#define RP_A_ROWS 80                      /* rows of level l a block holds (= RS_ROWS: the second step's source window) */
#define RP_S_ROWS 98                      /* rows of level l-1 staged for them: 80 * 1.2 + 2 */
#define RP_B_WORDS 20                     /* pitch of the level-l window in LDS: 64 computed bytes + 16 a funnel may touch */

__global__ __launch_bounds__(256) void k_resize_pair(uint8_t *__restrict__ pyr, const ss_geom *__restrict__ g,
                                                     const ss_rtab *__restrict__ rtab, int level,
                                                     const uint8_t *__restrict__ lvl0, int lvl0_pitch, int64_t lvl0_fs)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_s[RP_S_ROWS + 1][RS_WORDS]; /* level l-1 window */
    __shared__ __attribute__((aligned(8))) uint16_t hbuf[RP_S_ROWS][SS_TILE_W];      /* horizontal sums of either step */
    __shared__ uint32_t lds_a[RP_A_ROWS + 1][RP_B_WORDS];                            /* level l window */
    __shared__ ss_rtab yt_a[RP_A_ROWS], yt_b[RS_TILE_H];
    const ss_level &S = g->lv[level - 1];
    const ss_level &A = g->lv[level];
    const ss_level &B = g->lv[level + 1];
    const int tiles_x = (B.w + RP_TILE_W - 1) / RP_TILE_W, tiles_y = (B.h + RS_TILE_H - 1) / RS_TILE_H;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int x0 = tile_x * RP_TILE_W, y0 = tile_y * RS_TILE_H;
    uint8_t *base = pyr + (size_t)blockIdx.y * g->block_bytes;
    const bool inplace = level == 1 && lvl0 != nullptr; /* level 0 lives in the caller's buffer */
    const uint8_t *src = inplace ? lvl0 + (int64_t)blockIdx.y * lvl0_fs : base + S.off;
    const int spitch = inplace ? lvl0_pitch : S.pitch;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;

    /* the level-l window: origin, what of it this block stores, how many rows it computes */
    const int ax0 = (int)rtab[B.xtab_off + x0].s0 & ~3;
    const int ay0 = (int)rtab[B.ytab_off + y0].s0;
    const int ax_end = tile_x == tiles_x - 1 ? A.pitch : ((int)rtab[B.xtab_off + x0 + RP_TILE_W].s0 & ~3);
    const int ay_end = tile_y == tiles_y - 1 ? A.h : (int)rtab[B.ytab_off + y0 + RS_TILE_H].s0;
    const int ay_need = (int)rtab[B.ytab_off + imin(y0 + RS_TILE_H - 1, B.h - 1)].s1 + 1;
    const int n_a = imax(ay_need, ay_end) - ay0; /* <= RP_A_ROWS (host check) */

    /* step 1 taps: this thread's four columns of level l; row taps of the window to LDS */
    ss_rtab rx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) rx[i] = rtab[A.xtab_off + ax0 + 4 * tx + i]; /* x tables are padded past w */
    if (threadIdx.x < RP_A_ROWS) yt_a[threadIdx.x] = rtab[A.ytab_off + imin(ay0 + (int)threadIdx.x, A.h - 1)];
    if (threadIdx.x >= 128 && threadIdx.x < 128 + RS_TILE_H) yt_b[threadIdx.x - 128] = rtab[B.ytab_off + imin(y0 + (int)threadIdx.x - 128, B.h - 1)];
    const int gx0 = (int)rtab[A.xtab_off + ax0].s0 & ~15;
    const int gy0 = (int)rtab[A.ytab_off + ay0].s0;
    const int gy1 = (int)rtab[A.ytab_off + imin(ay0 + n_a - 1, A.h - 1)].s1; /* <= gy0 + RP_S_ROWS - 1 (host check) */
    {
        constexpr int VEC = RS_WORDS / 4, ROUNDS = (RP_S_ROWS * VEC + 255) / 256;
        uint4 v[ROUNDS];
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = (int)threadIdx.x + 256 * it;
            const int r = idx / VEC, c = idx - r * VEC;
            const int gy = gy0 + r, gx = gx0 + 16 * c;
            v[it] = (idx < RP_S_ROWS * VEC && gy <= gy1 && gx < spitch) ? *(const uint4 *)(src + (__umul24((uint32_t)gy, (uint32_t)spitch) + (uint32_t)gx)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < ROUNDS; it++) {
            const int idx = (int)threadIdx.x + 256 * it;
            if (idx < RP_S_ROWS * VEC) ((uint4 *)&lds_s[0][0])[idx] = v[it];
        }
    }
    /* horizontal pass of one step: source rows [0, n_rows) of a window with `pitch` dwords per row whose byte 0 is source
     * column `origin`; four destination columns per thread with taps t[4] (k_resize_lds pass 1) */
    auto h_pass = [&](const uint32_t *win, int pitch, int origin, const ss_rtab (&t)[4], int n_rows, int max_rounds) {
        const int o0 = (int)t[0].s0 - origin, bword = o0 >> 2;
        const uint32_t fsh = (uint32_t)o0 & 3u;
        uint32_t sel[4], taps[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t k = (uint32_t)((int)t[i].s0 - (int)t[0].s0);
            sel[i] = 0x0C000C00u | k | ((k + 1) << 16);
            taps[i] = ((uint32_t)(uint16_t)t[i].a0 << 4) | ((uint32_t)(uint16_t)t[i].a1 << 20);
        }
        for (int rr = 0; rr < max_rounds; rr++) {
            const int r = ty + 16 * rr;
            if (r >= n_rows) break;
            const uint32_t *w = win + r * pitch + bword;
            const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
            const uint32_t fa = __builtin_amdgcn_alignbyte(d1, d0, fsh), fb = __builtin_amdgcn_alignbyte(d2, d1, fsh);
            uint32_t hs[4];
#pragma unroll
            for (int i = 0; i < 4; i++) hs[i] = dot2_u16(__builtin_amdgcn_perm(fb, fa, sel[i]), taps[i], 0u);
            *(uint2 *)&hbuf[r][4 * tx] = make_uint2(__builtin_amdgcn_perm(hs[1], hs[0], 0x06050201u), __builtin_amdgcn_perm(hs[3], hs[2], 0x06050201u));
        }
    };
    /* vertical pass: four pixels of destination row `ry` from the sums of its two source rows (k_resize_lds pass 2) */
    auto v_dword = [&](const ss_rtab ry, int row0) -> uint32_t {
        const uint2 h0 = *(const uint2 *)&hbuf[ry.s0 - row0][4 * tx], h1 = *(const uint2 *)&hbuf[ry.s1 - row0][4 * tx];
        const int b0 = ry.a0, b1 = ry.a1;
        const uint32_t p0[4] = {h0.x & 0xFFFFu, h0.x >> 16, h0.y & 0xFFFFu, h0.y >> 16};
        const uint32_t p1[4] = {h1.x & 0xFFFFu, h1.x >> 16, h1.y & 0xFFFFu, h1.y >> 16};
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int v = ((__mul24(b0, (int)p0[i]) >> 16) + (__mul24(b1, (int)p1[i]) >> 16) + 2) >> 2;
            out |= ((uint32_t)v & 0xFFu) << (8 * i);
        }
        return out;
    };
    __syncthreads();
    h_pass(&lds_s[0][0], RS_WORDS, gx0, rx, gy1 - gy0 + 1, (RP_S_ROWS + 15) / 16);
    __syncthreads();
    {
        const int dx4 = ax0 + 4 * tx;
        const bool store_col = dx4 < ax_end && dx4 < A.w;
        for (int rr = 0; rr < RP_A_ROWS / 16; rr++) {
            const int ly = ty + 16 * rr, dy = ay0 + ly;
            if (ly >= n_a) break;
            const uint32_t out = v_dword(yt_a[ly], gy0);
            lds_a[ly][tx] = out;
            if (store_col && dy < ay_end) *(uint32_t *)(base + A.off + (__umul24((uint32_t)dy, (uint32_t)A.pitch) + (uint32_t)dx4)) = out;
        }
    }
    /* step 2: the tile of level l+1 from the window just built */
#pragma unroll
    for (int i = 0; i < 4; i++) rx[i] = rtab[B.xtab_off + x0 + 4 * tx + i];
    __syncthreads();
    h_pass(&lds_a[0][0], RP_B_WORDS, ax0, rx, n_a, RP_A_ROWS / 16);
    __syncthreads();
    const int dx4 = x0 + 4 * tx;
    if (4 * tx >= RP_TILE_W || dx4 >= B.w) return;
    for (int rr = 0; rr < RS_TILE_H / 16; rr++) {
        const int ly = ty + 16 * rr, dy = y0 + ly;
        if (dy >= B.h) break;
        *(uint32_t *)(base + B.off + (__umul24((uint32_t)dy, (uint32_t)B.pitch) + (uint32_t)dx4)) = v_dword(yt_b[ly], ay0);
    }
}

/* ------------------------------------------------------------------------------------ */
/* K2 + K3a + K6a: FAST-9-16 response, non-maximum suppression, Gaussian blur.            */
/* R = max over the 16 arcs of 9 contiguous ring pixels of min(v - p) and of min(p - v); a  */
/* pixel is a corner at threshold t iff R > t and its cv::cornerScore is R - 1 for every    */
/* such t, so ONE map serves iniTh and minTh.  Stored: R - 1 if R > minTh else 0.           */
/* 64x32 tile per 256-thread block.  The tile, a 1-px ring of neighbours (whose scores the  */
/* NMS needs) and their 3-px FAST rings are staged in LDS (96-byte rows, 16-byte loads); 7x7 */
/* Gaussian (K6a) needs the same bytes, so it rides in the same kernel: one global read      */
/* feeds all three.                                                                         */
/* NMS (K3a) runs inside the FAST cell windows: cv::FAST is called per cell sub-image, so a */
/* pixel competes only with neighbours of ITS window (FAST_t's ring buffers hold 0 outside  */
/* it); the per-column / per-row tables say whether a pixel is evaluated at all and whether  */
/* it is the first / last of its window.  A survivor at minTh that scores >= iniTh is also   */
/* a survivor at iniTh and vice versa (every neighbour below iniTh is below it), so one      */
/* unordered bucket per cell serves both thresholds: the cell's count word holds the number */
/* of survivors (low half, = bucket slot allocator) and of those >= iniTh (high half).      */
/* ------------------------------------------------------------------------------------ */
#ifndef FT_BLUR_MFMA
/* 2 (default): both passes of the Gaussian on the matrix pipe (below).  0: on the vector pipe (v_dot4_u32_u8 rows into LDS,
 * v_dot2_u32_u16 columns): 6 % more vector instructions in k_fast_score, 0.3036 against 0.2896 ms per 64 frames alone, 116.9 k
 * against 118.4 k frames/s with four batches in flight (profiles/tools/build_variant.sh valub -DFT_BLUR_MFMA=0).  An earlier
 * form that put only the horizontal pass on the matrix pipe and handed the sums over through LDS measured slower than the
 * vector form. */
#define FT_BLUR_MFMA 2
#endif
#if FT_BLUR_MFMA == 2
/* 2: BOTH passes of K6a on the matrix pipe, the sums of the first handed to the second in registers (no LDS in between).
 * A wave takes 16 columns of the tile and all 32 rows; v_mfma_i32_16x16x32_i8 throughout (lane = (index & 15, group g = lane >> 4),
 * operand bytes k = 8 g .. 8 g + 7; result registers r = rows 4 g + r of the lane's column).
 *   Horizontal: D1[row][col] = sum_k A[row][k] Bh[k][col] for three blocks of 16 staged rows (blur rows rho = 16 mb + 4 g + r);
 *     A = staged bytes 16 w + 8 + k of the row, as int8 = pixel - 128; Bh[k][j] = tap k - j - 5: a constant band.  D1 = the
 *     8.8 fixed-point row sum less 32768: fits int16.
 *   Vertical, transposed so that a lane ends up with four adjacent pixels of one row: D2[col][out row] = sum_k H[col][k]
 *     Bv[k][out row], H = D1 as the A operand -- the layouts agree, a lane holds rows of ITS column in both -- once for the
 *     high bytes of the sums and once for the low bytes (xor 0x80: a signed digit, + 128): out rows 16 nb + n read blur rows
 *     16 nb + n .. + 6, which live in the row blocks nb (operand bytes 0-3) and nb + 1 (bytes 4-7); Bv[k][n] = tap rho - n of
 *     the byte's row rho = 4 g + s (s < 4) or 16 + 4 g + s - 4.  Contraction indices are free to permute: that is all the
 *     'transposition' costs.  sum = 256 * (high part) + (low part) + 256 * (32768 + 128); + 2^15 >> 16 as cv::GaussianBlur. */
struct blur_frag { uint32_t w[64][2]; };
constexpr int blur_tap(int t) { return t == 0 || t == 6 ? SS_GAUSS_K0 : t == 1 || t == 5 ? SS_GAUSS_K1 : t == 2 || t == 4 ? SS_GAUSS_K2 : t == 3 ? SS_GAUSS_K3 : 0; }
constexpr blur_frag make_blur_h()
{
    blur_frag f{};
    for (int lane = 0; lane < 64; lane++)
        for (int b = 0; b < 8; b++) {
            const int j = lane & 15, k = 8 * (lane >> 4) + b;
            f.w[lane][b >> 2] |= (uint32_t)blur_tap(k - j - 5) << (8 * (b & 3));
        }
    return f;
}
constexpr blur_frag make_blur_v()
{
    blur_frag f{};
    for (int lane = 0; lane < 64; lane++)
        for (int s = 0; s < 8; s++) {
            const int n = lane & 15, g = lane >> 4, rho = s < 4 ? 4 * g + s : 16 + 4 * g + (s - 4);
            f.w[lane][s >> 2] |= (uint32_t)blur_tap(rho - n) << (8 * (s & 3));
        }
    return f;
}
__device__ const blur_frag g_blur_h = make_blur_h(), g_blur_v = make_blur_v();
static_assert(SS_GAUSS_K0 * 2 + SS_GAUSS_K1 * 2 + SS_GAUSS_K2 * 2 + SS_GAUSS_K3 == 256 && SS_GAUSS_K3 < 128, "taps: int8, sum 256");
#endif

#define FT_ROWS (SS_TILE_H2 + 8)        /* staged rows: y0 - 4 .. y0 + 35 */
#define FT_BLUR_ROWS (SS_TILE_H2 + 6)   /* of which the blur uses y0 - 3 .. y0 + 34 */
#define FT_WORDS (SS_TILE_W / 4 + 8)    /* staged bytes: x0 - 16 .. x0 + 79, of which x0 - 4 .. x0 + 67 are used: 16-byte
                                         * aligned rows, so an interior tile is staged by ONE 16-byte load per thread, and a
                                         * 24-dword pitch, on which the four rows a wave reads at once share no bank */
#define FT_XB 16                        /* staged byte of the tile's first pixel */
#define FT_XW (FT_XB / 4)
#define FT_OWORDS (SS_TILE_W / 4 + 2)   /* score tile: bytes x0 - 4 .. x0 + 67 */
#define FT_HALO_PIXELS (2 * (SS_TILE_W + 2) + 2 * SS_TILE_H2)

typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i16x2 as_i16x2(uint32_t v) { return __builtin_bit_cast(i16x2, v); }

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p < 0 ? 0 : (p >= n ? n - 1 : p);
}


#define FT_THREADS (8 * SS_TILE_H2)
#ifndef FT_SKIP
#define FT_SKIP 0 /* accounting builds only (profiles/tools/fast_accounting.sh; results invalid): 1 halo ring, 2 blur H,
                   * 4 arc search, 8 NMS, 16 blur V, 32 compass + queue */
#endif
__global__ __launch_bounds__(FT_THREADS) void k_fast_score(const uint8_t *__restrict__ pyr,
                                                    uint8_t *__restrict__ score,
                                                    const ss_geom *__restrict__ g,
                                                    const uint32_t *__restrict__ tiles,
                                                    const uint16_t *__restrict__ cinfo,
                                                    uint32_t *__restrict__ tsurv,
                                                    uint32_t *__restrict__ thdr,
                                                    ss_level_state *__restrict__ state,
                                                    uint8_t *__restrict__ blur,
                                                    const uint8_t *__restrict__ lvl0, int lvl0_pitch, int64_t lvl0_fs)
{
    /* `score` may be NULL: no later kernel reads the response map (it exists for stage-by-stage tests) */
    /* horizontal Gaussian sums (8 fractional bits, less 32768: int16), packed as (row 2p, row 2p+1) per pixel
     * so the vertical pass is four v_dot2_i32_i16 per output */
#if FT_BLUR_MFMA != 2
    __shared__ __attribute__((aligned(16))) uint32_t hpair[FT_BLUR_ROWS / 2][SS_TILE_W];
#endif
    /* + 1 row: the blur's A operand of the last row / last 16 columns reads up to 40 bytes past it (against zero taps) */
    __shared__ __attribute__((aligned(16))) uint32_t lds[FT_ROWS + 1][FT_WORDS];
    /* scores of the tile and of its 1-px ring: row ly + 1, byte lx + 4 (tile pixels dword-aligned) */
    __shared__ __attribute__((aligned(16))) uint32_t out_tile[SS_TILE_H2 + 2][FT_OWORDS];
    __shared__ uint16_t list[SS_TILE_W * SS_TILE_H2 + FT_HALO_PIXELS + 4];
    __shared__ uint16_t corners[SS_TILE_W * SS_TILE_H2];
    __shared__ uint16_t xinf[SS_TILE_W], yinf[SS_TILE_H2];
    __shared__ uint32_t s_kcnt[SS_TS_HDR]; /* count words of the tile's sub-lists */
    __shared__ int n_list, n_corner;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    /* everything a block needs to know about its tile in ONE 64-byte record (ss_geometry.cpp): the staging loads
     * wait for one scalar load instead of the chain tile word -> level -> level fields */
    const uint32_t *tr = tiles + (size_t)tile * SS_TILE_REC_WORDS;
    const int level = (int)tr[0], x0 = (int)tr[1], y0 = (int)tr[2];
    const int w = (int)tr[3], h = (int)tr[4], pitch = (int)tr[5];
    const int xinfo_off = (int)tr[7], yinfo_off = (int)tr[8];
    const int frame = blockIdx.y;
    const size_t fb = (size_t)frame * g->block_bytes + tr[6];
    /* level 0 may live in the caller's buffer (own pitch); the blurred level is always written with the geometry's */
    const bool inplace = level == 0 && lvl0 != nullptr;
    const uint8_t *img = inplace ? lvl0 + (int64_t)frame * lvl0_fs : pyr + fb;
    const int ipitch = inplace ? lvl0_pitch : pitch;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#if defined(FT_EXP) && FT_EXP == 1 /* timing experiment: every interior tile stages the same (cache-hot) pixels */
#define FT_EXP_HOT 1
#endif

#if FT_BLUR_MFMA == 2
    /* the blur's two constant operands: requested first, needed last */
    uint2 bh2 = *(const uint2 *)&g_blur_h.w[threadIdx.x & 63][0], bv2 = *(const uint2 *)&g_blur_v.w[threadIdx.x & 63][0];
#endif
    if (threadIdx.x == 0) { n_list = 0; n_corner = 0; }
    if (threadIdx.x < SS_TS_HDR) s_kcnt[threadIdx.x] = 0;
    static_assert(((SS_TILE_H2 + 2) * FT_OWORDS) % 4 == 0 && ((SS_TILE_H2 + 2) * FT_OWORDS) / 4 <= FT_THREADS, "out_tile is cleared by one 16-byte store per thread");
    if (threadIdx.x < (SS_TILE_H2 + 2) * FT_OWORDS / 4) ((uint4 *)&out_tile[0][0])[threadIdx.x] = make_uint4(0, 0, 0, 0);
    /* window info of the tile's columns / rows: requested now, stored to LDS after the staging loads have been issued,
     * so that the block waits for the two kinds of loads once, not one after the other */
    uint16_t cinf_v = 0;
    if (threadIdx.x < SS_TILE_W) cinf_v = x0 + (int)threadIdx.x < w ? cinfo[xinfo_off + x0 + threadIdx.x] : (uint16_t)0;
    else if (threadIdx.x < SS_TILE_W + SS_TILE_H2) {
        const int k = (int)threadIdx.x - SS_TILE_W;
        cinf_v = y0 + k < h ? cinfo[yinfo_off + y0 + k] : (uint16_t)0;
    }
    /* stage rows y0-4 .. y0+35, bytes x0-4 .. x0+67 (inside 96-byte rows that start at x0-16).  Pixels outside
     * the image are filled by BORDER_REFLECT_101 (what the blur needs; FAST never evaluates a
     * pixel whose ring leaves the image, so it does not care): only the tiles at the image rim
     * take that path (block-uniform), as dwords: thread (tx, ty) takes column tx of rows ty, ty+16,
     * ty+32; the two rightmost columns go to tx < 2. */
    const bool inner_x = x0 >= 4 && x0 + 68 <= w;
    const bool interior = inner_x && y0 >= 4 && y0 + SS_TILE_H2 + 4 <= h;
    /* rows start 16-byte aligned (levels: 64-byte pitch, 256-byte offsets; the caller's level 0: ss_api.cpp checks) */
    if (interior && x0 >= FT_XB && x0 - FT_XB + 4 * FT_WORDS <= ipitch) {
        /* interior tile (the common case): no reflection; thread t takes 16 bytes: row t / 6, bytes 16 (t % 6) */
        static_assert(FT_ROWS * (FT_WORDS / 4) <= FT_THREADS, "one 16-byte load per thread stages the tile");
#ifdef FT_EXP_HOT
        const uint8_t *tile0 = img + (size_t)60 * ipitch + 48;
#else
        const uint8_t *tile0 = img + (size_t)(y0 - 4) * ipitch + (x0 - FT_XB);
#endif
        const uint32_t r = __umulhi((uint32_t)threadIdx.x, 0xAAAAAAABu) >> 2, c = (uint32_t)threadIdx.x - 6u * r; /* t / 6 */
        static_assert(FT_WORDS / 4 == 6, "t / 6 above");
        if (r < FT_ROWS) *(uint4 *)&lds[r][4 * c] = *(const uint4 *)(tile0 + (__umul24(r, (uint32_t)ipitch) + 16u * c));
    } else {
        /* a tile at the image rim: rows by reflected row index; columns as the same aligned dwords wherever their
         * first byte is inside the row (a level's rows are padded to the pitch, the caller's rows to 16 bytes), and a
         * fix-up pass below for the few bytes BORDER_REFLECT_101 defines outside [0, w) */
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const int r = ty + (FT_THREADS / 16) * rr;
            if (r < FT_ROWS) {
                const uint8_t *row = img + (size_t)reflect101(y0 - 4 + r, h) * ipitch;
                const int gx = x0 - 4 + 4 * tx;
                lds[r][FT_XW - 1 + tx] = (gx >= 0 && gx < w) ? *(const uint32_t *)(row + gx) : 0u;
                if (tx < 2) lds[r][FT_XW + 15 + tx] = (gx + 64 < w) ? *(const uint32_t *)(row + gx + 64) : 0u;
            }
        }
        if (!inner_x) {
            /* the blur reads up to 3 px outside the row (x = -3 .. -1 mirror 3 .. 1, x = w .. w+2 mirror w-2 .. w-4); FAST
             * never evaluates a pixel whose ring leaves the image.  Sources and destinations are inside the staged window. */
            __syncthreads();
            uint8_t *t8 = (uint8_t *)&lds[0][0];
            for (int i = threadIdx.x; i < FT_ROWS * 6; i += FT_THREADS) {
                const int r = i / 6, k = i - 6 * r;
                const int x = k < 3 ? -1 - k : w + (k - 3), sx = k < 3 ? 1 + k : w - 2 - (k - 3);
                const int bx = x - (x0 - FT_XB), bs = sx - (x0 - FT_XB);
                if (bx >= 0 && bx < 4 * FT_WORDS && bs >= 0 && bs < 4 * FT_WORDS) t8[r * (4 * FT_WORDS) + bx] = t8[r * (4 * FT_WORDS) + bs];
            }
        }
    }
    if (threadIdx.x < SS_TILE_W) xinf[threadIdx.x] = cinf_v;
    else if (threadIdx.x < SS_TILE_W + SS_TILE_H2) yinf[threadIdx.x - SS_TILE_W] = cinf_v;
#if FT_BLUR_MFMA == 2
    /* pins the two loads above here, where the block waits for its staging loads anyway (left alone the compiler sinks them
     * to their first use, and the blur starts with a trip to the cache) */
    asm volatile("" : "+v"(bh2.x), "+v"(bh2.y), "+v"(bv2.x), "+v"(bv2.y));
#endif
    __syncthreads();

    /* Phase 1, every pixel: the compass test of cv::FAST.  Any 9 contiguous ring pixels contain
     * at least one pixel of every opposing pair {k, k+8}; with the pairs (0,8) and (4,12): a
     * corner at threshold t needs (d0 > t or d8 > t) and (d4 > t or d12 > t), or the same with
     * d < -t.  Only a few % of the pixels pass; they are queued in LDS and scored in phase 2,
     * the rest get 0 without the 100-op arc search. */
    const int min_th = g->min_th;
    const uint8_t *tile8 = (const uint8_t *)&lds[0][0];
    /* pixel i of row 2ty + rr passed <=> bit 8 i + 3 + 4 rr of cand_bits (the sign bits of min_th - c, gathered by one
     * v_perm_b32 per row) */
    uint32_t cand_bits = 0;
    const i16x2 th2 = as_i16x2((uint32_t)min_th * 0x00010001u);
    /* the four pixels of staged word wx of tile row ly (-1 .. SS_TILE_H2): verdicts in the top bits of the four bytes */
    auto compass_row = [&](int ly, int wx) -> uint32_t {
        /* two pixels per operation: ring values go to the 16-bit halves of a dword (v_perm_b32),
         * differences and the min/max tree are v_pk_*_i16 */
        const uint32_t *lw = &lds[0][0] + (ly + 4) * FT_WORDS + wx; /* flat: word -1 / 18 of a row is its neighbour row's */
        const uint32_t u1 = lw[-3 * FT_WORDS], n1 = lw[3 * FT_WORDS];
        const uint32_t m0 = lw[-1], m1 = lw[0], m2 = lw[1];
        uint32_t neg[2];
#pragma unroll
        for (int pr = 0; pr < 2; pr++) { /* pixels 2pr, 2pr+1 of the thread's four */
            constexpr uint32_t Z = 0x0C000C00u; /* selector bytes 1 and 3 = constant 0 */
            const uint32_t sel_c = Z | (uint32_t)(2 * pr) | ((uint32_t)(2 * pr + 1) << 16);  /* bytes 2pr, 2pr+1 of one dword */
            const i16x2 v = as_i16x2(__builtin_amdgcn_perm(0u, m1, sel_c));
            const i16x2 pu = as_i16x2(__builtin_amdgcn_perm(0u, u1, sel_c));
            const i16x2 pd = as_i16x2(__builtin_amdgcn_perm(0u, n1, sel_c));
            /* x+3: window bytes 7+2pr, 8+2pr = bytes 3+2pr, 4+2pr of {m2:m1}; x-3: bytes 1+2pr, 2+2pr of {m1:m0} */
            const i16x2 pr3 = as_i16x2(__builtin_amdgcn_perm(m2, m1, Z | (uint32_t)(3 + 2 * pr) | ((uint32_t)(4 + 2 * pr) << 16)));
            const i16x2 pl3 = as_i16x2(__builtin_amdgcn_perm(m1, m0, Z | (uint32_t)(1 + 2 * pr) | ((uint32_t)(2 + 2 * pr) << 16)));
            /* min over pairs of max(v - p_k, v - p_k+8) = v - max over pairs of min(p_k, p_k+8), and the mirror image
             * for the bright side: three min/max and one subtraction per side instead of four subtractions first */
            const i16x2 lo_pairs = __builtin_elementwise_max(__builtin_elementwise_min(pd, pu), __builtin_elementwise_min(pr3, pl3));
            const i16x2 hi_pairs = __builtin_elementwise_min(__builtin_elementwise_max(pd, pu), __builtin_elementwise_max(pr3, pl3));
            const i16x2 c = __builtin_elementwise_max(v - lo_pairs, hi_pairs - v);
            neg[pr] = __builtin_bit_cast(uint32_t, th2 - c); /* |c| <= 255: no wrap; negative <=> c > min_th */
        }
        /* bytes 1 and 3 of neg[0], then of neg[1]: their top bits are the four verdicts, in pixel order */
        return __builtin_amdgcn_perm(neg[1], neg[0], 0x07050301u) & 0x80808080u;
    };
#pragma unroll
    for (int rr = 0; rr < ((FT_SKIP & 32) ? 0 : 2); rr++) { /* two rows per thread */
        const uint32_t signs = compass_row(2 * ty + rr, tx + FT_XW);
        cand_bits |= rr ? signs : signs >> 4;
    }
    if (!interior) {
        /* rim tiles: FAST evaluates 3 <= x < w - 3, 3 <= y < h - 3 only */
        uint32_t ok = 0;
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = x0 + 4 * tx + i, y = y0 + 2 * ty + rr;
                ok |= (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) ? 1u << (8 * i + 3 + 4 * rr) : 0u;
            }
        cand_bits &= ok;
    }
    /* Queue slots: a wave-wide inclusive scan of the lanes' candidate counts by six DPP additions, ONE LDS atomic per
     * wave for the block-wide base (the queue is unordered).  Left to the compiler, atomicAdd(&n_list, popc) of a
     * non-uniform value becomes a scalar loop over the active lanes (eight instructions per lane with a candidate). */
    if (__ballot(cand_bits != 0) != 0) {
        const int cnt = (int)__popc(cand_bits);
        int incl = cnt;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false); /* row_shr:1 */
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false); /* row_shr:2 */
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false); /* row_shr:4 */
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false); /* row_shr:8 */
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false); /* row_bcast15 -> rows 1, 3 */
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false); /* row_bcast31 -> rows 2, 3 */
        const int total = __builtin_amdgcn_readlane(incl, 63); /* lane 63 holds the wave's total */
        int base = 0;
        if (lane_id() == 0) base = atomicAdd(&n_list, total);
        base = __builtin_amdgcn_readfirstlane(base);
        uint32_t slot = (uint32_t)(base + incl - cnt);
        const uint32_t code0 = (uint32_t)(((2 * ty + 1) << 8) | (4 * tx + 1));
        if (cand_bits)
        do {
            const uint32_t bit = (uint32_t)__builtin_ctz(cand_bits);
            cand_bits &= cand_bits - 1;
            list[slot++] = (uint16_t)(code0 + (bit >> 3) + ((bit & 4u) << 6)); /* (ly + 1) << 8 | (lx + 1) */
        } while (cand_bits);
    }
    /* the same test for the 1-px ring around the tile (scores the NMS of the edge pixels needs): the row above in wave 0,
     * the row below in wave 1, the columns left and right in wave 2.  The ring's four corner pixels would cost a fourth
     * wave the whole test: they are queued untested (phase 2 scores whatever is queued; the test only spares it work). */
    static_assert(SS_TILE_W == 64 && 2 * SS_TILE_H2 <= 64 * (FT_THREADS / 64 - 2), "ring layout: one wave per row, the columns after them");
    if (!(FT_SKIP & 1) && threadIdx.x < 2 * SS_TILE_W + 2 * SS_TILE_H2) {
        const int i = threadIdx.x;
        int lx, ly;
        if (i < SS_TILE_W) { lx = i; ly = -1; }
        else if (i < 2 * SS_TILE_W) { lx = i - SS_TILE_W; ly = SS_TILE_H2; }
        else if (i < 2 * SS_TILE_W + SS_TILE_H2) { lx = -1; ly = i - 2 * SS_TILE_W; }
        else { lx = SS_TILE_W; ly = i - 2 * SS_TILE_W - SS_TILE_H2; }
        const int x = x0 + lx, y = y0 + ly;
        if (interior || (x >= 3 && x < w - 3 && y >= 3 && y < h - 3)) {
            const uint8_t *c = tile8 + (ly + 4) * (FT_WORDS * 4) + FT_XB + lx;
            const int v = c[0];
            const int p0 = c[3 * (FT_WORDS * 4)], p8 = c[-3 * (FT_WORDS * 4)], p4 = c[3], p12 = c[-3];
            const int dark = v - imax(imin(p0, p8), imin(p4, p12)), bright = imin(imax(p0, p8), imax(p4, p12)) - v;
            if (imax(dark, bright) > min_th) list[atomicAdd(&n_list, 1)] = (uint16_t)(((ly + 1) << 8) | (lx + 1));
        }
    } else if (!(FT_SKIP & 1) && threadIdx.x >= FT_THREADS - 4) {
        const int k = (int)threadIdx.x - (FT_THREADS - 4);
        const int lx = (k & 1) ? SS_TILE_W : -1, ly = (k & 2) ? SS_TILE_H2 : -1;
        const int x = x0 + lx, y = y0 + ly;
        if (interior || (x >= 3 && x < w - 3 && y >= 3 && y < h - 3)) list[atomicAdd(&n_list, 1)] = (uint16_t)(((ly + 1) << 8) | (lx + 1));
    }
#if FT_BLUR_MFMA == 0
    /* K6a horizontal pass on the same staged tile.  The 7 taps of pixel i of a dword sit at bytes i + 1 .. i + 7 of the
     * three aligned dwords around it: one v_dot4_u32_u8 per dword that holds any of them, against the taps shifted into
     * place (zeros elsewhere) -- 2 + 3 + 3 + 2 products for four pixels, no byte funnels */
    if (!(FT_SKIP & 2)) {
        constexpr uint32_t K0 = SS_GAUSS_K0, K1 = SS_GAUSS_K1, K2 = SS_GAUSS_K2, K3 = SS_GAUSS_K3;
        /* pixel 0: bytes 1-3 of w0, 0-3 of w1; pixel 1: 2-3, 0-3, 0; pixel 2: 3, 0-3, 0-1; pixel 3: 0-3, 0-2 */
        constexpr uint32_t A0 = (K0 << 8) | (K1 << 16) | (K2 << 24), B0 = K3 | (K2 << 8) | (K1 << 16) | (K0 << 24);
        constexpr uint32_t A1 = (K0 << 16) | (K1 << 24), B1 = K2 | (K3 << 8) | (K2 << 16) | (K1 << 24), C1 = K0;
        constexpr uint32_t A2 = K0 << 24, B2 = K1 | (K2 << 8) | (K3 << 16) | (K2 << 24), C2 = K1 | (K0 << 8);
        constexpr uint32_t B3 = K0 | (K1 << 8) | (K2 << 16) | (K3 << 24), C3 = K2 | (K1 << 8) | (K0 << 16);
        /* one item = four pixels of the two rows of a pair: 20 dot products, ONE 16-byte LDS store */
#pragma unroll
        for (int it = 0; it < ((FT_BLUR_ROWS / 2) * 16 + FT_THREADS - 1) / FT_THREADS; it++) {
            /* the leftovers of the second round go to the LAST wave: the first has the longest way to the barrier */
            static_assert((FT_BLUR_ROWS / 2) * 16 <= 2 * FT_THREADS, "two rounds");
            const int idx = it ? 2 * FT_THREADS - 1 - (int)threadIdx.x : (int)threadIdx.x;
            if (idx >= (FT_BLUR_ROWS / 2) * 16) continue;
            const int pair = idx >> 4, q = idx & 15; /* blur rows 2 pair, 2 pair + 1 = staged rows + 1 */
            uint32_t hv[2][4];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t *lr = &lds[2 * pair + k + 1][FT_XW - 1 + q];
                const uint32_t w0 = lr[0], w1 = lr[1], w2 = lr[2];
                hv[k][0] = __builtin_amdgcn_udot4(w0, A0, __builtin_amdgcn_udot4(w1, B0, 0, false), false);
                hv[k][1] = __builtin_amdgcn_udot4(w0, A1, __builtin_amdgcn_udot4(w1, B1, __builtin_amdgcn_udot4(w2, C1, 0, false), false), false);
                hv[k][2] = __builtin_amdgcn_udot4(w0, A2, __builtin_amdgcn_udot4(w1, B2, __builtin_amdgcn_udot4(w2, C2, 0, false), false), false);
                hv[k][3] = __builtin_amdgcn_udot4(w1, B3, __builtin_amdgcn_udot4(w2, C3, 0, false), false);
            }
            /* sums stay below 2^16 (255 * 256) */
            *(uint4 *)&hpair[pair][4 * q] = make_uint4(hv[0][0] | (hv[1][0] << 16), hv[0][1] | (hv[1][1] << 16),
                                                       hv[0][2] | (hv[1][2] << 16), hv[0][3] | (hv[1][3] << 16));
        }
    }
#endif
    __syncthreads();

    /* K6a vertical pass, two output rows per thread (rows 2ty and 2ty+1 read the same four row
     * pairs); + 2^15 >> 16 as cv::GaussianBlur's fixed-point path.  The queue of phase 2 rarely reaches the upper half
     * of the block (a wave per 64 entries): those waves run this pass while the lower half scores, and take the NMS
     * afterwards, while the lower half runs this pass -- the block's critical path loses one of the two. */
#if FT_BLUR_MFMA == 2
    auto blur_cols = [&](int wv) { /* both passes for columns 16 wv .. 16 wv + 15: see g_blur_h */
        if (FT_SKIP & 16) return;
        const int lane = lane_id();
        const int i16 = lane & 15, g = lane >> 4;
        const long bh = (long)(((uint64_t)bh2.y << 32) | bh2.x), bv = (long)(((uint64_t)bv2.y << 32) | bv2.x);
        static_assert(SS_TILE_W == 16 * (FT_THREADS / 64) && SS_TILE_H2 == 32, "a wave per 16 columns, two blocks of 16 output rows");
        uint32_t lo[3], hi[3];
#pragma unroll
        for (int mb = 0; mb < 3; mb++) {
            /* blur row rho = staged row rho + 1; the last block's rows past the window meet zero taps only: any staged row serves */
            const int srow = mb < 2 ? 1 + 16 * mb + i16 : imin(1 + 32 + i16, FT_ROWS);
            const uint2 px = *(const uint2 *)(tile8 + srow * (FT_WORDS * 4) + 16 * wv + 8 + 8 * g);
            const long a = (long)(((uint64_t)(px.y ^ 0x80808080u) << 32) | (px.x ^ 0x80808080u));
            const v4i d = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, bh, v4i{0, 0, 0, 0}, 0, 0, 0);
            /* the lane's four sums (rows 4 g .. 4 g + 3 of the block): their low bytes in one dword, their high bytes in another */
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)d[1], (uint32_t)d[0], 0x05010400u);
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)d[3], (uint32_t)d[2], 0x05010400u);
            lo[mb] = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;
            hi[mb] = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
        }
        constexpr int KC = 256 * (32768 + 128) + 32768;
        const int col = x0 + 16 * wv + 4 * g;
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            const long ah = (long)(((uint64_t)hi[nb + 1] << 32) | hi[nb]), al = (long)(((uint64_t)lo[nb + 1] << 32) | lo[nb]);
            const v4i e = __builtin_amdgcn_mfma_i32_16x16x32_i8(ah, bv, v4i{0, 0, 0, 0}, 0, 0, 0);
            const v4i c = v4i{(int)(((uint32_t)e[0] << 8) + (uint32_t)KC), (int)(((uint32_t)e[1] << 8) + (uint32_t)KC),
                              (int)(((uint32_t)e[2] << 8) + (uint32_t)KC), (int)(((uint32_t)e[3] << 8) + (uint32_t)KC)};
            const v4i f = __builtin_amdgcn_mfma_i32_16x16x32_i8(al, bv, c, 0, 0, 0);
            /* (sum + 2^15) >> 16 is byte 2 of each sum (sums stay below 2^24) */
            const uint32_t out = __builtin_amdgcn_perm((uint32_t)f[1], (uint32_t)f[0], 0x0C0C0602u) | __builtin_amdgcn_perm((uint32_t)f[3], (uint32_t)f[2], 0x06020C0Cu);
            const int y = y0 + 16 * nb + i16;
            if (y < h && col < pitch) *(uint32_t *)(blur + fb + (__umul24((uint32_t)y, (uint32_t)pitch) + (uint32_t)col)) = out;
        }
    };
#ifndef FT_BLUR_SPLIT
#define FT_BLUR_SPLIT 0 /* 1: the upper half of the block blurs all four column blocks while the lower half runs the arc search:
                         * measured 0.2905 against 0.2865 ms per 64 frames alone, 117.3 k against 117.9 k frames/s in the pipeline */
#endif
    auto blur_v = [&]() {
        const int wv = rfl((int)(threadIdx.x >> 6));
        if (FT_BLUR_SPLIT) {
            if (wv >= 2) {
                blur_cols(wv);
                blur_cols(wv - 2);
            }
        } else {
            blur_cols(wv);
        }
    };
#else
    auto blur_v = [&]() {
        if (FT_SKIP & 16) return;
        constexpr uint32_t KA0 = SS_GAUSS_K0 | (SS_GAUSS_K1 << 16), KA1 = SS_GAUSS_K2 | (SS_GAUSS_K3 << 16);
        constexpr uint32_t KA2 = SS_GAUSS_K2 | (SS_GAUSS_K1 << 16), KA3 = SS_GAUSS_K0;
        constexpr uint32_t KB0 = (uint32_t)SS_GAUSS_K0 << 16, KB1 = SS_GAUSS_K1 | (SS_GAUSS_K2 << 16);
        constexpr uint32_t KB2 = SS_GAUSS_K3 | (SS_GAUSS_K2 << 16), KB3 = SS_GAUSS_K1 | (SS_GAUSS_K0 << 16);
        uint32_t va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t p0 = hpair[ty][4 * tx + i], p1 = hpair[ty + 1][4 * tx + i];
            const uint32_t p2 = hpair[ty + 2][4 * tx + i], p3 = hpair[ty + 3][4 * tx + i];
            va[i] = dot2_u16(p0, KA0, dot2_u16(p1, KA1, dot2_u16(p2, KA2, dot2_u16(p3, KA3, 32768u))));
            vb[i] = dot2_u16(p0, KB0, dot2_u16(p1, KB1, dot2_u16(p2, KB2, dot2_u16(p3, KB3, 32768u))));
        }
        /* (sum + 2^15) >> 16 is byte 2 of each sum (sums stay below 2^24): three v_perm_b32 gather four of them */
        const uint32_t out_a = __builtin_amdgcn_perm(va[1], va[0], 0x0C0C0602u) | __builtin_amdgcn_perm(va[3], va[2], 0x06020C0Cu);
        const uint32_t out_b = __builtin_amdgcn_perm(vb[1], vb[0], 0x0C0C0602u) | __builtin_amdgcn_perm(vb[3], vb[2], 0x06020C0Cu);
        const int ya = y0 + 2 * ty;
        if (x0 + 4 * tx < pitch) {
            /* 24-bit multiply + 32-bit offset (a level is far below 2^32 bytes) instead of a 64-bit v_mad_i64_i32 */
            const uint32_t o = __umul24((uint32_t)ya, (uint32_t)pitch) + (uint32_t)(x0 + 4 * tx);
            if (ya < h) *(uint32_t *)(blur + fb + o) = out_a;
            if (ya + 1 < h) *(uint32_t *)(blur + fb + (o + (uint32_t)pitch)) = out_b;
        }
    };
#endif
    const bool upper_half = threadIdx.x >= FT_THREADS / 2; /* wave-uniform */
    if (upper_half) blur_v();

    /* Phase 2, queued pixels only: R = max over the 16 arcs of min9(v - p) and of min9(p - v) */
    constexpr int RDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int RDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    uint8_t *out8 = (uint8_t *)&out_tile[0][0];
    const int n = (FT_SKIP & 4) ? 0 : n_list;
    for (int e = threadIdx.x; e < n; e += FT_THREADS) {
        const int ly = (int)(list[e] >> 8) - 1, lx = (int)(list[e] & 0xFF) - 1;
        const uint8_t *c = tile8 + (ly + 4) * (FT_WORDS * 4) + FT_XB + lx;
        const int v = c[0];
        /* max over arcs of min9(v - p) = v - (min over arcs of max9(p)) and max over arcs of min9(p - v) =
         * (max over arcs of min9(p)) - v: the 16 differences are never formed */
        int p[16];
#pragma unroll
        for (int k = 0; k < 16; k++) p[k] = c[RDY[k] * (FT_WORDS * 4) + RDX[k]];
        int lo3[16], hi3[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            lo3[k] = min3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
            hi3[k] = max3(p[k], p[(k + 1) & 15], p[(k + 2) & 15]);
        }
        int arc_hi = 256, arc_lo = -1; /* min over arcs of max9(p); max over arcs of min9(p) */
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            arc_hi = min3(arc_hi, max3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]),
                          max3(hi3[k + 1], hi3[(k + 4) & 15], hi3[(k + 7) & 15]));
            arc_lo = max3(arc_lo, min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]),
                          min3(lo3[k + 1], lo3[(k + 4) & 15], lo3[(k + 7) & 15]));
        }
        const int R = imax(v - arc_hi, arc_lo - v);
        if (R > min_th) {
            out8[(ly + 1) * (FT_OWORDS * 4) + 4 + lx] = (uint8_t)(R - 1);
            if ((unsigned)lx < SS_TILE_W && (unsigned)ly < SS_TILE_H2) corners[atomicAdd(&n_corner, 1)] = (uint16_t)((ly << 8) | lx);
        }
    }
    __syncthreads();
    /* K3a on the tile's corners: 8 neighbours from LDS, window rules from the tables.  A survivor
     * joins the sub-list of its cell window (a tile meets at most 3 x 2 of them); its slot comes
     * from an LDS counter, so no global atomic is involved: pass 1 here allots the slots, pass 2
     * (after the blur's vertical pass) knows the sub-list offsets and writes the records. */
    const int nc = (FT_SKIP & 8) ? 0 : n_corner, ini_th = g->ini_th;
    const uint32_t tc = tr[9];
    const int col0 = (int)(tc & 0xFFFFu), row0 = (int)(tc >> 16);
    for (int e = (int)((threadIdx.x + FT_THREADS / 2) % FT_THREADS); e < nc; e += FT_THREADS) { /* upper half first */
        const int ly = corners[e] >> 8, lx = corners[e] & 0xFF;
        const uint32_t xi = xinf[lx], yi = yinf[ly];
        uint16_t code = 0xFFFFu;
        if ((xi & SS_CI_VALID) && (yi & SS_CI_VALID)) {
            const uint8_t *c = out8 + (ly + 1) * (FT_OWORDS * 4) + 4 + lx;
            const int sc = c[0];
            /* all eight neighbours are read at once (the score tile has a ring, so every address is valid) and the ones
             * outside the pixel's cell window are masked to 0 -- as conditional reads they were eight dependent LDS round
             * trips on the critical path of the one wave that runs the NMS */
            constexpr int P = FT_OWORDS * 4;
            const int lm = (xi & SS_CI_LOW) ? 0 : 0xFF, rm = (xi & SS_CI_HIGH) ? 0 : 0xFF;
            const int um = (yi & SS_CI_LOW) ? 0 : 0xFF, dm = (yi & SS_CI_HIGH) ? 0 : 0xFF;
            const int n00 = c[-P - 1], n01 = c[-P], n02 = c[-P + 1], n10 = c[-1], n12 = c[1], n20 = c[P - 1], n21 = c[P], n22 = c[P + 1];
            const int m = max3(max3(n00 & (um & lm), n01 & um, n02 & (um & rm)), max3(n10 & lm, n12 & rm, n20 & (dm & lm)),
                               imax(n21 & dm, n22 & (dm & rm)));
            if (sc > m) {
                const int k = ((int)(yi & SS_CI_CELL) - row0) * 3 + ((int)(xi & SS_CI_CELL) - col0);
                const uint32_t old = atomicAdd(&s_kcnt[k], 1u | (sc >= ini_th ? 0x10000u : 0u));
                code = (uint16_t)((k << 12) | (old & 0xFFFu));
            }
        }
        list[e] = code; /* the queue of phase 1 is free again */
    }

    if (!upper_half) blur_v();
    if (score) {
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const int ly = 2 * ty + rr, y = y0 + ly;
            if (y < h && x0 + 4 * tx < pitch) *(uint32_t *)(score + fb + (size_t)y * pitch + x0 + 4 * tx) = out_tile[ly + 1][tx + 1];
        }
    }
    __syncthreads();
    const size_t tslot = (size_t)frame * g->tiles2_total + tile;
    if (threadIdx.x < SS_TS_HDR) thdr[tslot * SS_TS_HDR + threadIdx.x] = s_kcnt[threadIdx.x];
    for (int e = threadIdx.x; e < nc; e += FT_THREADS) {
        const uint32_t code = list[e];
        if (code == 0xFFFFu) continue;
        const int k = (int)(code >> 12);
        int idx = (int)(code & 0xFFFu);
#pragma unroll
        for (int j = 0; j < SS_TS_CELLS - 1; j++) idx += j < k ? (int)(s_kcnt[j] & 0xFFFFu) : 0;
        const int ly = corners[e] >> 8, lx = corners[e] & 0xFF;
        if (idx < SS_TS_CAP)
            tsurv[tslot * SS_TS_CAP + idx] = SS_PACK(x0 + lx - SS_MIN_BORDER, y0 + ly - SS_MIN_BORDER, out8[(ly + 1) * (FT_OWORDS * 4) + 4 + lx]);
        else
            atomicExch(&state[(size_t)frame * SS_MAX_LEVELS_ + level].error, -5);
    }
}

/* ------------------------------------------------------------------------------------ */
/* K3a': 16 lanes per grid cell gather the cell's survivors from the sub-lists of the tiles   */
/* its window is spread over (table built by the host) into the cell's bucket and write the   */
/* cell's count word.  Plain loads and stores: every word is written by exactly one thread,    */
/* nothing needs clearing between batches.                                                     */
/* ------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(256) void k_bucket_gather(const ss_geom *__restrict__ g, const uint32_t *__restrict__ cell_units,
                                                      const uint32_t *__restrict__ tsurv, const uint32_t *__restrict__ thdr,
                                                      uint32_t *__restrict__ bucket, uint32_t *__restrict__ cell_cnt,
                                                      ss_level_state *__restrict__ state)
{
    /* 16 lanes per cell, one per (tile, sub-list) unit: the loads of all units are in flight together */
    const int gid = (int)(blockIdx.x * 256 + threadIdx.x), frame = blockIdx.y;
    const int cell = gid >> 4, u = gid & 15;
    const bool live = cell < g->n_cells;
    int level = 0;
    for (int l = 1; l < g->n_levels; l++)
        if (live && cell >= g->lv[l].cell_base) level = l;
    const ss_level &L = g->lv[level];
    const uint32_t d = live && u < SS_CELL_UNITS ? cell_units[(size_t)cell * SS_CELL_UNITS + u] : 0xFFFFFFFFu;
    int cnt = 0, ini = 0;
    const uint32_t *src = tsurv;
    if (d != 0xFFFFFFFFu) {
        const size_t tslot = (size_t)frame * g->tiles2_total + (d & 0xFFFFFFu);
        const int k = (int)(d >> 24);
        const uint32_t *hdr = thdr + tslot * SS_TS_HDR;
        int pre = 0;
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < SS_TS_CELLS; j++) {
            const uint32_t hw = hdr[j];
            pre += j < k ? (int)(hw & 0xFFFFu) : 0;
            word = j == k ? hw : word;
        }
        cnt = imin((int)(word & 0xFFFFu), imax(SS_TS_CAP - pre, 0));
        ini = (int)(word >> 16);
        src = tsurv + tslot * SS_TS_CAP + pre;
    }
    /* inclusive scan over the cell's 16 lanes = one DPP row: both counts in one word (each < 2^16), four row shifts (a
     * lane whose source is left of the row keeps the 0 of `old`); lane 0 of the row reads the total from lane 15 by a row
     * mirror.  (__shfl_up is ds_bpermute_b32, an LDS round trip per step.) */
    int both = cnt | (ini << 16);
    both += __builtin_amdgcn_update_dpp(0, both, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    both += __builtin_amdgcn_update_dpp(0, both, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    both += __builtin_amdgcn_update_dpp(0, both, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    both += __builtin_amdgcn_update_dpp(0, both, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    const int incl = both & 0xFFFF;
    const int totals = __builtin_amdgcn_update_dpp(0, both, 0x140, 0xF, 0xF, false); /* row_mirror: lane 0 <- lane 15 */
    const int total = totals & 0xFFFF, total_ini = (int)((uint32_t)totals >> 16);
    if (!live) return;
    uint32_t *bk = bucket + (size_t)frame * g->bucket_total + L.bucket_base + (size_t)(cell - L.cell_base) * L.bucket_cap;
    const int pos = incl - cnt;
    for (int i = 0; i < cnt; i++)
        if (pos + i < L.bucket_cap) bk[pos + i] = src[i];
    if (u == 0) {
        if (total > L.bucket_cap) atomicExch(&state[(size_t)frame * SS_MAX_LEVELS_ + level].error, -5);
        cell_cnt[(size_t)frame * g->n_cells + cell] = (uint32_t)imin(total, L.bucket_cap) | ((uint32_t)total_ini << 16);
    }
}

/* ------------------------------------------------------------------------------------ */
/* K3b: ordered compaction.  A cell that kept anything at iniTh uses its survivors that     */
/* score >= iniTh, otherwise all its (minTh) survivors -- the FAST(..., minThFAST) retry.     */
/* Output order = upstream's push_back order: cells row-major, pixels row-major inside a     */
/* cell.  One 256-thread block per 64 consecutive cells of a level: offsets by a scan of the */
/* count words, then one thread per bucket entry computes its rank among the entries of its  */
/* cell (key = y : x) and writes it to its final position.                                   */
/* ------------------------------------------------------------------------------------ */
#define CE_CELLS 64
__device__ __forceinline__ int cell_count_of(uint32_t c) { return (c >> 16) ? (int)(c >> 16) : (int)(c & 0xFFFFu); }

__global__ __launch_bounds__(256) void k_cells_emit(const uint32_t *__restrict__ bucket, const ss_geom *__restrict__ g,
                                                   const uint32_t *__restrict__ cell_cnt, uint32_t *__restrict__ cand,
                                                   ss_level_state *__restrict__ state)
{
    __shared__ int s_all[CE_CELLS + 1], s_out[CE_CELLS + 1];
    __shared__ uint32_t s_cnt[CE_CELLS];
    __shared__ int s_red[4];
    const int frame = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = lane_id();
    int level = 0;
    for (int l = 1; l < g->n_levels; l++)
        if (chunk >= g->lv[l].chunk_base) level = l;
    const ss_level &L = g->lv[level];
    const int n_level_cells = L.n_cols * L.n_rows;
    const int first = (chunk - L.chunk_base) * CE_CELLS;
    const int nc = imin(CE_CELLS, n_level_cells - first);
    const uint32_t *cnt = cell_cnt + (size_t)frame * g->n_cells + L.cell_base;
    ss_level_state *st = state + (size_t)frame * SS_MAX_LEVELS_ + level;

    int part = 0;
    for (int c = tid; c < first; c += 256) part += cell_count_of(cnt[c]);
    part = wave_sum(part);
    if (lane == 0) s_red[tid >> 6] = part;
    if (tid < CE_CELLS) s_cnt[tid] = tid < nc ? cnt[first + tid] : 0u;
    __syncthreads();
    const int before = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tid < WAVE) {
        const uint32_t word = s_cnt[tid];
        const int ia = wave_scan_incl(imin((int)(word & 0xFFFFu), L.bucket_cap)), io = wave_scan_incl(cell_count_of(word));
        s_all[tid + 1] = ia;
        s_out[tid + 1] = before + io;
        if (tid == 0) { s_all[0] = 0; s_out[0] = before; }
    }
    __syncthreads();
    const int total_items = s_all[nc], total_out = s_out[nc];
    if (total_out > L.cand_cap) {
        if (tid == 0) atomicExch(&st->error, -5);
        return;
    }
    if (first + nc == n_level_cells && tid == 0) st->n_cand = total_out;
    const int ini_th = g->ini_th;
    const uint32_t *bk_level = bucket + (size_t)frame * g->bucket_total + L.bucket_base;
    uint32_t *out = cand + (size_t)frame * g->cand_total + L.cand_base;
    for (int t = tid; t < total_items; t += 256) {
        int c = 0; /* the cell whose entry range [s_all[c], s_all[c+1]) holds t */
#pragma unroll
        for (int step = CE_CELLS / 2; step >= 1; step >>= 1)
            if (c + step < nc && s_all[c + step] <= t) c += step;
        const uint32_t word = s_cnt[c];
        const bool use_ini = (word >> 16) != 0;
        const int n_all = s_all[c + 1] - s_all[c];
        const uint32_t *bk = bk_level + (size_t)(first + c) * L.bucket_cap;
        const uint32_t rec = bk[t - s_all[c]];
        if (use_ini && (int)(rec >> 24) < ini_th) continue;
        const uint32_t key = rec & 0xFFFFFFu; /* y in bits 12..23 above x: row-major order */
        int rank = 0;
        for (int j0 = 0; j0 < n_all; j0 += 4) { /* four entries per round: their loads are in flight together */
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; q++) o[q] = bk[imin(j0 + q, n_all - 1)];
#pragma unroll
            for (int q = 0; q < 4; q++)
                rank += (j0 + q < n_all && (!use_ini || (int)(o[q] >> 24) >= ini_th) && (o[q] & 0xFFFFFFu) < key) ? 1 : 0;
        }
        out[s_out[c] + rank] = rec;
    }
}

/* ------------------------------------------------------------------------------------ */
/* K4: DistributeOctTree, array form (DESIGN.md "quadtree"): std::list with push_front /   */
/* erase == append-only node table whose list order is DESCENDING creation index; every    */
/* node owns a contiguous, order-preserving segment of a record array (ping-pong buffers), */
/* so DivideNode is a stable 4-way partition done with ballots.  One 4-wave workgroup per   */
/* tree: the waves divide four nodes of a pass at once.                                     */
/* std::sort(compareNodes) is libstdc++ introsort restated on lane 0 (equal keys must come */
/* out in libstdc++'s order; tests pin the restatement against the real std::sort).        */
/* ------------------------------------------------------------------------------------ */
#define QT_MAX_ITEMS 2048

/* sort item: size[63:32] | UL.x[31:20] | node index[19:0]; compareNodes looks at (size, UL.x) only */
__device__ __forceinline__ bool item_less(uint64_t a, uint64_t b) { return (a >> 20) < (b >> 20); }

__device__ void sort_push_heap(uint64_t *first, int hole, int top, uint64_t value)
{
    int parent = (hole - 1) / 2;
    while (hole > top && item_less(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
__device__ void sort_adjust_heap(uint64_t *first, int hole, int len, uint64_t value)
{
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (item_less(first[second], first[second - 1])) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    sort_push_heap(first, hole, top, value);
}
__device__ void sort_heap_range(uint64_t *first, int n)
{
    if (n >= 2) {
        int parent = (n - 2) / 2;
        for (;;) {
            sort_adjust_heap(first, parent, n, first[parent]);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = n;
    while (last > 1) {
        --last;
        const uint64_t v = first[last];
        first[last] = first[0];
        sort_adjust_heap(first, 0, last, v);
    }
}
/* Lanes of ONE wave hand data to each other through memory: a wave's memory instructions are
 * issued and serviced in program order (LLVM AMDGPU memory model: wavefront scope needs no cache
 * action or wait), so only the COMPILER must be kept from moving accesses.  No s_waitcnt vmcnt(0),
 * no s_barrier: a dependent load simply waits for its own data. */
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

/* libstdc++ std::sort(first, last, compareNodes) -- introsort: median-of-3 partitions down to 16
 * elements with a 2*lg(n) depth limit (heapsort beyond it), then one insertion pass.  The partition
 * loop is executed by ONE WAVE: the sequence of element moves is the sequential algorithm's (so
 * equal keys land exactly where libstdc++ leaves them), but the two inner scans of a partition look
 * at up to 64 elements per LDS round trip (ballot + count trailing zeros) instead of one; control
 * flow is wave-uniform.  The insertion pass is done by the whole workgroup as a rank computation
 * (sort_final_rank). */
__device__ __forceinline__ uint64_t lane_bcast64(uint64_t v, int src_lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

/* `stack`: 64 words of LDS for the pending right-hand ranges (first | last << 12 | depth << 24); a local array
 * would be indexed dynamically and live in scratch memory.  `pos`: scratch LDS for 2 x pos_half u16 (>= n each). */
__device__ __forceinline__ void std_sort_items_wave(uint64_t *a, int n, int lane, uint32_t *stack, uint16_t *pos, int pos_half)
{
    if (n <= 0) return;
    const uint64_t lt = lanemask_lt();
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) lg++;
    int sp = 1;
    if (lane == 0) stack[0] = 0u | ((uint32_t)n << 12) | ((uint32_t)(lg * 2) << 24);
    wave_sync();
    while (sp > 0) {
        --sp;
        const uint32_t top = stack[sp];
        int first = (int)(top & 0xFFFu), last = (int)((top >> 12) & 0xFFFu), depth = (int)(top >> 24);
        while (last - first > 16) {
            if (depth == 0) {
                wave_sync();
                if (lane == 0) sort_heap_range(a + first, last - first);
                wave_sync();
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            const int ia = first + 1, ib = mid, ic = last - 1;
            const uint64_t va = a[ia], vb = a[ib], vc = a[ic];
            int pick; /* __move_median_to_first */
            if (item_less(va, vb)) {
                if (item_less(vb, vc)) pick = ib;
                else if (item_less(va, vc)) pick = ic;
                else pick = ia;
            } else if (item_less(va, vc)) pick = ia;
            else if (item_less(vb, vc)) pick = ic;
            else pick = ib;
            {
                const uint64_t tf = a[first], tp = a[pick];
                wave_sync();
                if (lane == 0) { a[first] = tp; a[pick] = tf; }
                wave_sync();
            }
            const uint64_t pivot = a[first];
            /* __unguarded_partition(first + 1, last, pivot), all of its swaps at once.  The left pointer stops at
             * the positions whose element is not < pivot, in ascending order; the right pointer at those whose
             * element is not > pivot, in descending order; neither ever reads a position the other side has
             * written before the two meet.  So swap k exchanges the k-th stop of each side for as long as the
             * left one is left of the right one, and the returned cut is the first position after the last left
             * stop used that holds an element not < pivot: the next left stop or the last right stop used,
             * whichever comes first.  `pos` (scratch LDS, u16) lists the stops: left ones from the front, right
             * ones -- ascending -- from n_cap. */
            uint16_t *posL = pos, *posR = pos + pos_half;
            int nL = 0, nR = 0;
            for (int base = first + 1; base < last; base += WAVE) {
                const int p = base + lane;
                const bool in = p < last;
                const uint64_t v = a[in ? p : last - 1];
                const bool fl = in && !item_less(v, pivot), fr = in && !item_less(pivot, v);
                const uint64_t ml = __ballot(fl), mr = __ballot(fr);
                if (fl) posL[nL + __popcll(ml & lt)] = (uint16_t)p;
                if (fr) posR[nR + __popcll(mr & lt)] = (uint16_t)p;
                nL += __popcll(ml);
                nR += __popcll(mr);
            }
            wave_sync();
            const int n_both = imin(nL, nR);
            int n_swaps = 0;
            for (int base = 0; base < n_both; base += WAVE) {
                const int k = base + lane;
                const bool ok = k < n_both && posL[k] < posR[nR - 1 - k];
                const uint64_t m = __ballot(ok);
                n_swaps += __popcll(m);
                if (m != ~0ull) break; /* the predicate is monotone in k */
            }
            for (int base = 0; base < n_swaps; base += WAVE) {
                const int k = base + lane;
                if (k < n_swaps) {
                    const int pl = posL[k], pr = posR[nR - 1 - k];
                    const uint64_t vl = a[pl], vr = a[pr];
                    a[pl] = vr;
                    a[pr] = vl;
                }
            }
            wave_sync();
            int lo = n_swaps < nL ? (int)posL[n_swaps] : last;
            if (n_swaps > 0) lo = imin(lo, (int)posR[nR - n_swaps]);
            lo = rfl(lo);
            if (sp < 64) {
                if (lane == 0) stack[sp] = (uint32_t)lo | ((uint32_t)last << 12) | ((uint32_t)depth << 24);
                wave_sync();
                ++sp;
            }
            last = lo;
        }
    }
    wave_sync();
}

/* __final_insertion_sort, without inserting anything.  After the loop above every range longer than 16 has been
 * partitioned (or heap-sorted), so the array is a sequence of blocks of <= 16 elements with every key of a block <=
 * every key of the next.  libstdc++'s final pass is an insertion sort whose inner loop moves an element left only
 * past STRICTLY greater ones: it is stable, and no element can cross into a block of smaller-or-equal keys.  Its
 * result is therefore the stable sort by key of the array as the loop left it:
 *     final position of a[i] = #{j : key_j < key_i} + #{j < i : key_j == key_i},
 * which every thread of the caller computes for its own elements (n broadcast reads of LDS each) instead of one
 * wave shifting elements one insertion at a time -- the insertions were half of the quadtree kernel's time.
 * `tmp` is a second array of n items; the caller separates the two phases with its barrier. */
__device__ __forceinline__ void sort_final_rank(const uint64_t *a, uint64_t *tmp, int n, int tid, int n_threads)
{
    /* (key_j < key_i) or (key_j == key_i and j < i) is ONE unsigned 64-bit comparison once the node index in the low 20
     * bits of an item (not part of the key) is replaced by the item's position */
    for (int i = tid; i < n; i += n_threads) {
        const uint64_t v = a[i];
        const uint64_t mine = (v & ~0xFFFFFull) | (uint32_t)i;
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += 8) { /* eight broadcast reads in flight */
            uint64_t kj[8];
#pragma unroll
            for (int u = 0; u < 8; u++) kj[u] = a[imin(j0 + u, n - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = imin(j0 + u, n - 1); /* a repeat of the last item (j0 + u >= n) compares as itself: counted once below n only */
                rank += (j0 + u < n && ((kj[u] & ~0xFFFFFull) | (uint32_t)j) < mine) ? 1 : 0;
            }
        }
        tmp[rank] = v;
    }
}

/* The first nodes of a tree live in LDS, later ones in the global table.  The cache is kept small on purpose:
 * quadtree blocks share CUs with the FAST / match blocks of the other batches in flight, and every KB they hold
 * is a KB those cannot use (block LDS 37 KB -> 17.5 KB: +4 % frames/s, quadtree time unchanged; shrinking the
 * sort buffer as well packs the trees onto fewer CUs and loses it again). */
#ifndef QT_LDS_NODES
#define QT_LDS_NODES 64
#endif

struct qt_ctx {
    /* the ping-pong record buffers as base + b * distance: an array indexed by a node's buffer bit (and a select between two
     * members just the same) made the compiler keep the whole struct in scratch memory, a memory round trip per use */
    uint32_t *buf0;
    int64_t buf_step; /* bytes from buffer 0 to buffer 1 */
    ss_qnode *nodes;      /* global table */
    ss_qnode *lds_nodes;  /* LDS cache for indices < QT_LDS_NODES */
    int node_cap;
    int n_nodes;
    int size; /* lNodes.size() */
    int error;
};

__device__ __forceinline__ ss_qnode *qt_node(const qt_ctx &q, int idx)
{
    return idx < QT_LDS_NODES ? q.lds_nodes + idx : q.nodes + idx;
}
__device__ __forceinline__ uint32_t *qt_buf(const qt_ctx &q, int b) { return (uint32_t *)((char *)q.buf0 + (int64_t)b * q.buf_step); }
/* stores with the address space spelled out: through generic pointers they would be FLAT instructions, which
 * count against the LDS counter too and would make lds_barrier() wait for global memory */
typedef __attribute__((address_space(1))) uint32_t qt_gu32;
typedef __attribute__((address_space(3))) uint32_t qt_lu32;
/* ... and loads: a FLAT load (generic pointer: "LDS or global?") counts against the LDS counter as well, so the
 * s_waitcnt lgkmcnt(0) of lds_barrier() waited for every record / node load in flight, the prefetched ones included */
typedef const __attribute__((address_space(1))) uint32_t qt_cgu32;
typedef const __attribute__((address_space(3))) uint32_t qt_clu32;
static_assert(sizeof(ss_qnode) == 20, "ss_qnode is five dwords");
__device__ __forceinline__ void qt_store_node(const qt_ctx &q, int idx, const ss_qnode &n)
{
    uint32_t w[5];
    __builtin_memcpy(w, &n, sizeof(w));
    if (idx < QT_LDS_NODES) {
        qt_lu32 *p = (qt_lu32 *)(uint32_t *)(q.lds_nodes + idx);
#pragma unroll
        for (int i = 0; i < 5; i++) p[i] = w[i];
    } else {
        qt_gu32 *p = (qt_gu32 *)(uint32_t *)(q.nodes + idx);
#pragma unroll
        for (int i = 0; i < 5; i++) p[i] = w[i];
    }
}
/* node record by a wave-uniform index: ds_read or global_load, never flat */
__device__ __forceinline__ ss_qnode qt_load_node(const qt_ctx &q, int idx)
{
    uint32_t w[5];
    if (idx < QT_LDS_NODES) {
        qt_clu32 *p = (qt_clu32 *)(const uint32_t *)(q.lds_nodes + idx);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = p[i];
    } else {
        qt_cgu32 *p = (qt_cgu32 *)(const uint32_t *)(q.nodes + idx);
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = p[i];
    }
    ss_qnode n;
    __builtin_memcpy(&n, w, sizeof(w));
    return n;
}
/* a node record that is known to live in the global table (index >= QT_LDS_NODES): 16 + 4 bytes, two stores */
typedef uint32_t qt_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(1))) qt_u32x4 qt_gu32x4;
__device__ __forceinline__ void qt_store_node_global(const qt_ctx &q, int idx, const ss_qnode &n)
{
    uint32_t w[5];
    __builtin_memcpy(w, &n, sizeof(w));
    uint32_t *p = (uint32_t *)(q.nodes + idx);
    qt_u32x4 v;
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    *(qt_gu32x4 *)p = v;
    ((qt_gu32 *)p)[4] = w[4];
}
__device__ __forceinline__ void qt_store_flags(const qt_ctx &q, int idx, int flags)
{
    if (idx < QT_LDS_NODES) ((qt_lu32 *)(uint32_t *)(q.lds_nodes + idx))[4] = (uint32_t)flags;
    else ((qt_gu32 *)(uint32_t *)(q.nodes + idx))[4] = (uint32_t)flags;
}
/* Workgroup barrier that orders LDS traffic only.  The steps of one sweep exchange their counts through LDS;
 * what they write to global memory (moved records, child nodes) is read in the NEXT sweep, after a full
 * __syncthreads().  Waiting for those stores to be acknowledged at every step was 40 % of the kernel's time. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/* ExtractorNode::DivideNode is done in two steps so that the four waves of the workgroup can
 * divide four nodes of a pass at once and still number the children exactly as the sequential
 * algorithm does: (1) qt_count: quadrant populations of one node, by one wave; (2) after the
 * workgroup has exchanged the populations and every wave knows where its children go,
 * qt_scatter: the stable 4-way partition of the node's record segment (ballot + prefix between
 * the ping-pong buffers) and the child node records. */
struct qt_div {
    int x0, x1, y0, y1, xm, ym, beg, cnt, b, flags;
    uint32_t rec0; /* first 64 records stay in registers between the two steps */
    int q0;
};

/* what a division starts from: the node record and the first 64 records of its segment -- two DEPENDENT memory round trips
 * (most node records live in global memory).  The nodes of a sweep's list are final when the sweep starts, so a wave
 * fetches the node record two steps ahead of the step that divides the node and its records one step ahead: neither
 * latency is left in a step's dependency chain. */
struct qt_pre {
    int idx;
    ss_qnode nd;
    uint32_t rec;
};
__device__ __forceinline__ void qt_fetch_node(const qt_ctx &q, int idx, qt_pre &p)
{
    p.idx = idx;
    p.nd = qt_load_node(q, idx);
}
__device__ __forceinline__ void qt_fetch_records(const qt_ctx &q, qt_pre &p)
{
    const int b = (rfl(p.nd.flags) >> 2) & 1;
    p.rec = lane_id() < rfl(p.nd.cnt) ? ((qt_cgu32 *)qt_buf(q, b))[rfl(p.nd.beg) + lane_id()] : 0u;
}

__device__ __forceinline__ void qt_count(const qt_ctx &q, const qt_pre &p, qt_div &d, int c[4])
{
    const ss_qnode &nd = p.nd;
    d.x0 = rfl(nd.x0); d.x1 = rfl(nd.x1); d.y0 = rfl(nd.y0); d.y1 = rfl(nd.y1);
    d.beg = rfl(nd.beg); d.cnt = rfl(nd.cnt); d.flags = rfl(nd.flags);
    d.b = (d.flags >> 2) & 1;
    d.xm = d.x0 + ((d.x1 - d.x0 + 1) >> 1); /* ceil((float)(UR.x-UL.x)/2) */
    d.ym = d.y0 + ((d.y1 - d.y0 + 1) >> 1);
    qt_cgu32 *src = (qt_cgu32 *)(qt_buf(q, d.b) + d.beg);
    const int lane = lane_id();
    d.rec0 = p.rec;
    d.q0 = lane < d.cnt ? (SS_PX(p.rec) >= d.xm ? 1 : 0) | (SS_PY(p.rec) >= d.ym ? 2 : 0) : 4;
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = __popcll(__ballot(d.q0 == k));
    /* a large node (the first passes): four chunks of 64 per round, their loads in flight together */
    for (int base = WAVE; base < d.cnt; base += 4 * WAVE) {
        uint32_t rec[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * WAVE + lane;
            rec[u] = i < d.cnt ? src[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * WAVE + lane;
            const int quad = i < d.cnt ? (SS_PX(rec[u]) >= d.xm ? 1 : 0) | (SS_PY(rec[u]) >= d.ym ? 2 : 0) : 4;
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] += __popcll(__ballot(quad == k));
        }
    }
}

/* children get node indices first_child, first_child+1, ... in n1..n4 order (empty ones skipped) */
__device__ __forceinline__ void qt_scatter(const qt_ctx &q, int idx, const qt_div &d, const int c[4], int first_child)
{
    qt_cgu32 *src = (qt_cgu32 *)(qt_buf(q, d.b) + d.beg);
    qt_gu32 *dst = (qt_gu32 *)(qt_buf(q, d.b ^ 1) + d.beg); /* global_store, not flat_store (lds_barrier) */
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    int run[4];
    run[0] = 0;
    run[1] = c[0];
    run[2] = c[0] + c[1];
    run[3] = c[0] + c[1] + c[2];
    const int o[4] = {run[0], run[1], run[2], run[3]};
    /* one store per chunk of 64 records: a lane picks the ballot of ITS quadrant and the quadrant's running offset (selects)
     * instead of four masked stores, each with its own prefix and exec-mask switch */
    auto place = [&](uint32_t rec, int quad) {
        const uint64_t m0 = __ballot(quad == 0), m1 = __ballot(quad == 1), m2 = __ballot(quad == 2), m3 = __ballot(quad == 3);
        const uint64_t mine = quad == 0 ? m0 : quad == 1 ? m1 : quad == 2 ? m2 : m3;
        const int base = quad == 0 ? run[0] : quad == 1 ? run[1] : quad == 2 ? run[2] : run[3];
        if (quad < 4) dst[base + __popcll(mine & lt)] = rec;
        run[0] += __popcll(m0);
        run[1] += __popcll(m1);
        run[2] += __popcll(m2);
        run[3] += __popcll(m3);
    };
    place(d.rec0, d.q0);
    for (int base = WAVE; base < d.cnt; base += 4 * WAVE) {
        uint32_t rec[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * WAVE + lane;
            rec[u] = i < d.cnt ? src[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + u * WAVE >= d.cnt) break;
            const int i = base + u * WAVE + lane;
            place(rec[u], i < d.cnt ? (SS_PX(rec[u]) >= d.xm ? 1 : 0) | (SS_PY(rec[u]) >= d.ym ? 2 : 0) : 4);
        }
    }
    /* the (up to) four child records: lane k writes child k, so the block is five store instructions, not twenty issued
     * one after the other by lane 0 */
    if (lane < 4) {
        const int k = lane;
        const int ck = k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3];
        const int ok = k == 0 ? o[0] : k == 1 ? o[1] : k == 2 ? o[2] : o[3];
        const int before = (k > 0 && c[0] > 0) + (k > 1 && c[1] > 0) + (k > 2 && c[2] > 0); /* non-empty quadrants before k */
        if (ck > 0) {
            ss_qnode ch;
            ch.x0 = (uint16_t)((k & 1) ? d.xm : d.x0);
            ch.x1 = (uint16_t)((k & 1) ? d.x1 : d.xm);
            ch.y0 = (uint16_t)((k & 2) ? d.ym : d.y0);
            ch.y1 = (uint16_t)((k & 2) ? d.y1 : d.ym);
            ch.beg = d.beg + ok;
            ch.cnt = ck;
            ch.flags = 1 | (ck == 1 ? 2 : 0) | ((d.b ^ 1) << 2);
            if (first_child >= QT_LDS_NODES) qt_store_node_global(q, first_child + before, ch); /* wave-uniform: all but the first nodes */
            else qt_store_node(q, first_child + before, ch);
        }
    }
    if (lane == 0) qt_store_flags(q, idx, d.flags & ~1); /* lNodes.erase */
}

#ifndef QT_WAVES
#define QT_WAVES 4
#endif
#define QT_THREADS (QT_WAVES * 64)

__global__ __launch_bounds__(QT_THREADS) void k_quadtree(const ss_geom *__restrict__ g, const uint32_t *__restrict__ cand,
                                                  uint32_t *__restrict__ qbuf0, uint32_t *__restrict__ qbuf1,
                                                  ss_qnode *__restrict__ nodes_all, int32_t *__restrict__ lists_all,
                                                  uint32_t *__restrict__ sel, ss_level_state *__restrict__ state, int items_cap)
{
    /* dynamic LDS, sized by the launch to the largest per-level list of the geometry (2 x items_cap sort entries -- the array and
     * the target of its final rank pass -- and the two expandable-node lists of items_cap ints): LDS a tree
     * holds is LDS the other batches' FAST blocks cannot use while it is resident */
    extern __shared__ uint64_t items[];
    __shared__ ss_qnode lds_nodes[QT_LDS_NODES];
    __shared__ __attribute__((aligned(16))) int grp_cnt[QT_WAVES][4];
    __shared__ int wave_alive[QT_WAVES];
    __shared__ uint32_t sort_stack[64];
    /* frame fastest: the eight XCDs take workgroups round-robin, and with the level fastest every level-0 tree
     * (the long ones) of a batch landed on one XCD, the other kernels in flight waiting for that XCD's share */
    const int frame = blockIdx.x, level = blockIdx.y;
    const ss_level &L = g->lv[level];
    ss_level_state *st = state + (size_t)frame * SS_MAX_LEVELS_ + level;
    const int lane = lane_id();
    const int wave = rfl((int)(threadIdx.x >> 6));
    const uint64_t lt = lanemask_lt();
    const int n_cand = rfl(st->n_cand);
    if (rfl(st->error) != 0 || n_cand == 0) {
        if (threadIdx.x == 0) st->n_sel = 0;
        return;
    }
    const int N = L.quota;
    const uint32_t *in = cand + (size_t)frame * g->cand_total + L.cand_base;
    qt_ctx q;
    q.buf0 = qbuf0 + (size_t)frame * g->cand_total + L.cand_base;
    q.buf_step = (int64_t)((intptr_t)qbuf1 - (intptr_t)qbuf0);
    q.nodes = nodes_all + (size_t)frame * g->node_total + L.node_base;
    q.lds_nodes = lds_nodes;
    q.node_cap = L.node_cap;
    q.n_nodes = 0;
    q.size = 0;
    q.error = 0;
    /* two expandable-node lists of item_cap ints each, swapped per pass */
    /* ... in LDS behind the sort arrays (items_cap >= every level's item_cap): reading the next node to divide is
     * then not a global-memory round trip at the head of every step's dependency chain */
    (void)lists_all;
    int32_t *list_a = (int32_t *)(items + 2 * items_cap);
    int32_t *list_b = list_a + items_cap;
    const int list_cap = L.item_cap;

    /* roots: vpIniNodes[kp.pt.x / hX]; list order r0, r1, ... == descending creation index,
     * so root i gets node index n_ini-1-i.  Stable partition of the candidates by root (wave 0). */
    const int n_ini = L.n_ini;
    if (wave == 0) {
        const float hx = L.hx;
        const int height = (L.h - SS_EDGE_THRESHOLD + 3) - SS_MIN_BORDER;
        int root_beg = 0;
        for (int r = 0; r < n_ini; r++) {
            int cnt_r = 0;
            for (int base = 0; base < n_cand; base += 4 * WAVE) { /* four chunks per round: their loads in flight together */
                uint32_t rec[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = base + u * WAVE + lane;
                    rec[u] = i < n_cand ? in[i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = base + u * WAVE + lane;
                    const bool mine = i < n_cand && (int)((float)SS_PX(rec[u]) / hx) == r;
                    const uint64_t m = __ballot(mine);
                    if (mine) q.buf0[root_beg + cnt_r + __popcll(m & lt)] = rec[u];
                    cnt_r += __popcll(m);
                }
            }
            ss_qnode nd;
            nd.x0 = (uint16_t)(int)(hx * (float)r);
            nd.x1 = (uint16_t)(int)(hx * (float)(r + 1));
            nd.y0 = 0;
            nd.y1 = (uint16_t)height;
            nd.beg = root_beg;
            nd.cnt = cnt_r;
            nd.flags = (cnt_r > 0 ? 1 : 0) | (cnt_r == 1 ? 2 : 0); /* empty roots are erased */
            if (lane == 0) qt_store_node(q, n_ini - 1 - r, nd);
            root_beg += cnt_r;
        }
        if (lane == 0) grp_cnt[0][0] = root_beg;
    }
    __syncthreads();
    if (grp_cnt[0][0] != n_cand) q.error = -5; /* a candidate outside every root: cannot happen */
    q.n_nodes = n_ini;

    /* expandable roots in creation order (= root n_ini-1 first) */
    int32_t *cur = list_a, *nxt = list_b;
    int n_cur = 0;
    for (int idx = 0; idx < n_ini; idx++) {
        const int fl = rfl(qt_load_node(q, idx).flags);
        if (fl & 1) q.size++;
        if ((fl & 1) && !(fl & 2)) {
            if (threadIdx.x == 0) cur[n_cur] = idx;
            n_cur++;
        }
    }
    __syncthreads();

    /* One sweep over a list of nodes to divide, in the given order, four at a time (one per
     * wave).  `stop_at_n`: the sorted phase, which stops as soon as lNodes.size() >= N.  Returns
     * through n_nxt the expandable children appended to `nxt`, through n_to_expand their count. */
    auto sweep = [&](auto node_at, int n_list, bool stop_at_n, int &n_nxt, int &n_to_expand) -> bool {
        bool stopped = false;
        qt_pre pre, pre2; /* this wave's next step (node + records), and the one after (node) */
        pre.idx = pre2.idx = -1;
        pre.rec = 0;
        if (wave < n_list) {
            qt_fetch_node(q, rfl(node_at(wave)), pre);
            qt_fetch_records(q, pre);
        }
        if (wave + QT_WAVES < n_list) qt_fetch_node(q, rfl(node_at(wave + QT_WAVES)), pre2);
        for (int k0 = 0; k0 < n_list && !stopped && q.error == 0; k0 += QT_WAVES) {
            const int k = k0 + wave;
            const bool have = k < n_list;
            qt_div d;
            int c[4] = {0, 0, 0, 0};
            const qt_pre now = pre;
            const int idx = now.idx;
            pre = pre2;
            if (k + QT_WAVES < n_list) qt_fetch_records(q, pre); /* its node record was requested a step ago */
            if (k + 2 * QT_WAVES < n_list) qt_fetch_node(q, rfl(node_at(k + 2 * QT_WAVES)), pre2);
            if (have) qt_count(q, now, d, c);
            /* what the other waves need of this division: how many children (non-empty quadrants) and how many of them can be
             * divided again, in one word; the four words are one 16-byte LDS read */
            if (lane == 0)
                grp_cnt[wave >> 2][wave & 3] = ((c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0)) | (((c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1)) << 8);
            lds_barrier();
            /* every thread replays the sequential bookkeeping of the (up to) four divisions */
            int my_first_child = 0, my_first_nxt = 0;
            bool my_go = false;
            /* made wave-uniform by name (v_readfirstlane): as values loaded from LDS the compiler must treat them as per-lane,
             * and the bookkeeping below, which every lane replays identically, came out as ~200 dependent vector instructions
             * and exec-mask branches per step (2000 cycles on a SIMD this wave has to itself); on the scalar unit, and from
             * the two sums per division instead of its four counts, it is a fraction */
            static_assert(QT_WAVES <= 4 || QT_WAVES == 8, "grp_cnt rows of four words");
            int words[QT_WAVES];
#pragma unroll
            for (int w4 = 0; w4 < QT_WAVES; w4 += 4) {
                const int4 v = *(const int4 *)&grp_cnt[w4 / 4][0];
                words[w4] = rfl(v.x);
                if (w4 + 1 < QT_WAVES) words[w4 + 1] = rfl(v.y);
                if (w4 + 2 < QT_WAVES) words[w4 + 2] = rfl(v.z);
                if (w4 + 3 < QT_WAVES) words[w4 + 3] = rfl(v.w);
            }
#pragma unroll
            for (int w = 0; w < QT_WAVES; w++) {
                if (k0 + w >= n_list) break;
                const int made = words[w] & 0xFF;
                const int expandable = words[w] >> 8;
                if (q.n_nodes + 4 > q.node_cap || n_nxt + expandable > list_cap) {
                    q.error = -5;
                    break;
                }
                if (w == wave) {
                    my_go = true;
                    my_first_child = q.n_nodes;
                    my_first_nxt = n_nxt;
                }
                q.n_nodes += made;
                q.size += made - 1;
                n_nxt += expandable;
                n_to_expand += expandable;
                if (stop_at_n && q.size >= N) {
                    stopped = true;
                    break;
                }
            }
            if (have && my_go && q.error == 0) {
                qt_scatter(q, idx, d, c, my_first_child);
                if (lane < 4) { /* lane k appends child k to the next list if it can be divided again */
                    const int k = lane;
                    const int ck = k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3];
                    const int child = my_first_child + (k > 0 && c[0] > 0) + (k > 1 && c[1] > 0) + (k > 2 && c[2] > 0);
                    const int pos = my_first_nxt + (k > 0 && c[0] > 1) + (k > 1 && c[1] > 1) + (k > 2 && c[2] > 1);
                    if (ck > 1) nxt[pos] = child;
                }
            }
            lds_barrier();
        }
        __syncthreads(); /* the sweep's global stores are visible to the whole workgroup from here */
        return stopped;
    };

    bool finish = false;
    while (!finish && q.error == 0) {
        const int prev_size = q.size;
        int n_to_expand = 0, n_nxt = 0;
        /* one full pass: lit walks the list (descending creation index); every node that existed
         * at the start and is not final is divided */
        {
            const int32_t *cl = cur;
            const int nc = n_cur;
            sweep([&](int k) { return cl[nc - 1 - k]; }, n_cur, false, n_nxt, n_to_expand);
        }
        { int32_t *t = cur; cur = nxt; nxt = t; }
        n_cur = n_nxt;
        if (q.error) break;
        if (q.size >= N || q.size == prev_size) {
            finish = true;
        } else if (q.size + n_to_expand * 3 > N) {
            while (!finish && q.error == 0) {
                const int prev2 = q.size;
                const int n_prev = n_cur;
                if (n_prev > items_cap) { q.error = -5; break; }
                /* vPrevSizeAndPointerToNode, in creation order; key = size, then UL.x */
                for (int j = threadIdx.x; j < n_prev; j += QT_THREADS) {
                    const int idx = cur[j];
                    const ss_qnode nd = *qt_node(q, idx);
                    items[j] = ((uint64_t)(uint32_t)nd.cnt << 32) | ((uint64_t)nd.x0 << 20) | (uint32_t)idx;
                }
                __syncthreads();
                if (wave == 0) std_sort_items_wave(items, n_prev, lane, sort_stack, (uint16_t *)(items + items_cap), items_cap);
                __syncthreads();
                sort_final_rank(items, items + items_cap, n_prev, (int)threadIdx.x, QT_THREADS);
                __syncthreads();
                for (int j = threadIdx.x; j < n_prev; j += QT_THREADS) items[j] = items[items_cap + j];
                __syncthreads();
                n_nxt = 0;
                int dummy = 0;
                const bool stopped = sweep([&](int k) { return (int)((uint32_t)items[n_prev - 1 - k] & 0xFFFFFu); }, n_prev, true,
                                           n_nxt, dummy);
                (void)stopped;
                { int32_t *t = cur; cur = nxt; nxt = t; }
                n_cur = n_nxt;
                if (q.size >= N || q.size == prev2) finish = true;
            }
        }
    }

    /* retain the best point of each node, list order (descending creation index) */
    uint32_t *out = sel + (size_t)frame * g->sel_total + L.sel_base;
    int n_out = 0;
    if (q.error == 0) {
        for (int hi = q.n_nodes - 1; hi >= 0; hi -= QT_THREADS) {
            const int idx = hi - (int)threadIdx.x;
            ss_qnode nd;
            nd.flags = 0;
            if (idx >= 0) nd = *qt_node(q, idx);
            const bool alive = idx >= 0 && (nd.flags & 1);
            const uint64_t m = __ballot(alive);
            if (lane == 0) wave_alive[wave] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < QT_WAVES; w++) {
                if (w < wave) before += wave_alive[w];
                total += wave_alive[w];
            }
            const int pos = n_out + before + __popcll(m & lt);
            if (alive && pos < L.sel_cap) {
                const uint32_t *seg = qt_buf(q, (nd.flags >> 2) & 1) + nd.beg;
                uint32_t best = seg[0];
                for (int k = 1; k < nd.cnt; k++) {
                    const uint32_t r = seg[k];
                    if (SS_PR(r) > SS_PR(best)) best = r;
                }
                out[pos] = SS_PACK(SS_PX(best) + SS_MIN_BORDER, SS_PY(best) + SS_MIN_BORDER, SS_PR(best));
            }
            n_out += total;
            __syncthreads();
        }
        if (n_out > L.sel_cap) q.error = -5;
    }
    if (threadIdx.x == 0) {
        st->n_sel = q.error ? 0 : n_out;
        if (q.error) atomicExch(&st->error, q.error);
    }
}

/* test hook: the device's std::sort restatement on caller data (ss_debug_sort) */
__global__ __launch_bounds__(64) void k_debug_sort(uint64_t *__restrict__ data, int n)
{
    __shared__ uint64_t items[QT_MAX_ITEMS];
    const int lane = lane_id();
    for (int i = lane; i < n; i += WAVE) items[i] = data[i];
    wave_sync();
    __shared__ uint32_t sort_stack[64];
    __shared__ uint64_t sorted[QT_MAX_ITEMS];
    std_sort_items_wave(items, n, lane, sort_stack, (uint16_t *)sorted, QT_MAX_ITEMS);
    wave_sync();
    sort_final_rank(items, sorted, n, lane, WAVE);
    wave_sync();
    for (int i = lane; i < n; i += WAVE) data[i] = sorted[i];
}

/* ------------------------------------------------------------------------------------ */
/* ORBextractor::operator() output order: keypoints whose scaled x lies in the lapping     */
/* area fill the arrays from the back (stereoIndex--), the others from the front           */
/* (monoIndex++), levels in order.  One wave per frame computes each keypoint's slot.      */
/* kp_ref[slot] = level << 16 | index in the level's selected list.                        */
/* ------------------------------------------------------------------------------------ */
__global__ __launch_bounds__(64) void k_slots(const ss_geom *__restrict__ g, const uint32_t *__restrict__ sel,
                                              const ss_level_state *__restrict__ state,
                                              uint32_t *__restrict__ kp_ref, int32_t *__restrict__ n_kp,
                                              int32_t *__restrict__ level_counts, int32_t *__restrict__ frame_error)
{
    const int frame = blockIdx.x, lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const ss_level_state *st = state + (size_t)frame * SS_MAX_LEVELS_;
    int total = 0, err = 0;
    for (int l = 0; l < g->n_levels; l++) {
        total += st[l].n_sel;
        if (st[l].error) err = st[l].error;
    }
    if (total > g->kcap) err = -5;
    if (lane < SS_MAX_LEVELS_) level_counts[(size_t)frame * SS_MAX_LEVELS_ + lane] = lane < g->n_levels ? st[lane].n_sel : 0;
    if (lane == 0) {
        n_kp[frame] = err ? 0 : total;
        frame_error[frame] = err;
    }
    if (err) return;
    const float lap0 = (float)g->lap_x0, lap1 = (float)g->lap_x1;
    uint2 *ref = (uint2 *)kp_ref + (size_t)frame * g->kcap; /* (level << 16 | index, the record itself) */
    int mono = 0, stereo = total - 1;
    for (int l = 0; l < g->n_levels; l++) {
        const ss_level &L = g->lv[l];
        const uint32_t *s = sel + (size_t)frame * g->sel_total + L.sel_base;
        const int n = st[l].n_sel;
        for (int base = 0; base < n; base += WAVE) {
            const int i = base + lane;
            bool is_st = false, valid = i < n;
            if (valid) {
                float x = (float)SS_PX(s[i]);
                if (l != 0) x = __fmul_rn(x, L.scale);
                is_st = x >= lap0 && x <= lap1;
            }
            const uint64_t ms = __ballot(valid && is_st), mm = __ballot(valid && !is_st);
            if (valid) {
                const int slot = is_st ? stereo - __popcll(ms & lt) : mono + __popcll(mm & lt);
                ref[slot] = make_uint2(((uint32_t)l << 16) | (uint32_t)i, s[i]);
            }
            stereo -= __popcll(ms);
            mono += __popcll(mm);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* K5 + K6b: one wave per keypoint.  IC_Angle: lanes = 31 rows x 2 halves of the radius-15 */
/* disc, integer moments reduced across the wave, fastAtan2 restated.  rBRIEF: lane L does */
/* pairs L, L+64, L+128, L+192; __ballot(t0 < t1) IS descriptor bytes 8k .. 8k+7 (bit i of  */
/* byte j = test 8j+i).                                                                    */
/* ------------------------------------------------------------------------------------ */
static_assert(offsetof(ss_geom, ic_mask) % 16 == 0 && offsetof(ss_geom, pat4) % 16 == 0, "ic_mask and pat4 are loaded as dwordx4");

/* Operand format of the matrix-core matcher (k_match_mfma_x): a descriptor is 256 FP4 (e2m1) values, bit set -> +1.0
 * (nibble 0x2), bit clear -> -1.0 (nibble 0xA), bit b in nibble b of the 128-byte row.  Four bits -> four nibbles:
 * nibble -> four 0 / 1 bytes by one multiplication, bytes squeezed to nibbles by two shift-or-and steps, then
 * 0xA ^ (bit << 3). */
#define SS_X_ROW 128
__device__ __forceinline__ uint32_t fp4_of_4bits(uint32_t nib)
{
    uint32_t p = (nib * 0x00204081u) & 0x01010101u;
    p = (p | (p >> 4)) & 0x00110011u;
    p = (p | (p >> 8)) & 0x00001111u;
    return 0xAAAAu ^ (p << 3);
}

/* STEER_FMA: how the rotated tap coordinates cvRound(x*b + y*a), cvRound(x*a - y*b) are evaluated.  false = two
 * products and one sum, each rounded (the C expression as written); true = what GCC's FMA contraction makes of it
 * when upstream is built -O3 -march=native (CMakeLists.txt:10-13): the FIRST product fused into the sum,
 * fma(x, b, y*a) and fma(x, a, -(y*b)).  Which one the reference binary runs is machine- and compiler-dependent and
 * unpinned (tests/golden/ref_dump/README.md); both are implemented and tested against the oracle's two forms.
 * KP keypoints per wave, side by side: a keypoint is a chain of dependent memory round trips (reference record ->
 * level geometry -> patch + window -> LDS) with little arithmetic in between, so a wave that walks two chains at once
 * keeps twice the bytes in flight for the same lifetime (the loads of both are issued together, stage by stage). */
#ifndef OD_KP
#define OD_KP 1 /* measured in the four-context pipeline: 1 -> 93.5-95.4 k frames/s, 2 -> 92.5 k (twice the LDS per block) */
#endif
#ifndef OD_SEQ
#define OD_SEQ 0 /* 1 with OD_KP = 2 or 4: bit-exact, fewer vector instructions, 113.2 k against 117.8 k frames/s in the pipeline */
#endif
#ifndef OD_PITCH
#define OD_PITCH 80
#endif
template <bool STEER_FMA, int KP>
__global__ __launch_bounds__(256) void k_orient_describe(const ss_geom *__restrict__ g, const uint8_t *__restrict__ pyr,
                                                         const uint8_t *__restrict__ blur,
                                                         const uint32_t *__restrict__ sel,
                                                         const uint32_t *__restrict__ kp_ref,
                                                         const int32_t *__restrict__ n_kp,
                                                         ss_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
                                                         const uint8_t *__restrict__ lvl0, int lvl0_pitch, int64_t lvl0_fs,
                                                         uint8_t *__restrict__ desc_x)
{
    /* all blocks of a frame on one XCD: keypoints whose patches share 64-B lines then share an L2 */
    const int logical = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    const int frame = logical / (int)gridDim.x;
    const int slot0 = rfl(((logical - frame * (int)gridDim.x) * 4 + (int)(threadIdx.x >> 6)) * KP);
    const int n_frame = n_kp[frame];
    if (slot0 >= n_frame) return;
    const int lane = lane_id();
    const int kcap = g->kcap;
    /* IC_Angle: the 31-row patch around the keypoint is staged in LDS as aligned dwords (5 coalesced loads per lane
     * instead of 16 byte gathers); lane = (row, half) of the disc.  The rBRIEF taps land anywhere in the 37 x 37 blurred
     * window around the keypoint (tap radius <= sqrt(338)): as global byte gathers, every one of a lane's 8 taps was a
     * separate cache-line lookup in the CU's vector cache (~40 distinct lines per wave-instruction: the kernel was bound
     * by that).  The window does not depend on the angle, so it is staged with the patch: 37 rows x 10 aligned dwords,
     * coalesced, one memory latency for both. */
    /* 16-byte pieces: the texture addresser, which takes 64 lane addresses per load instruction whatever their width,
     * was the busiest unit of this kernel with dword loads (TA_BUSY 75 %); a row of the patch is 3 pieces from the
     * 16-byte boundary at or below kx - 15, a row of the window 4 pieces from the one at or below kx - 18 */
    /* OD_SEQ: the KP keypoints of a wave go through ONE patch buffer and ONE window buffer one after the other (their loads
     * are still issued together and wait in registers): the LDS of KP = 1, the angle / sine / cosine evaluation shared by KP */
    constexpr int KL = OD_SEQ ? 1 : KP;
    __shared__ __attribute__((aligned(16))) uint32_t patch_all[4][KL][31 * 12 + 4]; /* + 4: the last row's masked-off tail read */
    /* window rows at an OD_PITCH-byte pitch (>= 64, multiple of 16): at 64 bytes rows r and r + 2 share their banks; at 80 only
     * rows r and r + 8 do, and the byte gathers of the 64 lanes (anywhere in the window) collide less */
    __shared__ __attribute__((aligned(16))) uint32_t win_all[4][KL][37 * (OD_PITCH / 4)];
    int slot[KP], level[KP], kx[KP], ky[KP], resp[KP], pitch[KP], px0[KP], wx0[KP];
    size_t fb[KP];
    uint2 ref[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        slot[k] = slot0 + k < n_frame ? slot0 + k : slot0; /* an odd count: the last wave walks its keypoint twice, writes it once */
        ref[k] = ((const uint2 *)kp_ref)[(size_t)frame * kcap + slot[k]]; /* k_slots left the record next to its reference */
    }
    /* this lane's four sample pairs of the pattern: independent of the keypoint, requested first */
    const uint4 pat4 = ((const uint4 *)g->pat4)[lane];
    const uint32_t pat[4] = {pat4.x, pat4.y, pat4.z, pat4.w};
    /* this lane's share of the disc (row lane & 31, left or right half): first u and the byte masks of its <= 16 pixels */
    const uint4 icm = ((const uint4 *)g->ic_mask)[lane];
    const int u0 = g->ic_u0[lane];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        level[k] = rfl((int)(ref[k].x >> 16));
        const ss_level &L = g->lv[level[k]];
        const uint32_t rec = (uint32_t)rfl((int)ref[k].y);
        kx[k] = SS_PX(rec);
        ky[k] = SS_PY(rec);
        resp[k] = SS_PR(rec);
        fb[k] = (size_t)frame * g->block_bytes + L.off;
        pitch[k] = L.pitch;
        px0[k] = (kx[k] - SS_HALF_PATCH) & ~15; /* >= 0: keypoints stay 19 px inside the level */
        wx0[k] = (kx[k] - 18) & ~15;
    }
    uint4 pv[KP][2], wv[KP][3];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        const bool inplace = level[k] == 0 && lvl0 != nullptr; /* level 0 lives in the caller's buffer */
        const int ipitch = inplace ? lvl0_pitch : pitch[k];
        const uint8_t *p0 = (inplace ? lvl0 + (int64_t)frame * lvl0_fs : pyr + fb[k]) + (size_t)(ky[k] - SS_HALF_PATCH) * ipitch + px0[k];
        const uint8_t *b0 = blur + fb[k] + (size_t)(ky[k] - 18) * pitch[k] + wx0[k];
#pragma unroll
        for (int it = 0; it < 2; it++) { /* 31 rows x 3 pieces = 93 */
            const int idx = lane + WAVE * it;
            const int r = idx / 3, c = idx - r * 3;
            pv[k][it] = idx < 31 * 3 ? *(const uint4 *)(p0 + (__umul24((uint32_t)r, (uint32_t)ipitch) + 16u * (uint32_t)c)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 3; it++) { /* 37 rows x 4 pieces = 148 */
            const int idx = lane + WAVE * it;
            wv[k][it] = idx < 37 * 4 ? *(const uint4 *)(b0 + (__umul24((uint32_t)(idx >> 2), (uint32_t)pitch[k]) + 16u * (uint32_t)(idx & 3))) : make_uint4(0, 0, 0, 0);
        }
    }
    auto store_patch = [&](int k) {
        uint4 *pl = (uint4 *)&patch_all[threadIdx.x >> 6][OD_SEQ ? 0 : k][0];
#pragma unroll
        for (int it = 0; it < 2; it++)
            if (lane + WAVE * it < 31 * 3) pl[lane + WAVE * it] = pv[k][it];
    };
    auto store_window = [&](int k) {
        uint4 *wl = (uint4 *)&win_all[threadIdx.x >> 6][OD_SEQ ? 0 : k][0];
#pragma unroll
        for (int it = 0; it < 3; it++)
            if (lane + WAVE * it < 37 * 4) wl[((lane + WAVE * it) >> 2) * (OD_PITCH / 16) + ((lane + WAVE * it) & 3)] = wv[k][it];
    };
    if (!OD_SEQ) {
#pragma unroll
        for (int k = 0; k < KP; k++) {
            store_patch(k);
            store_window(k);
        }
        wave_sync();
    }
    int m10[KP], m01[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        if (OD_SEQ) {
            if (k > 0) wave_sync(); /* the reads of the keypoint before are done */
            store_patch(k);
            wave_sync();
        }
        /* the lane's pixels are 16 consecutive bytes of its staged row starting at u0 (the tail masked off): five
         * aligned dwords, four v_alignbyte, then m10 = sum u * I = u0 * sum I + sum k * I_k and m01 = v * sum I as
         * eight v_dot4_u32_u8 -- integer sums, so the order of additions is free */
        const int row = imin(lane & 31, 30); /* lanes 31 and 63 carry zero masks */
        const int sb = (kx[k] - px0[k]) + u0; /* 0 .. 30: byte offset in the staged row */
        const uint32_t *rw = &patch_all[threadIdx.x >> 6][OD_SEQ ? 0 : k][row * 12 + (sb >> 2)];
        const uint32_t sh = (uint32_t)sb & 3u;
        const uint32_t w0 = rw[0], w1 = rw[1], w2 = rw[2], w3 = rw[3], w4 = rw[4];
        const uint32_t a0 = __builtin_amdgcn_alignbyte(w1, w0, sh) & icm.x, a1 = __builtin_amdgcn_alignbyte(w2, w1, sh) & icm.y;
        const uint32_t a2 = __builtin_amdgcn_alignbyte(w3, w2, sh) & icm.z, a3 = __builtin_amdgcn_alignbyte(w4, w3, sh) & icm.w;
        const uint32_t rs = __builtin_amdgcn_udot4(a0, 0x01010101u, __builtin_amdgcn_udot4(a1, 0x01010101u,
                            __builtin_amdgcn_udot4(a2, 0x01010101u, __builtin_amdgcn_udot4(a3, 0x01010101u, 0u, false), false), false), false);
        const uint32_t ws = __builtin_amdgcn_udot4(a0, 0x03020100u, __builtin_amdgcn_udot4(a1, 0x07060504u,
                            __builtin_amdgcn_udot4(a2, 0x0B0A0908u, __builtin_amdgcn_udot4(a3, 0x0F0E0D0Cu, 0u, false), false), false), false);
        m10[k] = __mul24(u0, (int)rs) + (int)ws;
        m01[k] = __mul24((lane & 31) - SS_HALF_PATCH, (int)rs);
    }
    float angle[KP], sb_[KP], ca_[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        m10[k] = wave_sum(m10[k]);
        m01[k] = wave_sum(m01[k]);
    }
    {
        /* fastAtan2 and the double-precision sine / cosine are the same for all 64 lanes of a keypoint: lane k evaluates
         * them for keypoint k, so the wave pays for ONE evaluation however many keypoints it walks */
        int vm01 = m01[0], vm10 = m10[0];
#pragma unroll
        for (int k = 1; k < KP; k++) {
            vm01 = lane == k ? m01[k] : vm01;
            vm10 = lane == k ? m10[k] : vm10;
        }
        const float ang_v = ss_fast_atan2((float)vm01, (float)vm10);
        float sin_v, cos_v;
        ss_sincosf_deg(ang_v, &sin_v, &cos_v);
#pragma unroll
        for (int k = 0; k < KP; k++) {
            angle[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ang_v), k));
            sb_[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sin_v), k));
            ca_[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cos_v), k));
        }
    }

    /* steered rBRIEF: the eight sample addresses of a lane first, then the eight byte reads together, then the
     * four ballots (the pattern words were requested before the IC stage) */
    /* cvRound without v_rndne + v_cvt: x + (2^23 + 32) has ulp 1, so the sum is the integer nearest to x (ties to even,
     * like cvRound) and its bits are 0x4B000020 + round(x) for |x| <= 32.  The low 24 bits, 32 + round(x), are what the
     * 24-bit multiplier reads; the biases move into one wave-uniform constant: byte index in the staged window =
     * (18 + row) * OD_PITCH + (kx - wx0) + col */
    constexpr float RN_MAGIC = 8388640.f;
    constexpr int RN_BIAS = 0x4B000020;
    uint32_t off0[KP][4], off1[KP][4];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        const float b = sb_[k], a = ca_[k];
        const uint32_t kbias = (uint32_t)((18 - 32) * OD_PITCH + (kx[k] - wx0[k])) - (uint32_t)RN_BIAS;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t pt = pat[q]; /* x0 y0 x1 y1 as int8 */
            const float x0 = (float)(int8_t)(pt & 0xFF), y0 = (float)(int8_t)((pt >> 8) & 0xFF);
            const float x1 = (float)(int8_t)((pt >> 16) & 0xFF), y1 = (float)(int8_t)(pt >> 24);
            const float fr0 = STEER_FMA ? __fmaf_rn(x0, b, __fmul_rn(y0, a)) : __fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a));
            const float fc0 = STEER_FMA ? __fmaf_rn(x0, a, -__fmul_rn(y0, b)) : __fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b));
            const float fr1 = STEER_FMA ? __fmaf_rn(x1, b, __fmul_rn(y1, a)) : __fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a));
            const float fc1 = STEER_FMA ? __fmaf_rn(x1, a, -__fmul_rn(y1, b)) : __fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b));
            const int r0 = __float_as_int(__fadd_rn(fr0, RN_MAGIC));
            const int c0 = __float_as_int(__fadd_rn(fc0, RN_MAGIC));
            const int r1 = __float_as_int(__fadd_rn(fr1, RN_MAGIC));
            const int c1 = __float_as_int(__fadd_rn(fc1, RN_MAGIC));
            off0[k][q] = (uint32_t)(__mul24(r0, OD_PITCH) + c0) + kbias;
            off1[k][q] = (uint32_t)(__mul24(r1, OD_PITCH) + c1) + kbias;
        }
    }
    int t0[KP][4], t1[KP][4];
#pragma unroll
    for (int k = 0; k < KP; k++) {
        if (OD_SEQ) {
            if (k > 0) wave_sync();
            store_window(k);
            wave_sync();
        }
        const uint8_t *center = (const uint8_t *)&win_all[threadIdx.x >> 6][OD_SEQ ? 0 : k][0];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            t0[k][q] = center[off0[k][q]];
            t1[k][q] = center[off1[k][q]];
        }
    }
#pragma unroll
    for (int k = 0; k < KP; k++) {
        if (k > 0 && slot0 + k >= n_frame) break; /* the duplicate of an odd tail is not written */
        uint64_t words[4];
#pragma unroll
        for (int q = 0; q < 4; q++) words[q] = __ballot(t0[k][q] < t1[k][q]);
        if (desc_x) {
            /* the same 256 bits as the matcher's operand row (fp4_of_4bits): lane L writes nibbles 4 L .. 4 L + 3 = bits
             * 4 L .. 4 L + 3, two bytes: one coalesced 128-byte store per wave */
            const uint64_t wsel = lane < 16 ? words[0] : lane < 32 ? words[1] : lane < 48 ? words[2] : words[3];
            const uint32_t nib = (uint32_t)(wsel >> (4 * (lane & 15))) & 15u;
            *(uint16_t *)(desc_x + ((size_t)frame * kcap + slot[k]) * SS_X_ROW + 2 * lane) = (uint16_t)fp4_of_4bits(nib);
        }
        if (lane == 0) {
            const ss_level &L = g->lv[level[k]];
            uint64_t *d = (uint64_t *)(desc + ((size_t)frame * kcap + slot[k]) * SS_DESC_BYTES);
            d[0] = words[0];
            d[1] = words[1];
            d[2] = words[2];
            d[3] = words[3];
            ss_keypoint kp;
            kp.x = (float)kx[k];
            kp.y = (float)ky[k];
            if (level[k] != 0) {
                kp.x = __fmul_rn(kp.x, L.scale);
                kp.y = __fmul_rn(kp.y, L.scale);
            }
            kp.size = (float)L.scaled_patch;
            kp.angle = angle[k];
            kp.response = (float)resp[k];
            kp.octave = level[k];
            kps[(size_t)frame * kcap + slot[k]] = kp;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* K7: Hamming best / second best.  Lane = one query descriptor (8 dwords in VGPRs); the   */
/* train descriptor of an iteration is wave-uniform, so its 8 dwords arrive by scalar      */
/* loads and feed v_xor / v_bcnt (popcount-accumulate) as SGPR operands: 16 VALU per pair  */
/* + 3 to keep the two smallest keys.  key = distance << 16 | local train index: the       */
/* minimum key is the best distance with the LOWEST index, the second-smallest key carries */
/* the second-best distance (DESIGN.md "match").  A block = 64 queries x one train chunk,   */
/* its 4 waves take quarter-chunks and merge through LDS.                                  */
/* ------------------------------------------------------------------------------------ */
struct match_partial {
    uint16_t d1, d2;
    int32_t j1;
};

__device__ __forceinline__ void merge_partial(int &d1, int &j1, int &d2, int e1, int ej, int e2)
{
    /* (d1, j1, d2) covers lower train indices than (e1, ej, e2); ties keep the lower */
    if (e1 < d1) {
        d2 = imin(d1, e2);
        d1 = e1;
        j1 = ej;
    } else {
        d2 = imin(d2, e1);
    }
}

__global__ __launch_bounds__(256) void k_match(const uint32_t *__restrict__ query, const uint32_t *__restrict__ train,
                                               const int32_t *__restrict__ nq_arr, const int32_t *__restrict__ nt_arr,
                                               int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride,
                                               int train_frame_shift, int chunk_len, int n_chunks, int exclude_self_mode,
                                               int th, int rnum, int rden, int out_stride,
                                               match_partial *__restrict__ partial, int32_t *__restrict__ idx_out,
                                               uint16_t *__restrict__ d1_out, uint16_t *__restrict__ d2_out)
{
    __shared__ int s_d1[4][64], s_j1[4][64], s_d2[4][64];
    const int frame = blockIdx.z, chunk = blockIdx.y;
    /* batch mode: train = frame + shift (clamped to 0); exclude j == i when train == query frame */
    int tframe = frame + train_frame_shift;
    if (tframe < 0) tframe = 0;
    const int nq = nq_arr ? nq_arr[frame] : nq_fixed;
    const int nt = nt_arr ? nt_arr[tframe] : nt_fixed;
    const bool excl = exclude_self_mode == 1 || (exclude_self_mode == 2 && tframe == frame);
    const uint32_t *qf = query + (size_t)frame * q_frame_stride;
    const uint32_t *tf = train + (size_t)tframe * t_frame_stride;
    const int lane = lane_id();
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int qi = blockIdx.x * 64 + lane;
    const bool qvalid = qi < nq;

    uint32_t qw[8];
    {
        const uint4 *p = (const uint4 *)(qf + (size_t)(qvalid ? qi : 0) * 8);
        const uint4 lo = p[0], hi = p[1];
        qw[0] = lo.x; qw[1] = lo.y; qw[2] = lo.z; qw[3] = lo.w;
        qw[4] = hi.x; qw[5] = hi.y; qw[6] = hi.z; qw[7] = hi.w;
    }
    const int c0 = chunk * chunk_len, c1 = imin(c0 + chunk_len, nt);
    const int quarter = (imax(c1 - c0, 0) + 3) >> 2;
    const int t0 = imin(c0 + wave * quarter, c1), t1 = imin(t0 + quarter, c1);

    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    /* 16 VALU per pair for the distance (XOR + popcount-accumulate with SGPR operands) + 3 to form the key and
     * keep the two smallest keys (k1 <= k2 always, so the new second is the median of k1, k2 and the key: one
     * v_med3_u32) + 2 to mask the self pair -- which only the waves whose train rows overlap the block's own
     * 64 query rows have to do (wave-uniform test).  Unrolled by 4 so several s_load_dwordx8 are in flight
     * before the first XOR needs its operand. */
    const bool self_in_range = excl && t0 < (int)(blockIdx.x * 64 + 64) && t1 > (int)(blockIdx.x * 64);
    if (self_in_range) {
        const uint32_t skip = (uint32_t)(qi - t0); /* local index this lane must skip */
#pragma unroll 4
        for (int j = t0; j < t1; j++) {
            const uint32_t *tj = tf + (size_t)j * 8;
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) d += __popc(qw[k] ^ tj[k]);
            const uint32_t jl = (uint32_t)(j - t0);
            uint32_t key = (d << 16) | jl;
            key = jl == skip ? 0xFFFFFFFFu : key;
            k2 = min(max(k1, k2), max(min(k1, k2), key));
            k1 = min(k1, key);
        }
    } else {
#pragma unroll 4
        for (int j = t0; j < t1; j++) {
            const uint32_t *tj = tf + (size_t)j * 8;
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) d += __popc(qw[k] ^ tj[k]);
            const uint32_t key = (d << 16) | (uint32_t)(j - t0);
            k2 = min(max(k1, k2), max(min(k1, k2), key));
            k1 = min(k1, key);
        }
    }
    int d1 = (int)(k1 >> 16), d2 = (int)(k2 >> 16);
    int j1 = d1 == 0xFFFF ? -1 : t0 + (int)(k1 & 0xFFFF);
    s_d1[wave][lane] = d1;
    s_j1[wave][lane] = j1;
    s_d2[wave][lane] = d2;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int wv = 1; wv < 4; wv++) merge_partial(d1, j1, d2, s_d1[wv][lane], s_j1[wv][lane], s_d2[wv][lane]);
    if (n_chunks > 1) {
        if (qi < out_stride) {
            match_partial mp;
            mp.d1 = (uint16_t)d1;
            mp.d2 = (uint16_t)d2;
            mp.j1 = j1;
            partial[((size_t)frame * n_chunks + chunk) * out_stride + qi] = mp;
        }
        return;
    }
    if (qi < out_stride) {
        const size_t o = (size_t)frame * out_stride + qi;
        const bool ok = qvalid && j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
        idx_out[o] = ok ? j1 : -1;
        d1_out[o] = qvalid ? (uint16_t)d1 : (uint16_t)0xFFFF;
        d2_out[o] = qvalid ? (uint16_t)d2 : (uint16_t)0xFFFF;
    }
}

/* ------------------------------------------------------------------------------------ */
/* K7 on the matrix cores.  Hamming distance IS a contraction: with descriptor bits spread to    */
/* bytes, query bit -> +1 / -1 and train bit -> -1 / +1, the i8 dot product of two rows is        */
/* 2 * hamming - 256, exactly, in int32.  v_mfma_i32_32x32x32_i8 does 32 x 32 pairs x 32 bits in   */
/* one instruction, so the 16 XOR + popcount VALU per pair of k_match (profiles/r01_valu_rates.json: */
/* v_bcnt issues at half rate) move to the otherwise idle MFMA pipe and the VALU keeps only the     */
/* three instructions per pair that select best / second best.                                       */
/* Block = 4 waves x NU x 32 queries.  Queries are the B operand (column = lane & 31, resident in */
/* registers for the whole chunk), 32-row train tiles the A operand (expanded once per block into   */
/* LDS through a 256-entry byte -> 8-byte table, rows padded to 272 B so ds_read_b128 is conflict   */
/* free).  C/D layout: lane holds column (query) lane & 31 and train rows (r & 3) + 8 (r >> 2) +      */
/* 4 (lane >> 5) of the tile in registers r = 0..15, so the selection runs per lane over registers  */
/* with the same key = distance << 16 | local row as k_match, and lanes l / l + 32 merge at the end. */
/* Same grid contract, same partial / final outputs as k_match.                                      */
/* ------------------------------------------------------------------------------------ */
#define MM_ROW_BYTES 272
#define MM_TILE 32
/* NU = 32-query B tiles per wave.  1: 117 VGPRs, four waves per SIMD -- as fast alone on the 2000 x 2000 frames and 2 % more
 * frames/s with four batches in flight (it leaves room next to the other kernels).  2: 198 VGPRs, two waves per SIMD, half the
 * LDS reads per MFMA -- 17 % faster on a large database that has the chip to itself (2000 x 20 M: 10.0 vs 11.7 ms). */
#ifndef MM_WAVES
#define MM_WAVES 4 /* waves per block sharing the train tiles (8: -2.5 % frames/s with four batches in flight) */
#endif
#define MM_QBLOCK(NU) (32 * (NU) * MM_WAVES)

/* best / second best per lane, kept as FOUR independent (k1, k2) chains (register groups r >> 2) so the three
 * dependent instructions of one element overlap with those of its neighbours; the chains merge once, at the end */
__device__ __forceinline__ void mm_select(const v16i &acc, uint32_t kb0, uint32_t (&k1)[4], uint32_t (&k2)[4])
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        /* acc = 2 * hamming - 256; (acc + 256) << 15 = hamming << 16, local row in the low half */
        const uint32_t key = ((uint32_t)acc[r] << 15) + (kb0 + (uint32_t)((r & 3) + 8 * (r >> 2)));
        const int c = r >> 2;
        k2[c] = min(max(k1[c], k2[c]), max(min(k1[c], k2[c]), key));
        k1[c] = min(k1[c], key);
    }
}

template <int NU>
__global__ __launch_bounds__(64 * MM_WAVES, NU == 2 ? 2 : 4) void k_match_mfma(const uint32_t *__restrict__ query, const uint32_t *__restrict__ train,
                                                    const int32_t *__restrict__ nq_arr, const int32_t *__restrict__ nt_arr,
                                                    int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride,
                                                    int train_frame_shift, int chunk_len, int n_chunks, int exclude_self_mode,
                                                    int th, int rnum, int rden, int out_stride,
                                                    match_partial *__restrict__ partial, int32_t *__restrict__ idx_out,
                                                    uint16_t *__restrict__ d1_out, uint16_t *__restrict__ d2_out)
{
    __shared__ uint2 lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2][MM_TILE * MM_ROW_BYTES];
    const int frame = blockIdx.z, chunk = blockIdx.y;
    int tframe = frame + train_frame_shift;
    if (tframe < 0) tframe = 0;
    const int nq = nq_arr ? nq_arr[frame] : nq_fixed;
    const int nt = nt_arr ? nt_arr[tframe] : nt_fixed;
    const bool excl = exclude_self_mode == 1 || (exclude_self_mode == 2 && tframe == frame);
    const uint32_t *qf = query + (size_t)frame * q_frame_stride;
    const uint32_t *tf = train + (size_t)tframe * t_frame_stride;
    const int lane = lane_id(), col = lane & 31, half = lane >> 5;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int qbase = blockIdx.x * MM_QBLOCK(NU) + wave * (32 * NU); /* this wave's queries: NU 32-column B tiles */
    const int c0 = chunk * chunk_len, c1 = imin(c0 + chunk_len, nt);
    /* a block whose 256 query rows are all past the frame's count has nothing to select: no tiles */
    const int n_tiles = (int)(blockIdx.x * MM_QBLOCK(NU)) < nq ? (imax(c1 - c0, 0) + MM_TILE - 1) / MM_TILE : 0;

    /* byte -> eight +1 / -1 bytes (bit i of the byte -> byte i): spread the nibble's bits to byte lanes by one
     * multiplication, then 1 -> 0x01, 0 -> 0xFF */
    {
        const uint32_t t = threadIdx.x & 255u;
        const uint32_t lo = ((t & 15u) * 0x204081u) & 0x01010101u, hi = ((t >> 4) * 0x204081u) & 0x01010101u;
        if (threadIdx.x < 256) lut[t] = make_uint2(lo | ((lo ^ 0x01010101u) * 0xFFu), hi | ((hi ^ 0x01010101u) * 0xFFu));
    }
    /* the packed word of the first tile this thread will expand: row t >> 3 of the tile, word t & 7 */
    const bool expander = MM_WAVES == 4 || threadIdx.x < 256; /* the first 256 threads expand the tiles */
    const int erow = (int)((threadIdx.x & 255u) >> 3), eword = (int)(threadIdx.x & 7);
    uint32_t wnext = (expander && n_tiles > 0 && c0 + erow < c1) ? tf[(size_t)(c0 + erow) * 8 + eword] : 0u;
    uint32_t qw[2][8];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int qi = qbase + 32 * u + col;
        const uint4 *p = (const uint4 *)(qf + (size_t)(qi < nq ? qi : 0) * 8);
        const uint4 a = p[0], b = p[1];
        qw[u][0] = a.x; qw[u][1] = a.y; qw[u][2] = a.z; qw[u][3] = a.w;
        qw[u][4] = b.x; qw[u][5] = b.y; qw[u][6] = b.z; qw[u][7] = b.w;
    }
    __syncthreads();
    /* B fragments: k-step s, lane half h <-> bits 32 s + 16 h .. + 15 of the descriptor (the A rows in LDS use the
     * same correspondence, which is all the contraction needs) */
    v4i bq[2][8];
#pragma unroll
    for (int u = 0; u < NU; u++)
#pragma unroll
        for (int sstep = 0; sstep < 8; sstep++) {
            const uint32_t hw = (qw[u][sstep] >> (16 * half)) & 0xFFFFu;
            const uint2 e0 = lut[hw & 0xFFu], e1 = lut[hw >> 8];
            bq[u][sstep] = v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
        }
    auto expand = [&](int buf, uint32_t w) {
        if (!expander) return;
        w = ~w; /* the train side carries the minus sign */
        const uint2 e0 = lut[w & 0xFFu], e1 = lut[(w >> 8) & 0xFFu], e2 = lut[(w >> 16) & 0xFFu], e3 = lut[w >> 24];
        uint4 *dst = (uint4 *)&tiles[buf][erow * MM_ROW_BYTES + eword * 32];
        dst[0] = make_uint4(e0.x, e0.y, e1.x, e1.y);
        dst[1] = make_uint4(e2.x, e2.y, e3.x, e3.y);
    };
    auto load_word = [&](int tile) -> uint32_t {
        const int j = c0 + tile * MM_TILE + erow;
        return (expander && tile < n_tiles && j < c1) ? tf[(size_t)j * 8 + eword] : 0u;
    };
    const bool active = qbase < nq; /* wave-uniform: a wave without valid queries only helps with the tiles */
    const int q_lo = qbase, q_hi = qbase + 32 * NU;
    /* Rows that must not compete -- past the end of the chunk, or the query itself in a self-match -- enter through
     * the accumulator input: C = 2^15 there, which puts their keys above every real key (folded to "none" at the
     * end).  Only the tiles that contain such rows (wave-uniform test) build a C vector; all others start from the
     * inline constant 0, and the selection code is the same straight line for both. */
    auto mfma_tile = [&](int tile, v16i &acc0, v16i &acc1) {
        const int j0 = c0 + tile * MM_TILE;
        const uint8_t *arow = &tiles[tile & 1][col * MM_ROW_BYTES + 16 * half];
        const v4i a0 = *(const v4i *)arow;
        const bool masked = j0 + MM_TILE > c1 || (excl && j0 < q_hi && j0 + MM_TILE > q_lo);
        if (masked) {
            const int rows_valid = c1 - j0, skip0 = excl ? qbase + col - j0 : -1, skip1 = excl ? qbase + 32 + col - j0 : -1;
            v16i ci0, ci1;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 4 * half + (r & 3) + 8 * (r >> 2);
                ci0[r] = (row >= rows_valid || row == skip0) ? 32768 : 0;
                ci1[r] = (row >= rows_valid || row == skip1) ? 32768 : 0;
            }
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0][0], ci0, 0, 0, 0);
            if (NU > 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[1][0], ci1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[0][0], v16i{0}, 0, 0, 0);
            if (NU > 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[1][0], v16i{0}, 0, 0, 0);
        }
    };
    auto mfma_rest = [&](int tile, v16i &acc0, v16i &acc1) {
        const uint8_t *arow = &tiles[tile & 1][col * MM_ROW_BYTES + 16 * half];
#pragma unroll
        for (int sstep = 1; sstep < 8; sstep++) {
            const v4i a = *(const v4i *)(arow + 32 * sstep);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][sstep], acc0, 0, 0, 0);
            if (NU > 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][sstep], acc1, 0, 0, 0);
        }
    };
    uint32_t k1[2][4], k2[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int c = 0; c < 4; c++) k1[u][c] = k2[u][c] = 0xFFFFFFFFu;
    auto select_tile = [&](int i, const v16i &acc0, const v16i &acc1) {
        const uint32_t kb0 = (256u << 15) + (uint32_t)(i * MM_TILE + 4 * half);
        mm_select(acc0, kb0, k1[0], k2[0]);
        if (NU > 1) mm_select(acc1, kb0, k1[1], k2[1]);
    };
    /* software pipeline: the MFMAs of tile i + 1 and the selection from tile i's accumulators sit in one basic block,
     * interleaved (one MFMA, then the VALU that fits in its 32 cycles), while tile i + 2 is expanded into the buffer
     * tile i was read from; one barrier per tile */
    v16i accA0, accA1, accB0, accB1;
    if (n_tiles > 0) expand(0, wnext);
    wnext = load_word(1);
    __syncthreads();
    if (n_tiles > 0 && active) {
        mfma_tile(0, accA0, accA1);
        mfma_rest(0, accA0, accA1);
    }
    if (n_tiles > 1) expand(1, wnext);
    wnext = load_word(2);
    __syncthreads();
    auto phase_mid = [&](int i, const v16i &cur0, const v16i &cur1, v16i &nxt0, v16i &nxt1) {
        if (active) {
            mfma_tile(i + 1, nxt0, nxt1);
            mfma_rest(i + 1, nxt0, nxt1);
            select_tile(i, cur0, cur1);
#ifndef MM_SCHED_VALU
#define MM_SCHED_VALU 9
#endif
#if MM_SCHED_VALU > 0
#pragma unroll
            for (int g = 0; g < 8 * NU - 2; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             /* one MFMA */
                __builtin_amdgcn_sched_group_barrier(0x002, MM_SCHED_VALU, 0); /* nine VALU */
            }
#endif
        }
        if (i + 2 < n_tiles) expand(i & 1, wnext);
        wnext = load_word(i + 3);
        __syncthreads();
    };
    bool done = n_tiles == 0;
    for (int i = 0; i + 1 < n_tiles; i += 2) {
        phase_mid(i, accA0, accA1, accB0, accB1);
        if (i + 2 < n_tiles) {
            phase_mid(i + 1, accB0, accB1, accA0, accA1);
        } else {
            if (active) select_tile(i + 1, accB0, accB1);
            done = true;
        }
    }
    if (!done && active) select_tile(n_tiles - 1, accA0, accA1);
    /* fold the four chains of a lane: (a1 <= a2), (b1 <= b2) -> smallest two of the four */
    uint32_t f1[2], f2[2];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        f1[u] = k1[u][0];
        f2[u] = k2[u][0];
#pragma unroll
        for (int c = 1; c < 4; c++) {
            const uint32_t lo = min(f1[u], k1[u][c]), hi = max(f1[u], k1[u][c]);
            f2[u] = min(hi, min(f2[u], k2[u][c]));
            f1[u] = lo;
        }
    }
    /* lanes l and l + 32 hold the same query over different train rows */
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)f1[u], 32, 64), o2 = (uint32_t)__shfl_xor((int)f2[u], 32, 64);
        uint32_t m1 = min(f1[u], o1), m2 = min(max(f1[u], o1), min(f2[u], o2));
        if (m1 >= 0x20000000u) m1 = 0xFFFFFFFFu; /* only excluded rows were seen */
        if (m2 >= 0x20000000u) m2 = 0xFFFFFFFFu;
        const int qi = qbase + 32 * u + col;
        if (half != 0 || qi >= out_stride) continue;
        const bool qvalid = qi < nq;
        const int d1 = (int)(m1 >> 16), d2 = (int)(m2 >> 16);
        const int j1 = d1 == 0xFFFF ? -1 : c0 + (int)(m1 & 0xFFFFu);
        if (n_chunks > 1) {
            match_partial mp;
            mp.d1 = (uint16_t)d1;
            mp.d2 = (uint16_t)d2;
            mp.j1 = j1;
            partial[((size_t)frame * n_chunks + chunk) * out_stride + qi] = mp;
        } else {
            const size_t o = (size_t)frame * out_stride + qi;
            const bool ok = qvalid && j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
            idx_out[o] = ok ? j1 : -1;
            d1_out[o] = qvalid ? (uint16_t)d1 : (uint16_t)0xFFFF;
            d2_out[o] = qvalid ? (uint16_t)d2 : (uint16_t)0xFFFF;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* K7 on the matrix cores, operands already expanded (k_orient_describe's desc_x / k_expand_desc: one FP4 value per           */
/* descriptor bit, +1 / -1, 128 bytes per row).                                                                                  */
/* * v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands takes the time of the i8 32x32x32 instruction for twice the K           */
/*   (profiles/tools/fp4_probe.hip): 4 instructions per 32 x 32 tile of pairs.  The E8M0 block scale 2^12 on the train           */
/*   operand makes every product +-4096, so the 256-term dot product of a row with a sign-flipped row is                         */
/*   8192 * hamming - 2^20, exact in f32 (|values| < 2^22).                                                                      */
/* * a wave owns TWO 32-query B tiles (64 queries, a block 256): every train fragment read from LDS feeds two MFMAs and a        */
/*   train tile is copied into LDS once per 256 queries.  Round 2 had one tile per wave: its L2 -> LDS stream (32 B per           */
/*   clock and CU at the matrix rate) sat at what the L2 delivers.                                                               */
/* * a 32-row train tile is 4 KB of consecutive bytes that the block copies global -> LDS by LDS-DMA                            */
/*   (global_load_lds_dwordx4: no VGPR, no VALU) MX_NBUF - 1 phases ahead of the MFMAs that read it.  LDS image: rows at a       */
/*   144-byte pitch (ds_read_b128 of 32 rows x 16 bytes is conflict-free there); a DMA instruction writes 64 consecutive         */
/*   16-byte units, so each lane's SOURCE address places the padding (unit u -> row u / 9, piece u % 9; piece 8 = pad).          */
/* * the key is the accumulator: its INPUT is the constant 2^20 + 8160 + row-in-tile, so the MFMA result is                      */
/*   8192 * hamming + 8160 + row as a non-negative float whose bit pattern orders like its value.  Instead of advancing the     */
/*   row field of the NEW keys per tile (round 2: a fifth MFMA), the two RUNNING keys of a lane lose 32 per tile (two            */
/*   v_sub_f32): older rows then always carry the smaller field, which is all the ordering needs; the field is decoded at        */
/*   the end.  Chunks are <= 8192 rows (13-bit field).                                                                            */
/* * selection by GROUPS.  A lane's 16 accumulator registers are, for one query, four runs of four consecutive train rows        */
/*   (aligned blocks of four).  The lane reduces each GROUP of MX_RUNS runs to its minimum (v_min3_u32 / v_min_u32) and keeps the  */
/*   two smallest group minima (v_med3_u32 + v_min_u32 per group): 16 vector instructions per tile and query tile with one-run     */
/*   groups, 10 with one group of all four runs, where the pairwise best / second-best took 32 -- round 2's kernel was bound by    */
/*   vector ISSUE (every MFMA also holds the issue port for 8 cycles), not by the matrix pipe.  What is kept is exact for the      */
/*   best (distance, row) and for the second best OVER THE OTHER GROUPS; the second best inside the best row's own group --         */
/*   3 rows in one 128-byte line of packed descriptors (15 rows in four lines with four-run groups) -- is recomputed once the       */
/*   best is final: in this kernel's epilogue when one chunk covers the train set (FUSED), else by k_match_finish_x after the       */
/*   chunk partials have been folded (second = min(second over the other groups, best of those rows)).  Nothing is approximated.    */
/* * rows that must not compete (past the chunk, or the query itself) get 2^28 added to their keys, in the few tiles that         */
/*   contain such rows (wave-uniform test).                                                                                       */
/* ------------------------------------------------------------------------------------ */
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define MX_PITCH (SS_X_ROW + 16)
#define MX_UNITS (MM_TILE * (MX_PITCH / 16))     /* 288 16-byte units per tile image */
#define MX_DMAS ((MX_UNITS + 63) / 64)           /* 5 DMA instructions per tile (the last one half used) */
#define MX_BUF (MX_DMAS * 1024)
#define MX_ROW_BITS 13
#define MX_FIELD0 8160         /* row field of a new key: 8160 + row in tile; a running key loses 32 per later tile */
#define MX_NONE_F 268435456.0f /* 2^28 */
#define MX_INIT_KEY 0x4E800000u /* 2^30 as a float: "nothing seen"; 2^30 - 32 rounds back to 2^30 */
#define MX_FMT_FP4 4
#define MX_SCALE_ONE 127       /* E8M0: 2^(x - 127) */
/* Two forms of one kernel (template parameters QT, PIPE):
 * compact   QT = 1, PIPE = false: a wave owns one 32-query tile, fragments are read and the selection runs inside the step of
 *           their own tile, LDS ring of three; ~90 registers, five waves per SIMD, blocks of 128 queries.  The form for the
 *           frames of a batch: its blocks are small and short, so they share the CUs with the other batches' kernels.
 * pipelined QT = 2, PIPE = true: two query tiles per wave (every fragment feeds two MFMAs, a train tile is copied into LDS once
 *           per 256 queries), fragments and accumulators double buffered (no LDS latency and no selection between a barrier
 *           and the MFMAs behind it), ring of four; 214 registers, two waves per SIMD.  The form for a large database that has
 *           the chip to itself: 2000 x 20 M rows 5.6 -> 3.8 ms.  In the four-batch pipeline it is 6 % SLOWER than the compact
 *           form although 15 % faster alone: two resident blocks per CU hold most of the register file for 30 us. */
#define MX_QBLOCK_OF(QT) (32 * (QT) * 4)
#define MX_NBUF_OF(PIPE) ((PIPE) ? 4 : 3) /* LDS ring: the DMA of a tile is issued NBUF - 1 steps before the MFMAs that read it */

/* A lane's 16 keys are, for one query, four RUNS of four consecutive train rows (registers 4 k .. 4 k + 3 = rows
 * 8 k + 4 half .. + 3 of the tile).  MX_RUNS consecutive runs form a GROUP: the minimum of each group, then the two smallest
 * group minima seen so far.  Larger groups = fewer vector instructions here (4 runs: 10, 2 runs: 12, 1 run: 16 per tile) and
 * more rows to re-examine once the best is final (15 / 7 / 3 rows in 4 / 2 / 1 lines of packed descriptors). */
#ifndef MX_RUNS
#define MX_RUNS 1 /* measured, 64 x 2000^2 in one fused launch: 1 run 41.8 us, 2 runs 43.2, 4 runs 46.7 (the epilogue's lines decide) */
#endif
__device__ __forceinline__ void mx_select(const v16f &acc, uint32_t &k1, uint32_t &k2)
{
#pragma unroll
    for (int gr = 0; gr < 4 / MX_RUNS; gr++) {
        const int r0 = 4 * MX_RUNS * gr;
        uint32_t m = min(min(__float_as_uint(acc[r0]), __float_as_uint(acc[r0 + 1])), __float_as_uint(acc[r0 + 2])); /* v_min3_u32 */
#pragma unroll
        for (int r = r0 + 3; r + 1 < r0 + 4 * MX_RUNS; r += 2) m = min(min(m, __float_as_uint(acc[r])), __float_as_uint(acc[r + 1]));
        m = min(m, __float_as_uint(acc[r0 + 4 * MX_RUNS - 1]));
        k2 = min(max(k1, k2), max(min(k1, k2), m)); /* v_med3_u32 */
        k1 = min(k1, m);
    }
}
/* first row of the first run of row j's group */
__device__ __forceinline__ int mx_group_base(int j) { return (j & ~31) + 8 * MX_RUNS * ((j & 31) / (8 * MX_RUNS)) + 4 * ((j >> 2) & 1); }

template <int N> __device__ __forceinline__ void mx_wait_vm()
{
    /* all but the N youngest vector-memory operations of this wave have completed (LDS-DMA included) */
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

/* what the FUSED epilogue needs: the packed descriptors of both sides and the outputs */
struct mx_finish {
    const uint8_t *query_p, *train_p; /* packed rows, 32 bytes each */
    int64_t qp_frame_stride, tp_frame_stride;
    int th, rnum, rden;
    int32_t *idx_out;
    uint16_t *d1_out, *d2_out;
};

template <bool FUSED, int MX_QT, bool PIPE> /* FUSED: one chunk covers the train set, the kernel finishes its queries itself */
__global__ __launch_bounds__(256, PIPE ? 2 : 5) void k_match_mfma_x(const uint8_t *__restrict__ query_x, const uint8_t *__restrict__ train_x,
                                                       const int32_t *__restrict__ nq_arr, const int32_t *__restrict__ nt_arr,
                                                       int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride,
                                                       int train_frame_shift, int chunk_len, int n_chunks, int exclude_self_mode,
                                                       int out_stride, match_partial *__restrict__ partial, mx_finish fin)
{
    constexpr int MX_NBUF = MX_NBUF_OF(PIPE), MX_QBLOCK = MX_QBLOCK_OF(MX_QT);
    __shared__ __attribute__((aligned(16))) uint8_t tiles[MX_NBUF][MX_BUF];
    /* 1-D grid, XCD-aware: every XCD gets a contiguous run of (frame, chunk, query block) triples, so the blocks that
     * stream the same train rows share one L2 (dealt round-robin, the blocks of a frame would pull its rows through all
     * eight L2s) */
    const int n_qblocks = (out_stride + MX_QBLOCK - 1) / MX_QBLOCK;
    const int logical = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int qblock = logical % n_qblocks, chunk = (logical / n_qblocks) % n_chunks, frame = logical / (n_qblocks * n_chunks);
    int tframe = frame + train_frame_shift;
    if (tframe < 0) tframe = 0;
    const int nq = nq_arr ? nq_arr[frame] : nq_fixed;
    const int nt = nt_arr ? nt_arr[tframe] : nt_fixed;
    const bool excl = exclude_self_mode == 1 || (exclude_self_mode == 2 && tframe == frame);
    const uint8_t *qf = query_x + (size_t)frame * q_frame_stride; /* strides in bytes: rows of SS_X_ROW */
    const uint8_t *tf = train_x + (size_t)tframe * t_frame_stride;
    const int lane = lane_id(), col = lane & 31, half = lane >> 5;
    const int wave = rfl((int)(threadIdx.x >> 6));
    const int qbase = qblock * MX_QBLOCK + wave * (32 * MX_QT); /* this wave's 64 queries: two B tiles */
    const int c0 = chunk * chunk_len, c1 = imin(c0 + chunk_len, nt);
    const int n_tiles = qblock * MX_QBLOCK < nq ? (imax(c1 - c0, 0) + MM_TILE - 1) / MM_TILE : 0;

    /* source offsets of this lane's pieces inside a tile (the same for every tile): DMA m of the tile covers units
     * 64 m .. 64 m + 63; wave w issues m = w (and m = 4 for wave 0) */
    uint32_t src_off[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int u = imin(64 * (wave + 4 * k) + lane, MX_UNITS - 1);
        const int r = u / (MX_PITCH / 16), j = imin(u - r * (MX_PITCH / 16), SS_X_ROW / 16 - 1);
        src_off[k] = (uint32_t)(r * SS_X_ROW + j * 16);
    }
    const uint8_t *tsrc = tf + (size_t)c0 * SS_X_ROW; /* first row of the chunk */
    const int last_tile = imax(n_tiles - 1, 0);
    /* every wave issues its DMAs for EVERY tile slot of the ring, also past the last tile (a harmless re-read of the
     * last tile): the count of vector-memory operations in flight is then the same in every phase, which is what the
     * counted waits below rely on.  `slot_off` = byte offset of the ring slot (wave-uniform). */
    auto dma_tile = [&](int tile, int slot_off, auto ndma_c) {
        constexpr int NDMA = decltype(ndma_c)::value;
#if defined(MX_EXP) && MX_EXP == 1 /* timing experiment: no DMA after the prologue (stale tiles: wrong results) */
        if (tile >= MX_NBUF - 1) return;
#endif
        const uint8_t *src = tsrc + (size_t)(uint32_t)(imin(tile, last_tile) * (MM_TILE * SS_X_ROW));
        uint8_t *dst = &tiles[0][0] + slot_off + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + src_off[0]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        if (NDMA == 2)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + src_off[1]),
                                             (__attribute__((address_space(3))) void *)(dst + 4096), 16, 0, 0);
    };
    /* B fragments of query tile u: k-step s, lane half h <-> bytes 32 s + 16 h .. + 15 of the query row (bits 64 s + 32 h .. + 31),
     * sign flipped */
    v4i bq4[MX_QT][4];
#pragma unroll
    for (int u = 0; u < MX_QT; u++) {
        const int qi = qbase + 32 * u + col;
        const uint8_t *qrow = qf + (size_t)(qi < nq ? qi : 0) * SS_X_ROW + 16 * half;
#pragma unroll
        for (int sstep = 0; sstep < 4; sstep++) bq4[u][sstep] = *(const v4i *)(qrow + 32 * sstep);
    }
#if defined(MX_EXP) && MX_EXP == 3 /* timing experiment: the tile stream alone, no MFMA, no selection */
    const bool active = false;
#else
    const bool active = qbase < nq; /* wave-uniform: a wave without valid queries only helps with the tiles */
#endif
    uint32_t k1[MX_QT], k2[MX_QT];
#pragma unroll
    for (int u = 0; u < MX_QT; u++) k1[u] = k2[u] = MX_INIT_KEY;

    /* The loop, in one version per wave role (DMAs per tile: wave 0 issues two, the others one; a wave without valid
     * queries only streams) so that nothing role-dependent is tested inside it.
     * Step t: { wait: this wave's DMAs of tile t + 1 have landed (those of the MX_NBUF - 3 younger tiles may still fly) |
     * barrier: so have everybody's | DMA of tile t + MX_NBUF - 1 into the slot tile t - 1 had (its last reader finished a
     * step ago) | 4 x 2 MFMAs on tile t, whose fragments are in registers since the previous step | LDS reads of tile t + 1's
     * fragments | between the MFMAs: the selection of tile t - 1 }.  Fragments and accumulators are double buffered, so no
     * LDS latency and no selection stands between a barrier and the MFMAs behind it: with the reads and the selection inside
     * the step of their own tile the waves of a CU fell into step and the matrix pipe idled through both.
     * An LDS-DMA is a pending LDS write that only the issuing wave's vmcnt tracks; hipcc does not always count it when it
     * places the waits of __syncthreads() (one loop barrier came out with lgkmcnt(0) only), hence the explicit counted waits
     * and the raw barrier. */
    auto run = [&](auto ndma_c, auto active_c) {
        constexpr int NDMA = decltype(ndma_c)::value;
        constexpr bool ACTIVE = decltype(active_c)::value;
#pragma unroll
        for (int t = 0; t < MX_NBUF - 1; t++) dma_tile(t, t * MX_BUF, ndma_c);
        int wr_off = (MX_NBUF - 1) * MX_BUF;
        auto sync_step = [&](int t) {
            mx_wait_vm<NDMA * (MX_NBUF - 2 - (PIPE ? 1 : 0))>(); /* PIPE: tile t + 1 has landed; else: tile t */
            /* every LDS read this wave has issued is complete before it arrives: the compiler otherwise lets the last fragment
             * read of a tile (and the MFMA behind it) sink below the barrier, and the DMA another wave issues right after the
             * barrier refills exactly that slot */
#if !(defined(MX_EXP) && MX_EXP == 7) /* experiment 7: the hazard itself (profiles/tools/repro_partial.py shows the wrong partials) */
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#if !(defined(MX_EXP) && MX_EXP == 5) /* timing experiment 5: no barrier (races: wrong results) */
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            dma_tile(t + MX_NBUF - 1, wr_off, ndma_c);
            wr_off = wr_off == (MX_NBUF - 1) * MX_BUF ? 0 : wr_off + MX_BUF;
        };
        if (PIPE) { /* tile 0 has landed everywhere before anybody reads it */
            mx_wait_vm<NDMA * (MX_NBUF - 2)>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (!ACTIVE) {
            for (int t = 0; t < n_tiles; t++) sync_step(t);
            return;
        }
        v8i bq[MX_QT][4];
        v16f crow; /* accumulator input of every tile: 2^20 + 8160 + row in the tile */
#pragma unroll
        for (int u = 0; u < MX_QT; u++)
#pragma unroll
            for (int sstep = 0; sstep < 4; sstep++) {
                const v4i f = bq4[u][sstep] ^ (int)0x88888888;
                bq[u][sstep] = v8i{f[0], f[1], f[2], f[3], 0, 0, 0, 0};
            }
#pragma unroll
        for (int r = 0; r < 16; r++) crow[r] = (float)((1 << 20) + MX_FIELD0 + 4 * half + (r & 3) + 8 * (r >> 2));
        /* tiles whose keys need masking: the last one when the chunk does not end on a tile, and the (up to two) tiles that
         * hold this wave's own 64 rows in a self-match.  The tile range is cut into plain and masked SEGMENTS so that the
         * plain loop tests nothing: A = [0, e1) plain, B = [e1, e2) masked, C = [e2, e3) plain, D = [e3, n) masked. */
        const int p_part = ((c1 - c0) & (MM_TILE - 1)) ? n_tiles - 1 : n_tiles;
        const bool self_here = excl && qbase + 32 * MX_QT > c0 && qbase < c1;
        const int m0 = self_here ? imin(imax(qbase - c0, 0) / MM_TILE, n_tiles) : n_tiles;
        const int m1 = self_here ? imin((qbase + 32 * MX_QT - c0) / MM_TILE, n_tiles) : n_tiles;
        const int e1 = imin(m0, p_part), e2 = m0 <= p_part ? m1 : n_tiles, e3 = imax(e2, p_part);
        const uint32_t a_base = (uint32_t)(col * MX_PITCH + 16 * half);
        int rd_off = 0;
        v4i afr[PIPE ? 2 : 1][4]; /* A fragments of the tile of this step (PIPE: parity P, and of the next one) */
        v16f acc[PIPE ? 2 : 1][MX_QT];
        auto read_frags = [&](v4i(&a4)[4]) {
            const uint8_t *arow = &tiles[0][0] + (a_base + (uint32_t)rd_off);
            rd_off = rd_off == (MX_NBUF - 1) * MX_BUF ? 0 : rd_off + MX_BUF;
#if defined(MX_EXP) && MX_EXP == 6 /* timing experiment 6: no LDS reads (wrong results) */
#pragma unroll
            for (int sstep = 0; sstep < 4; sstep++) a4[sstep] = bq4[0][sstep] + (int)(uintptr_t)arow;
#else
#pragma unroll
            for (int sstep = 0; sstep < 4; sstep++) a4[sstep] = *(const v4i *)(arow + 32 * sstep);
#endif
        };
        /* the MFMAs of tile t.  PIPE: the fragments are in registers (a4); compact: each k-step's fragment is read from the
         * tile's LDS slot right before its MFMAs (four registers live instead of sixteen) */
        auto mfma_tile = [&](int t, const v4i(&a4)[4], v16f(&ac)[MX_QT], auto masked_c) {
            constexpr bool MASKED = decltype(masked_c)::value;
            const uint8_t *arow = nullptr;
            if (!PIPE) {
                arow = &tiles[0][0] + (a_base + (uint32_t)rd_off);
                rd_off = rd_off == (MX_NBUF - 1) * MX_BUF ? 0 : rd_off + MX_BUF;
            }
#ifdef MX_SETPRIO /* timing experiment: the wave asks for issue priority while its matrix instructions go out */
            __builtin_amdgcn_s_setprio(MX_SETPRIO);
#endif
#pragma unroll
            for (int sstep = 0; sstep < 4; sstep++) {
                const v4i f = PIPE ? a4[sstep] : *(const v4i *)(arow + 32 * sstep);
                const v8i a = v8i{f[0], f[1], f[2], f[3], 0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < MX_QT; u++) {
                    v16f ci = sstep == 0 ? crow : ac[u];
                    if (MASKED && sstep == 0) { /* rows that must not compete enter with 2^28 */
                        const int j0 = c0 + t * MM_TILE, rows_valid = c1 - j0, skip = excl ? qbase + 32 * u + col - j0 : -1;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int row = 4 * half + (r & 3) + 8 * (r >> 2);
                            ci[r] += (row >= rows_valid || row == skip) ? MX_NONE_F : 0.0f;
                        }
                    }
                    ac[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bq[u][sstep], ci, MX_FMT_FP4, MX_FMT_FP4, 0, MX_SCALE_ONE + 12, 0, MX_SCALE_ONE);
                }
            }
#ifdef MX_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto select_tile = [&](v16f(&ac)[MX_QT]) {
#if defined(MX_EXP) && MX_EXP == 4 /* timing experiment 4: no selection (one key per tile keeps the MFMAs alive) */
            k1[0] = min(k1[0], __float_as_uint(ac[0][0]));
            k1[1] = min(k1[1], __float_as_uint(ac[1][0]));
            return;
#endif
            /* the running keys fall behind every key of this tile */
#pragma unroll
            for (int u = 0; u < MX_QT; u++) {
                k1[u] = __float_as_uint(__uint_as_float(k1[u]) - 32.0f);
                k2[u] = __float_as_uint(__uint_as_float(k2[u]) - 32.0f);
            }
#pragma unroll
            for (int u = 0; u < MX_QT; u++) mx_select(ac[u], k1[u], k2[u]);
        };
        /* step t: P = t & 1 names the buffers of tile t */
        auto step = [&](int t, auto p_c, auto masked_c, auto first_c) {
            constexpr int P = decltype(p_c)::value;
            sync_step(t);
            if (PIPE) {
                mfma_tile(t, afr[P], acc[P], masked_c);
                read_frags(afr[1 - P]); /* tile t + 1 (past the end: a landed slot, never used) */
                if (!decltype(first_c)::value) select_tile(acc[1 - P]);
            } else { /* compact: everything of tile t inside its own step */
                mfma_tile(t, afr[0], acc[0], masked_c);
                select_tile(acc[0]);
            }
        };
        auto segment = [&](int b, int e, auto masked_c) {
            int t = b;
            if (t < e && (t & 1)) {
                step(t, std::integral_constant<int, 1>{}, masked_c, std::false_type{});
                t++;
            }
            for (; t + 1 < e; t += 2) {
                step(t, std::integral_constant<int, 0>{}, masked_c, std::false_type{});
                step(t + 1, std::integral_constant<int, 1>{}, masked_c, std::false_type{});
            }
            if (t < e) step(t, std::integral_constant<int, 0>{}, masked_c, std::false_type{});
        };
        if (PIPE) read_frags(afr[0]);
        /* tile 0 has no predecessor to select */
        if (e1 > 0) step(0, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{});
        else step(0, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{});
        segment(1, e1, std::false_type{});
        segment(imax(e1, 1), e2, std::true_type{});
        segment(imax(e2, 1), e3, std::false_type{});
        segment(imax(e3, 1), n_tiles, std::true_type{});
        if (PIPE) {
            if (n_tiles & 1) select_tile(acc[0]);
            else select_tile(acc[PIPE ? 1 : 0]);
        }
    };
    if (n_tiles > 0) {
        if (active) {
            if (wave == 0) run(std::integral_constant<int, 2>{}, std::true_type{});
            else run(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            if (wave == 0) run(std::integral_constant<int, 2>{}, std::false_type{});
            else run(std::integral_constant<int, 1>{}, std::false_type{});
        }
    }
    mx_wait_vm<0>(); /* nothing of this block may still be writing LDS when the block retires */
    /* lanes l and l + 32 hold the same queries over the two groups of every tile; keys are bit patterns of non-negative
     * floats 8192 * hamming + field (< 2^22), or >= 2^28 for "none"; a key born in tile t has lost 32 for each of the
     * n_tiles - 1 - t later tiles */
    int r_d1[MX_QT], r_d2[MX_QT], r_j1[MX_QT];
#pragma unroll
    for (int u = 0; u < MX_QT; u++) {
        const uint32_t o1 = (uint32_t)__shfl_xor((int)k1[u], 32, 64), o2 = (uint32_t)__shfl_xor((int)k2[u], 32, 64);
        const uint32_t b1 = min(k1[u], o1), b2 = min(max(k1[u], o1), min(k2[u], o2));
        const bool none1 = b1 >= __float_as_uint(MX_NONE_F), none2 = b2 >= __float_as_uint(MX_NONE_F);
        const uint32_t m1 = none1 ? 0u : (uint32_t)__uint_as_float(b1), m2 = none2 ? 0u : (uint32_t)__uint_as_float(b2);
        r_d1[u] = none1 ? 0xFFFF : (int)(m1 >> MX_ROW_BITS); /* only excluded rows were seen: none */
        r_d2[u] = none2 ? 0xFFFF : (int)(m2 >> MX_ROW_BITS);
        r_j1[u] = none1 ? -1 : c0 + (int)(m1 & ((1u << MX_ROW_BITS) - 1)) - MX_FIELD0 + MM_TILE * (n_tiles - 1);
    }
    if (!FUSED) {
#pragma unroll
        for (int u = 0; u < MX_QT; u++) {
            const int qi = qbase + 32 * u + col;
            if (half != 0 || qi >= out_stride) continue;
            match_partial mp;
            mp.d1 = (uint16_t)r_d1[u];
            mp.d2 = (uint16_t)r_d2[u];
            mp.j1 = r_j1[u];
            partial[((size_t)frame * n_chunks + chunk) * out_stride + qi] = mp;
        }
        return;
    }
    /* FUSED: the best is final.  Eight passes of eight queries; in a pass the eight lanes 8 s .. 8 s + 7 serve one query of
     * the wave (query tile u = pass >> 2, column 8 (pass & 3) + s): lane g reads piece g (16 bytes: row g >> 1, half g & 1) of
     * each of the four 128-byte runs of the best row's group (the eight read whole lines), popcounts it against its half of the
     * query row, and the minimum over the group's other rows completes the second best.  Lane 8 s writes the query's outputs. */
    const int g = lane & 7, slot = lane >> 3;
    const uint8_t *qp = fin.query_p + (size_t)frame * fin.qp_frame_stride;
    const uint8_t *tp = fin.train_p + (size_t)tframe * fin.tp_frame_stride;
    /* a round of four passes per query tile: the loads of a round are issued together (one memory round trip per round, not per pass) */
#pragma unroll
    for (int round = 0; round < MX_QT; round++) {
        int pj1[4], pd1[4], pd2[4];
        uint4 pa[4], pb[4][MX_RUNS];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int src = 8 * k + slot; /* u = round is uniform in a pass */
            pj1[k] = __shfl(r_j1[round], src, 64);
            pd1[k] = __shfl(r_d1[round], src, 64);
            pd2[k] = __shfl(r_d2[round], src, 64);
            const int qi = qbase + 32 * round + src;
            const bool live = qi < nq && pj1[k] >= 0;
            const int base = live ? mx_group_base(pj1[k]) : 0;
            pa[k] = *(const uint4 *)(qp + (size_t)(live ? qi : 0) * SS_DESC_BYTES + 16 * (g & 1));
#pragma unroll
            for (int run = 0; run < MX_RUNS; run++) {
                const int row = base + 8 * run + (g >> 1);
                /* rows past the train set are not allocated everywhere: read the query piece's address again instead */
                pb[k][run] = live && row < nt ? *(const uint4 *)(tp + (size_t)row * SS_DESC_BYTES + 16 * (g & 1)) : pa[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int src = 8 * k + slot, qi = qbase + 32 * round + src;
            const bool qvalid = qi < nq, live = qvalid && pj1[k] >= 0;
            const int base = mx_group_base(pj1[k]);
            int cand = 0xFFFF;
#pragma unroll
            for (int run = 0; run < MX_RUNS; run++) {
                const int row = base + 8 * run + (g >> 1);
                const uint4 a = pa[k], b = pb[k][run];
                int h = __builtin_popcount(a.x ^ b.x) + __builtin_popcount(a.y ^ b.y) + __builtin_popcount(a.z ^ b.z) + __builtin_popcount(a.w ^ b.w);
                h += __builtin_amdgcn_update_dpp(0, h, 0xB1, 0xF, 0xF, false); /* quad_perm [1,0,3,2]: the other half of the row */
                if (live && row < nt && row != pj1[k] && !(excl && row == qi)) cand = imin(cand, h);
            }
            cand = imin(cand, __builtin_amdgcn_update_dpp(cand, cand, 0x4E, 0xF, 0xF, false));  /* quad_perm [2,3,0,1]: the other row of the quad */
            cand = imin(cand, __builtin_amdgcn_update_dpp(cand, cand, 0x141, 0xF, 0xF, false)); /* row_half_mirror: the other quad */
            const int d1 = pd1[k], d2 = imin(pd2[k], cand);
            if (g == 0 && qi < out_stride) {
                const size_t o = (size_t)frame * out_stride + qi;
                const bool ok = live && (fin.th < 0 || (d1 <= fin.th && d1 * fin.rden < d2 * fin.rnum));
                fin.idx_out[o] = ok ? pj1[k] : -1;
                fin.d1_out[o] = qvalid ? (uint16_t)d1 : (uint16_t)0xFFFF;
                fin.d2_out[o] = qvalid ? (uint16_t)d2 : (uint16_t)0xFFFF;
            }
        }
    }
}

/* The second half of k_match_mfma_x when the train set was matched in several chunks: folds the chunk partials of a query
 * (n_chunks >= 1; 0 = they have been folded into the output arrays already, by k_match_merge_wide in raw mode), then
 * recomputes the distances of the query to the other rows of the best row's GROUP (mx_select: MX_RUNS runs of four consecutive
 * rows) -- the only rows the partials' second best does not cover -- and applies the acceptance test.
 * Four lanes per query, lane g < MX_RUNS with run g of the group (4 ROWB contiguous bytes).  The rows are read
 * as packed descriptors where the caller has them (32 bytes: a quarter of the traffic), else as the matcher's operand rows:
 * 128 bytes of FP4 +1 (0x2) / -1 (0xA) nibbles, two rows differ in a bit where the nibbles' sign bits differ. */
template <int ROWB>
__global__ __launch_bounds__(256) void k_match_finish_x(const uint8_t *__restrict__ query_x, const uint8_t *__restrict__ train_x,
                                                        const int32_t *__restrict__ nq_arr, const int32_t *__restrict__ nt_arr, int nq_fixed,
                                                        int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride, int train_frame_shift,
                                                        int exclude_self_mode, const match_partial *__restrict__ partial, int n_chunks, int th,
                                                        int rnum, int rden, int out_stride, int32_t *__restrict__ idx_out,
                                                        uint16_t *__restrict__ d1_out, uint16_t *__restrict__ d2_out)
{
    const int frame = blockIdx.y, g = threadIdx.x & 3;
    const int qi = blockIdx.x * 64 + (threadIdx.x >> 2);
    if (qi >= out_stride) return; /* a whole quad leaves together */
    int tframe = frame + train_frame_shift;
    if (tframe < 0) tframe = 0;
    const int nq = nq_arr ? nq_arr[frame] : nq_fixed;
    const int nt = nt_arr ? nt_arr[tframe] : nt_fixed;
    const bool excl = exclude_self_mode == 1 || (exclude_self_mode == 2 && tframe == frame);
    const size_t o = (size_t)frame * out_stride + qi;
    int d1 = 0xFFFF, d2 = 0xFFFF, j1 = -1;
    if (n_chunks == 0) {
        d1 = d1_out[o];
        d2 = d2_out[o];
        j1 = idx_out[o];
    } else {
        for (int c = 0; c < n_chunks; c++) {
            const match_partial mp = partial[((size_t)frame * n_chunks + c) * out_stride + qi];
            merge_partial(d1, j1, d2, mp.d1, mp.j1, mp.d2);
        }
    }
    const bool qvalid = qi < nq;
    int cand = 0xFFFF;
    if (qvalid && j1 >= 0) {
        const int base = mx_group_base(j1) + 8 * g; /* rows base .. base + 3: this lane's run (lanes g >= MX_RUNS have none) */
        const uint4 *q = (const uint4 *)(query_x + (size_t)frame * q_frame_stride + (size_t)qi * ROWB);
        const uint4 *t = (const uint4 *)(train_x + (size_t)tframe * t_frame_stride + (size_t)base * ROWB);
        const uint32_t mask = ROWB == SS_X_ROW ? 0x88888888u : 0xFFFFFFFFu;
        uint4 qa[ROWB / 16];
#pragma unroll
        for (int k = 0; k < ROWB / 16; k++) qa[k] = q[k];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = base + r;
            if (row >= nt || g >= MX_RUNS) break; /* rows past the train set are not allocated everywhere */
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < ROWB / 16; k++) {
                const uint4 b = t[r * (ROWB / 16) + k];
                h = __builtin_popcount((qa[k].x ^ b.x) & mask) + h;
                h = __builtin_popcount((qa[k].y ^ b.y) & mask) + h;
                h = __builtin_popcount((qa[k].z ^ b.z) & mask) + h;
                h = __builtin_popcount((qa[k].w ^ b.w) & mask) + h;
            }
            if (row != j1 && !(excl && row == qi)) cand = imin(cand, (int)h);
        }
    }
    cand = imin(cand, __builtin_amdgcn_update_dpp(cand, cand, 0xB1, 0xF, 0xF, false)); /* quad_perm [1,0,3,2] */
    cand = imin(cand, __builtin_amdgcn_update_dpp(cand, cand, 0x4E, 0xF, 0xF, false)); /* quad_perm [2,3,0,1] */
    d2 = imin(d2, cand);
    if (g != 0) return;
    const bool ok = qvalid && j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
    idx_out[o] = ok ? j1 : -1;
    d1_out[o] = qvalid ? (uint16_t)d1 : (uint16_t)0xFFFF;
    d2_out[o] = qvalid ? (uint16_t)d2 : (uint16_t)0xFFFF;
}

__global__ __launch_bounds__(256) void k_match_merge(const match_partial *__restrict__ partial, const int32_t *__restrict__ nq_arr,
                                                     int nq_fixed, int n_chunks, int th, int rnum, int rden, int out_stride,
                                                     int32_t *__restrict__ idx_out, uint16_t *__restrict__ d1_out,
                                                     uint16_t *__restrict__ d2_out)
{
    const int frame = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= out_stride) return;
    const int nq = nq_arr ? nq_arr[frame] : nq_fixed;
    int d1 = 0xFFFF, d2 = 0xFFFF, j1 = -1;
    for (int c = 0; c < n_chunks; c++) {
        const match_partial mp = partial[((size_t)frame * n_chunks + c) * out_stride + qi];
        merge_partial(d1, j1, d2, mp.d1, mp.j1, mp.d2);
    }
    const size_t o = (size_t)frame * out_stride + qi;
    const bool qvalid = qi < nq;
    const bool ok = qvalid && j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
    idx_out[o] = ok ? j1 : -1;
    d1_out[o] = qvalid ? (uint16_t)d1 : (uint16_t)0xFFFF;
    d2_out[o] = qvalid ? (uint16_t)d2 : (uint16_t)0xFFFF;
}

/* packed descriptors (32 B) -> the matrix-core matcher's operand rows (SS_X_ROW bytes, fp4_of_4bits), the same format
 * k_orient_describe writes for the frames of a batch.  One wave per row, one coalesced 128-byte store; rows n .. n_alloc - 1
 * (n_alloc = n rounded up to the 32-row tile) are zero-filled (FP4 zeros: they contribute 0 and are masked anyway). */
__global__ __launch_bounds__(256) void k_expand_desc(const uint32_t *__restrict__ packed, int n, int n_alloc, uint8_t *__restrict__ out,
                                                     int64_t src_frame_words, int64_t dst_frame_bytes)
{
    /* blockIdx.y = frame of a batch ([frames][n][32] packed -> [frames][n_alloc][SS_X_ROW]); a single set has one frame */
    const int row = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = lane_id();
    if (row >= n_alloc) return;
    packed += (size_t)blockIdx.y * src_frame_words;
    out += (size_t)blockIdx.y * dst_frame_bytes;
    uint32_t v = 0;
    if (row < n) {
        const uint32_t w = packed[(size_t)row * 8 + (lane >> 3)];
        v = fp4_of_4bits((w >> (4 * (lane & 7))) & 15u);
    }
    *(uint16_t *)(out + (size_t)row * SS_X_ROW + 2 * lane) = (uint16_t)v;
}

/* raw local match of a database shard -> the 8-byte records ranks exchange (include/sendslam_orb.h ss_match_part) */
static_assert(sizeof(match_partial) == sizeof(ss_match_part) && offsetof(match_partial, j1) == offsetof(ss_match_part, row),
              "the chunk partial IS the cross-shard record");
__global__ __launch_bounds__(256) void k_pack_partial(const int32_t *__restrict__ idx, const uint16_t *__restrict__ d1,
                                                      const uint16_t *__restrict__ d2, int n, int32_t row_offset,
                                                      match_partial *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    match_partial mp;
    mp.d1 = d1[i];
    mp.d2 = d2[i];
    mp.j1 = idx[i] < 0 ? -1 : idx[i] + row_offset;
    out[i] = mp;
}

/* Same fold for MANY chunks (large databases): one wave per query, lanes stride over the chunks,
 * then a cross-lane fold on the 64-bit key (distance, global row) -- a lower row wins a tie, the
 * second best is the smallest of all second distances and the losing first distances. */
__global__ __launch_bounds__(64) void k_match_merge_wide(const match_partial *__restrict__ partial, int nq, int n_chunks, int th,
                                                         int rnum, int rden, int out_stride, int32_t *__restrict__ idx_out,
                                                         uint16_t *__restrict__ d1_out, uint16_t *__restrict__ d2_out)
{
    const int qi = blockIdx.x, lane = lane_id();
    uint64_t best = ~0ull;
    uint32_t second = 0xFFFFu;
    for (int c = lane; c < n_chunks; c += WAVE) {
        const match_partial mp = partial[(size_t)c * out_stride + qi];
        const uint64_t k = mp.j1 < 0 ? ~0ull : ((uint64_t)mp.d1 << 32) | (uint32_t)mp.j1;
        const uint64_t loser = k < best ? best : k;
        best = k < best ? k : best;
        const uint32_t ld = loser == ~0ull ? 0xFFFFu : (uint32_t)(loser >> 32);
        second = min(min(second, (uint32_t)mp.d2), ld);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t ob = __shfl_xor(best, o, 64);
        const uint32_t os = (uint32_t)__shfl_xor((int)second, o, 64);
        const uint64_t loser = ob < best ? best : ob;
        best = ob < best ? ob : best;
        const uint32_t ld = loser == ~0ull ? 0xFFFFu : (uint32_t)(loser >> 32);
        second = min(min(second, os), ld);
    }
    if (lane == 0) {
        const int d1 = best == ~0ull ? 0xFFFF : (int)(best >> 32), d2 = (int)second;
        const int j1 = best == ~0ull ? -1 : (int)(uint32_t)best;
        const bool ok = qi < nq && j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
        idx_out[qi] = ok ? j1 : -1;
        d1_out[qi] = (uint16_t)d1;
        d2_out[qi] = (uint16_t)d2;
    }
}

/* ------------------------------------------------------------------------------------ */
/* K7, database-streaming form (loop closure / relocalisation with a HANDFUL of query         */
/* descriptors against millions of keyframe descriptors: BASELINE.json config 5 in the regime   */
/* SURVEY.md section 8(d) calls HBM-bound).  Roles are swapped: a lane owns a TRAIN row (two    */
/* coalesced 16-B loads per row, the database is read exactly once), the <= 8 queries are       */
/* wave-uniform and sit in SGPRs.  19 VALU per (query, row): at 8 queries the integer pipe      */
/* still keeps up with ~5 TB/s of rows.  Each block owns a contiguous chunk so that partials    */
/* stay ordered by index and the ordinary k_match_merge folds them.                             */
/* ------------------------------------------------------------------------------------ */
template <int NQ>
__global__ __launch_bounds__(256) void k_match_stream(const uint32_t *__restrict__ query, const uint4 *__restrict__ train,
                                                      int nq, int nt, int chunk_len, int n_chunks,
                                                      match_partial *__restrict__ partial)
{
    __shared__ uint64_t s_k1[4][NQ];
    __shared__ uint32_t s_d2[4][NQ];
    const int chunk = blockIdx.x;
    const int c0 = chunk * chunk_len, c1 = imin(c0 + chunk_len, nt);
    uint32_t k1[NQ], k2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) k1[q] = k2[q] = 0xFFFFFFFFu;
    /* key = distance << 22 | iteration: per lane, iterations visit ascending rows, so the
     * smallest key is the lane's best distance at its lowest row */
    /* two rows per trip: four 16-B loads are in flight before the first XOR needs its operand */
    auto score_row = [&](const uint4 lo, const uint4 hi, uint32_t it) {
        const uint32_t tw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t *qp = query + (size_t)imin(q, nq - 1) * 8; /* uniform: scalar loads */
            uint32_t d = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) d += __popc(tw[k] ^ qp[k]);
            const uint32_t key = (d << 22) | it;
            k2[q] = min(k2[q], max(k1[q], key));
            k1[q] = min(k1[q], key);
        }
    };
    uint32_t it = 0;
    int j = c0 + (int)threadIdx.x;
    for (; j + 256 < c1; j += 512, it += 2) {
        const uint4 lo0 = train[(size_t)j * 2], hi0 = train[(size_t)j * 2 + 1];
        const uint4 lo1 = train[(size_t)(j + 256) * 2], hi1 = train[(size_t)(j + 256) * 2 + 1];
        score_row(lo0, hi0, it);
        score_row(lo1, hi1, it + 1);
    }
    if (j < c1) score_row(train[(size_t)j * 2], train[(size_t)j * 2 + 1], it);
    /* fold the 256 lanes: compare (distance, global row) as one 64-bit key; the second best is
     * the smallest distance among every lane's second key and the losing lanes' first keys */
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t d1 = k1[q] >> 22;
        const uint32_t row = (uint32_t)c0 + (k1[q] & 0x3FFFFFu) * 256u + threadIdx.x;
        uint64_t best = k1[q] == 0xFFFFFFFFu ? ~0ull : ((uint64_t)d1 << 32) | row;
        uint32_t second = k2[q] == 0xFFFFFFFFu ? 0xFFFFu : k2[q] >> 22;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t ob = __shfl_xor(best, o, 64);
            const uint32_t os = (uint32_t)__shfl_xor((int)second, o, 64);
            const uint64_t loser = ob < best ? best : ob;
            best = ob < best ? ob : best;
            const uint32_t ld = loser == ~0ull ? 0xFFFFu : (uint32_t)(loser >> 32);
            second = min(min(second, os), ld);
        }
        if (lane == 0) {
            s_k1[wave][q] = best;
            s_d2[wave][q] = second;
        }
    }
    __syncthreads();
    if (threadIdx.x < NQ && (int)threadIdx.x < nq) {
        const int q = threadIdx.x;
        uint64_t best = s_k1[0][q];
        uint32_t second = s_d2[0][q];
        for (int wv = 1; wv < 4; wv++) {
            const uint64_t ob = s_k1[wv][q];
            const uint64_t loser = ob < best ? best : ob;
            best = ob < best ? ob : best;
            const uint32_t ld = loser == ~0ull ? 0xFFFFu : (uint32_t)(loser >> 32);
            second = min(min(second, s_d2[wv][q]), ld);
        }
        match_partial mp;
        mp.d1 = best == ~0ull ? (uint16_t)0xFFFF : (uint16_t)(best >> 32);
        mp.d2 = (uint16_t)second;
        mp.j1 = best == ~0ull ? -1 : (int32_t)(uint32_t)best;
        partial[(size_t)chunk * nq + q] = mp;
    }
}

} // namespace

/* ------------------------------------------------------------------------------------ */
/* launch wrappers (host)                                                                */
/* ------------------------------------------------------------------------------------ */
void ssk_ingest(hipStream_t s, const void *src, int channels, int64_t row_stride, int64_t frame_stride,
                int c0, int c1, int c2, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, int n_frames)
{
    if (channels == 1 && ((uintptr_t)src % 16) == 0 && row_stride % 16 == 0 && frame_stride % 16 == 0) {
        dim3 grid16((hg.lv[0].w + 1023) / 1024, (hg.lv[0].h + 3) / 4, n_frames);
        hipLaunchKernelGGL(k_ingest_gray16, grid16, dim3(256), 0, s, (const uint8_t *)src, row_stride, frame_stride, pyr, dg);
        return;
    }
    dim3 grid((hg.lv[0].w + 255) / 256, (hg.lv[0].h + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, s, (const uint8_t *)src, channels, row_stride, frame_stride,
                       c0, c1, c2, pyr, dg);
}

void ssk_resize(hipStream_t s, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, const ss_rtab *rtab,
                int level, int n_frames, const ss_lvl0 &l0)
{
    /* source window of a 64x64 tile: (64 * scale + 1 + 15 alignment) bytes x (64 * scale + 2) rows */
    const float sx = (float)hg.lv[level - 1].w / (float)hg.lv[level].w, sy = (float)hg.lv[level - 1].h / (float)hg.lv[level].h;
    if (64.f * sx + 18.f <= 4.f * RS_WORDS && (float)RS_TILE_H * sy + 3.f <= (float)RS_ROWS) {
        dim3 grid(((hg.lv[level].w + SS_TILE_W - 1) / SS_TILE_W) * ((hg.lv[level].h + RS_TILE_H - 1) / RS_TILE_H), n_frames);
        hipLaunchKernelGGL(k_resize_lds, grid, dim3(256), 0, s, pyr, dg, rtab, level, l0.ptr, l0.pitch, l0.frame_stride);
    } else {
        dim3 grid((hg.lv[level].w + 255) / 256, (hg.lv[level].h + 3) / 4, n_frames);
        hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, pyr, dg, rtab, level, l0.ptr, l0.pitch, l0.frame_stride);
    }
}

/* May levels `level` and `level + 1` be built by one k_resize_pair launch?  Checked on the real tap tables, tile by
 * tile: every window the kernel forms must fit the buffers it is compiled with. */
bool ssk_resize_pair_fits(const ss_geom &hg, const ss_rtab *t, int level)
{
    if (level < 1 || level + 1 >= hg.n_levels) return false;
    const ss_level &S = hg.lv[level - 1], &A = hg.lv[level], &B = hg.lv[level + 1];
    if (A.w < 64 || A.h < 8 || B.w < 8 || B.h < 8) return false;
    const int tiles_x = (B.w + RP_TILE_W - 1) / RP_TILE_W, tiles_y = (B.h + RS_TILE_H - 1) / RS_TILE_H;
    for (int ty = 0; ty < tiles_y; ty++) {
        const int y0 = ty * RS_TILE_H;
        const int ay0 = t[B.ytab_off + y0].s0;
        const int ay_end = ty == tiles_y - 1 ? A.h : (int)t[B.ytab_off + y0 + RS_TILE_H].s0;
        const int ay_need = (int)t[B.ytab_off + std::min(y0 + RS_TILE_H - 1, B.h - 1)].s1 + 1;
        const int n_a = std::max(ay_need, ay_end) - ay0;
        if (n_a < 1 || n_a > RP_A_ROWS || ay_end < ay0) return false;
        const int gy0 = t[A.ytab_off + ay0].s0, gy1 = t[A.ytab_off + std::min(ay0 + n_a - 1, A.h - 1)].s1;
        if (gy1 - gy0 + 1 > RP_S_ROWS || gy1 >= S.h) return false;
        /* every row tap of the window must lie inside it */
        for (int r = 0; r < n_a; r++) {
            const ss_rtab &e = t[A.ytab_off + std::min(ay0 + r, A.h - 1)];
            if (e.s0 < gy0 || e.s1 > gy1) return false;
        }
        for (int r = 0; r < RS_TILE_H && y0 + r < B.h; r++) {
            const ss_rtab &e = t[B.ytab_off + y0 + r];
            if (e.s0 < ay0 || e.s1 >= ay0 + n_a) return false;
        }
    }
    for (int tx = 0; tx < tiles_x; tx++) {
        const int x0 = tx * RP_TILE_W;
        const int ax0 = (int)t[B.xtab_off + x0].s0 & ~3;
        const int ax_end = tx == tiles_x - 1 ? A.pitch : ((int)t[B.xtab_off + x0 + RP_TILE_W].s0 & ~3);
        /* stored columns: [ax0, ax_end) below A.w must be among the 64 computed ones */
        if (std::min(ax_end, (A.w + 3) & ~3) > ax0 + 64 || ax_end < ax0) return false;
        const int gx0 = (int)t[A.xtab_off + ax0].s0 & ~15;
        for (int c = 0; c < 64; c += 4) {
            /* a thread's funnel: 12 bytes from the dword of its first tap; its taps within 8 bytes of it */
            const ss_rtab &f = t[A.xtab_off + ax0 + c];
            if (((f.s0 - gx0) >> 2) * 4 + 12 > 4 * RS_WORDS) return false;
            for (int i = 0; i < 4; i++) {
                const ss_rtab &e = t[A.xtab_off + ax0 + c + i];
                if (e.s0 - f.s0 > 6 || (e.a1 != 0 && e.s1 != e.s0 + 1)) return false;
            }
        }
        for (int c = 0; c < RP_TILE_W && x0 + c < B.w; c += 4) {
            const ss_rtab &f = t[B.xtab_off + x0 + c];
            if (f.s0 < ax0 || ((f.s0 - ax0) >> 2) * 4 + 12 > 4 * RP_B_WORDS) return false;
            for (int i = 0; i < 4; i++) {
                const ss_rtab &e = t[B.xtab_off + x0 + c + i];
                if (e.s0 - f.s0 > 6 || (e.a1 != 0 && e.s1 != e.s0 + 1)) return false;
                if (x0 + c + i < B.w && (int)e.s1 >= ax0 + 64) return false; /* a tap outside the computed columns */
            }
        }
    }
    return true;
}

void ssk_resize_pair(hipStream_t s, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, const ss_rtab *rtab, int level,
                     int n_frames, const ss_lvl0 &l0)
{
    const ss_level &B = hg.lv[level + 1];
    dim3 grid(((B.w + RP_TILE_W - 1) / RP_TILE_W) * ((B.h + RS_TILE_H - 1) / RS_TILE_H), n_frames);
    hipLaunchKernelGGL(k_resize_pair, grid, dim3(256), 0, s, pyr, dg, rtab, level, l0.ptr, l0.pitch, l0.frame_stride);
}

void ssk_fast_blur_nms(hipStream_t s, const uint8_t *pyr, uint8_t *score, uint8_t *blur, const ss_geom *dg, const ss_geom &hg,
                       const uint32_t *tile_recs, const uint16_t *cinfo, uint32_t *tsurv, uint32_t *thdr, ss_level_state *state,
                       int n_frames, const ss_lvl0 &l0)
{
    hipLaunchKernelGGL(k_fast_score, dim3(hg.tiles2_total, n_frames), dim3(FT_THREADS), 0, s, pyr, score, dg, tile_recs, cinfo, tsurv, thdr,
                       state, blur, l0.ptr, l0.pitch, l0.frame_stride);
}
void ssk_bucket_gather(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint32_t *cell_units, const uint32_t *tsurv,
                       const uint32_t *thdr, uint32_t *bucket, uint32_t *cell_cnt, ss_level_state *state, int n_frames)
{
    hipLaunchKernelGGL(k_bucket_gather, dim3((hg.n_cells * 16 + 255) / 256, n_frames), dim3(256), 0, s, dg, cell_units, tsurv, thdr, bucket,
                       cell_cnt, state);
}
void ssk_cells_emit(hipStream_t s, const uint32_t *bucket, const ss_geom *dg, const ss_geom &hg, const uint32_t *cell_cnt,
                    uint32_t *cand, ss_level_state *state, int n_frames)
{
    hipLaunchKernelGGL(k_cells_emit, dim3(hg.chunks_total, n_frames), dim3(256), 0, s, bucket, dg, cell_cnt, cand, state);
}
void ssk_quadtree(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint32_t *cand, uint32_t *qbuf0,
                  uint32_t *qbuf1, ss_qnode *nodes, int32_t *lists, uint32_t *sel, ss_level_state *state, int n_frames)
{
    int need = 0;
    for (int l = 0; l < hg.n_levels; l++) need = hg.lv[l].item_cap > need ? hg.lv[l].item_cap : need;
    need = (need + 63) & ~63;
    hipLaunchKernelGGL(k_quadtree, dim3(n_frames, hg.n_levels), dim3(QT_THREADS), (size_t)need * (2 * sizeof(uint64_t) + 2 * sizeof(int32_t)), s, dg, cand, qbuf0,
                       qbuf1, nodes, lists, sel, state, need);
}

void ssk_slots(hipStream_t s, const ss_geom *dg, const uint32_t *sel, const ss_level_state *state, uint32_t *kp_ref,
               int32_t *n_kp, int32_t *level_counts, int32_t *frame_error, int n_frames)
{
    hipLaunchKernelGGL(k_slots, dim3(n_frames), dim3(64), 0, s, dg, sel, state, kp_ref, n_kp, level_counts, frame_error);
}

void ssk_orient_describe(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint8_t *pyr, const uint8_t *blur,
                         const uint32_t *sel, const uint32_t *kp_ref, const int32_t *n_kp, ss_keypoint *kps,
                         uint8_t *desc, int n_frames, const ss_lvl0 &l0, bool steer_fma, uint8_t *desc_x)
{
    /* kcap is a multiple of 64: kcap / (4 waves x OD_KP keypoints) blocks per frame */
    if (steer_fma)
        hipLaunchKernelGGL((k_orient_describe<true, OD_KP>), dim3(hg.kcap / (4 * OD_KP), n_frames), dim3(256), 0, s, dg, pyr, blur, sel,
                           kp_ref, n_kp, kps, desc, l0.ptr, l0.pitch, l0.frame_stride, desc_x);
    else
        hipLaunchKernelGGL((k_orient_describe<false, OD_KP>), dim3(hg.kcap / (4 * OD_KP), n_frames), dim3(256), 0, s, dg, pyr, blur, sel,
                           kp_ref, n_kp, kps, desc, l0.ptr, l0.pitch, l0.frame_stride, desc_x);
}

/* which form of the matrix-core kernel: a single large database has the chip to itself (NU = 2), batches of frames share it */
static int mm_tiles_per_wave(int n_train_max, int n_frames) { return n_frames == 1 && n_train_max >= 65536 ? 2 : 1; }

int ssk_match_chunks(int n_query_max, int n_train_max, int n_frames, int *chunk_len)
{
    /* the matrix-core kernel: 256 queries per block, 2 resident blocks per CU, and a per-block prologue (table, query
     * fragments) that wants >= 8 tiles of 32 train rows behind it; the VALU kernel: 64 queries per block, 8 per CU */
    const bool mfma = n_query_max >= SSK_MATCH_MFMA_MIN_QUERIES;
    const int qblock = MM_QBLOCK(mm_tiles_per_wave(n_train_max, n_frames));
    const int q_groups = mfma ? (n_query_max + qblock - 1) / qblock : (n_query_max + 63) / 64;
    const long blocks_wanted = mfma ? 2048 : 16384; /* >> resident blocks: keeps the last partial round of blocks small */
    long chunks = (blocks_wanted + (long)q_groups * n_frames - 1) / ((long)q_groups * n_frames > 0 ? (long)q_groups * n_frames : 1);
    const long max_chunks = (n_train_max + 255) / 256; /* >= 64 train rows per wave / >= 8 tiles per block */
    if (chunks > max_chunks) chunks = max_chunks;
    /* 16-bit local row index in the keys: per quarter-chunk of a wave in k_match, per chunk in k_match_mfma */
    const long max_len = mfma ? 65536 : 4 * 32768;
    const long min_chunks = ((long)n_train_max + max_len - 1) / max_len;
    if (chunks < min_chunks) chunks = min_chunks;
    if (chunks < 1) chunks = 1;
    int len = (int)(((long)n_train_max + chunks - 1) / chunks);
    len = mfma ? (len + MM_TILE - 1) & ~(MM_TILE - 1) : (len + 3) & ~3;
    if (len < 4) len = 4;
    *chunk_len = len;
    return (int)(((long)n_train_max + len - 1) / len > 0 ? ((long)n_train_max + len - 1) / len : 1);
}

/* which form of k_match_mfma_x: the pipelined one (two query tiles per wave, two waves per SIMD) for ONE query set against a
 * large train set, which has the chip to itself; the compact one (five waves per SIMD, 128-query blocks) for the frames of
 * a batch and for small sets, which share the chip with the other batches' kernels.  SENDSLAM_MX_FORM=compact|pipelined
 * overrides (A/B measurements). */
static bool mx_pipelined(int n_frames, int rows_t)
{
    if (const char *e = getenv("SENDSLAM_MX_FORM")) return e[0] == 'p';
    return n_frames == 1 && rows_t >= 65536;
}

template <bool FUSED, int QT, bool PIPE>
static void mx_launch(hipStream_t s, dim3 grid, const uint8_t *query_x, const uint8_t *train_x, const int32_t *nq_arr, const int32_t *nt_arr, int nq_fixed,
                      int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride, int train_frame_shift, int chunk_len, int n_chunks,
                      int exclude_self_mode, int out_stride, void *partial, const mx_finish &fin)
{
    hipLaunchKernelGGL((k_match_mfma_x<FUSED, QT, PIPE>), grid, dim3(256), 0, s, query_x, train_x, nq_arr, nt_arr, nq_fixed, nt_fixed, q_frame_stride,
                       t_frame_stride, train_frame_shift, chunk_len, n_chunks, exclude_self_mode, out_stride, (match_partial *)partial, fin);
}

/* the matcher's first launch (or only one, when fused); returns true when the outputs are final */
static bool mx_match(hipStream_t s, bool pipelined, int n_frames, const uint8_t *query_x, const uint8_t *train_x, const int32_t *nq_arr,
                     const int32_t *nt_arr, int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride, int train_frame_shift,
                     int chunk_len, int n_chunks, int exclude_self_mode, int out_stride, void *partial, const mx_finish &fin)
{
    const int qblock = pipelined ? MX_QBLOCK_OF(2) : MX_QBLOCK_OF(1);
    dim3 grid(((out_stride + qblock - 1) / qblock) * n_chunks * n_frames);
    const bool fused = n_chunks == 1 && fin.query_p && fin.train_p; /* one launch: the kernel finishes its queries itself */
#define MX_GO(F, Q, P)                                                                                                                      \
    mx_launch<F, Q, P>(s, grid, query_x, train_x, nq_arr, nt_arr, nq_fixed, nt_fixed, q_frame_stride, t_frame_stride, train_frame_shift, chunk_len, \
                       n_chunks, exclude_self_mode, out_stride, partial, fin)
    if (pipelined) {
        if (fused) MX_GO(true, 2, true);
        else MX_GO(false, 2, true);
    } else {
        if (fused) MX_GO(true, 1, false);
        else MX_GO(false, 1, false);
    }
#undef MX_GO
    return fused;
}

/* batch form on expanded descriptors (desc_x of the extraction): same arguments as ssk_match, strides in BYTES.  With several
 * chunks two launches: k_match_mfma_x writes one partial per (query, chunk), k_match_finish_x folds them, adds the second best
 * inside the best row's group and applies the acceptance test. */
void ssk_match_x(hipStream_t s, const uint8_t *query_x, const uint8_t *train_x, const int32_t *nq_arr, const int32_t *nt_arr,
                 int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride, int train_frame_shift, int chunk_len,
                 int n_chunks, int exclude_self_mode, int th, int rnum, int rden, int out_stride, void *partial, int32_t *idx,
                 uint16_t *d1, uint16_t *d2, int n_frames, const uint8_t *query_p, const uint8_t *train_p, int64_t qp_frame_stride,
                 int64_t tp_frame_stride)
{
    mx_finish fin{query_p, train_p, qp_frame_stride, tp_frame_stride, th, rnum, rden, idx, d1, d2};
    if (mx_match(s, mx_pipelined(n_frames, out_stride), n_frames, query_x, train_x, nq_arr, nt_arr, nq_fixed, nt_fixed, q_frame_stride, t_frame_stride,
                 train_frame_shift, chunk_len, n_chunks, exclude_self_mode, out_stride, partial, fin))
        return;
    dim3 g2((out_stride + 63) / 64, n_frames);
    if (query_p && train_p) /* the same rows as packed descriptors: the finish reads those */
        hipLaunchKernelGGL(k_match_finish_x<32>, g2, dim3(256), 0, s, query_p, train_p, nq_arr, nt_arr, nq_fixed, nt_fixed, qp_frame_stride,
                           tp_frame_stride, train_frame_shift, exclude_self_mode, (const match_partial *)partial, n_chunks, th, rnum, rden, out_stride,
                           idx, d1, d2);
    else
        hipLaunchKernelGGL(k_match_finish_x<SS_X_ROW>, g2, dim3(256), 0, s, query_x, train_x, nq_arr, nt_arr, nq_fixed, nt_fixed, q_frame_stride,
                           t_frame_stride, train_frame_shift, exclude_self_mode, (const match_partial *)partial, n_chunks, th, rnum, rden, out_stride,
                           idx, d1, d2);
}

/* chunk plan: rows_q query rows and rows_t train rows per frame.  Once the query blocks alone fill the chip, one chunk per
 * block (and the fused epilogue); fewer query blocks are spread over more chunks of >= 8 tiles. */
int ssk_match_x_batch_chunks(int rows_q, int rows_t, int n_frames, int *chunk_len)
{
    const bool pipelined = mx_pipelined(n_frames, rows_t);
    const int qblock = pipelined ? MX_QBLOCK_OF(2) : MX_QBLOCK_OF(1);
    const int q_groups = ((rows_q + qblock - 1) / qblock) * (n_frames > 0 ? n_frames : 1);
    const int fill = pipelined ? 512 : 1024; /* resident blocks: 2 resp. 5 per CU */
    int want = q_groups >= fill ? 1 : (2 * fill + q_groups / 2) / q_groups;
    if (const char *e = getenv("SENDSLAM_MX_CHUNKS")) want = atoi(e); /* experiments */
    const int max_chunks = (rows_t + 255) / 256;
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    int len = ((rows_t + want - 1) / want + MM_TILE - 1) & ~(MM_TILE - 1);
    if (len > (1 << MX_ROW_BITS)) len = 1 << MX_ROW_BITS;
    if (len < MM_TILE) len = MM_TILE;
    *chunk_len = len;
    const int n = (rows_t + len - 1) / len;
    return n < 1 ? 1 : n;
}

void ssk_expand_desc(hipStream_t s, const void *packed, int n, void *out)
{
    const int n_alloc = (n + MM_TILE - 1) & ~(MM_TILE - 1);
    if (n_alloc > 0)
        hipLaunchKernelGGL(k_expand_desc, dim3((n_alloc + 3) / 4), dim3(256), 0, s, (const uint32_t *)packed, n, n_alloc, (uint8_t *)out, (int64_t)0,
                           (int64_t)0);
}

/* [n_frames][rows][32] packed -> [n_frames][rows rounded up to 32][SSK_X_ROW] */
void ssk_expand_desc_frames(hipStream_t s, const void *packed, int rows, int n_frames, void *out)
{
    const int n_alloc = (rows + MM_TILE - 1) & ~(MM_TILE - 1);
    if (n_alloc > 0 && n_frames > 0)
        hipLaunchKernelGGL(k_expand_desc, dim3((n_alloc + 3) / 4, n_frames), dim3(256), 0, s, (const uint32_t *)packed, rows, n_alloc, (uint8_t *)out,
                           (int64_t)rows * 8, (int64_t)n_alloc * SS_X_ROW);
}

/* one query set against one (large) train set, both expanded: chunks of <= 8192 rows (the key's row field), the query
 * blocks of a chunk adjacent in the grid (one XCD streams the chunk once) */
int ssk_match_x_chunks(int n_query, int n_train, int *chunk_len) { return ssk_match_x_batch_chunks(n_query, n_train, 1, chunk_len); }

void ssk_match_x_single(hipStream_t s, const uint8_t *query_x, int nq, const uint8_t *train_x, int nt, int chunk_len, int n_chunks,
                        int exclude_self, int th, int rnum, int rden, void *partial, int32_t *idx, uint16_t *d1, uint16_t *d2,
                        const uint8_t *query_p, const uint8_t *train_p)
{
    mx_finish fin{query_p, train_p, 0, 0, th, rnum, rden, idx, d1, d2};
    if (mx_match(s, mx_pipelined(1, nt), 1, query_x, train_x, nullptr, nullptr, nq, nt, 0, 0, 0, chunk_len, n_chunks, exclude_self ? 1 : 0, nq, partial, fin))
        return;
    int fin_chunks = n_chunks;
    if (n_chunks >= 32) { /* many chunks: one wave per query folds them (raw: no acceptance test yet), the finish reads the outputs */
        hipLaunchKernelGGL(k_match_merge_wide, dim3(nq), dim3(64), 0, s, (const match_partial *)partial, nq, n_chunks, -1, 1, 1, nq, idx, d1, d2);
        fin_chunks = 0;
    }
    if (query_p && train_p)
        hipLaunchKernelGGL(k_match_finish_x<32>, dim3((nq + 63) / 64, 1), dim3(256), 0, s, query_p, train_p, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, nq, nt, (int64_t)0, (int64_t)0, 0, exclude_self ? 1 : 0, (const match_partial *)partial,
                           fin_chunks, th, rnum, rden, nq, idx, d1, d2);
    else
        hipLaunchKernelGGL(k_match_finish_x<SS_X_ROW>, dim3((nq + 63) / 64, 1), dim3(256), 0, s, query_x, train_x, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, nq, nt, (int64_t)0, (int64_t)0, 0, exclude_self ? 1 : 0, (const match_partial *)partial,
                           fin_chunks, th, rnum, rden, nq, idx, d1, d2);
}

void ssk_match(hipStream_t s, const void *query, const void *train, const int32_t *nq_arr, const int32_t *nt_arr,
               int nq_fixed, int nt_fixed, int64_t q_frame_stride_words, int64_t t_frame_stride_words,
               int train_frame_shift, int chunk_len, int n_chunks, int exclude_self_mode, int th, int rnum, int rden,
               int out_stride, void *partial, int32_t *idx, uint16_t *d1, uint16_t *d2, int n_frames)
{
    if (out_stride >= SSK_MATCH_MFMA_MIN_QUERIES) {
        /* many queries: the matrix-core form */
        const int nu = mm_tiles_per_wave(nt_fixed, n_frames);
        dim3 grid((out_stride + MM_QBLOCK(nu) - 1) / MM_QBLOCK(nu), n_chunks, n_frames);
        if (nu == 2)
            hipLaunchKernelGGL(k_match_mfma<2>, grid, dim3(64 * MM_WAVES), 0, s, (const uint32_t *)query, (const uint32_t *)train, nq_arr,
                               nt_arr, nq_fixed, nt_fixed, q_frame_stride_words, t_frame_stride_words, train_frame_shift, chunk_len,
                               n_chunks, exclude_self_mode, th, rnum, rden, out_stride, (match_partial *)partial, idx, d1, d2);
        else
            hipLaunchKernelGGL(k_match_mfma<1>, grid, dim3(64 * MM_WAVES), 0, s, (const uint32_t *)query, (const uint32_t *)train, nq_arr,
                               nt_arr, nq_fixed, nt_fixed, q_frame_stride_words, t_frame_stride_words, train_frame_shift, chunk_len,
                               n_chunks, exclude_self_mode, th, rnum, rden, out_stride, (match_partial *)partial, idx, d1, d2);
    } else {
        dim3 grid((out_stride + 63) / 64, n_chunks, n_frames);
        hipLaunchKernelGGL(k_match, grid, dim3(256), 0, s, (const uint32_t *)query, (const uint32_t *)train, nq_arr, nt_arr,
                           nq_fixed, nt_fixed, q_frame_stride_words, t_frame_stride_words, train_frame_shift, chunk_len,
                           n_chunks, exclude_self_mode, th, rnum, rden, out_stride, (match_partial *)partial, idx, d1, d2);
    }
    if (n_chunks >= 32 && n_frames == 1 && !nq_arr) {
        hipLaunchKernelGGL(k_match_merge_wide, dim3(out_stride), dim3(64), 0, s, (const match_partial *)partial, nq_fixed, n_chunks,
                           th, rnum, rden, out_stride, idx, d1, d2);
    } else if (n_chunks > 1) {
        dim3 g2((out_stride + 255) / 256, n_frames);
        hipLaunchKernelGGL(k_match_merge, g2, dim3(256), 0, s, (const match_partial *)partial, nq_arr, nq_fixed,
                           n_chunks, th, rnum, rden, out_stride, idx, d1, d2);
    }
}

/* database-streaming match for n_query <= 8 (see k_match_stream): plan (false = not applicable), kernel launch, merge
 * launch -- separate calls so that the caller can time the HBM-bound kernel on its own */
bool ssk_match_stream_plan(int nq, int nt, size_t partial_bytes, int *chunk_len_out, int *n_chunks_out)
{
    if (nq < 1 || nq > 8 || nt < 65536) return false;
    int n_chunks = 256 * 8;                       /* 8 resident blocks per CU */
    int chunk_len = (nt + n_chunks - 1) / n_chunks;
    chunk_len = (chunk_len + 255) & ~255;          /* whole iterations of the block */
    if (chunk_len > (1 << 22) * 256 / 256) return false;
    n_chunks = (nt + chunk_len - 1) / chunk_len;
    if ((size_t)n_chunks * nq * SSK_MATCH_PARTIAL_BYTES > partial_bytes) return false;
    *chunk_len_out = chunk_len;
    *n_chunks_out = n_chunks;
    return true;
}

void ssk_match_stream_kernel(hipStream_t s, const void *query, const void *train, int nq, int nt, int chunk_len, int n_chunks,
                             void *partial)
{
    const uint32_t *q = (const uint32_t *)query;
    const uint4 *t = (const uint4 *)train;
    match_partial *p = (match_partial *)partial;
    if (nq == 1) hipLaunchKernelGGL(k_match_stream<1>, dim3(n_chunks), dim3(256), 0, s, q, t, nq, nt, chunk_len, n_chunks, p);
    else if (nq == 2) hipLaunchKernelGGL(k_match_stream<2>, dim3(n_chunks), dim3(256), 0, s, q, t, nq, nt, chunk_len, n_chunks, p);
    else if (nq <= 4) hipLaunchKernelGGL(k_match_stream<4>, dim3(n_chunks), dim3(256), 0, s, q, t, nq, nt, chunk_len, n_chunks, p);
    else hipLaunchKernelGGL(k_match_stream<8>, dim3(n_chunks), dim3(256), 0, s, q, t, nq, nt, chunk_len, n_chunks, p);
}

void ssk_match_stream_merge(hipStream_t s, const void *partial, int nq, int n_chunks, int th, int rnum, int rden, int32_t *idx,
                            uint16_t *d1, uint16_t *d2)
{
    hipLaunchKernelGGL(k_match_merge_wide, dim3(nq), dim3(64), 0, s, (const match_partial *)partial, nq, n_chunks, th, rnum, rden, nq,
                       idx, d1, d2);
}

void ssk_pack_partial(hipStream_t s, const int32_t *idx, const uint16_t *d1, const uint16_t *d2, int n, int32_t row_offset,
                      void *part)
{
    hipLaunchKernelGGL(k_pack_partial, dim3((n + 255) / 256), dim3(256), 0, s, idx, d1, d2, n, row_offset, (match_partial *)part);
}

void ssk_match_fold(hipStream_t s, const void *parts, int n_parts, int nq, int th, int rnum, int rden, int32_t *idx,
                    uint16_t *d1, uint16_t *d2)
{
    hipLaunchKernelGGL(k_match_merge, dim3((nq + 255) / 256, 1), dim3(256), 0, s, (const match_partial *)parts, (const int32_t *)nullptr,
                       nq, n_parts, th, rnum, rden, nq, idx, d1, d2);
}

/* the cross-shard fold on parts that sit part_stride_bytes apart (what ss_xchg_allgather leaves) */
__global__ __launch_bounds__(256) void k_match_fold_strided(const uint8_t *__restrict__ parts, int64_t part_stride_bytes, int n_parts, int nq, int th,
                                                            int rnum, int rden, int32_t *__restrict__ idx_out, uint16_t *__restrict__ d1_out,
                                                            uint16_t *__restrict__ d2_out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    int d1 = 0xFFFF, d2 = 0xFFFF, j1 = -1;
    for (int c = 0; c < n_parts; c++) {
        const match_partial mp = ((const match_partial *)(parts + (size_t)c * part_stride_bytes))[qi];
        merge_partial(d1, j1, d2, mp.d1, mp.j1, mp.d2);
    }
    const bool ok = j1 >= 0 && (th < 0 || (d1 <= th && d1 * rden < d2 * rnum));
    idx_out[qi] = ok ? j1 : -1;
    d1_out[qi] = (uint16_t)d1;
    d2_out[qi] = (uint16_t)d2;
}

void ssk_match_fold_strided(hipStream_t s, const void *parts, int64_t part_stride_bytes, int n_parts, int nq, int th, int rnum, int rden,
                            int32_t *idx, uint16_t *d1, uint16_t *d2)
{
    hipLaunchKernelGGL(k_match_fold_strided, dim3((nq + 255) / 256), dim3(256), 0, s, (const uint8_t *)parts, part_stride_bytes, n_parts, nq, th,
                       rnum, rden, idx, d1, d2);
}

int ssk_debug_sort(hipStream_t s, uint64_t *d_items, int n)
{
    if (n < 0 || n > QT_MAX_ITEMS) return -1;
    hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(64), 0, s, d_items, n);
    return 0;
}
