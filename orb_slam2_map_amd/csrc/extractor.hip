// ORB extraction pipeline for MI355X (gfx950): image pyramid, per-cell FAST-9/16 with NMS and
// threshold fallback, quadtree distribution, intensity-centroid orientation, 7x7 Gaussian blur and
// 256-bit steered BRIEF -- one launch per stage over (cells | tiles | keypoints) x frames.
//
// Replaces ORB_SLAM2::ORBextractor::operator() (reference src/ORBextractor.cc:1043-1105) and the
// OpenCV 2.4 calls it makes.  Integer stages are bit-exact by
// construction; float stages are written without contraction (see common.h).
#include "common.h"

#include <atomic>
#include <chrono>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include "trig_base.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orbgpu {

constexpr int EDGE = 19;           // EDGE_THRESHOLD, ORBextractor.cc:74
constexpr int BORDER0 = EDGE - 3;  // minBorderX/Y, ORBextractor.cc:775-776
constexpr int HALF_PATCH = 15;     // HALF_PATCH_SIZE, ORBextractor.cc:73
constexpr int PATCH = 31;          // PATCH_SIZE
constexpr float CELL_W = 30.f;     // W, ORBextractor.cc:769

// ---------------------------------------------------------------------------------------------
// Geometry tables (host-built, device-read)
// ---------------------------------------------------------------------------------------------
struct LevelGeom {
    int w, h, pitch;     // level size; padded row pitch in bytes (multiple of 64)
    int plane_off;       // byte offset of the padded plane inside one frame's pyramid block
    int max_bx, max_by;  // maxBorderX/Y = w-16, h-16
    int ncols, nrows, wcell, hcell;
    int cell_first, ncells;  // this level's (non-skipped) cells in the cell table
    int slot_off, slot_cnt;  // this level's key slots inside one frame's slot block (u32 units)
    int quota, sel_cap, sel_off;
    int n_ini;
    float hx;
    float scale;
    int patch;
    int xtab_off, ytab_off;  // resize tables (levels >= 1)
    int rs_off, rs_fast;     // fast-path strip tables (k_resize_fast); rs_fast = 0 -> k_resize_level
    int ncols_eff;           // cell columns that are not skipped (ORBextractor.cc:797)
    uint32_t inv_wcell, inv_hcell;  // mul_hi(v, inv) == v / wcell (hcell) for every key coordinate (host-verified)
    int yrow_off;                   // k_resize_fast: first YRow of the level (one per PADDED output row)
};

struct CellDesc {
    short level;
    short x0, y0, x1, y1;  // sub-image [x0,x1) x [y0,y1) in level coordinates (FAST apron included)
    short addx, addy;      // j*wCell, i*hCell  (ORBextractor.cc:822-823)
    short cap;             // slot capacity
    int slot_off;          // offset of this cell's slots inside one frame's slot block
    int pitch, plane_off;  // copies of the level's LevelGeom fields
    int pad[1];
};
static_assert(sizeof(CellDesc) == 32, "CellDesc is read as two 16-byte words");

struct XTab {  // cv::resize horizontal table entry (A2)
    uint16_t sx, sx1, a0, a1;
};
struct YTab {
    uint16_t sy0, sy1;
    int16_t b0, b1;
};
struct YRow {  // the same per PADDED output row (border rows = the entry of the row they reflect to), as four ints: one
    int sy0, sy1, b0, b1;  // 16-byte load, no reflection and no field extraction in k_resize_fast's row loop
};


// packed FAST key: y[31:20] x[19:8] response[7:0]; x,y relative to (minBorderX,minBorderY)
__host__ __device__ __forceinline__ uint32_t pack_key(int x, int y, int resp)
{
    return ((uint32_t)y << 20) | ((uint32_t)x << 8) | (uint32_t)resp;
}
__host__ __device__ __forceinline__ int key_x(uint32_t k) { return (int)((k >> 8) & 0xFFF); }
__host__ __device__ __forceinline__ int key_y(uint32_t k) { return (int)(k >> 20); }
__host__ __device__ __forceinline__ int key_resp(uint32_t k) { return (int)(k & 0xFF); }

__host__ __device__ __forceinline__ int reflect101(int p, int len)
{
    // single bounce is enough: every level is >= 62 px and the border is 19 px
    if (p < 0)
        p = -p;
    else if (p >= len)
        p = 2 * (len - 1) - p;
    return p;
}

// Level 0 read straight from the caller's image ("direct" mode).  ComputePyramid's level 0 is copyMakeBorder(image) into a
// padded plane (ORBextractor.cc:1125-1129) -- 0.67 MB of traffic per 640x480 frame that no stage needs: FAST, the
// orientation disc and the resize to level 1 only read the image interior, and the blur can reflect its 3-px
// rim itself.  With aligned inputs (launch_pipeline) the kernels that read level 0 take the image from here and the
// padded copy is only materialised when mvImagePyramid[0] is asked for (orbgpu_extractor_get_pyramid_level).
// Pixel (x, y) of frame f = p[f * frame_stride + y * pitch + x]; the padded planes have the pixel at (x + EDGE, y + EDGE).
struct Src0 {
    const uint8_t *p;
    size_t frame_stride;
    uint32_t pitch;
    int direct;  // 0: level 0 is the padded plane like every other level
};
constexpr int BLUR0_OX = EDGE + 1;  // direct mode: column of pixel 0 in the BLURRED level-0 plane (k_blur0_direct stores 8 aligned bytes per strip)

// ---------------------------------------------------------------------------------------------
// K1a: level 0 = copyMakeBorder(image, REFLECT_101)           (ORBextractor.cc:1125-1129, A6)
// ---------------------------------------------------------------------------------------------
// byte offset of a row inside a padded plane: rows and pitches are < 2^24, so a 24-bit multiply (full rate) is exact;
// the 64-bit (size_t) product the obvious expression asks for costs two quarter-rate multiplies per use
__device__ __forceinline__ uint32_t rowoff(int row, int pitch) { return __umul24((unsigned)row, (unsigned)pitch); }

constexpr int PYR_ROWS = 8;  // padded rows per thread in the pyramid kernels

__global__ __launch_bounds__(256) void k_border0(const uint8_t *__restrict__ src, size_t stride,
                                                 size_t frame_stride, uint8_t *__restrict__ pyr,
                                                 size_t frame_pyr, const LevelGeom *__restrict__ geom)
{
    const LevelGeom g = geom[0];
    // work item = (group of PYR_ROWS padded rows, aligned dword of the row), flattened so that every lane
    // of a workgroup has work whatever the row length
    const int ndw = g.pitch >> 2;
    int bx, f;
    xcd_frame_block(bx, f);
    const int item = bx * 256 + threadIdx.x;
    const int rg = item / ndw;
    const int x4 = (item - rg * ndw) * 4;
    if (rg * PYR_ROWS >= g.h + 2 * EDGE)
        return;
    const int sx0 = x4 - EDGE;  // source column of the first of the 4 output pixels
    const bool interior = sx0 >= 0 && sx0 + 7 < g.w;
#pragma unroll 4
    for (int rr = 0; rr < PYR_ROWS; rr++) {
        const int py = rg * PYR_ROWS + rr;
        if (py >= g.h + 2 * EDGE)
            break;
        const int sy = reflect101(py - EDGE, g.h);
        const uint8_t *s = src + (size_t)f * frame_stride + __umul24((unsigned)sy, (unsigned)stride);  // stride < 2^24 (host check)
        uint32_t v = 0;
        if (interior && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {
            // interior: two aligned source dwords + v_alignbyte (the 19-px border shifts rows by 3 bytes)
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(s) + (sx0 >> 2);
            v = __builtin_amdgcn_alignbyte(s32[1], s32[0], sx0 & 3);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int px = min(x4 + k, g.w + 2 * EDGE - 1);
                int sx = reflect101(px - EDGE, g.w);
                v |= (uint32_t)s[sx] << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(pyr + (size_t)f * frame_pyr + g.plane_off + rowoff(py, g.pitch) + x4) = v;
    }
}

// Fast path of K1a for 4-byte aligned rows whose width is a multiple of 4 (host-checked; k_border0 otherwise): every
// aligned dword of the padded row -- interior, reflected border, or the fold between them -- is 4 bytes out of two
// adjacent source dwords, picked by one v_perm_b32 with a selector that only depends on the column (BorderCol, built
// on the host).  No per-byte path and no divergence: a thread owns 16 output bytes x BORDER_ROWS rows.
constexpr int BORDER_ROWS = 4;

struct BorderCol {
    uint32_t d;    // first of the two adjacent source dwords
    uint32_t sel;  // v_perm_b32 selector over {src[d+1], src[d]}
};

__global__ __launch_bounds__(256) void k_border0_fast(const uint8_t *__restrict__ src, size_t stride, size_t frame_stride,
                                                      uint8_t *__restrict__ pyr, size_t frame_pyr,
                                                      const LevelGeom *__restrict__ geom,
                                                      const BorderCol *__restrict__ cols)
{
    const LevelGeom g = geom[0];
    const int n16 = g.pitch >> 4;
    int bx, f;
    xcd_frame_block(bx, f);
    const int item = bx * 256 + threadIdx.x;
    const int rg = item / n16;
    const int c16 = item - rg * n16;
    if (rg * BORDER_ROWS >= g.h + 2 * EDGE)
        return;
    const uint4 t01 = reinterpret_cast<const uint4 *>(cols)[c16 * 2], t23 = reinterpret_cast<const uint4 *>(cols)[c16 * 2 + 1];
    const uint8_t *sf = src + (size_t)f * frame_stride;
    uint8_t *dst = pyr + (size_t)f * frame_pyr + g.plane_off + c16 * 16;
    // interior items (their 16 bytes are source bytes 16 c - 19 .. 16 c - 4): 5 consecutive dwords from 4 c - 5, every
    // output dword = bytes 1..4 of a dword pair.  Only the two items at each end of a row go through the table.
    const bool interior = c16 >= 2 && c16 * 16 - EDGE + 15 < g.w - 1 && c16 * 4 - 5 + 4 < (g.w >> 2);
    struct __attribute__((packed, aligned(4))) Win {
        uint32_t d[5];
    };
    if (interior) {
        Win wn[BORDER_ROWS];
#pragma unroll
        for (int rr = 0; rr < BORDER_ROWS; rr++) {
            const int py = min(rg * BORDER_ROWS + rr, g.h + 2 * EDGE - 1);
            wn[rr] = *reinterpret_cast<const Win *>(sf + __umul24((unsigned)reflect101(py - EDGE, g.h), (unsigned)stride) +
                                                    (c16 * 16 - 20));
        }
#pragma unroll
        for (int rr = 0; rr < BORDER_ROWS; rr++) {
            const int py = rg * BORDER_ROWS + rr;
            if (py >= g.h + 2 * EDGE)
                break;
            uint4 v;
            v.x = __builtin_amdgcn_alignbyte(wn[rr].d[1], wn[rr].d[0], 1);
            v.y = __builtin_amdgcn_alignbyte(wn[rr].d[2], wn[rr].d[1], 1);
            v.z = __builtin_amdgcn_alignbyte(wn[rr].d[3], wn[rr].d[2], 1);
            v.w = __builtin_amdgcn_alignbyte(wn[rr].d[4], wn[rr].d[3], 1);
            *reinterpret_cast<uint4 *>(dst + rowoff(py, g.pitch)) = v;
        }
        return;
    }
#pragma unroll
    for (int rr = 0; rr < BORDER_ROWS; rr++) {
        const int py = rg * BORDER_ROWS + rr;
        if (py >= g.h + 2 * EDGE)
            break;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(sf + __umul24((unsigned)reflect101(py - EDGE, g.h), (unsigned)stride));
        uint4 v;
        v.x = __builtin_amdgcn_perm(s32[t01.x + 1], s32[t01.x], t01.y);
        v.y = __builtin_amdgcn_perm(s32[t01.z + 1], s32[t01.z], t01.w);
        v.z = __builtin_amdgcn_perm(s32[t23.x + 1], s32[t23.x], t23.y);
        v.w = __builtin_amdgcn_perm(s32[t23.z + 1], s32[t23.z], t23.w);
        *reinterpret_cast<uint4 *>(dst + rowoff(py, g.pitch)) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K1b: level l = resize(level l-1, INTER_LINEAR) + copyMakeBorder(REFLECT_101|ISOLATED)
//      (ORBextractor.cc:1118-1124; OpenCV 2.4 8-bit fixed-point bilinear, A2).  Border pixels are
//      produced by evaluating the resize at the reflected coordinate, so one pass writes the
//      whole padded plane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_level(uint8_t *__restrict__ pyr, size_t frame_pyr,
                                                      const LevelGeom *__restrict__ geom, int level,
                                                      const XTab *__restrict__ xtab,
                                                      const YTab *__restrict__ ytab)
{
    const LevelGeom g = geom[level];
    const LevelGeom gs = geom[level - 1];
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int py = blockIdx.y;
    const int f = blockIdx.z;
    if (x4 >= g.pitch)
        return;
    const int dy = reflect101(py - EDGE, g.h);
    const YTab yt = ytab[g.ytab_off + dy];
    const uint8_t *base = pyr + (size_t)f * frame_pyr + gs.plane_off + rowoff(EDGE, gs.pitch) + EDGE;
    const uint8_t *S0 = base + rowoff(yt.sy0, gs.pitch);
    const uint8_t *S1 = base + rowoff(yt.sy1, gs.pitch);
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int px = min(x4 + k, g.w + 2 * EDGE - 1);
        int dx = reflect101(px - EDGE, g.w);
        const XTab xt = xtab[g.xtab_off + dx];
        int t0 = S0[xt.sx] * xt.a0 + S0[xt.sx1] * xt.a1;
        int t1 = S1[xt.sx] * xt.a0 + S1[xt.sx1] * xt.a1;
        int o = ((((int)yt.b0 * (t0 >> 4)) >> 16) + (((int)yt.b1 * (t1 >> 4)) >> 16) + 2) >> 2;
        v |= (uint32_t)(o & 0xFF) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(pyr + (size_t)f * frame_pyr + g.plane_off + rowoff(py, g.pitch) + x4) = v;
}

// Fast path of K1b: one thread per aligned output dword (4 pixels) of the padded plane.  The 8 source
// bytes the 4 outputs need per source row lie inside one 12-byte aligned window (true for scale
// factors <= 2; checked on the host, otherwise k_resize_level runs), which is fetched with 3 dword
// loads and shifted by the column's byte offset (two v_alignbyte) so that they lie in one 8-byte pair; v_perm_b32 with
// host-precomputed selectors forms (left tap | right tap << 16) pairs and v_dot2_u32_u16 applies the 11-bit weights.
// Same integer arithmetic as k_resize_level.
struct ResizeStrip {  // per padded output dword column of a level
    uint32_t base_q;  // bits 0..15: window base (padded source column, multiple of 4); bits 16..17: byte shift of the window;
                      // bit 18 (direct level-0 source only): the window is the last 12 bytes of the row and the pair comes from its dwords 1, 2
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// REUSE: keep a source row's horizontal interpolation for the next output row (pays on the large levels).
// DIRECT: level 1 from the caller's image (Src0) instead of the padded level-0 plane; the strip tables are built for
// unpadded source columns and a window that would run past the end of an image row re-reads its second dword.
template <bool REUSE, bool DIRECT>
__global__ __launch_bounds__(256) void k_resize_fast(uint8_t *__restrict__ pyr, size_t frame_pyr,
                                                     const LevelGeom *__restrict__ geom, int level,
                                                     const ResizeStrip *__restrict__ strips,
                                                     const uint4 *__restrict__ sels, const uint4 *__restrict__ wts,
                                                     const YRow *__restrict__ yrows, int strip_off, int rows, Src0 s0)
{
    const LevelGeom g = geom[level];
    const LevelGeom gs = geom[level - 1];
    const int ndw = g.pitch >> 2;
    int bx, f;
    xcd_frame_block(bx, f);
    const int item = bx * 256 + threadIdx.x;  // (row group, output dword column), flattened
    const int rg = item / ndw;
    const int sdw = item - rg * ndw;
    if (rg * rows >= g.h + 2 * EDGE)
        return;
    const uint32_t bq = strips[strip_off + sdw].base_q;
    const uint32_t sh = (bq >> 16) & 3u;  // byte shift of the 12-byte window (0..3)
    const uint4 sel = sels[strip_off + sdw], wt = wts[strip_off + sdw];
    const uint32_t selv[4] = {sel.x, sel.y, sel.z, sel.w}, wv[4] = {wt.x, wt.y, wt.z, wt.w};
    // wave-uniform frame base + 32-bit offsets
    uint8_t *fb = pyr + (size_t)f * frame_pyr;
    const uint8_t *sb = DIRECT ? s0.p + (size_t)f * s0.frame_stride : fb;  // source plane: padded level l-1, or the image itself
    const int spitch = DIRECT ? (int)s0.pitch : gs.pitch;
    // (source pixel row 0 is row EDGE of a padded source plane: folded into the base offset)
    const uint32_t base = (DIRECT ? 0u : (uint32_t)gs.plane_off + (uint32_t)EDGE * (uint32_t)gs.pitch) + (bq & 0xFFFFu);
    // DIRECT, a window at the end of an image row: it is placed on the row's LAST 12 bytes (nothing past the row is read)
    // and the 8-byte pair is taken from its second and third dword instead of the first and second
    const bool edge = DIRECT && ((bq >> 18) & 1u) != 0u;
    const uint32_t dst = (uint32_t)g.plane_off + (uint32_t)sdw * 4u;
    // the row table entries of all rows of the item first (one round trip), so that the source loads of a row do
    // not wait for a table load of their own
    YRow yts[PYR_ROWS];
#pragma unroll
    for (int rr = 0; rr < PYR_ROWS; rr++)
        yts[rr] = yrows[g.yrow_off + min(rg * rows + rr, g.h + 2 * EDGE - 1)];
    // consecutive output rows share a source row (row y's lower tap row is usually row y+1's upper one): its horizontal
    // interpolation is kept instead of being loaded and computed again
    int kept_row = -1, kept[4] = {0, 0, 0, 0};
#pragma unroll
    for (int rr = 0; rr < PYR_ROWS; rr++) {
        const int py = rg * rows + rr;
        if (rr >= rows || py >= g.h + 2 * EDGE)
            break;
        const YRow yt = yts[rr];
        const uint32_t *S1 = reinterpret_cast<const uint32_t *>(sb + (base + rowoff(yt.sy1, spitch)));
        uint32_t c0 = S1[0], c1 = S1[1];
        const uint32_t c2 = S1[2];
        if (DIRECT) {
            c0 = edge ? c1 : c0;
            c1 = edge ? c2 : c1;
        }
        int t0[4];
        if (!REUSE || yt.sy0 != kept_row) {
            const uint32_t *S0 = reinterpret_cast<const uint32_t *>(sb + (base + rowoff(yt.sy0, spitch)));
            uint32_t a0 = S0[0], a1 = S0[1];
            const uint32_t a2 = S0[2];
            if (DIRECT) {
                a0 = edge ? a1 : a0;
                a1 = edge ? a2 : a1;
            }
            const uint32_t lo = __builtin_amdgcn_alignbyte(a1, a0, sh), hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t p0 = __builtin_amdgcn_perm(hi, lo, selv[k]);
                t0[k] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, p0), __builtin_bit_cast(us2, wv[k]), 0u, false);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                t0[k] = kept[k];
        }
        uint32_t v = 0;
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, sh), hi1 = __builtin_amdgcn_alignbyte(c2, c1, sh);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t p1 = __builtin_amdgcn_perm(hi1, lo1, selv[k]);
            const int t1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, p1), __builtin_bit_cast(us2, wv[k]), 0u, false);
            const int o = ((__mul24(yt.b0, t0[k] >> 4) >> 16) + (__mul24(yt.b1, t1 >> 4) >> 16) + 2) >> 2;  // 12 x 15 bits
            v |= (uint32_t)o << (8 * k);  // o <= 255: the two weights add up to 2048 (+-1) and t >> 4 <= 255 * 128
            kept[k] = t1;
        }
        kept_row = yt.sy1;
        *reinterpret_cast<uint32_t *>(fb + (dst + rowoff(py, g.pitch))) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: FAST-9/16 with NMS and per-cell threshold fallback
//     (ORBextractor.cc:789-829 calling cv::FAST(sub, kps, th, true), A4)
//
// One kernel, k_fast_detect (below).  The corner score s(x,y) = max over the 16 arcs of 9 contiguous ring pixels of
// the minimum |centre - ring| (dark and bright polarity), clamped to [0,255]: a pixel is a FAST corner for threshold t
// <=>  s > t, and cv::FAST's cornerScore is s-1, so one score serves both thresholds.  Register-only streaming like
// k_blur: a thread owns 4 pixels x the rows of a band, keeps a 7-row window in registers -- every row unpacked once
// with v_perm_b32 into the eight packed-16-bit forms its seven uses need (RowU) -- and runs the min/max network with
// three-input packed extrema
// (v_pk_maximum3_f16 / v_pk_minimum3_f16 on the integer bit patterns: two pixels per instruction, no divergence).
// ---------------------------------------------------------------------------------------------
constexpr int MAX_CELL = 66;  // max cell interior edge

typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 as_pk(uint32_t u) { return __builtin_bit_cast(pk16, u); }
__device__ __forceinline__ uint32_t as_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 pkmin(pk16 a, pk16 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ pk16 pkmax(pk16 a, pk16 b) { return __builtin_elementwise_max(a, b); }

// score of two pixels at once from their centre c and 16 packed ring values p[k] (all zero-extended
// bytes).  With d = c - p:  max over arcs of min(d) = c - min over arcs of max(p)  (dark arcs) and
// max over arcs of min(-d) = max over arcs of min(p) - c  (bright arcs), so the networks run on the raw
// ring values and the 16 subtractions disappear.
// gfx950 has packed THREE-input extrema only for f16 (v_pk_maximum3_f16 / v_pk_minimum3_f16).  A 16-bit lane
// holding an integer 0..255 is, read as f16, the subnormal n * 2^-24: positive f16 bit patterns order like the
// integers they are, f16 denormals are never flushed on this target, and no value is a NaN, so the f16 extrema of
// the bit patterns ARE the integer extrema.  Windows are taken in pairs: with M8[j] = ext(v[j..j+7]),
//   ext'(W9[j-1], W9[j]) = ext'(ext(v[j-1], M8[j]), ext(M8[j], v[j+8])) = ext(M8[j], ext'(v[j-1], v[j+8]))
// (ext = the window extremum, ext' = the opposite extremum taken over the windows), so only the 8 odd j are needed:
//   M2[j] = ext(v[j], v[j+1]), M4[j] = ext(M2[j], M2[j+2])      (8 + 8 ops, j odd)
//   E[j]  = ext'(v[j-1], v[j+8])                                 (8 ops)
//   T[j]  = ext3(M4[j], M4[j+4], E[j])                           (8 ops)
// and the reduction over the 8 pairs 4 more: 36 packed ops per polarity (40 with one 3-input window per arc, 59
// with the two-input van Herk prefix/suffix scheme before that, 79 at the start).
typedef _Float16 pkh __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pkh as_h(pk16 v) { return __builtin_bit_cast(pkh, v); }
__device__ __forceinline__ pk16 h_as_pk(pkh v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ pkh hmax3(pkh a, pkh b, pkh c)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
__device__ __forceinline__ pkh hmin3(pkh a, pkh b, pkh c)
{
    return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c);
}

__device__ __forceinline__ pk16 fast_score_pk(pk16 c, const pk16 p[16])
{
    pkh v[16], m2[8], m4[8], t[8];
#pragma unroll
    for (int k = 0; k < 16; k++)
        v[k] = as_h(p[k]);
    // ---- dark arcs: A = min over windows of the window maximum (index i stands for j = 2 i + 1)
#pragma unroll
    for (int i = 0; i < 8; i++)
        m2[i] = __builtin_elementwise_maximum(v[2 * i + 1], v[(2 * i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 8; i++)
        m4[i] = __builtin_elementwise_maximum(m2[i], m2[(i + 1) & 7]);
#pragma unroll
    for (int i = 0; i < 8; i++)
        t[i] = hmax3(m4[i], m4[(i + 2) & 7], __builtin_elementwise_minimum(v[2 * i], v[(2 * i + 9) & 15]));
    const pkh A = __builtin_elementwise_minimum(hmin3(hmin3(t[0], t[1], t[2]), hmin3(t[3], t[4], t[5]), t[6]), t[7]);
    // ---- bright arcs: B = max over windows of the window minimum
#pragma unroll
    for (int i = 0; i < 8; i++)
        m2[i] = __builtin_elementwise_minimum(v[2 * i + 1], v[(2 * i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 8; i++)
        m4[i] = __builtin_elementwise_minimum(m2[i], m2[(i + 1) & 7]);
#pragma unroll
    for (int i = 0; i < 8; i++)
        t[i] = hmin3(m4[i], m4[(i + 2) & 7], __builtin_elementwise_maximum(v[2 * i], v[(2 * i + 9) & 15]));
    const pkh B = __builtin_elementwise_maximum(hmax3(hmax3(t[0], t[1], t[2]), hmax3(t[3], t[4], t[5]), t[6]), t[7]);
    // max(c - A, B - c, 0) with saturating unsigned subtractions (v_pk_sub_u16 clamp): 3 operations instead of 4
    typedef unsigned short pku16 __attribute__((ext_vector_type(2)));
    const pku16 cu = __builtin_bit_cast(pku16, c);
    const pku16 dk = __builtin_elementwise_sub_sat(cu, __builtin_bit_cast(pku16, A));
    const pku16 br = __builtin_elementwise_sub_sat(__builtin_bit_cast(pku16, B), cu);
    return __builtin_bit_cast(pk16, __builtin_elementwise_max(dk, br));  // values are in [0,255]
}

// v_perm selectors that zero-extend bytes (o, o+2) resp. (o+1, o+3) of the 8-byte pair {lo,hi}
#define ORBGPU_SEL_EVEN(o) ((uint32_t)(o) | 0x0c00u | ((uint32_t)((o) + 2) << 16) | 0x0c000000u)
#define ORBGPU_SEL_ODD(o) ((uint32_t)((o) + 1) | 0x0c00u | ((uint32_t)((o) + 3) << 16) | 0x0c000000u)

struct Row3 {
    uint32_t d[3];  // 12 bytes of one padded row: columns 4c-4 .. 4c+7
};

// A row of the 7-row window in unpacked form: f[o-1] = (byte o, byte o+2) of the 12-byte window as packed 16-bit
// lanes, o = 1..8.  The even pixel pair (px0, px2) at horizontal offset dx is f at o = 4 + dx, the odd pair (px1, px3)
// at o = 5 + dx.  A row serves seven steps (as row y+3 down to y-3) and its eight forms are used 30 times in all:
// unpacking once per row costs 8 v_perm_b32 per step instead of 30 -- at 56 registers for the window instead of 21.
struct RowU {
    uint32_t f[8];
};
__device__ __forceinline__ void unpack_row(const Row3 &R, RowU &U)
{
#pragma unroll
    for (int o = 1; o <= 8; o++) {
        const int q = o >> 2;
        U.f[o - 1] = __builtin_amdgcn_perm(R.d[q < 2 ? q + 1 : 2], R.d[q], ORBGPU_SEL_EVEN(o & 3));
    }
}
template <int DX, int ODD> __device__ __forceinline__ pk16 ring_half(const RowU &R)
{
    return as_pk(R.f[4 + DX + ODD - 1]);
}

// score of one pixel pair of the strip in the row whose window is r3 (r0..r6 = rows y-3..y+3)
template <int ODD>
__device__ __forceinline__ pk16 fast_score_half(const RowU &r0, const RowU &r1, const RowU &r2, const RowU &r3,
                                                const RowU &r4, const RowU &r5, const RowU &r6)
{
    pk16 p[16];
    const pk16 c = ring_half<0, ODD>(r3);
    // ring order A4: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    p[0] = ring_half<0, ODD>(r6);
    p[1] = ring_half<1, ODD>(r6);
    p[2] = ring_half<2, ODD>(r5);
    p[3] = ring_half<3, ODD>(r4);
    p[4] = ring_half<3, ODD>(r3);
    p[5] = ring_half<3, ODD>(r2);
    p[6] = ring_half<2, ODD>(r1);
    p[7] = ring_half<1, ODD>(r0);
    p[8] = ring_half<0, ODD>(r0);
    p[9] = ring_half<-1, ODD>(r0);
    p[10] = ring_half<-2, ODD>(r1);
    p[11] = ring_half<-3, ODD>(r2);
    p[12] = ring_half<-3, ODD>(r3);
    p[13] = ring_half<-3, ODD>(r4);
    p[14] = ring_half<-2, ODD>(r5);
    p[15] = ring_half<-1, ODD>(r6);
    return fast_score_pk(c, p);
}

// Exact upper bound of the corner score from the four compass pixels (0,3) (3,0) (0,-3) (-3,0): an arc of 9 contiguous
// ring pixels contains two ADJACENT compass pixels, so
//   bright score <= max over adjacent pairs of min(p_a, p_b) - c,   dark score <= c - min over adjacent pairs of max(p_a, p_b).
// Returns the sign bits of the lanes' two pixels that MAY exceed the threshold (0 = neither can be a corner at t, whose
// score may then be written as 0 without changing any output: a non-corner's score is only ever compared as "<= t").
// 18 packed operations against the 72 of the two score networks of a pixel pair.
template <int ODD>
__device__ __forceinline__ uint32_t fast_maybe_half(const RowU &r0, const RowU &r3, const RowU &r6, pkh tpk)
{
    const pkh c = as_h(ring_half<0, ODD>(r3));
    const pkh p0 = as_h(ring_half<0, ODD>(r6)), p4 = as_h(ring_half<3, ODD>(r3));
    const pkh p8 = as_h(ring_half<0, ODD>(r0)), p12 = as_h(ring_half<-3, ODD>(r3));
    const pkh bm = hmax3(__builtin_elementwise_minimum(p0, p4), __builtin_elementwise_minimum(p4, p8),
                         __builtin_elementwise_maximum(__builtin_elementwise_minimum(p8, p12), __builtin_elementwise_minimum(p12, p0)));
    const pkh dm = hmin3(__builtin_elementwise_maximum(p0, p4), __builtin_elementwise_maximum(p4, p8),
                         __builtin_elementwise_minimum(__builtin_elementwise_maximum(p8, p12), __builtin_elementwise_maximum(p12, p0)));
    const pk16 ci = h_as_pk(c), ti = h_as_pk(tpk);
    const uint32_t eb = as_u32((ci + ti) - h_as_pk(bm));  // negative <=> bm - c > t
    const uint32_t ed = as_u32(h_as_pk(dm) - (ci - ti));  // negative <=> c - dm > t
    return (eb | ed) & 0x80008000u;
}

struct StripGeom {  // strips of all levels, flattened (k_blur)
    int first[ORBGPU_MAX_LEVELS + 1];  // first strip index of each level
    int nsx[ORBGPU_MAX_LEVELS];        // strips per row of strips
    int nlevels;
};


// ---------------------------------------------------------------------------------------------
// k_fast_detect: corner score, in-cell 3x3 non-maximum suppression and key emission in one pass over the pyramid --
// the score map never reaches memory.
//
// Work item = (level, band, aligned dword column): a band is one row of cells (its interior rows), so a thread scores
// the 4 pixels of its column for every row of the band and, one row behind, decides which
// of them survive.  cv::FAST runs on a cell's sub-image, so a neighbour outside the cell interior counts as 0: rows
// outside the band are zero by construction, and the columns where a cell starts / ends are known per dword column
// (ColumnInfo) and turn into four constant keep-masks per thread.  With S = the row's scores as packed pairs
// E = (px0, px2), O = (px1, px3), the neighbour pairs come from the adjacent lanes (the strips of a wave are
// consecutive columns; lanes 0 and 63 of a wave are halo columns that score but do not emit), and
//     T[x] = max(s[x-1], s[x], s[x+1]),  H[x] = max(s[x-1], s[x+1])      (masked neighbours)
//     survivor(y, x)  <=>  s > max(t, 1)  and  s > max(H(y), T(y-1), T(y+1))
// is one NMS for both thresholds: a survivor at the lower threshold with s > t_hi is also a survivor at t_hi (a
// neighbour with s_n <= t_hi cannot beat it), so cv::FAST(iniThFAST) of a cell is the subset of its survivors with
// s > iniThFAST and the fallback to minThFAST (ORBextractor.cc:809-816) is a per-cell choice k_quadtree makes from two
// counts.  A row with survivors is queued in LDS as one record (4 score bytes, 4 survivor flags, row, lane) -- 8
// instructions in the scoring loop -- and the wave turns its records into keys when its band is done: per-wave
// per-cell counts in LDS, one global atomic per (wave, cell) on the cell's counter (low half = survivors above the
// lower threshold, high half = those above iniThFAST) to reserve slots, positions handed out from LDS.  The keys of
// a cell land in no particular order: nothing downstream depends on it (k_quadtree's "first maximum" rule rebuilds
// vToDistributeKeys order from the key itself).
// Measured (B = 256, 640x480): 375 us against 324 + 158 us for the score-map / per-cell pair it replaces, and
// 1.8 MB per frame less HBM traffic (143 VGPRs, three waves per SIMD, every window row unpacked once; 390 us with raw
// rows in 128 VGPRs and 30 v_perm_b32 per step instead of 8); 338 us with the paired-window score network below, which
// also fits four waves per SIMD (121 VGPRs).
// ---------------------------------------------------------------------------------------------
constexpr int FD_OWN = 62;    // columns a wave owns (lanes 1..62)
constexpr int FD_QCAP = 512;  // row records a wave can queue before it falls back to emitting them directly
constexpr int FD_CELLS = 32;  // cells a wave's 64 columns can touch (consecutive ids)
constexpr int FD_EARLY_HOLD = 3;  // k_fast_detect<true>: rows scored without the early-out test after a test that found work

struct DetectGeom {
    int first[ORBGPU_MAX_LEVELS + 1];  // first flat column strip of each level (bands x dword columns)
    int nsx[ORBGPU_MAX_LEVELS];        // dword columns per band: padded dwords 9 .. (w-1)/4
    int ncols[ORBGPU_MAX_LEVELS];      // cell columns of the level's cell grid (skipped ones excluded)
    int ctab_off[ORBGPU_MAX_LEVELS];   // first ColumnInfo of the level
    int xfirst[ORBGPU_MAX_LEVELS];     // image column of pixel 0 of the level's first strip: 17 for a padded plane (its aligned
                                       // dword 9), 16 for level 0 read directly from the image (Src0: aligned to the image)
    int nlevels;
};

struct ColumnInfo {
    uint32_t cellj;  // byte p: cell column of pixel p of the dword (0xFF: outside every cell interior)
    uint32_t flags;  // bit p: pixel p lies in a cell interior; bit 4+p: first column of its cell; bit 8+p: last column
};

struct DetectLane {  // what the key emission needs to know about the lane that queued a record
    int xrel, yrel;  // key coordinates of pixel 0 / row 0 of the strip (relative to minBorderX/Y)
    int cell_base;   // cell id of the band's first cell (frame-relative)
    uint32_t cellj;
};

// keys of one queued record: f(pixel p, score s, cell id) for every survivor flag of the record
template <typename F>
__device__ __forceinline__ void detect_record(uint32_t sc4, uint32_t rec, const DetectLane &dl, F &&fn)
{
    uint32_t m = rec & 0x01010101u;
    while (m) {
        const int b = __ffs((int)m) - 1;  // 8 p
        m &= m - 1u;
        fn(b >> 3, (int)((sc4 >> b) & 255u), dl.cell_base + (int)((dl.cellj >> b) & 255u));
    }
}

// EARLY: skip the score network of a half row (a pixel pair per lane) when the compass bound clears it for the whole wave --
// exact, pays on images with flat regions (orbgpu_extractor_set_fast_early_out); off, the kernel is the round-3 one.
template <bool EARLY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fast_detect(const uint8_t *__restrict__ pyr, size_t frame_pyr,
                                                     const LevelGeom *__restrict__ geom, DetectGeom dg,
                                                     const ColumnInfo *__restrict__ ctab,
                                                     const CellDesc *__restrict__ cells, int ncells_total,
                                                     uint32_t *__restrict__ slots, size_t frame_slots,
                                                     int *__restrict__ cell_cnt, int t_lo, int t_ini, int qcap, Src0 s0)
{
    __shared__ uint2 s_q[4][FD_QCAP];
    __shared__ DetectLane s_lane[4][64];
    __shared__ int s_cc[4][FD_CELLS], s_cid[4][FD_CELLS];
    int bx, f;
    xcd_frame_block(bx, f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < FD_CELLS)
        s_cc[wave][lane] = 0;
    const int q = (bx * 4 + wave) * FD_OWN + lane - 1;
    const bool in_range = q >= 0 && q < dg.first[dg.nlevels];
    int level = 0;
#pragma unroll
    for (int l = 1; l < ORBGPU_MAX_LEVELS; l++)
        level += (l < dg.nlevels && q >= dg.first[l]) ? 1 : 0;
    const LevelGeom g = geom[level];
    const int local = max(q, 0) - dg.first[level];
    const int band = local / dg.nsx[level], sx = local - band * dg.nsx[level];
    const int xs = dg.xfirst[level] + 4 * sx;  // image column of the strip's pixel 0
    const int y0 = EDGE + band * g.hcell;      // first row of the band (image coordinates)
    // The wave's plane: the padded pyramid, or -- direct mode, level 0 -- the caller's image.  In direct mode the host ends
    // level 0's strips on a wave boundary (DetectGeom::first[1] is a multiple of FD_OWN), so the choice is wave-uniform for
    // the owned lanes: a scalar base + 32-bit offsets, as before.  A halo lane that belongs to the other side of that
    // boundary sits out (rows = 0): the neighbour it would deliver is outside every cell and masked anyway.
    const bool wave_dir0 = s0.direct != 0 && __builtin_amdgcn_readlane(level, 1) == 0;
    const bool dir0 = s0.direct != 0 && level == 0;
    const int rows = max(in_range && dir0 == wave_dir0 ? min(g.hcell, g.h - EDGE - y0) : 0, 0);  // (0 for the padding strips)
    // The row loop is WAVE-uniform: every lane runs as many steps as the wave's tallest band (a wave can straddle bands and
    // levels of different height).  A lane past the end of its own band keeps scoring -- on the last row it may fetch
    // (fetch_max; a lane without rows reads the first rows of the plane) -- and its scores are masked to zero, which is what
    // "rows outside the band count as 0" asks for anyway.  Uniform control flow costs no copies at branch joins and lets
    // the record queue count in a scalar register.
    int wrows = rows;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        wrows = max(wrows, __shfl_xor(wrows, off, 64));
    wrows = __builtin_amdgcn_readfirstlane(wrows);
    const int fetch_max = rows + 5;  // the last input row the lane's band needs
    const uint8_t *fsrc = wave_dir0 ? s0.p + (size_t)f * s0.frame_stride : pyr + (size_t)f * frame_pyr;
    const int lpitch = dir0 ? (int)s0.pitch : g.pitch;
    const uint32_t src0 = rows == 0 ? 0u
                          : dir0    ? (uint32_t)(y0 - 3) * (uint32_t)lpitch + (uint32_t)(xs - 4)
                                    : (uint32_t)g.plane_off + (uint32_t)(y0 + EDGE - 3) * (uint32_t)lpitch + (uint32_t)(xs + EDGE - 4);
    uint32_t *fslots = slots + (size_t)f * frame_slots;
    int *fcnt = cell_cnt + (size_t)f * (ncells_total + ORBGPU_MAX_LEVELS);  // per-cell counters ...
    int *lcnt = fcnt + ncells_total;                                         // ... then the stored keys of every level

    const ColumnInfo ci = in_range ? ctab[dg.ctab_off[level] + sx] : ColumnInfo{0xFFFFFFFFu, 0u};
    {
        DetectLane dl;
        dl.xrel = xs - BORDER0;
        dl.yrel = y0 - BORDER0;
        dl.cell_base = g.cell_first + band * dg.ncols[level];
        dl.cellj = ci.cellj;
        s_lane[wave][lane] = dl;
    }
    // neighbour pairs of E = (px0, px2) and O = (px1, px3).  A neighbour outside the pixel's cell counts as 0: for the
    // pairs that come from the adjacent lanes the v_perm_b32 selector that assembles them zeroes the half word
    // (selector byte 0x0c), the in-lane pairs are masked.
    auto flag = [&](int bit) { return ((ci.flags >> bit) & 1u) != 0u; };
    const uint32_t selLE = (flag(4) ? 0x0c0cu : 0x0302u) | (flag(6) ? 0x0c0c0000u : 0x05040000u);   // (px-1, px1) of {so, left so}
    const uint32_t selRO = (flag(9) ? 0x0c0cu : 0x0302u) | (flag(11) ? 0x0c0c0000u : 0x05040000u);  // (px2, px4) of {right se, se}
    const uint32_t kRE = (flag(8) ? 0u : 0x0000FFFFu) | (flag(10) ? 0u : 0xFFFF0000u);              // (px1, px3) right of E
    const uint32_t kLO = (flag(5) ? 0u : 0x0000FFFFu) | (flag(7) ? 0u : 0xFFFF0000u);               // (px0, px2) left of O
    // survivor sign bits only count inside a cell interior and in the columns this wave owns
    const bool own = lane >= 1 && lane <= FD_OWN;
    const uint32_t vE = own ? ((flag(0) ? 0x8000u : 0u) | (flag(2) ? 0x80000000u : 0u)) : 0u;
    const uint32_t vO = own ? ((flag(1) ? 0x8000u : 0u) | (flag(3) ? 0x80000000u : 0u)) : 0u;
    const uint32_t lane_tag = (uint32_t)lane << 9;
    const uint32_t tl = (uint32_t)max(t_lo, 1);
    const pkh tpk = __builtin_bit_cast(pkh, tl | (tl << 16));

    // NMS state, one row behind the scores: T of rows k-2 and k-1; H of row k-1 with the threshold folded in
    // (H = max(left, right, t)); S of row k-1 (packed pairs)
    uint32_t TpE = 0u, TpO = 0u, TcE = 0u, TcO = 0u, HcE = 0u, HcO = 0u, ScE = 0u, ScO = 0u;
    int qn = 0;  // records the wave has queued: wave-uniform, the row loop's control flow is
    auto finish_row = [&](uint32_t TnE, uint32_t TnO, int row) {
        const pkh nbE = hmax3(__builtin_bit_cast(pkh, HcE), __builtin_bit_cast(pkh, TpE), __builtin_bit_cast(pkh, TnE));
        const pkh nbO = hmax3(__builtin_bit_cast(pkh, HcO), __builtin_bit_cast(pkh, TpO), __builtin_bit_cast(pkh, TnO));
        // x > y  <=>  sign(y - x) for values in [0,255]: survivor <=> s > max(every neighbour, t)
        const uint32_t sE = as_u32(h_as_pk(nbE) - as_pk(ScE)) & vE;
        const uint32_t sO = as_u32(h_as_pk(nbO) - as_pk(ScO)) & vO;
        // a lane's slot in the wave's queue: the fill count + the survivor lanes below it (the compare that feeds the branch
        // is the ballot; until round 4 an LDS atomic with return that nearly every row of this stream waited for)
        const unsigned long long srv = __ballot((sE | sO) != 0u);
        if (sE | sO) {
            const uint32_t sg = (sE >> 15) | (sO >> 7);  // flag bytes in pixel order: px0 bit 0, px1 bit 8, px2 bit 16, px3 bit 24
            const uint32_t sc4 = __builtin_amdgcn_perm(ScO, ScE, 0x06020400u);
            const uint32_t rec = sg | ((uint32_t)row << 1) | lane_tag;
            const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(srv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)srv, 0u));
            if (slot < qcap)
                s_q[wave][slot] = make_uint2(sc4, rec);
            else
                detect_record(sc4, rec, s_lane[wave][lane], [&](int p, int sc, int cell) {  // queue full: two atomics per key
                    const DetectLane &dl = s_lane[wave][lane];
                    atomicAdd(&fcnt[cell], 1 + (sc > t_ini ? 65536 : 0));
                    fslots[geom[level].slot_off + atomicAdd(&lcnt[level], 1)] = pack_key(dl.xrel + p, dl.yrel + row, sc - 1);
                });
        }
        qn += (int)__popcll(srv);
    };
    auto nms_row = [&](uint32_t se, uint32_t so, int k) {
        // wave-wide lane shifts (DPP wave_shr:1 / wave_shl:1): an inactive or missing source lane yields 0
        const uint32_t oL = __builtin_amdgcn_update_dpp(0u, so, 0x138, 0xf, 0xf, true);  // left column's (px1, px3)
        const uint32_t eR = __builtin_amdgcn_update_dpp(0u, se, 0x130, 0xf, 0xf, true);  // right column's (px0, px2)
        const uint32_t le = __builtin_amdgcn_perm(so, oL, selLE);                    // (px-1, px1)
        const uint32_t re = so & kRE;                                                 // (px1, px3)
        const uint32_t lo = se & kLO;                                                 // (px0, px2)
        const uint32_t ro = __builtin_amdgcn_perm(eR, se, selRO);                    // (px2, px4)
        const uint32_t TnE = __builtin_bit_cast(uint32_t, hmax3(__builtin_bit_cast(pkh, le), __builtin_bit_cast(pkh, se), __builtin_bit_cast(pkh, re)));
        const uint32_t TnO = __builtin_bit_cast(uint32_t, hmax3(__builtin_bit_cast(pkh, lo), __builtin_bit_cast(pkh, so), __builtin_bit_cast(pkh, ro)));
        finish_row(TnE, TnO, k - 1);
        TpE = TcE;
        TpO = TcO;
        TcE = TnE;
        TcO = TnO;
        HcE = __builtin_bit_cast(uint32_t, hmax3(__builtin_bit_cast(pkh, le), __builtin_bit_cast(pkh, re), tpk));
        HcO = __builtin_bit_cast(uint32_t, hmax3(__builtin_bit_cast(pkh, lo), __builtin_bit_cast(pkh, ro), tpk));
        ScE = se;
        ScO = so;
    };

    RowU r0, r1, r2, r3, r4, r5, r6;
#define FD_FETCH(row)                                                                                        \
    {                                                                                                        \
        const uint32_t *qq = reinterpret_cast<const uint32_t *>(fsrc + (src0 + rowoff(min((int)(row), fetch_max), lpitch))); \
        nx.d[0] = qq[0];                                                                                     \
        nx.d[1] = qq[1];                                                                                     \
        nx.d[2] = qq[2];                                                                                     \
    }
#define FD_LOAD(R, row)                                                                                      \
    {                                                                                                        \
        FD_FETCH(row)                                                                                        \
        unpack_row(nx, R);                                                                                   \
    }
#define FD_STEP(A, B, C, D, E, F, G, k)                                                                      \
    if ((k) < wrows) {                                                                                       \
        unpack_row(nx, G);                                                                                   \
        FD_FETCH((k) + 7)                                                                                    \
        const uint32_t lm = (uint32_t)(((k) - rows) >> 31);  /* all ones while the lane's own band lasts */   \
        pk16 se = {0, 0}, so = {0, 0};                                                                       \
        bool runE = true, runO = true;                                                                       \
        if (EARLY) {                                                                                         \
            if (hold == 0) {                                                                                 \
                runE = __builtin_amdgcn_ballot_w64((fast_maybe_half<0>(A, D, G, tpk) & lm) != 0u) != 0ull;   \
                runO = __builtin_amdgcn_ballot_w64((fast_maybe_half<1>(A, D, G, tpk) & lm) != 0u) != 0ull;   \
                hold = (runE && runO) ? FD_EARLY_HOLD : 0;                                                   \
            } else                                                                                           \
                hold--;                                                                                      \
        }                                                                                                    \
        if (runE)                                                                                            \
            se = fast_score_half<0>(A, B, C, D, E, F, G);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (runO)                                                                                            \
            so = fast_score_half<1>(A, B, C, D, E, F, G);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        nms_row(as_u32(se) & lm, as_u32(so) & lm, (k));                                                      \
    }
    // EARLY: rows to go before the bound is evaluated again -- after a row whose segment was NOT clear the next
    // FD_EARLY_HOLD rows run the network untested (textured regions pay a quarter of the test), after a clear one every row is
    // tested (flat regions are contiguous).  Only the cost depends on it: the bound is exact whenever it is used.
    int hold = 0;
    Row3 nx;  // next input row (raw), fetched one step ahead
    nx.d[0] = nx.d[1] = nx.d[2] = 0u;
#pragma unroll
    for (int o = 0; o < 8; o++)
        r0.f[o] = r1.f[o] = r2.f[o] = r3.f[o] = r4.f[o] = r5.f[o] = r6.f[o] = 0u;
    if (wrows > 0) {
        FD_LOAD(r0, 0) FD_LOAD(r1, 1) FD_LOAD(r2, 2) FD_LOAD(r3, 3) FD_LOAD(r4, 4) FD_LOAD(r5, 5)
        FD_FETCH(6)
    }
#pragma unroll 1
    for (int k = 0; k < wrows; k += 7) {
        FD_STEP(r0, r1, r2, r3, r4, r5, r6, k)
        FD_STEP(r1, r2, r3, r4, r5, r6, r0, k + 1)
        FD_STEP(r2, r3, r4, r5, r6, r0, r1, k + 2)
        FD_STEP(r3, r4, r5, r6, r0, r1, r2, k + 3)
        FD_STEP(r4, r5, r6, r0, r1, r2, r3, k + 4)
        FD_STEP(r5, r6, r0, r1, r2, r3, r4, k + 5)
        FD_STEP(r6, r0, r1, r2, r3, r4, r5, k + 6)
    }
#undef FD_STEP
#undef FD_LOAD
#undef FD_FETCH
    if (wrows > 0)
        finish_row(0u, 0u, wrows - 1);  // the row below the band counts as 0 (a shorter band was finished inside the loop)

    // ---- the wave's records become keys.  The cells a wave touches are a contiguous range of at most FD_CELLS ids
    //      (host-checked), so id mod FD_CELLS addresses per-wave counters in LDS.  Per (wave, cell): one global atomic
    //      on the cell's counter (survivors | those above iniThFAST << 16 -- all the quadtree needs to pick the cell's
    //      threshold).  Per (wave, level): one global atomic that reserves a range of the level's key array; the keys of
    //      a level are stored densely, in no particular order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nrec = min(qn, qcap);
    for (int e = lane; e < nrec; e += 64) {
        const uint2 r = s_q[wave][e];
        detect_record(r.x, r.y, s_lane[wave][(r.y >> 9) & 63u], [&](int, int sc, int cell) {
            atomicAdd(&s_cc[wave][cell & (FD_CELLS - 1)], 1 + (sc > t_ini ? 65536 : 0));
            s_cid[wave][cell & (FD_CELLS - 1)] = cell;
        });
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int c = lane < FD_CELLS ? s_cc[wave][lane] : 0;
        const int mine = c & 0xFFFF;
        // the items of a wave are consecutive in (level, band, column) order: its cells belong to the levels of its
        // first .. last owned lane (one level, two at a boundary, more only for tiny images); a cell belongs to the
        // last of them whose first cell id is not above its own
        const int lvmin = __builtin_amdgcn_readlane(level, 1), lvmax = __builtin_amdgcn_readlane(level, FD_OWN);
        int lv = -1;
        if (c) {
            const int cell = s_cid[wave][lane];
            atomicAdd(&fcnt[cell], c);
            lv = lvmin;
            for (int L = lvmin + 1; L <= lvmax; L++)
                lv += cell >= geom[L].cell_first ? 1 : 0;
        }
        for (int L = lvmin; L <= lvmax; L++) {
            const int v = lv == L ? mine : 0;
            int incl = v;
#pragma unroll
            for (int off = 1; off < FD_CELLS; off <<= 1) {  // the counts live in lanes 0 .. FD_CELLS-1
                const int o = __shfl_up(incl, off, 64);
                incl += lane >= off ? o : 0;
            }
            const int total = __shfl(incl, FD_CELLS - 1, 64);
            int base = 0;
            if (lane == 0 && total)
                base = atomicAdd(&lcnt[L], total);
            base = __shfl(base, 0, 64);
            if (lv == L)
                s_cc[wave][lane] = geom[L].slot_off + base + incl - v;  // first position of this wave's keys of the cell
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < nrec; e += 64) {
        const uint2 r = s_q[wave][e];
        const DetectLane dl = s_lane[wave][(r.y >> 9) & 63u];
        const int row = (int)((r.y >> 1) & 127u);
        detect_record(r.x, r.y, dl, [&](int p, int sc, int cell) {
            const int pos = atomicAdd(&s_cc[wave][cell & (FD_CELLS - 1)], 1);
            fslots[pos] = pack_key(dl.xrel + p, dl.yrel + row, sc - 1);
        });
    }
}

// ---------------------------------------------------------------------------------------------
// K3: quadtree distribution                       (DistributeOctTree, ORBextractor.cc:539-763)
//
// One workgroup per (level, frame).  The std::list of the reference is kept as arrays in list
// order in LDS (ping-pong A/B); keys live in HBM (L2-resident) with their node id.  A pass
// expands a set E of nodes in a processing order:
//   phase 1 (:591-667): E = every node with >1 keys, processed front-to-back;
//   phase 2 (:674-736): nodes with >1 keys sorted by (size, creation) descending, E = the shortest
//            prefix that brings the list to >= N nodes (the reference's `break`).
// Children are created in processing order (n1..n4, empty ones skipped) and push_front'ed, so the
// new list is reverse(creation order) followed by the untouched nodes in their old order.
// Tie-break of the (size, pointer) sort at :684: the node created later compares greater (the
// reference compares heap addresses, which is not deterministic; the oracle uses the same rule).
// ---------------------------------------------------------------------------------------------
struct QtShared {
    int *ccnt_next;     // [ncap*4] child key counts of the list being built (fused relabel + count)
    int *ccnt;          // [ncap*4] child key counts (also: cell scan, best-key scratch)
    int *sa, *sb;       // scan scratch
    uint16_t *cpos;     // [ncap*4] new list position of child q
    uint16_t *opos;     // new list position of an untouched node
    uint16_t *order;    // processing order (node ids)
    uint8_t *inE;       // node is expanded in this pass
    uint32_t *soff;     // [cells of the level] slot offset of the cell; bit 31: the cell has corners above iniThFAST
};

// ExtractorNode::DivideNode, ORBextractor.cc:481-484, 513-526: a key goes to child 0..3 by comparing it with the
// node's split point (UL.x + halfX, UL.y + halfY), halfX = ceil((float)(UR.x - UL.x) / 2) = (d + 1) >> 1 for the
// non-negative integer d.  The split point is kept per node as one word (x | y << 16) next to the bounds, so that a
// key sweep reads 4 bytes per node and compares.
__device__ __forceinline__ uint32_t split_of(short4 b)
{
    const int mx = b.x + ((b.z - b.x + 1) >> 1), my = b.y + ((b.w - b.y + 1) >> 1);
    return (uint32_t)mx | ((uint32_t)my << 16);
}
__device__ __forceinline__ int quadrant_at(uint32_t key, uint32_t split)
{
    const int x = key_x(key), y = key_y(key);
    return (x < (int)(split & 0xFFFFu) ? 0 : 1) + (y < (int)(split >> 16) ? 0 : 2);
}

// workgroup size of k_quadtree: a single frame is bound by the latency of its level-0 workgroup (more threads per
// key loop help), a full batch by barrier cost and workgroups per CU (fewer threads help): 155 -> 122 us at B = 256
constexpr int QT_THREADS = 1024, QT_THREADS_SMALL = 512, QT_THREADS_BATCH = 256, QT_BATCH_MIN = 32;
constexpr int QT_LARGE_PIXELS = 700000;  // single frames from about 1024x768 use QT_THREADS (measured: 137 -> 128 us at 1280x960, 51 -> 54 us at 640x480)

// Node lists up to this length are processed by one wave (k_quadtree, sections (2)..(6))
constexpr int QT_SOLO_MAX = 128;
constexpr int QT_LDS_MAX = 160 * 1024 - 1024;  // dynamic LDS k_quadtree may be launched with (the CU has 160 KB)

// LDS traffic of one wave is ordered; this keeps the compiler from moving accesses across the point and drains the queue
__device__ __forceinline__ void qt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In-place exclusive scan of an LDS int array by ONE wave (all 64 lanes call); returns the total.
__device__ int wave_excl_scan(int *a, int n)
{
    const int lane = threadIdx.x & 63;
    const int ipt = (n + 63) / 64;
    const int beg = min(lane * ipt, n), end = min(beg + ipt, n);
    int sum = 0;
    for (int i = beg; i < end; i++)
        sum += a[i];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off, 64);
        if (lane >= off)
            inc += t;
    }
    const int total = __shfl(inc, 63, 64);
    int run = inc - sum;
    qt_wave_sync();  // every lane has read its items before any lane overwrites
    for (int i = beg; i < end; i++) {
        int t = a[i];
        a[i] = run;
        run += t;
    }
    qt_wave_sync();
    return total;
}

// In-place exclusive scan of an LDS int array by the whole workgroup; returns the total.
__device__ int block_excl_scan(int *a, int n, int *s_tmp /*>= 16 ints*/)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const int ipt = (n + nt - 1) / nt;
    const int beg = min(tid * ipt, n), end = min(beg + ipt, n);
    int sum = 0;
    for (int i = beg; i < end; i++)
        sum += a[i];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(inc, off, 64);
        if (lane >= off)
            inc += t;
    }
    if (lane == 63)
        s_tmp[wave] = inc;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < nw; w++) {
        int t = s_tmp[w];
        if (w < wave)
            woff += t;
        total += t;
    }
    int run = woff + inc - sum;
    for (int i = beg; i < end; i++) {
        int t = a[i];
        a[i] = run;
        run += t;
    }
    __syncthreads();
    return total;
}

// counter[target] += 1 for every lane with `valid`; called by all 64 lanes of the wave (no lane masked off).  The keys
// of a level are stored cell by cell, so neighbouring lanes mostly hold key points of the same node: a plain LDS atomic
// takes one turn per lane on the same word (measured: 10 of the 18 us of a sweep over 18 k keys).  Here a run of equal
// targets in consecutive lanes costs one atomic, issued by its first lane with the run length.
__device__ __forceinline__ void count_runs(int *cnt, int target, bool valid)
{
    const int lane = threadIdx.x & 63;
    const int t = valid ? target : -1;
    const int prev = __builtin_amdgcn_update_dpp(-2, t, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);  // lane 0: -2
    const bool head = t != prev;
    const unsigned long long heads = __ballot(head);
    if (head && t >= 0) {
        const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
        atomicAdd(&cnt[t], rest ? __ffsll((long long)rest) : 64 - lane);  // lanes up to the next run
    }
}

constexpr int QT_CELL_ILP = 8;  // ... of one cell in the single-frame compaction
constexpr int QT_ILP = 4;  // keys a thread carries through a sweep side by side (independent LDS / memory chains)

// Single frames only: step 0 of k_quadtree (filter the level's stored keys by the per-cell threshold rule, compact them,
// count them per initial node) and the child counts of the initial nodes (what step 1's key sweep produced) for ALL
// levels of a frame, spread over the device -- the level-0 workgroup of k_quadtree<true> spent 31 of its 102 us on that
// one sweep over ~45 k stored keys of a 1280x960 frame, and 7 us more on step 1's.  Per (frame, level) the kernel
// leaves in `aux`: [0] the number of kept keys, [1 + 4 b + q] the kept keys of initial node (bin) b that fall into its
// quadrant q.  k_quadtree<true> reads and re-arms them.  Batches keep the in-kernel step 0 (one workgroup per (frame,
// level) is already thousands of workgroups there).
constexpr int QT_INI_MAX = 80;                       // initial nodes of a level this path supports (round(width / height))
constexpr int QT_AUX_LEVEL = 1 + 4 * QT_INI_MAX;     // ints per (frame, level)
constexpr int QT_AUX_FRAME = QT_AUX_LEVEL * ORBGPU_MAX_LEVELS;
constexpr int QT_PRE_KEYS = 256 * QT_ILP;            // stored keys a workgroup of the prefilter takes per trip
constexpr int QT_PRE_GRID = 48;                      // workgroups per (frame, level): one trip each up to 49 k stored keys

__global__ __launch_bounds__(256) void k_qt_prefilter(const LevelGeom *__restrict__ geom, int ncells_total,
                                                      const uint32_t *__restrict__ slots, size_t frame_slots,
                                                      const int *__restrict__ cell_cnt, uint32_t *__restrict__ dense_key,
                                                      int *__restrict__ aux, int t_ini)
{
    __shared__ int s_cnt[4 * QT_INI_MAX];
    __shared__ int s_w[4 * QT_ILP], s_base;
    const int level = blockIdx.y, f = blockIdx.z;
    const LevelGeom g = geom[level];
    const int *fcnt = cell_cnt + (size_t)f * (ncells_total + ORBGPU_MAX_LEVELS);
    const int nst = min(fcnt[ncells_total + level], g.slot_cnt);
    if ((int)blockIdx.x * QT_PRE_KEYS >= nst)
        return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_ini = g.n_ini;
    for (int i = tid; i < 4 * n_ini; i += 256)
        s_cnt[i] = 0;
    __syncthreads();
    const int *ccounts = fcnt + g.cell_first;
    const uint32_t *lslots = slots + (size_t)f * frame_slots + g.slot_off;
    uint32_t *dkey = dense_key + (size_t)f * frame_slots + g.slot_off;
    int *laux = aux + (size_t)f * QT_AUX_FRAME + level * QT_AUX_LEVEL;
    // the grid is sized for a typical level (the host cannot know the key counts); a workgroup strides over the tiles
    for (int i_first = blockIdx.x * QT_PRE_KEYS; i_first < nst; i_first += gridDim.x * QT_PRE_KEYS) {
    uint32_t key[QT_ILP];
    int hi[QT_ILP];
#pragma unroll
    for (int u = 0; u < QT_ILP; u++)
        key[u] = lslots[min(i_first + u * 256 + tid, nst - 1)];
#pragma unroll
    for (int u = 0; u < QT_ILP; u++) {
        const uint32_t cj = __umulhi((uint32_t)max(key_x(key[u]) - 3, 0), g.inv_wcell);
        const uint32_t ci = __umulhi((uint32_t)max(key_y(key[u]) - 3, 0), g.inv_hcell);
        hi[u] = ccounts[min((int)(ci * (uint32_t)g.ncols_eff + cj), g.ncells - 1)] >> 16;  // the cell's corners above iniThFAST
    }
    // one range of the level's dense array per workgroup and trip (a same-address global atomic per WAVE serialised
    // 700 of them on level 0's counter: 7 of the kernel's 12 us)
    bool keep[QT_ILP];
    unsigned long long m[QT_ILP];
#pragma unroll
    for (int u = 0; u < QT_ILP; u++) {
        // cornerScore = s - 1:  s > t  <=>  response >= t
        keep[u] = i_first + u * 256 + tid < nst && (!hi[u] || key_resp(key[u]) >= t_ini);
        m[u] = __ballot(keep[u]);
        if (lane == 0)
            s_w[u * 4 + (tid >> 6)] = __popcll(m[u]);
    }
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < 4 * QT_ILP; k++)
            tot += s_w[k];
        s_base = tot ? atomicAdd(&laux[0], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < QT_ILP; u++) {
        int base = s_base;
        for (int k = 0; k < u * 4 + (tid >> 6); k++)
            base += s_w[k];
        if (keep[u])
            dkey[base + __popcll(m[u] & ((1ull << lane) - 1ull))] = key[u];
        int bin = (int)((float)key_x(key[u]) / g.hx);
        bin = min(max(bin, 0), n_ini - 1);
        // the initial node of bin b (ORBextractor.cc:548-557) and the child its split point sends the key to
        const short4 b = make_short4((short)(int)(g.hx * (float)bin), 0, (short)(int)(g.hx * (float)(bin + 1)),
                                     (short)(g.max_by - BORDER0));
        count_runs(s_cnt, bin * 4 + quadrant_at(key[u], split_of(b)), keep[u]);
    }
    __syncthreads();  // s_w is rewritten by the next trip
    }
    __syncthreads();
    for (int i = tid; i < 4 * n_ini; i += 256)
        if (s_cnt[i])
            atomicAdd(&laux[1 + i], s_cnt[i]);
}

template <bool LDS_KEYS>  // keep the level's first kcap keys and node ids in LDS (single frames: see the launch)
__global__ __launch_bounds__(QT_THREADS) void k_quadtree(const LevelGeom *__restrict__ geom,
                                                  const CellDesc *__restrict__ cells, int ncells_total,
                                                  const uint32_t *__restrict__ slots, size_t frame_slots,
                                                  int *__restrict__ cell_cnt,
                                                  uint32_t *__restrict__ dense_key,
                                                  uint16_t *__restrict__ dense_node,
                                                  uint32_t *__restrict__ sel, int sel_cap_total,
                                                  int *__restrict__ nsel, int *__restrict__ ncand, int nlevels,
                                                  int ncap, int t_ini, int kcap_arg, long long *dbg_t,
                                                  int *__restrict__ pre_aux /* k_qt_prefilter's results, or nullptr */)
{
    // section timestamps of workgroup (frame 0, level 0) for tools/qt_sections.py; compiled in with -DORBGPU_QT_TIMING
#ifdef ORBGPU_QT_TIMING
    int dbg_k = 0;
#define QT_MARK(id) if (dbg_t && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == 0 && dbg_k < 500) { dbg_t[2 * dbg_k] = (id); dbg_t[2 * dbg_k + 1] = (long long)wall_clock64(); dbg_k++; }
#else
#define QT_MARK(id)
#endif

    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_tmp[16];
    const int kcap = LDS_KEYS ? kcap_arg : 0;
    __shared__ int s_n, s_phase, s_done, s_nexp, s_cut, s_nst, s_nk;

    const int tid = threadIdx.x, nt = blockDim.x;
    // grid = (frames, levels): all workgroups of level 0 (the longest) are dispatched first
    const int level = blockIdx.y, f = blockIdx.x;
    const LevelGeom g = geom[level];
    const int N = g.quota;

    QtShared S;
    uint32_t *lkey;
    uint16_t *lnode;
    uint32_t *dkey = dense_key + (size_t)f * frame_slots + g.slot_off;
    uint16_t *dnode = dense_node + (size_t)f * frame_slots + g.slot_off;
    // list-ordered node arrays, ping-pong A (current) / B (next): bounds (ulx,uly,brx,bry), key
    // count, creation index inside the pass that created the node
    short4 *bndA, *bndB;
    int *cntA, *cntB;
    uint16_t *creA, *creB;
    uint32_t *midA, *midB;  // split point of the node (split_of)
    {
        uint8_t *p = smem;
        bndA = (short4 *)p; p += sizeof(short4) * ncap;
        bndB = (short4 *)p; p += sizeof(short4) * ncap;
        cntA = (int *)p; p += sizeof(int) * ncap;
        cntB = (int *)p; p += sizeof(int) * ncap;
        S.ccnt = (int *)p; p += sizeof(int) * ncap * 4;
        S.ccnt_next = (int *)p; p += sizeof(int) * ncap * 4;
        S.sa = (int *)p; p += sizeof(int) * ncap;
        S.sb = (int *)p; p += sizeof(int) * ncap;
        S.cpos = (uint16_t *)p; p += sizeof(uint16_t) * ncap * 4;
        creA = (uint16_t *)p; p += sizeof(uint16_t) * ncap;
        creB = (uint16_t *)p; p += sizeof(uint16_t) * ncap;
        S.opos = (uint16_t *)p; p += sizeof(uint16_t) * ncap;
        S.order = (uint16_t *)p; p += sizeof(uint16_t) * ncap;
        S.inE = (uint8_t *)p; p += ((size_t)ncap + 3) / 4 * 4;
        midA = (uint32_t *)p; p += sizeof(uint32_t) * ncap;
        midB = (uint32_t *)p; p += sizeof(uint32_t) * ncap;
        // S.soff lives on the node-bound arrays (bndA + bndB = 4 ncap ints >= the cells of a level), which step 0 does not use
        S.soff = (uint32_t *)smem;
        lkey = (uint32_t *)p; p += sizeof(uint32_t) * kcap;
        lnode = (uint16_t *)p;
    }
    // LDS_KEYS: every pass sweeps the keys, and a sweep out of L2 is a chain of memory round trips -- the whole run
    // time of a single frame's level-0 workgroup.  A batch is better served by eight workgroups per CU than by short
    // sweeps (136 us against 198 us at B = 256 with 4096 keys per workgroup in LDS), so it runs with kcap = 0.
    auto key_at = [&](int i) -> uint32_t { return i < kcap ? lkey[i] : dkey[i]; };
    auto node_at = [&](int i) -> int { return i < kcap ? (int)lnode[i] : (int)dnode[i]; };
    auto set_node = [&](int i, int nd) {
        if (i < kcap)
            lnode[i] = (uint16_t)nd;
        else
            dnode[i] = (uint16_t)nd;
    };

    const uint32_t *fslots = slots + (size_t)f * frame_slots;
    int *ccounts = cell_cnt + (size_t)f * (ncells_total + ORBGPU_MAX_LEVELS) + g.cell_first;
    int *lcount = cell_cnt + (size_t)f * (ncells_total + ORBGPU_MAX_LEVELS) + ncells_total + level;

    QT_MARK(0)
    // ---- step 0: this level's keys, filtered and compacted.  k_fast_detect leaves the NMS survivors above the lower
    //      threshold of the whole level in one dense array (no particular order; *lcount of them) and per cell the
    //      counter survivors | (those above iniThFAST) << 16.  cv::FAST(iniThFAST) of a cell is the subset above
    //      iniThFAST (a survivor of the 3x3 suppression at the lower threshold with s > t_hi also survives at t_hi: a
    //      neighbour with s_n <= t_hi cannot beat it), and only a cell where that is empty keeps the rest
    //      (ORBextractor.cc:809-816).  The cell of a key follows from its coordinates: a key of cell (i, j) has
    //      x in [j wCell + 3, (j + 1) wCell + 3), y likewise (:797-825).  The order inside the dense array is arbitrary;
    //      the one place where vToDistributeKeys order matters (step 3) derives it from the key.  On the way the keys
    //      are counted per initial node (:542-585).
    const int n_ini = g.n_ini;
    const float hx = g.hx;
    const uint32_t *lslots = slots + (size_t)f * frame_slots + g.slot_off;
    __shared__ uint16_t s_bin2node[QT_INI_MAX];  // prefiltered path: list position of the initial node of every bin
    const bool pre = LDS_KEYS && pre_aux != nullptr;
    bool labelled = !pre;  // prefiltered path: no key carries a node id until the first pass has re-labelled them
    if (pre) {
        // k_qt_prefilter has filtered, compacted and counted: fetch the counts, pull the LDS share of the keys in, re-arm
        int *laux = pre_aux + (size_t)f * QT_AUX_FRAME + level * QT_AUX_LEVEL;
        if (tid == 0)
            s_nk = laux[0];
        for (int i = tid; i < 4 * n_ini; i += nt)
            S.ccnt_next[i] = laux[1 + i];  // (bin, quadrant) counts; turned into the initial nodes' child counts below
        __syncthreads();
        for (int i = tid; i < 1 + 4 * n_ini; i += nt)
            laux[i] = 0;
        for (int c = tid; c < g.ncells; c += nt)
            ccounts[c] = 0;
        if (tid == 0)
            *lcount = 0;
        for (int b = tid; b < n_ini; b += nt)
            S.sb[b] = S.ccnt_next[4 * b] + S.ccnt_next[4 * b + 1] + S.ccnt_next[4 * b + 2] + S.ccnt_next[4 * b + 3];
        const int nk = min(s_nk, kcap);
        for (int i = tid; i < nk; i += nt)
            lkey[i] = dkey[i];
    } else {
    for (int b = tid; b < n_ini; b += nt)
        S.sb[b] = 0;
    for (int c = tid; c < g.ncells; c += nt) {
        const int cnt = ccounts[c];
        ccounts[c] = 0;                        // ready for the next extraction
        S.soff[c] = (cnt >> 16) ? 1u : 0u;     // the cell has corners above iniThFAST: only those count
    }
    if (tid == 0) {
        s_nst = min(*lcount, g.slot_cnt);  // (never more than the level's array holds: k_fast_detect's bound; a clamp, not a trust)
        *lcount = 0;
        s_nk = 0;
    }
    __syncthreads();
    {
        const int nst = s_nst;
        const int lane = tid & 63;
        for (int i0 = tid; i0 - tid < nst; i0 += QT_ILP * nt) {  // (all lanes iterate: ballots and count_runs work wave-wide)
            uint32_t key[QT_ILP];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++)
                key[u] = lslots[min(i0 + u * nt, nst - 1)];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                const uint32_t cj = __umulhi((uint32_t)max(key_x(key[u]) - 3, 0), g.inv_wcell);
                const uint32_t ci = __umulhi((uint32_t)max(key_y(key[u]) - 3, 0), g.inv_hcell);
                const int c = min((int)(ci * (uint32_t)g.ncols_eff + cj), g.ncells - 1);
                // cornerScore = s - 1:  s > t  <=>  response >= t
                const bool keep = i0 + u * nt < nst && (!S.soff[c] || key_resp(key[u]) >= t_ini);
                const unsigned long long m = __ballot(keep);
                int base = 0;
                if (lane == 0 && m)
                    base = atomicAdd(&s_nk, __popcll(m));
                base = __shfl(base, 0, 64);
                if (keep) {
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    dkey[pos] = key[u];  // also what the debug API reads
                    if (LDS_KEYS && pos < kcap)
                        lkey[pos] = key[u];
                }
                int bin = (int)((float)key_x(key[u]) / hx);
                bin = min(max(bin, 0), n_ini - 1);
                count_runs(S.sb, bin, keep);
            }
        }
    }
    }  // !pre
    __syncthreads();  // dkey written by this workgroup only; visible after the barrier (same CU)
    const int nkeys = s_nk;
    if (tid == 0)
        ncand[(size_t)f * nlevels + level] = nkeys;

    QT_MARK(1)
    // ---- step 1: initial nodes (:542-585); S.sb[b] = keys of bin b, counted by step 0
    if (tid == 0) {
        int n = 0;
        for (int b = 0; b < n_ini; b++) {
            int c = S.sb[b];
            S.sa[b] = n;  // bin -> list position
            if (pre)
                s_bin2node[b] = (uint16_t)n;
            if (c > 0) {
                bndA[n] = make_short4((short)(int)(hx * (float)b), 0, (short)(int)(hx * (float)(b + 1)),
                                          (short)(g.max_by - BORDER0));
                midA[n] = split_of(bndA[n]);
                cntA[n] = c;
                creA[n] = (uint16_t)n;
                n++;
            }
        }
        s_n = n;
        s_phase = 1;
        s_done = 0;
    }
    __syncthreads();
    // Loop control lives in registers.  s_n / s_phase / s_done are only written by section (6) of a pass, which one wave
    // may run on its own (QT_SOLO_MAX) while the others wait at the barrier behind the sections: a wave must therefore
    // never read them between the barrier that ends a pass and the barrier behind the next pass's sections.  They are
    // read here (the next write is two barriers away) and after the barrier behind the sections of every pass.
    int n_cur = s_n, phase_cur = s_phase, done_cur = s_done;
    if (pre) {
        // the child counts of the initial nodes are the prefilter's (bin, quadrant) counts, moved from bin order to list
        // order (empty bins have no node); a node with one key is never expanded and counts nothing (as the sweep below)
        int mine[4] = {0, 0, 0, 0};
        const int b = tid;
        const bool have = b < n_ini && S.sb[b] > 0;
        if (have)
            for (int q = 0; q < 4; q++)
                mine[q] = S.sb[b] > 1 ? S.ccnt_next[4 * b + q] : 0;
        __syncthreads();  // every count read before the array is rewritten in list order (n_ini <= QT_INI_MAX < nt)
        if (have)
            for (int q = 0; q < 4; q++)
                S.ccnt_next[4 * S.sa[b] + q] = mine[q];
    } else {
    for (int i = tid; i < 4 * n_ini; i += nt)
        S.ccnt_next[i] = 0;
    __syncthreads();
    {
        for (int i0 = tid; i0 - tid < nkeys; i0 += QT_ILP * nt) {
            uint32_t key[QT_ILP];
            int nd[QT_ILP];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++)
                key[u] = key_at(min(i0 + u * nt, nkeys - 1));
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                int b = (int)((float)key_x(key[u]) / hx);
                b = min(max(b, 0), n_ini - 1);
                nd[u] = S.sa[b];
            }
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                const bool have = i0 + u * nt < nkeys;
                if (have)
                    set_node(i0 + u * nt, nd[u]);
                count_runs(S.ccnt_next, nd[u] * 4 + quadrant_at(key[u], midA[nd[u]]), have && cntA[nd[u]] > 1);
            }
        }
    }
    }  // !pre
    __syncthreads();
    {
        int *t = S.ccnt;
        S.ccnt = S.ccnt_next;
        S.ccnt_next = t;
    }
    // node of key i: its label, or (prefiltered path, before the first re-labelling sweep) the initial node of its bin
    auto node_of = [&](int i, uint32_t key) -> int {
        if (labelled)
            return node_at(i);
        int b = (int)((float)key_x(key) / hx);
        b = min(max(b, 0), n_ini - 1);
        return (int)s_bin2node[b];
    };

    // step 3's reduction key of a candidate (see there): response + 1 in bits 40.., then the inverted vToDistributeKeys order
    const float inv_w = 1.0f / (float)g.wcell, inv_h = 1.0f / (float)g.hcell;
    auto best_of = [&](uint32_t key) -> unsigned long long {
        const int x = key_x(key), y = key_y(key);  // relative to (minBorderX, minBorderY): cell interiors start at 3
        // exact floor((v - 3) / cell) for v < 4096, cell >= 1: the float quotient is off by less than one ulp
        int cj = (int)((float)(x - 3) * inv_w), ci = (int)((float)(y - 3) * inv_h);
        cj += ((cj + 1) * g.wcell <= x - 3) ? 1 : 0;
        cj -= (cj * g.wcell > x - 3) ? 1 : 0;
        ci += ((ci + 1) * g.hcell <= y - 3) ? 1 : 0;
        ci -= (ci * g.hcell > y - 3) ? 1 : 0;
        const unsigned long long ord = ((unsigned long long)ci << 32) | ((unsigned long long)cj << 24) | (key >> 8);
        return ((unsigned long long)(key_resp(key) + 1) << 40) | (0xFFFFFFFFFFull - ord);
    };
    bool best_done = false;  // the last pass's sweep has already reduced the keys per final node (below)

    QT_MARK(2)
    // ---- passes
    for (int iter = 0; iter < 64; iter++) {
        const int n = n_cur;
        const int phase = phase_cur;
        if (done_cur || n == 0)
            break;
        // (1) S.ccnt already holds the child key counts of every node with >1 keys: they are
        //     accumulated by the key loop of the previous pass (or of the initial assignment)

        // Sections (2)..(6) work on the node list (n <= N entries), not on the keys.  While the list is short one wave does
        // them alone, with wave-level synchronisation and scans -- the other waves of the workgroup would only walk
        // through the barriers and the scans' bookkeeping (which was over a third of this kernel's instructions); a long
        // list is shared by the whole workgroup as before.
        auto node_sections = [&](const int tid_, const int nt_, auto &&SYNC, auto &&SCAN) {
            QT_MARK(10)
            // (2) processing order of the expandable nodes
            for (int p = tid_; p < n; p += nt_)
                S.sa[p] = cntA[p] > 1 ? 1 : 0;
            SYNC();
            const int nv = SCAN(S.sa, n);  // sa[p] = rank among expandable (list order)
            if (phase == 1) {
                for (int p = tid_; p < n; p += nt_)
                    if (cntA[p] > 1)
                        S.order[S.sa[p]] = (uint16_t)p;
            } else {
                // sort by (cnt, creation) descending: rank by counting (keys are unique).  The sort keys of the expandable
                // nodes are first laid out densely (S.sa is free between the scan above and section (4)) and padded with
                // zeros to a multiple of 4, so that the counting loop reads them four at a time
                uint32_t *skey = reinterpret_cast<uint32_t *>(S.sa);
                for (int p = tid_; p < n; p += nt_)
                    if (cntA[p] > 1)
                        S.opos[S.sa[p]] = (uint16_t)p;  // opos as temporary list of expandable nodes
                SYNC();  // (also: every S.sa[p] has been read before the array is reused)
                const int nv4 = (nv + 3) & ~3;
                for (int a = tid_; a < nv4; a += nt_) {
                    const int p = S.opos[min(a, nv - 1)];
                    skey[a] = a < nv ? (((uint32_t)cntA[p] << 16) | creA[p]) : 0u;
                }
                SYNC();
                for (int a = tid_; a < nv; a += nt_) {
                    const uint32_t ka = skey[a];
                    int rank = 0;
                    for (int b = 0; b < nv4; b += 4) {
                        const uint4 kb = *reinterpret_cast<const uint4 *>(&skey[b]);
                        rank += (kb.x > ka) + (kb.y > ka) + (kb.z > ka) + (kb.w > ka);
                    }
                    S.order[rank] = S.opos[a];
                }
            }
            SYNC();

            QT_MARK(11)
            // (3) children per expandable node in processing order; find the cut for phase 2
            for (int j = tid_; j < nv; j += nt_) {
                const int p = S.order[j];
                const int *cc = &S.ccnt[p * 4];
                S.sb[j] = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0);
            }
            if (tid_ == 0)
                s_cut = nv;  // number of nodes to expand
            SYNC();
            const int call = SCAN(S.sb, nv);  // sb[j] = creation index of first child
            if (phase == 2) {
                // list size after expanding j+1 nodes = n + (children so far) - (j+1) >= N ?  (:730-731)
                for (int j = tid_; j < nv; j += nt_) {
                    const int kids_incl = (j + 1 < nv) ? S.sb[j + 1] : call;
                    if (n + kids_incl - (j + 1) >= N)
                        atomicMin(&s_cut, j + 1);
                }
                SYNC();
            }
            const int nE = s_cut;
            const int C = (nE < nv) ? S.sb[nE] : call;  // children created in this pass

            QT_MARK(12)
            // (4) untouched nodes keep their order behind the children
            for (int p = tid_; p < n; p += nt_)
                S.inE[p] = 0;
            SYNC();
            for (int j = tid_; j < nE; j += nt_)
                S.inE[S.order[j]] = 1;
            SYNC();
            for (int p = tid_; p < n; p += nt_)
                S.sa[p] = S.inE[p] ? 0 : 1;
            SYNC();
            const int nkept = SCAN(S.sa, n);
            const int n2 = C + nkept;
            if (n2 > ncap) {  // cannot happen for validated geometry; fail safe
                if (tid_ == 0)
                    s_done = 2;
                return;
            }
            if (tid_ == 0)
                s_nexp = 0;
            SYNC();

            QT_MARK(13)
            // (5) build the new list
            for (int p = tid_; p < n; p += nt_) {
                if (!S.inE[p]) {
                    const int pos = C + S.sa[p];
                    bndB[pos] = bndA[p];
                    midB[pos] = midA[p];
                    cntB[pos] = cntA[p];
                    creB[pos] = creA[p];
                    S.opos[p] = (uint16_t)pos;
                }
            }
            int my_exp = 0;
            for (int j = tid_; j < nE; j += nt_) {
                const int p = S.order[j];
                const short4 b = bndA[p];
                const int halfx = (int)ceilf((float)(b.z - b.x) / 2);
                const int halfy = (int)ceilf((float)(b.w - b.y) / 2);
                const short mx = (short)(b.x + halfx), my = (short)(b.y + halfy);
                int k = S.sb[j];
    #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = S.ccnt[p * 4 + q];
                    if (c == 0)
                        continue;
                    const int pos = C - 1 - k;
                    short4 cb;
                    cb.x = (q & 1) ? mx : b.x;
                    cb.z = (q & 1) ? b.z : mx;
                    cb.y = (q & 2) ? my : b.y;
                    cb.w = (q & 2) ? b.w : my;
                    bndB[pos] = cb;
                    midB[pos] = split_of(cb);
                    cntB[pos] = c;
                    creB[pos] = (uint16_t)k;
                    S.cpos[p * 4 + q] = (uint16_t)pos;
                    my_exp += c > 1;
                    k++;
                }
            }
            if (my_exp)
                atomicAdd(&s_nexp, my_exp);
            SYNC();

            QT_MARK(14)
            // (6) termination / phase switch (:669-673, :733-734), decided before the key loop so that a
            //     final pass does not count children
            if (tid_ == 0) {
                s_n = n2;
                if (n2 >= N || n2 == n)
                    s_done = 1;
                else if (phase == 1 && n2 + 3 * s_nexp > N)
                    s_phase = 2;
            }
            SYNC();
        };
        if (n <= QT_SOLO_MAX) {
            if ((tid >> 6) == 0)
                node_sections(tid & 63, 64, [] { qt_wave_sync(); }, [](int *arr, int cnt) { return wave_excl_scan(arr, cnt); });
        } else
            node_sections(tid, nt, [] { __syncthreads(); }, [&](int *arr, int cnt) { return block_excl_scan(arr, cnt, s_tmp); });
        __syncthreads();
        // the list of the next pass: written by section (6) above, not written again before the next pass's sections
        done_cur = s_done;
        if (done_cur == 2)
            break;
        n_cur = s_n;
        phase_cur = s_phase;
        for (int i = tid; i < n_cur * 4; i += nt)
            S.ccnt_next[i] = 0;
        __syncthreads();
        const bool count_next = done_cur == 0;
        // The last pass (the list is final): nobody reads the labels again except step 3, so the sweep feeds step 3's
        // per-node maximum directly instead of writing them -- one sweep over the keys less.  The maxima accumulate in the
        // child-count array of the list being built, which the loop above has just zeroed for n_cur nodes (4 ints = two
        // 64-bit words each) and which is S.ccnt after the swap below, where step 3 expects it.
        const bool last_pass = done_cur == 1;
        unsigned long long *best_next = reinterpret_cast<unsigned long long *>(S.ccnt_next);

        QT_MARK(15)
        // (7) re-label the keys and, in the same sweep, count the children of the NEW list's nodes
        for (int i0 = tid; i0 - tid < nkeys; i0 += QT_ILP * nt) {
            uint32_t key[QT_ILP];
            int nd[QT_ILP], nn[QT_ILP];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                const int i = min(i0 + u * nt, nkeys - 1);
                key[u] = key_at(i);
                nd[u] = node_of(i, key[u]);
            }
#pragma unroll
            for (int u = 0; u < QT_ILP; u++)
                nn[u] = S.inE[nd[u]] ? S.cpos[nd[u] * 4 + quadrant_at(key[u], midA[nd[u]])] : S.opos[nd[u]];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                const bool have = i0 + u * nt < nkeys;
                if (last_pass) {
                    if (have)
                        atomicMax(&best_next[nn[u]], best_of(key[u]));
                    continue;
                }
                if (have)
                    set_node(i0 + u * nt, nn[u]);
                if (count_next)
                    count_runs(S.ccnt_next, nn[u] * 4 + quadrant_at(key[u], midB[nn[u]]), have && cntB[nn[u]] > 1);
            }
        }
        __syncthreads();
        best_done = last_pass;
        labelled = labelled || !last_pass;  // every key carries the id of its node in the new list (not after the last pass)
        {
            int *t = S.ccnt;
            S.ccnt = S.ccnt_next;
            S.ccnt_next = t;
        }
        {
            short4 *tb = bndA; bndA = bndB; bndB = tb;
            int *tc = cntA; cntA = cntB; cntB = tc;
            uint16_t *tr = creA; creA = creB; creB = tr;
            uint32_t *tm = midA; midA = midB; midB = tm;
        }
        __syncthreads();
    }

    QT_MARK(3)
    // ---- step 3: best response per node, first maximum in vToDistributeKeys order (:741-760): cells row-major
    //      (:789-829), inside a cell cv::FAST's row-major output -- i.e. ascending (cell row, cell column, y, x), all of
    //      which the key holds.  One 64-bit maximum per node: response + 1 (bits 40..), then the inverted order (cell
    //      row 32..39, cell column 24..31, y 12..23, x 0..11), which is also enough to rebuild the winning key.
    const int n = n_cur;
    unsigned long long *best = reinterpret_cast<unsigned long long *>(S.ccnt);  // [ncap*4] ints >= n 64-bit words
    if (!best_done) {  // (no last pass ran: the initial list was final, or the pass limit was hit)
        for (int p = tid; p < n; p += nt)
            best[p] = 0ull;
        __syncthreads();
        for (int i0 = tid; i0 < nkeys; i0 += QT_ILP * nt) {
            uint32_t key[QT_ILP];
            int nd[QT_ILP];
#pragma unroll
            for (int u = 0; u < QT_ILP; u++) {
                const int i = min(i0 + u * nt, nkeys - 1);
                key[u] = key_at(i);
                nd[u] = node_of(i, key[u]);
            }
#pragma unroll
            for (int u = 0; u < QT_ILP; u++)
                if (i0 + u * nt < nkeys)
                    atomicMax(&best[nd[u]], best_of(key[u]));
        }
    }
    __syncthreads();
    uint32_t *osel = sel + (size_t)f * sel_cap_total + g.sel_off;
    const bool ok = done_cur != 2 && n <= g.sel_cap;
    for (int p = tid; p < n && ok; p += nt) {
        const unsigned long long b = best[p];
        const uint32_t yx = (uint32_t)((0xFFFFFFFFFFull - (b & 0xFFFFFFFFFFull)) & 0xFFFFFFull);
        osel[p] = (yx << 8) | (uint32_t)((b >> 40) - 1ull);
    }
    QT_MARK(4)
    if (tid == 0)
        nsel[(size_t)f * nlevels + level] = ok ? n : -1;
}

// ---------------------------------------------------------------------------------------------
// K4: orientation + key-point assembly
//     (IC_Angle :77-104, fastAtan2 A5, fix-up :837-847, scaling :1095-1101)
// ---------------------------------------------------------------------------------------------
struct KpAux {  // x: the key point's column in its blurred plane minus EDGE (= its image column, except level 0 in direct mode)
    int x, y, level;
    float angle;
    float ca, sb;  // cosf / sinf of the angle in radians as the handle's trig mode defines them (k_trig)
    int plane_off, pitch;  // of the key point's level: k_describe's patch loads then depend on this record only
};

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float s = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0)
        a = 180.f - a;
    if (y < 0)
        a = 360.f - a;
    return a;
}

struct UMax {
    int v[16];
};

// Half a wave (32 lanes) per selected key point, lane = one row v of the radius-15 disc.  A lane
// reads its row as 9 aligned dwords and reduces it with v_dot4_u32_u8 against per-(alignment, |v|)
// byte-weight tables (built on the host, copied to LDS by every workgroup): W10 holds (u+16) inside the disc (0 outside),
// M01 holds 1 inside the disc, so  sum u*I = dot(W10) - 16*dot(M01)  and  sum I = dot(M01).
// Integer moments: exact in any summation order.
constexpr int OR_ITERS = 4;  // key points per half wave for full batches (1 below OR_BATCH_MIN frames: latency)
constexpr int OR_BATCH_MIN = 32;

__global__ __launch_bounds__(256) void k_orient(const uint8_t *__restrict__ pyr, size_t frame_pyr,
                                                const LevelGeom *__restrict__ geom, int nlevels,
                                                const uint32_t *__restrict__ sel, int sel_cap_total,
                                                const int *__restrict__ nsel, const uint32_t *__restrict__ orw,
                                                orbgpu_keypoint *__restrict__ kps, KpAux *__restrict__ aux,
                                                int cap, int *__restrict__ n_out, int iters, Src0 s0)
{
    __shared__ uint32_t W10[4 * 16 * 9], M01[4 * 16 * 9];
    __shared__ __align__(16) uint8_t stage[8][31 * 48];
    // the weight tables are built once on the host (orient_tables): a workgroup only copies them
    for (int i = threadIdx.x; i < 4 * 16 * 9; i += 256) {
        W10[i] = orw[i];
        M01[i] = orw[4 * 16 * 9 + i];
    }
    __syncthreads();

    const int hl = threadIdx.x & 31;  // lane inside the half wave = disc row index
    int bx, f;
    xcd_frame_block(bx, f);
    f = serpentine(f);
    const int *ns = nsel + (size_t)f * nlevels;
    // `iters` slots per half wave: the weight tables above are built once per 8 * iters key points
    for (int it = 0; it < iters; it++) {
    const int slot = (bx * iters + it) * 8 + (threadIdx.x >> 5);
    // locate (level, j) of this slot and the output offset
    int level = -1, j = 0, out_off = 0, total = 0, cnt_l = 0;
    bool bad = false;
    // the level's geometry is picked up while the (wave-uniform, scalar) level records are walked anyway, so the
    // key load below does not wait for a per-lane LevelGeom fetch
    struct {
        int plane_off, pitch, sel_off, patch;
        float scale;
    } g = {0, 0, 0, 0, 1.f};
    for (int l = 0; l < nlevels; l++) {
        const int c = ns[l];
        if (c < 0)
            bad = true;
        const int so = geom[l].sel_off;
        if (slot >= so && slot < so + geom[l].sel_cap) {
            level = l;
            j = slot - so;
            out_off = total;
            cnt_l = c;
            g.plane_off = geom[l].plane_off;
            g.pitch = geom[l].pitch;
            g.sel_off = so;
            g.patch = geom[l].patch;
            g.scale = geom[l].scale;
        }
        total += max(c, 0);
    }
    if (slot == 0 && hl == 0)
        n_out[f] = bad ? -1 - total : (total > cap ? -1 - total : total);
    if (level < 0 || bad || total > cap || j >= cnt_l)
        continue;
    const uint32_t key = sel[(size_t)f * sel_cap_total + g.sel_off + j];
    const int x = key_x(key) + BORDER0, y = key_y(key) + BORDER0;
    // the key point's plane: the padded level, or -- level 0 in direct mode -- the caller's image (pixel 0 at column 0)
    const bool dir0 = s0.direct != 0 && level == 0;
    const int porg = dir0 ? 0 : EDGE, ppitch = dir0 ? (int)s0.pitch : g.pitch;
    const int xl = x + porg - HALF_PATCH;  // leftmost column of the disc in that plane
    const int a = xl & 3;
    const int v = hl - HALF_PATCH;         // rows -15..15 (lane 31 idles)
    int m10 = 0, m01 = 0;
    {
        // The disc's 31 rows staged in LDS as 16-byte pieces of the 48-byte window that starts at the aligned column
        // xl - a.  A row of half width u needs the window's bytes 15 - u + a .. 15 + u + a: the first two
        // pieces always, the third only where u + a >= 17 -- 62 to 75 of the 93 pieces (the weights of the bytes that are
        // not loaded are zero, so what LDS holds there does not matter).  Until round 4 all 93 were loaded.
        const uint8_t *base = (dir0 ? s0.p + (size_t)f * s0.frame_stride : pyr + (size_t)f * frame_pyr + g.plane_off) + (xl - a);
        struct __attribute__((packed, aligned(4))) Piece {
            uint32_t d[4];
        };
        // lane = (row, piece) for the 62 pieces every disc needs: two neighbouring lanes share a row, so a wave-load touches
        // ~16 rows per key point; then the rows around the centre whose third piece is needed: |v| <= lim
        const int lim = a == 2 ? 3 : (a == 3 ? 6 : -1);  // umax[|v|] + a >= 17 (umax: ORBextractor.cc:452-469, host-checked)
        const uint8_t *rowbase = base + rowoff(y + porg - HALF_PATCH, ppitch);
        uint4 *pl = reinterpret_cast<uint4 *>(stage[threadIdx.x >> 5]);
        Piece pc[3];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int q = hl + 32 * k, r = q >> 1, part = q & 1;
            if (q < 62)
                pc[k] = *reinterpret_cast<const Piece *>(rowbase + rowoff(r, ppitch) + part * 16);
        }
        const int r2 = HALF_PATCH - lim + hl;
        if (hl < 2 * lim + 1)
            pc[2] = *reinterpret_cast<const Piece *>(rowbase + rowoff(r2, ppitch) + 32);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int q = hl + 32 * k;
            if (q < 62)
                pl[3 * (q >> 1) + (q & 1)] = make_uint4(pc[k].d[0], pc[k].d[1], pc[k].d[2], pc[k].d[3]);
        }
        if (hl < 2 * lim + 1)
            pl[3 * r2 + 2] = make_uint4(pc[2].d[0], pc[2].d[1], pc[2].d[2], pc[2].d[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (hl < 31) {
            const int av = v < 0 ? -v : v;
            const uint32_t *row = reinterpret_cast<const uint32_t *>(stage[threadIdx.x >> 5]) + hl * 12;
            const uint32_t *w = &W10[(a * 16 + av) * 9], *m = &M01[(a * 16 + av) * 9];
            uint32_t sw = 0, sm = 0;
#pragma unroll
            for (int q = 0; q < 9; q++) {
                const uint32_t d = row[q];
                sw = __builtin_amdgcn_udot4(d, w[q], sw, false);
                sm = __builtin_amdgcn_udot4(d, m[q], sm, false);
            }
            m10 = (int)sw - 16 * (int)sm;
            m01 = v * (int)sm;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next item overwrites the slot
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off, 64);
        m01 += __shfl_xor(m01, off, 64);
    }
    if (hl == 0) {
        const float angle = fast_atan2_deg((float)m01, (float)m10);
        orbgpu_keypoint kp;
        kp.x = (float)x;
        kp.y = (float)y;
        if (level != 0) {
            kp.x *= g.scale;
            kp.y *= g.scale;
        }
        kp.size = (float)g.patch;
        kp.angle = angle;
        kp.response = (float)key_resp(key);
        kp.octave = level;
        kp.class_id = -1;
        kps[(size_t)f * cap + out_off + j] = kp;
        KpAux ax;
        ax.x = x + (dir0 ? BLUR0_OX - EDGE : 0);  // k_describe reads the BLURRED plane: column x + EDGE there (direct level 0: BLUR0_OX)
        ax.y = y;
        ax.level = level;
        ax.plane_off = g.plane_off;
        ax.pitch = g.pitch;
        ax.angle = angle;
        aux[(size_t)f * cap + out_off + j] = ax;
    }
    }
}
// computeOrbDescriptor's  a = (float)cos(angle), b = (float)sin(angle)  (ORBextractor.cc:112-113): std::cos(float) /
// std::sin(float) of the reference's host, i.e. its libm's cosf / sinf (trig_base.h).  The fixed double-precision
// sequence gives the correctly rounded value; the table (if the handle has one) replaces it where the host's libm
// returns something else.  One thread per key point: 64 key points per wave share the double-precision evaluation.
__global__ __launch_bounds__(256) void k_trig(KpAux *__restrict__ aux, const int *__restrict__ n_out, int cap, TrigTable tt)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out[f])
        return;
    KpAux *a = aux + (size_t)f * cap + i;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float angle = a->angle * factorPI;
    float c, s;
    orbgpu_trig_device(angle, tt, &c, &s);
    a->ca = c;
    a->sb = s;
}

// ---------------------------------------------------------------------------------------------
// K5: GaussianBlur(7x7, sigma 2, REFLECT_101), OpenCV 2.4 8-bit fixed-point path (A3).
// taps x256 = [18,34,49,55,49,34,18]; dst = sat_u8((sum_y sum_x k_y k_x p + 32768) >> 16).  Integer
// arithmetic, so any summation order is exact.  The border the reference gets from REFLECT_101 on
// the cloned level equals the pyramid's own 19-px reflect-101 border: the kernel reads the padded
// plane and never branches on image edges.
//
// Register-only streaming kernel, no LDS: a thread owns a column strip of 8 output pixels (two
// aligned dwords of the padded plane) x BLUR_ROWS rows.  Per input row it loads 4 aligned dwords in one
// 16-byte access (they cover the 14 bytes it needs), forms the eight 7-tap horizontal sums with v_alignbyte +
// v_dot4_u32_u8, keeps the last row sums in registers as six PAIRS of consecutive rows (blur_vsum) and emits 8 bytes per row.  (4 pixels per thread with
// 12-byte loads and 4-byte stores: 176 us against 166 us; a second input row in flight: no gain.)
// HBM-bound: each pixel is fetched once from HBM (neighbouring strips re-read through L1/L2).
// ---------------------------------------------------------------------------------------------
constexpr int BLUR_ROWS = 30;  // output rows per strip (5 x 6: the 6-slot register ring of row pairs unrolls evenly)

__device__ __forceinline__ void blur_hsum(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t h[4])
{
    // window bytes 0..11 = padded columns 4s-4 .. 4s+7; output pixel p has its centre at byte 4+p, its taps at bytes
    // 1+p .. 7+p.  The taps [18,34,49,55,49,34,18] are laid over the three dwords as per-pixel weight constants, so no
    // data is shifted: 2 + 3 + 3 + 2 v_dot4_u32_u8 for the four pixels (6 v_alignbyte + 8 v_dot4 before).
#define W4(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
    h[0] = __builtin_amdgcn_udot4(d1, W4(55, 49, 34, 18), __builtin_amdgcn_udot4(d0, W4(0, 18, 34, 49), 0u, false), false);
    h[1] = __builtin_amdgcn_udot4(d2, W4(18, 0, 0, 0), __builtin_amdgcn_udot4(d1, W4(49, 55, 49, 34), __builtin_amdgcn_udot4(d0, W4(0, 0, 18, 34), 0u, false), false), false);
    h[2] = __builtin_amdgcn_udot4(d2, W4(34, 18, 0, 0), __builtin_amdgcn_udot4(d1, W4(34, 49, 55, 49), __builtin_amdgcn_udot4(d0, W4(0, 0, 0, 18), 0u, false), false), false);
    h[3] = __builtin_amdgcn_udot4(d2, W4(49, 34, 18, 0), __builtin_amdgcn_udot4(d1, W4(18, 34, 49, 55), 0u, false), false);
#undef W4
}

// Vertical pass on PAIRS of row sums.  A row sum is <= 255 * 257 = 65535: it fits 16 bits exactly, so P_r = (h_{r-1},
// h_r) packs two consecutive rows of a pixel into one register (one v_alignbit when row r arrives) and
// v_dot2_u32_u16 applies two taps per instruction: for the output row c
//   s = 32768 + P_{c-2}.(18,34) + P_c.(49,55) + P_{c+2}.(49,34) + P_{c+3}.(0,18)
// (4 instructions instead of 3 adds + 4 multiply-adds + the rounding add), the ring holds the 6 pairs P_{c-2..c+3}.
// The four results are narrowed two at a time: v_perm takes the high halves (s >> 16 <= 257), v_pk_min_u16 saturates.
__device__ __forceinline__ uint32_t blur_vsum(const uint32_t A[4], const uint32_t C[4], const uint32_t E[4],
                                              const uint32_t F[4])
{
    const us2 K01 = {18, 34}, K23 = {49, 55}, K45 = {49, 34}, K6 = {0, 18};
    uint32_t sv[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        uint32_t acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, A[p]), K01, 32768u, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, C[p]), K23, acc, false);
        acc = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, E[p]), K45, acc, false);
        sv[p] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, F[p]), K6, acc, false);
    }
    const us2 cap = {255, 255};
    const us2 lo = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm(sv[1], sv[0], 0x07060302u)), cap);
    const us2 hi = __builtin_elementwise_min(__builtin_bit_cast(us2, __builtin_amdgcn_perm(sv[3], sv[2], 0x07060302u)), cap);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                               size_t frame_pyr, const LevelGeom *__restrict__ geom, StripGeom bg, int strip0)
{
    int bx, f;
    xcd_frame_block(bx, f);
    f = serpentine(f);
    const int strip = strip0 + bx * 256 + threadIdx.x;  // strip0 = first strip of level 1 when level 0 is blurred by k_blur0_direct
    if (strip >= bg.first[bg.nlevels])
        return;
    int level = 0;
#pragma unroll
    for (int l = 1; l < ORBGPU_MAX_LEVELS; l++)
        level += (l < bg.nlevels && strip >= bg.first[l]) ? 1 : 0;
    const LevelGeom g = geom[level];
    const int local = strip - bg.first[level];
    const int sy = local / bg.nsx[level], sx = local - sy * bg.nsx[level];
    const int col = (4 + 2 * sx) * 4;  // first of the strip's two padded dword columns
    const int y0 = sy * BLUR_ROWS;
    const int rows = min(BLUR_ROWS, g.h - y0);
    // wave-uniform frame bases + 32-bit offsets stepped by one pitch per row (a frame's planes are far below 4 GB):
    // one v_add per access instead of a 64-bit multiply-add
    const uint8_t *fsrc = pyr + (size_t)f * frame_pyr;
    uint8_t *fdst = blur + (size_t)f * frame_pyr;
    const uint32_t pitch = (uint32_t)g.pitch;
    const uint32_t src0 = (uint32_t)g.plane_off + (uint32_t)(y0 + EDGE - 3) * pitch + (uint32_t)(col - 4);
    uint32_t soff = src0 + 6u * pitch;  // the row fetched one step ahead
    uint32_t doff = (uint32_t)g.plane_off + (uint32_t)(y0 + EDGE) * pitch + (uint32_t)col;
    struct __attribute__((packed, aligned(4))) Q4 {
        uint32_t d[4];
    };
    uint32_t p0[8], p1[8], p2[8], p3[8], p4[8], p5[8], hn[8];  // six row PAIRS (blur_vsum) and the newest row's sums
    Q4 nx;
#define B8_HSUM(R, Q)                                                                                        \
    {                                                                                                        \
        blur_hsum(Q.d[0], Q.d[1], Q.d[2], R);                                                                \
        blur_hsum(Q.d[1], Q.d[2], Q.d[3], R + 4);                                                            \
    }
// the pair (previous row, this row): the previous row is the high half of the previous pair
#define B8_PAIR(P, PREV, H)                                                                                  \
    _Pragma("unroll") for (int p = 0; p < 8; p++) P[p] = __builtin_amdgcn_alignbit(H[p], PREV[p], 16);
#define B8_LOAD(P, PREV, row)                                                                                \
    {                                                                                                        \
        const Q4 q = *reinterpret_cast<const Q4 *>(fsrc + (src0 + (uint32_t)(row) * pitch));                \
        B8_HSUM(hn, q)                                                                                       \
        B8_PAIR(P, PREV, hn)                                                                                 \
    }
// A = P_{c-2}, C = P_c, E = P_{c+2}; F receives P_{c+3} (it held P_{c-3})
#define B8_STEP(A, C, E, F, k)                                                                               \
    if ((k) < rows) {                                                                                        \
        B8_HSUM(hn, nx)                                                                                      \
        B8_PAIR(F, E, hn)                                                                                    \
        soff += pitch;                                                                                       \
        if ((k) + 1 < rows)                                                                                  \
            nx = *reinterpret_cast<const Q4 *>(fsrc + soff);                                                 \
        uint2 o;                                                                                             \
        o.x = blur_vsum(A, C, E, F);                                                                         \
        o.y = blur_vsum(A + 4, C + 4, E + 4, F + 4);                                                         \
        *reinterpret_cast<uint2 *>(fdst + doff) = o;                                                         \
        doff += pitch;                                                                                       \
    }
    // rows y0-3 .. y0+2: the pairs P_{-2} .. P_{2} of the first output row (p5's low half, row y0-4, is never used: zero)
#pragma unroll
    for (int p = 0; p < 8; p++)
        p5[p] = 0;
    B8_LOAD(p5, p5, 0) B8_LOAD(p0, p5, 1) B8_LOAD(p1, p0, 2) B8_LOAD(p2, p1, 3) B8_LOAD(p3, p2, 4) B8_LOAD(p4, p3, 5)
    nx = *reinterpret_cast<const Q4 *>(fsrc + soff);  // next input row, fetched one step ahead (two: no gain)
    // now p0 = P_{-2} = (row -3, row -2), p1 = P_{-1}, p2 = P_0, p3 = P_1, p4 = P_2; p5 = (-, row -3) is free for P_3
#pragma unroll 1
    for (int k = 0; k < BLUR_ROWS; k += 6) {
        B8_STEP(p0, p2, p4, p5, k)
        B8_STEP(p1, p3, p5, p0, k + 1)
        B8_STEP(p2, p4, p0, p1, k + 2)
        B8_STEP(p3, p5, p1, p2, k + 3)
        B8_STEP(p4, p0, p2, p3, k + 4)
        B8_STEP(p5, p1, p3, p4, k + 5)
    }
#undef B8_PAIR
#undef B8_STEP
#undef B8_LOAD
#undef B8_HSUM
}

// Level 0 of the blur in direct mode: the same arithmetic on the caller's image (Src0).  A strip is 8 pixels aligned to the
// IMAGE (x0 = 8 sx; requires width % 8 == 0), its 16-byte window x0-4 .. x0+11 one aligned load; what the padded plane
// gave for free is done here: rows outside the image are reflected (REFLECT_101 on the cloned level, ORBextractor.cc:1085-
// 1086), the first strip of a row builds x = -4..-1 and the last one x = w..w+3 from the neighbouring pixels with one
// v_perm each (their loads start at x = 0 resp. w-16, so nothing outside an image row is ever read).  Output: the
// blurred level-0 plane with pixel 0 at column BLUR0_OX (8 aligned bytes per strip and row).
__global__ __launch_bounds__(256) void k_blur0_direct(Src0 s0, uint8_t *__restrict__ blur, size_t frame_pyr,
                                                       const LevelGeom *__restrict__ geom)
{
    int bx, f;
    xcd_frame_block(bx, f);
    f = serpentine(f);
    const LevelGeom g = geom[0];
    const int nsx = g.w >> 3;
    const int strip = bx * 256 + threadIdx.x;
    const int sy = strip / nsx, sx = strip - sy * nsx;
    const int y0 = sy * BLUR_ROWS;
    if (y0 >= g.h)
        return;
    const int rows = min(BLUR_ROWS, g.h - y0);
    const bool first = sx == 0, last = sx == nsx - 1;
    const uint8_t *fsrc = s0.p + (size_t)f * s0.frame_stride + (first ? 0 : (last ? g.w - 16 : 8 * sx - 4));
    uint8_t *fdst = blur + (size_t)f * frame_pyr;
    const uint32_t pitch = (uint32_t)g.pitch;
    uint32_t doff = (uint32_t)g.plane_off + (uint32_t)(y0 + EDGE) * pitch + (uint32_t)(8 * sx + BLUR0_OX);
    struct __attribute__((packed, aligned(4))) Q4 {
        uint32_t d[4];
    };
    struct __attribute__((packed, aligned(4))) O2 {
        uint32_t d[2];
    };
    // the window of image row y (any y in [-3, h+2]) as the four dwords x0-4 .. x0+11
    auto fetch = [&](int y) -> Q4 {
        const int ry = y < 0 ? -y : (y >= g.h ? 2 * g.h - 2 - y : y);
        return *reinterpret_cast<const Q4 *>(fsrc + __umul24((unsigned)ry, s0.pitch));
    };
    auto window = [&](const Q4 &q) -> Q4 {
        Q4 w;
        // first strip: q = x 0..15 -> (x4 x3 x2 x1 | q0 | q1 | q2); last strip: q = x w-16..w-1 -> (q1 | q2 | q3 | w-2 w-3 w-4 w-5)
        const uint32_t left = __builtin_amdgcn_perm(q.d[1], q.d[0], 0x01020304u);
        const uint32_t right = __builtin_amdgcn_perm(q.d[3], q.d[2], 0x03040506u);
        w.d[0] = first ? left : (last ? q.d[1] : q.d[0]);
        w.d[1] = first ? q.d[0] : (last ? q.d[2] : q.d[1]);
        w.d[2] = first ? q.d[1] : (last ? q.d[3] : q.d[2]);
        w.d[3] = first ? q.d[2] : (last ? right : q.d[3]);
        return w;
    };
    uint32_t p0[8], p1[8], p2[8], p3[8], p4[8], p5[8], hn[8];  // six row PAIRS (blur_vsum) and the newest row's sums
    Q4 nx;
#define B8_HSUM(R, Q)                                                                                        \
    {                                                                                                        \
        const Q4 wq = window(Q);                                                                             \
        blur_hsum(wq.d[0], wq.d[1], wq.d[2], R);                                                             \
        blur_hsum(wq.d[1], wq.d[2], wq.d[3], R + 4);                                                         \
    }
#define B8_PAIR(P, PREV, H)                                                                                  \
    _Pragma("unroll") for (int p = 0; p < 8; p++) P[p] = __builtin_amdgcn_alignbit(H[p], PREV[p], 16);
#define B8_LOAD(P, PREV, row)                                                                                \
    {                                                                                                        \
        const Q4 q = fetch(y0 - 3 + (row));                                                                  \
        B8_HSUM(hn, q)                                                                                       \
        B8_PAIR(P, PREV, hn)                                                                                 \
    }
#define B8_STEP(A, C, E, F, k)                                                                               \
    if ((k) < rows) {                                                                                        \
        B8_HSUM(hn, nx)                                                                                      \
        B8_PAIR(F, E, hn)                                                                                    \
        if ((k) + 1 < rows)                                                                                  \
            nx = fetch(y0 + 4 + (k));                                                                        \
        O2 o;                                                                                                \
        o.d[0] = blur_vsum(A, C, E, F);                                                                      \
        o.d[1] = blur_vsum(A + 4, C + 4, E + 4, F + 4);                                                      \
        *reinterpret_cast<O2 *>(fdst + doff) = o;                                                            \
        doff += pitch;                                                                                       \
    }
#pragma unroll
    for (int p = 0; p < 8; p++)
        p5[p] = 0;
    B8_LOAD(p5, p5, 0) B8_LOAD(p0, p5, 1) B8_LOAD(p1, p0, 2) B8_LOAD(p2, p1, 3) B8_LOAD(p3, p2, 4) B8_LOAD(p4, p3, 5)
    nx = fetch(y0 + 3);
#pragma unroll 1
    for (int k = 0; k < BLUR_ROWS; k += 6) {
        B8_STEP(p0, p2, p4, p5, k)
        B8_STEP(p1, p3, p5, p0, k + 1)
        B8_STEP(p2, p4, p0, p1, k + 2)
        B8_STEP(p3, p5, p1, p2, k + 3)
        B8_STEP(p4, p0, p2, p3, k + 4)
        B8_STEP(p5, p1, p3, p4, k + 5)
    }
#undef B8_PAIR
#undef B8_STEP
#undef B8_LOAD
#undef B8_HSUM
}

// ---------------------------------------------------------------------------------------------
// K6: steered BRIEF, 256 bits                         (computeOrbDescriptor, :108-147)
// 32 lanes per key point, one descriptor byte (8 tests, 16 samples) per lane.
// ---------------------------------------------------------------------------------------------
constexpr int DP_R = 18;                  // |rotated pattern offset| <= 18 (radius^2 <= 338, A6)
constexpr int DP_ROWS = 2 * DP_R + 1;     // 37 rows
constexpr int DP_DW = 12;                 // 37 bytes + up to 3 bytes of alignment slack = 10 dwords, staged as 3 x 16 bytes
constexpr int DP_BYTES = DP_ROWS * DP_DW * 4;

__global__ __launch_bounds__(256) void k_describe(const uint8_t *__restrict__ blur, size_t frame_pyr,
                                                  const LevelGeom *__restrict__ geom,
                                                  const KpAux *__restrict__ aux, const int *__restrict__ n_out,
                                                  int cap, const int8_t *__restrict__ pattern,
                                                  uint8_t *__restrict__ desc)
{
    // the 37x37 neighbourhood of each key point is staged in LDS with coalesced aligned dword loads
    // (10 consecutive dwords per row); the 512 rotated samples are then LDS byte reads instead of 512
    // scattered global byte loads
    __shared__ __align__(16) uint8_t patch[8][DP_BYTES];
    int bx, f;
    xcd_frame_block(bx, f);
    const int n = n_out[f];
    const int slot = threadIdx.x >> 5;
    const int kpi = bx * 8 + slot;
    if (kpi >= n)
        return;  // whole half-wave leaves together; the other half of the wave is independent
    const int byte = threadIdx.x & 31;
    const KpAux a = aux[(size_t)f * cap + kpi];
    // this lane's 8 tests = 32 pattern bytes (x0,y0,x1,y1 per test), L1-resident table
    const uint4 pq0 = reinterpret_cast<const uint4 *>(pattern)[byte * 2];
    const uint4 pq1 = reinterpret_cast<const uint4 *>(pattern)[byte * 2 + 1];
    const uint32_t pw[8] = {pq0.x, pq0.y, pq0.z, pq0.w, pq1.x, pq1.y, pq1.z, pq1.w};
    const float ca = a.ca, sb = a.sb;
    const int xl = a.x + EDGE - DP_R;  // leftmost padded column of the patch
    const int al = xl & 3;
    // staging: 37 rows x three 16-byte pieces = 111 pieces over the 32 lanes of the key point, four x4 loads per lane (the
    // 8 bytes past the 40 a row needs are still inside the padded row: the key point is >= 16 pixels from the image edge).
    // The frame base is wave-uniform and the rest a 32-bit offset
    const uint8_t *fb = blur + (size_t)f * frame_pyr;
    const uint32_t vbase = (uint32_t)a.plane_off + (uint32_t)(a.y + EDGE - DP_R) * (uint32_t)a.pitch + (uint32_t)(xl - al);
    struct __attribute__((packed, aligned(4))) Piece {
        uint32_t d[4];
    };
    constexpr int NPIECE = DP_ROWS * 3, NK = (NPIECE + 31) / 32;  // 111, 4
    Piece stage[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int q = byte + 32 * k, r = (q * 171) >> 9, part = q - 3 * r;  // q / 3 for 0..127
        if (q < NPIECE)
            stage[k] = *reinterpret_cast<const Piece *>(fb + (vbase + (uint32_t)r * (uint32_t)a.pitch + (uint32_t)part * 16u));
    }
    uint4 *pl = reinterpret_cast<uint4 *>(patch[slot]);
#pragma unroll
    for (int k = 0; k < NK; k++)
        if (byte + 32 * k < NPIECE)
            pl[byte + 32 * k] = make_uint4(stage[k].d[0], stage[k].d[1], stage[k].d[2], stage[k].d[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // cvRound by the magic constant 1.5 * 2^23: t = v + MAGIC is v rounded to the nearest-even integer (|v| < 2^22), and the
    // low 24 bits of its pattern are 0x400000 + round(v).  v_mad_u32_u24 reads exactly those bits, so the byte offset
    // 40 r + c costs one multiply-add and one add of a per-lane constant that takes the biases out (mod 2^32)
    constexpr float MAGIC = 12582912.0f;
    constexpr uint32_t BIAS = 0x400000u * (DP_DW * 4) + 0x4B400000u;
    const uint32_t kc = (uint32_t)(slot * DP_BYTES + DP_R * (DP_DW * 4) + DP_R + al) - BIAS;
    const uint8_t *lds = &patch[0][0];
    int val = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float x0 = (float)(int8_t)(pw[k] & 0xFF), y0 = (float)(int8_t)((pw[k] >> 8) & 0xFF);
        const float x1 = (float)(int8_t)((pw[k] >> 16) & 0xFF), y1 = (float)(int8_t)(pw[k] >> 24);
        const uint32_t r0b = __float_as_uint((x0 * sb + y0 * ca) + MAGIC), c0b = __float_as_uint((x0 * ca - y0 * sb) + MAGIC);
        const uint32_t r1b = __float_as_uint((x1 * sb + y1 * ca) + MAGIC), c1b = __float_as_uint((x1 * ca - y1 * sb) + MAGIC);
        const int t0 = lds[__umul24(r0b, DP_DW * 4) + c0b + kc];
        const int t1 = lds[__umul24(r1b, DP_DW * 4) + c1b + kc];
        val |= (t0 < t1) << k;
    }
    desc[((size_t)f * cap + kpi) * 32 + byte] = (uint8_t)val;
}

// small helper kernels for the debug API
__global__ void k_unpack_keys(const uint32_t *__restrict__ keys, int n, int *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint32_t k = keys[i];
        out[3 * i] = key_x(k);
        out[3 * i + 1] = key_y(k);
        out[3 * i + 2] = key_resp(k);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
static const int8_t k_pattern_host[1024] = {
#include "orbgpu_pattern.inc"
};

enum Stage { ST_PYRAMID = 0, ST_BLUR, ST_FAST, ST_QUADTREE, ST_ORIENT, ST_DESCRIBE, ST_COUNT };  // in launch order
static const char *k_stage_names[ST_COUNT] = {"pyramid", "blur", "fast", "quadtree", "orient", "describe"};

#ifdef ORBGPU_QT_TIMING
static long long *g_qt_dbg = nullptr;
static long long *qt_dbg()
{
    if (!g_qt_dbg) {
        (void)hipMalloc(&g_qt_dbg, 8000);
        (void)hipMemset(g_qt_dbg, 0, 8000);
    }
    return g_qt_dbg;
}
extern "C" void orbgpu_qt_dbg_dump()
{
    long long h[1000];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, g_qt_dbg, 8000, hipMemcpyDeviceToHost);
    for (int i = 0; i < 500 && (i == 0 || h[2 * i + 1]); i++)
        printf("mark %lld t %lld\n", h[2 * i], h[2 * i + 1]);
}
#else
static long long *qt_dbg() { return nullptr; }
#endif
static inline int cv_round_host(double v) { return (int)lrint(v); }

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_extractor {
    orbgpu_extractor_params prm;
    double scale_factor_d;
    int nlevels;
    float scale[ORBGPU_MAX_LEVELS], inv_scale[ORBGPU_MAX_LEVELS], sigma2[ORBGPU_MAX_LEVELS],
        inv_sigma2[ORBGPU_MAX_LEVELS];
    int quota[ORBGPU_MAX_LEVELS];
    UMax umax;
    // geometry of the configured image size
    int cfg_w = 0, cfg_h = 0, cfg_batch = 0;
    std::vector<LevelGeom> geom;
    std::vector<CellDesc> cells;
    StripGeom blur_geom;
    DetectGeom det_geom;
    size_t frame_pyr = 0, frame_slots = 0;
    int sel_cap_total = 0, ncap = 0, max_kp = 0;
    size_t qt_lds = 0;
    int qt_kcap = 0;  // keys of a level k_quadtree<true> keeps in LDS
    // device state
    DevBuf d_geom, d_cells, d_xtab, d_ytab, d_yrow, d_pattern, d_rstrip, d_rsel, d_rwt, d_ctab, d_bcol;
    DevBuf d_pyr, d_blur, d_slots, d_cellcnt, d_dkey, d_dnode, d_sel, d_nsel, d_ncand, d_aux;
    DevBuf d_qtaux;  // k_qt_prefilter -> k_quadtree<true> (single frames): QT_AUX_FRAME ints per frame, < QT_BATCH_MIN frames
    bool qt_prefilter = false;  // every level has <= QT_INI_MAX initial nodes
    int max_slots_level = 0;
    DevBuf d_in, d_kps, d_desc, d_nout;  // staging for the host entry points
    void *h_pin = nullptr;               // ... and its pinned host side (input images | counts | key points | descriptors)
    size_t h_pin_bytes = 0;
    bool pinned_staging = true;          // ORBGPU_DEBUG_NO_PINNED: the runtime's pageable copies instead (A/B)
    DevBuf d_dbg;
    hipStream_t stream = nullptr;
    int last_batch = 0, last_cap = 0;
    // profiling: one event set per profiled call (up to PROF_SLOTS), averaged by stage_times()
    static constexpr int PROF_SLOTS = 256;
    bool profiling = false;
    std::vector<hipEvent_t> ev;  // PROF_SLOTS * 2 * ST_COUNT slots (ST_COUNT + 1 boundary events used per call), created lazily
    int prof_calls = 0;
    // host entry points: the 14-launch sequence is captured once per (size, batch, buffers) and replayed as a hipGraph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_key = 0;
    bool border_fast = false;  // level-0 column table present (width % 4 == 0)
    hipEvent_t stage_signal[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // caller's events
    hipEvent_t pipe_signal[8] = {};  // the same for an orbgpu_pipeline that owns this handle (stagger of its parts)
    bool counters_dirty = false;  // the cell counters may hold counts no k_quadtree has consumed
    bool force_batch_quadtree = false;  // ORBGPU_DEBUG_QT_BATCH: the batch variant of k_quadtree for any batch size (tests)
    // optional: the blur (HBM-bound) on a stream of the handle's own next to the FAST pass (VALU-bound) and the quadtree
    bool concurrent_blur = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_fork0 = nullptr, ev_join = nullptr;
    bool qt_no_prefilter = false;  // ORBGPU_DEBUG_QT_NOPRE: single frames filter their keys inside k_quadtree<true> (tests, A/B)
    int qt_keys_hook = -1;  // ORBGPU_DEBUG_QT_KEYS (read at creation): LDS key share of k_quadtree<true>; -1 = as many as fit
    int fast_queue_cap = FD_QCAP;  // row records a wave of k_fast_detect queues (ORBGPU_DEBUG_FAST_QUEUE shrinks it: tests)
    bool fast_early_out = false;   // k_fast_detect<true>: exact wave-level early-out (orbgpu_extractor_set_fast_early_out; ORBGPU_FAST_EARLY_OUT=1 at creation)
    int graph_state = 0;  // 0 = not tried, 1 = usable, -1 = capture failed: plain launches from then on
    // direct mode (Src0): level 0 read from the caller's image, no padded copy
    bool direct0_ok = false;       // the configured geometry has the tables for it (width % 8 == 0, >= 2 levels on the fast resize path)
    bool no_early_blur0 = false;   // ORBGPU_DEBUG_NO_EARLY_BLUR0: concurrent blur forks after the pyramid for level 0 too (A/B)
    int direct0_min_batch = 8;     // frames per call from which direct mode is used (ORBGPU_DEBUG_DIRECT0_MIN)
    bool no_direct0 = false;       // ORBGPU_DEBUG_NO_DIRECT0 (read at creation): always make the padded copy (tests, A/B)
    int rs_off_direct = 0;         // level 1's strip tables for the unpadded source
    DetectGeom det_geom_direct;    // k_fast_detect's strips with level 0 aligned to the image
    Src0 last_src = {nullptr, 0, 0u, 0};  // level-0 source of the last call (direct != 0: the call ran in direct mode)
    bool level0_materialized = true;      // d_pyr holds level 0 of the last call (false after a direct-mode call until a getter asks)
    TrigTable trig = {nullptr, nullptr, nullptr, 0u};  // the host libm's cosf / sinf exceptions (trig.hip); none in ORBGPU_TRIG_ROUNDED_DOUBLE mode

};

namespace orbgpu {

int trig_table_for_device(int device_id, TrigTable *out);  // trig.hip

// E0: constructor tables, ORBextractor.cc:410-470
static void build_tables(orbgpu_extractor *e)
{
    const int nl = e->nlevels;
    e->scale_factor_d = (double)e->prm.scale_factor;  // ORBextractor.h:98 keeps a double
    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        e->scale[i] = (float)((double)e->scale[i - 1] * e->scale_factor_d);
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    const float factor = (float)(1.0 / e->scale_factor_d);
    float desired = (float)e->prm.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        e->quota[l] = cv_round_host(desired);
        sum += e->quota[l];
        desired *= factor;
    }
    e->quota[nl - 1] = std::max(e->prm.nfeatures - sum, 0);

    int *um = e->umax.v;
    const int vmax = (int)floor(HALF_PATCH * sqrtf(2.f) / 2 + 1);
    const int vmin = (int)ceil(HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (int v = 0; v <= vmax; ++v)
        um[v] = cv_round_host(sqrt(hp2 - v * v));
    for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (um[v0] == um[v0 + 1])
            ++v0;
        um[v] = v0;
        ++v0;
    }
    // k_orient's "third piece" rule is written for these values (half widths 15 up to |v| = 3, >= 14 up to |v| = 6)
    static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int v = 0; v <= HALF_PATCH; v++)
        if (um[v] != expect[v])
            abort();
}

static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

// Level sizes, FAST cell grid, resize tables, slot layout for one image size.
static int configure(orbgpu_extractor *e, int w, int h, int batch)
{
    if (e->cfg_w == w && e->cfg_h == h && batch <= e->cfg_batch)
        return ORBGPU_OK;
    // a real reconfigure rewrites tables that kernels of an earlier call (on the caller's stream) may still read
    if (e->cfg_w != 0)
        ORBGPU_HIP_TRY(hipDeviceSynchronize());
    const int nl = e->nlevels;
    std::vector<LevelGeom> geom(nl);
    std::vector<CellDesc> cells;
    std::vector<XTab> xtab;
    std::vector<YTab> ytab;
    std::vector<YRow> yrow;
    std::vector<ResizeStrip> rstrip;
    std::vector<uint4> rsel, rwt;
    size_t plane_off = 0;
    int slot_off = 0, sel_off = 0, max_cells_level = 0, ncap = 0;
    for (int l = 0; l < nl; l++) {
        LevelGeom &g = geom[l];
        memset(&g, 0, sizeof(g));
        g.w = cv_round_host((float)w * e->inv_scale[l]);  // :1112
        g.h = cv_round_host((float)h * e->inv_scale[l]);
        g.max_bx = g.w - EDGE + 3;
        g.max_by = g.h - EDGE + 3;
        const float width = (float)(g.max_bx - BORDER0), height = (float)(g.max_by - BORDER0);
        g.ncols = (int)(width / CELL_W);
        g.nrows = (int)(height / CELL_W);
        ORBGPU_REQUIRE(g.ncols >= 1 && g.nrows >= 1 && g.w < 4096 + BORDER0 && g.h < 4096 + BORDER0,
                       "image %dx%d: pyramid level %d (%dx%d) is outside the supported range "
                       "(each level needs >= 62 px per side and < 4112 px)",
                       w, h, l, g.w, g.h);
        g.wcell = (int)ceilf(width / g.ncols);
        g.hcell = (int)ceilf(height / g.nrows);
        ORBGPU_REQUIRE(g.wcell <= MAX_CELL && g.hcell <= MAX_CELL, "FAST cell %dx%d too large", g.wcell, g.hcell);
        g.pitch = ((g.w + 2 * EDGE + 63) / 64) * 64;
        g.plane_off = (int)plane_off;
        plane_off += (((size_t)g.pitch * (g.h + 2 * EDGE) + 255) / 256) * 256;
        g.quota = e->quota[l];
        g.scale = e->scale[l];
        g.patch = (int)(PATCH * e->scale[l]);  // :837
        // DistributeOctTree initial nodes, :542-544
        g.n_ini = (int)roundf(width / height);
        ORBGPU_REQUIRE(g.n_ini >= 1, "level %d aspect ratio unsupported (nIni = 0 divides by zero in the reference)", l);
        g.hx = width / (float)g.n_ini;
        // cells, :789-808
        g.cell_first = (int)cells.size();
        g.slot_off = slot_off;
        for (int i = 0; i < g.nrows; i++) {
            const float iniY = (float)(BORDER0 + i * g.hcell);
            float maxY = iniY + g.hcell + 6;
            if (iniY >= g.max_by - 3)
                continue;
            if (maxY > g.max_by)
                maxY = (float)g.max_by;
            for (int j = 0; j < g.ncols; j++) {
                const float iniX = (float)(BORDER0 + j * g.wcell);
                float maxX = iniX + g.wcell + 6;
                if (iniX >= g.max_bx - 6)
                    continue;
                if (maxX > g.max_bx)
                    maxX = (float)g.max_bx;
                CellDesc c;
                c.level = (short)l;
                c.x0 = (short)iniX;
                c.x1 = (short)maxX;
                c.y0 = (short)iniY;
                c.y1 = (short)maxY;
                c.addx = (short)(j * g.wcell);
                c.addy = (short)(i * g.hcell);
                const int iw = std::max(c.x1 - c.x0 - 6, 0), ih = std::max(c.y1 - c.y0 - 6, 0);
                // strict 3x3 NMS: no two survivors are 8-neighbours
                c.cap = (short)(((iw + 1) / 2) * ((ih + 1) / 2));
                c.slot_off = slot_off;
                c.pitch = g.pitch;
                c.plane_off = g.plane_off;
                c.pad[0] = 0;
                slot_off += c.cap;
                cells.push_back(c);
            }
        }
        g.ncells = (int)cells.size() - g.cell_first;
        g.slot_cnt = slot_off - g.slot_off;
        max_cells_level = std::max(max_cells_level, g.ncells);
        // E3': a level yields at most max(4*nIni, quota+2) key points
        g.sel_cap = std::max(4 * g.n_ini, g.quota + 3) + 1;
        g.sel_off = sel_off;
        sel_off += g.sel_cap;
        ncap = std::max(ncap, g.sel_cap);
        // resize tables (cv::resize INTER_LINEAR 8U, A2)
        g.xtab_off = (int)xtab.size();
        g.ytab_off = (int)ytab.size();
        if (l > 0) {
            const int sw = geom[l - 1].w, sh = geom[l - 1].h;
            const double scale_x = 1. / ((double)g.w / sw), scale_y = 1. / ((double)g.h / sh);
            for (int dx = 0; dx < g.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floor(fx);
                fx -= sx;
                if (sx < 0) {
                    fx = 0;
                    sx = 0;
                }
                if (sx >= sw - 1) {
                    fx = 0;
                    sx = sw - 1;
                }
                XTab t;
                t.sx = (uint16_t)sx;
                t.sx1 = (uint16_t)std::min(sx + 1, sw - 1);
                t.a0 = (uint16_t)sat_short(cv_round_host((1.f - fx) * 2048));
                t.a1 = (uint16_t)sat_short(cv_round_host(fx * 2048));
                xtab.push_back(t);
            }
            for (int dy = 0; dy < g.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floor(fy);
                fy -= sy;
                YTab t;
                t.sy0 = (uint16_t)std::min(std::max(sy, 0), sh - 1);
                t.sy1 = (uint16_t)std::min(std::max(sy + 1, 0), sh - 1);
                t.b0 = sat_short(cv_round_host((1.f - fy) * 2048));
                t.b1 = sat_short(cv_round_host(fy * 2048));
                ytab.push_back(t);
            }
            g.yrow_off = (int)yrow.size();
            for (int py = 0; py < g.h + 2 * EDGE; py++) {
                const YTab &t = ytab[g.ytab_off + reflect101(py - EDGE, g.h)];
                yrow.push_back(YRow{(int)t.sy0, (int)t.sy1, (int)t.b0, (int)t.b1});
            }
            // fast-path strip tables: one entry per aligned output dword of the padded row
            g.rs_off = (int)rstrip.size();
            g.rs_fast = 1;
            for (int sdw = 0; sdw < g.pitch / 4; sdw++) {
                int cl[4], cr[4];
                const XTab *xt[4];
                int mn = 1 << 30;
                for (int k = 0; k < 4; k++) {
                    const int px = std::min(sdw * 4 + k, g.w + 2 * EDGE - 1);
                    const int dx = reflect101(px - EDGE, g.w);
                    xt[k] = &xtab[g.xtab_off + dx];
                    cl[k] = xt[k]->sx + EDGE;
                    cr[k] = xt[k]->sx1 + EDGE;
                    mn = std::min(mn, cl[k]);
                }
                // the 12-byte window starts at the aligned column wbase; the kernel shifts it by sh = mn - wbase bytes (two
                // v_alignbyte per source row) so that all eight taps of the four outputs lie in ONE 8-byte pair
                const int wbase = mn & ~3, sh = mn - wbase;
                ResizeStrip rsx;
                rsx.base_q = (uint32_t)wbase | ((uint32_t)sh << 16);
                uint32_t sel[4], wt[4];
                for (int k = 0; k < 4; k++) {
                    const int ol = cl[k] - mn, orr = cr[k] - mn;
                    if (ol < 0 || orr < ol || orr > 7)
                        g.rs_fast = 0;
                    sel[k] = (uint32_t)(ol & 7) | 0x0c00u | ((uint32_t)(orr & 7) << 16) | 0x0c000000u;
                    wt[k] = (uint32_t)xt[k]->a0 | ((uint32_t)xt[k]->a1 << 16);
                }
                rstrip.push_back(rsx);
                rsel.push_back(uint4{sel[0], sel[1], sel[2], sel[3]});
                rwt.push_back(uint4{wt[0], wt[1], wt[2], wt[3]});
            }
        }
    }
    // direct mode: level 1's strips once more, for source columns counted from the image's own column 0 (Src0).  A window
    // that would run past the end of an image row (the next row, or -- last row of the last frame -- the end of the
    // caller's buffer) is placed on the row's last 12 bytes and the pair is cut from its second and third dword (bit 18).
    bool direct_ok = nl >= 2 && w % 8 == 0 && w >= 64 && geom[std::min(1, nl - 1)].rs_fast != 0;
    int rs_off_direct = 0;
    if (direct_ok) {
        const LevelGeom &g = geom[1];
        const int sw = geom[0].w;
        rs_off_direct = (int)rstrip.size();
        for (int sdw = 0; sdw < g.pitch / 4 && direct_ok; sdw++) {
            int cl[4], cr[4], mn = 1 << 30;
            const XTab *xt[4];
            for (int k = 0; k < 4; k++) {
                const int px = std::min(sdw * 4 + k, g.w + 2 * EDGE - 1);
                xt[k] = &xtab[g.xtab_off + reflect101(px - EDGE, g.w)];
                cl[k] = xt[k]->sx;
                cr[k] = xt[k]->sx1;
                mn = std::min(mn, cl[k]);
            }
            int wbase = mn & ~3, ps = mn;  // ps: first byte of the 8-byte pair the selectors index
            int pbase = wbase;             // first byte of the two dwords the pair is cut from
            uint32_t edge = 0;
            if (wbase + 12 > sw) {  // the window would pass the end of the row: the row's last 12 bytes, pair from dwords 1 and 2
                wbase = sw - 12;
                pbase = sw - 8;
                ps = std::min(mn, pbase + 3);
                edge = 1u << 18;
            }
            const int sh = ps - pbase;
            uint32_t sel[4], wt[4];
            for (int k = 0; k < 4; k++) {
                const int ol = cl[k] - ps, orr = cr[k] - ps;
                // inside the pair, and -- at a row end -- inside its valid part (the bytes above 7 - sh come from the re-read dword)
                if (sh < 0 || sh > 3 || wbase < 0 || ol < 0 || orr < ol || orr > (edge ? 7 - sh : 7))
                    direct_ok = false;
                sel[k] = (uint32_t)(ol & 7) | 0x0c00u | ((uint32_t)(orr & 7) << 16) | 0x0c000000u;
                wt[k] = (uint32_t)xt[k]->a0 | ((uint32_t)xt[k]->a1 << 16);
            }
            rstrip.push_back(ResizeStrip{(uint32_t)wbase | ((uint32_t)sh << 16) | edge});
            rsel.push_back(uint4{sel[0], sel[1], sel[2], sel[3]});
            rwt.push_back(uint4{wt[0], wt[1], wt[2], wt[3]});
        }
    }
    ncap = std::max(ncap, (max_cells_level + 3) / 4);  // the cell scan reuses the [ncap*4] child-count array
    ncap = ((ncap + 7) / 8) * 8;
    const size_t qt_lds = (size_t)ncap * (2 * sizeof(short4) + 2 * sizeof(int) + 8 * sizeof(int) + 2 * sizeof(int) +
                                          4 * sizeof(uint16_t) + 4 * sizeof(uint16_t) + 1 + 2 * sizeof(uint32_t)) + 64;
    // single frames: keys (4 B) and node ids (2 B) of a level in LDS, as many as fit (never more than a level can hold)
    int max_slots_level = 0;
    for (int l = 0; l < nl; l++)
        max_slots_level = std::max(max_slots_level, geom[l].slot_cnt);
    int qt_kcap = max_slots_level;
    if (e->qt_keys_hook >= 0)  // test hook: forces the mixed LDS / memory path
        qt_kcap = e->qt_keys_hook;
    qt_kcap = (int)std::min<size_t>((size_t)(qt_kcap + 7) / 8 * 8, qt_lds < 150 * 1024 ? (150 * 1024 - qt_lds) / 6 / 8 * 8 : 0);
    ORBGPU_REQUIRE(qt_lds <= (size_t)QT_LDS_MAX, "nfeatures too large for the quadtree kernel (needs %zu B of LDS)", qt_lds);
    ORBGPU_REQUIRE((size_t)slot_off < (1u << 23), "too many FAST key slots per frame");

    e->geom = geom;
    e->cells = cells;
    {
        StripGeom &bgm = e->blur_geom;
        memset(&bgm, 0, sizeof(bgm));
        bgm.nlevels = nl;
        int acc = 0;
        for (int l = 0; l < nl; l++) {
            // 8-pixel strips (two dwords) start at padded dword 4 (bytes 16..19 hold image column 0) and must cover
            // column w-1
            const int nsx = ((geom[l].w + EDGE - 1) / 4 - 4 + 1 + 1) / 2;
            const int nsy = (geom[l].h + BLUR_ROWS - 1) / BLUR_ROWS;
            bgm.first[l] = acc;
            bgm.nsx[l] = nsx;
            acc += nsx * nsy;
        }
        for (int l = nl; l <= ORBGPU_MAX_LEVELS; l++)
            bgm.first[l] = acc;
    }
    // k_fast_detect: bands (rows of cells) x dword columns, and what every dword column knows about the cell grid.  Built
    // twice: every level a padded plane (strips = its aligned dwords from dword 9: image column 17), and with level 0 read
    // from the image itself (direct mode: strips aligned to the image, from column 16)
    std::vector<ColumnInfo> ctab;
    for (int l = 0; l < nl; l++) {
        const LevelGeom &g = geom[l];
        int ncols_eff = 0;  // cell columns that are not skipped (ORBextractor.cc:797: iniX >= maxBorderX-6)
        for (int j = 0; j < g.ncols; j++)
            if (BORDER0 + j * g.wcell < g.max_bx - 6)
                ncols_eff++;
        const int nbands = (g.h - 2 * EDGE + g.hcell - 1) / g.hcell;  // bands with interior rows
        ORBGPU_REQUIRE(ncols_eff > 0 && g.ncells % ncols_eff == 0 && nbands <= g.ncells / ncols_eff && ncols_eff < 255 &&
                           g.ncells / ncols_eff < 256 && g.hcell < 128,
                       "unexpected cell grid at level %d", l);
        geom[l].ncols_eff = ncols_eff;
        for (int which = 0; which < 2; which++) {
            const int d = which ? g.hcell : g.wcell;
            const uint32_t m = (uint32_t)(((1ull << 32) + (uint64_t)d - 1u) / (uint64_t)d);
            for (uint32_t v = 0; v < 8192u; v++)
                ORBGPU_REQUIRE((uint32_t)(((uint64_t)v * m) >> 32) == v / (uint32_t)d, "cell size %d at level %d has no exact reciprocal", d, l);
            (which ? geom[l].inv_hcell : geom[l].inv_wcell) = m;
        }
    }
    auto build_detect = [&](bool direct0, DetectGeom &dgm) -> bool {
        memset(&dgm, 0, sizeof(dgm));
        dgm.nlevels = nl;
        int acc = 0;
        for (int l = 0; l < nl; l++) {
            const LevelGeom &g = geom[l];
            const bool dir = direct0 && l == 0;
            const int xfirst = dir ? 16 : 9 * 4 - EDGE;
            // strips must reach the last scored column, w - 20 (padded planes: up to their aligned dword (w-1)/4)
            const int nsx = dir ? (g.w - 20 - xfirst) / 4 + 1 : (g.w - 1) / 4 - 9 + 1;
            const int nbands = (g.h - 2 * EDGE + g.hcell - 1) / g.hcell;
            dgm.first[l] = acc;
            dgm.nsx[l] = nsx;
            dgm.ncols[l] = g.ncols_eff;
            dgm.ctab_off[l] = (int)ctab.size();
            dgm.xfirst[l] = xfirst;
            acc += nsx * nbands;
            if (dir)  // level 0's strips end on a wave boundary: a wave of k_fast_detect then reads ONE plane (image or pyramid)
                acc = (acc + FD_OWN - 1) / FD_OWN * FD_OWN;
            for (int sx = 0; sx < nsx; sx++) {
                ColumnInfo c{0u, 0u};
                for (int p = 0; p < 4; p++) {
                    const int x = xfirst + 4 * sx + p;  // image column
                    int j = 0xFF;
                    if (x >= EDGE && x < g.w - EDGE) {
                        j = (x - EDGE) / g.wcell;
                        const int xs = EDGE + j * g.wcell, xe = std::min(xs + g.wcell, g.w - EDGE);  // interior [xs, xe)
                        if (j < g.ncols_eff) {
                            c.flags |= 1u << p;
                            if (x == xs)
                                c.flags |= 16u << p;
                            if (x == xe - 1)
                                c.flags |= 256u << p;
                        } else
                            j = 0xFF;
                    }
                    c.cellj |= (uint32_t)j << (8 * p);
                }
                ctab.push_back(c);
            }
        }
        for (int l = nl; l <= ORBGPU_MAX_LEVELS; l++)
            dgm.first[l] = acc;
        // the key emission of k_fast_detect addresses per-wave counters with (cell id mod FD_CELLS): the cells of the
        // columns a wave owns must be a range of at most FD_CELLS consecutive ids
        std::vector<std::pair<int, int>> strip_cells;  // per flat strip: lowest / highest cell id (-1: none)
        for (int l = 0; l < nl; l++) {
            const LevelGeom &g = geom[l];
            const int nbands = (g.h - 2 * EDGE + g.hcell - 1) / g.hcell;
            strip_cells.resize((size_t)dgm.first[l], {-1, -1});  // (the padding strips in front of this level)
            for (int b = 0; b < nbands; b++)
                for (int sx = 0; sx < dgm.nsx[l]; sx++) {
                    const ColumnInfo &c = ctab[dgm.ctab_off[l] + sx];
                    int lo = -1, hi = -1;
                    for (int p = 0; p < 4; p++)
                        if ((c.flags >> p) & 1u) {
                            const int id = g.cell_first + b * dgm.ncols[l] + (int)((c.cellj >> (8 * p)) & 255u);
                            lo = lo < 0 ? id : std::min(lo, id);
                            hi = std::max(hi, id);
                        }
                    strip_cells.push_back({lo, hi});
                }
        }
        for (size_t w0 = 0; w0 < strip_cells.size(); w0 += FD_OWN) {
            int lo = -1, hi = -1;
            for (size_t q = w0; q < std::min(w0 + FD_OWN, strip_cells.size()); q++)
                if (strip_cells[q].first >= 0) {
                    lo = lo < 0 ? strip_cells[q].first : std::min(lo, strip_cells[q].first);
                    hi = std::max(hi, strip_cells[q].second);
                }
            if (hi - lo >= FD_CELLS)
                return false;
        }
        return true;
    };
    ORBGPU_REQUIRE(build_detect(false, e->det_geom), "image too small for the FAST kernel's cell bookkeeping (more than %d cells in one wave)", FD_CELLS);
    if (direct_ok)
        direct_ok = build_detect(true, e->det_geom_direct);
    e->direct0_ok = direct_ok;
    e->rs_off_direct = rs_off_direct;
    e->geom = geom;  // (with the cell-grid fields filled in above)
    // k_border0_fast: per aligned dword of the padded level-0 row, the two adjacent source dwords and the byte selector
    std::vector<BorderCol> bcol;
    if (w % 4 == 0 && w >= 8) {
        const LevelGeom &g0 = geom[0];
        for (int c = 0; c < g0.pitch / 4; c++) {
            int sx[4], lo = 1 << 30, hi = 0;
            for (int k = 0; k < 4; k++) {
                sx[k] = reflect101(std::min(c * 4 + k, g0.w + 2 * EDGE - 1) - EDGE, g0.w);
                lo = std::min(lo, sx[k]);
                hi = std::max(hi, sx[k]);
            }
            const int d = std::min(lo / 4, g0.w / 4 - 2);  // src[d + 1] stays inside the row
            ORBGPU_REQUIRE(hi < d * 4 + 8 && lo >= d * 4, "level-0 border table: span of column %d", c);
            BorderCol bc{(uint32_t)d, 0u};
            for (int k = 0; k < 4; k++)
                bc.sel |= (uint32_t)(sx[k] - d * 4) << (8 * k);  // bytes 0..3 = src[d], 4..7 = src[d+1]
            bcol.push_back(bc);
        }
    }
    e->frame_pyr = plane_off;
    e->frame_slots = (size_t)slot_off;
    e->sel_cap_total = sel_off;
    e->ncap = ncap;
    e->qt_lds = qt_lds;
    e->qt_kcap = qt_kcap;
    e->max_slots_level = max_slots_level;
    e->qt_prefilter = !e->qt_no_prefilter;
    for (int l = 0; l < nl; l++)
        e->qt_prefilter = e->qt_prefilter && geom[l].n_ini <= QT_INI_MAX;
    int max_kp = 0;
    for (int l = 0; l < nl; l++)
        max_kp += geom[l].sel_cap - 1;
    e->max_kp = max_kp;

    int rc;
#define RSV(buf, n) if ((rc = (buf).reserve(n)) != ORBGPU_OK) return rc
    RSV(e->d_geom, sizeof(LevelGeom) * nl);
    RSV(e->d_cells, sizeof(CellDesc) * cells.size());
    RSV(e->d_ctab, sizeof(ColumnInfo) * ctab.size());
    RSV(e->d_bcol, sizeof(BorderCol) * std::max<size_t>(bcol.size(), 1));
    RSV(e->d_xtab, sizeof(XTab) * std::max<size_t>(xtab.size(), 1));
    RSV(e->d_ytab, sizeof(YTab) * std::max<size_t>(ytab.size(), 1));
    RSV(e->d_yrow, sizeof(YRow) * std::max<size_t>(yrow.size(), 1));
    RSV(e->d_rstrip, sizeof(ResizeStrip) * std::max<size_t>(rstrip.size(), 1));
    RSV(e->d_rsel, sizeof(uint4) * std::max<size_t>(rsel.size(), 1));
    RSV(e->d_rwt, sizeof(uint4) * std::max<size_t>(rwt.size(), 1));
    RSV(e->d_pattern, 1024 + 2 * 4 * 16 * 9 * sizeof(uint32_t));  // rBRIEF pattern, then k_orient's weight tables
    const size_t B = (size_t)batch;
    RSV(e->d_pyr, e->frame_pyr * B);
    RSV(e->d_blur, e->frame_pyr * B);
    RSV(e->d_slots, sizeof(uint32_t) * e->frame_slots * B);
    RSV(e->d_cellcnt, sizeof(int) * (cells.size() + ORBGPU_MAX_LEVELS) * B);  // per frame: cell counters, then one stored-key counter per level
    RSV(e->d_dkey, sizeof(uint32_t) * e->frame_slots * B);
    RSV(e->d_dnode, sizeof(uint16_t) * e->frame_slots * B);
    RSV(e->d_sel, sizeof(uint32_t) * (size_t)sel_off * B);
    RSV(e->d_nsel, sizeof(int) * nl * B);
    RSV(e->d_ncand, sizeof(int) * nl * B);
    RSV(e->d_qtaux, sizeof(int) * QT_AUX_FRAME * QT_BATCH_MIN);
#undef RSV
    // Tables and initial values go through the handle's OWN stream and are waited for below.  The synchronous forms are
    // not a substitute: hipMemset (like cudaMemset) is asynchronous with respect to the host and runs on the null stream,
    // which the handle's non-blocking stream is not ordered with -- under load (several host threads making their first
    // calls at once) the zeroing of the blurred planes and of the cell counters arrived AFTER k_blur / k_fast_detect had
    // written them: descriptors of zeros, wrong key points, once a memory fault (round 4, tools/stress_first_calls.py).
    ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_geom.p, geom.data(), sizeof(LevelGeom) * nl, hipMemcpyHostToDevice, e->stream));
    ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_cells.p, cells.data(), sizeof(CellDesc) * cells.size(), hipMemcpyHostToDevice, e->stream));
    ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_ctab.p, ctab.data(), sizeof(ColumnInfo) * ctab.size(), hipMemcpyHostToDevice, e->stream));
    if (!bcol.empty())
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_bcol.p, bcol.data(), sizeof(BorderCol) * bcol.size(), hipMemcpyHostToDevice, e->stream));
    e->border_fast = !bcol.empty();
    if (!xtab.empty()) {
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_xtab.p, xtab.data(), sizeof(XTab) * xtab.size(), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_ytab.p, ytab.data(), sizeof(YTab) * ytab.size(), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_yrow.p, yrow.data(), sizeof(YRow) * yrow.size(), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_rstrip.p, rstrip.data(), sizeof(ResizeStrip) * rstrip.size(), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_rsel.p, rsel.data(), sizeof(uint4) * rsel.size(), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_rwt.p, rwt.data(), sizeof(uint4) * rwt.size(), hipMemcpyHostToDevice, e->stream));
    }
    ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_pattern.p, k_pattern_host, 1024, hipMemcpyHostToDevice, e->stream));
    {
        // k_orient's byte-weight tables per (alignment a of the disc's first column, |v|, dword j of the 9-dword row):
        // W10 holds u + 16 inside the disc (0 outside), M01 holds 1 inside the disc
        std::vector<uint32_t> orw(2 * 4 * 16 * 9, 0u);
        for (int i = 0; i < 4 * 16 * 9; i++) {
            const int a = i / (16 * 9), av = (i / 9) % 16, j = i % 9;
            for (int k = 0; k < 4; k++) {
                const int u = 4 * j + k - a - HALF_PATCH;
                if (std::abs(u) <= e->umax.v[av]) {
                    orw[i] |= (uint32_t)(u + 16) << (8 * k);
                    orw[4 * 16 * 9 + i] |= 1u << (8 * k);
                }
            }
        }
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_pattern.as<uint8_t>() + 1024, orw.data(), orw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
        ORBGPU_HIP_TRY(hipStreamSynchronize(e->stream));  // (orw leaves scope)
    }
    // the blurred planes are only written inside the image; define the rest once
    ORBGPU_HIP_TRY(hipMemsetAsync(e->d_blur.p, 0, e->frame_pyr * B, e->stream));
    // cell counters start at zero; k_quadtree re-arms them after reading
    ORBGPU_HIP_TRY(hipMemsetAsync(e->d_cellcnt.p, 0, sizeof(int) * (cells.size() + ORBGPU_MAX_LEVELS) * B, e->stream));
    ORBGPU_HIP_TRY(hipMemsetAsync(e->d_qtaux.p, 0, e->d_qtaux.bytes, e->stream));  // likewise (k_quadtree<true> re-arms what it reads)
    ORBGPU_HIP_TRY(hipStreamSynchronize(e->stream));  // ... before any stream (the caller's, for the device entry points) reads them
    // The attribute belongs to the function (per device), not to this handle: another handle with a larger geometry may
    // have raised it and still launch, so it is set once to the most the kernel can ever be launched with here
    // (qt_lds <= 159 KB and qt_lds + 6 qt_kcap <= 150 KB are enforced above; the static part is < 200 B).
    ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_quadtree<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, QT_LDS_MAX));
    ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_quadtree<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, QT_LDS_MAX));
    e->cfg_w = w;
    e->cfg_h = h;
    e->cfg_batch = batch;
    return ORBGPU_OK;
}

// Where the kernels of a call go: onto a stream, or -- host entry points, from the second call of a configuration on --
// into a hipGraph as a chain of kernel nodes.  The graph is BUILT (hipGraphAddKernelNode), not captured: a stream
// capture is process-wide state on this runtime -- another thread's hipFree / hipMalloc / synchronous copy invalidated
// captures in flight (in every capture mode), and launches of other threads then failed with "operation failed due to a
// previous error during capture" (round 4, tools/stress_first_calls.py).  Nothing here touches global state.
struct Launcher {
    hipStream_t st = nullptr;
    hipGraph_t graph = nullptr;      // non-null: record instead of launching
    hipGraphNode_t last = nullptr;   // tail of the chain
    hipError_t err = hipSuccess;     // first error while recording

    Launcher() = default;
    explicit Launcher(hipStream_t s) : st(s) {}

    template <typename... P, typename... A> void launch(void (*kern)(P...), dim3 grid, dim3 block, size_t lds, A &&...a)
    {
        static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
        if (!graph) {
            hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<P>(a)...);
            return;
        }
        std::tuple<std::remove_cv_t<P>...> vals(static_cast<P>(a)...);  // the node copies the values when it is added
        void *ptrs[sizeof...(P) + 1];
        fill(ptrs, vals, std::index_sequence_for<P...>{});
        hipKernelNodeParams kp = {};
        kp.func = reinterpret_cast<void *>(kern);
        kp.gridDim = grid;
        kp.blockDim = block;
        kp.sharedMemBytes = (unsigned)lds;
        kp.kernelParams = ptrs;
        kp.extra = nullptr;
        add([&](hipGraphNode_t *n, const hipGraphNode_t *dep, size_t ndep) { return hipGraphAddKernelNode(n, graph, dep, ndep, &kp); });
    }
    int memset32(void *dst, size_t bytes)  // zero `bytes` (a multiple of 4) at dst
    {
        if (!graph) {
            ORBGPU_HIP_TRY(hipMemsetAsync(dst, 0, bytes, st));
            return ORBGPU_OK;
        }
        hipMemsetParams mp = {};
        mp.dst = dst;
        mp.elementSize = 4;
        mp.width = bytes / 4;
        mp.height = 1;
        mp.pitch = bytes;
        mp.value = 0;
        add([&](hipGraphNode_t *n, const hipGraphNode_t *dep, size_t ndep) { return hipGraphAddMemsetNode(n, graph, dep, ndep, &mp); });
        return ORBGPU_OK;
    }

private:
    template <typename T, size_t... I> static void fill(void **ptrs, T &vals, std::index_sequence<I...>)
    {
        ((ptrs[I] = const_cast<void *>(static_cast<const void *>(&std::get<I>(vals)))), ...);
    }
    template <typename F> void add(F f)
    {
        hipGraphNode_t n = nullptr;
        const hipError_t e = f(&n, last ? &last : nullptr, last ? 1 : 0);
        if (e != hipSuccess && err == hipSuccess)
            err = e;
        if (e == hipSuccess)
            last = n;
    }
};

// Level 0 as a padded plane: copyMakeBorder(image, REFLECT_101) (ORBextractor.cc:1125-1129).  Every call when the input is
// not aligned for direct mode; otherwise only when mvImagePyramid[0] is asked for (the getters below).
static int materialize_level0(orbgpu_extractor *e, const Src0 &s0, int batch, Launcher &L)
{
    const LevelGeom &g = e->geom[0];
    const LevelGeom *dg = e->d_geom.as<LevelGeom>();
    uint8_t *pyr = e->d_pyr.as<uint8_t>();
    const size_t stride = s0.pitch;
    if (e->border_fast && stride % 4 == 0 && s0.frame_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(s0.p) & 3) == 0) {
        dim3 gridf(((g.pitch / 16) * ((g.h + 2 * EDGE + BORDER_ROWS - 1) / BORDER_ROWS) + 255) / 256, batch);
        L.launch(k_border0_fast, gridf, dim3(256), 0, s0.p, stride, s0.frame_stride, pyr, e->frame_pyr, dg,
                           e->d_bcol.as<BorderCol>());
    } else {
        dim3 grid(((g.pitch / 4) * ((g.h + 2 * EDGE + PYR_ROWS - 1) / PYR_ROWS) + 255) / 256, batch);
        L.launch(k_border0, grid, dim3(256), 0, s0.p, stride, s0.frame_stride, pyr, e->frame_pyr, dg);
    }
    if (!L.graph)
        ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

// the blur of all levels: one launch, or -- direct mode -- level 0 from the image (k_blur0_direct) + the other levels
// (which: 0 = everything, 1 = only the direct-mode level 0, which needs nothing but the image, 2 = only the rest)
static void launch_blur(orbgpu_extractor *e, const Src0 &s0, int batch, Launcher &L, int which = 0)
{
    const int nl = e->nlevels;
    const LevelGeom *dg = e->d_geom.as<LevelGeom>();
    uint8_t *pyr = e->d_pyr.as<uint8_t>(), *blur = e->d_blur.as<uint8_t>();
    int strip0 = 0;
    if (s0.direct) {
        const LevelGeom &g = e->geom[0];
        const int nstrips = (g.w / 8) * ((g.h + BLUR_ROWS - 1) / BLUR_ROWS);
        if (which != 2)
            L.launch(k_blur0_direct, dim3((nstrips + 255) / 256, batch), dim3(256), 0, s0, blur, e->frame_pyr, dg);
        strip0 = e->blur_geom.first[1];
    }
    if (which == 1)
        return;
    if (e->blur_geom.first[nl] > strip0)
        L.launch(k_blur, dim3((e->blur_geom.first[nl] - strip0 + 255) / 256, batch), dim3(256), 0, pyr, blur,
                           e->frame_pyr, dg, e->blur_geom, strip0);
}

static int launch_pipeline(orbgpu_extractor *e, const uint8_t *d_gray, int batch, int w, int h, size_t stride,
                           size_t frame_stride, orbgpu_keypoint *d_kps, uint8_t *d_desc, int cap, int *d_n_out,
                           Launcher &L)
{
    hipStream_t st = L.st;  // (events and the side stream below: plain launches only)
    const int nl = e->nlevels;
    const LevelGeom *dg = e->d_geom.as<LevelGeom>();
    uint8_t *pyr = e->d_pyr.as<uint8_t>();
    uint8_t *blur = e->d_blur.as<uint8_t>();
    int rc = e->d_aux.reserve(sizeof(KpAux) * (size_t)cap * batch);
    if (rc != ORBGPU_OK)
        return rc;
    const bool rec = L.graph != nullptr;  // recording a graph: no events, no side stream
    const bool prof = !rec && e->profiling && e->prof_calls < orbgpu_extractor::PROF_SLOTS;
    if (prof && e->ev.empty()) {
        e->ev.resize((size_t)orbgpu_extractor::PROF_SLOTS * 2 * ST_COUNT, nullptr);
        for (auto &x : e->ev)
            ORBGPU_HIP_TRY(hipEventCreate(&x));
    }
    hipEvent_t *evs = prof ? &e->ev[(size_t)e->prof_calls * 2 * ST_COUNT] : nullptr;
// one event per stage boundary (an event costs the stream several microseconds): evs[0] before the first stage,
// evs[1 + stage] after each; stage time = evs[1 + stage] - evs[stage]
#define BEGIN(stage, s) if (prof && (stage) == 0) ORBGPU_HIP_TRY(hipEventRecord(evs[0], s))
#define END(stage, s)                                                          \
    {                                                                          \
        if (prof)                                                              \
            ORBGPU_HIP_TRY(hipEventRecord(evs[1 + (stage)], s));               \
        if (!rec && e->stage_signal[stage])                                    \
            ORBGPU_HIP_TRY(hipEventRecord(e->stage_signal[stage], s));         \
        if (!rec && e->pipe_signal[stage])                                     \
            ORBGPU_HIP_TRY(hipEventRecord(e->pipe_signal[stage], s));          \
    }
    // direct mode: aligned rows, and strides the 24-bit row-offset multiplies cover (checked by the callers: stride < 2^24)
    const bool aligned4 = stride % 4 == 0 && frame_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(d_gray) & 3) == 0;
    // ... and a batch: for a handful of frames the kernels are latency-bound, the traffic direct mode saves is served from
    // L2 anyway, and its second blur launch costs more (0.030 against 0.016 ms for one 640x480 frame) than the border kernel
    // it removes (0.036 against 0.039 ms); ORBGPU_DEBUG_DIRECT0_MIN overrides the threshold (tests run batch 1 - 3 in direct mode)
    const bool direct = e->direct0_ok && !e->no_direct0 && aligned4 && batch >= e->direct0_min_batch;
    const Src0 s0{d_gray, frame_stride, (uint32_t)stride, direct ? 1 : 0};
    e->last_src = s0;
    e->level0_materialized = !direct;
    BEGIN(ST_PYRAMID, st);
    // concurrent blur in direct mode: level 0's blur depends on the image only -- it starts on the side stream now, next
    // to the resize chain (seven dependent launches whose small levels leave most of the device idle)
    const bool side_blur = e->concurrent_blur && !rec;  // (a recorded graph is a chain)
    Launcher Ls(e->side);
    const bool early_blur0 = side_blur && direct && !e->no_early_blur0;
    if (early_blur0) {
        ORBGPU_HIP_TRY(hipEventRecord(e->ev_fork0, st));
        ORBGPU_HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork0, 0));
        if (prof)
            ORBGPU_HIP_TRY(hipEventRecord(evs[ST_COUNT + 1], e->side));
        launch_blur(e, s0, batch, Ls, 1);
    }
    {
        if (!direct) {
            int rcb = materialize_level0(e, s0, batch, L);
            if (rcb != ORBGPU_OK)
                return rcb;
        }
        for (int l = 1; l < nl; l++) {
            const LevelGeom &gl = e->geom[l];
            dim3 gr((gl.pitch / 4 + 255) / 256, gl.h + 2 * EDGE, batch);
            // rows per thread: 8 on the large levels; the small ones get more (shorter) threads to hide latency with
            const int ndw_l = (gl.pitch / 4) * (gl.h + 2 * EDGE);
            const int rows = ndw_l >= 32768 ? PYR_ROWS : ndw_l >= 16384 ? PYR_ROWS / 2 : PYR_ROWS / 4;
            dim3 grf(((gl.pitch / 4) * ((gl.h + 2 * EDGE + rows - 1) / rows) + 255) / 256, batch);
            if (gl.rs_fast) {
                const bool dir1 = direct && l == 1;  // level 1 straight from the image
                auto kern = dir1 ? (rows == PYR_ROWS ? k_resize_fast<true, true> : k_resize_fast<false, true>)
                                 : (rows == PYR_ROWS ? k_resize_fast<true, false> : k_resize_fast<false, false>);
                L.launch(kern, grf, dim3(256), 0, pyr, e->frame_pyr, dg, l, e->d_rstrip.as<ResizeStrip>(),
                                   e->d_rsel.as<uint4>(), e->d_rwt.as<uint4>(), e->d_yrow.as<YRow>(),
                                   dir1 ? e->rs_off_direct : gl.rs_off, rows, s0);
            }
            else
                L.launch(k_resize_level, gr, dim3(256), 0, pyr, e->frame_pyr, dg, l,
                                   e->d_xtab.as<XTab>(), e->d_ytab.as<YTab>());
        }
    }
    END(ST_PYRAMID, st);
    // The blur only depends on the pyramid.  It runs here, in front of the VALU-bound FAST pass, because the kernel that
    // follows it pays for the write-back of the blurred planes (k_describe 105 instead of 87 us behind it, k_orient 119
    // instead of 93), and FAST has bandwidth to spare.  (On a side stream next to the quadtree both were slower.)
    BEGIN(ST_BLUR, st);
    if (side_blur) {
        // fork: the side stream waits for the pyramid, blurs, and is joined in front of the descriptor stage
        ORBGPU_HIP_TRY(hipEventRecord(e->ev_fork, st));
        ORBGPU_HIP_TRY(hipStreamWaitEvent(e->side, e->ev_fork, 0));
        if (prof && !early_blur0)
            ORBGPU_HIP_TRY(hipEventRecord(evs[ST_COUNT + 1], e->side));
        launch_blur(e, s0, batch, Ls, early_blur0 ? 2 : 0);
        if (prof)
            ORBGPU_HIP_TRY(hipEventRecord(evs[ST_COUNT + 2], e->side));
        ORBGPU_HIP_TRY(hipEventRecord(e->ev_join, e->side));
    } else
        launch_blur(e, s0, batch, L);
    END(ST_BLUR, st);
    BEGIN(ST_FAST, st);
    if (e->counters_dirty) {  // a previous call enqueued the FAST pass but not the quadtree that re-arms the counters
        int rcm = L.memset32(e->d_cellcnt.p, e->d_cellcnt.bytes & ~(size_t)3);
        if (rcm == ORBGPU_OK)
            rcm = L.memset32(e->d_qtaux.p, e->d_qtaux.bytes & ~(size_t)3);
        if (rcm != ORBGPU_OK)
            return rcm;
    }
    e->counters_dirty = true;
    {
        const int t_ini = std::max(e->prm.ini_th_fast, 1), t_min = std::max(e->prm.min_th_fast, 1);
        const DetectGeom &dgeo = direct ? e->det_geom_direct : e->det_geom;
        const int nwaves = (dgeo.first[nl] + FD_OWN - 1) / FD_OWN;
        L.launch(e->fast_early_out ? k_fast_detect<true> : k_fast_detect<false>, dim3((nwaves + 3) / 4, batch), dim3(256), 0, pyr, e->frame_pyr, dg,
                           dgeo, e->d_ctab.as<ColumnInfo>(), e->d_cells.as<CellDesc>(), (int)e->cells.size(),
                           e->d_slots.as<uint32_t>(), e->frame_slots, e->d_cellcnt.as<int>(), std::min(t_ini, t_min),
                           t_ini, e->fast_queue_cap, s0);
    }
    END(ST_FAST, st);
    BEGIN(ST_QUADTREE, st);
    if (batch >= QT_BATCH_MIN || e->force_batch_quadtree)
        L.launch(k_quadtree<false>, dim3(batch, nl), dim3(QT_THREADS_BATCH), e->qt_lds, dg,
                           e->d_cells.as<CellDesc>(), (int)e->cells.size(), e->d_slots.as<uint32_t>(), e->frame_slots,
                           e->d_cellcnt.as<int>(), e->d_dkey.as<uint32_t>(), e->d_dnode.as<uint16_t>(),
                           e->d_sel.as<uint32_t>(), e->sel_cap_total, e->d_nsel.as<int>(), e->d_ncand.as<int>(), nl,
                           e->ncap, std::max(e->prm.ini_th_fast, 1), 0, qt_dbg(), (int *)nullptr);
    else {
        // the filtering / counting sweep over the stored keys of every level first, spread over the device
        if (e->qt_prefilter)
            L.launch(k_qt_prefilter, dim3(std::min((e->max_slots_level + QT_PRE_KEYS - 1) / QT_PRE_KEYS, QT_PRE_GRID), nl, batch), dim3(256), 0, dg, (int)e->cells.size(), e->d_slots.as<uint32_t>(), e->frame_slots, e->d_cellcnt.as<int>(),
                               e->d_dkey.as<uint32_t>(), e->d_qtaux.as<int>(), std::max(e->prm.ini_th_fast, 1));
        L.launch(k_quadtree<true>, dim3(batch, nl), dim3(w * h >= QT_LARGE_PIXELS ? QT_THREADS : QT_THREADS_SMALL), e->qt_lds + (size_t)e->qt_kcap * 6, dg,
                           e->d_cells.as<CellDesc>(), (int)e->cells.size(), e->d_slots.as<uint32_t>(), e->frame_slots,
                           e->d_cellcnt.as<int>(), e->d_dkey.as<uint32_t>(), e->d_dnode.as<uint16_t>(),
                           e->d_sel.as<uint32_t>(), e->sel_cap_total, e->d_nsel.as<int>(), e->d_ncand.as<int>(), nl,
                           e->ncap, std::max(e->prm.ini_th_fast, 1), e->qt_kcap, qt_dbg(),
                           e->qt_prefilter ? e->d_qtaux.as<int>() : (int *)nullptr);
    }
    if (!rec)
        ORBGPU_HIP_TRY(hipGetLastError());
    e->counters_dirty = false;
    END(ST_QUADTREE, st);
    BEGIN(ST_ORIENT, st);
    const int or_iters = batch >= OR_BATCH_MIN ? OR_ITERS : 1;
    L.launch(k_orient, dim3((e->sel_cap_total + 8 * or_iters - 1) / (8 * or_iters), batch), dim3(256), 0, pyr, e->frame_pyr, dg, nl,
                       e->d_sel.as<uint32_t>(), e->sel_cap_total, e->d_nsel.as<int>(),
                       reinterpret_cast<const uint32_t *>(e->d_pattern.as<uint8_t>() + 1024), d_kps,
                       e->d_aux.as<KpAux>(), cap, d_n_out, or_iters, s0);
    L.launch(k_trig, dim3((std::min(cap, e->max_kp) + 255) / 256, batch), dim3(256), 0,
                       e->d_aux.as<KpAux>(), d_n_out, cap, e->trig);
    END(ST_ORIENT, st);
    BEGIN(ST_DESCRIBE, st);
    if (side_blur)
        ORBGPU_HIP_TRY(hipStreamWaitEvent(st, e->ev_join, 0));  // join: the descriptors read the blurred planes
    L.launch(k_describe, dim3((std::min(cap, e->max_kp) + 7) / 8, batch), dim3(256), 0, blur,
                       e->frame_pyr, dg, e->d_aux.as<KpAux>(), d_n_out, cap, e->d_pattern.as<int8_t>(), d_desc);
    END(ST_DESCRIBE, st);
#undef BEGIN
#undef END
    if (!rec)
        ORBGPU_HIP_TRY(hipGetLastError());
    else if (L.err != hipSuccess) {
        set_error("recording the extraction graph: %s", hipGetErrorString(L.err));
        return ORBGPU_EHIP;
    }
    if (prof)
        e->prof_calls++;
    e->last_batch = batch;
    e->last_cap = cap;
    (void)w;
    (void)h;
    return ORBGPU_OK;
}

} // namespace orbgpu

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int orbgpu_extractor_create(const orbgpu_extractor_params *p, orbgpu_extractor **out)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    ORBGPU_REQUIRE(p && out, "null argument");
    ORBGPU_REQUIRE(p->nlevels >= 1 && p->nlevels <= ORBGPU_MAX_LEVELS, "nlevels must be in [1,%d]", ORBGPU_MAX_LEVELS);
    ORBGPU_REQUIRE(p->nfeatures >= 1, "nfeatures must be positive");
    ORBGPU_REQUIRE(p->scale_factor > 1.0f, "scale_factor must be > 1");
    ORBGPU_REQUIRE(p->ini_th_fast >= 0 && p->ini_th_fast <= 255 && p->min_th_fast >= 0 && p->min_th_fast <= 255,
                   "FAST thresholds must be in [0,255]");
    int rc = select_device(p->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_extractor *e = new (std::nothrow) orbgpu_extractor();
    if (!e) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    e->prm = *p;
    if (e->prm.max_batch < 1)
        e->prm.max_batch = 1;
    e->force_batch_quadtree = getenv("ORBGPU_DEBUG_QT_BATCH") != nullptr;
    e->qt_no_prefilter = getenv("ORBGPU_DEBUG_QT_NOPRE") != nullptr;
    if (const char *q = getenv("ORBGPU_DEBUG_QT_KEYS"))  // test hook, read here like the others (configure() runs at the first extraction)
        e->qt_keys_hook = std::max(atoi(q), 0);
    e->no_direct0 = getenv("ORBGPU_DEBUG_NO_DIRECT0") != nullptr;
    e->no_early_blur0 = getenv("ORBGPU_DEBUG_NO_EARLY_BLUR0") != nullptr;
    if (const char *q = getenv("ORBGPU_DEBUG_DIRECT0_MIN"))
        e->direct0_min_batch = std::max(atoi(q), 1);
    if (const char *q = getenv("ORBGPU_FAST_EARLY_OUT"))  // default of the option for handles created from now on (fuzzing, A/B)
        e->fast_early_out = atoi(q) != 0;
    e->pinned_staging = getenv("ORBGPU_DEBUG_NO_PINNED") == nullptr;
    if (getenv("ORBGPU_DEBUG_NO_GRAPH"))  // test hook / A-B: the host entry points always launch plainly
        e->graph_state = -1;
    if (const char *q = getenv("ORBGPU_DEBUG_FAST_QUEUE"))  // test hook: forces k_fast_detect's queue-full path
        e->fast_queue_cap = std::min(std::max(atoi(q), 0), FD_QCAP);
    e->nlevels = p->nlevels;
    build_tables(e);
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(he));
        delete e;
        return ORBGPU_EHIP;
    }
    rc = trig_table_for_device(p->device_id, &e->trig);
    if (rc != ORBGPU_OK) {
        (void)hipStreamDestroy(e->stream);
        delete e;
        return rc;
    }
    *out = e;
    return ORBGPU_OK;
}

int orbgpu_extractor_destroy(orbgpu_extractor *e)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!e)
        return ORBGPU_OK;
    (void)hipSetDevice(e->prm.device_id);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&e->d_geom, &e->d_cells, &e->d_xtab, &e->d_ytab, &e->d_yrow, &e->d_pattern, &e->d_rstrip, &e->d_rsel,
                      &e->d_rwt, &e->d_ctab, &e->d_bcol, &e->d_pyr,
                      &e->d_blur, &e->d_slots, &e->d_cellcnt, &e->d_dkey, &e->d_dnode, &e->d_sel, &e->d_nsel,
                      &e->d_ncand, &e->d_aux, &e->d_in, &e->d_kps, &e->d_desc, &e->d_nout, &e->d_dbg, &e->d_qtaux};
    for (DevBuf *b : bufs)
        b->release();
    for (auto &x : e->ev)
        if (x)
            (void)hipEventDestroy(x);
    if (e->graph_exec)
        (void)hipGraphExecDestroy(e->graph_exec);
    if (e->graph)
        (void)hipGraphDestroy(e->graph);
    if (e->h_pin)
        (void)hipHostFree(e->h_pin);
    if (e->stream)
        (void)hipStreamDestroy(e->stream);
    if (e->side)
        (void)hipStreamDestroy(e->side);
    if (e->ev_fork)
        (void)hipEventDestroy(e->ev_fork);
    if (e->ev_fork0)
        (void)hipEventDestroy(e->ev_fork0);
    if (e->ev_join)
        (void)hipEventDestroy(e->ev_join);
    delete e;
    return ORBGPU_OK;
}

int orbgpu_extractor_get_levels(const orbgpu_extractor *e, int32_t *n)
{
    ORBGPU_REQUIRE(e && n, "null argument");
    *n = e->nlevels;
    return ORBGPU_OK;
}
int orbgpu_extractor_get_scale_factor(const orbgpu_extractor *e, float *s)
{
    ORBGPU_REQUIRE(e && s, "null argument");
    *s = (float)e->scale_factor_d;
    return ORBGPU_OK;
}
#define ORBGPU_GETTER(name, field, T)                                                                        \
    int name(const orbgpu_extractor *e, T *out)                                                              \
    {                                                                                                        \
        ORBGPU_REQUIRE(e && out, "null argument");                                                           \
        for (int i = 0; i < e->nlevels; i++)                                                                 \
            out[i] = e->field[i];                                                                            \
        return ORBGPU_OK;                                                                                    \
    }
ORBGPU_GETTER(orbgpu_extractor_get_scale_factors, scale, float)
ORBGPU_GETTER(orbgpu_extractor_get_inv_scale_factors, inv_scale, float)
ORBGPU_GETTER(orbgpu_extractor_get_sigma2, sigma2, float)
ORBGPU_GETTER(orbgpu_extractor_get_inv_sigma2, inv_sigma2, float)
ORBGPU_GETTER(orbgpu_extractor_get_quotas, quota, int32_t)
#undef ORBGPU_GETTER

int orbgpu_extractor_max_keypoints(const orbgpu_extractor *e, int32_t w, int32_t h, int32_t *cap)
{
    ORBGPU_REQUIRE(e && cap, "null argument");
    ORBGPU_REQUIRE(w > 0 && h > 0, "empty image");
    int total = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const int lw = cv_round_host((float)w * e->inv_scale[l]), lh = cv_round_host((float)h * e->inv_scale[l]);
        const float width = (float)(lw - 2 * BORDER0), height = (float)(lh - 2 * BORDER0);
        ORBGPU_REQUIRE(width >= 30 && height >= 30, "pyramid level %d too small", l);
        const int n_ini = (int)roundf(width / height);
        total += std::max(4 * n_ini, e->quota[l] + 3);
    }
    *cap = total;
    return ORBGPU_OK;
}

int orbgpu_extract_batch_device(orbgpu_extractor *e, const uint8_t *d_gray, int32_t batch, int32_t w, int32_t h,
                                size_t stride, size_t frame_stride, orbgpu_keypoint *d_kps, uint8_t *d_desc,
                                int32_t cap, int32_t *d_n_out, void *hip_stream)
{
    ORBGPU_REQUIRE(e && d_gray && d_kps && d_desc && d_n_out, "null argument");
    ORBGPU_REQUIRE(batch >= 1 && w > 0 && h > 0 && cap >= 1, "bad batch/size/cap");
    ORBGPU_REQUIRE(stride >= (size_t)w && stride < ((size_t)1 << 24) && frame_stride >= stride * (size_t)(h - 1) + (size_t)w,
                   "bad strides");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    rc = configure(e, w, h, std::max(batch, e->prm.max_batch));
    if (rc != ORBGPU_OK)
        return rc;
    Launcher L((hipStream_t)hip_stream);
    return launch_pipeline(e, d_gray, batch, w, h, stride, frame_stride, d_kps, d_desc, cap, d_n_out, L);
}

int orbgpu_extract_batch(orbgpu_extractor *e, const uint8_t *gray, int32_t batch, int32_t w, int32_t h,
                         size_t stride, size_t frame_stride, orbgpu_keypoint *kps, uint8_t *desc, int32_t cap,
                         int32_t *n_out)
{
    ORBGPU_REQUIRE(e && n_out, "null argument");
    ORBGPU_REQUIRE(batch >= 1, "bad batch");
    if (w <= 0 || h <= 0 || !gray) {  // ORBextractor.cc:1046: empty image -> silent return
        for (int i = 0; i < batch; i++)
            n_out[i] = 0;
        return ORBGPU_OK;
    }
    ORBGPU_REQUIRE(kps && desc && cap >= 1, "null output / bad cap");
    ORBGPU_REQUIRE(stride >= (size_t)w && stride < ((size_t)1 << 24), "bad stride");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const size_t img = (size_t)w * h;
    if ((rc = e->d_in.reserve(img * batch)) != ORBGPU_OK)
        return rc;
    if ((rc = e->d_kps.reserve(sizeof(orbgpu_keypoint) * (size_t)cap * batch)) != ORBGPU_OK)
        return rc;
    if ((rc = e->d_desc.reserve((size_t)32 * cap * batch)) != ORBGPU_OK)
        return rc;
    if ((rc = e->d_nout.reserve(sizeof(int) * batch)) != ORBGPU_OK)
        return rc;
    // Everything that allocates or uploads tables comes first (a first call of a configuration), while the stream is
    // empty; only then is work enqueued.
    if ((rc = configure(e, w, h, std::max(batch, e->prm.max_batch))) != ORBGPU_OK)
        return rc;
    if ((rc = e->d_aux.reserve(sizeof(KpAux) * (size_t)cap * batch)) != ORBGPU_OK)
        return rc;
    uint8_t *pin_in = nullptr;
    int *pin_n = nullptr;
    orbgpu_keypoint *pin_k = nullptr;
    uint8_t *pin_d = nullptr;
    // One frame at a time is what the reference does (C1): its image and results go through a pinned buffer of the
    // handle's own (one memcpy each way + real DMA) rather than through the runtime's pageable path -- 0.226 -> 0.220 ms
    // per 640x480 frame, and a tighter p95.  Large batches keep the pageable path (no hundreds of MB of pinned memory).
    const size_t pin_need = ((img * batch + 255) & ~(size_t)255) + ((sizeof(int) * batch + 255) & ~(size_t)255) +
                            (sizeof(orbgpu_keypoint) + 32) * (size_t)cap * batch;
    if (e->pinned_staging && pin_need <= ((size_t)64 << 20)) {
        const size_t o_n = (img * batch + 255) & ~(size_t)255, o_k = o_n + ((sizeof(int) * batch + 255) & ~(size_t)255);
        const size_t o_d = o_k + sizeof(orbgpu_keypoint) * (size_t)cap * batch, need = o_d + (size_t)32 * cap * batch;
        if (need > e->h_pin_bytes) {
            if (e->h_pin)
                (void)hipHostFree(e->h_pin);
            e->h_pin = nullptr;
            e->h_pin_bytes = 0;
            ORBGPU_HIP_TRY(hipHostMalloc(&e->h_pin, need, hipHostMallocDefault));
            e->h_pin_bytes = need;
        }
        pin_in = static_cast<uint8_t *>(e->h_pin);
        pin_n = reinterpret_cast<int *>(pin_in + o_n);
        pin_k = reinterpret_cast<orbgpu_keypoint *>(pin_in + o_k);
        pin_d = pin_in + o_d;
        for (int f = 0; f < batch; f++)
            for (int y = 0; y < h; y++)
                memcpy(pin_in + img * f + (size_t)y * w, gray + frame_stride * f + stride * (size_t)y, (size_t)w);
        ORBGPU_HIP_TRY(hipMemcpyAsync(e->d_in.p, pin_in, img * batch, hipMemcpyHostToDevice, e->stream));
    } else
    for (int f = 0; f < batch; f++)
        ORBGPU_HIP_TRY(hipMemcpy2DAsync(e->d_in.as<uint8_t>() + img * f, (size_t)w, gray + frame_stride * f, stride,
                                        (size_t)w, (size_t)h, hipMemcpyHostToDevice, e->stream));
    // Single frames are launch bound (17 dependent launches for 0.3 MB): replay them as one hipGraph.  The graph bakes
    // in the handle-owned buffers, so it is keyed on them and on the geometry; a first call of a configuration runs
    // plain, the second captures.  Profiling (events between stages) and capture failures fall
    // back to plain launches.
    uint64_t key = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)w, (uint64_t)h, (uint64_t)batch, (uint64_t)cap, (uint64_t)(uintptr_t)e->d_in.p,
                       (uint64_t)(uintptr_t)e->d_kps.p, (uint64_t)(uintptr_t)e->d_desc.p, (uint64_t)(uintptr_t)e->d_nout.p,
                       (uint64_t)(uintptr_t)e->d_pyr.p, (uint64_t)(uintptr_t)e->d_aux.p, (uint64_t)(uintptr_t)e->d_sel.p})
        key = (key ^ v) * 1099511628211ull;
    bool launched = false;
    if (!e->profiling && e->graph_state >= 0) {
        if (e->graph_exec && e->graph_key == key) {
            if (hipGraphLaunch(e->graph_exec, e->stream) == hipSuccess) {
                launched = true;
                e->level0_materialized = !e->last_src.direct;  // the replay did what launch_pipeline recorded: no padded level 0 in direct mode
            } else
                e->graph_state = -1;
        } else if (e->graph_key == key) {  // second call of this configuration: record the graph
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            bool ok = hipGraphCreate(&g, 0) == hipSuccess;
            if (ok) {
                Launcher L;
                L.graph = g;
                const int lrc = launch_pipeline(e, e->d_in.as<uint8_t>(), batch, w, h, (size_t)w, img,
                                                e->d_kps.as<orbgpu_keypoint>(), e->d_desc.as<uint8_t>(), cap,
                                                e->d_nout.as<int32_t>(), L);
                ok = lrc == ORBGPU_OK && L.err == hipSuccess;
            }
            if (ok)
                ok = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
            if (ok) {
                if (e->graph_exec)
                    (void)hipGraphExecDestroy(e->graph_exec);
                if (e->graph)
                    (void)hipGraphDestroy(e->graph);
                e->graph = g;
                e->graph_exec = ge;
                e->graph_state = 1;
                ok = hipGraphLaunch(e->graph_exec, e->stream) == hipSuccess;
                launched = ok;
                e->level0_materialized = !e->last_src.direct;
            }
            if (!ok) {
                (void)hipGetLastError();
                if (ge && ge != e->graph_exec)
                    (void)hipGraphExecDestroy(ge);
                if (g && g != e->graph)
                    (void)hipGraphDestroy(g);
                e->graph_state = -1;
            }
        } else {
            e->graph_key = key;  // first call of a new configuration: plain launches, the graph next time
            if (e->graph_exec) {
                (void)hipGraphExecDestroy(e->graph_exec);
                e->graph_exec = nullptr;
            }
        }
    }
    if (!launched) {
        rc = orbgpu_extract_batch_device(e, e->d_in.as<uint8_t>(), batch, w, h, (size_t)w, img,
                                         e->d_kps.as<orbgpu_keypoint>(), e->d_desc.as<uint8_t>(), cap,
                                         e->d_nout.as<int32_t>(), e->stream);
        if (rc != ORBGPU_OK)
            return rc;
    }
    ORBGPU_HIP_TRY(hipMemcpyAsync(pin_n ? pin_n : n_out, e->d_nout.p, sizeof(int) * batch, hipMemcpyDeviceToHost, e->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(e->stream));
    if (pin_n)
        memcpy(n_out, pin_n, sizeof(int) * batch);
    for (int f = 0; f < batch; f++) {
        if (n_out[f] < 0) {
            set_error("frame %d needs %d key points but cap is %d", f, -1 - n_out[f], cap);
            return ORBGPU_ECAPACITY;
        }
    }
    for (int f = 0; f < batch; f++) {
        if (n_out[f] == 0)
            continue;
        ORBGPU_HIP_TRY(hipMemcpyAsync((pin_k ? pin_k : kps) + (size_t)cap * f, e->d_kps.as<orbgpu_keypoint>() + (size_t)cap * f,
                                      sizeof(orbgpu_keypoint) * n_out[f], hipMemcpyDeviceToHost, e->stream));
        ORBGPU_HIP_TRY(hipMemcpyAsync((pin_d ? pin_d : desc) + (size_t)32 * cap * f, e->d_desc.as<uint8_t>() + (size_t)32 * cap * f,
                                      (size_t)32 * n_out[f], hipMemcpyDeviceToHost, e->stream));
    }
    ORBGPU_HIP_TRY(hipStreamSynchronize(e->stream));
    if (pin_k)
        for (int f = 0; f < batch; f++) {
            memcpy(kps + (size_t)cap * f, pin_k + (size_t)cap * f, sizeof(orbgpu_keypoint) * n_out[f]);
            memcpy(desc + (size_t)32 * cap * f, pin_d + (size_t)32 * cap * f, (size_t)32 * n_out[f]);
        }
    return ORBGPU_OK;
}

int orbgpu_extract(orbgpu_extractor *e, const uint8_t *gray, int32_t w, int32_t h, size_t stride,
                   orbgpu_keypoint *kps, uint8_t *desc, int32_t cap, int32_t *n_out)
{
    return orbgpu_extract_batch(e, gray, 1, w, h, stride, stride * (size_t)(h > 0 ? h : 0), kps, desc, cap, n_out);
}

// Direct mode leaves level 0 of the padded pyramid unwritten (no stage reads it); whoever asks for it gets it now, from the
// image of the last call -- the handle's own staging copy for the host entry points, the caller's device buffer for
// orbgpu_extract_batch_device (which must still hold the images: orbgpu.h).
static int ensure_level0(orbgpu_extractor *e)
{
    if (e->level0_materialized)
        return ORBGPU_OK;
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    Launcher L0(nullptr);
    int rc = materialize_level0(e, e->last_src, e->last_batch, L0);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    e->level0_materialized = true;
    return ORBGPU_OK;
}

int orbgpu_extractor_get_pyramid_level(orbgpu_extractor *e, int32_t frame, int32_t level, uint8_t *dst,
                                       size_t dst_stride, int32_t *width, int32_t *height)
{
    ORBGPU_REQUIRE(e && dst, "null argument");
    ORBGPU_REQUIRE(e->last_batch > 0 && frame >= 0 && frame < e->last_batch && level >= 0 && level < e->nlevels,
                   "no such frame/level in the last call");
    const LevelGeom &g = e->geom[level];
    ORBGPU_REQUIRE(dst_stride >= (size_t)g.w, "bad dst_stride");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (level == 0 && (rc = ensure_level0(e)) != ORBGPU_OK)
        return rc;
    const uint8_t *src = e->d_pyr.as<uint8_t>() + e->frame_pyr * frame + g.plane_off + (size_t)EDGE * g.pitch + EDGE;
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    ORBGPU_HIP_TRY(hipMemcpy2D(dst, dst_stride, src, (size_t)g.pitch, (size_t)g.w, (size_t)g.h, hipMemcpyDeviceToHost));
    if (width)
        *width = g.w;
    if (height)
        *height = g.h;
    return ORBGPU_OK;
}

int orbgpu_extractor_debug_read(orbgpu_extractor *e, int32_t what, int32_t frame, int32_t level, void *dst,
                                size_t dst_bytes, size_t *n, int32_t *aux)
{
    ORBGPU_REQUIRE(e && dst && n, "null argument");
    ORBGPU_REQUIRE(e->last_batch > 0 && frame >= 0 && frame < e->last_batch && level >= 0 && level < e->nlevels,
                   "no such frame/level in the last call");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    const LevelGeom &g = e->geom[level];
    if (what == ORBGPU_DBG_PYRAMID_PADDED || what == ORBGPU_DBG_BLURRED_PADDED) {
        const size_t bytes = (size_t)g.pitch * (g.h + 2 * EDGE);
        ORBGPU_REQUIRE(dst_bytes >= bytes, "dst too small (%zu needed)", bytes);
        if (what == ORBGPU_DBG_PYRAMID_PADDED && level == 0 && (rc = ensure_level0(e)) != ORBGPU_OK)
            return rc;
        const uint8_t *base = (what == ORBGPU_DBG_PYRAMID_PADDED ? e->d_pyr : e->d_blur).as<uint8_t>();
        ORBGPU_HIP_TRY(hipMemcpy(dst, base + e->frame_pyr * frame + g.plane_off, bytes, hipMemcpyDeviceToHost));
        if (what == ORBGPU_DBG_BLURRED_PADDED && level == 0 && e->last_src.direct) {
            // direct mode keeps the blurred level 0 with pixel 0 at column BLUR0_OX: present it in the common layout
            uint8_t *rowp = static_cast<uint8_t *>(dst);
            for (int r = 0; r < g.h + 2 * EDGE; r++, rowp += g.pitch)
                memmove(rowp, rowp + (BLUR0_OX - EDGE), (size_t)g.pitch - (BLUR0_OX - EDGE));
        }
        *n = bytes;
        if (aux)
            *aux = g.pitch;
        return ORBGPU_OK;
    }
    if (what == ORBGPU_DBG_CANDIDATES || what == ORBGPU_DBG_SELECTED) {
        int cnt = 0;
        const int *cp = (what == ORBGPU_DBG_CANDIDATES ? e->d_ncand : e->d_nsel).as<int>() + (size_t)frame * e->nlevels + level;
        ORBGPU_HIP_TRY(hipMemcpy(&cnt, cp, sizeof(int), hipMemcpyDeviceToHost));
        ORBGPU_REQUIRE(cnt >= 0, "stage reported failure (%d)", cnt);
        ORBGPU_REQUIRE(dst_bytes >= (size_t)cnt * 12, "dst too small (%zu needed)", (size_t)cnt * 12);
        const uint32_t *keys = what == ORBGPU_DBG_CANDIDATES
                                   ? e->d_dkey.as<uint32_t>() + e->frame_slots * frame + g.slot_off
                                   : e->d_sel.as<uint32_t>() + (size_t)e->sel_cap_total * frame + g.sel_off;
        if (cnt > 0) {
            if ((rc = e->d_dbg.reserve((size_t)cnt * 12)) != ORBGPU_OK)
                return rc;
            hipLaunchKernelGGL(k_unpack_keys, dim3((cnt + 255) / 256), dim3(256), 0, 0, keys, cnt, e->d_dbg.as<int>());
            ORBGPU_HIP_TRY(hipMemcpy(dst, e->d_dbg.p, (size_t)cnt * 12, hipMemcpyDeviceToHost));
            if (what == ORBGPU_DBG_CANDIDATES) {
                // the device keeps a level's candidates in no particular order; report them in vToDistributeKeys
                // order (cells row-major, row-major inside a cell: ORBextractor.cc:789-829)
                struct K {
                    int x, y, r;
                };
                K *k = static_cast<K *>(dst);
                std::sort(k, k + cnt, [&](const K &a, const K &b) {
                    const int ai = (a.y - 3) / g.hcell, bi = (b.y - 3) / g.hcell;
                    const int aj = (a.x - 3) / g.wcell, bj = (b.x - 3) / g.wcell;
                    if (ai != bi)
                        return ai < bi;
                    if (aj != bj)
                        return aj < bj;
                    return a.y != b.y ? a.y < b.y : a.x < b.x;
                });
            }
        }
        *n = (size_t)cnt;
        if (aux)
            *aux = 0;
        return ORBGPU_OK;
    }
    set_error("unknown debug selector %d", what);
    return ORBGPU_EINVAL;
}

int orbgpu_extractor_set_stage_signal(orbgpu_extractor *e, int32_t stage, void *hip_event)
{
    ORBGPU_REQUIRE(e && stage >= 0 && stage < ST_COUNT, "bad stage");
    e->stage_signal[stage] = (hipEvent_t)hip_event;
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------------------------
// orbgpu_pipeline: a batch as `parts` sub-batches on `parts` streams, each with an extractor handle of its own.  Part k
// starts when part k-1 has passed the pyramid stage (part 0: the last part of the previous call), so the stages of
// neighbouring parts -- HBM-bound pyramid / blur / gathers, VALU-bound FAST, latency-bound quadtree -- run next to each
// other instead of one after the other, across calls as well: a call only enqueues, and hands back an event.
// ---------------------------------------------------------------------------------------------
struct orbgpu_pipeline {
    std::vector<orbgpu_extractor *> part;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ev_stage, ev_done;
    int device_id = 0;
    int last_part = -1;  // last part the previous call used (its ev_stage holds a record)
};

int orbgpu_pipeline_destroy(orbgpu_pipeline *pl)
{
    if (!pl)
        return ORBGPU_OK;
    select_device(pl->device_id);
    for (auto st : pl->stream)
        if (st)
            (void)hipStreamSynchronize(st);
    for (auto e : pl->part)
        orbgpu_extractor_destroy(e);
    for (auto ev : pl->ev_stage)
        if (ev)
            (void)hipEventDestroy(ev);
    for (auto ev : pl->ev_done)
        if (ev)
            (void)hipEventDestroy(ev);
    for (auto st : pl->stream)
        if (st)
            (void)hipStreamDestroy(st);
    delete pl;
    return ORBGPU_OK;
}

int orbgpu_pipeline_create(const orbgpu_extractor_params *p, int32_t parts, orbgpu_pipeline **out)
{
    ORBGPU_REQUIRE(p && out, "null argument");
    ORBGPU_REQUIRE(parts >= 1 && parts <= 16, "parts must be in [1,16]");
    int rc = select_device(p->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_pipeline *pl = new (std::nothrow) orbgpu_pipeline();
    if (!pl) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    pl->device_id = p->device_id;
    orbgpu_extractor_params pp = *p;
    pp.max_batch = (std::max(p->max_batch, 1) + parts - 1) / parts;
    for (int k = 0; k < parts; k++) {
        orbgpu_extractor *e = nullptr;
        if ((rc = orbgpu_extractor_create(&pp, &e)) != ORBGPU_OK) {
            orbgpu_pipeline_destroy(pl);
            return rc;
        }
        pl->part.push_back(e);
        hipStream_t st = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        const bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                        hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess;
        pl->stream.push_back(st);
        pl->ev_stage.push_back(a);
        pl->ev_done.push_back(b);
        if (!ok) {
            orbgpu_pipeline_destroy(pl);
            set_error("orbgpu_pipeline_create: stream / event creation failed");
            return ORBGPU_EHIP;
        }
        e->pipe_signal[ST_PYRAMID] = a;
    }
    *out = pl;
    return ORBGPU_OK;
}

int orbgpu_pipeline_parts(const orbgpu_pipeline *pl, int32_t *parts)
{
    ORBGPU_REQUIRE(pl && parts, "null argument");
    *parts = (int32_t)pl->part.size();
    return ORBGPU_OK;
}

int orbgpu_pipeline_part(orbgpu_pipeline *pl, int32_t k, orbgpu_extractor **part)
{
    ORBGPU_REQUIRE(pl && part && k >= 0 && k < (int)pl->part.size(), "bad part index");
    *part = pl->part[k];
    return ORBGPU_OK;
}

int orbgpu_pipeline_extract_device(orbgpu_pipeline *pl, const uint8_t *d_gray, int32_t batch, int32_t w, int32_t h,
                                   size_t stride, size_t frame_stride, orbgpu_keypoint *d_kps, uint8_t *d_desc,
                                   int32_t cap, int32_t *d_n_out, void *wait_event, void *done_event)
{
    ORBGPU_REQUIRE(pl && d_gray && d_kps && d_desc && d_n_out, "null argument");
    ORBGPU_REQUIRE(batch >= 1 && cap >= 1, "bad batch/cap");
    int rc = select_device(pl->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const int P = (int)pl->part.size();
    const int n = (batch + P - 1) / P;
    int used = 0;
    for (int k = 0; k < P && k * n < batch; k++, used++) {
        const int f0 = k * n, cnt = std::min(n, batch - f0);
        hipStream_t st = pl->stream[k];
        if (wait_event)
            ORBGPU_HIP_TRY(hipStreamWaitEvent(st, (hipEvent_t)wait_event, 0));
        const int prev = k > 0 ? k - 1 : pl->last_part;
        if (prev >= 0 && prev != k)
            ORBGPU_HIP_TRY(hipStreamWaitEvent(st, pl->ev_stage[prev], 0));
        rc = orbgpu_extract_batch_device(pl->part[k], d_gray + (size_t)f0 * frame_stride, cnt, w, h, stride, frame_stride,
                                         d_kps + (size_t)f0 * cap, d_desc + (size_t)f0 * cap * 32, cap, d_n_out + f0, st);
        if (rc != ORBGPU_OK)
            return rc;
        ORBGPU_HIP_TRY(hipEventRecord(pl->ev_done[k], st));
    }
    pl->last_part = used - 1;
    if (done_event) {
        // on the last part's stream once it has seen the others finish (a stream of its own for the join cost the step
        // one more cross-stream hop: 2.27 against 2.13 ms with a single part)
        for (int k = 0; k + 1 < used; k++)
            ORBGPU_HIP_TRY(hipStreamWaitEvent(pl->stream[used - 1], pl->ev_done[k], 0));
        ORBGPU_HIP_TRY(hipEventRecord((hipEvent_t)done_event, pl->stream[used - 1]));
    }
    return ORBGPU_OK;
}

int orbgpu_pipeline_wait(orbgpu_pipeline *pl, void *hip_stream)
{
    ORBGPU_REQUIRE(pl, "null argument");
    int rc = select_device(pl->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    for (size_t k = 0; k < pl->part.size(); k++)  // an event never recorded is no wait
        ORBGPU_HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, pl->ev_done[k], 0));
    return ORBGPU_OK;
}

int orbgpu_extractor_set_profiling(orbgpu_extractor *e, int32_t enable)
{
    ORBGPU_REQUIRE(e, "null argument");
    e->profiling = enable != 0;  // the averaging window is kept: stage_times() reads and resets it
    return ORBGPU_OK;
}
int orbgpu_extractor_set_fast_early_out(orbgpu_extractor *e, int32_t enable)
{
    ORBGPU_REQUIRE(e, "null handle");
    if (e->fast_early_out != (enable != 0)) {
        e->fast_early_out = enable != 0;
        e->graph_key = 0;  // a captured launch sequence holds the other kernel
    }
    return ORBGPU_OK;
}

int orbgpu_extractor_set_concurrent_blur(orbgpu_extractor *e, int32_t enable)
{
    ORBGPU_REQUIRE(e, "null argument");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (enable && !e->side) {
        ORBGPU_HIP_TRY(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
        ORBGPU_HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        ORBGPU_HIP_TRY(hipEventCreateWithFlags(&e->ev_fork0, hipEventDisableTiming));
        ORBGPU_HIP_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    if ((enable != 0) != e->concurrent_blur) {
        ORBGPU_HIP_TRY(hipDeviceSynchronize());  // nothing of the other schedule in flight; a captured graph bakes the schedule in
        e->graph_key = 0;
        e->prof_calls = 0;
    }
    e->concurrent_blur = enable != 0;
    return ORBGPU_OK;
}
int orbgpu_extractor_graph_state(const orbgpu_extractor *e, int32_t *state)
{
    ORBGPU_REQUIRE(e && state, "null argument");
    *state = e->graph_state;
    return ORBGPU_OK;
}
int orbgpu_extractor_debug_quadtree_config(const orbgpu_extractor *e, int32_t batch, int32_t *lds_keys, int32_t *threads,
                                           int32_t *lds_bytes)
{
    ORBGPU_REQUIRE(e && lds_keys && threads && lds_bytes, "null argument");
    ORBGPU_REQUIRE(e->cfg_w > 0, "no image size configured yet (call an extraction first)");
    const bool batch_variant = batch >= QT_BATCH_MIN || e->force_batch_quadtree;
    *lds_keys = batch_variant ? 0 : e->qt_kcap;
    *threads = batch_variant ? QT_THREADS_BATCH : (e->cfg_w * e->cfg_h >= QT_LARGE_PIXELS ? QT_THREADS : QT_THREADS_SMALL);
    *lds_bytes = (int32_t)(e->qt_lds + (batch_variant ? 0 : (size_t)e->qt_kcap * 6));
    return ORBGPU_OK;
}
int orbgpu_extractor_stage_count(void) { return ST_COUNT; }
const char *orbgpu_extractor_stage_name(int32_t i) { return (i >= 0 && i < ST_COUNT) ? k_stage_names[i] : ""; }
int orbgpu_extractor_stage_times(orbgpu_extractor *e, float *ms)
{
    ORBGPU_REQUIRE(e && ms, "null argument");
    ORBGPU_REQUIRE(e->prof_calls > 0, "no profiled call recorded");
    int rc = select_device(e->prm.device_id);
    if (rc != ORBGPU_OK)
        return rc;
    for (int i = 0; i < ST_COUNT; i++)
        ms[i] = 0.f;
    for (int c = 0; c < e->prof_calls; c++) {
        hipEvent_t *evs = &e->ev[(size_t)c * 2 * ST_COUNT];
        ORBGPU_HIP_TRY(hipEventSynchronize(evs[ST_COUNT]));  // the last boundary event of the profiled call
        for (int i = 0; i < ST_COUNT; i++) {
            float t = 0.f;
            if (i == ST_BLUR && e->concurrent_blur) {  // timed on the side stream it ran on
                ORBGPU_HIP_TRY(hipEventSynchronize(evs[ST_COUNT + 2]));
                ORBGPU_HIP_TRY(hipEventElapsedTime(&t, evs[ST_COUNT + 1], evs[ST_COUNT + 2]));
            } else
                ORBGPU_HIP_TRY(hipEventElapsedTime(&t, evs[i], evs[i + 1]));
            ms[i] += t;
        }
    }
    for (int i = 0; i < ST_COUNT; i++)
        ms[i] /= (float)e->prof_calls;
    e->prof_calls = 0;  // start a new averaging window
    return ORBGPU_OK;
}

} // extern "C"
