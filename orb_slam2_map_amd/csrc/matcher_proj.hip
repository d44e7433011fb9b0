// Grid-accelerated projection matchers for MI355X (gfx950).
//
// Replaces ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
// (reference src/ORBmatcher.cc:45-129) and ORBmatcher::SearchByProjection(CurrentFrame, LastFrame,
// th, bMono) (:1328-1470), including Frame::GetFeaturesInArea (src/Frame.cc:327-380).
//
// Both reference functions are greedy and sequential: a key point claimed by an earlier row
// (map point) whose MapPoint::Observations()>0 is skipped by every later row (:87-89, :1403-1405).
// As in the brute-force matcher the GPU evaluates all rows in parallel and iterates to the unique
// triangular fixpoint ("row i sees the claims of rows < i"), which equals the sequential result.
// Pass 1 (k_proj_lists): one wave per row walks the grid window ONCE -- lanes take the window's cells
// (ix outer, iy inner: the reference's candidate order, which decides ties) -- and caches the 16 best
// candidates in preference order (distance, visiting order).  Pass 2 (k_proj_resolve): one workgroup runs
// the sweeps from the cached lists (first four words of a row in registers) with the claim tables in LDS;
// only a row whose 16 cached candidates cannot decide is walked again with the claim filter.
#include "common.h"
#include "matcher_common.h"
#include "workspace.h"
#include "proj_internal.h"

#include <algorithm>
#include <climits>
#include <new>
#include <cstring>
#include <cmath>
#include <vector>

namespace orbgpu {

constexpr int GC = ORBGPU_GRID_COLS, GR = ORBGPU_GRID_ROWS;

struct Query {  // one row of the matcher: a projected map point
    float x, y, r;      // window centre and half-size (r already multiplied by the level scale)
    float ur;           // predicted right coordinate (mTrackProjXR / u - mbf*invz)
    int min_level, max_level;
    int active;         // 0: the reference `continue`s before the candidate loop
    int blocking;       // a claim by this row hides the key point from later rows
    int check_ur;       // apply the mvuRight gate (ORBmatcher.cc:91-96 / 1407-1413); off for :1472-1599
    int gate;           // 1: Fuse's reprojection-error gates (ORBmatcher.cc:908-933) against F.inv_sigma2
};

struct FrameDev {
    int n;             // key point count, or the array capacity when n_dev is set
    const int *n_dev;  // device-resident count (frames straight out of the extractor); nullptr: use n
    int kp_stride;     // element stride of kp_x / kp_y / kp_octave: 1 (SoA upload) or 7 (orbgpu_keypoint records)
    const float *kp_x, *kp_y;
    const int *kp_octave;
    const float *u_right;
    const uint8_t *desc;
    float min_x, min_y, inv_w, inv_h;
    const int *cell_start, *cell_items;
    const float *inv_sigma2;  // [nlevels] mvInvLevelSigma2, only read when a query has gate set
};

constexpr uint64_t KEY_NONE = ((uint64_t)256 << 44) | 0xFFFFFFFFFFFull;

__device__ __forceinline__ void keep2(uint64_t &k1, uint64_t &k2, uint64_t k)
{
    if (k < k1) {
        k2 = k1;
        k1 = k;
    } else if (k < k2) {
        k2 = k;
    }
}

__device__ __forceinline__ void cswap64(uint64_t &a, uint64_t &b)
{
    const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}
__device__ __forceinline__ void top4_insert64(uint64_t t[4], uint64_t k)
{
    if (k < t[3]) {
        t[3] = k;
        cswap64(t[2], t[3]);
        cswap64(t[1], t[2]);
        cswap64(t[0], t[1]);
    }
}

// Walks the grid window of one row (Frame::GetFeaturesInArea, Frame.cc:327-380, and the candidate
// filters of ORBmatcher.cc:82-97 / 1399-1412) with one wave: lanes take the window's cells in the
// reference's visiting order (ix outer, iy inner), and every surviving candidate becomes a key
//   distance << 44 | cell sequence << 32 | position in cell << 20 | key point index,
// whose ascending order is exactly the order in which the sequential loop would prefer candidates.
// `claim` == nullptr: no claim filtering (pass 1); otherwise key points with claim[idx] < i are hidden.
// sink(key, octave) is called by the lane that found the candidate.
template <int LPR = 64, class Sink>
__device__ __forceinline__ void proj_walk_each(const Query &Q, const uint64_t a[4], const FrameDev &F,
                                               const int *claim, int i, Sink &&sink)
{
    const int lane = threadIdx.x & (LPR - 1);  // LPR lanes share a row (a whole wave, or 16 lanes in pass 1)
    const int c0 = (int)floorf((Q.x - F.min_x - Q.r) * F.inv_w);
    const int c1 = (int)ceilf((Q.x - F.min_x + Q.r) * F.inv_w);
    const int r0 = (int)floorf((Q.y - F.min_y - Q.r) * F.inv_h);
    const int r1 = (int)ceilf((Q.y - F.min_y + Q.r) * F.inv_h);
    const int minCx = max(0, c0), maxCx = min(GC - 1, c1);
    const int minCy = max(0, r0), maxCy = min(GR - 1, r1);
    if (minCx >= GC || maxCx < 0 || minCy >= GR || maxCy < 0)
        return;
    const int ny = maxCy - minCy + 1;
    const int ncell = (maxCx - minCx + 1) * ny;
    const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
    for (int seq = lane; seq < ncell; seq += LPR) {
        const int ix = minCx + seq / ny, iy = minCy + seq % ny;
        const int cell = ix * GR + iy;
        const int beg = F.cell_start[cell], end = F.cell_start[cell + 1];
        for (int p = beg; p < end; p++) {
            const int idx = F.cell_items[p];
            const int oct = F.kp_octave[idx * F.kp_stride];
            if (check_levels) {
                if (oct < Q.min_level)
                    continue;
                if (Q.max_level >= 0 && oct > Q.max_level)
                    continue;
            }
            const float dx = F.kp_x[idx * F.kp_stride] - Q.x, dy = F.kp_y[idx * F.kp_stride] - Q.y;
            if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r))
                continue;
            if (claim && claim[idx] < i)
                continue;  // held by an earlier row / a pre-existing association
            const float ur = F.u_right[idx];
            if (Q.check_ur && ur > 0) {
                const float er = fabsf(Q.ur - ur);
                if (er > Q.r)
                    continue;
            }
            if (Q.gate) {  // ORBmatcher.cc:908-933: chi-square gates on the reprojection error (float e2, double bound)
                const float ex = -dx, ey = -dy;  // u - kpx, v - kpy
                if (ur >= 0) {
                    const float er = Q.ur - ur;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if ((double)(e2 * F.inv_sigma2[oct]) > 7.8)
                        continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if ((double)(e2 * F.inv_sigma2[oct]) > 5.99)
                        continue;
                }
            }
            const uint64_t *db = reinterpret_cast<const uint64_t *>(F.desc) + (size_t)idx * 4;
            uint64_t b[4] = {db[0], db[1], db[2], db[3]};
            const uint64_t d = (uint64_t)hamming256(a, b);
            sink((d << 44) | ((uint64_t)seq << 32) | ((uint64_t)(p - beg) << 20) | (uint64_t)idx, oct);
        }
    }
}

// the 4 best keys per lane (the claim-filtered re-walk of pass 2); the caller merges across lanes
__device__ __forceinline__ void proj_walk(const Query &Q, const uint64_t a[4], const FrameDev &F, const int *claim,
                                          int i, uint64_t t[4])
{
    t[0] = t[1] = t[2] = t[3] = KEY_NONE;
    proj_walk_each(Q, a, F, claim, i, [&](uint64_t key, int) { top4_insert64(t, key); });
}

// wave merge of per-lane sorted 4-lists -> every lane holds the 4 smallest keys of the wave
__device__ __forceinline__ void top4_wave_merge64(uint64_t t[4])
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint64_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = __shfl_xor(t[k], off, 64);
        uint64_t c0 = t[0] < o[3] ? t[0] : o[3], c1 = t[1] < o[2] ? t[1] : o[2];
        uint64_t c2 = t[2] < o[1] ? t[2] : o[1], c3 = t[3] < o[0] ? t[3] : o[0];
        cswap64(c0, c2);
        cswap64(c1, c3);
        cswap64(c0, c1);
        cswap64(c2, c3);
        cswap64(c1, c2);
        t[0] = c0;
        t[1] = c1;
        t[2] = c2;
        t[3] = c3;
    }
}

// Candidate lists as the sweeps see them: one 32-bit word per candidate,
//   key point index (14 bits) | distance (9 bits) << 14 | octave (4 bits) << 23,
// in the preference order of the 64-bit keys, PJ_LIST words per row, PJ_NONE after the last candidate (a list
// that contains PJ_NONE is complete: it holds every candidate of the window).
constexpr uint32_t PJ_NONE = 0xFFFFFFFFu;
constexpr uint32_t PJ_REWALK = 0xFFFFFFFEu;  // word 0: the window held more candidates than a wave can rank
constexpr int PJ_LIST = 16;                  // cached candidates per row (4 stay in registers in pass 2)
constexpr int PJ_WBUF = 64;                  // candidates a wave ranks in LDS
__device__ __forceinline__ int pj_idx(uint32_t w) { return (int)(w & 0x3FFFu); }
__device__ __forceinline__ int pj_dist(uint32_t w) { return w == PJ_NONE ? 256 : (int)((w >> 14) & 0x1FFu); }
__device__ __forceinline__ int pj_oct(uint32_t w) { return (int)((w >> 23) & 0xFu); }

// Pass 1: one wave per row walks the window once (no claim filter), drops every candidate into LDS, ranks them
// by key and writes the PJ_LIST best as 32-bit words.
// A window holds a handful of grid cells with about one key point each, so 16 lanes per row (16 rows per
// workgroup) keep the lanes busy; a whole wave per row left 55 of 64 lanes idle (C3: 14 -> see DESIGN.md).
constexpr int PJ_LPR = 16;
constexpr int PJ_ROWS_PER_BLOCK = 256 / PJ_LPR;

__device__ __forceinline__ void proj_lists_body(int m, const Query *__restrict__ q,
                                                const uint8_t *__restrict__ row_desc, const FrameDev &F,
                                                uint32_t *__restrict__ lists)
{
    __shared__ uint64_t s_key[PJ_ROWS_PER_BLOCK][PJ_WBUF];
    __shared__ uint32_t s_word[PJ_ROWS_PER_BLOCK][PJ_WBUF];
    __shared__ int s_cnt[PJ_ROWS_PER_BLOCK];
    const int lane = threadIdx.x & (PJ_LPR - 1), w = threadIdx.x / PJ_LPR;
    const int i = blockIdx.x * PJ_ROWS_PER_BLOCK + w;
    if (lane == 0)
        s_cnt[w] = 0;
    __syncthreads();
    const bool live = i < m;
    if (live) {
        const Query Q = q[i];
        if (Q.active) {
            const uint64_t *da = reinterpret_cast<const uint64_t *>(row_desc) + (size_t)i * 4;
            const uint64_t a[4] = {da[0], da[1], da[2], da[3]};
            proj_walk_each<PJ_LPR>(Q, a, F, nullptr, i, [&](uint64_t key, int oct) {
                const int slot = atomicAdd(&s_cnt[w], 1);
                if (slot < PJ_WBUF) {
                    s_key[w][slot] = key;
                    s_word[w][slot] = (uint32_t)(key & 0x3FFFu) | ((uint32_t)(key >> 44) << 14) | ((uint32_t)(oct & 0xF) << 23);
                }
            });
        }
    }
    __syncthreads();
    if (!live)
        return;
    uint32_t *out = lists + (size_t)i * PJ_LIST;
    const int total = s_cnt[w];
    if (total > PJ_WBUF) {
        for (int c = lane; c < PJ_LIST; c += PJ_LPR)
            out[c] = PJ_REWALK;
        return;
    }
    for (int c = lane; c < total; c += PJ_LPR) {
        const uint64_t key = s_key[w][c];
        int rank = 0;
        for (int j = 0; j < total; j++)
            rank += s_key[w][j] < key ? 1 : 0;  // keys are unique (they carry the key point index)
        if (rank < PJ_LIST)
            out[rank] = s_word[w][c];
    }
    for (int c = lane; c < PJ_LIST; c += PJ_LPR)
        if (c >= total)
            out[c] = PJ_NONE;
}

__global__ __launch_bounds__(256) void k_proj_lists(int m, const Query *__restrict__ q,
                                                    const uint8_t *__restrict__ row_desc, FrameDev F,
                                                    uint32_t *__restrict__ lists)
{
    proj_lists_body(m, q, row_desc, F, lists);
}

// accept rule.  mode 0: ORBmatcher.cc:114-125 (TH_HIGH, ratio test only on equal levels);
//               mode 1: ORBmatcher.cc:1424-1430 (TH_HIGH only) and :1554 (ORBdist only)
template <int MODE>
__device__ __forceinline__ int proj_accept(uint64_t k1, uint64_t k2, const FrameDev &F, float nnratio, int th_dist)
{
    const int bestDist = (int)(k1 >> 44);
    if (bestDist > th_dist)
        return -1;
    const int bestIdx = (int)(k1 & 0xFFFFF);
    if (MODE == 0) {
        const int bestDist2 = (int)(k2 >> 44);
        const int bestLevel = F.kp_octave[bestIdx * F.kp_stride];
        const int bestLevel2 = bestDist2 < 256 ? F.kp_octave[(int)(k2 & 0xFFFFF) * F.kp_stride] : -1;
        if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
            return -1;
    }
    return bestIdx;
}

// ---- pass 2 decision from the cached list ----------------------------------------------------------------
constexpr int PJ_RC = 8;  // rows per thread whose first four words stay in registers (8 * 1024 = 8192 rows)

struct PjScan {
    uint32_t k1 = PJ_NONE, k2 = PJ_NONE;  // best / second visible candidate so far
    int found = 0;
    bool complete = false;  // the end of the list was reached: nothing else is in the window
};

// continue the scan over four more words under the claim table of the previous sweep
__device__ __forceinline__ void pj_scan4(const uint32_t w[4], int i, const int *claimA, PjScan &s)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (s.complete || s.found >= 2)
            continue;
        if (w[k] == PJ_NONE) {
            s.complete = true;
        } else if (!(claimA[pj_idx(w[k])] < i)) {
            if (s.found == 0)
                s.k1 = w[k];
            else
                s.k2 = w[k];
            s.found++;
        }
    }
}

// decision after scanning the list up to (and including) a word of distance d_last (only meaningful when the
// scan is not complete).  mode 0: ORBmatcher.cc:114-125 (TH_HIGH, ratio test only on equal levels);
// mode 1: :1424-1430 / :1554 (threshold only).
template <int MODE>
__device__ __forceinline__ int pj_finish(const PjScan &s, int d_last, float nnratio, int th_dist, bool &decided)
{
    const int need = MODE == 0 ? 2 : 1;
    const int b1 = pj_dist(s.k1);
    decided = s.found >= need || s.complete || (s.found == 1 && b1 > th_dist);
    int result = -1;
    if (decided && s.found > 0 && b1 <= th_dist) {
        result = pj_idx(s.k1);
        if (MODE == 0) {
            const int b2 = pj_dist(s.k2);
            const int l2 = b2 < 256 ? pj_oct(s.k2) : -1;
            if (pj_oct(s.k1) == l2 && (float)b1 > nnratio * (float)b2)
                result = -1;
        }
    }
    if (MODE == 0 && !decided && s.found == 1) {
        // every candidate beyond the scanned part of the list is at least d_last away, so the unknown second
        // distance is >= d_last: a best that is not rejected against d_last is not rejected against it either,
        // whatever its level (:114-118)
        if (!((float)b1 > nnratio * (float)d_last)) {
            decided = true;
            result = pj_idx(s.k1);
        }
    }
    return result;
}

// The words beyond the first four, fetched only when those cannot decide (kept out of line: it is the rare path
// and would otherwise be replicated for every register-resident row).
// decision of a row from its 16 list words (registers) under a claim table: four words at a time until the scan
// decides.  decided = false: more than the list holds is needed (or the list overflowed): whole-wave walk.
template <int MODE>
__device__ __forceinline__ int pj_decide16(const uint32_t w[PJ_LIST], int i, const int *claimA, float nnratio,
                                           int th_dist, bool &decided)
{
    static_assert(PJ_LIST == 16, "four groups of four");
    decided = false;
    if (w[0] == PJ_REWALK)
        return -1;
    PjScan s;
    int result = -1;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (decided)
            continue;
        pj_scan4(w + 4 * g, i, claimA, s);
        result = pj_finish<MODE>(s, pj_dist(w[4 * g + 3]), nnratio, th_dist, decided);
    }
    return decided ? result : -1;
}

// section timestamps of the first workgroup for tools/pj_sections.py; compiled in with -DORBGPU_PJ_TIMING
#ifdef ORBGPU_PJ_TIMING
__device__ long long g_pj_dbg[2048];
__device__ int g_pj_k;
#define PJ_MARK(id) if (threadIdx.x == 0 && blockIdx.x == 0 && g_pj_k < 1000) { g_pj_dbg[2 * g_pj_k] = (id); g_pj_dbg[2 * g_pj_k + 1] = (long long)wall_clock64(); g_pj_k++; }
extern "C" void orbgpu_pj_dbg_dump()
{
    static long long h[2048];
    int k = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pj_dbg), sizeof(h));
    (void)hipMemcpyFromSymbol(&k, HIP_SYMBOL(g_pj_k), sizeof(int));
    for (int i = 0; i < k; i++)
        printf("mark %lld t %lld\n", h[2 * i], h[2 * i + 1]);
    k = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pj_k), &k, sizeof(int));
}
#else
#define PJ_MARK(id)
#endif
// The claim fixpoint as a blocked Gauss-Seidel.  Rows are visited in blocks of blockDim.x consecutive rows, one row
// per thread with its whole candidate list in registers.  Every row of an earlier block is final, so `base` (claims
// of the earlier blocks on top of claim_init) never changes while a block is worked on; inside the block the usual
// sweeps run -- sweep s re-decides every row of the block, hiding what earlier rows claimed in sweep s-1 -- until a
// sweep changes nothing, which makes the block's rows final (row p of the block after at most p+1 sweeps).  The
// kernel is one workgroup on one CU and bound by that CU's instruction issue (tools/pj_sections.py: a 16-wave sweep
// takes ~3 us whatever the barrier count), so what counts is row decisions: 10 blocks x ~3 sweeps x 1024 rows = 31 k
// against 6 sweeps x 10 k rows = 60 k for the all-rows sweeps this replaces, and no list word is read twice.
template <int MODE>
__device__ __forceinline__ void proj_resolve_body(int m, const Query *__restrict__ q,
                                                       const uint8_t *__restrict__ row_desc, const FrameDev &F,
                                                       float nnratio, int th_dist,
                                                       const int *__restrict__ claim_init,
                                                       const uint32_t *__restrict__ lists, int *__restrict__ match,
                                                       int *__restrict__ slow, const float *__restrict__ row_angle,
                                                       int row_angle_stride,
                                                       const float *__restrict__ kp_angle, int check_orientation,
                                                       int *__restrict__ kp_to_mp, int *__restrict__ nmatches,
                                                       int *__restrict__ sweeps_out, int /*unused*/)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int histo[ORBGPU_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count, s_changed, s_nslow;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n = F.n_dev ? min(max(*F.n_dev, 0), F.n) : F.n;
    const int kp_angle_stride = F.kp_stride;
    // LDS: three claim tables of F.n entries that rotate: base / previous sweep / the sweep being built
    int *T0 = reinterpret_cast<int *>(smem), *T1 = T0 + F.n, *T2 = T1 + F.n;
    int *base = T0;
    for (int j = tid; j < n; j += nt)
        base[j] = claim_init[j];
    __syncthreads();
    int sweeps = 0, rewalked = 0;
    const int nblocks = (m + nt - 1) / nt;
    // the next block's row is fetched while the current block is being decided
    auto load_row = [&](int i, uint32_t w[PJ_LIST], bool &active, bool &blocking) {
        active = blocking = false;
#pragma unroll
        for (int k = 0; k < PJ_LIST; k++)
            w[k] = PJ_NONE;
        if (i < m) {
            active = q[i].active;
            blocking = q[i].blocking;
            if (active) {
                const uint4 *src = reinterpret_cast<const uint4 *>(lists + (size_t)i * PJ_LIST);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint4 v = src[g];
                    w[4 * g] = v.x, w[4 * g + 1] = v.y, w[4 * g + 2] = v.z, w[4 * g + 3] = v.w;
                }
            }
        }
    };
    uint32_t wn[PJ_LIST];
    bool act_n, blk_n;
    load_row(tid, wn, act_n, blk_n);
    for (int b = 0; b < nblocks; b++) {
        PJ_MARK(1)  // block start
        const int i = b * nt + tid;
        uint32_t w[PJ_LIST];
#pragma unroll
        for (int k = 0; k < PJ_LIST; k++)
            w[k] = wn[k];
        const bool active = act_n, blocking = blk_n;
        if (b + 1 < nblocks)
            load_row(i + nt, wn, act_n, blk_n);
        int *claimA = base;                       // sweep 0 of the block sees the earlier blocks only
        int *claimB = base == T0 ? T1 : T0;       // the two tables that are not `base`
        int *spare = base == T2 ? T1 : T2;
        int res = -2;
        const int rows_here = min(nt, m - b * nt);
        for (int iter = 0; iter <= rows_here + 1; iter++) {
            for (int j = tid; j < n; j += nt)
                claimB[j] = base[j];
            if (tid == 0) {
                s_changed = 0;
                s_nslow = 0;
            }
            __syncthreads();
            bool changed = false, queued = false;
            if (i < m) {
                int result = -1;
                bool decided = true;
                if (active)
                    result = pj_decide16<MODE>(w, i, claimA, nnratio, th_dist, decided);
                if (!decided) {
                    slow[atomicAdd(&s_nslow, 1)] = i;
                    queued = true;
                } else {
                    if (res != result) {
                        changed = true;
                        res = result;
                    }
                    if (result >= 0 && blocking)
                        atomicMin(&claimB[result], i);
                }
            }
            __syncthreads();
            const int nslow = s_nslow;
            if (nslow) {  // rows whose list cannot decide: exact re-walk of the window by a whole wave
                rewalked += nslow;
                const int wave = tid >> 6, nw = nt >> 6;
                for (int r = wave; r < nslow; r += nw) {
                    const int ir = slow[r];
                    const Query Q = q[ir];
                    const uint64_t *da = reinterpret_cast<const uint64_t *>(row_desc) + (size_t)ir * 4;
                    const uint64_t a[4] = {da[0], da[1], da[2], da[3]};
                    uint64_t t[4];
                    proj_walk(Q, a, F, claimA, ir, t);
                    top4_wave_merge64(t);
                    if ((tid & 63) == 0) {
                        int result = -1;
                        if (t[0] != KEY_NONE)
                            result = proj_accept<MODE>(t[0], t[1], F, nnratio, th_dist);
                        match[ir] = result;  // handed back to the row's thread below
                        if (result >= 0 && Q.blocking)
                            atomicMin(&claimB[result], ir);
                    }
                }
                __syncthreads();
                if (queued) {
                    const int result = match[i];
                    if (res != result) {
                        changed = true;
                        res = result;
                    }
                }
            }
            if (changed)
                s_changed = 1;
            __syncthreads();
            PJ_MARK(6)  // sweep end
            sweeps++;
            const bool again = s_changed != 0;
            __syncthreads();
            if (!again) {
                base = claimB;  // nothing changed: the table just built is the block's final one
                break;
            }
            // rotate: the table just built becomes the previous sweep's, the old previous one (never `base`) is rebuilt
            int *prev = claimA;
            claimA = claimB;
            claimB = prev == base ? spare : prev;
        }
        if (i < m)
            match[i] = res;
    }
    __syncthreads();
    PJ_MARK(7)  // finish
    int *claimA = base == T0 ? T1 : T0;  // scratch for the finish

    // ---- finish
    int *last_claim = claimA;  // reuse
    for (int j = tid; j < n; j += nt)
        last_claim[j] = -1;
    if (tid < ORBGPU_HISTO_LENGTH)
        histo[tid] = 0;
    if (tid == 0)
        s_count = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < m; i += nt) {
        const int j = match[i];
        if (j < 0)
            continue;
        cnt++;
        atomicMax(&last_claim[j], i);
        if (check_orientation)
            atomicAdd(&histo[rot_bin(row_angle[(size_t)i * row_angle_stride], kp_angle[j * kp_angle_stride])], 1);
    }
    __syncthreads();
    for (int j = tid; j < n; j += nt)
        if (last_claim[j] >= 0)
            kp_to_mp[j] = last_claim[j];  // any claim overrides a previous (non-blocking) association
    __syncthreads();
    if (check_orientation) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(histo, ORBGPU_HISTO_LENGTH, i1, i2, i3);
            s_keep[0] = i1;
            s_keep[1] = i2;
            s_keep[2] = i3;
        }
        __syncthreads();
        for (int i = tid; i < m; i += nt) {
            const int j = match[i];
            if (j < 0)
                continue;
            const int b = rot_bin(row_angle[(size_t)i * row_angle_stride], kp_angle[j * kp_angle_stride]);
            if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {
                kp_to_mp[j] = -1;
                cnt--;
            }
        }
    }
    cnt = wave_reduce_add(cnt);
    if ((tid & 63) == 0)
        atomicAdd(&s_count, cnt);
    __syncthreads();
    PJ_MARK(8)  // end
    if (tid == 0) {
        *nmatches = s_count;
        sweeps_out[0] = sweeps;
        sweeps_out[1] = rewalked;
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_proj_resolve(int m, const Query *__restrict__ q,
                                                       const uint8_t *__restrict__ row_desc, FrameDev F,
                                                       float nnratio, int th_dist,
                                                       const int *__restrict__ claim_init,
                                                       const uint32_t *__restrict__ lists, int *__restrict__ match,
                                                       int *__restrict__ slow, const float *__restrict__ row_angle,
                                                       int row_angle_stride,
                                                       const float *__restrict__ kp_angle, int check_orientation,
                                                       int *__restrict__ kp_to_mp, int *__restrict__ nmatches,
                                                       int *__restrict__ sweeps_out, int novf)
{
    proj_resolve_body<MODE>(m, q, row_desc, F, nnratio, th_dist, claim_init, lists, match, slow, row_angle,
                            row_angle_stride, kp_angle, check_orientation, kp_to_mp, nmatches, sweeps_out, novf);
}

// LDS of k_proj_resolve: three claim tables of one int per key point slot
constexpr size_t PJ_RESOLVE_MAX_LDS = 150 * 1024;
static inline size_t resolve_lds(int /*m*/, int ncap, int *novf)
{
    *novf = 0;
    return (size_t)12 * ncap + 16;
}

// ---- host side -----------------------------------------------------------------------------
// Per-thread workspace: device buffers that grow on demand and are reused by every call of this host
// thread (the reference constructs an ORBmatcher on the stack per call site; allocation per call would
// dominate the kernel time).
struct ProjWorkspace {
    int device = -1;
    hipStream_t stream = nullptr;
    unsigned attr_set = 0;  // bit per call site: the dynamic-LDS limit of its kernels has been raised ON THIS DEVICE
    DevBuf kp_x, kp_y, kp_octave, u_right, desc, cell_start, cell_items, kp_angle;
    DevBuf queries, row_desc, row_angle, claim_init, topk, match, slow, k2m, out, inv_sigma2, tri, problems, sweeps;
    void release_device_resources()
    {
        DevBuf *bufs[] = {&kp_x,  &kp_y,  &kp_octave, &u_right, &desc, &cell_start, &cell_items, &kp_angle, &queries, &row_desc, &row_angle,
                          &claim_init, &topk, &match, &slow, &k2m, &out, &inv_sigma2, &tri, &problems, &sweeps};
        for (DevBuf *b : bufs)
            b->release();
        if (stream)
            (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
    ~ProjWorkspace()
    {
        // end of the owning thread (or proj_workspace_delete): give the stream and the buffers back -- unless the process
        // is exiting, when the HIP runtime may already be gone and everything goes with the process (workspace.h)
        if (device >= 0 && !process_exiting().load()) {
            (void)hipSetDevice(device);
            if (stream)
                (void)hipStreamSynchronize(stream);
            release_device_resources();
        }
    }
};

// A caller that brings its own workspace (the MapPoint table: its calls run on the table's stream and must not share
// buffers with asynchronous calls the same thread has in flight on another stream) installs it for the duration of a call.
static thread_local ProjWorkspace *t_ws_override = nullptr;
ProjWorkspace *proj_workspace_new() { return new (std::nothrow) ProjWorkspace(); }
void proj_workspace_delete(ProjWorkspace *ws) { delete ws; }  // the destructor releases
ProjWorkspaceScope::ProjWorkspaceScope(ProjWorkspace *ws) : prev_(t_ws_override) { t_ws_override = ws; }
ProjWorkspaceScope::~ProjWorkspaceScope() { t_ws_override = prev_; }

static int workspace(int device_id, ProjWorkspace **out)
{
    ProjWorkspace &ws = t_ws_override ? *t_ws_override : per_device_workspace<ProjWorkspace>(device_id);  // (the caller has selected device_id)
    if (ws.device != device_id) {  // first use of this device by this thread
        ws.device = device_id;
        hipError_t e = hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            ws.device = -1;
            set_error("hipStreamCreate: %s", hipGetErrorString(e));
            return ORBGPU_EHIP;
        }
    }
    *out = &ws;
    return ORBGPU_OK;
}

static int put(DevBuf &b, const void *src, size_t bytes, hipStream_t st)
{
    int rc = b.reserve(std::max<size_t>(bytes, 16));
    if (rc != ORBGPU_OK)
        return rc;
    if (src && bytes)
        ORBGPU_HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    return ORBGPU_OK;
}

int validate_frame(const orbgpu_frame_view *f)
{
    ORBGPU_REQUIRE(f, "null frame view");
    ORBGPU_REQUIRE(f->n >= 0 && f->n <= 16384, "frame key point count out of range (max 16384)");
    ORBGPU_REQUIRE(f->nlevels >= 1 && f->nlevels <= ORBGPU_MAX_LEVELS && f->scale_factors, "bad scale factors");
    ORBGPU_REQUIRE(f->cell_start && f->cell_items, "null grid");
    if (f->n > 0)
        ORBGPU_REQUIRE(f->kp_x && f->kp_y && f->kp_octave && f->u_right && f->desc, "null frame arrays");
    const int nc = GC * GR;
    ORBGPU_REQUIRE(f->cell_start[0] == 0 && f->cell_start[nc] <= f->n, "grid CSR inconsistent");
    for (int c = 0; c < nc; c++) {
        const int cnt = f->cell_start[c + 1] - f->cell_start[c];
        ORBGPU_REQUIRE(cnt >= 0 && cnt < 4096, "grid cell %d holds %d items (limit 4095)", c, cnt);
    }
    for (int t = 0; t < f->cell_start[nc]; t++)
        ORBGPU_REQUIRE(f->cell_items[t] >= 0 && f->cell_items[t] < f->n, "grid item out of range");
    return ORBGPU_OK;
}

#define PJ_TRY(x)                                                                                            \
    do {                                                                                                     \
        int rc__ = (x);                                                                                      \
        if (rc__ != ORBGPU_OK)                                                                               \
            return rc__;                                                                                     \
    } while (0)

static int upload_frame(ProjWorkspace &ws, const orbgpu_frame_view *f, FrameDev &F)
{
    const size_t n = (size_t)f->n;
    hipStream_t st = ws.stream;
    PJ_TRY(put(ws.kp_x, f->kp_x, n * 4, st));
    PJ_TRY(put(ws.kp_y, f->kp_y, n * 4, st));
    PJ_TRY(put(ws.kp_octave, f->kp_octave, n * 4, st));
    PJ_TRY(put(ws.u_right, f->u_right, n * 4, st));
    PJ_TRY(put(ws.desc, f->desc, n * 32, st));
    PJ_TRY(put(ws.cell_start, f->cell_start, (size_t)(GC * GR + 1) * 4, st));
    PJ_TRY(put(ws.cell_items, f->cell_items, std::max<size_t>((size_t)f->cell_start[GC * GR], 1) * 4, st));
    F.n = f->n;
    F.n_dev = nullptr;
    F.kp_stride = 1;
    F.kp_x = ws.kp_x.as<float>();
    F.kp_y = ws.kp_y.as<float>();
    F.kp_octave = ws.kp_octave.as<int>();
    F.u_right = ws.u_right.as<float>();
    F.desc = ws.desc.as<uint8_t>();
    F.min_x = f->min_x;
    F.min_y = f->min_y;
    F.inv_w = f->grid_inv_w;
    F.inv_h = f->grid_inv_h;
    F.cell_start = ws.cell_start.as<int>();
    F.cell_items = ws.cell_items.as<int>();
    F.inv_sigma2 = nullptr;
    return ORBGPU_OK;
}

// Runs pass 1 + pass 2 for prepared queries. kp_to_mp (host, in/out).
template <int MODE>
static int run_projection(ProjWorkspace &ws, const FrameDev &F, const std::vector<Query> &queries,
                          const uint8_t *row_desc_host, const float *row_angle_host, const float *kp_angle_host,
                          const std::vector<int> &claim_init, float nnratio, int th_dist, int check_orientation,
                          int32_t *kp_to_mp, int32_t *nmatches, int32_t *row_match = nullptr)
{
    const int m = (int)queries.size(), n = F.n;
    if (m == 0 || n == 0) {
        *nmatches = 0;
        for (int i = 0; row_match && i < m; i++)
            row_match[i] = -1;
        return ORBGPU_OK;
    }
    hipStream_t st = ws.stream;
    PJ_TRY(put(ws.queries, queries.data(), sizeof(Query) * m, st));
    PJ_TRY(put(ws.row_desc, row_desc_host, (size_t)m * 32, st));
    PJ_TRY(put(ws.claim_init, claim_init.data(), sizeof(int) * n, st));
    PJ_TRY(put(ws.k2m, kp_to_mp, sizeof(int) * n, st));
    PJ_TRY(ws.topk.reserve(sizeof(uint32_t) * PJ_LIST * (size_t)m));
    PJ_TRY(ws.match.reserve(sizeof(int) * m));
    PJ_TRY(ws.slow.reserve(sizeof(int) * m));
    PJ_TRY(ws.out.reserve(4 * sizeof(int)));
    if (check_orientation) {
        PJ_TRY(put(ws.row_angle, row_angle_host, sizeof(float) * m, st));
        PJ_TRY(put(ws.kp_angle, kp_angle_host, sizeof(float) * n, st));
    }
    if (!(ws.attr_set & 1u)) {
        ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PJ_RESOLVE_MAX_LDS + 64));
        ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PJ_RESOLVE_MAX_LDS + 64));
        ws.attr_set |= 1u;
    }
    hipLaunchKernelGGL(k_proj_lists, dim3((m + PJ_ROWS_PER_BLOCK - 1) / PJ_ROWS_PER_BLOCK), dim3(256), 0, st, m, ws.queries.as<Query>(),
                       ws.row_desc.as<uint8_t>(), F, ws.topk.as<uint32_t>());
    int novf = 0;
    const size_t lds = resolve_lds(m, n, &novf);
    hipLaunchKernelGGL(k_proj_resolve<MODE>, dim3(1), dim3(1024), lds, st, m, ws.queries.as<Query>(),
                       ws.row_desc.as<uint8_t>(), F, nnratio, th_dist, ws.claim_init.as<int>(), ws.topk.as<uint32_t>(),
                       ws.match.as<int>(), ws.slow.as<int>(), ws.row_angle.as<float>(), 1, ws.kp_angle.as<float>(),
                       check_orientation, ws.k2m.as<int>(), ws.out.as<int>(), ws.out.as<int>() + 1, novf);
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpyAsync(kp_to_mp, ws.k2m.p, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(nmatches, ws.out.p, sizeof(int), hipMemcpyDeviceToHost, st));
    if (row_match)  // the decision of every row (key point index or -1): what the claim-free matchers return
        ORBGPU_HIP_TRY(hipMemcpyAsync(row_match, ws.match.p, sizeof(int) * m, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

// cv::Mat 3x3 * 3x1 + 3x1 (CV_32F): cv::gemm small-matrix path -> float products summed left to
// right, then one add of the C term (adopted convention, DESIGN.md "float conventions").
static void rt_apply(const float *T, const float *p, float *out)
{
    for (int i = 0; i < 3; i++) {
        volatile float a = T[4 * i + 0] * p[0];
        volatile float b = T[4 * i + 1] * p[1];
        volatile float c = T[4 * i + 2] * p[2];
        volatile float t0 = a + b;
        volatile float t1 = t0 + c;
        out[i] = t1 + T[4 * i + 3];
    }
}
static void minus_rt_t(const float *T, float *out)
{
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++)
            s += (double)T[4 * k + i] * (double)T[4 * k + 3];
        out[i] = (float)(s * -1.0);
    }
}

// ---- device-resident Tracking::SearchLocalPoints (Tracking.cc:1447-1497) -----------------------------
struct FrustumParams {
    float T[12];   // rows of [Rcw | tcw]
    float Ow[3];   // camera centre, -Rcw^T tcw (Frame.cc:266)
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float log_sf, cos_limit, th;
    int nlevels;
    float scale_factors[ORBGPU_MAX_LEVELS];
};

// Thread i < m: Frame::isInFrustum (Frame.cc:269-325) + MapPoint::PredictScale (MapPoint.cc:385-394) for map
// point i, then the query row of ORBmatcher::SearchByProjection (:45-137: radius by viewing cosine, levels
// [l-1, l]).  Thread j < cap: the claim table entry of key point j from the incoming association.
__device__ __forceinline__ void frustum_queries_body(int m, const float *__restrict__ world_pos,
                                                         const float *__restrict__ normal,
                                                         const float *__restrict__ min_dist,
                                                         const float *__restrict__ max_dist,
                                                         const uint8_t *__restrict__ skip,
                                                         const uint8_t *__restrict__ obs_pos, const FrustumParams &P,
                                                         Query *__restrict__ q, int cap,
                                                         const int *__restrict__ kp_to_mp, int *__restrict__ claim_init,
                                                         const orbgpu_track_scratch &out, int *__restrict__ bad_levels)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        const int v = kp_to_mp[i];
        const bool held = v == -2 || (v >= 0 && v < m && (obs_pos ? obs_pos[v] != 0 : true));
        claim_init[i] = held ? -1 : INT_MAX;
    }
    if (i >= m)
        return;
    Query Q{};
    Q.blocking = obs_pos ? (obs_pos[i] != 0) : 1;
    bool in_view = false;
    float u = 0.f, v = 0.f, ur = 0.f, view_cos = 0.f;
    int level = 0;
    if (!(skip && skip[i])) {
        const float X = world_pos[3 * i], Y = world_pos[3 * i + 1], Z = world_pos[3 * i + 2];
        // mRcw*P + mtcw: float products summed left to right, then the translation (same convention as the host
        // entry points, DESIGN.md "float conventions")
        float Pc[3];
#pragma unroll
        for (int r = 0; r < 3; r++)
            Pc[r] = ((P.T[4 * r] * X + P.T[4 * r + 1] * Y) + P.T[4 * r + 2] * Z) + P.T[4 * r + 3];
        do {
            if (Pc[2] < 0.0f)
                break;
            const float invz = 1.0f / Pc[2];
            u = P.fx * Pc[0] * invz + P.cx;
            v = P.fy * Pc[1] * invz + P.cy;
            if (u < P.min_x || u > P.max_x || v < P.min_y || v > P.max_y)
                break;
            const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
            const float PO[3] = {X - P.Ow[0], Y - P.Ow[1], Z - P.Ow[2]};
            const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
            if (dist < minDistance || dist > maxDistance)
                break;
            const double dot = (double)PO[0] * normal[3 * i] + (double)PO[1] * normal[3 * i + 1] +
                               (double)PO[2] * normal[3 * i + 2];
            view_cos = (float)(dot / (double)dist);
            if (view_cos < P.cos_limit)
                break;
            const float ratio = max_dist[i] / dist;
            // std::log(float): evaluated in double and rounded, which is the correctly rounded float logarithm
            level = (int)ceilf((float)log((double)ratio) / P.log_sf);
            ur = u - P.mbf * invz;
            in_view = true;
        } while (0);
    }
    if (in_view && (level < 0 || level >= P.nlevels)) {
        // the reference indexes mvScaleFactors out of range here (undefined); such points are left unmatched
        atomicAdd(bad_levels, 1);
        in_view = false;
    }
    if (in_view) {
        float r = (double)view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:131-137
        if (P.th != 1.0f)
            r *= P.th;
        Q.r = r * P.scale_factors[level];
        Q.x = u;
        Q.y = v;
        Q.ur = ur;
        Q.min_level = level - 1;
        Q.max_level = level;
        Q.check_ur = 1;
        Q.active = 1;
    }
    q[i] = Q;
    if (out.in_view)
        out.in_view[i] = in_view ? 1 : 0;
    if (in_view) {
        if (out.proj_x)
            out.proj_x[i] = u;
        if (out.proj_y)
            out.proj_y[i] = v;
        if (out.proj_xr)
            out.proj_xr[i] = ur;
        if (out.level)
            out.level[i] = level;
        if (out.view_cos)
            out.view_cos[i] = view_cos;
    }
}

__global__ __launch_bounds__(256) void k_frustum_queries(int m, const float *__restrict__ world_pos,
                                                         const float *__restrict__ normal,
                                                         const float *__restrict__ min_dist,
                                                         const float *__restrict__ max_dist,
                                                         const uint8_t *__restrict__ skip,
                                                         const uint8_t *__restrict__ obs_pos, FrustumParams P,
                                                         Query *__restrict__ q, int cap,
                                                         const int *__restrict__ kp_to_mp, int *__restrict__ claim_init,
                                                         orbgpu_track_scratch out, int *__restrict__ bad_levels)
{
    frustum_queries_body(m, world_pos, normal, min_dist, max_dist, skip, obs_pos, P, q, cap, kp_to_mp, claim_init, out,
                         bad_levels);
}

// The same rows from MapPoint::mTrack* members the caller's own Frame::isInFrustum already filled (the drop-in at the
// ORBmatcher level: ORBmatcher.cc:53-75 reads mbTrackInView, mnTrackScaleLevel, mTrackViewCos, mTrackProjX/Y/XR); the
// arrays were uploaded for this call, skip / obs_pos come from the device-resident MapPoint table.
__global__ __launch_bounds__(256) void k_scratch_queries(int m, ScratchDev sc, const uint8_t *__restrict__ skip,
                                                         const uint8_t *__restrict__ obs_pos, float th, int nlevels,
                                                         FrustumParams P, Query *__restrict__ q, int cap,
                                                         const int *__restrict__ kp_to_mp, int *__restrict__ claim_init,
                                                         int *__restrict__ bad_levels)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        const int v = kp_to_mp[i];
        const bool held = v == -2 || (v >= 0 && v < m && (obs_pos ? obs_pos[v] != 0 : true));
        claim_init[i] = held ? -1 : INT_MAX;
    }
    if (i >= m)
        return;
    Query Q{};
    Q.blocking = obs_pos ? (obs_pos[i] != 0) : 1;
    if (sc.in_view[i] && !(skip && skip[i])) {
        const int lvl = sc.level[i];
        if (lvl < 0 || lvl >= nlevels) {
            atomicAdd(bad_levels, 1);  // H5: the reference indexes mvScaleFactors out of range
        } else {
            float r = (double)sc.view_cos[i] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, ORBmatcher.cc:131-137
            if (th != 1.0f)
                r *= th;
            Q.r = r * P.scale_factors[lvl];
            Q.x = sc.proj_x[i];
            Q.y = sc.proj_y[i];
            Q.ur = sc.proj_xr[i];
            Q.min_level = lvl - 1;
            Q.max_level = lvl;
            Q.check_ur = 1;
            Q.active = 1;
        }
    }
    q[i] = Q;
}

// ---- many independent SearchLocalPoints problems (one per sequence) in three launches ------------------------
// The claim fixpoint of one frame is one workgroup by construction; independent sequences (the unit of
// sharding) fill the device: blockIdx.y (queries, lists) / blockIdx.x (resolve) selects the problem.
struct ProjProblem {
    int m, cap, novf, pad;
    const float *world_pos, *normal, *min_dist, *max_dist;
    const uint8_t *desc, *skip, *obs_pos;
    FrustumParams P;
    FrameDev F;
    Query *q;
    int *kp_to_mp, *claim_init, *counts, *match, *slow, *sweeps;
    uint32_t *lists;
    float nnratio;
    orbgpu_track_scratch track;
};

__global__ void k_batch_zero_counts(const ProjProblem *__restrict__ problems, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        problems[i].counts[0] = 0;
        problems[i].counts[1] = 0;
    }
}

__global__ __launch_bounds__(256) void k_frustum_queries_batch(const ProjProblem *__restrict__ problems)
{
    const ProjProblem &p = problems[blockIdx.y];
    if ((int)(blockIdx.x * blockDim.x) >= max(p.m, p.cap))
        return;
    frustum_queries_body(p.m, p.world_pos, p.normal, p.min_dist, p.max_dist, p.skip, p.obs_pos, p.P, p.q, p.cap,
                         p.kp_to_mp, p.claim_init, p.track, p.counts + 1);
}

__global__ __launch_bounds__(256) void k_proj_lists_batch(const ProjProblem *__restrict__ problems)
{
    const ProjProblem &p = problems[blockIdx.y];
    if ((int)(blockIdx.x * PJ_ROWS_PER_BLOCK) >= p.m)
        return;
    proj_lists_body(p.m, p.q, p.desc, p.F, p.lists);
}

__global__ __launch_bounds__(1024) void k_proj_resolve_batch(const ProjProblem *__restrict__ problems)
{
    const ProjProblem &p = problems[blockIdx.x];
    if (p.m <= 0)
        return;
    proj_resolve_body<0>(p.m, p.q, p.desc, p.F, p.nnratio, (int)ORBGPU_TH_HIGH, p.claim_init, p.lists, p.match, p.slow,
                         (const float *)nullptr, 1, (const float *)nullptr, 0, p.kp_to_mp, p.counts, p.sweeps, p.novf);
}

// ---- device-resident ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (:1328-1470) --------
struct LastParams {
    float T[12];  // rows of the current [Rcw | tcw]
    float fx, fy, cx, cy, mbf;
    float min_x, max_x, min_y, max_y;
    float th;
    int forward, backward;  // :1348-1349
    int nlevels;
    float scale_factors[ORBGPU_MAX_LEVELS];
};

// Thread i < rows: the query of last-frame key point i (:1353-1397): its map point projected with the current
// pose, window th * scale[octave of the last key point], level window by the direction of motion.  Same float
// conventions as the host entry point (products summed left to right, no contraction; invzc = 1.0 / z in double).
// Thread j < cap: claim table entry of current key point j from the incoming association.
__global__ __launch_bounds__(256) void k_project_last_queries(int rows, const int *__restrict__ n_last,
                                                              const orbgpu_keypoint *__restrict__ last_kps,
                                                              const uint8_t *__restrict__ has_mp,
                                                              const uint8_t *__restrict__ outlier,
                                                              const uint8_t *__restrict__ obs_pos,
                                                              const float *__restrict__ world_pos, LastParams P,
                                                              Query *__restrict__ q, int cap,
                                                              const int *__restrict__ kp_to_mp,
                                                              int *__restrict__ claim_init, int *__restrict__ bad_levels)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        const int v = kp_to_mp[i];
        const bool held = v == -2 || (v >= 0 && v < rows && (obs_pos ? obs_pos[v] != 0 : true));
        claim_init[i] = held ? -1 : INT_MAX;
    }
    if (i >= rows)
        return;
    Query Q{};
    Q.blocking = obs_pos ? (obs_pos[i] != 0) : 1;
    const int nl = min(max(*n_last, 0), rows);
    if (i < nl && has_mp[i] && !(outlier && outlier[i])) {
        const float X = world_pos[3 * i], Y = world_pos[3 * i + 1], Z = world_pos[3 * i + 2];
        float xc[3];
#pragma unroll
        for (int r = 0; r < 3; r++)
            xc[r] = ((P.T[4 * r] * X + P.T[4 * r + 1] * Y) + P.T[4 * r + 2] * Z) + P.T[4 * r + 3];
        const float invzc = (float)(1.0 / (double)xc[2]);
        if (!(invzc < 0)) {
            const float u = P.fx * xc[0] * invzc + P.cx;
            const float v = P.fy * xc[1] * invzc + P.cy;
            if (!(u < P.min_x || u > P.max_x) && !(v < P.min_y || v > P.max_y)) {
                const int oct = last_kps[i].octave;
                if (oct < 0 || oct >= P.nlevels) {
                    atomicAdd(bad_levels, 1);  // the reference reads mvScaleFactors out of range here
                } else {
                    Q.r = P.th * P.scale_factors[oct];
                    Q.x = u;
                    Q.y = v;
                    Q.ur = u - P.mbf * invzc;
                    if (P.forward) {
                        Q.min_level = oct;
                        Q.max_level = -1;
                    } else if (P.backward) {
                        Q.min_level = 0;
                        Q.max_level = oct;
                    } else {
                        Q.min_level = oct - 1;
                        Q.max_level = oct + 1;
                    }
                    Q.check_ur = 1;
                    Q.active = 1;
                }
            }
        }
    }
    q[i] = Q;
}

// ---- ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823) ---------------------------------------------
struct TriParams {
    float F12[9];
    float ex, ey;  // epipole in the second image (:664-670)
    int only_stereo;
    float sigma2[ORBGPU_MAX_LEVELS], scale[ORBGPU_MAX_LEVELS];  // of key frame 2
};

// One wave per key point of key frame 1 without a map point: its candidates are the key points of key frame 2 under
// the same vocabulary node (node2 == node1 >= 0) without a map point; Hamming <= TH_LOW, the epipole distance test
// for monocular pairs, the epipolar-line test (:140-157), least distance, the LAST among equals in the node's list
// (`dist > bestDist` is the skip, :739), i.e. the largest index.  The reference never sets vbMatched2, so rows do
// not interact.  Rotation histogram: k_tri_finish.
__global__ __launch_bounds__(256) void k_tri_match(int n1, const float *__restrict__ x1, const float *__restrict__ y1,
                                                   const float *__restrict__ ur1, const uint8_t *__restrict__ has_mp1,
                                                   const int *__restrict__ node1, const uint8_t *__restrict__ desc1,
                                                   int n2, const float *__restrict__ x2, const float *__restrict__ y2,
                                                   const int *__restrict__ oct2, const float *__restrict__ ur2,
                                                   const uint8_t *__restrict__ has_mp2, const int *__restrict__ node2,
                                                   const uint8_t *__restrict__ desc2, TriParams P,
                                                   int *__restrict__ match12)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n1)
        return;
    int best = -1;
    const int nd = node1[i];
    const bool stereo1 = ur1[i] >= 0;
    if (nd >= 0 && !has_mp1[i] && !(P.only_stereo && !stereo1)) {
        const uint64_t *pa = reinterpret_cast<const uint64_t *>(desc1) + (size_t)i * 4;
        const uint64_t a[4] = {pa[0], pa[1], pa[2], pa[3]};
        const float kx = x1[i], ky = y1[i];
        // epipolar line in the second image l = x1' F12 (:143-145)
        const float la = kx * P.F12[0] + ky * P.F12[3] + P.F12[6];
        const float lb = kx * P.F12[1] + ky * P.F12[4] + P.F12[7];
        const float lc = kx * P.F12[2] + ky * P.F12[5] + P.F12[8];
        const float den = la * la + lb * lb;
        uint32_t key = 0xFFFFFFFFu;  // distance << 16 | (65535 - idx2): least distance, then largest index
        for (int j = lane; j < n2; j += 64) {
            if (node2[j] != nd || has_mp2[j])
                continue;
            const bool stereo2 = ur2[j] >= 0;
            if (P.only_stereo && !stereo2)
                continue;
            const uint64_t *pb = reinterpret_cast<const uint64_t *>(desc2) + (size_t)j * 4;
            const uint64_t b[4] = {pb[0], pb[1], pb[2], pb[3]};
            const int dist = hamming256(a, b);
            if (dist > ORBGPU_TH_LOW)
                continue;
            const float px = x2[j], py = y2[j];
            const int o = oct2[j];
            if (!stereo1 && !stereo2) {
                const float dex = P.ex - px, dey = P.ey - py;
                if (dex * dex + dey * dey < 100 * P.scale[o])
                    continue;
            }
            const float num = la * px + lb * py + lc;
            if (den == 0)
                continue;
            const float dsqr = num * num / den;
            if (!((double)dsqr < 3.84 * (double)P.sigma2[o]))
                continue;
            key = min(key, ((uint32_t)dist << 16) | (uint32_t)(65535 - j));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            key = min(key, (uint32_t)__shfl_xor((int)key, off, 64));
        if (key != 0xFFFFFFFFu)
            best = 65535 - (int)(key & 0xFFFFu);
    }
    if (lane == 0)
        match12[i] = best;
}

// rotation consistency of the accepted pairs (:772-795) and the count; one workgroup
__global__ __launch_bounds__(1024) void k_tri_finish(int n1, const float *__restrict__ angle1,
                                                     const float *__restrict__ angle2, int check_orientation,
                                                     int *__restrict__ match12, int *__restrict__ nmatches)
{
    __shared__ int histo[ORBGPU_HISTO_LENGTH];
    __shared__ int s_keep[3], s_count;
    const int tid = threadIdx.x;
    if (tid < ORBGPU_HISTO_LENGTH)
        histo[tid] = 0;
    if (tid == 0)
        s_count = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = tid; i < n1; i += 1024) {
        const int j = match12[i];
        if (j < 0)
            continue;
        cnt++;
        if (check_orientation)
            atomicAdd(&histo[rot_bin(angle1[i], angle2[j])], 1);
    }
    __syncthreads();
    if (check_orientation) {
        if (tid == 0) {
            int i1, i2, i3;
            three_maxima(histo, ORBGPU_HISTO_LENGTH, i1, i2, i3);
            s_keep[0] = i1, s_keep[1] = i2, s_keep[2] = i3;
        }
        __syncthreads();
        for (int i = tid; i < n1; i += 1024) {
            const int j = match12[i];
            if (j < 0)
                continue;
            const int b = rot_bin(angle1[i], angle2[j]);
            if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {
                match12[i] = -1;
                cnt--;
            }
        }
    }
    cnt = wave_reduce_add(cnt);
    if ((tid & 63) == 0)
        atomicAdd(&s_count, cnt);
    __syncthreads();
    if (tid == 0)
        *nmatches = s_count;
}

// ---- ORBmatcher::SearchForInitialization (ORBmatcher.cc:405-520): distances of every window candidate ---------
// One wave per level-0 key point of F1: all key points of F2 inside its window (GetFeaturesInArea(x, y, windowSize,
// 0, 0)) with their Hamming distance, as 64-bit keys distance << 44 | cell sequence << 32 | position in cell << 20
// | index (the visiting order of the reference is the order of bits 43..20).  The steal-if-strictly-closer
// bookkeeping (:441-470) is a sequential dependency between rows and runs on the host over these lists.
__global__ __launch_bounds__(256) void k_window_candidates(int rows, const Query *__restrict__ q,
                                                           const uint8_t *__restrict__ row_desc, FrameDev F, int cap,
                                                           uint64_t *__restrict__ keys, int *__restrict__ counts)
{
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w;
    if (lane == 0)
        s_cnt[w] = 0;
    __syncthreads();
    if (i < rows) {
        const Query Q = q[i];
        if (Q.active) {
            const uint64_t *da = reinterpret_cast<const uint64_t *>(row_desc) + (size_t)i * 4;
            const uint64_t a[4] = {da[0], da[1], da[2], da[3]};
            proj_walk_each(Q, a, F, nullptr, i, [&](uint64_t key, int) {
                const int slot = atomicAdd(&s_cnt[w], 1);
                if (slot < cap)
                    keys[(size_t)i * cap + slot] = key;
            });
        }
    }
    __syncthreads();
    if (i < rows && lane == 0)
        counts[i] = s_cnt[w];
}

} // namespace orbgpu

using namespace orbgpu;

extern "C" {

int orbgpu_assign_features_to_grid(int32_t n, const float *kp_x, const float *kp_y, float min_x, float min_y,
                                   float inv_w, float inv_h, int32_t *cell_start, int32_t *cell_items)
{
    // Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:230-245, 382-392). O(n) bookkeeping that the
    // reference also does on the host right after extraction; it is a layout conversion of the
    // boundary (mGrid -> CSR), not a compute stage.
    ORBGPU_REQUIRE(n >= 0 && cell_start && (n == 0 || (kp_x && kp_y && cell_items)), "bad arguments");
    const int NC = GC * GR;
    std::vector<int> cell_of((size_t)std::max(n, 1));
    for (int c = 0; c <= NC; c++)
        cell_start[c] = 0;
    for (int i = 0; i < n; i++) {
        const int px = (int)roundf((kp_x[i] - min_x) * inv_w);
        const int py = (int)roundf((kp_y[i] - min_y) * inv_h);
        if (px < 0 || px >= GC || py < 0 || py >= GR) {
            cell_of[i] = -1;
            continue;
        }
        cell_of[i] = px * GR + py;
        cell_start[cell_of[i] + 1]++;
    }
    for (int c = 0; c < NC; c++)
        cell_start[c + 1] += cell_start[c];
    std::vector<int> pos(cell_start, cell_start + NC);
    for (int i = 0; i < n; i++)
        if (cell_of[i] >= 0)
            cell_items[pos[cell_of[i]]++] = i;
    return ORBGPU_OK;
}

int orbgpu_search_by_projection(const orbgpu_frame_view *f, const orbgpu_mappoint_view *mp, float th,
                                float nnratio, int32_t *kp_to_mp, int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(mp && kp_to_mp && nmatches, "null argument");
    int rc = validate_frame(f);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(mp->m >= 0, "bad map point count");
    if (mp->m > 0)
        ORBGPU_REQUIRE(mp->in_view && mp->level && mp->view_cos && mp->proj_x && mp->proj_y && mp->proj_xr && mp->desc,
                       "null map point arrays");
    rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const bool bFactor = th != 1.0;
    std::vector<Query> q((size_t)mp->m);
    for (int i = 0; i < mp->m; i++) {
        Query &Q = q[i];
        Q = Query{};
        Q.blocking = mp->obs_pos ? (mp->obs_pos[i] != 0) : 1;
        if (!mp->in_view[i] || (mp->bad && mp->bad[i]))
            continue;
        const int lvl = mp->level[i];
        if (lvl < 0 || lvl >= f->nlevels) {
            set_error("map point %d: predicted level %d outside [0,%d)", i, lvl, f->nlevels);
            return ORBGPU_ELEVEL;  // H5
        }
        float r = (double)mp->view_cos[i] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, :131-137
        if (bFactor)
            r *= th;
        Q.r = r * f->scale_factors[lvl];
        Q.x = mp->proj_x[i];
        Q.y = mp->proj_y[i];
        Q.ur = mp->proj_xr[i];
        Q.min_level = lvl - 1;
        Q.max_level = lvl;
        Q.check_ur = 1;
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(f->n, 1));
    for (int j = 0; j < f->n; j++) {
        const int v = kp_to_mp[j];
        ORBGPU_REQUIRE(v >= -2 && v < mp->m, "kp_to_mp[%d] = %d out of range", j, v);
        const bool held = v == -2 || (v >= 0 && (mp->obs_pos ? mp->obs_pos[v] != 0 : true));
        init[j] = held ? -1 : INT_MAX;
    }
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, f, F)) != ORBGPU_OK)
        return rc;
    return run_projection<0>(*ws, F, q, mp->desc, nullptr, nullptr, init, nnratio, ORBGPU_TH_HIGH, 0, kp_to_mp, nmatches);
}

int orbgpu_search_local_points_device(const orbgpu_device_frame_view *f, const orbgpu_device_mappoint_table *mp,
                                      const float *Tcw, float fx, float fy, float cx, float cy, float mbf,
                                      float log_scale_factor, float cos_limit, float th, float nnratio,
                                      int32_t *d_kp_to_mp, int32_t *d_counts, const orbgpu_track_scratch *d_track,
                                      int32_t device_id, void *hip_stream)
{
    return orbgpu::search_local_points_device_impl(f, mp, nullptr, Tcw, fx, fy, cx, cy, mbf, log_scale_factor, cos_limit,
                                                   th, nnratio, d_kp_to_mp, d_counts, d_track, device_id, hip_stream);
}
} // extern "C"

// scratch != nullptr: the rows come from uploaded mTrack* members (k_scratch_queries; Tcw and the camera are not read),
// otherwise from Frame::isInFrustum on the device (k_frustum_queries)
int orbgpu::search_local_points_device_impl(const orbgpu_device_frame_view *f, const orbgpu_device_mappoint_table *mp,
                                            const ScratchDev *scratch, const float *Tcw, float fx, float fy, float cx,
                                            float cy, float mbf, float log_scale_factor, float cos_limit, float th,
                                            float nnratio, int32_t *d_kp_to_mp, int32_t *d_counts,
                                            const orbgpu_track_scratch *d_track, int32_t device_id, void *hip_stream)
{
    ORBGPU_REQUIRE(f && mp && (Tcw || scratch) && d_kp_to_mp && d_counts, "null argument");
    ORBGPU_REQUIRE(f->cap >= 1 && f->cap <= 16384, "frame capacity out of range (max 16384)");
    ORBGPU_REQUIRE(f->n && f->kps && f->desc && f->u_right && f->cell_start && f->cell_items, "null frame arrays");
    ORBGPU_REQUIRE(f->nlevels >= 1 && f->nlevels <= ORBGPU_MAX_LEVELS && f->scale_factors, "bad scale factors");
    ORBGPU_REQUIRE(f->max_x > f->min_x && f->max_y > f->min_y, "empty image bounds");
    ORBGPU_REQUIRE(mp->m >= 0 && mp->m < (1 << 20), "bad map point count");
    if (mp->m > 0)
        ORBGPU_REQUIRE(mp->desc && (scratch || (mp->world_pos && mp->normal && mp->min_dist && mp->max_dist)),
                       "null map point arrays");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    ORBGPU_HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(int32_t), st));
    if (mp->m == 0)
        return ORBGPU_OK;
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    const int m = mp->m, cap = f->cap;
    PJ_TRY(ws->queries.reserve(sizeof(Query) * (size_t)m));
    PJ_TRY(ws->claim_init.reserve(sizeof(int) * (size_t)cap));
    PJ_TRY(ws->topk.reserve(sizeof(uint32_t) * PJ_LIST * (size_t)m));
    PJ_TRY(ws->match.reserve(sizeof(int) * (size_t)m));
    PJ_TRY(ws->slow.reserve(sizeof(int) * (size_t)m));
    PJ_TRY(ws->out.reserve(4 * sizeof(int)));
    FrustumParams P{};
    if (!scratch) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                P.T[4 * r + c] = Tcw[4 * r + c];
        minus_rt_t(Tcw, P.Ow);
    }
    P.fx = fx, P.fy = fy, P.cx = cx, P.cy = cy, P.mbf = mbf;
    P.min_x = f->min_x, P.max_x = f->max_x, P.min_y = f->min_y, P.max_y = f->max_y;
    P.log_sf = log_scale_factor, P.cos_limit = cos_limit, P.th = th;
    P.nlevels = f->nlevels;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++)
        P.scale_factors[l] = l < f->nlevels ? f->scale_factors[l] : 0.f;
    FrameDev F;
    F.n = cap;
    F.n_dev = f->n;
    F.kp_stride = (int)(sizeof(orbgpu_keypoint) / sizeof(float));
    F.kp_x = reinterpret_cast<const float *>(f->kps);
    F.kp_y = F.kp_x + 1;
    F.kp_octave = reinterpret_cast<const int *>(f->kps) + 5;
    F.u_right = f->u_right;
    F.desc = f->desc;
    F.min_x = f->min_x;
    F.min_y = f->min_y;
    F.inv_w = (float)GC / (f->max_x - f->min_x);  // Frame.cc:155-156
    F.inv_h = (float)GR / (f->max_y - f->min_y);
    F.cell_start = f->cell_start;
    F.cell_items = f->cell_items;
    F.inv_sigma2 = nullptr;
    if (!(ws->attr_set & 2u)) {
        ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PJ_RESOLVE_MAX_LDS + 64));
        ws->attr_set |= 2u;
    }
    orbgpu_track_scratch none{};
    const int cover = std::max(m, cap);
    if (scratch)
        hipLaunchKernelGGL(k_scratch_queries, dim3((cover + 255) / 256), dim3(256), 0, st, m, *scratch, mp->skip, mp->obs_pos,
                           th, f->nlevels, P, ws->queries.as<Query>(), cap, d_kp_to_mp, ws->claim_init.as<int>(),
                           d_counts + 1);
    else
        hipLaunchKernelGGL(k_frustum_queries, dim3((cover + 255) / 256), dim3(256), 0, st, m, mp->world_pos, mp->normal,
                           mp->min_dist, mp->max_dist, mp->skip, mp->obs_pos, P, ws->queries.as<Query>(), cap, d_kp_to_mp,
                           ws->claim_init.as<int>(), d_track ? *d_track : none, d_counts + 1);
    hipLaunchKernelGGL(k_proj_lists, dim3((m + PJ_ROWS_PER_BLOCK - 1) / PJ_ROWS_PER_BLOCK), dim3(256), 0, st, m, ws->queries.as<Query>(), mp->desc, F,
                       ws->topk.as<uint32_t>());
    int novf = 0;
    const size_t lds = resolve_lds(m, cap, &novf);
    hipLaunchKernelGGL(k_proj_resolve<0>, dim3(1), dim3(1024), lds, st, m, ws->queries.as<Query>(), mp->desc,
                       F, nnratio, (int)ORBGPU_TH_HIGH, ws->claim_init.as<int>(), ws->topk.as<uint32_t>(),
                       ws->match.as<int>(), ws->slow.as<int>(), (const float *)nullptr, 1, (const float *)nullptr, 0,
                       d_kp_to_mp, d_counts, ws->out.as<int>() + 1, novf);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

extern "C" {

int orbgpu_search_by_projection_last_device(const orbgpu_device_frame_view *cur, const float *cur_Tcw,
                                            const orbgpu_device_lastframe_view *last, const float *last_Tcw, float fx,
                                            float fy, float cx, float cy, float mbf, float mb, float th, int32_t mono,
                                            int32_t check_orientation, int32_t *d_kp_to_mp, int32_t *d_counts,
                                            int32_t device_id, void *hip_stream)
{
    ORBGPU_REQUIRE(cur && last && cur_Tcw && last_Tcw && d_kp_to_mp && d_counts, "null argument");
    ORBGPU_REQUIRE(cur->cap >= 1 && cur->cap <= 16384, "frame capacity out of range (max 16384)");
    ORBGPU_REQUIRE(cur->n && cur->kps && cur->desc && cur->u_right && cur->cell_start && cur->cell_items, "null frame arrays");
    ORBGPU_REQUIRE(cur->nlevels >= 1 && cur->nlevels <= ORBGPU_MAX_LEVELS && cur->scale_factors, "bad scale factors");
    ORBGPU_REQUIRE(cur->max_x > cur->min_x && cur->max_y > cur->min_y, "empty image bounds");
    ORBGPU_REQUIRE(last->cap >= 0 && last->cap <= 16384, "last-frame capacity out of range (max 16384)");
    if (last->cap > 0)
        ORBGPU_REQUIRE(last->n && last->kps && last->has_mp && last->world_pos && last->desc, "null last-frame arrays");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    ORBGPU_HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(int32_t), st));
    if (last->cap == 0)
        return ORBGPU_OK;
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    const int m = last->cap, cap = cur->cap;
    PJ_TRY(ws->queries.reserve(sizeof(Query) * (size_t)m));
    PJ_TRY(ws->claim_init.reserve(sizeof(int) * (size_t)cap));
    PJ_TRY(ws->topk.reserve(sizeof(uint32_t) * PJ_LIST * (size_t)m));
    PJ_TRY(ws->match.reserve(sizeof(int) * (size_t)m));
    PJ_TRY(ws->slow.reserve(sizeof(int) * (size_t)m));
    PJ_TRY(ws->out.reserve(4 * sizeof(int)));
    // :1339-1349 forward / backward motion from the two poses (host: 2 x 12 floats)
    float twc[3], tlc[3];
    minus_rt_t(cur_Tcw, twc);
    rt_apply(last_Tcw, twc, tlc);
    LastParams P;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++)
            P.T[4 * r + c] = cur_Tcw[4 * r + c];
    P.fx = fx, P.fy = fy, P.cx = cx, P.cy = cy, P.mbf = mbf;
    P.min_x = cur->min_x, P.max_x = cur->max_x, P.min_y = cur->min_y, P.max_y = cur->max_y;
    P.th = th;
    P.forward = (tlc[2] > mb && !mono) ? 1 : 0;
    P.backward = (-tlc[2] > mb && !mono) ? 1 : 0;
    P.nlevels = cur->nlevels;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++)
        P.scale_factors[l] = l < cur->nlevels ? cur->scale_factors[l] : 0.f;
    FrameDev F;
    F.n = cap;
    F.n_dev = cur->n;
    F.kp_stride = (int)(sizeof(orbgpu_keypoint) / sizeof(float));
    F.kp_x = reinterpret_cast<const float *>(cur->kps);
    F.kp_y = F.kp_x + 1;
    F.kp_octave = reinterpret_cast<const int *>(cur->kps) + 5;
    F.u_right = cur->u_right;
    F.desc = cur->desc;
    F.min_x = cur->min_x;
    F.min_y = cur->min_y;
    F.inv_w = (float)GC / (cur->max_x - cur->min_x);  // Frame.cc:155-156
    F.inv_h = (float)GR / (cur->max_y - cur->min_y);
    F.cell_start = cur->cell_start;
    F.cell_items = cur->cell_items;
    F.inv_sigma2 = nullptr;
    if (!(ws->attr_set & 4u)) {
        ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PJ_RESOLVE_MAX_LDS + 64));
        ws->attr_set |= 4u;
    }
    const int cover = std::max(m, cap);
    const int kstride = (int)(sizeof(orbgpu_keypoint) / sizeof(float));
    hipLaunchKernelGGL(k_project_last_queries, dim3((cover + 255) / 256), dim3(256), 0, st, m, last->n, last->kps,
                       last->has_mp, last->outlier, last->obs_pos, last->world_pos, P, ws->queries.as<Query>(), cap,
                       d_kp_to_mp, ws->claim_init.as<int>(), d_counts + 1);
    hipLaunchKernelGGL(k_proj_lists, dim3((m + PJ_ROWS_PER_BLOCK - 1) / PJ_ROWS_PER_BLOCK), dim3(256), 0, st, m, ws->queries.as<Query>(), last->desc, F,
                       ws->topk.as<uint32_t>());
    int novf = 0;
    const size_t lds = resolve_lds(m, cap, &novf);
    hipLaunchKernelGGL(k_proj_resolve<1>, dim3(1), dim3(1024), lds, st, m, ws->queries.as<Query>(),
                       last->desc, F, 0.f, (int)ORBGPU_TH_HIGH, ws->claim_init.as<int>(), ws->topk.as<uint32_t>(),
                       ws->match.as<int>(), ws->slow.as<int>(), reinterpret_cast<const float *>(last->kps) + 3, kstride,
                       reinterpret_cast<const float *>(cur->kps) + 3, check_orientation ? 1 : 0, d_kp_to_mp, d_counts,
                       ws->out.as<int>() + 1, novf);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

int orbgpu_search_local_points_batch_device(int32_t n, const orbgpu_local_points_problem *problems, float cos_limit,
                                            float th, float nnratio, int32_t device_id, void *hip_stream)
{
    ORBGPU_REQUIRE(n >= 0 && (n == 0 || problems), "bad arguments");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK || n == 0)
        return rc;
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    size_t tot_m = 0, tot_cap = 0;
    int max_m = 0, max_cover = 0;
    size_t lds = 0;
    std::vector<int> novfs((size_t)n);
    for (int k = 0; k < n; k++) {
        const orbgpu_local_points_problem &pr = problems[k];
        const orbgpu_device_frame_view *f = pr.frame;
        const orbgpu_device_mappoint_table *mp = pr.table;
        ORBGPU_REQUIRE(f && mp && pr.Tcw && pr.d_kp_to_mp && pr.d_counts, "problem %d: null argument", k);
        ORBGPU_REQUIRE(f->cap >= 1 && f->cap <= 16384, "problem %d: frame capacity out of range (max 16384)", k);
        ORBGPU_REQUIRE(f->n && f->kps && f->desc && f->u_right && f->cell_start && f->cell_items, "problem %d: null frame arrays", k);
        ORBGPU_REQUIRE(f->nlevels >= 1 && f->nlevels <= ORBGPU_MAX_LEVELS && f->scale_factors, "problem %d: bad scale factors", k);
        ORBGPU_REQUIRE(f->max_x > f->min_x && f->max_y > f->min_y, "problem %d: empty image bounds", k);
        ORBGPU_REQUIRE(mp->m >= 0 && mp->m < (1 << 20), "problem %d: bad map point count", k);
        if (mp->m > 0)
            ORBGPU_REQUIRE(mp->world_pos && mp->normal && mp->min_dist && mp->max_dist && mp->desc, "problem %d: null map point arrays", k);
        tot_m += (size_t)std::max(mp->m, 1);
        tot_cap += (size_t)f->cap;
        max_m = std::max(max_m, mp->m);
        max_cover = std::max(max_cover, std::max(mp->m, f->cap));
        lds = std::max(lds, resolve_lds(mp->m, f->cap, &novfs[k]));
    }
    PJ_TRY(ws->queries.reserve(sizeof(Query) * tot_m));
    PJ_TRY(ws->claim_init.reserve(sizeof(int) * tot_cap));
    PJ_TRY(ws->topk.reserve(sizeof(uint32_t) * PJ_LIST * tot_m));
    PJ_TRY(ws->match.reserve(sizeof(int) * tot_m));
    PJ_TRY(ws->slow.reserve(sizeof(int) * tot_m));
    PJ_TRY(ws->sweeps.reserve(sizeof(int) * 2 * (size_t)n));
    PJ_TRY(ws->problems.reserve(sizeof(ProjProblem) * (size_t)n));
    PJ_TRY(ws->out.reserve(4 * sizeof(int)));
    std::vector<ProjProblem> hp((size_t)n);
    size_t om = 0, oc = 0;
    for (int k = 0; k < n; k++) {
        const orbgpu_local_points_problem &pr = problems[k];
        const orbgpu_device_frame_view *f = pr.frame;
        const orbgpu_device_mappoint_table *mp = pr.table;
        ProjProblem &P = hp[k];
        memset(&P, 0, sizeof(P));
        P.m = mp->m, P.cap = f->cap, P.novf = novfs[k];
        P.world_pos = mp->world_pos, P.normal = mp->normal, P.min_dist = mp->min_dist, P.max_dist = mp->max_dist;
        P.desc = mp->desc, P.skip = mp->skip, P.obs_pos = mp->obs_pos;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++)
                P.P.T[4 * r + c] = pr.Tcw[4 * r + c];
        minus_rt_t(pr.Tcw, P.P.Ow);
        P.P.fx = pr.fx, P.P.fy = pr.fy, P.P.cx = pr.cx, P.P.cy = pr.cy, P.P.mbf = pr.mbf;
        P.P.min_x = f->min_x, P.P.max_x = f->max_x, P.P.min_y = f->min_y, P.P.max_y = f->max_y;
        P.P.log_sf = pr.log_scale_factor, P.P.cos_limit = cos_limit, P.P.th = th;
        P.P.nlevels = f->nlevels;
        for (int l = 0; l < ORBGPU_MAX_LEVELS; l++)
            P.P.scale_factors[l] = l < f->nlevels ? f->scale_factors[l] : 0.f;
        P.F.n = f->cap;
        P.F.n_dev = f->n;
        P.F.kp_stride = (int)(sizeof(orbgpu_keypoint) / sizeof(float));
        P.F.kp_x = reinterpret_cast<const float *>(f->kps);
        P.F.kp_y = P.F.kp_x + 1;
        P.F.kp_octave = reinterpret_cast<const int *>(f->kps) + 5;
        P.F.u_right = f->u_right;
        P.F.desc = f->desc;
        P.F.min_x = f->min_x, P.F.min_y = f->min_y;
        P.F.inv_w = (float)GC / (f->max_x - f->min_x);
        P.F.inv_h = (float)GR / (f->max_y - f->min_y);
        P.F.cell_start = f->cell_start, P.F.cell_items = f->cell_items;
        P.F.inv_sigma2 = nullptr;
        P.q = ws->queries.as<Query>() + om;
        P.lists = ws->topk.as<uint32_t>() + om * PJ_LIST;
        P.match = ws->match.as<int>() + om;
        P.slow = ws->slow.as<int>() + om;
        P.claim_init = ws->claim_init.as<int>() + oc;
        P.sweeps = ws->sweeps.as<int>() + 2 * (size_t)k;
        P.kp_to_mp = pr.d_kp_to_mp;
        P.counts = pr.d_counts;
        P.nnratio = nnratio;
        if (pr.d_track)
            P.track = *pr.d_track;
        om += (size_t)std::max(mp->m, 1);
        oc += (size_t)f->cap;
    }
    ORBGPU_HIP_TRY(hipMemcpyAsync(ws->problems.p, hp.data(), sizeof(ProjProblem) * (size_t)n, hipMemcpyHostToDevice, st));
    if (!(ws->attr_set & 8u)) {
        ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve_batch),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)PJ_RESOLVE_MAX_LDS + 64));
        ws->attr_set |= 8u;
    }
    const ProjProblem *dp = ws->problems.as<ProjProblem>();
    hipLaunchKernelGGL(k_batch_zero_counts, dim3((n + 255) / 256), dim3(256), 0, st, dp, n);
    if (max_m > 0) {
        hipLaunchKernelGGL(k_frustum_queries_batch, dim3((max_cover + 255) / 256, n), dim3(256), 0, st, dp);
        hipLaunchKernelGGL(k_proj_lists_batch, dim3((max_m + PJ_ROWS_PER_BLOCK - 1) / PJ_ROWS_PER_BLOCK, n), dim3(256), 0, st, dp);
        hipLaunchKernelGGL(k_proj_resolve_batch, dim3(n), dim3(1024), lds, st, dp);
    }
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

int orbgpu_projection_last_sweeps(int32_t *sweeps, int32_t *rewalked_rows)
{
    // convergence diagnostics of the calling thread's most recent projection match (any of the four entry points)
    ORBGPU_REQUIRE(sweeps && rewalked_rows, "null argument");
    static thread_local int dummy;
    (void)dummy;
    ProjWorkspace *ws = nullptr;
    int dev = 0;
    ORBGPU_HIP_TRY(hipGetDevice(&dev));
    int rc = workspace(dev, &ws);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(ws->out.p, "no projection match recorded on this thread");
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    int h[2] = {0, 0};
    ORBGPU_HIP_TRY(hipMemcpy(h, ws->out.as<int>() + 1, sizeof(h), hipMemcpyDeviceToHost));
    *sweeps = h[0];
    *rewalked_rows = h[1];
    return ORBGPU_OK;
}

int orbgpu_search_by_projection_last(const orbgpu_frame_view *cur, const float *cur_Tcw, float fx, float fy,
                                     float cx, float cy, float mbf, float mb, const orbgpu_lastframe_view *last,
                                     float th, int32_t mono, int32_t check_orientation, int32_t *kp_to_mp,
                                     int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(last && cur_Tcw && kp_to_mp && nmatches, "null argument");
    int rc = validate_frame(cur);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(last->n >= 0 && last->Tcw, "bad last frame");
    if (last->n > 0)
        ORBGPU_REQUIRE(last->has_mp && last->world_pos && last->desc && last->kp_octave, "null last-frame arrays");
    ORBGPU_REQUIRE(!check_orientation || ((last->n == 0 || last->kp_angle) && (cur->n == 0 || cur->kp_angle)),
                   "orientation check needs angles");
    rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    // :1339-1349 forward / backward motion
    float twc[3], tlc[3];
    minus_rt_t(cur_Tcw, twc);
    rt_apply(last->Tcw, twc, tlc);
    const bool bForward = tlc[2] > mb && !mono;
    const bool bBackward = -tlc[2] > mb && !mono;
    std::vector<Query> q((size_t)last->n);
    for (int i = 0; i < last->n; i++) {
        Query &Q = q[i];
        Q = Query{};
        Q.blocking = last->obs_pos ? (last->obs_pos[i] != 0) : 1;
        if (!last->has_mp[i] || (last->outlier && last->outlier[i]))
            continue;
        // :1360-1376 projection (per-point float arithmetic of the boundary, O(n))
        float xc3[3];
        rt_apply(cur_Tcw, last->world_pos + 3 * (size_t)i, xc3);
        const float invzc = (float)(1.0 / (double)xc3[2]);
        if (invzc < 0)
            continue;
        volatile float ux = fx * xc3[0];
        volatile float ux2 = ux * invzc;
        const float u = ux2 + cx;
        volatile float vy = fy * xc3[1];
        volatile float vy2 = vy * invzc;
        const float v = vy2 + cy;
        if (u < cur->min_x || u > cur->max_x)
            continue;
        if (v < cur->min_y || v > cur->max_y)
            continue;
        const int oct = last->kp_octave[i];
        if (oct < 0 || oct >= cur->nlevels) {
            set_error("last-frame key point %d: octave %d outside [0,%d)", i, oct, cur->nlevels);
            return ORBGPU_ELEVEL;
        }
        Q.r = th * cur->scale_factors[oct];
        Q.x = u;
        Q.y = v;
        volatile float bz = mbf * invzc;
        Q.ur = u - bz;
        if (bForward) {
            Q.min_level = oct;
            Q.max_level = -1;
        } else if (bBackward) {
            Q.min_level = 0;
            Q.max_level = oct;
        } else {
            Q.min_level = oct - 1;
            Q.max_level = oct + 1;
        }
        Q.check_ur = 1;
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(cur->n, 1));
    for (int j = 0; j < cur->n; j++) {
        const int v = kp_to_mp[j];
        ORBGPU_REQUIRE(v >= -2 && v < last->n, "kp_to_mp[%d] = %d out of range", j, v);
        const bool held = v == -2 || (v >= 0 && (last->obs_pos ? last->obs_pos[v] != 0 : true));
        init[j] = held ? -1 : INT_MAX;
    }
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, cur, F)) != ORBGPU_OK)
        return rc;
    return run_projection<1>(*ws, F, q, last->desc, last->kp_angle, cur->kp_angle, init, 0.f, ORBGPU_TH_HIGH,
                             check_orientation, kp_to_mp, nmatches);
}

int orbgpu_search_by_projection_keyframe(const orbgpu_frame_view *cur, const float *cur_Tcw, float fx, float fy,
                                         float cx, float cy, float log_scale_factor,
                                         const orbgpu_keyframe_view *kf, float th, int32_t orb_dist,
                                         int32_t check_orientation, int32_t *kp_to_mp, int32_t *nmatches,
                                         int32_t device_id)
{
    ORBGPU_REQUIRE(kf && cur_Tcw && kp_to_mp && nmatches, "null argument");
    int rc = validate_frame(cur);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(kf->n >= 0, "bad key frame");
    if (kf->n > 0)
        ORBGPU_REQUIRE(kf->has_mp && kf->world_pos && kf->min_dist_inv && kf->max_dist_inv && kf->max_dist && kf->desc, "null key-frame arrays");
    ORBGPU_REQUIRE(!check_orientation || ((kf->n == 0 || kf->kp_angle) && (cur->n == 0 || cur->kp_angle)),
                   "orientation check needs angles");
    ORBGPU_REQUIRE(log_scale_factor > 0, "log_scale_factor must be positive");
    rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    float Ow[3];
    minus_rt_t(cur_Tcw, Ow);  // :1478
    std::vector<Query> q((size_t)kf->n);
    for (int i = 0; i < kf->n; i++) {
        Query &Q = q[i];
        Q = Query{};
        Q.blocking = 1;  // :1540-1541: any association hides the key point
        if (!kf->has_mp[i] || (kf->bad && kf->bad[i]) || (kf->already_found && kf->already_found[i]))
            continue;
        const float *Pw = kf->world_pos + 3 * (size_t)i;
        float xc3[3];
        rt_apply(cur_Tcw, Pw, xc3);
        const float invzc = (float)(1.0 / (double)xc3[2]);
        volatile float ux = fx * xc3[0];
        volatile float ux2 = ux * invzc;
        const float u = ux2 + cx;
        volatile float vy = fy * xc3[1];
        volatile float vy2 = vy * invzc;
        const float v = vy2 + cy;
        if (u < cur->min_x || u > cur->max_x)
            continue;
        if (v < cur->min_y || v > cur->max_y)
            continue;
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        // cv::norm accumulates in double
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        const float maxDistance = kf->max_dist_inv[i], minDistance = kf->min_dist_inv[i];
        if (dist3D < minDistance || dist3D > maxDistance)
            continue;
        const float ratio = kf->max_dist[i] / dist3D;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);  // MapPoint::PredictScale
        if (lvl < 0 || lvl >= cur->nlevels) {
            set_error("key-frame map point %d: predicted level %d outside [0,%d)", i, lvl, cur->nlevels);
            return ORBGPU_ELEVEL;
        }
        Q.r = th * cur->scale_factors[lvl];
        Q.x = u;
        Q.y = v;
        Q.min_level = lvl - 1;
        Q.max_level = lvl + 1;
        Q.check_ur = 0;
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(cur->n, 1));
    for (int j = 0; j < cur->n; j++)
        init[j] = kp_to_mp[j] == -1 ? INT_MAX : -1;
    // the matcher writes indices of `kf` rows; occupied key points keep their value
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, cur, F)) != ORBGPU_OK)
        return rc;
    return run_projection<1>(*ws, F, q, kf->desc, kf->kp_angle, cur->kp_angle, init, 0.f, orb_dist, check_orientation,
                             kp_to_mp, nmatches);
}

int orbgpu_search_by_projection_sim3(const orbgpu_frame_view *kf, const float *Scw, float fx, float fy, float cx,
                                     float cy, float log_scale_factor, const orbgpu_points_view *pts, int32_t th,
                                     int32_t *kp_to_mp, int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(pts && Scw && kp_to_mp && nmatches, "null argument");
    int rc = validate_frame(kf);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(pts->m >= 0, "bad point count");
    if (pts->m > 0)
        ORBGPU_REQUIRE(pts->world_pos && pts->normal && pts->min_dist && pts->max_dist && pts->desc, "null point arrays");
    ORBGPU_REQUIRE(log_scale_factor > 0, "log_scale_factor must be positive");
    rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    // Scw = [s R | s t] decomposed as ORBmatcher.cc:299-303 does: scw from the first row (double dot), then
    // cv::Mat / scalar (float multiply by (float)(1/scw)), Ow = -Rcw^T tcw
    float T[16] = {0}, Ow[3];
    {
        const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
        const float scw = (float)sqrt(d);
        ORBGPU_REQUIRE(scw > 0.f, "degenerate Scw");
        const float alpha = (float)(1.0 / (double)scw);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                volatile float v = Scw[4 * r + c] * alpha;
                T[4 * r + c] = v;
            }
        minus_rt_t(T, Ow);
    }
    // spAlreadyFound (:306-307): points some key point of the key frame already holds
    std::vector<uint8_t> found((size_t)std::max(pts->m, 1), 0);
    for (int j = 0; j < kf->n; j++) {
        const int v = kp_to_mp[j];
        ORBGPU_REQUIRE(v >= -2 && v < pts->m, "kp_to_mp[%d] = %d out of range", j, v);
        if (v >= 0)
            found[v] = 1;
    }
    std::vector<Query> q((size_t)pts->m);
    for (int i = 0; i < pts->m; i++) {
        Query &Q = q[i];
        Q = Query{};
        Q.blocking = 1;  // :394 vpMatched[bestIdx] = pMP hides the key point from every later point
        if ((pts->bad && pts->bad[i]) || found[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(T, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = 1 / pc[2];
        volatile float x = pc[0] * invz, y = pc[1] * invz;
        volatile float ux = fx * x, vy = fy * y;
        const float u = ux + cx, v = vy + cy;
        if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y))  // KeyFrame::IsInImage
            continue;
        const float maxDistance = 1.2f * pts->max_dist[i], minDistance = 0.8f * pts->min_dist[i];
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist < minDistance || dist > maxDistance)
            continue;
        const float *Pn = pts->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
        if (dot < 0.5 * dist)  // viewing angle below 60 degrees (:352)
            continue;
        const float ratio = pts->max_dist[i] / dist;
        const int lvl = (int)ceilf(logf(ratio) / log_scale_factor);  // MapPoint::PredictScale
        if (lvl < 0 || lvl >= kf->nlevels) {
            set_error("point %d: predicted level %d outside [0,%d)", i, lvl, kf->nlevels);
            return ORBGPU_ELEVEL;
        }
        Q.r = (float)th * kf->scale_factors[lvl];
        Q.x = u;
        Q.y = v;
        Q.min_level = lvl - 1;
        Q.max_level = lvl;
        Q.check_ur = 0;
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(kf->n, 1));
    for (int j = 0; j < kf->n; j++)
        init[j] = kp_to_mp[j] == -1 ? INT_MAX : -1;  // :373 vpMatched[idx] set: skipped
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, kf, F)) != ORBGPU_OK)
        return rc;
    return run_projection<1>(*ws, F, q, pts->desc, nullptr, nullptr, init, 0.f, ORBGPU_TH_LOW, 0, kp_to_mp, nmatches);
}

// ---- claim-free "best key point in the window" matchers: Fuse, Fuse(Sim3), SearchBySim3 ---------------------
// Host side = the O(m) projection of the boundary in the reference's float conventions; device side = the same
// window walk / ranking as the projection matchers with rows that never block, so every row keeps the first
// candidate of least distance (`dist < bestDist`), subject to the distance threshold.
namespace {

struct PointGate {  // outcome of the per-point tests up to the window search
    bool ok = false;
    float u = 0, v = 0, ur = 0;
    int lvl = 0;
};

// common tail of the per-point tests: image bounds (KeyFrame::IsInImage), scale-invariance range, viewing angle
// (optional), PredictScale.  Returns ORBGPU_ELEVEL through rc when the level is out of range.
static PointGate point_gate(const float pc[3], float invz, float fx, float fy, float cx, float cy, float bf,
                            const orbgpu_frame_view *kf, float dist3D, const float *PO, const float *Pn, float min_dist,
                            float max_dist, float log_sf, int &rc, int i)
{
    PointGate g;
    volatile float x = pc[0] * invz, y = pc[1] * invz;
    volatile float ux = fx * x, vy = fy * y;
    const float u = ux + cx, v = vy + cy;
    if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y))
        return g;
    const float maxDistance = 1.2f * max_dist, minDistance = 0.8f * min_dist;
    if (dist3D < minDistance || dist3D > maxDistance)
        return g;
    if (Pn) {
        const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
        if (dot < 0.5 * dist3D)
            return g;
    }
    const float ratio = max_dist / dist3D;
    const int lvl = (int)ceilf(logf(ratio) / log_sf);  // MapPoint::PredictScale
    if (lvl < 0 || lvl >= kf->nlevels) {
        set_error("point %d: predicted level %d outside [0,%d)", i, lvl, kf->nlevels);
        rc = ORBGPU_ELEVEL;
        return g;
    }
    volatile float bz = bf * invz;
    g.ok = true;
    g.u = u, g.v = v, g.ur = u - bz, g.lvl = lvl;
    return g;
}

static int validate_points(const orbgpu_points_view *pts, bool need_normal)
{
    ORBGPU_REQUIRE(pts && pts->m >= 0, "bad point view");
    if (pts->m > 0)
        ORBGPU_REQUIRE(pts->world_pos && pts->min_dist && pts->max_dist && pts->desc && (!need_normal || pts->normal),
                       "null point arrays");
    return ORBGPU_OK;
}

// queries -> best key point per row (first of least distance, <= th_dist), -1 otherwise
static int best_rows(const orbgpu_frame_view *kf, std::vector<Query> &q, const uint8_t *row_desc, int th_dist,
                     const float *inv_sigma2, int32_t *best_idx, int32_t device_id)
{
    ProjWorkspace *ws = nullptr;
    int rc = workspace(device_id, &ws);
    if (rc != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, kf, F)) != ORBGPU_OK)
        return rc;
    if (inv_sigma2) {
        PJ_TRY(put(ws->inv_sigma2, inv_sigma2, sizeof(float) * kf->nlevels, ws->stream));
        F.inv_sigma2 = ws->inv_sigma2.as<float>();
    }
    std::vector<int> init((size_t)std::max(kf->n, 1), INT_MAX);  // nothing is claimed, nobody blocks
    std::vector<int32_t> k2m((size_t)std::max(kf->n, 1), -1);
    int32_t nm = 0;
    return run_projection<1>(*ws, F, q, row_desc, nullptr, nullptr, init, 0.f, th_dist, 0, k2m.data(), &nm, best_idx);
}

} // namespace

int orbgpu_fuse(const orbgpu_frame_view *kf, const float *Tcw, float fx, float fy, float cx, float cy, float bf,
                float log_scale_factor, const orbgpu_points_view *pts, float th, const float *inv_level_sigma2,
                int32_t *best_idx, int32_t *n_candidates, int32_t device_id)
{
    ORBGPU_REQUIRE(Tcw && best_idx && n_candidates && inv_level_sigma2, "null argument");
    int rc = validate_frame(kf);
    if (rc != ORBGPU_OK || (rc = validate_points(pts, true)) != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(log_scale_factor > 0, "log_scale_factor must be positive");
    if ((rc = select_device(device_id)) != ORBGPU_OK)
        return rc;
    float Ow[3];
    minus_rt_t(Tcw, Ow);
    std::vector<Query> q((size_t)pts->m);
    for (int i = 0; i < pts->m; i++) {
        Query &Q = q[i];
        Q = Query{};
        if (pts->bad && pts->bad[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(Tcw, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = 1 / pc[2];  // :857 (float division)
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        const PointGate g = point_gate(pc, invz, fx, fy, cx, cy, bf, kf, dist3D, PO, pts->normal + 3 * (size_t)i,
                                       pts->min_dist[i], pts->max_dist[i], log_scale_factor, rc, i);
        if (rc != ORBGPU_OK)
            return rc;
        if (!g.ok)
            continue;
        Q.r = th * kf->scale_factors[g.lvl];
        Q.x = g.u, Q.y = g.v, Q.ur = g.ur;
        Q.min_level = g.lvl - 1, Q.max_level = g.lvl;
        Q.gate = 1;
        Q.active = 1;
    }
    if ((rc = best_rows(kf, q, pts->desc, ORBGPU_TH_LOW, inv_level_sigma2, best_idx, device_id)) != ORBGPU_OK)
        return rc;
    int n = 0;
    for (int i = 0; i < pts->m; i++)
        n += best_idx[i] >= 0;
    *n_candidates = n;
    return ORBGPU_OK;
}

int orbgpu_fuse_sim3(const orbgpu_frame_view *kf, const float *Scw, float fx, float fy, float cx, float cy,
                     float log_scale_factor, const orbgpu_points_view *pts, float th, int32_t *best_idx,
                     int32_t *n_candidates, int32_t device_id)
{
    ORBGPU_REQUIRE(Scw && best_idx && n_candidates, "null argument");
    int rc = validate_frame(kf);
    if (rc != ORBGPU_OK || (rc = validate_points(pts, true)) != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(log_scale_factor > 0, "log_scale_factor must be positive");
    if ((rc = select_device(device_id)) != ORBGPU_OK)
        return rc;
    float T[16] = {0}, Ow[3];
    {  // :985-989, as orbgpu_search_by_projection_sim3
        const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
        const float scw = (float)sqrt(d);
        ORBGPU_REQUIRE(scw > 0.f, "degenerate Scw");
        const float alpha = (float)(1.0 / (double)scw);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) {
                volatile float v = Scw[4 * r + c] * alpha;
                T[4 * r + c] = v;
            }
        minus_rt_t(T, Ow);
    }
    std::vector<Query> q((size_t)pts->m);
    for (int i = 0; i < pts->m; i++) {
        Query &Q = q[i];
        Q = Query{};
        if (pts->bad && pts->bad[i])
            continue;
        const float *Pw = pts->world_pos + 3 * (size_t)i;
        float pc[3];
        rt_apply(T, Pw, pc);
        if (pc[2] < 0.0f)
            continue;
        const float invz = (float)(1.0 / (double)pc[2]);  // :1017
        const float PO[3] = {Pw[0] - Ow[0], Pw[1] - Ow[1], Pw[2] - Ow[2]};
        const float dist3D = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        const PointGate g = point_gate(pc, invz, fx, fy, cx, cy, 0.f, kf, dist3D, PO, pts->normal + 3 * (size_t)i,
                                       pts->min_dist[i], pts->max_dist[i], log_scale_factor, rc, i);
        if (rc != ORBGPU_OK)
            return rc;
        if (!g.ok)
            continue;
        Q.r = th * kf->scale_factors[g.lvl];
        Q.x = g.u, Q.y = g.v;
        Q.min_level = g.lvl - 1, Q.max_level = g.lvl;
        Q.active = 1;
    }
    if ((rc = best_rows(kf, q, pts->desc, ORBGPU_TH_LOW, nullptr, best_idx, device_id)) != ORBGPU_OK)
        return rc;
    int n = 0;
    for (int i = 0; i < pts->m; i++)
        n += best_idx[i] >= 0;
    *n_candidates = n;
    return ORBGPU_OK;
}

// one direction of SearchBySim3 (:1143-1227 / :1229-1307)
static int sim3_direction(const orbgpu_frame_view *kfB, const float *Taw, const float sR[9], const float t[3], float fx,
                          float fy, float cx, float cy, float log_sfB, const orbgpu_points_view *ptsA,
                          const uint8_t *skipA, float th, int32_t *match, int32_t device_id)
{
    int rc = ORBGPU_OK;
    std::vector<Query> q((size_t)ptsA->m);
    for (int i = 0; i < ptsA->m; i++) {
        Query &Q = q[i];
        Q = Query{};
        if ((ptsA->bad && ptsA->bad[i]) || (skipA && skipA[i]))
            continue;
        float pa[3], pb[3];
        rt_apply(Taw, ptsA->world_pos + 3 * (size_t)i, pa);
        for (int r = 0; r < 3; r++) {
            volatile float a = sR[3 * r] * pa[0], b = sR[3 * r + 1] * pa[1], c = sR[3 * r + 2] * pa[2];
            volatile float t0 = a + b;
            volatile float t1 = t0 + c;
            pb[r] = t1 + t[r];
        }
        if ((double)pb[2] < 0.0)
            continue;
        const float invz = (float)(1.0 / (double)pb[2]);
        const float dist3D = (float)sqrt((double)pb[0] * pb[0] + (double)pb[1] * pb[1] + (double)pb[2] * pb[2]);
        const PointGate g = point_gate(pb, invz, fx, fy, cx, cy, 0.f, kfB, dist3D, nullptr, nullptr, ptsA->min_dist[i],
                                       ptsA->max_dist[i], log_sfB, rc, i);
        if (rc != ORBGPU_OK)
            return rc;
        if (!g.ok)
            continue;
        Q.r = th * kfB->scale_factors[g.lvl];
        Q.x = g.u, Q.y = g.v;
        Q.min_level = g.lvl - 1, Q.max_level = g.lvl;
        Q.active = 1;
    }
    return best_rows(kfB, q, ptsA->desc, ORBGPU_TH_HIGH, nullptr, match, device_id);
}

int orbgpu_search_by_sim3(const orbgpu_frame_view *kf1, const orbgpu_frame_view *kf2, const float *T1w, const float *T2w,
                          float s12, const float *R12, const float *t12, float fx, float fy, float cx, float cy,
                          float log_sf1, float log_sf2, const orbgpu_points_view *pts1, const uint8_t *already1,
                          const orbgpu_points_view *pts2, const uint8_t *already2, float th, int32_t *match12,
                          int32_t *nfound, int32_t device_id)
{
    ORBGPU_REQUIRE(T1w && T2w && R12 && t12 && match12 && nfound, "null argument");
    int rc = validate_frame(kf1);
    if (rc != ORBGPU_OK || (rc = validate_frame(kf2)) != ORBGPU_OK || (rc = validate_points(pts1, false)) != ORBGPU_OK ||
        (rc = validate_points(pts2, false)) != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(pts1->m == kf1->n && pts2->m == kf2->n, "points views must have one row per key point");
    ORBGPU_REQUIRE(s12 > 0 && log_sf1 > 0 && log_sf2 > 0, "bad scale");
    if ((rc = select_device(device_id)) != ORBGPU_OK)
        return rc;
    float sR12[9], sR21[9], t21[3];
    const double inv_s = 1.0 / (double)s12;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            volatile float a = s12 * R12[3 * r + c];  // :1121
            sR12[3 * r + c] = a;
            sR21[3 * r + c] = (float)((double)R12[3 * c + r] * inv_s);  // :1122 (1.0/s12)*R12.t()
        }
    for (int r = 0; r < 3; r++) {  // :1123 t21 = -sR21*t12
        volatile float a = sR21[3 * r] * t12[0], b = sR21[3 * r + 1] * t12[1], c = sR21[3 * r + 2] * t12[2];
        volatile float t0 = a + b;
        volatile float t1 = t0 + c;
        t21[r] = -t1;
    }
    std::vector<int32_t> m1((size_t)std::max(pts1->m, 1), -1), m2((size_t)std::max(pts2->m, 1), -1);
    if ((rc = sim3_direction(kf2, T1w, sR21, t21, fx, fy, cx, cy, log_sf2, pts1, already1, th, m1.data(), device_id)) != ORBGPU_OK)
        return rc;
    if ((rc = sim3_direction(kf1, T2w, sR12, t12, fx, fy, cx, cy, log_sf1, pts2, already2, th, m2.data(), device_id)) != ORBGPU_OK)
        return rc;
    int n = 0;
    for (int i1 = 0; i1 < pts1->m; i1++) {  // :1309-1323 agreement of the two directions
        const int idx2 = m1[i1];
        match12[i1] = (idx2 >= 0 && m2[idx2] == i1) ? idx2 : -1;
        n += match12[i1] >= 0;
    }
    *nfound = n;
    return ORBGPU_OK;
}

int orbgpu_search_for_triangulation(const orbgpu_frame_view *kf1, const uint8_t *has_mp1, const int32_t *node1,
                                    const orbgpu_frame_view *kf2, const uint8_t *has_mp2, const int32_t *node2,
                                    const float *F12, float ex, float ey, const float *level_sigma2_2,
                                    int32_t only_stereo, int32_t check_orientation, int32_t *match12,
                                    int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(kf1 && kf2 && F12 && level_sigma2_2 && match12 && nmatches, "null argument");
    ORBGPU_REQUIRE(kf1->n >= 0 && kf1->n <= 65535 && kf2->n >= 0 && kf2->n <= 65535, "key point count out of range");
    ORBGPU_REQUIRE(kf2->nlevels >= 1 && kf2->nlevels <= ORBGPU_MAX_LEVELS && kf2->scale_factors, "bad scale factors");
    if (kf1->n > 0)
        ORBGPU_REQUIRE(kf1->kp_x && kf1->kp_y && kf1->u_right && kf1->desc && has_mp1 && node1, "null key-frame 1 arrays");
    if (kf2->n > 0)
        ORBGPU_REQUIRE(kf2->kp_x && kf2->kp_y && kf2->kp_octave && kf2->u_right && kf2->desc && has_mp2 && node2,
                       "null key-frame 2 arrays");
    ORBGPU_REQUIRE(!check_orientation || ((kf1->n == 0 || kf1->kp_angle) && (kf2->n == 0 || kf2->kp_angle)),
                   "orientation check needs angles");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    *nmatches = 0;
    const int n1 = kf1->n, n2 = kf2->n;
    if (n1 == 0)
        return ORBGPU_OK;
    if (n2 == 0) {
        for (int i = 0; i < n1; i++)
            match12[i] = -1;
        return ORBGPU_OK;
    }
    for (int j = 0; j < n2; j++)
        ORBGPU_REQUIRE(kf2->kp_octave[j] >= 0 && kf2->kp_octave[j] < kf2->nlevels, "key point %d: octave out of range", j);
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    hipStream_t st = ws->stream;
    // one staging buffer: descriptors (32-byte rows) first, then the 4-byte arrays, the flags, the outputs
    const size_t w1 = (size_t)n1, w2 = (size_t)n2;
    const size_t off_desc1 = 0, off_desc2 = 32 * w1, off_arr = 32 * (w1 + w2), off_f1 = off_arr + 4 * (5 * w1 + 6 * w2),
                 off_f2 = off_f1 + w1, off_out = ((off_f2 + w2 + 15) / 16) * 16, total = off_out + 4 * w1 + 16;
    PJ_TRY(ws->tri.reserve(total));
    uint8_t *base = ws->tri.as<uint8_t>();
    auto up = [&](size_t off, const void *src, size_t bytes) -> int {
        ORBGPU_HIP_TRY(hipMemcpyAsync(base + off, src, bytes, hipMemcpyHostToDevice, st));
        return ORBGPU_OK;
    };
    std::vector<float> zeros(std::max(w1, w2), 0.f);
    size_t o = off_arr;
    const size_t o_x1 = o; PJ_TRY(up(o, kf1->kp_x, 4 * w1)); o += 4 * w1;
    const size_t o_y1 = o; PJ_TRY(up(o, kf1->kp_y, 4 * w1)); o += 4 * w1;
    const size_t o_u1 = o; PJ_TRY(up(o, kf1->u_right, 4 * w1)); o += 4 * w1;
    const size_t o_n1 = o; PJ_TRY(up(o, node1, 4 * w1)); o += 4 * w1;
    const size_t o_a1 = o; PJ_TRY(up(o, check_orientation ? kf1->kp_angle : zeros.data(), 4 * w1)); o += 4 * w1;
    const size_t o_x2 = o; PJ_TRY(up(o, kf2->kp_x, 4 * w2)); o += 4 * w2;
    const size_t o_y2 = o; PJ_TRY(up(o, kf2->kp_y, 4 * w2)); o += 4 * w2;
    const size_t o_u2 = o; PJ_TRY(up(o, kf2->u_right, 4 * w2)); o += 4 * w2;
    const size_t o_n2 = o; PJ_TRY(up(o, node2, 4 * w2)); o += 4 * w2;
    const size_t o_o2 = o; PJ_TRY(up(o, kf2->kp_octave, 4 * w2)); o += 4 * w2;
    const size_t o_a2 = o; PJ_TRY(up(o, check_orientation ? kf2->kp_angle : zeros.data(), 4 * w2)); o += 4 * w2;
    PJ_TRY(up(off_desc1, kf1->desc, 32 * w1));
    PJ_TRY(up(off_desc2, kf2->desc, 32 * w2));
    PJ_TRY(up(off_f1, has_mp1, w1));
    PJ_TRY(up(off_f2, has_mp2, w2));
    TriParams P;
    for (int k = 0; k < 9; k++)
        P.F12[k] = F12[k];
    P.ex = ex, P.ey = ey, P.only_stereo = only_stereo ? 1 : 0;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) {
        P.sigma2[l] = l < kf2->nlevels ? level_sigma2_2[l] : 0.f;
        P.scale[l] = l < kf2->nlevels ? kf2->scale_factors[l] : 0.f;
    }
    int *d_match = reinterpret_cast<int *>(base + off_out), *d_n = d_match + n1;
#define TRI_F(off) reinterpret_cast<const float *>(base + (off))
#define TRI_I(off) reinterpret_cast<const int *>(base + (off))
    hipLaunchKernelGGL(k_tri_match, dim3((n1 + 3) / 4), dim3(256), 0, st, n1, TRI_F(o_x1), TRI_F(o_y1), TRI_F(o_u1),
                       base + off_f1, TRI_I(o_n1), base + off_desc1, n2, TRI_F(o_x2), TRI_F(o_y2), TRI_I(o_o2),
                       TRI_F(o_u2), base + off_f2, TRI_I(o_n2), base + off_desc2, P, d_match);
    hipLaunchKernelGGL(k_tri_finish, dim3(1), dim3(1024), 0, st, n1, TRI_F(o_a1), TRI_F(o_a2), check_orientation ? 1 : 0,
                       d_match, d_n);
#undef TRI_F
#undef TRI_I
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpyAsync(match12, d_match, 4 * w1, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(nmatches, d_n, 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

int orbgpu_search_for_initialization(const orbgpu_frame_view *f1, const orbgpu_frame_view *f2, float *prev_matched,
                                     int32_t window_size, float nnratio, int32_t check_orientation,
                                     int32_t *matches12, int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(f1 && prev_matched && matches12 && nmatches && window_size > 0, "bad arguments");
    ORBGPU_REQUIRE(f1->n >= 0 && (f1->n == 0 || (f1->kp_octave && f1->desc)), "null frame-1 arrays");
    ORBGPU_REQUIRE(!check_orientation || f1->n == 0 || f1->kp_angle, "orientation check needs angles");
    int rc = validate_frame(f2);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(!check_orientation || f2->n == 0 || f2->kp_angle, "orientation check needs angles");
    if ((rc = select_device(device_id)) != ORBGPU_OK)
        return rc;
    const int n1 = f1->n, n2 = f2->n;
    *nmatches = 0;
    for (int i = 0; i < n1; i++)
        matches12[i] = -1;
    if (n1 == 0 || n2 == 0)
        return ORBGPU_OK;
    // rows = the level-0 key points of F1 (:419-422)
    std::vector<int> row_of;
    std::vector<Query> q;
    std::vector<uint8_t> rdesc;
    for (int i1 = 0; i1 < n1; i1++) {
        if (f1->kp_octave[i1] > 0)
            continue;
        Query Q{};
        Q.x = prev_matched[2 * i1], Q.y = prev_matched[2 * i1 + 1];
        Q.r = (float)window_size;
        Q.min_level = f1->kp_octave[i1], Q.max_level = f1->kp_octave[i1];  // GetFeaturesInArea(.., level1, level1)
        Q.active = 1;
        row_of.push_back(i1);
        q.push_back(Q);
        rdesc.insert(rdesc.end(), f1->desc + (size_t)i1 * 32, f1->desc + (size_t)i1 * 32 + 32);
    }
    const int rows = (int)row_of.size();
    if (rows == 0)
        return ORBGPU_OK;
    ProjWorkspace *ws = nullptr;
    if ((rc = workspace(device_id, &ws)) != ORBGPU_OK)
        return rc;
    FrameDev F;
    if ((rc = upload_frame(*ws, f2, F)) != ORBGPU_OK)
        return rc;
    hipStream_t st = ws->stream;
    const int cap = n2;
    PJ_TRY(put(ws->queries, q.data(), sizeof(Query) * rows, st));
    PJ_TRY(put(ws->row_desc, rdesc.data(), (size_t)rows * 32, st));
    PJ_TRY(ws->tri.reserve(sizeof(uint64_t) * (size_t)rows * cap + sizeof(int) * (size_t)rows + 64));
    uint64_t *d_keys = ws->tri.as<uint64_t>();
    int *d_counts = reinterpret_cast<int *>(d_keys + (size_t)rows * cap);
    hipLaunchKernelGGL(k_window_candidates, dim3((rows + 3) / 4), dim3(256), 0, st, rows, ws->queries.as<Query>(),
                       ws->row_desc.as<uint8_t>(), F, cap, d_keys, d_counts);
    ORBGPU_HIP_TRY(hipGetLastError());
    std::vector<int> counts((size_t)rows);
    ORBGPU_HIP_TRY(hipMemcpyAsync(counts.data(), d_counts, sizeof(int) * (size_t)rows, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    std::vector<size_t> off((size_t)rows + 1, 0);
    for (int r = 0; r < rows; r++) {
        ORBGPU_REQUIRE(counts[r] <= cap, "window candidate list overflow");  // cannot happen: cap = all key points
        off[r + 1] = off[r] + (size_t)counts[r];
    }
    std::vector<uint64_t> keys(std::max<size_t>(off[rows], 1));
    for (int r = 0; r < rows; r++)
        if (counts[r])
            ORBGPU_HIP_TRY(hipMemcpyAsync(&keys[off[r]], d_keys + (size_t)r * cap, sizeof(uint64_t) * (size_t)counts[r],
                                          hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    // ---- the sequential part of the reference, verbatim, over the device-computed distances (:414-517)
    std::vector<int> vMatchedDistance((size_t)n2, INT_MAX), vnMatches21((size_t)n2, -1);
    std::vector<std::pair<int, int>> pushes;  // (bin, i1) in push order
    int histo[ORBGPU_HISTO_LENGTH] = {0};
    int nm = 0;
    const float factor = 1.0f / ORBGPU_HISTO_LENGTH;
    for (int r = 0; r < rows; r++) {
        const int i1 = row_of[r];
        uint64_t *b = &keys[off[r]], *e = b + counts[r];
        if (b == e)
            continue;  // :426-427
        std::sort(b, e, [](uint64_t x, uint64_t y) { return ((x >> 20) & 0xFFFFFFull) < ((y >> 20) & 0xFFFFFFull); });
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (uint64_t *p = b; p != e; ++p) {
            const int i2 = (int)(*p & 0xFFFFF), dist = (int)(*p >> 44);
            if (vMatchedDistance[i2] <= dist)
                continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= ORBGPU_TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
            if (vnMatches21[bestIdx2] >= 0) {
                matches12[vnMatches21[bestIdx2]] = -1;
                nm--;
            }
            matches12[i1] = bestIdx2;
            vnMatches21[bestIdx2] = i1;
            vMatchedDistance[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) {
                float rot = f1->kp_angle[i1] - f2->kp_angle[bestIdx2];
                if (rot < 0.0)
                    rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == ORBGPU_HISTO_LENGTH)
                    bin = 0;
                pushes.emplace_back(bin, i1);
                histo[bin]++;
            }
        }
    }
    if (check_orientation) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;  // ComputeThreeMaxima, :1601-1642
        for (int i = 0; i < ORBGPU_HISTO_LENGTH; i++) {
            const int sz = histo[i];
            if (sz > max1) {
                max3 = max2, max2 = max1, max1 = sz;
                ind3 = ind2, ind2 = ind1, ind1 = i;
            } else if (sz > max2) {
                max3 = max2, max2 = sz;
                ind3 = ind2, ind2 = i;
            } else if (sz > max3) {
                max3 = sz, ind3 = i;
            }
        }
        if ((float)max2 < 0.1f * (float)max1)
            ind2 = ind3 = -1;
        else if ((float)max3 < 0.1f * (float)max1)
            ind3 = -1;
        for (const auto &pr : pushes) {
            if (pr.first == ind1 || pr.first == ind2 || pr.first == ind3)
                continue;
            if (matches12[pr.second] >= 0) {
                matches12[pr.second] = -1;
                nm--;
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)  // :514-517
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = f2->kp_x[matches12[i1]];
            prev_matched[2 * i1 + 1] = f2->kp_y[matches12[i1]];
        }
    *nmatches = nm;
    return ORBGPU_OK;
}

} // extern "C"
