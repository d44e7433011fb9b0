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
// One wave per row: lanes take the grid cells of the search window (ix outer, iy inner -- the
// reference's candidate order, which decides ties), each lane keeps its two best candidates as
// 64-bit keys (distance, order, index) and a wave butterfly merges them.
#include "common.h"
#include "matcher_common.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

namespace orbgpu {

constexpr int PJ_MAX_SWEEPS = 16;
constexpr int GC = ORBGPU_GRID_COLS, GR = ORBGPU_GRID_ROWS;

struct Query {  // one row of the matcher: a projected map point
    float x, y, r;      // window centre and half-size (r already multiplied by the level scale)
    float ur;           // predicted right coordinate (mTrackProjXR / u - mbf*invz)
    int min_level, max_level;
    int active;         // 0: the reference `continue`s before the candidate loop
    int blocking;       // MapPoint::Observations() > 0
};

struct FrameDev {
    int n;
    const float *kp_x, *kp_y;
    const int *kp_octave;
    const float *u_right;
    const uint8_t *desc;
    float min_x, min_y, inv_w, inv_h;
    const int *cell_start, *cell_items;
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

// mode 0: SearchByProjection(F, MapPoints)  -- best + second, ratio test only on equal levels
// mode 1: SearchByProjection(Cur, Last)     -- best only
template <int MODE>
__global__ __launch_bounds__(256) void k_proj_sweep(int sweep, int m, const Query *__restrict__ q,
                                                    const uint8_t *__restrict__ row_desc, FrameDev F,
                                                    float nnratio, const int *__restrict__ claim_init,
                                                    int *__restrict__ match, int *__restrict__ claim3,
                                                    int *__restrict__ changed)
{
    if (sweep > 0 && changed[sweep - 1] == 0)
        return;
    const int n = F.n;
    int *c_rd = claim3 + (size_t)(sweep % 3) * n;
    int *c_wr = claim3 + (size_t)((sweep + 1) % 3) * n;
    int *c_cl = claim3 + (size_t)((sweep + 2) % 3) * n;
    // reset the table sweep+1 will write to its initial state (pre-existing associations)
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256)
        c_cl[j] = claim_init[j];

    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m)
        return;
    const Query Q = q[i];
    int result = -1;
    if (Q.active) {
        // Frame::GetFeaturesInArea, Frame.cc:332-346
        const int c0 = (int)floorf((Q.x - F.min_x - Q.r) * F.inv_w);
        const int c1 = (int)ceilf((Q.x - F.min_x + Q.r) * F.inv_w);
        const int r0 = (int)floorf((Q.y - F.min_y - Q.r) * F.inv_h);
        const int r1 = (int)ceilf((Q.y - F.min_y + Q.r) * F.inv_h);
        const int minCx = max(0, c0), maxCx = min(GC - 1, c1);
        const int minCy = max(0, r0), maxCy = min(GR - 1, r1);
        if (!(minCx >= GC || maxCx < 0 || minCy >= GR || maxCy < 0)) {
            const int ny = maxCy - minCy + 1;
            const int ncell = (maxCx - minCx + 1) * ny;
            const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
            uint64_t a[4];
            const uint64_t *da = reinterpret_cast<const uint64_t *>(row_desc) + (size_t)i * 4;
#pragma unroll
            for (int w = 0; w < 4; w++)
                a[w] = da[w];
            uint64_t k1 = KEY_NONE, k2 = KEY_NONE;
            for (int seq = lane; seq < ncell; seq += 64) {
                const int ix = minCx + seq / ny, iy = minCy + seq % ny;
                const int cell = ix * GR + iy;
                const int beg = F.cell_start[cell], end = F.cell_start[cell + 1];
                for (int t = beg; t < end; t++) {
                    const int idx = F.cell_items[t];
                    if (check_levels) {
                        const int oct = F.kp_octave[idx];
                        if (oct < Q.min_level)
                            continue;
                        if (Q.max_level >= 0 && oct > Q.max_level)
                            continue;
                    }
                    const float dx = F.kp_x[idx] - Q.x, dy = F.kp_y[idx] - Q.y;
                    if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r))
                        continue;
                    if (c_rd[idx] < i)
                        continue;  // held by an earlier row / a pre-existing association
                    const float ur = F.u_right[idx];
                    if (ur > 0) {
                        const float er = fabsf(Q.ur - ur);
                        if (er > Q.r)
                            continue;
                    }
                    const uint64_t *db = reinterpret_cast<const uint64_t *>(F.desc) + (size_t)idx * 4;
                    uint64_t b[4] = {db[0], db[1], db[2], db[3]};
                    const uint64_t d = (uint64_t)hamming256(a, b);
                    const uint64_t key = (d << 44) | ((uint64_t)seq << 32) | ((uint64_t)(t - beg) << 20) | (uint64_t)idx;
                    keep2(k1, k2, key);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint64_t o1 = __shfl_xor(k1, off, 64), o2 = __shfl_xor(k2, off, 64);
                keep2(k1, k2, o1);
                keep2(k1, k2, o2);
            }
            const int bestDist = (int)(k1 >> 44);
            if (bestDist <= ORBGPU_TH_HIGH) {
                const int bestIdx = (int)(k1 & 0xFFFFF);
                bool accept = true;
                if (MODE == 0) {
                    const int bestDist2 = (int)(k2 >> 44);
                    const int bestLevel = F.kp_octave[bestIdx];
                    const int bestLevel2 = bestDist2 < 256 ? F.kp_octave[(int)(k2 & 0xFFFFF)] : -1;
                    if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
                        accept = false;  // ORBmatcher.cc:118-121
                }
                if (accept)
                    result = bestIdx;
            }
        }
    }
    if (lane == 0) {
        if (sweep == 0 || match[i] != result)
            atomicAdd(&changed[sweep], 1);
        match[i] = result;
        if (result >= 0 && Q.blocking)
            atomicMin(&c_wr[result], i);
    }
}

// NB on distance ties inside k_proj_sweep: the sequential loop keeps the first candidate with the
// smallest distance and, as second, the next one in (distance, visiting order) -- exactly the two
// smallest keys, because the visiting order (cell sequence, position in cell) is part of the key.

// Exact serial fallback (one wave), used only if PJ_MAX_SWEEPS sweeps did not converge.
template <int MODE>
__global__ __launch_bounds__(64) void k_proj_serial(int m, const Query *__restrict__ q,
                                                    const uint8_t *__restrict__ row_desc, FrameDev F,
                                                    float nnratio, const int *__restrict__ claim_init,
                                                    int *__restrict__ match, int *__restrict__ claim3,
                                                    int *__restrict__ changed)
{
    if (changed[PJ_MAX_SWEEPS - 1] == 0)
        return;
    const int lane = threadIdx.x;
    int *held = claim3;  // 1 = held
    for (int j = lane; j < F.n; j += 64)
        held[j] = claim_init[j] < 0 ? 1 : 0;
    __syncthreads();
    for (int i = 0; i < m; i++) {
        const Query Q = q[i];
        int result = -1;
        if (Q.active) {
            const int c0 = (int)floorf((Q.x - F.min_x - Q.r) * F.inv_w);
            const int c1 = (int)ceilf((Q.x - F.min_x + Q.r) * F.inv_w);
            const int r0 = (int)floorf((Q.y - F.min_y - Q.r) * F.inv_h);
            const int r1 = (int)ceilf((Q.y - F.min_y + Q.r) * F.inv_h);
            const int minCx = max(0, c0), maxCx = min(GC - 1, c1);
            const int minCy = max(0, r0), maxCy = min(GR - 1, r1);
            if (!(minCx >= GC || maxCx < 0 || minCy >= GR || maxCy < 0)) {
                const int ny = maxCy - minCy + 1;
                const int ncell = (maxCx - minCx + 1) * ny;
                const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
                uint64_t a[4];
                const uint64_t *da = reinterpret_cast<const uint64_t *>(row_desc) + (size_t)i * 4;
                for (int w = 0; w < 4; w++)
                    a[w] = da[w];
                uint64_t k1 = KEY_NONE, k2 = KEY_NONE;
                for (int seq = lane; seq < ncell; seq += 64) {
                    const int ix = minCx + seq / ny, iy = minCy + seq % ny;
                    const int cell = ix * GR + iy;
                    const int beg = F.cell_start[cell], end = F.cell_start[cell + 1];
                    for (int t = beg; t < end; t++) {
                        const int idx = F.cell_items[t];
                        if (check_levels) {
                            const int oct = F.kp_octave[idx];
                            if (oct < Q.min_level)
                                continue;
                            if (Q.max_level >= 0 && oct > Q.max_level)
                                continue;
                        }
                        const float dx = F.kp_x[idx] - Q.x, dy = F.kp_y[idx] - Q.y;
                        if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r))
                            continue;
                        if (held[idx])
                            continue;
                        const float ur = F.u_right[idx];
                        if (ur > 0 && fabsf(Q.ur - ur) > Q.r)
                            continue;
                        const uint64_t *db = reinterpret_cast<const uint64_t *>(F.desc) + (size_t)idx * 4;
                        uint64_t b[4] = {db[0], db[1], db[2], db[3]};
                        const uint64_t d = (uint64_t)hamming256(a, b);
                        keep2(k1, k2, (d << 44) | ((uint64_t)seq << 32) | ((uint64_t)(t - beg) << 20) | (uint64_t)idx);
                    }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const uint64_t o1 = __shfl_xor(k1, off, 64), o2 = __shfl_xor(k2, off, 64);
                    keep2(k1, k2, o1);
                    keep2(k1, k2, o2);
                }
                const int bestDist = (int)(k1 >> 44);
                if (bestDist <= ORBGPU_TH_HIGH) {
                    const int bestIdx = (int)(k1 & 0xFFFFF);
                    bool accept = true;
                    if (MODE == 0) {
                        const int bestDist2 = (int)(k2 >> 44);
                        const int bestLevel = F.kp_octave[bestIdx];
                        const int bestLevel2 = bestDist2 < 256 ? F.kp_octave[(int)(k2 & 0xFFFFF)] : -1;
                        if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)
                            accept = false;
                    }
                    if (accept)
                        result = bestIdx;
                }
            }
        }
        if (lane == 0) {
            match[i] = result;
            if (result >= 0 && Q.blocking)
                held[result] = 1;
        }
        __syncthreads();
    }
}

// Write F.mvpMapPoints: the LAST claimant of a key point wins (rows are visited in order and a
// non-blocking claimant can be overwritten, :123 / :1428); then the rotation consistency of
// :1448-1467: every accepted row whose bin is not among the three maxima clears its key point.
__global__ __launch_bounds__(256) void k_proj_finish(int m, int n, const int *__restrict__ match,
                                                     const float *__restrict__ row_angle,
                                                     const float *__restrict__ kp_angle, int check_orientation,
                                                     int *__restrict__ last_claim /*scratch [n]*/,
                                                     int *__restrict__ kp_to_mp, int *__restrict__ nmatches)
{
    __shared__ int histo[ORBGPU_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    if (threadIdx.x < ORBGPU_HISTO_LENGTH)
        histo[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_count = 0;
    for (int j = threadIdx.x; j < n; j += 256)
        last_claim[j] = -1;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < m; i += 256) {
        const int j = match[i];
        if (j < 0)
            continue;
        cnt++;
        atomicMax(&last_claim[j], i);
        if (check_orientation)
            atomicAdd(&histo[rot_bin(row_angle[i], kp_angle[j])], 1);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 256)
        if (last_claim[j] >= 0)
            kp_to_mp[j] = last_claim[j];  // any claim overrides a previous (non-blocking) association
    __syncthreads();
    if (check_orientation) {
        if (threadIdx.x == 0) {
            int i1, i2, i3;
            three_maxima(histo, ORBGPU_HISTO_LENGTH, i1, i2, i3);
            s_keep[0] = i1;
            s_keep[1] = i2;
            s_keep[2] = i3;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 256) {
            const int j = match[i];
            if (j < 0)
                continue;
            const int b = rot_bin(row_angle[i], kp_angle[j]);
            if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) {
                kp_to_mp[j] = -1;
                cnt--;
            }
        }
    }
    cnt = wave_reduce_add(cnt);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&s_count, cnt);
    __syncthreads();
    if (threadIdx.x == 0)
        *nmatches = s_count;
}

// ---- host helpers ---------------------------------------------------------------------------
struct Uploader {
    std::vector<DevBuf> bufs;
    int rc = ORBGPU_OK;
    ~Uploader()
    {
        for (auto &b : bufs)
            b.release();
    }
    void *put(const void *src, size_t bytes)
    {
        bufs.emplace_back();
        DevBuf &b = bufs.back();
        if (rc != ORBGPU_OK)
            return nullptr;
        rc = b.reserve(std::max<size_t>(bytes, 16));
        if (rc != ORBGPU_OK)
            return nullptr;
        if (src && bytes) {
            hipError_t e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                set_error("hipMemcpy H2D: %s", hipGetErrorString(e));
                rc = ORBGPU_EHIP;
            }
        }
        return b.p;
    }
    void *alloc(size_t bytes) { return put(nullptr, bytes); }
};

static int validate_frame(const orbgpu_frame_view *f)
{
    ORBGPU_REQUIRE(f, "null frame view");
    ORBGPU_REQUIRE(f->n >= 0 && f->n < (1 << 20), "frame key point count out of range");
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

static FrameDev upload_frame(Uploader &up, const orbgpu_frame_view *f)
{
    FrameDev F;
    const size_t n = (size_t)f->n;
    F.n = f->n;
    F.kp_x = (const float *)up.put(f->kp_x, n * 4);
    F.kp_y = (const float *)up.put(f->kp_y, n * 4);
    F.kp_octave = (const int *)up.put(f->kp_octave, n * 4);
    F.u_right = (const float *)up.put(f->u_right, n * 4);
    F.desc = (const uint8_t *)up.put(f->desc, n * 32);
    F.min_x = f->min_x;
    F.min_y = f->min_y;
    F.inv_w = f->grid_inv_w;
    F.inv_h = f->grid_inv_h;
    F.cell_start = (const int *)up.put(f->cell_start, (size_t)(GC * GR + 1) * 4);
    F.cell_items = (const int *)up.put(f->cell_items, std::max<size_t>((size_t)f->cell_start[GC * GR], 1) * 4);
    return F;
}

// Runs sweeps + fallback + finish for prepared queries. kp_to_mp (host, in/out).
template <int MODE>
static int run_projection(Uploader &up, const FrameDev &F, const std::vector<Query> &queries,
                          const uint8_t *row_desc_host, const float *row_angle_host, const float *kp_angle_host,
                          const std::vector<int> &claim_init, float nnratio, int check_orientation,
                          int32_t *kp_to_mp, int32_t *nmatches)
{
    const int m = (int)queries.size(), n = F.n;
    if (m == 0 || n == 0) {
        *nmatches = 0;
        return ORBGPU_OK;
    }
    Query *dq = (Query *)up.put(queries.data(), sizeof(Query) * m);
    uint8_t *ddesc = (uint8_t *)up.put(row_desc_host, (size_t)m * 32);
    int *dinit = (int *)up.put(claim_init.data(), sizeof(int) * n);
    int *dmatch = (int *)up.alloc(sizeof(int) * m);
    int *dclaim = (int *)up.alloc(sizeof(int) * 3 * (size_t)n);
    int *dchanged = (int *)up.alloc(sizeof(int) * (PJ_MAX_SWEEPS + 1));
    int *dk2m = (int *)up.put(kp_to_mp, sizeof(int) * n);
    int *dnm = (int *)up.alloc(sizeof(int));
    float *drang = nullptr, *dkang = nullptr;
    if (check_orientation) {
        drang = (float *)up.put(row_angle_host, sizeof(float) * m);
        dkang = (float *)up.put(kp_angle_host, sizeof(float) * n);
    }
    if (up.rc != ORBGPU_OK)
        return up.rc;
    // tables 0 and 1 start at the initial state
    ORBGPU_HIP_TRY(hipMemcpy(dclaim, dinit, sizeof(int) * n, hipMemcpyDeviceToDevice));
    ORBGPU_HIP_TRY(hipMemcpy(dclaim + n, dinit, sizeof(int) * n, hipMemcpyDeviceToDevice));
    ORBGPU_HIP_TRY(hipMemset(dchanged, 0, sizeof(int) * (PJ_MAX_SWEEPS + 1)));
    const dim3 grid((m + 3) / 4);
    for (int s = 0; s < PJ_MAX_SWEEPS; s++)
        hipLaunchKernelGGL(k_proj_sweep<MODE>, grid, dim3(256), 0, 0, s, m, dq, ddesc, F, nnratio, dinit, dmatch, dclaim,
                           dchanged);
    hipLaunchKernelGGL(k_proj_serial<MODE>, dim3(1), dim3(64), 0, 0, m, dq, ddesc, F, nnratio, dinit, dmatch, dclaim,
                       dchanged);
    hipLaunchKernelGGL(k_proj_finish, dim3(1), dim3(256), 0, 0, m, n, dmatch, drang, dkang, check_orientation,
                       dclaim + n, dk2m, dnm);
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    ORBGPU_HIP_TRY(hipMemcpy(kp_to_mp, dk2m, sizeof(int) * n, hipMemcpyDeviceToHost));
    ORBGPU_HIP_TRY(hipMemcpy(nmatches, dnm, sizeof(int), hipMemcpyDeviceToHost));
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
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(f->n, 1));
    for (int j = 0; j < f->n; j++) {
        const int v = kp_to_mp[j];
        ORBGPU_REQUIRE(v >= -2 && v < mp->m, "kp_to_mp[%d] = %d out of range", j, v);
        const bool held = v == -2 || (v >= 0 && (mp->obs_pos ? mp->obs_pos[v] != 0 : true));
        init[j] = held ? -1 : INT_MAX;
    }
    Uploader up;
    FrameDev F = upload_frame(up, f);
    return run_projection<0>(up, F, q, mp->desc, nullptr, nullptr, init, nnratio, 0, kp_to_mp, nmatches);
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
        Q.active = 1;
    }
    std::vector<int> init((size_t)std::max(cur->n, 1));
    for (int j = 0; j < cur->n; j++) {
        const int v = kp_to_mp[j];
        ORBGPU_REQUIRE(v >= -2 && v < last->n, "kp_to_mp[%d] = %d out of range", j, v);
        const bool held = v == -2 || (v >= 0 && (last->obs_pos ? last->obs_pos[v] != 0 : true));
        init[j] = held ? -1 : INT_MAX;
    }
    Uploader up;
    FrameDev F = upload_frame(up, cur);
    return run_projection<1>(up, F, q, last->desc, last->kp_angle, cur->kp_angle, init, 0.f, check_orientation, kp_to_mp,
                             nmatches);
}

} // extern "C"
