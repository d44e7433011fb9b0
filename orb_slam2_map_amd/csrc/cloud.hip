// RGB-D dense map for MI355X (gfx950): stride-3 back-projection (+ SE3 transform) and voxel-grid
// down-sampling.
//
// Replaces the arithmetic of ORB_SLAM2::PointCloudMapping (reference src/PointCloudMap.cc):
//   convertToPointCloud / generatePointCloud (:78-138), pcl::transformPointCloud (:105,144,172),
//   pcl::VoxelGrid<PointXYZRGBA>::filter (:240-243, 259-262, 277-278; PCL 1.7 semantics, A7).
//
// Voxel filter = (1) bounding box of the finite points, (2) per-point voxel index exactly as PCL
// computes it, (3) a stable LSD radix sort of (index, point id) -- 8-bit digits, wave-ballot
// ranking, so equal indices keep input order and the per-voxel float sums are reproducible --
// (4) head flags + device-wide scan, (5) one sequential float sum per voxel, output in ascending
// index order (PCL's output order).  HBM-bound integer/float streaming; no MFMA.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

namespace orbgpu {

using Point = orbgpu_point_xyzrgba;

// ------------------------------------------------------------------------------------------
// P1/P2 back-projection
// ------------------------------------------------------------------------------------------
struct Pose {
    double R[9], t[3];
    int apply;
};

__device__ __forceinline__ bool depth_valid(float d)
{
    // PointCloudMap.cc:121  `if (d < 0.01 || d>10) continue;`  (0.01 is a double literal)
    return !((double)d < 0.01 || d > 10);
}

// valid samples per sample-row
__global__ __launch_bounds__(256) void k_bp_count(const float *__restrict__ depth, size_t dstride, int w, int h,
                                                  int *__restrict__ row_cnt)
{
    const int row = blockIdx.x;
    const int m = row * 3;
    const int gw = (w + 2) / 3;
    int c = 0;
    for (int j = threadIdx.x; j < gw; j += 256)
        c += depth_valid(depth[(size_t)m * dstride + (size_t)j * 3]) ? 1 : 0;
    c = wave_reduce_add(c);
    __shared__ int s[4];
    if ((threadIdx.x & 63) == 0)
        s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        row_cnt[row] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of the row counts (gh <= a few thousand): one workgroup
__global__ __launch_bounds__(256) void k_bp_scan(int *__restrict__ row_cnt, int gh, long long *__restrict__ total,
                                                 long long base)
{
    __shared__ int s_carry;
    __shared__ int s_w[4];
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < gh; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int v = i < gh ? row_cnt[i] : 0;
        int inc = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int t = __shfl_up(inc, off, 64);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            s_w[wave] = inc;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) {
            if (k < wave)
                woff += s_w[k];
            tot += s_w[k];
        }
        if (i < gh)
            row_cnt[i] = s_carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0)
            s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *total = base + s_carry;
}

__global__ __launch_bounds__(256) void k_bp_write(const float *__restrict__ depth, size_t dstride,
                                                  const uint8_t *__restrict__ rgb, size_t cstride, int w, int h,
                                                  float fx, float fy, float cx, float cy, Pose pose,
                                                  const int *__restrict__ row_off, Point *__restrict__ out)
{
    __shared__ int s_w[4];
    __shared__ int s_base;
    const int row = blockIdx.x;
    const int m = row * 3;
    const int gw = (w + 2) / 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_base = row_off[row];
    __syncthreads();
    for (int j0 = 0; j0 < gw; j0 += 256) {
        const int j = j0 + threadIdx.x;
        const int n = j * 3;
        float d = 0.f;
        bool ok = false;
        if (j < gw) {
            d = depth[(size_t)m * dstride + (size_t)n];
            ok = depth_valid(d);
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0)
            s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
        for (int k = 0; k < 4; k++) {
            if (k < wave)
                off += s_w[k];
            tot += s_w[k];
        }
        off += __popcll(bal & ((1ull << lane) - 1ull));
        if (ok) {
            Point p;
            p.z = d;
            p.x = ((float)n - cx) * p.z / fx;  // :124-125, this exact operation order
            p.y = ((float)m - cy) * p.z / fy;
            const uint8_t *px = rgb + (size_t)m * cstride + (size_t)n * 3;
            p.rgba = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
            if (pose.apply && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                // pcl::transformPointCloud with a double matrix (A8)
                const double x = p.x, y = p.y, z = p.z;
                const float ox = (float)(pose.R[0] * x + pose.R[1] * y + pose.R[2] * z + pose.t[0]);
                const float oy = (float)(pose.R[3] * x + pose.R[4] * y + pose.R[5] * z + pose.t[1]);
                const float oz = (float)(pose.R[6] * x + pose.R[7] * y + pose.R[8] * z + pose.t[2]);
                p.x = ox;
                p.y = oy;
                p.z = oz;
            }
            out[off] = p;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            s_base += tot;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// P3 voxel grid
// ------------------------------------------------------------------------------------------
struct VoxState {
    unsigned mn[3], mx[3];  // order-preserving encodings of float min / max
    int nfinite;
    int min_b[3], div_b[3], mul[3];
    int overflow;
    int nout;
};

__device__ __forceinline__ unsigned f2ord(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void k_vox_init(VoxState *st)
{
    for (int a = 0; a < 3; a++) {
        st->mn[a] = 0xFFFFFFFFu;
        st->mx[a] = 0u;
    }
    st->nfinite = 0;
    st->overflow = 0;
    st->nout = 0;
}

__global__ __launch_bounds__(256) void k_vox_minmax(const Point *__restrict__ pts, long long n, VoxState *st)
{
    unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0};
    int nf = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const Point p = pts[i];
        if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z))
            continue;
        nf++;
        const unsigned o[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = min(mn[a], o[a]);
            mx[a] = max(mx[a], o[a]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            mn[a] = min(mn[a], (unsigned)__shfl_xor((int)mn[a], off, 64));
            mx[a] = max(mx[a], (unsigned)__shfl_xor((int)mx[a], off, 64));
        }
        nf += __shfl_xor(nf, off, 64);
    }
    // one set of atomics per workgroup (all workgroups hit the same 7 words: keep them few)
    __shared__ unsigned s_mn[4][3], s_mx[4][3];
    __shared__ int s_nf[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            s_mn[wave][a] = mn[a];
            s_mx[wave][a] = mx[a];
        }
        s_nf[wave] = nf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 4; w++)
            tot += s_nf[w];
        if (tot) {
#pragma unroll
            for (int a = 0; a < 3; a++) {
                unsigned lo = 0xFFFFFFFFu, hi = 0u;
                for (int w = 0; w < 4; w++) {
                    lo = min(lo, s_mn[w][a]);
                    hi = max(hi, s_mx[w][a]);
                }
                atomicMin(&st->mn[a], lo);
                atomicMax(&st->mx[a], hi);
            }
            atomicAdd(&st->nfinite, tot);
        }
    }
}

// PCL 1.7 VoxelGrid::applyFilter set-up (A7): overflow test, min_b, div_b, divb_mul
__global__ void k_vox_setup(VoxState *st, float inv)
{
    if (st->nfinite == 0)
        return;
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) {
        mn[a] = ord2f(st->mn[a]);
        mx[a] = ord2f(st->mx[a]);
    }
    const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1;
    const long long dy = (long long)((mx[1] - mn[1]) * inv) + 1;
    const long long dz = (long long)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > 2147483647ll) {
        st->overflow = 1;
        return;
    }
    for (int a = 0; a < 3; a++) {
        st->min_b[a] = (int)floorf(mn[a] * inv);
        const int max_b = (int)floorf(mx[a] * inv);
        st->div_b[a] = max_b - st->min_b[a] + 1;
    }
    st->mul[0] = 1;
    st->mul[1] = st->div_b[0];
    st->mul[2] = st->div_b[0] * st->div_b[1];
}

__global__ __launch_bounds__(256) void k_vox_keys(const Point *__restrict__ pts, long long n, const VoxState *st,
                                                  float inv, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || st->overflow)
        return;
    const Point p = pts[i];
    uint32_t key = 0xFFFFFFFFu;  // non-finite points sort to the end and are dropped
    if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        const int i0 = (int)(floorf(p.x * inv) - (float)st->min_b[0]);
        const int i1 = (int)(floorf(p.y * inv) - (float)st->min_b[1]);
        const int i2 = (int)(floorf(p.z * inv) - (float)st->min_b[2]);
        key = (uint32_t)(i0 * st->mul[0] + i1 * st->mul[1] + i2 * st->mul[2]);
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}

// ---- stable LSD radix sort, 8-bit digits, tiles of 1024 = 4 rounds x 256 threads -----------
constexpr int RS_TILE = 1024;

__device__ __forceinline__ unsigned long long match_digit(unsigned d, bool valid)
{
    // lanes of the wave holding the same 8-bit digit (invalid lanes match nobody)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return peers;
}

__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t *__restrict__ keys, long long n, int shift,
                                                 int nblocks, const VoxState *st, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[256];
    if (st->overflow)
        return;
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n)
            atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major
}

// device-wide exclusive scan (three kernels): per-tile sums, scan of the sums, add back
__global__ __launch_bounds__(256) void k_scan_tiles(unsigned *__restrict__ a, long long n,
                                                    unsigned *__restrict__ tile_sum)
{
    __shared__ unsigned s_w[4];
    const long long base = (long long)blockIdx.x * RS_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + (long long)threadIdx.x * 4 + k;
        v[k] = i < n ? a[i] : 0u;
        sum += v[k];
    }
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned t = (unsigned)__shfl_up((int)inc, off, 64);
        if (lane >= off)
            inc += t;
    }
    if (lane == 63)
        s_w[wave] = inc;
    __syncthreads();
    unsigned woff = 0, tot = 0;
    for (int k = 0; k < 4; k++) {
        if (k < wave)
            woff += s_w[k];
        tot += s_w[k];
    }
    unsigned run = woff + inc - sum;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + (long long)threadIdx.x * 4 + k;
        if (i < n)
            a[i] = run;
        run += v[k];
    }
    if (threadIdx.x == 0)
        tile_sum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_sums(unsigned *__restrict__ tile_sum, int ntiles,
                                                   unsigned *__restrict__ total)
{
    __shared__ unsigned s_carry;
    __shared__ unsigned s_w[4];
    if (threadIdx.x == 0)
        s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < ntiles; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const unsigned v = i < ntiles ? tile_sum[i] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned t = (unsigned)__shfl_up((int)inc, off, 64);
            if (lane >= off)
                inc += t;
        }
        if (lane == 63)
            s_w[wave] = inc;
        __syncthreads();
        unsigned woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) {
            if (k < wave)
                woff += s_w[k];
            tot += s_w[k];
        }
        if (i < ntiles)
            tile_sum[i] = s_carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0)
            s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total)
        *total = s_carry;
}

__global__ __launch_bounds__(256) void k_scan_add(unsigned *__restrict__ a, long long n,
                                                  const unsigned *__restrict__ tile_sum)
{
    const long long base = (long long)blockIdx.x * RS_TILE;
    const unsigned add = tile_sum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < n)
            a[i] += add;
    }
}

__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                    const uint32_t *__restrict__ vals_in, long long n, int shift,
                                                    int nblocks, const VoxState *st,
                                                    const unsigned *__restrict__ hist_scanned,
                                                    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out)
{
    // h[round][wave][digit]: elements of the tile are ordered (round, wave, lane)
    __shared__ unsigned h[16 * 256];
    if (st->overflow)
        return;
    for (int i = threadIdx.x; i < 16 * 256; i += 256)
        h[i] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * RS_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t key[4], val[4];
    unsigned rank[4];
    bool valid[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = base + k * 256 + threadIdx.x;
        valid[k] = i < n;
        key[k] = valid[k] ? keys_in[i] : 0u;
        val[k] = valid[k] ? vals_in[i] : 0u;
        const unsigned d = (key[k] >> shift) & 255u;
        const unsigned long long peers = match_digit(d, valid[k]);
        rank[k] = __popcll(peers & ((1ull << lane) - 1ull));
        if (valid[k] && rank[k] == 0)
            h[(k * 4 + wave) * 256 + d] = __popcll(peers);
    }
    __syncthreads();
    {  // exclusive prefix over the 16 (round, wave) slots, one digit per thread
        unsigned run = hist_scanned[(size_t)threadIdx.x * nblocks + blockIdx.x];
        for (int s = 0; s < 16; s++) {
            const unsigned c = h[s * 256 + threadIdx.x];
            h[s * 256 + threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!valid[k])
            continue;
        const unsigned d = (key[k] >> shift) & 255u;
        const unsigned pos = h[(k * 4 + wave) * 256 + d] + rank[k];
        keys_out[pos] = key[k];
        vals_out[pos] = val[k];
    }
}

// head flag of every voxel run (finite points only)
__global__ __launch_bounds__(256) void k_vox_heads(const uint32_t *__restrict__ keys, long long n,
                                                   const VoxState *st, unsigned *__restrict__ flag)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    unsigned f = 0;
    if (!st->overflow && i < st->nfinite)
        f = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    flag[i] = f;
}

// one thread per voxel run: float sums in sorted (= input) order, PCL's centroid + rgb packing
__global__ __launch_bounds__(256) void k_vox_reduce(const Point *__restrict__ pts, const uint32_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ vals, long long n,
                                                    const unsigned *__restrict__ pos, VoxState *st,
                                                    Point *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (st->overflow) {
        if (i < n)
            out[i] = pts[i];  // PCL: "Leaf size is too small" -> output = input
        if (i == 0)
            st->nout = (int)n;
        return;
    }
    const int nf = st->nfinite;
    if (i >= nf)
        return;
    const uint32_t k = keys[i];
    if (i != 0 && keys[i - 1] == k)
        return;
    float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
    long long j = i;
    for (; j < nf && keys[j] == k; j++) {
        const Point p = pts[vals[j]];
        const float r = (float)((p.rgba >> 16) & 255u), g = (float)((p.rgba >> 8) & 255u), b = (float)(p.rgba & 255u);
        if (j == i) {
            sx = p.x; sy = p.y; sz = p.z; sr = r; sg = g; sb = b;
        } else {
            sx += p.x; sy += p.y; sz += p.z; sr += r; sg += g; sb += b;
        }
    }
    const float cnt = (float)(j - i);
    Point o;
    o.x = sx / cnt;
    o.y = sy / cnt;
    o.z = sz / cnt;
    const int ri = (int)(sr / cnt), gi = (int)(sg / cnt), bi = (int)(sb / cnt);
    o.rgba = (uint32_t)((ri << 16) | (gi << 8) | bi);
    out[pos[i]] = o;
    if (j == nf)
        st->nout = (int)pos[i] + 1;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct VoxelWorkspace {
    DevBuf keys[2], vals[2], hist, tile_sum, flag, state;
    int reserve(long long n)
    {
        const long long nb = (n + RS_TILE - 1) / RS_TILE;
        int rc;
        for (int k = 0; k < 2; k++) {
            if ((rc = keys[k].reserve(sizeof(uint32_t) * (size_t)n)) != ORBGPU_OK)
                return rc;
            if ((rc = vals[k].reserve(sizeof(uint32_t) * (size_t)n)) != ORBGPU_OK)
                return rc;
        }
        if ((rc = hist.reserve(sizeof(unsigned) * 256 * (size_t)nb)) != ORBGPU_OK)
            return rc;
        const long long nt = std::max<long long>((std::max<long long>(n, 256 * nb) + RS_TILE - 1) / RS_TILE, 1);
        if ((rc = tile_sum.reserve(sizeof(unsigned) * (size_t)nt)) != ORBGPU_OK)
            return rc;
        if ((rc = flag.reserve(sizeof(unsigned) * (size_t)n)) != ORBGPU_OK)
            return rc;
        return state.reserve(sizeof(VoxState));
    }
    void release()
    {
        for (int k = 0; k < 2; k++) {
            keys[k].release();
            vals[k].release();
        }
        hist.release();
        tile_sum.release();
        flag.release();
        state.release();
    }
};

static int device_scan(unsigned *a, long long n, unsigned *tile_sum, hipStream_t st)
{
    const int nt = (int)((n + RS_TILE - 1) / RS_TILE);
    hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(256), 0, st, a, n, tile_sum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, tile_sum, nt, (unsigned *)nullptr);
    hipLaunchKernelGGL(k_scan_add, dim3(nt), dim3(256), 0, st, a, n, tile_sum);
    return ORBGPU_OK;
}

// in[0..n) -> out (capacity >= n); the output count lands in ws.state->nout.
static int voxel_filter_device(VoxelWorkspace &ws, const Point *in, long long n, float leaf, Point *out,
                               hipStream_t st)
{
    int rc = ws.reserve(std::max<long long>(n, 1));
    if (rc != ORBGPU_OK)
        return rc;
    VoxState *S = ws.state.as<VoxState>();
    const float inv = 1.0f / leaf;  // Eigen::Array4f::Ones() / leaf_size_
    hipLaunchKernelGGL(k_vox_init, dim3(1), dim3(1), 0, st, S);
    if (n == 0)
        return ORBGPU_OK;
    const int nb256 = (int)((n + 255) / 256);
    hipLaunchKernelGGL(k_vox_minmax, dim3(std::min(nb256, 256)), dim3(256), 0, st, in, n, S);
    hipLaunchKernelGGL(k_vox_setup, dim3(1), dim3(1), 0, st, S, inv);
    uint32_t *k0 = ws.keys[0].as<uint32_t>(), *k1 = ws.keys[1].as<uint32_t>();
    uint32_t *v0 = ws.vals[0].as<uint32_t>(), *v1 = ws.vals[1].as<uint32_t>();
    hipLaunchKernelGGL(k_vox_keys, dim3(nb256), dim3(256), 0, st, in, n, S, inv, k0, v0);
    const int nblocks = (int)((n + RS_TILE - 1) / RS_TILE);
    unsigned *hist = ws.hist.as<unsigned>();
    for (int pass = 0; pass < 4; pass++) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(256), 0, st, k0, n, shift, nblocks, S, hist);
        device_scan(hist, 256ll * nblocks, ws.tile_sum.as<unsigned>(), st);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblocks), dim3(256), 0, st, k0, v0, n, shift, nblocks, S, hist, k1, v1);
        std::swap(k0, k1);
        std::swap(v0, v1);
    }
    unsigned *flag = ws.flag.as<unsigned>();
    hipLaunchKernelGGL(k_vox_heads, dim3(nb256), dim3(256), 0, st, k0, n, S, flag);
    device_scan(flag, n, ws.tile_sum.as<unsigned>(), st);
    hipLaunchKernelGGL(k_vox_reduce, dim3(nb256), dim3(256), 0, st, in, k0, v0, n, flag, S, out);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

// Converter::toSE3Quat + g2o::SE3Quat normalisation + Isometry3d::inverse (Converter.cc:37-47,
// PointCloudMap.cc:103-105; A9): Tcw float 4x4 -> Twc (double).
static void pose_inverse(const float *Tcw, Pose &P)
{
    double m[3][3], tt[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            m[i][j] = (double)Tcw[4 * i + j];
        tt[i] = (double)Tcw[4 * i + 3];
    }
    double q[4];  // x y z w  (Eigen quaternion from a rotation matrix)
    const double tr = m[0][0] + m[1][1] + m[2][2];
    if (tr > 0) {
        double s = sqrt(tr + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (m[2][1] - m[1][2]) * s;
        q[1] = (m[0][2] - m[2][0]) * s;
        q[2] = (m[1][0] - m[0][1]) * s;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0])
            i = 1;
        if (m[2][2] > m[i][i])
            i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (m[k][j] - m[j][k]) * s;
        q[j] = (m[j][i] + m[i][j]) * s;
        q[k] = (m[k][i] + m[i][k]) * s;
    }
    if (q[3] < 0)
        for (int i = 0; i < 4; i++)
            q[i] = -q[i];
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++)
        q[i] /= nrm;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    double Rq[3][3];
    Rq[0][0] = 1 - (tyy + tzz);
    Rq[0][1] = txy - twz;
    Rq[0][2] = txz + twy;
    Rq[1][0] = txy + twz;
    Rq[1][1] = 1 - (txx + tzz);
    Rq[1][2] = tyz - twx;
    Rq[2][0] = txz - twy;
    Rq[2][1] = tyz + twx;
    Rq[2][2] = 1 - (txx + tyy);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            P.R[3 * i + j] = Rq[j][i];
    for (int i = 0; i < 3; i++)
        P.t[i] = -(P.R[3 * i + 0] * tt[0] + P.R[3 * i + 1] * tt[1] + P.R[3 * i + 2] * tt[2]);
    P.apply = 1;
}

struct FrameStage {
    DevBuf depth, rgb, row_cnt, total;
};

// depth/rgb host -> device, back-project (+ transform) appending to out[base..]; *d_total = base + count
static int backproject_device(FrameStage &fs, const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride,
                              int w, int h, float fx, float fy, float cx, float cy, const float *Tcw, Point *out,
                              long long base, hipStream_t st)
{
    int rc;
    const int gh = (h + 2) / 3;
    if ((rc = fs.depth.reserve(sizeof(float) * (size_t)w * h)) != ORBGPU_OK ||
        (rc = fs.rgb.reserve((size_t)w * 3 * h)) != ORBGPU_OK || (rc = fs.row_cnt.reserve(sizeof(int) * gh)) != ORBGPU_OK ||
        (rc = fs.total.reserve(sizeof(long long))) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpy2DAsync(fs.depth.p, sizeof(float) * w, depth, sizeof(float) * dstride, sizeof(float) * w, h,
                                    hipMemcpyHostToDevice, st));
    ORBGPU_HIP_TRY(hipMemcpy2DAsync(fs.rgb.p, (size_t)w * 3, rgb, cstride, (size_t)w * 3, h, hipMemcpyHostToDevice, st));
    Pose P;
    P.apply = 0;
    if (Tcw)
        pose_inverse(Tcw, P);
    hipLaunchKernelGGL(k_bp_count, dim3(gh), dim3(256), 0, st, fs.depth.as<float>(), (size_t)w, w, h, fs.row_cnt.as<int>());
    hipLaunchKernelGGL(k_bp_scan, dim3(1), dim3(256), 0, st, fs.row_cnt.as<int>(), gh, fs.total.as<long long>(), base);
    hipLaunchKernelGGL(k_bp_write, dim3(gh), dim3(256), 0, st, fs.depth.as<float>(), (size_t)w, fs.rgb.as<uint8_t>(),
                       (size_t)w * 3, w, h, fx, fy, cx, cy, P, fs.row_cnt.as<int>(), out + base);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_cloud {
    int device_id = 0;
    float leaf = 0.01f;
    DevBuf map[2];        // ping-pong: map[cur] holds the global map
    int cur = 0;
    long long size = 0;   // points in the global map
    int last_overflow = 0;
    VoxelWorkspace ws;
    FrameStage fs;
    hipStream_t stream = nullptr;
};

namespace orbgpu {

static int cloud_grow(orbgpu_cloud *c, int which, long long need_points, long long keep_points)
{
    DevBuf &b = c->map[which];
    const size_t need = sizeof(Point) * (size_t)need_points;
    if (need <= b.bytes)
        return ORBGPU_OK;
    DevBuf nb;
    int rc = nb.reserve(std::max(need * 2, (size_t)1 << 20));
    if (rc != ORBGPU_OK)
        return rc;
    if (keep_points > 0 && b.p) {
        hipError_t e = hipMemcpyAsync(nb.p, b.p, sizeof(Point) * (size_t)keep_points, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            set_error("grow map: %s", hipGetErrorString(e));
            nb.release();
            return ORBGPU_EHIP;
        }
    }
    b.release();
    b = nb;
    return ORBGPU_OK;
}

// filter map[cur][0..k) into map[cur^1], swap, read the new size
static int cloud_filter(orbgpu_cloud *c, long long k)
{
    int rc = cloud_grow(c, c->cur ^ 1, std::max<long long>(k, 1), 0);
    if (rc != ORBGPU_OK)
        return rc;
    rc = voxel_filter_device(c->ws, c->map[c->cur].as<Point>(), k, c->leaf, c->map[c->cur ^ 1].as<Point>(), c->stream);
    if (rc != ORBGPU_OK)
        return rc;
    VoxState hs;
    ORBGPU_HIP_TRY(hipMemcpyAsync(&hs, c->ws.state.p, sizeof(VoxState), hipMemcpyDeviceToHost, c->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(c->stream));
    c->cur ^= 1;
    c->size = hs.nout;
    c->last_overflow = hs.overflow;
    return ORBGPU_OK;
}

} // namespace orbgpu

extern "C" {

int orbgpu_cloud_create(double resolution, int32_t device_id, orbgpu_cloud **out)
{
    ORBGPU_REQUIRE(out, "null argument");
    ORBGPU_REQUIRE(resolution > 0 && std::isfinite(resolution), "resolution must be positive");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_cloud *c = new (std::nothrow) orbgpu_cloud();
    if (!c) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    c->device_id = device_id;
    c->leaf = (float)resolution;  // setLeafSize(float, float, float), PointCloudMap.cc:41
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(e));
        delete c;
        return ORBGPU_EHIP;
    }
    *out = c;
    return ORBGPU_OK;
}

int orbgpu_cloud_destroy(orbgpu_cloud *c)
{
    if (!c)
        return ORBGPU_OK;
    (void)hipSetDevice(c->device_id);
    (void)hipDeviceSynchronize();
    c->map[0].release();
    c->map[1].release();
    c->ws.release();
    c->fs.depth.release();
    c->fs.rgb.release();
    c->fs.row_cnt.release();
    c->fs.total.release();
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return ORBGPU_OK;
}

static int check_frame_args(const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride, int w, int h)
{
    ORBGPU_REQUIRE(depth && rgb, "null image");
    ORBGPU_REQUIRE(w > 0 && h > 0 && dstride >= (size_t)w && cstride >= (size_t)w * 3, "bad image size / strides");
    return ORBGPU_OK;
}

int orbgpu_cloud_insert(orbgpu_cloud *c, const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride,
                        int32_t w, int32_t h, float fx, float fy, float cx, float cy, const float *Tcw)
{
    ORBGPU_REQUIRE(c && Tcw, "null argument");
    int rc = check_frame_args(depth, dstride, rgb, cstride, w, h);
    if (rc != ORBGPU_OK)
        return rc;
    if ((rc = select_device(c->device_id)) != ORBGPU_OK)
        return rc;
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    if ((rc = cloud_grow(c, c->cur, c->size + maxnew, c->size)) != ORBGPU_OK)
        return rc;
    // globalMap += transform(convertToPointCloud(kf), Twc)   (:204-249)
    if ((rc = backproject_device(c->fs, depth, dstride, rgb, cstride, w, h, fx, fy, cx, cy, Tcw,
                                 c->map[c->cur].as<Point>(), c->size, c->stream)) != ORBGPU_OK)
        return rc;
    long long total = 0;
    ORBGPU_HIP_TRY(hipMemcpyAsync(&total, c->fs.total.p, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(c->stream));
    // voxel.setInputCloud(globalMap); voxel.filter(tmp); swap   (:259-262)
    return cloud_filter(c, total);
}

int orbgpu_cloud_rebuild(orbgpu_cloud *c, int32_t n, const float *const *depth, size_t dstride,
                         const uint8_t *const *rgb, size_t cstride, int32_t w, int32_t h, float fx, float fy, float cx,
                         float cy, const float *const *Tcw)
{
    ORBGPU_REQUIRE(c && n >= 0 && (n == 0 || (depth && rgb && Tcw)), "bad arguments");
    int rc = select_device(c->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    c->size = 0;  // globalMap.reset(new PointCloud)  (:225)
    if (n == 0)
        return ORBGPU_OK;
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    if ((rc = cloud_grow(c, c->cur, maxnew * n, 0)) != ORBGPU_OK)
        return rc;
    long long total = 0;
    for (int k = 0; k < n; k++) {
        ORBGPU_REQUIRE(Tcw[k], "null pose");
        if ((rc = check_frame_args(depth[k], dstride, rgb[k], cstride, w, h)) != ORBGPU_OK)
            return rc;
        if ((rc = backproject_device(c->fs, depth[k], dstride, rgb[k], cstride, w, h, fx, fy, cx, cy, Tcw[k],
                                     c->map[c->cur].as<Point>(), total, c->stream)) != ORBGPU_OK)
            return rc;
        ORBGPU_HIP_TRY(hipMemcpyAsync(&total, c->fs.total.p, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        ORBGPU_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return cloud_filter(c, total);
}

int orbgpu_cloud_size(orbgpu_cloud *c, int64_t *n)
{
    ORBGPU_REQUIRE(c && n, "null argument");
    *n = c->size;
    return ORBGPU_OK;
}

int orbgpu_cloud_download(orbgpu_cloud *c, orbgpu_point_xyzrgba *out, int64_t cap, int64_t *n)
{
    ORBGPU_REQUIRE(c && out && n, "null argument");
    if (cap < c->size) {
        set_error("cloud has %lld points, cap is %lld", (long long)c->size, (long long)cap);
        return ORBGPU_ECAPACITY;
    }
    int rc = select_device(c->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (c->size > 0)
        ORBGPU_HIP_TRY(hipMemcpy(out, c->map[c->cur].p, sizeof(Point) * (size_t)c->size, hipMemcpyDeviceToHost));
    *n = c->size;
    return ORBGPU_OK;
}

int orbgpu_cloud_last_overflow(orbgpu_cloud *c, int32_t *overflow)
{
    ORBGPU_REQUIRE(c && overflow, "null argument");
    *overflow = c->last_overflow;
    return ORBGPU_OK;
}

int orbgpu_backproject(const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride, int32_t w, int32_t h,
                       float fx, float fy, float cx, float cy, const float *Tcw, orbgpu_point_xyzrgba *out, int64_t cap,
                       int64_t *n, int32_t device_id)
{
    ORBGPU_REQUIRE(out && n, "null argument");
    int rc = check_frame_args(depth, dstride, rgb, cstride, w, h);
    if (rc != ORBGPU_OK)
        return rc;
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    if (cap < maxnew) {
        set_error("cap %lld < worst case %lld", (long long)cap, maxnew);
        return ORBGPU_ECAPACITY;
    }
    if ((rc = select_device(device_id)) != ORBGPU_OK)
        return rc;
    FrameStage fs;
    DevBuf dout;
    auto cleanup = [&]() {
        fs.depth.release();
        fs.rgb.release();
        fs.row_cnt.release();
        fs.total.release();
        dout.release();
    };
    if ((rc = dout.reserve(sizeof(Point) * (size_t)maxnew)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    rc = backproject_device(fs, depth, dstride, rgb, cstride, w, h, fx, fy, cx, cy, Tcw, dout.as<Point>(), 0, nullptr);
    long long total = 0;
    hipError_t e = hipSuccess;
    if (rc == ORBGPU_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess)
            e = hipMemcpy(&total, fs.total.p, sizeof(long long), hipMemcpyDeviceToHost);
        if (e == hipSuccess && total > 0)
            e = hipMemcpy(out, dout.p, sizeof(Point) * (size_t)total, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            set_error("backproject: %s", hipGetErrorString(e));
            rc = ORBGPU_EHIP;
        }
    }
    cleanup();
    *n = total;
    return rc;
}

int orbgpu_voxel_filter(const orbgpu_point_xyzrgba *in, int64_t n, double resolution, orbgpu_point_xyzrgba *out,
                        int64_t cap, int64_t *n_out, int32_t *overflow, int32_t device_id)
{
    ORBGPU_REQUIRE(n >= 0 && n < (1ll << 31) && (n == 0 || in) && out && n_out, "bad arguments");
    ORBGPU_REQUIRE(resolution > 0, "resolution must be positive");
    ORBGPU_REQUIRE(cap >= n, "cap must be >= n (the overflow path returns the input)");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    *n_out = 0;
    if (overflow)
        *overflow = 0;
    if (n == 0)
        return ORBGPU_OK;
    VoxelWorkspace ws;
    DevBuf din, dout;
    auto cleanup = [&]() {
        ws.release();
        din.release();
        dout.release();
    };
    if ((rc = din.reserve(sizeof(Point) * (size_t)n)) != ORBGPU_OK || (rc = dout.reserve(sizeof(Point) * (size_t)n)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpy(din.p, in, sizeof(Point) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = voxel_filter_device(ws, din.as<Point>(), n, (float)resolution, dout.as<Point>(), nullptr);
        if (rc == ORBGPU_OK) {
            VoxState hs;
            e = hipDeviceSynchronize();
            if (e == hipSuccess)
                e = hipMemcpy(&hs, ws.state.p, sizeof(VoxState), hipMemcpyDeviceToHost);
            if (e == hipSuccess && hs.nout > 0)
                e = hipMemcpy(out, dout.p, sizeof(Point) * (size_t)hs.nout, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                *n_out = hs.nout;
                if (overflow)
                    *overflow = hs.overflow;
            }
        }
    }
    if (e != hipSuccess) {
        set_error("voxel_filter: %s", hipGetErrorString(e));
        rc = ORBGPU_EHIP;
    }
    cleanup();
    return rc;
}

} // extern "C"
