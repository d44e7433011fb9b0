// RGB-D dense map for MI355X (gfx950): stride-3 back-projection (+ SE3 transform) and voxel-grid
// down-sampling.
//
// Replaces the arithmetic of ORB_SLAM2::PointCloudMapping (reference src/PointCloudMap.cc):
//   convertToPointCloud / generatePointCloud (:78-138), pcl::transformPointCloud (:105,144,172),
//   pcl::VoxelGrid<PointXYZRGBA>::filter (:240-243, 259-262, 277-278; PCL 1.7 semantics, A7).
//
// The reference re-filters the WHOLE accumulated map at every key frame (:259-262).  PCL's result is
// "stable sort of (old map ++ new points) by voxel index, one float sum per run, ascending output".
// The old map is itself such an output -- sorted, one point per voxel, and that order (z-major
// lexicographic on the lattice coordinates) does not depend on the bounding box -- so a key frame is a MERGE:
//   k_bp         back-project + transform + compact the <= 34 k new points (one launch, look-back offsets)
//   k_vox_keys   PCL's set-up from (box of the map, box of the new points), voxel indices of the new points; further
//                workgroups of the same launch index EVERY resident point under that set-up (4 bytes per point) and
//                sample 4096 of those indices into a table
//   k_sort_pass  x4: stable LSD radix sort of the new points only (one launch per digit, look-back)
//   k_merge_new  256 sorted new keys per workgroup: the table brackets the resident range they can fall into, its
//                indices are staged in LDS and every run of equal keys is ranked there -> centroid (seeded with the
//                map's point of that voxel if any) at its merged position
//   k_merge_old  untouched map points shift up by the number of new voxels before them (the map is read and
//                written once: K*16 B in, V*16 B out = the algorithmic bytes of SURVEY.md 8d)
// 8 launches per key frame instead of 35, no full-map sort.  What the merge assumes is checked on the fly
// (map strictly increasing under the new indices, no int32 overflow); otherwise the key frame is redone by
// the general path = the same sort over all points + a fused head-flag / scan / reduce kernel (7 launches),
// which also serves orbgpu_voxel_filter and the loop-closure rebuild.  Equal indices keep input order in both
// paths, so the per-voxel float sums are reproducible.  HBM-bound integer/float streaming; no MFMA.
//
// What bounds a key frame is latency, not bytes (s_memrealtime stamps inside the kernels): a
// dependent load inside a kernel is a 1 - 1.5 us round trip whichever cache answers it, a publish -> read hop between
// workgroups (look-back) ~3 us, a kernel boundary ~2.2 us, and ONE workgroup draws ~12 - 15 GB/s.  Hence: loads are
// requested before the chain that does not need them, data a workgroup searches is staged in LDS with one coalesced
// request, work that is wide (indexing 700 k resident points) rides in launches that are narrow (34 tiles of new points).
#include "common.h"

#include <cstdlib>
#include <cstring>

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

// ------------------------------------------------------------------------------------------
// Decoupled look-back (single-pass device-wide prefix sums over tiles).
//
// Every kernel that needs an ordered prefix over its workgroups (compaction offsets, radix-sort digit
// offsets, voxel output positions) takes a tile id -- blockIdx.x in launches small enough to be resident as a whole,
// else a dynamic one from a ticket counter, so that a tile only ever waits for tiles that have already started
// (lb_take_tile) -- publishes its aggregate, sums the published values of its
// predecessors until it meets an inclusive prefix, and publishes its own inclusive prefix.
// Status word: [63:32] launch epoch (stale words of earlier launches never match, nothing is cleared
// between launches), [31:30] 1 = aggregate / 2 = inclusive prefix, [29:0] value.  The last workgroup to
// leave re-arms the ticket counter for the next launch on the stream.
// ------------------------------------------------------------------------------------------
struct LbCtl {
    unsigned ticket, finished;
};
constexpr unsigned long long LB_AGG = 1ull << 30, LB_PREFIX = 2ull << 30, LB_FLAGS = 3ull << 30;

__device__ __forceinline__ void lb_publish(unsigned long long *slot, unsigned epoch, unsigned long long flag, unsigned v)
{
    // relaxed: the word carries its own payload, no other data is handed over, so no L2 write-back / invalidate
    __hip_atomic_store(slot, ((unsigned long long)epoch << 32) | flag | (unsigned long long)v, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of channel `chan` over tiles [0, tile); `stride` = channels per tile.
// (Requesting 16 or 32 predecessors' words at a time, with or without re-requesting the whole batch while one is
// missing, measured the same 3 - 3.4 us for the last of 34 tiles as this word-by-word walk: an agent-scope load is served
// from memory, not from the XCD's L2, and one publish -> read hop across XCDs is ~1.5 us whatever is batched around it.)
__device__ __forceinline__ unsigned lb_exclusive(unsigned long long *status, int stride, int chan, int tile,
                                                 unsigned epoch)
{
    unsigned sum = 0;
    for (int t = tile - 1; t >= 0; --t) {
        unsigned long long w;
        for (;;) {
            w = __hip_atomic_load(status + (size_t)t * stride + chan, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == epoch && (w & LB_FLAGS))
                break;
            __builtin_amdgcn_s_sleep(1);
        }
        sum += (unsigned)(w & 0x3FFFFFFFull);
        if (w & LB_PREFIX)
            break;
    }
    return sum;
}
__device__ __forceinline__ bool lb_valid(unsigned long long w, unsigned epoch)
{
    return (unsigned)(w >> 32) == epoch && (w & LB_FLAGS) != 0ull;
}

// one-channel convenience for a whole workgroup: wave 0 publishes / looks back (lane l reads the word of tile
// tile-1-l, 64 predecessors per round trip), everybody gets the base
__device__ __forceinline__ unsigned lb_tile_base(unsigned long long *status, int tile, unsigned epoch, unsigned agg,
                                                 unsigned *s_slot)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned base = 0;
        if (tile > 0) {
            if (lane == 0)
                lb_publish(status + tile, epoch, LB_AGG, agg);
            for (int t0 = tile - 1; t0 >= 0;) {
                const int t = t0 - lane;
                unsigned long long w = 0ull;
                if (t >= 0)
                    w = __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = t < 0 || lb_valid(w, epoch);
                const bool last = t < 0 || (ok && (w & LB_PREFIX) != 0ull);  // nothing below this lane is needed
                const unsigned long long okm = __ballot(ok), lastm = __ballot(last);
                const int L = lastm ? (int)__builtin_ctzll(lastm) : 63;      // lanes 0..L are wanted
                const unsigned long long want = L == 63 ? ~0ull : ((2ull << L) - 1ull);
                if ((okm & want) != want) {  // a wanted word is not published yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                base += (unsigned)wave_reduce_add((lane <= L && t >= 0) ? (int)(w & 0x3FFFFFFFull) : 0);
                if (lastm)
                    break;
                t0 -= 64;
            }
            base = (unsigned)__shfl((int)base, 0, 64);
        }
        if (lane == 0) {
            lb_publish(status + tile, epoch, LB_PREFIX, base + agg);
            *s_slot = base;
        }
    }
    __syncthreads();
    return *s_slot;
}

// Tile ids.  A launch of at most LB_STATIC_GRID workgroups uses blockIdx.x and no ticket: every one of its workgroups
// gets a slot on the device whatever the others do (256 CUs, several such workgroups each; kernels of other streams
// drain on their own), so a workgroup that spins on a lower tile always sees it arrive.  Larger launches take tickets,
// which makes the wait safe when only part of the grid is resident.  (The two agent-scope atomics -- ticket and
// "finished" -- were 1.3 of the 7.7 us of a radix pass over 34 tiles.)
constexpr unsigned LB_STATIC_GRID = 256;
__device__ __forceinline__ int lb_take_tile(LbCtl *ctl, int *s_tile)
{
    if (gridDim.x <= LB_STATIC_GRID)
        return (int)blockIdx.x;
    if (threadIdx.x == 0)
        *s_tile = (int)__hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *s_tile;
}

__device__ __forceinline__ void lb_leave(LbCtl *ctl)
{
    if (gridDim.x <= LB_STATIC_GRID)
        return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // every workgroup has taken its ticket before it gets here; the stores below become visible to the next
        // launch at the kernel boundary
        if (__hip_atomic_fetch_add(&ctl->finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(&ctl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

constexpr int TILE = 1024;  // elements per workgroup (256 threads x 4), order inside a tile: (round, wave, lane)
constexpr int MERGE_TABLE = 4096;  // sampled indices of the resident map (k_vox_keys -> k_merge_new)
static inline long long tiles_of_ll(long long n) { return (n + TILE - 1) / TILE; }

// ------------------------------------------------------------------------------------------
// bounding boxes (order-preserving encodings of float min / max) and PCL's voxel index
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

struct Box {  // of finite points
    unsigned mn[3], mx[3];
    int nfinite;
    int pad;
};

__device__ __forceinline__ void box_init(Box &b)
{
    for (int a = 0; a < 3; a++) {
        b.mn[a] = 0xFFFFFFFFu;
        b.mx[a] = 0u;
    }
    b.nfinite = 0;
    b.pad = 0;
}
__device__ __forceinline__ void box_add(Box &b, const Point &p)
{
    if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z))
        return;
    const unsigned o[3] = {f2ord(p.x), f2ord(p.y), f2ord(p.z)};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        b.mn[a] = min(b.mn[a], o[a]);
        b.mx[a] = max(b.mx[a], o[a]);
    }
    b.nfinite++;
}
__device__ __forceinline__ void box_merge(Box &b, const Box &o)
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        b.mn[a] = min(b.mn[a], o.mn[a]);
        b.mx[a] = max(b.mx[a], o.mx[a]);
    }
    b.nfinite += o.nfinite;
}
// workgroup (256 threads) reduction; the result is valid in every thread
__device__ __forceinline__ void box_block_reduce(Box &b, Box *s_box /* [4] */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            b.mn[a] = min(b.mn[a], (unsigned)__shfl_xor((int)b.mn[a], off, 64));
            b.mx[a] = max(b.mx[a], (unsigned)__shfl_xor((int)b.mx[a], off, 64));
        }
        b.nfinite += __shfl_xor(b.nfinite, off, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        s_box[threadIdx.x >> 6] = b;
    __syncthreads();
    b = s_box[0];
    for (int w = 1; w < 4; w++)
        box_merge(b, s_box[w]);
}
__device__ __forceinline__ void box_atomic(unsigned *box6, const Box &b)  // box6 = mn[3], mx[3]
{
    if (!b.nfinite)
        return;
    // hundreds of workgroups target the same six words: only those that would extend the box issue the atomic
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (b.mn[a] < __hip_atomic_load(&box6[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(&box6[a], b.mn[a]);
        if (b.mx[a] > __hip_atomic_load(&box6[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&box6[3 + a], b.mx[a]);
    }
}

// The bounding box of a map under construction: BOX_SLOTS copies of the six words, one cache line apart.  A workgroup
// folds its box into slot blockIdx.x % BOX_SLOTS with six atomics it does not wait for; the consumer (the next insert's
// k_vox_keys) folds the slots.  Until round 4 every workgroup first LOOKED at the six shared words (to skip atomics that
// would not extend the box): six dependent agent-scope loads, ~1.5 us each, at the tail of every workgroup of the merge
// kernels -- 9 of k_merge_new's 35 us (s_memrealtime stamps).
constexpr int BOX_SLOTS = 16, BOX_SLOT_WORDS = 32, BOX_WORDS = BOX_SLOTS * BOX_SLOT_WORDS;
__device__ __forceinline__ void box_slots_init(unsigned *boxs)  // by one workgroup of 256 threads
{
    for (int i = threadIdx.x; i < BOX_WORDS; i += 256)
        boxs[i] = (i & (BOX_SLOT_WORDS - 1)) < 3 ? 0xFFFFFFFFu : 0u;
}
__device__ __forceinline__ void box_slots_atomic(unsigned *boxs, const Box &b)
{
    if (!b.nfinite)
        return;
    unsigned *box6 = boxs + (blockIdx.x % BOX_SLOTS) * BOX_SLOT_WORDS;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        atomicMin(&box6[a], b.mn[a]);
        atomicMax(&box6[3 + a], b.mx[a]);
    }
}
__device__ __forceinline__ void box_slots_fold(const unsigned *boxs, Box &b)  // every thread reads all slots (L2 hits)
{
    for (int a = 0; a < 3; a++) {
        b.mn[a] = 0xFFFFFFFFu;
        b.mx[a] = 0u;
    }
    for (int sl = 0; sl < BOX_SLOTS; sl++)
        for (int a = 0; a < 3; a++) {
            b.mn[a] = min(b.mn[a], boxs[sl * BOX_SLOT_WORDS + a]);
            b.mx[a] = max(b.mx[a], boxs[sl * BOX_SLOT_WORDS + 3 + a]);
        }
}

// state of one filter call, device resident; the host reads it back when the call has been queued
struct CloudState {
    int n_sort;    // elements in the sort (fast path: the new points; general path: the whole input)
    int nfinite;   // finite ones among them (they sort to the front, non-finite keys are 0xFFFFFFFF)
    int min_b[3], div_b[3], mul[3];
    int overflow;  // PCL's int32 voxel-index overflow rule fired
    int unsorted;  // fast path only: the old map is not strictly increasing under the new keys
    int nout;
};

struct VoxSetup {
    int min_b[3], mul[3];
    int overflow, valid;
};

// PCL 1.7 VoxelGrid::applyFilter set-up (A7): overflow test, min_b, div_b, divb_mul
__device__ __forceinline__ void vox_setup(const Box &b, float inv, VoxSetup &v, CloudState *st_out)
{
    v.overflow = 0;
    v.valid = b.nfinite > 0;
    for (int a = 0; a < 3; a++)
        v.min_b[a] = v.mul[a] = 0;
    int div_b[3] = {0, 0, 0};
    if (v.valid) {
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) {
            mn[a] = ord2f(b.mn[a]);
            mx[a] = ord2f(b.mx[a]);
        }
        const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1;
        const long long dy = (long long)((mx[1] - mn[1]) * inv) + 1;
        const long long dz = (long long)((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > 2147483647ll) {
            v.overflow = 1;
        } else {
            for (int a = 0; a < 3; a++) {
                v.min_b[a] = (int)floorf(mn[a] * inv);
                const int max_b = (int)floorf(mx[a] * inv);
                div_b[a] = max_b - v.min_b[a] + 1;
            }
            v.mul[0] = 1;
            v.mul[1] = div_b[0];
            v.mul[2] = div_b[0] * div_b[1];
        }
    }
    if (st_out) {
        for (int a = 0; a < 3; a++) {
            st_out->min_b[a] = v.min_b[a];
            st_out->div_b[a] = div_b[a];
            st_out->mul[a] = v.mul[a];
        }
        st_out->overflow = v.overflow;
        st_out->nfinite = b.nfinite;
        st_out->unsorted = 0;
        st_out->nout = 0;
    }
}

__device__ __forceinline__ uint32_t vox_key(const Point &p, float inv, const int min_b[3], const int mul[3])
{
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z)))
        return 0xFFFFFFFFu;  // non-finite points sort to the end and are dropped
    const int i0 = (int)(floorf(p.x * inv) - (float)min_b[0]);
    const int i1 = (int)(floorf(p.y * inv) - (float)min_b[1]);
    const int i2 = (int)(floorf(p.z * inv) - (float)min_b[2]);
    return (uint32_t)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]);
}

// ------------------------------------------------------------------------------------------
// P1/P2: stride-3 back-projection (+ SE3 transform), compacted in scan order in ONE launch
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool depth_valid(float d)
{
    // PointCloudMap.cc:121  `if (d < 0.01 || d>10) continue;`  (0.01 is a double literal)
    return !((double)d < 0.01 || d > 10);
}

// A tile = 1024 consecutive samples of the (ceil(w/3) x ceil(h/3)) sample grid in raster order.  Valid samples
// are counted with wave ballots, the tile's offset comes from the look-back, points are written in raster order
// (the reference's push_back order).  Also: the tile's bounding box (what the voxel set-up of the next kernel
// needs), the total count, and the re-initialisation of the accumulators later kernels of the sequence add into.
__global__ __launch_bounds__(256) void k_bp(const float *__restrict__ depth, size_t dstride,
                                            const uint8_t *__restrict__ rgb, size_t cstride, int w, int h, float fx,
                                            float fy, float cx, float cy, Pose pose, Point *__restrict__ out,
                                            LbCtl *ctl, unsigned long long *status, unsigned epoch,
                                            Box *__restrict__ part, int *__restrict__ n_out,
                                            unsigned *__restrict__ ghist, unsigned *__restrict__ outbox)
{
    __shared__ int s_tile;
    __shared__ unsigned s_cnt[16], s_base;
    __shared__ Box s_box[4];
    const int tile = lb_take_tile(ctl, &s_tile);
    const int gw = (w + 2) / 3, gh = (h + 2) / 3;
    const int ns = gw * gh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tile == 0) {
        if (ghist)
            for (int i = threadIdx.x; i < 4 * 256; i += 256)
                ghist[i] = 0u;
        if (outbox)
            box_slots_init(outbox);
    }
    float d[4];
    bool ok[4];
    int mm[4], nn[4];
    unsigned rank[4];
    uint32_t col[4];  // the sample's colour, requested with its depth (one round trip instead of two; unused if the depth is invalid)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = tile * TILE + k * 256 + threadIdx.x;
        ok[k] = false;
        d[k] = 0.f;
        mm[k] = nn[k] = 0;
        col[k] = 0u;
        if (s < ns) {
            const int r = s / gw;
            mm[k] = r * 3;
            nn[k] = (s - r * gw) * 3;
            d[k] = depth[(size_t)mm[k] * dstride + (size_t)nn[k]];
            const uint8_t *px = rgb + (size_t)mm[k] * cstride + (size_t)nn[k] * 3;
            col[k] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
        }
        ok[k] = s < ns && depth_valid(d[k]);
        const unsigned long long bal = __ballot(ok[k]);
        rank[k] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_cnt[k * 4 + wave] = __popcll(bal);
    }
    __syncthreads();
    unsigned tot = 0, pre[4];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        if ((s & 3) == wave)
            pre[s >> 2] = tot;
        tot += s_cnt[s];
    }
    // the points are formed before the look-back (their colour loads and the double transform do not need the base)
    Box bx;
    box_init(bx);
    Point pt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!ok[k])
            continue;
        Point &p = pt[k];
        p.z = d[k];
        p.x = ((float)nn[k] - cx) * p.z / fx;  // :124-125, this exact operation order
        p.y = ((float)mm[k] - cy) * p.z / fy;
        p.rgba = col[k];
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
        box_add(bx, p);
    }
    const unsigned base = lb_tile_base(status, tile, epoch, tot, &s_base);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (ok[k])
            out[base + pre[k] + rank[k]] = pt[k];
    box_block_reduce(bx, s_box);
    if (threadIdx.x == 0) {
        part[tile] = bx;
        if (tile == (ns + TILE - 1) / TILE - 1)
            *n_out = (int)(base + tot);
    }
    lb_leave(ctl);
}

// ------------------------------------------------------------------------------------------
// P3 voxel grid
// ------------------------------------------------------------------------------------------
// general path, first kernel: per-workgroup bounding boxes of the whole input + accumulator re-initialisation
__global__ __launch_bounds__(256) void k_vox_minmax(const Point *__restrict__ pts, int n, Box *__restrict__ part,
                                                    CloudState *st, unsigned *__restrict__ ghist,
                                                    unsigned *__restrict__ outbox)
{
    __shared__ Box s_box[4];
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < 4 * 256; i += 256)
            ghist[i] = 0u;
        box_slots_init(outbox);
        if (threadIdx.x == 0)
            st->n_sort = n;
    }
    Box bx;
    box_init(bx);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        box_add(bx, pts[i]);
    box_block_reduce(bx, s_box);
    if (threadIdx.x == 0)
        part[blockIdx.x] = bx;
}

// Every workgroup reduces the partial boxes (plus the box of the resident map on the fast path), evaluates PCL's
// set-up redundantly, computes the voxel index of its tile's points and adds its four 8-bit digit histograms to
// the global ones (the radix passes need them up front).  n is read from the device (`n_dev`).
__global__ __launch_bounds__(256) void k_vox_keys(const Point *__restrict__ pts, const int *__restrict__ n_dev,
                                                  const Box *__restrict__ part, int nparts,
                                                  const unsigned *__restrict__ oldbox, float inv, CloudState *st,
                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                  unsigned *__restrict__ ghist, const Point *__restrict__ old, int K,
                                                  uint32_t *__restrict__ samp, int pts_cap,
                                                  uint32_t *__restrict__ okeys, int okey_groups)
{
    __shared__ Box s_box[4];
    __shared__ unsigned s_h[4 * 256];
    // what does not depend on the set-up is requested first: the tile's points (the buffer holds pts_cap of them, all
    // readable whether or not they are part of this call's input) and this workgroup's share of the sample table
    Point own[4] = {}, smp = {};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = (long long)blockIdx.x * TILE + k * 256 + threadIdx.x;
        if (i < pts_cap)
            own[k] = pts[i];
    }
    const int np = samp ? min(K, MERGE_TABLE) : 0;
    const int sidx = blockIdx.x * 256 + threadIdx.x;  // (the launch has more than MERGE_TABLE threads, or loops below)
    if (sidx < np)
        smp = old[K <= MERGE_TABLE ? sidx : (int)((long long)sidx * K / MERGE_TABLE)];
    // ... and, in the workgroups that index the resident map (below), the first 4096 points of their chunk
    const int og = (int)blockIdx.x - ((int)gridDim.x - okey_groups);
    const int og_chunk = og >= 0 ? ((K + okey_groups - 1) / okey_groups + 255) & ~255 : 0;
    const int og_i0 = og * og_chunk, og_i1 = min(og_i0 + og_chunk, K);
    Point ogp[16];
    if (og >= 0) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = og_i0 + u * 256 + (int)threadIdx.x;
            if (i < og_i1)
                ogp[u] = old[i];
        }
    }
    const int n = *n_dev;
    if ((long long)blockIdx.x * TILE >= n && blockIdx.x != 0 && !samp)
        return;
    Box bx;
    box_init(bx);
    for (int i = threadIdx.x; i < nparts; i += 256)
        box_merge(bx, part[i]);
    box_block_reduce(bx, s_box);
    int nf_sort = bx.nfinite;
    if (oldbox) {  // the resident map: all finite, its box is maintained exactly by the previous call
        Box ob;
        box_slots_fold(oldbox, ob);
        ob.nfinite = 1;
        box_merge(bx, ob);
    }
    VoxSetup vs;
    vox_setup(bx, inv, vs, (blockIdx.x == 0 && threadIdx.x == 0) ? st : nullptr);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->n_sort = n;
        st->nfinite = nf_sort;
    }
    if (vs.overflow)
        return;
    // merge path: the indices of MERGE_TABLE evenly spaced resident points (all of them if the map is smaller), the
    // table k_merge_new brackets its tile with -- one scattered load per thread here, beside this kernel's own loads,
    // instead of the same 1024 scattered loads at the head of every workgroup's chain there (4 us of its 35)
    // merge path, the last okey_groups workgroups of the launch: the index of EVERY resident point under this call's
    // set-up, 4 bytes per point for k_merge_new to rank against (it staged the points themselves until this was added:
    // 16 bytes per point at the ~12 GB/s one workgroup draws, 9 of its 18 us).  These workgroups run beside the ones that
    // index the new points, on compute units the launch would leave idle.
    if (og >= 0) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int i = og_i0 + u * 256 + (int)threadIdx.x;
            if (i < og_i1)
                okeys[i] = vox_key(ogp[u], inv, vs.min_b, vs.mul);
        }
        for (int b = og_i0 + 4096; b < og_i1; b += 256 * 8) {  // (maps beyond ~900 k points: chunks longer than 4096)
            Point pt[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = b + u * 256 + (int)threadIdx.x;
                if (i < og_i1)
                    pt[u] = old[i];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = b + u * 256 + (int)threadIdx.x;
                if (i < og_i1)
                    okeys[i] = vox_key(pt[u], inv, vs.min_b, vs.mul);
            }
        }
    }
    if (samp) {
        if (sidx < np)
            samp[sidx] = vox_key(smp, inv, vs.min_b, vs.mul);
        for (int s = sidx + gridDim.x * 256; s < np; s += gridDim.x * 256)  // (small frames: fewer threads than samples)
            samp[s] = vox_key(old[K <= MERGE_TABLE ? s : (int)((long long)s * K / MERGE_TABLE)], inv, vs.min_b, vs.mul);
        if ((long long)blockIdx.x * TILE >= n && blockIdx.x != 0)
            return;
    }
    for (int i = threadIdx.x; i < 4 * 256; i += 256)
        s_h[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = (long long)blockIdx.x * TILE + k * 256 + threadIdx.x;
        if (i < n) {
            const uint32_t key = vox_key(own[k], inv, vs.min_b, vs.mul);
            keys[i] = key;
            vals[i] = (uint32_t)i;
#pragma unroll
            for (int ps = 0; ps < 4; ps++)
                atomicAdd(&s_h[ps * 256 + ((key >> (8 * ps)) & 255u)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256)
        if (s_h[i])
            atomicAdd(&ghist[i], s_h[i]);
}

// ---- stable LSD radix sort: one launch per 8-bit digit (look-back over tiles, wave-ballot ranking) ---------
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

__global__ __launch_bounds__(256) void k_sort_pass(const uint32_t *__restrict__ keys_in,
                                                   const uint32_t *__restrict__ vals_in,
                                                   uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                   const CloudState *__restrict__ st, int shift,
                                                   const unsigned *__restrict__ ghist /* this pass: [256] */,
                                                   LbCtl *ctl, unsigned long long *status, unsigned epoch)
{
    // h[round][wave][digit]: elements of the tile are ordered (round, wave, lane)
    __shared__ unsigned h[16 * 256];
    __shared__ unsigned s_scan[256];
    __shared__ int s_tile;
    const int tile = lb_take_tile(ctl, &s_tile);
    const int n = st->n_sort;
    const int ntiles = (n + TILE - 1) / TILE;
    if (tile < ntiles && !st->overflow) {
        const long long base = (long long)tile * TILE;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t key[4], val[4];
        bool valid[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = base + k * 256 + threadIdx.x;
            valid[k] = i < n;
            key[k] = valid[k] ? keys_in[i] : 0u;
            val[k] = valid[k] ? vals_in[i] : 0u;
        }
        // a digit every key shares (the high digits of a small grid): the pass is the identity
        const bool trivial = ghist[(keys_in[0] >> shift) & 255u] == (unsigned)n;
        if (trivial) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long i = base + k * 256 + threadIdx.x;
                if (valid[k]) {
                    keys_out[i] = key[k];
                    vals_out[i] = val[k];
                }
            }
        } else {
            for (int i = threadIdx.x; i < 16 * 256; i += 256)
                h[i] = 0;
            __syncthreads();
            unsigned rank[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned d = (key[k] >> shift) & 255u;
                const unsigned long long peers = match_digit(d, valid[k]);
                rank[k] = __popcll(peers & ((1ull << lane) - 1ull));
                if (valid[k] && rank[k] == 0)
                    h[(k * 4 + wave) * 256 + d] = __popcll(peers);
            }
            __syncthreads();
            // thread = digit: exclusive prefix over the 16 (round, wave) slots, tile total, look-back over the
            // earlier tiles, exclusive scan of the global histogram over the digits
            unsigned run = 0;
            for (int s = 0; s < 16; s++) {
                const unsigned c = h[s * 256 + threadIdx.x];
                h[s * 256 + threadIdx.x] = run;
                run += c;
            }
            unsigned long long *slot = status + (size_t)tile * 256 + threadIdx.x;
            unsigned before = 0;
            if (tile > 0) {
                lb_publish(slot, epoch, LB_AGG, run);
                before = lb_exclusive(status, 256, threadIdx.x, tile, epoch);
            }
            lb_publish(slot, epoch, LB_PREFIX, before + run);
            {
                const unsigned g = ghist[threadIdx.x];
                unsigned inc = g;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned t = (unsigned)__shfl_up((int)inc, off, 64);
                    if (lane >= off)
                        inc += t;
                }
                if (lane == 63)
                    s_scan[wave] = inc;
                __syncthreads();
                unsigned woff = 0;
                for (int k = 0; k < wave; k++)
                    woff += s_scan[k];
                __syncthreads();
                s_scan[threadIdx.x] = woff + inc - g + before;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!valid[k])
                    continue;
                const unsigned d = (key[k] >> shift) & 255u;
                const unsigned pos = s_scan[d] + h[(k * 4 + wave) * 256 + d] + rank[k];
                keys_out[pos] = key[k];
                vals_out[pos] = val[k];
            }
        }
    }
    lb_leave(ctl);
}

// voxel centroid of a run of sorted elements [j, ...) sharing `key`, optionally seeded with the resident map's
// point of that voxel (which precedes the new points in PCL's input order): float sums in input order (A7)
__device__ __forceinline__ Point vox_centroid(const Point *__restrict__ pts, const uint32_t *__restrict__ keys,
                                              const uint32_t *__restrict__ vals, int j, int nf, uint32_t key,
                                              const Point *seed)
{
    float sx = 0, sy = 0, sz = 0, sr = 0, sg = 0, sb = 0;
    int cnt = 0;
    if (seed) {
        sx = seed->x; sy = seed->y; sz = seed->z;
        sr = (float)((seed->rgba >> 16) & 255u); sg = (float)((seed->rgba >> 8) & 255u); sb = (float)(seed->rgba & 255u);
        cnt = 1;
    }
    for (; j < nf && keys[j] == key; j++, cnt++) {
        const Point p = pts[vals[j]];
        const float r = (float)((p.rgba >> 16) & 255u), g = (float)((p.rgba >> 8) & 255u), b = (float)(p.rgba & 255u);
        if (cnt == 0) {
            sx = p.x; sy = p.y; sz = p.z; sr = r; sg = g; sb = b;
        } else {
            sx += p.x; sy += p.y; sz += p.z; sr += r; sg += g; sb += b;
        }
    }
    const float c = (float)cnt;
    Point o;
    o.x = sx / c;
    o.y = sy / c;
    o.z = sz / c;
    const int ri = (int)(sr / c), gi = (int)(sg / c), bi = (int)(sb / c);
    o.rgba = (uint32_t)((ri << 16) | (gi << 8) | bi);
    return o;
}

// the same with the run's first new point already loaded (k_merge_new fetches the first points of its four runs
// together, so their latencies overlap); j = index of that first element
__device__ __forceinline__ Point vox_centroid_from(const Point &first, const Point *__restrict__ pts,
                                                   const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                   int j, int nf, uint32_t key, const Point *seed)
{
    float sx, sy, sz, sr, sg, sb;
    int cnt;
    const float r0 = (float)((first.rgba >> 16) & 255u), g0 = (float)((first.rgba >> 8) & 255u), b0 = (float)(first.rgba & 255u);
    if (seed) {
        sx = seed->x; sy = seed->y; sz = seed->z;
        sr = (float)((seed->rgba >> 16) & 255u); sg = (float)((seed->rgba >> 8) & 255u); sb = (float)(seed->rgba & 255u);
        sx += first.x; sy += first.y; sz += first.z; sr += r0; sg += g0; sb += b0;
        cnt = 2;
    } else {
        sx = first.x; sy = first.y; sz = first.z; sr = r0; sg = g0; sb = b0;
        cnt = 1;
    }
    for (j++; j < nf && keys[j] == key; j++, cnt++) {
        const Point p = pts[vals[j]];
        sx += p.x; sy += p.y; sz += p.z;
        sr += (float)((p.rgba >> 16) & 255u); sg += (float)((p.rgba >> 8) & 255u); sb += (float)(p.rgba & 255u);
    }
    const float c = (float)cnt;
    Point o;
    o.x = sx / c;
    o.y = sy / c;
    o.z = sz / c;
    const int ri = (int)(sr / c), gi = (int)(sg / c), bi = (int)(sb / c);
    o.rgba = (uint32_t)((ri << 16) | (gi << 8) | bi);
    return o;
}

// tile-local exclusive scan of one flag per element in (round, wave, lane) order; returns the tile total
__device__ __forceinline__ unsigned tile_flag_scan(const bool f[4], unsigned excl[4], unsigned *s_cnt /* [16] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long bal = __ballot(f[k]);
        excl[k] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_cnt[k * 4 + wave] = __popcll(bal);
    }
    __syncthreads();
    unsigned tot = 0;
#pragma unroll
    for (int s = 0; s < 16; s++) {
        if ((s & 3) == wave)
            excl[s >> 2] += tot;
        tot += s_cnt[s];
    }
    return tot;
}

// general path, last kernel: run heads of the sorted keys -> output positions (look-back) -> one thread per voxel
// sums its run; PCL's overflow rule copies the input instead.  Also the bounding box of the output (the next
// insert's set-up needs the box of the resident map).
__global__ __launch_bounds__(256) void k_vox_reduce(const Point *__restrict__ pts, int n_in,
                                                    const uint32_t *__restrict__ keys,
                                                    const uint32_t *__restrict__ vals, CloudState *st,
                                                    Point *__restrict__ out, LbCtl *ctl, unsigned long long *status,
                                                    unsigned epoch, unsigned *__restrict__ outbox)
{
    __shared__ int s_tile;
    __shared__ unsigned s_cnt[16], s_base;
    __shared__ Box s_box[4];
    const int tile = lb_take_tile(ctl, &s_tile);
    const int ntiles = (n_in + TILE - 1) / TILE;
    if (st->overflow) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = (long long)tile * TILE + k * 256 + threadIdx.x;
            if (i < n_in)
                out[i] = pts[i];  // PCL: "Leaf size is too small" -> output = input
        }
        if (tile == 0 && threadIdx.x == 0)
            st->nout = n_in;
    } else {
        const int nf = st->nfinite;
        bool head[4];
        unsigned excl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = (long long)tile * TILE + k * 256 + threadIdx.x;
            head[k] = i < nf && (i == 0 || keys[i] != keys[i - 1]);
        }
        const unsigned tot = tile_flag_scan(head, excl, s_cnt);
        const unsigned base = lb_tile_base(status, tile, epoch, tot, &s_base);
        Box bx;
        box_init(bx);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!head[k])
                continue;
            const int i = tile * TILE + k * 256 + threadIdx.x;
            const Point o = vox_centroid(pts, keys, vals, i, nf, keys[i], nullptr);
            out[base + excl[k]] = o;
            box_add(bx, o);
        }
        box_block_reduce(bx, s_box);
        if (threadIdx.x == 0) {
            box_slots_atomic(outbox, bx);
            if (tile == ntiles - 1)
                st->nout = (int)(base + tot);
        }
    }
    lb_leave(ctl);
}

// ---- fast path: merge the sorted new points into the resident (sorted, one point per voxel) map ----------------
__device__ __forceinline__ int old_upper_bound(const Point *__restrict__ old, int lo, int hi, uint32_t key, float inv,
                                               const int min_b[3], const int mul[3])
{
    while (lo < hi) {  // first index in [lo, hi) whose key is > key (hi if none)
        const int mid = (lo + hi) >> 1;
        if (vox_key(old[mid], inv, min_b, mul) <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
template <typename A> __device__ __forceinline__ int key_lower_bound(const A *a, int lo, int hi, uint32_t key)
{
    while (lo < hi) {  // first index in [lo, hi) whose key is >= key
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int lds_count_le(const uint32_t *a, int n, uint32_t key)
{
    int lo = 0, hi = n;  // number of entries <= key in the ascending array a[0..n)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

constexpr int MN_TILE = 256;     // sorted new keys per workgroup (one per thread)
constexpr int MN_CAP = 24576;    // resident indices a workgroup stages in LDS (96 KB)
constexpr int MN_SAMPLES = 1024; // fall-back when a tile's range is longer than that: samples of the range
constexpr int MN_LOADS = 16;     // loads a thread has in flight while it stages a range

// new-point side: every run of equal keys among the sorted new points becomes one output voxel -- merged with the
// resident map's point of that voxel if there is one (it comes first in PCL's input order), a new voxel otherwise.
// cexcl[j] = new voxels opened by runs before j (look-back); a run's output position is its rank among the
// resident points plus that count.
//
// The rank (number of resident points with index <= key) without a chain of dependent loads from memory -- inside a
// kernel every one of those is a 1 - 1.5 us round trip, whichever cache level answers (measured with s_memrealtime
// stamps; until round 4 this kernel had about twelve of them in a row and took 35 - 39 us):
//   * `table` (k_vox_keys) holds the indices of 4096 evenly spaced resident points: one coalesced load to LDS;
//   * the tile's first and last key, looked up there, bracket the range of resident points its ranks lie in: with
//     256 keys per workgroup that is ~6 k points of a 700 k map;
//   * the workgroup streams that whole range (coalesced 16-byte loads, many in flight), keeps the indices in LDS and
//     ranks its keys there.  A range longer than MN_CAP (new points sparse against a dense map) is sampled a second time
//     instead and the last levels are a search in global memory, as before.
// Side job: tile_lb[b] = first sorted new key >= the index of resident point b*1024, which brackets the new keys
// k_merge_old's workgroup b has to look at.  It is the work of the workgroups BEHIND the tiles of new points (the
// launch adds merge_side_groups(K) of them): a 15-deep chain of its own that runs beside the tiles' chain.
__global__ __launch_bounds__(256) void k_merge_new(const Point *__restrict__ old, const uint32_t *__restrict__ okeys, int K,
                                                   const Point *__restrict__ newp, const uint32_t *__restrict__ skeys,
                                                   const uint32_t *__restrict__ svals, CloudState *st,
                                                   const uint32_t *__restrict__ table, int cap,
                                                   unsigned *__restrict__ cexcl, int *__restrict__ tile_lb,
                                                   Point *__restrict__ out, LbCtl *ctl, unsigned long long *status,
                                                   unsigned epoch, unsigned *__restrict__ outbox)
{
    extern __shared__ uint32_t s_old[];  // max(cap, MERGE_TABLE) indices
    __shared__ int s_tile, s_lo, s_hi;
    __shared__ unsigned s_cnt[4], s_base;
    __shared__ Box s_box[4];
    const int tile = lb_take_tile(ctl, &s_tile);
    const int nf = st->nfinite;
    const int ntiles = max((nf + MN_TILE - 1) / MN_TILE, 1);
    const bool overflow = st->overflow != 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tile < ntiles && !overflow) {
        const int j = tile * MN_TILE + (int)threadIdx.x;
        const int jlast = min(tile * MN_TILE + MN_TILE, nf) - 1;  // the tile's last element
        uint32_t key = 0;
        bool head = false, opens = false;
        int rnk = 0;  // rank; negative: the run merges into resident point -rnk - 1
        if (j < nf) {
            key = skeys[j];
            head = j == 0 || skeys[j - 1] != key;
        }
        if (K > 0 && nf > 0) {
            // level A: the table of the whole resident map
            const int npa = min(K, MERGE_TABLE);
            for (int s = threadIdx.x; s < npa; s += 256)
                s_old[s] = table[s];
            __syncthreads();
            if (threadIdx.x == 0) {  // (holds the tile's first key)
                // the last table entry <= the first key STAYS in the range: it may be the resident point of that voxel
                const int c = lds_count_le(s_old, npa, key);
                s_lo = c == 0 ? 0 : (K <= MERGE_TABLE ? c - 1 : (int)((long long)(c - 1) * K / MERGE_TABLE));
            }
            if (j == jlast) {  // the thread that holds the tile's last key
                const int c = lds_count_le(s_old, npa, key);
                s_hi = c == npa ? K : (K <= MERGE_TABLE ? c : (int)((long long)c * K / MERGE_TABLE));
            }
            __syncthreads();
            const int lo = s_lo, hi = max(s_hi, s_lo), len = hi - lo;
            int r = 0;           // resident points with index <= key
            bool exists = false;  // ... the last of them has this index
            if (len <= cap) {
                // the range itself: the indices of old[lo, hi) (k_vox_keys) to LDS
                for (int s0 = 0; s0 < len; s0 += 256 * MN_LOADS) {
                    uint32_t kk[MN_LOADS];
#pragma unroll
                    for (int u = 0; u < MN_LOADS; u++) {
                        const int sidx = s0 + u * 256 + (int)threadIdx.x;
                        if (sidx < len)
                            kk[u] = okeys[lo + sidx];
                    }
#pragma unroll
                    for (int u = 0; u < MN_LOADS; u++) {
                        const int sidx = s0 + u * 256 + (int)threadIdx.x;
                        if (sidx < len)
                            s_old[sidx] = kk[u];
                    }
                }
                __syncthreads();
                if (head) {
                    const int c = lds_count_le(s_old, len, key);
                    r = lo + c;  // (every point below lo is smaller than the tile's first key)
                    exists = c > 0 && s_old[c - 1] == key;
                }
            } else {
                // a long range: MN_SAMPLES samples of it, then the last levels in global memory
                for (int s = threadIdx.x; s < MN_SAMPLES; s += 256)
                    s_old[s] = okeys[lo + (int)((long long)s * len / MN_SAMPLES)];
                __syncthreads();
                if (head) {
                    const int c = lds_count_le(s_old, MN_SAMPLES, key);
                    int l = c == 0 ? lo : lo + (int)((long long)(c - 1) * len / MN_SAMPLES) + 1;
                    int h = max(l, c == MN_SAMPLES ? hi : lo + (int)((long long)c * len / MN_SAMPLES));
                    while (l < h) {  // first index in [l, h) whose key is > key (h if none)
                        const int mid = (l + h) >> 1;
                        if (okeys[mid] <= key)
                            l = mid + 1;
                        else
                            h = mid;
                    }
                    r = l;
                    exists = l > 0 && okeys[l - 1] == key;
                }
            }
            if (head) {
                opens = !exists;
                rnk = exists ? -r : r;
            }
        } else
            opens = head;
        // value index, first new point, seed: nothing here depends on the look-back
        Box bx;
        box_init(bx);
        Point o;
        if (head) {
            const Point first = newp[svals[j]];
            Point seed;
            if (rnk < 0)
                seed = old[-rnk - 1];
            o = vox_centroid_from(first, newp, skeys, svals, j, nf, key, rnk < 0 ? &seed : nullptr);
            box_add(bx, o);
        }
        // new voxels opened before j: in the wave, in the tile, in the earlier tiles
        const unsigned long long bal = __ballot(opens);
        unsigned excl = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0)
            s_cnt[wave] = __popcll(bal);
        __syncthreads();
        unsigned tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w == wave)
                excl += tot;
            tot += s_cnt[w];
        }
        const unsigned base = lb_tile_base(status, tile, epoch, tot, &s_base);
        if (j < nf)
            cexcl[j] = base + excl;
        if (head)
            out[(rnk < 0 ? -rnk - 1 : rnk) + (int)(base + excl)] = o;
        box_block_reduce(bx, s_box);
        if (threadIdx.x == 0) {
            box_slots_atomic(outbox, bx);
            if (tile == ntiles - 1) {
                cexcl[nf] = base + tot;
                st->nout = K + (int)(base + tot);
            }
        }
    } else if (!overflow) {
        // side job (the workgroups behind the tiles): the bracket of every map tile
        const int ntk = (K + TILE - 1) / TILE, nside = (int)gridDim.x - ntiles;
        for (int b = (tile - ntiles) * 256 + (int)threadIdx.x; b < ntk; b += nside * 256)
            tile_lb[b] = key_lower_bound(skeys, 0, nf, okeys[(size_t)b * TILE]);
    }
    lb_leave(ctl);
}

// workgroups k_merge_new is launched with beyond the tiles of new points: one bracket search per thread up to 16 k map
// tiles (a 16 M point map), more per thread beyond
static inline int merge_side_groups(long long K) { return (int)std::min<long long>(std::max<long long>((tiles_of_ll(K) + 255) / 256, 1), 64); }

constexpr int MERGE_OLD_LDS = 2048;

// resident side: a point whose voxel received no new point moves up by the number of new voxels that sort before
// it.  One pass: K*16 B in, (K - touched)*16 B out.  The sorted new keys a workgroup can meet are bracketed by
// tile_lb[b], tile_lb[b+1] and staged in LDS.  Also checks that the resident map really is strictly increasing
// under the new keys (a centroid rounded onto a voxel boundary, or a map left unfiltered by the overflow rule, is
// not): the host then redoes the key frame through the general path.
__global__ __launch_bounds__(256) void k_merge_old(const Point *__restrict__ old, int K,
                                                   const uint32_t *__restrict__ skeys,
                                                   const unsigned *__restrict__ cexcl,
                                                   const int *__restrict__ tile_lb, float inv, CloudState *st,
                                                   Point *__restrict__ out, unsigned *__restrict__ outbox)
{
    __shared__ Box s_box[4];
    __shared__ uint32_t s_keys[MERGE_OLD_LDS];
    __shared__ unsigned s_cex[MERGE_OLD_LDS + 1];
    if (st->overflow)
        return;
    const int nf = st->nfinite;
    int min_b[3], mul[3];
    for (int a = 0; a < 3; a++) {
        min_b[a] = st->min_b[a];
        mul[a] = st->mul[a];
    }
    const int ntk = (K + TILE - 1) / TILE;
    const int i0 = blockIdx.x * TILE;
    // the tile's points are requested first -- and, by the first lane of every wave, the point in front of the wave's
    // first one, for the ordering check --: their latency covers the bracket / staging chain below
    const int lane = threadIdx.x & 63;
    Point p4[4] = {}, pv4[4] = {};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k * 256 + threadIdx.x;
        if (i < K)
            p4[k] = old[i];
        if (lane == 0 && i > 0 && i < K)
            pv4[k] = old[i - 1];
    }
    const int lo = min(max(tile_lb[blockIdx.x], 0), nf);
    const int hi = min(max((int)blockIdx.x + 1 < ntk ? tile_lb[blockIdx.x + 1] : nf, lo), nf);
    const int len = hi - lo;
    const bool staged = len <= MERGE_OLD_LDS;
    if (staged) {  // the new keys the tile can meet and the new-voxel counts in front of them (one more: lb may be hi)
        for (int s = threadIdx.x; s < len; s += 256)
            s_keys[s] = skeys[lo + s];
        for (int s = threadIdx.x; s <= len; s += 256)
            s_cex[s] = cexcl[lo + s];
    }
    __syncthreads();
    Box bx;
    box_init(bx);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k * 256 + threadIdx.x;
        const Point p = p4[k];
        uint32_t key = 0;
        if (i < K)
            key = vox_key(p, inv, min_b, mul);
        uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
        if (i >= K)
            continue;
        if (lane == 0 && i > 0)
            prev = vox_key(pv4[k], inv, min_b, mul);
        if (i > 0 && prev >= key)
            bad = true;
        int lb;
        bool touched;
        unsigned opened;  // new voxels in front of this point
        if (staged) {
            const int c = key_lower_bound(s_keys, 0, len, key);
            lb = lo + c;
            touched = c < len ? s_keys[c] == key : (lb < nf && skeys[lb] == key);
            opened = s_cex[c];
        } else {
            lb = key_lower_bound(skeys, lo, hi, key);
            touched = lb < nf && skeys[lb] == key;
            opened = touched ? 0u : cexcl[lb];
        }
        if (touched)
            continue;  // merged by k_merge_new
        out[i + (int)opened] = p;
        box_add(bx, p);
    }
    if (__any(bad) && lane == 0)
        st->unsorted = 1;
    box_block_reduce(bx, s_box);
    if (threadIdx.x == 0)
        box_slots_atomic(outbox, bx);
}

// ------------------------------------------------------------------------------------------
// StatisticalOutlierRemoval: the mean distance of every point to its mean_k nearest neighbours
// (PointCloudMap.cc:46-47, 283-285; PCL 1.7 filters/impl/statistical_outlier_removal.hpp).
//
// The exact kd-tree search becomes an exact search on a uniform grid: points sorted by cell index (the radix sort
// above), so the points of a run of cells along x are one contiguous range found with two binary searches.  One wave
// per query point scans the (2S+1)^3 block of cells around its own cell, S = 1, 2, ..., keeping the mean_k + 1
// smallest squared distances (only the distances matter); once the block holds mean_k + 1 points whose farthest is
// nearer than S cells, nothing outside can be nearer and the search ends.  Squared distances are FLANN's
// L2_Simple<float> ((dx*dx + dy*dy) + dz*dz, no contraction), the mean is summed in double in ascending order like
// PCL's loop over the sorted neighbour list.
// ------------------------------------------------------------------------------------------
struct SorGrid {
    float mn[3];
    float h, inv_h;
    int dim[3];
};

constexpr int SOR_MAX_K = 63;  // mean_k + 1 distances fit one wave
constexpr int SOR_BUF = 128;   // candidate distances a wave holds between selections

__global__ __launch_bounds__(256) void k_sor_box(const Point *__restrict__ pts, int n, unsigned *__restrict__ box6,
                                                 int *__restrict__ nfinite)
{
    __shared__ Box s_box[4];
    Box bx;
    box_init(bx);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        box_add(bx, pts[i]);
    box_block_reduce(bx, s_box);
    if (threadIdx.x == 0) {
        box_atomic(box6, bx);
        if (bx.nfinite)
            atomicAdd(nfinite, bx.nfinite);
    }
}

__device__ __forceinline__ uint32_t sor_key(const Point &p, const SorGrid &g)
{
    if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z))
        return 0xFFFFFFFFu;
    const int cx = min(max((int)floorf((p.x - g.mn[0]) * g.inv_h), 0), g.dim[0] - 1);
    const int cy = min(max((int)floorf((p.y - g.mn[1]) * g.inv_h), 0), g.dim[1] - 1);
    const int cz = min(max((int)floorf((p.z - g.mn[2]) * g.inv_h), 0), g.dim[2] - 1);
    return (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
}

__global__ __launch_bounds__(256) void k_sor_keys(const Point *__restrict__ pts, int n, SorGrid g,
                                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                  unsigned *__restrict__ ghist, CloudState *__restrict__ st)
{
    __shared__ unsigned s_h[4 * 256];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->n_sort = n;
        st->overflow = 0;
    }
    for (int i = threadIdx.x; i < 4 * 256; i += 256)
        s_h[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long long i = (long long)blockIdx.x * TILE + k * 256 + threadIdx.x;
        if (i < n) {
            const uint32_t key = sor_key(pts[i], g);
            keys[i] = key;
            vals[i] = (uint32_t)i;
#pragma unroll
            for (int ps = 0; ps < 4; ps++)
                atomicAdd(&s_h[ps * 256 + ((key >> (8 * ps)) & 255u)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256)
        if (s_h[i])
            atomicAdd(&ghist[i], s_h[i]);
}

// occupied cells (distinct keys among the first nfinite sorted keys), for the choice of the cell size
__global__ __launch_bounds__(256) void k_sor_count_cells(const uint32_t *__restrict__ keys, int nfinite,
                                                         int *__restrict__ cells)
{
    int c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nfinite; i += (long long)gridDim.x * 256)
        c += (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    c = wave_reduce_add(c);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(cells, c);
}

__global__ __launch_bounds__(256) void k_sor_gather(const Point *__restrict__ pts, const uint32_t *__restrict__ vals,
                                                    int n, float4 *__restrict__ sorted)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        const Point p = pts[vals[r]];
        sorted[r] = make_float4(p.x, p.y, p.z, 0.f);
    }
}

// ascending bitonic sort of the SOR_BUF floats of one wave's LDS buffer (two elements per lane and step)
__device__ __forceinline__ void sor_sort(float *buf, int lane)
{
    for (int k = 2; k <= SOR_BUF; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int t = lane + 64 * half;
                const int o = t ^ j;
                if (o > t) {
                    const float a = buf[t], b = buf[o];
                    const bool up = (t & k) == 0;
                    if ((a > b) == up) {
                        buf[t] = b;
                        buf[o] = a;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_sor_knn(const float4 *__restrict__ sorted, const uint32_t *__restrict__ keys,
                                                 const uint32_t *__restrict__ vals, int n, int nfinite, SorGrid g,
                                                 int mean_k, float *__restrict__ mean_dist)
{
    __shared__ float s_buf[4][SOR_BUF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 4 + wave;  // query = r-th point in cell order
    if (r >= n)
        return;
    if (r >= nfinite) {  // non-finite points sort to the end: distance 0, no neighbours (PCL: `continue`)
        if (lane == 0)
            mean_dist[vals[r]] = 0.0f;
        return;
    }
    float *buf = s_buf[wave];
    const float4 q = sorted[r];
    const uint32_t key = keys[r];
    const int cx = (int)(key % (uint32_t)g.dim[0]), cy = (int)((key / (uint32_t)g.dim[0]) % (uint32_t)g.dim[1]),
              cz = (int)(key / ((uint32_t)g.dim[0] * (uint32_t)g.dim[1]));
    const int need = mean_k + 1;
    for (int S = 1;; S++) {
        const int x0 = max(cx - S, 0), x1 = min(cx + S, g.dim[0] - 1);
        const int y0 = max(cy - S, 0), y1 = min(cy + S, g.dim[1] - 1);
        const int z0 = max(cz - S, 0), z1 = min(cz + S, g.dim[2] - 1);
        const int ny = y1 - y0 + 1, nruns = ny * (z1 - z0 + 1);
        int cnt = 0;              // distances in buf (wave-uniform)
        float tau = INFINITY;     // nothing above it can be among the `need` smallest
        for (int run0 = 0; run0 < nruns; run0 += 32) {
            // lanes 2i / 2i+1: first / one-past-last sorted position of run run0 + i (the cells x0..x1 of one (y, z))
            const int ri = run0 + (lane >> 1);
            int pos = 0;
            if (ri < nruns) {
                const int yy = y0 + ri % ny, zz = z0 + ri / ny;
                const uint32_t base = (uint32_t)((zz * g.dim[1] + yy) * g.dim[0]);
                const uint32_t target = (lane & 1) ? base + (uint32_t)x1 : base + (uint32_t)x0;
                int lo = 0, hi = nfinite;  // lower_bound (even lanes) / upper_bound (odd lanes)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const uint32_t km = keys[mid];
                    if ((lane & 1) ? km <= target : km < target)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                pos = lo;
            }
            const int nr = min(32, nruns - run0);
            for (int i = 0; i < nr; i++) {
                const int beg = __shfl(pos, 2 * i, 64), end = __shfl(pos, 2 * i + 1, 64);
                for (int j0 = beg; j0 < end; j0 += 64) {
                    const int j = j0 + lane;
                    float d2 = INFINITY;
                    if (j < end) {
                        const float4 p = sorted[j];
                        const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
                        d2 = dx * dx;
                        d2 += dy * dy;
                        d2 += dz * dz;
                    }
                    const bool keep = j < end && d2 <= tau;
                    const unsigned long long m = __ballot(keep);
                    if (keep)
                        buf[cnt + __popcll(m & ((1ull << lane) - 1ull))] = d2;
                    cnt += __popcll(m);
                    if (cnt > SOR_BUF - 64) {  // make room for the next 64: keep the `need` smallest
                        for (int t = cnt + lane; t < SOR_BUF; t += 64)
                            buf[t] = INFINITY;
                        sor_sort(buf, lane);
                        cnt = min(cnt, need);
                        if (cnt == need)
                            tau = buf[need - 1];
                    }
                }
            }
        }
        for (int t = cnt + lane; t < SOR_BUF; t += 64)
            buf[t] = INFINITY;
        sor_sort(buf, lane);
        const bool covered = x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dim[0] - 1 && y1 == g.dim[1] - 1 && z1 == g.dim[2] - 1;
        // a point outside the block is at least S cells away along one axis (0.1 % slack for the rounding of the cell
        // coordinates)
        const float reach = (float)S * g.h * 0.999f;
        if (covered || (cnt >= need && buf[need - 1] <= reach * reach))
            break;
    }
    if (lane == 0) {
        double dist_sum = 0.0;
        for (int k = 1; k < need; k++)  // entry 0 is the query point itself (distance 0)
            dist_sum += sqrt((double)buf[k]);
        mean_dist[vals[r]] = (float)(dist_sum / (double)mean_k);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct VoxelWorkspace {
    DevBuf keys[2], vals[2], status, part, cexcl, tile_lb, ctl, state, ghist, table, okeys;
    unsigned epoch = 0;
    bool armed = false;
    int reserve(long long n_sort, int nparts)
    {
        const long long nt = std::max<long long>((n_sort + TILE - 1) / TILE, 1);
        int rc;
        for (int k = 0; k < 2; k++) {
            if ((rc = keys[k].reserve(sizeof(uint32_t) * (size_t)std::max<long long>(n_sort, 1))) != ORBGPU_OK)
                return rc;
            if ((rc = vals[k].reserve(sizeof(uint32_t) * (size_t)std::max<long long>(n_sort, 1))) != ORBGPU_OK)
                return rc;
        }
        // status words: 256 channels per tile for the radix passes; never cleared (epoch-tagged), but fresh
        // allocations must not alias a live epoch: zero them
        const size_t sb = sizeof(unsigned long long) * 256 * (size_t)nt;
        if (sb > status.bytes) {
            if ((rc = status.reserve(sb)) != ORBGPU_OK)
                return rc;
            ORBGPU_HIP_TRY(hipMemset(status.p, 0, status.bytes));
            ORBGPU_HIP_TRY(hipDeviceSynchronize());  // the work streams are non-blocking: not ordered behind this
        }
        if ((rc = part.reserve(sizeof(Box) * (size_t)std::max(nparts, 1))) != ORBGPU_OK)
            return rc;
        if ((rc = cexcl.reserve(sizeof(unsigned) * (size_t)(n_sort + 1))) != ORBGPU_OK)
            return rc;
        if ((rc = state.reserve(sizeof(CloudState))) != ORBGPU_OK || (rc = ghist.reserve(sizeof(unsigned) * 4 * 256)) != ORBGPU_OK ||
            (rc = table.reserve(sizeof(uint32_t) * MERGE_TABLE)) != ORBGPU_OK)
            return rc;
        if (!armed) {
            if ((rc = ctl.reserve(sizeof(LbCtl))) != ORBGPU_OK)
                return rc;
            ORBGPU_HIP_TRY(hipMemset(ctl.p, 0, sizeof(LbCtl)));
            ORBGPU_HIP_TRY(hipMemset(state.p, 0, sizeof(CloudState)));
            ORBGPU_HIP_TRY(hipDeviceSynchronize());
            epoch = 0;
            armed = true;
        }
        return ORBGPU_OK;
    }
    unsigned next_epoch()
    {
        if (++epoch == 0)  // 2^32 launches later: stale words could match again -> start over with clean words
            (void)hipMemset(status.p, 0, status.bytes), (void)hipDeviceSynchronize(), epoch = 1;
        return epoch;
    }
    void release()
    {
        for (int k = 0; k < 2; k++) {
            keys[k].release();
            vals[k].release();
        }
        status.release(), part.release(), cexcl.release(), tile_lb.release(), ctl.release(), state.release(), ghist.release(), table.release(), okeys.release();
        armed = false;
    }
};

static inline int tiles_of(long long n) { return (int)((n + TILE - 1) / TILE); }

// the four digit passes over ws.keys/vals[0] (result back in [0]); `grid_tiles` >= tiles of the device-side count
static void radix_sort_device(VoxelWorkspace &ws, int grid_tiles, hipStream_t st)
{
    uint32_t *k0 = ws.keys[0].as<uint32_t>(), *k1 = ws.keys[1].as<uint32_t>();
    uint32_t *v0 = ws.vals[0].as<uint32_t>(), *v1 = ws.vals[1].as<uint32_t>();
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(k_sort_pass, dim3(grid_tiles), dim3(256), 0, st, k0, v0, k1, v1, ws.state.as<CloudState>(),
                           pass * 8, ws.ghist.as<unsigned>() + pass * 256, ws.ctl.as<LbCtl>(),
                           ws.status.as<unsigned long long>(), ws.next_epoch());
        std::swap(k0, k1);
        std::swap(v0, v1);
    }
}

// General path: in[0..n) -> out (capacity >= n): 7 launches.  The output count lands in ws.state->nout, the
// bounding box of the output in the slots of outbox (BOX_WORDS words).
static int voxel_filter_device(VoxelWorkspace &ws, const Point *in, long long n, float leaf, Point *out,
                               unsigned *outbox, hipStream_t st)
{
    ORBGPU_REQUIRE(n < (1ll << 30), "too many points for one voxel filter (%lld)", n);
    const int nb = (int)std::min<long long>(std::max<long long>((n + 255) / 256, 1), 256);
    int rc = ws.reserve(n, nb);
    if (rc != ORBGPU_OK)
        return rc;
    CloudState *S = ws.state.as<CloudState>();
    const float inv = 1.0f / leaf;  // Eigen::Array4f::Ones() / leaf_size_
    if (n == 0) {
        ORBGPU_HIP_TRY(hipMemsetAsync(S, 0, sizeof(CloudState), st));
        return ORBGPU_OK;
    }
    const int nt = tiles_of(n);
    hipLaunchKernelGGL(k_vox_minmax, dim3(nb), dim3(256), 0, st, in, (int)n, ws.part.as<Box>(), S, ws.ghist.as<unsigned>(),
                       outbox);
    hipLaunchKernelGGL(k_vox_keys, dim3(nt), dim3(256), 0, st, in, &S->n_sort, ws.part.as<Box>(), nb,
                       (const unsigned *)nullptr, inv, S, ws.keys[0].as<uint32_t>(), ws.vals[0].as<uint32_t>(),
                       ws.ghist.as<unsigned>(), (const Point *)nullptr, 0, (uint32_t *)nullptr, (int)n, (uint32_t *)nullptr, 0);
    radix_sort_device(ws, nt, st);
    hipLaunchKernelGGL(k_vox_reduce, dim3(nt), dim3(256), 0, st, in, (int)n, ws.keys[0].as<uint32_t>(),
                       ws.vals[0].as<uint32_t>(), S, out, ws.ctl.as<LbCtl>(), ws.status.as<unsigned long long>(),
                       ws.next_epoch(), outbox);
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
    DevBuf depth, rgb, status, part, ctl;
    unsigned epoch = 0;
    bool armed = false;
    int reserve_lb(int ntiles)
    {
        int rc;
        const size_t sb = sizeof(unsigned long long) * (size_t)ntiles;
        if (sb > status.bytes) {
            if ((rc = status.reserve(sb)) != ORBGPU_OK)
                return rc;
            ORBGPU_HIP_TRY(hipMemset(status.p, 0, status.bytes));
            ORBGPU_HIP_TRY(hipDeviceSynchronize());
        }
        if ((rc = part.reserve(sizeof(Box) * (size_t)ntiles)) != ORBGPU_OK)
            return rc;
        if (!armed) {
            if ((rc = ctl.reserve(sizeof(LbCtl))) != ORBGPU_OK)
                return rc;
            ORBGPU_HIP_TRY(hipMemset(ctl.p, 0, sizeof(LbCtl)));
            ORBGPU_HIP_TRY(hipDeviceSynchronize());
            armed = true;
        }
        return ORBGPU_OK;
    }
    void release()
    {
        depth.release(), rgb.release(), status.release(), part.release(), ctl.release();
        armed = false;
    }
};

static inline int bp_tiles(int w, int h) { return tiles_of((long long)((w + 2) / 3) * ((h + 2) / 3)); }

// back-project device images (+ transform) into out[0..count); *d_count = count.  One launch.
static int backproject_launch(FrameStage &fs, const float *d_depth, size_t dstride, const uint8_t *d_rgb,
                              size_t cstride, int w, int h, float fx, float fy, float cx, float cy, const float *Tcw,
                              Point *out, int *d_count, unsigned *ghist, unsigned *outbox, hipStream_t st)
{
    const int nt = bp_tiles(w, h);
    int rc = fs.reserve_lb(nt);
    if (rc != ORBGPU_OK)
        return rc;
    Pose P;
    P.apply = 0;
    if (Tcw)
        pose_inverse(Tcw, P);
    if (++fs.epoch == 0) {
        ORBGPU_HIP_TRY(hipMemsetAsync(fs.status.p, 0, fs.status.bytes, st));
        fs.epoch = 1;
    }
    hipLaunchKernelGGL(k_bp, dim3(nt), dim3(256), 0, st, d_depth, dstride, d_rgb, cstride, w, h, fx, fy, cx, cy, P, out,
                       fs.ctl.as<LbCtl>(), fs.status.as<unsigned long long>(), fs.epoch, fs.part.as<Box>(), d_count,
                       ghist, outbox);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

// host images -> staging buffers of the frame stage
static int upload_frame(FrameStage &fs, const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride, int w,
                        int h, hipStream_t st)
{
    int rc;
    if ((rc = fs.depth.reserve(sizeof(float) * (size_t)w * h)) != ORBGPU_OK ||
        (rc = fs.rgb.reserve((size_t)w * 3 * h)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpy2DAsync(fs.depth.p, sizeof(float) * w, depth, sizeof(float) * dstride, sizeof(float) * w, h,
                                    hipMemcpyHostToDevice, st));
    ORBGPU_HIP_TRY(hipMemcpy2DAsync(fs.rgb.p, (size_t)w * 3, rgb, cstride, (size_t)w * 3, h, hipMemcpyHostToDevice, st));
    return ORBGPU_OK;
}

// ---- StatisticalOutlierRemoval on device points: mean neighbour distance of every point -> d_mean_dist[n] ----------
// Cell size: a pilot pass at a coarse cell size counts the occupied cells; the clouds this runs on are surfaces, where
// the points per cell grow with h^2, so h is rescaled to about SOR_CELL_POINTS points per cell (any h is exact, the
// choice only decides how many candidates a query scans and how often it needs a second shell).
constexpr double SOR_CELL_POINTS = 8.0;

static int sor_grid_for(const float mn[3], const float mx[3], double h, SorGrid &g)
{
    for (;;) {
        double cells = 1.0;
        for (int a = 0; a < 3; a++) {
            g.mn[a] = mn[a];
            const double d = std::floor(((double)mx[a] - (double)mn[a]) / h) + 1.0;
            g.dim[a] = (int)std::min(d, 2.0e9);
            cells *= d;
        }
        if (cells < 4.0e9)
            break;
        h *= 1.5;  // the cell index must fit 32 bits
    }
    g.h = (float)h;
    g.inv_h = 1.0f / g.h;
    return ORBGPU_OK;
}

static int sor_sort_cells(VoxelWorkspace &ws, const Point *d_pts, long long n, const SorGrid &g, hipStream_t st)
{
    const int nt = tiles_of(n);
    ORBGPU_HIP_TRY(hipMemsetAsync(ws.ghist.p, 0, sizeof(unsigned) * 4 * 256, st));
    hipLaunchKernelGGL(k_sor_keys, dim3(nt), dim3(256), 0, st, d_pts, (int)n, g, ws.keys[0].as<uint32_t>(),
                       ws.vals[0].as<uint32_t>(), ws.ghist.as<unsigned>(), ws.state.as<CloudState>());
    radix_sort_device(ws, nt, st);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

static int sor_mean_distances(VoxelWorkspace &ws, const Point *d_pts, long long n, int mean_k, float *d_mean_dist,
                              long long *nfinite_out, hipStream_t st)
{
    ORBGPU_REQUIRE(n >= 1 && n < (1ll << 30), "statistical outlier removal: bad point count %lld", n);
    ORBGPU_REQUIRE(mean_k >= 1 && mean_k <= SOR_MAX_K, "mean_k must be in [1, %d]", SOR_MAX_K);
    int rc = ws.reserve(n, 1);
    if (rc != ORBGPU_OK)
        return rc;
    DevBuf scratch, sorted;
    auto done = [&](int code) {
        scratch.release();
        sorted.release();
        return code;
    };
    if ((rc = scratch.reserve(sizeof(unsigned) * 8)) != ORBGPU_OK || (rc = sorted.reserve(sizeof(float4) * (size_t)n)) != ORBGPU_OK)
        return done(rc);
    // bounding box and count of the finite points
    unsigned hbox[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};  // mn[3], mx[3], nfinite, cells
    hipError_t e = hipMemcpyAsync(scratch.p, hbox, sizeof(hbox), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const int nb = (int)std::min<long long>((n + 255) / 256, 1024);
        hipLaunchKernelGGL(k_sor_box, dim3(nb), dim3(256), 0, st, d_pts, (int)n, scratch.as<unsigned>(),
                           reinterpret_cast<int *>(scratch.as<unsigned>() + 6));
        e = hipMemcpyAsync(hbox, scratch.p, sizeof(hbox), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error("statistical outlier removal: %s", hipGetErrorString(e));
        return done(ORBGPU_EHIP);
    }
    const long long nfinite = (long long)hbox[6];
    *nfinite_out = nfinite;
    if (nfinite <= mean_k) {
        set_error("statistical outlier removal needs more than mean_k = %d finite points (%lld given): PCL reads past "
                  "the neighbour list", mean_k, nfinite);
        return done(ORBGPU_EINVAL);
    }
    auto ord2f_host = [](unsigned o) {
        const unsigned b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        float f;
        memcpy(&f, &b, 4);
        return f;
    };
    float mn[3], mx[3];
    double ext = 0;
    for (int a = 0; a < 3; a++) {
        mn[a] = ord2f_host(hbox[a]);
        mx[a] = ord2f_host(hbox[3 + a]);
        ext = std::max(ext, (double)mx[a] - (double)mn[a]);
    }
    SorGrid g;
    double h = std::max(ext / 256.0, 1e-6);
    if (ext > 0) {
        // pilot: occupied cells at a coarse size
        sor_grid_for(mn, mx, h, g);
        if ((rc = sor_sort_cells(ws, d_pts, n, g, st)) != ORBGPU_OK)
            return done(rc);
        const int nbc = (int)std::min<long long>((nfinite + 255) / 256, 1024);
        hipLaunchKernelGGL(k_sor_count_cells, dim3(nbc), dim3(256), 0, st, ws.keys[0].as<uint32_t>(), (int)nfinite,
                           reinterpret_cast<int *>(scratch.as<unsigned>() + 7));
        e = hipMemcpyAsync(hbox, scratch.p, sizeof(hbox), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            set_error("statistical outlier removal: %s", hipGetErrorString(e));
            return done(ORBGPU_EHIP);
        }
        const double per_cell = (double)nfinite / (double)std::max(hbox[7], 1u);
        h = g.h * std::sqrt(SOR_CELL_POINTS / per_cell);
        h = std::min(std::max(h, ext / 4096.0), ext);
    }
    sor_grid_for(mn, mx, h, g);
    if ((rc = sor_sort_cells(ws, d_pts, n, g, st)) != ORBGPU_OK)
        return done(rc);
    hipLaunchKernelGGL(k_sor_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_pts, ws.vals[0].as<uint32_t>(),
                       (int)n, sorted.as<float4>());
    hipLaunchKernelGGL(k_sor_knn, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, sorted.as<float4>(),
                       ws.keys[0].as<uint32_t>(), ws.vals[0].as<uint32_t>(), (int)n, (int)nfinite, g, mean_k, d_mean_dist);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error("statistical outlier removal: %s", hipGetErrorString(e));
        return done(ORBGPU_EHIP);
    }
    return done(ORBGPU_OK);
}

// PCL's statistics and selection (applyFilterIndices): sequential double sums over the float distances in index order
// -- host work, like the std::vector loops they restate.  keep[i] = 1 for the points that stay.
static long long sor_select(const float *dist, long long n, long long nfinite, double stddev_mul, uint8_t *keep)
{
    double sum = 0, sq_sum = 0;
    for (long long i = 0; i < n; i++) {
        sum += dist[i];
        sq_sum += dist[i] * dist[i];  // a float product, as in PCL
    }
    const double valid = (double)nfinite;
    const double mean = sum / valid;
    const double variance = (sq_sum - sum * sum / valid) / (valid - 1);
    const double stddev = sqrt(variance);
    const double distance_threshold = mean + stddev_mul * stddev;
    long long kept = 0;
    for (long long i = 0; i < n; i++) {
        keep[i] = !(dist[i] > distance_threshold);
        kept += keep[i];
    }
    return kept;
}

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_cloud {
    int device_id = 0;
    float leaf = 0.01f;
    DevBuf map[2];        // ping-pong: map[cur] holds the global map
    DevBuf box;           // 2 x 6 words: bounding box of map[0] / map[1] (order-preserving float encodings)
    int cur = 0;
    long long size = 0;   // points in the global map
    int last_overflow = 0;
    bool sorted_map = false;  // map[cur] came out of a voxel filter (sorted, one point per voxel): fast path allowed
    int last_path = 0;    // 1 = merge (fast) path, 2 = general path, 3 = merge attempted, redone by the general path
    VoxelWorkspace ws;
    FrameStage fs;
    hipStream_t stream = nullptr;
    bool profiling = false;  // HIP events on the handle's stream around the kernels of an insert
    hipEvent_t ev[2] = {nullptr, nullptr};
    float last_ms = -1.f;
    int merge_cap = 0;    // resident indices k_merge_new stages per workgroup (MN_CAP; ORBGPU_DEBUG_MERGE_CAP lowers it for tests)
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

static int read_state(orbgpu_cloud *c, CloudState &hs)
{
    ORBGPU_HIP_TRY(hipMemcpyAsync(&hs, c->ws.state.p, sizeof(CloudState), hipMemcpyDeviceToHost, c->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(c->stream));
    return ORBGPU_OK;
}

// general path: filter map[cur][0..k) into map[cur^1], swap, read the new size
static int cloud_filter(orbgpu_cloud *c, long long k)
{
    int rc = cloud_grow(c, c->cur ^ 1, std::max<long long>(k, 1), 0);
    if (rc != ORBGPU_OK)
        return rc;
    rc = voxel_filter_device(c->ws, c->map[c->cur].as<Point>(), k, c->leaf, c->map[c->cur ^ 1].as<Point>(),
                             c->box.as<unsigned>() + BOX_WORDS * (c->cur ^ 1), c->stream);
    if (rc != ORBGPU_OK)
        return rc;
    CloudState hs;
    if ((rc = read_state(c, hs)) != ORBGPU_OK)
        return rc;
    c->cur ^= 1;
    c->size = hs.nout;
    c->last_overflow = hs.overflow;
    c->sorted_map = !hs.overflow;
    return ORBGPU_OK;
}

// One key frame (PointCloudMap.cc:204-262): new points appended behind the resident map, then the voxel filter
// over map + new.  Merge path (8 launches, the resident map is read and written once): back-projection, keys of
// the new points, four digit passes, new-side merge, resident-side merge.  It assumes what a filtered map
// guarantees -- sorted, one point per voxel -- and checks it; overflow / unsorted maps take the general path.
static int cloud_insert_device(orbgpu_cloud *c, const float *d_depth, size_t dstride, const uint8_t *d_rgb,
                               size_t cstride, int w, int h, float fx, float fy, float cx, float cy, const float *Tcw)
{
    int rc;
    const long long K = c->size;
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    ORBGPU_REQUIRE(K + maxnew < (1ll << 30), "dense map too large (%lld points)", K + maxnew);
    if ((rc = cloud_grow(c, c->cur, K + maxnew, K)) != ORBGPU_OK || (rc = cloud_grow(c, c->cur ^ 1, K + maxnew, 0)) != ORBGPU_OK)
        return rc;
    if ((rc = c->ws.reserve(maxnew, bp_tiles(w, h))) != ORBGPU_OK)
        return rc;
    if ((rc = c->box.reserve(sizeof(unsigned) * 2 * BOX_WORDS)) != ORBGPU_OK ||
        (rc = c->ws.tile_lb.reserve(sizeof(int) * (size_t)(tiles_of(K) + 1) * 2)) != ORBGPU_OK)
        return rc;
    if (sizeof(uint32_t) * (size_t)K > c->ws.okeys.bytes &&  // (grows with the map: with slack, not at every key frame)
        (rc = c->ws.okeys.reserve(sizeof(uint32_t) * (size_t)(K + K / 2 + 4 * maxnew))) != ORBGPU_OK)
        return rc;
    VoxelWorkspace &ws = c->ws;
    CloudState *S = ws.state.as<CloudState>();
    Point *old = c->map[c->cur].as<Point>(), *newp = old + K, *out = c->map[c->cur ^ 1].as<Point>();
    unsigned *box_in = c->box.as<unsigned>() + BOX_WORDS * c->cur, *box_out = c->box.as<unsigned>() + BOX_WORDS * (c->cur ^ 1);
    const float inv = 1.0f / c->leaf;
    const bool merge = K == 0 || c->sorted_map;
    c->last_ms = -1.f;
    if (c->profiling)
        ORBGPU_HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    // globalMap += transform(convertToPointCloud(kf), Twc)   (:204-249)
    if ((rc = backproject_launch(c->fs, d_depth, dstride, d_rgb, cstride, w, h, fx, fy, cx, cy, Tcw, newp, &S->n_sort,
                                 ws.ghist.as<unsigned>(), box_out, c->stream)) != ORBGPU_OK)
        return rc;
    CloudState hs;
    if (merge) {
        const int ntn = tiles_of(maxnew);
        // + the workgroups that index the resident map: ~4 k points each, at most as many as fill the device with the others
        const int okg = K > 0 ? (int)std::min<long long>((K + 4095) / 4096, std::max(256 - ntn, 64)) : 0;
        hipLaunchKernelGGL(k_vox_keys, dim3(ntn + okg), dim3(256), 0, c->stream, newp, &S->n_sort, c->fs.part.as<Box>(),
                           bp_tiles(w, h), K > 0 ? box_in : (const unsigned *)nullptr, inv, S, ws.keys[0].as<uint32_t>(),
                           ws.vals[0].as<uint32_t>(), ws.ghist.as<unsigned>(), (const Point *)old, (int)K,
                           ws.table.as<uint32_t>(), (int)maxnew, ws.okeys.as<uint32_t>(), okg);
        radix_sort_device(ws, ntn, c->stream);
        const int mn_tiles = (int)((maxnew + MN_TILE - 1) / MN_TILE);
        hipLaunchKernelGGL(k_merge_new, dim3(mn_tiles + merge_side_groups(K)), dim3(256),
                           sizeof(uint32_t) * (size_t)std::max(c->merge_cap, MERGE_TABLE), c->stream, old,
                           ws.okeys.as<uint32_t>(), (int)K, newp,
                           ws.keys[0].as<uint32_t>(), ws.vals[0].as<uint32_t>(), S, ws.table.as<uint32_t>(), c->merge_cap,
                           ws.cexcl.as<unsigned>(), ws.tile_lb.as<int>(), out, ws.ctl.as<LbCtl>(),
                           ws.status.as<unsigned long long>(), ws.next_epoch(), box_out);
        if (K > 0)
            hipLaunchKernelGGL(k_merge_old, dim3(tiles_of(K)), dim3(256), 0, c->stream, old, (int)K,
                               ws.keys[0].as<uint32_t>(), ws.cexcl.as<unsigned>(), ws.tile_lb.as<int>(), inv, S, out,
                               box_out);
        ORBGPU_HIP_TRY(hipGetLastError());
        if (c->profiling)
            ORBGPU_HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        if ((rc = read_state(c, hs)) != ORBGPU_OK)
            return rc;
        if (!hs.overflow && !hs.unsorted) {
            if (c->profiling)
                ORBGPU_HIP_TRY(hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
            c->cur ^= 1;
            c->size = hs.nout;
            c->last_overflow = 0;
            c->sorted_map = true;
            c->last_path = 1;
            return ORBGPU_OK;
        }
        c->last_path = 3;
    } else {
        if ((rc = read_state(c, hs)) != ORBGPU_OK)
            return rc;
        c->last_path = 2;
    }
    // voxel.setInputCloud(globalMap); voxel.filter(tmp); swap   (:259-262), general path
    rc = cloud_filter(c, K + hs.n_sort);
    if (rc == ORBGPU_OK && c->profiling) {
        ORBGPU_HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        ORBGPU_HIP_TRY(hipEventSynchronize(c->ev[1]));
        ORBGPU_HIP_TRY(hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
    }
    return rc;
}

} // namespace orbgpu

extern "C" {

int orbgpu_cloud_create(double resolution, int32_t device_id, orbgpu_cloud **out)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
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
    // k_merge_new stages up to MN_CAP resident indices (96 KB of dynamic LDS)
    c->merge_cap = orbgpu::MN_CAP;
    if (const char *dbg = getenv("ORBGPU_DEBUG_MERGE_CAP"))  // tests: a small cap sends tiles down the sampled fall-back
        c->merge_cap = std::min(std::max(atoi(dbg), 0), orbgpu::MN_CAP);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(orbgpu::k_merge_new), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(uint32_t) * orbgpu::MN_CAP));
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(k_merge_new): %s", hipGetErrorString(e));
        (void)hipStreamDestroy(c->stream);
        delete c;
        return ORBGPU_EHIP;
    }
    *out = c;
    return ORBGPU_OK;
}

int orbgpu_cloud_destroy(orbgpu_cloud *c)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!c)
        return ORBGPU_OK;
    (void)hipSetDevice(c->device_id);
    (void)hipDeviceSynchronize();
    c->map[0].release();
    c->map[1].release();
    c->box.release();
    c->ws.release();
    c->fs.release();
    for (hipEvent_t e : c->ev)
        if (e)
            (void)hipEventDestroy(e);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
    return ORBGPU_OK;
}

static int check_frame_args(const void *depth, size_t dstride, const void *rgb, size_t cstride, int w, int h)
{
    ORBGPU_REQUIRE(depth && rgb, "null image");
    ORBGPU_REQUIRE(w > 0 && h > 0 && dstride >= (size_t)w && cstride >= (size_t)w * 3, "bad image size / strides");
    ORBGPU_REQUIRE((long long)w * h < (1ll << 30), "image too large");
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
    if ((rc = upload_frame(c->fs, depth, dstride, rgb, cstride, w, h, c->stream)) != ORBGPU_OK)
        return rc;
    return cloud_insert_device(c, c->fs.depth.as<float>(), (size_t)w, c->fs.rgb.as<uint8_t>(), (size_t)w * 3, w, h, fx, fy,
                               cx, cy, Tcw);
}

int orbgpu_cloud_insert_device(orbgpu_cloud *c, const float *d_depth, size_t dstride, const uint8_t *d_rgb,
                               size_t cstride, int32_t w, int32_t h, float fx, float fy, float cx, float cy,
                               const float *Tcw)
{
    ORBGPU_REQUIRE(c && Tcw, "null argument");
    int rc = check_frame_args(d_depth, dstride, d_rgb, cstride, w, h);
    if (rc != ORBGPU_OK)
        return rc;
    if ((rc = select_device(c->device_id)) != ORBGPU_OK)
        return rc;
    return cloud_insert_device(c, d_depth, dstride, d_rgb, cstride, w, h, fx, fy, cx, cy, Tcw);
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
    c->sorted_map = false;
    if (n == 0)
        return ORBGPU_OK;
    ORBGPU_REQUIRE(w > 0 && h > 0, "bad image size");
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    ORBGPU_REQUIRE(maxnew * n < (1ll << 30), "too many points for one rebuild");
    if ((rc = cloud_grow(c, c->cur, maxnew * n, 0)) != ORBGPU_OK)
        return rc;
    if ((rc = c->ws.reserve(maxnew, bp_tiles(w, h))) != ORBGPU_OK || (rc = c->box.reserve(sizeof(unsigned) * 2 * BOX_WORDS)) != ORBGPU_OK)
        return rc;
    CloudState *S = c->ws.state.as<CloudState>();
    long long total = 0;
    for (int k = 0; k < n; k++) {
        ORBGPU_REQUIRE(Tcw[k], "null pose");
        if ((rc = check_frame_args(depth[k], dstride, rgb[k], cstride, w, h)) != ORBGPU_OK)
            return rc;
        if ((rc = upload_frame(c->fs, depth[k], dstride, rgb[k], cstride, w, h, c->stream)) != ORBGPU_OK)
            return rc;
        if ((rc = backproject_launch(c->fs, c->fs.depth.as<float>(), (size_t)w, c->fs.rgb.as<uint8_t>(), (size_t)w * 3, w,
                                     h, fx, fy, cx, cy, Tcw[k], c->map[c->cur].as<Point>() + total, &S->n_sort, nullptr,
                                     nullptr, c->stream)) != ORBGPU_OK)
            return rc;
        CloudState hs;
        if ((rc = read_state(c, hs)) != ORBGPU_OK)
            return rc;
        total += hs.n_sort;
    }
    c->last_path = 2;
    return cloud_filter(c, total);
}

int orbgpu_cloud_clear(orbgpu_cloud *c)
{
    ORBGPU_REQUIRE(c, "null argument");
    c->size = 0;  // globalMap->clear()  (:270)
    c->sorted_map = false;
    c->last_overflow = 0;
    return ORBGPU_OK;
}

// One iteration of the shutdown loop (PointCloudMap.cc:272-282): generatePointCloud(kf), voxel.filter of THAT cloud
// alone, globalMap += the result.  The map is a concatenation afterwards (not sorted, not one point per voxel).
static int cloud_append_filtered_device(orbgpu_cloud *c, const float *d_depth, size_t dstride, const uint8_t *d_rgb,
                                        size_t cstride, int w, int h, float fx, float fy, float cx, float cy,
                                        const float *Tcw)
{
    int rc;
    const long long K = c->size;
    const long long maxnew = (long long)((h + 2) / 3) * ((w + 2) / 3);
    ORBGPU_REQUIRE(K + maxnew < (1ll << 30), "dense map too large (%lld points)", K + maxnew);
    if ((rc = cloud_grow(c, c->cur, K + maxnew, K)) != ORBGPU_OK || (rc = cloud_grow(c, c->cur ^ 1, maxnew, 0)) != ORBGPU_OK)
        return rc;
    if ((rc = c->ws.reserve(maxnew, bp_tiles(w, h))) != ORBGPU_OK || (rc = c->box.reserve(sizeof(unsigned) * 2 * BOX_WORDS)) != ORBGPU_OK)
        return rc;
    CloudState *S = c->ws.state.as<CloudState>();
    Point *scratch = c->map[c->cur ^ 1].as<Point>();
    if ((rc = backproject_launch(c->fs, d_depth, dstride, d_rgb, cstride, w, h, fx, fy, cx, cy, Tcw, scratch, &S->n_sort,
                                 nullptr, nullptr, c->stream)) != ORBGPU_OK)
        return rc;
    CloudState hs;
    if ((rc = read_state(c, hs)) != ORBGPU_OK)
        return rc;
    rc = voxel_filter_device(c->ws, scratch, hs.n_sort, c->leaf, c->map[c->cur].as<Point>() + K,
                             c->box.as<unsigned>() + BOX_WORDS * (c->cur ^ 1), c->stream);
    if (rc != ORBGPU_OK || (rc = read_state(c, hs)) != ORBGPU_OK)
        return rc;
    c->size = K + hs.nout;
    c->last_overflow = hs.overflow;
    c->sorted_map = false;
    c->last_path = 2;
    return ORBGPU_OK;
}

int orbgpu_cloud_append_filtered(orbgpu_cloud *c, const float *depth, size_t dstride, const uint8_t *rgb, size_t cstride,
                                 int32_t w, int32_t h, float fx, float fy, float cx, float cy, const float *Tcw)
{
    ORBGPU_REQUIRE(c && Tcw, "null argument");
    int rc = check_frame_args(depth, dstride, rgb, cstride, w, h);
    if (rc != ORBGPU_OK)
        return rc;
    if ((rc = select_device(c->device_id)) != ORBGPU_OK)
        return rc;
    if ((rc = upload_frame(c->fs, depth, dstride, rgb, cstride, w, h, c->stream)) != ORBGPU_OK)
        return rc;
    return cloud_append_filtered_device(c, c->fs.depth.as<float>(), (size_t)w, c->fs.rgb.as<uint8_t>(), (size_t)w * 3, w, h,
                                        fx, fy, cx, cy, Tcw);
}

// sor.setInputCloud(globalMap); sor.filter(*tmp); globalMap->swap(*tmp)  (PointCloudMap.cc:283-285)
int orbgpu_cloud_remove_outliers(orbgpu_cloud *c, int32_t mean_k, double stddev_mul, int64_t *removed)
{
    ORBGPU_REQUIRE(c, "null argument");
    int rc = select_device(c->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (removed)
        *removed = 0;
    const long long n = c->size;
    ORBGPU_REQUIRE(n > mean_k, "statistical outlier removal needs more than mean_k = %d points (the map holds %lld)", mean_k, n);
    DevBuf ddist;
    if ((rc = ddist.reserve(sizeof(float) * (size_t)n)) != ORBGPU_OK)
        return rc;
    long long nfinite = 0;
    rc = sor_mean_distances(c->ws, c->map[c->cur].as<Point>(), n, mean_k, ddist.as<float>(), &nfinite, c->stream);
    std::vector<float> dist;
    std::vector<Point> pts;
    std::vector<uint8_t> keep;
    if (rc == ORBGPU_OK) {
        dist.resize((size_t)n);
        pts.resize((size_t)n);
        keep.resize((size_t)n);
        hipError_t e = hipMemcpy(dist.data(), ddist.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(pts.data(), c->map[c->cur].p, sizeof(Point) * (size_t)n, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            const long long kept = sor_select(dist.data(), n, nfinite, stddev_mul, keep.data());
            long long o = 0;
            for (long long i = 0; i < n; i++)
                if (keep[(size_t)i])
                    pts[(size_t)o++] = pts[(size_t)i];
            if (kept > 0)
                e = hipMemcpy(c->map[c->cur].p, pts.data(), sizeof(Point) * (size_t)kept, hipMemcpyHostToDevice);
            if (e == hipSuccess)
                e = hipStreamSynchronize(nullptr);  // null-stream work is not ordered with the handle's non-blocking stream
            if (e == hipSuccess) {
                c->size = kept;
                c->sorted_map = false;  // the next insert re-derives order and bounding box from the points
                if (removed)
                    *removed = n - kept;
            }
        }
        if (e != hipSuccess) {
            set_error("cloud_remove_outliers: %s", hipGetErrorString(e));
            rc = ORBGPU_EHIP;
        }
    }
    ddist.release();
    return rc;
}

// pcl::StatisticalOutlierRemoval<PointXYZRGBA>::filter on host points (stateless, for parity tests and other callers)
int orbgpu_statistical_outlier_removal(const orbgpu_point_xyzrgba *in, int64_t n, int32_t mean_k, double stddev_mul,
                                       orbgpu_point_xyzrgba *out, int64_t cap, int64_t *n_out, float *mean_dist,
                                       int32_t device_id)
{
    ORBGPU_REQUIRE(in && out && n_out && n >= 1 && n < (1ll << 30), "bad arguments");
    ORBGPU_REQUIRE(cap >= n, "cap must be >= n");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    *n_out = 0;
    VoxelWorkspace ws;
    DevBuf din, ddist;
    auto cleanup = [&]() {
        ws.release();
        din.release();
        ddist.release();
    };
    if ((rc = din.reserve(sizeof(Point) * (size_t)n)) != ORBGPU_OK || (rc = ddist.reserve(sizeof(float) * (size_t)n)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpy(din.p, in, sizeof(Point) * (size_t)n, hipMemcpyHostToDevice);
    long long nfinite = 0;
    if (e == hipSuccess) {
        rc = sor_mean_distances(ws, din.as<Point>(), n, mean_k, ddist.as<float>(), &nfinite, nullptr);
        if (rc == ORBGPU_OK) {
            std::vector<float> dist((size_t)n);
            std::vector<uint8_t> keep((size_t)n);
            e = hipMemcpy(dist.data(), ddist.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                sor_select(dist.data(), n, nfinite, stddev_mul, keep.data());
                int64_t o = 0;
                for (int64_t i = 0; i < n; i++) {
                    if (mean_dist)
                        mean_dist[i] = dist[(size_t)i];
                    if (keep[(size_t)i])
                        out[o++] = in[i];
                }
                *n_out = o;
            }
        }
    }
    if (e != hipSuccess) {
        set_error("statistical_outlier_removal: %s", hipGetErrorString(e));
        rc = ORBGPU_EHIP;
    }
    cleanup();
    return rc;
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

int orbgpu_cloud_set_profiling(orbgpu_cloud *c, int32_t enable)
{
    ORBGPU_REQUIRE(c, "null argument");
    int rc = select_device(c->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (enable && !c->ev[0]) {
        ORBGPU_HIP_TRY(hipEventCreate(&c->ev[0]));
        ORBGPU_HIP_TRY(hipEventCreate(&c->ev[1]));
    }
    c->profiling = enable != 0;
    return ORBGPU_OK;
}

int orbgpu_cloud_last_insert_ms(orbgpu_cloud *c, float *ms)
{
    ORBGPU_REQUIRE(c && ms, "null argument");
    ORBGPU_REQUIRE(c->last_ms >= 0.f, "no profiled insert recorded (orbgpu_cloud_set_profiling)");
    *ms = c->last_ms;
    return ORBGPU_OK;
}

int orbgpu_cloud_last_path(orbgpu_cloud *c, int32_t *path)
{
    ORBGPU_REQUIRE(c && path, "null argument");
    *path = c->last_path;
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
    DevBuf dout, dcount;
    auto cleanup = [&]() {
        fs.release();
        dout.release();
        dcount.release();
    };
    if ((rc = dout.reserve(sizeof(Point) * (size_t)maxnew)) != ORBGPU_OK || (rc = dcount.reserve(sizeof(int))) != ORBGPU_OK ||
        (rc = upload_frame(fs, depth, dstride, rgb, cstride, w, h, nullptr)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    rc = backproject_launch(fs, fs.depth.as<float>(), (size_t)w, fs.rgb.as<uint8_t>(), (size_t)w * 3, w, h, fx, fy, cx, cy,
                            Tcw, dout.as<Point>(), dcount.as<int>(), nullptr, nullptr, nullptr);
    int total = 0;
    hipError_t e = hipSuccess;
    if (rc == ORBGPU_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess)
            e = hipMemcpy(&total, dcount.p, sizeof(int), hipMemcpyDeviceToHost);
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
    ORBGPU_REQUIRE(n >= 0 && n < (1ll << 30) && (n == 0 || in) && out && n_out, "bad arguments");
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
    DevBuf din, dout, dbox;
    auto cleanup = [&]() {
        ws.release();
        din.release();
        dout.release();
        dbox.release();
    };
    if ((rc = din.reserve(sizeof(Point) * (size_t)n)) != ORBGPU_OK || (rc = dout.reserve(sizeof(Point) * (size_t)n)) != ORBGPU_OK ||
        (rc = dbox.reserve(sizeof(unsigned) * BOX_WORDS)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpy(din.p, in, sizeof(Point) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = voxel_filter_device(ws, din.as<Point>(), n, (float)resolution, dout.as<Point>(), dbox.as<unsigned>(), nullptr);
        if (rc == ORBGPU_OK) {
            CloudState hs;
            e = hipDeviceSynchronize();
            if (e == hipSuccess)
                e = hipMemcpy(&hs, ws.state.p, sizeof(CloudState), hipMemcpyDeviceToHost);
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
