// 256-bit Hamming brute-force matcher for MI355X (gfx950).
//
// Replaces ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1647-1663) and the
// all-pairs loop of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (:159-288) for the case where
// every feature shares one vocabulary node (the "BF-Hamming" configuration).
//
// The reference is a greedy sequential algorithm: A rows are visited in index order and a B row
// claimed by an earlier A row is invisible to later ones (:209-210, :232).  The GPU computes the
// same result as the unique fixpoint of a triangular system: sweep s recomputes EVERY row in
// parallel, hiding from row i the B rows claimed by rows < i in sweep s-1.  Row 0 is final after
// sweep 0, row i after at most i+1 sweeps; in practice 2-4 sweeps suffice.  A sweep that changes
// nothing proves convergence; a serial single-wave pass is the bounded fallback.
#include "common.h"
#include "matcher_common.h"

#include <algorithm>
#include <climits>
#include <new>

namespace orbgpu {

constexpr int BF_ROWS_PER_WAVE = 8;
constexpr int BF_ROWS_PER_BLOCK = 4 * BF_ROWS_PER_WAVE;
constexpr int BF_MAX_SWEEPS = 12;

struct Best2 {
    int b1, i1, b2;
};

__device__ __forceinline__ Best2 merge_best2(Best2 a, int ob1, int oi1, int ob2)
{
    Best2 r;
    const bool other_first = (ob1 < a.b1) || (ob1 == a.b1 && oi1 < a.i1);
    r.b1 = other_first ? ob1 : a.b1;
    r.i1 = other_first ? oi1 : a.i1;
    r.b2 = min(min(a.b2, ob2), max(a.b1, ob1));
    return r;
}

__device__ __forceinline__ Best2 wave_best2(Best2 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int ob1 = __shfl_xor(v.b1, off, 64);
        const int oi1 = __shfl_xor(v.i1, off, 64);
        const int ob2 = __shfl_xor(v.b2, off, 64);
        v = merge_best2(v, ob1, oi1, ob2);
    }
    return v;
}

// One sweep.  grid = (ceil(cap / BF_ROWS_PER_BLOCK), pairs).  Dynamic LDS: B descriptors as four
// u64 planes (conflict-free ds_read_b64 per lane) + the claim table.
__global__ __launch_bounds__(256) void k_bf_sweep(int sweep, int cap, const uint8_t *__restrict__ desc_a,
                                                  const uint8_t *__restrict__ valid_a,
                                                  const int *__restrict__ na_p, const uint8_t *__restrict__ desc_b,
                                                  const int *__restrict__ nb_p, int th_low, float nnratio,
                                                  int *__restrict__ match_a, int *__restrict__ claim3,
                                                  int *__restrict__ changed)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int pair = blockIdx.y;
    const int na = min(na_p[pair], cap), nb = min(nb_p[pair], cap);
    int *chg = changed + (size_t)pair * (BF_MAX_SWEEPS + 1);
    if (sweep > 0 && chg[sweep - 1] == 0)
        return;  // converged in an earlier sweep (match_a is final)
    const int row0 = blockIdx.x * BF_ROWS_PER_BLOCK;
    int *c_rd = claim3 + ((size_t)pair * 3 + (sweep % 3)) * cap;
    int *c_wr = claim3 + ((size_t)pair * 3 + ((sweep + 1) % 3)) * cap;
    int *c_cl = claim3 + ((size_t)pair * 3 + ((sweep + 2) % 3)) * cap;
    // clear the table sweep+1 will write (grid-strided over this pair's blocks)
    for (int j = row0 + threadIdx.x; j < min(row0 + BF_ROWS_PER_BLOCK, cap); j += 256)
        c_cl[j] = INT_MAX;
    if (row0 >= na || na <= 0)
        return;

    uint64_t *bplane = reinterpret_cast<uint64_t *>(smem);          // [4][nbp]
    const int nbp = (nb + 63) & ~63;
    int *claim = reinterpret_cast<int *>(smem + (size_t)32 * nbp);  // [nbp]
    const uint64_t *gb = reinterpret_cast<const uint64_t *>(desc_b + (size_t)pair * cap * 32);
    for (int i = threadIdx.x; i < nb * 4; i += 256) {
        const int j = i >> 2, w = i & 3;
        bplane[w * nbp + j] = gb[i];
    }
    for (int j = threadIdx.x; j < nb; j += 256)
        claim[j] = c_rd[j];
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t *ga = reinterpret_cast<const uint64_t *>(desc_a + (size_t)pair * cap * 32);
    int *m = match_a + (size_t)pair * cap;
    const uint8_t *va = valid_a ? valid_a + (size_t)pair * cap : nullptr;
    int n_changed = 0;
    for (int r = 0; r < BF_ROWS_PER_WAVE; r++) {
        const int i = row0 + wave * BF_ROWS_PER_WAVE + r;
        if (i >= na)
            break;
        int result = -1;
        if (!va || va[i]) {
            uint64_t a[4];
#pragma unroll
            for (int w = 0; w < 4; w++)
                a[w] = ga[(size_t)i * 4 + w];
            Best2 v{256, INT_MAX, 256};
            for (int j = lane; j < nb; j += 64) {
                if (claim[j] < i)
                    continue;  // claimed by an earlier A row (:209-210)
                const int d = __popcll(a[0] ^ bplane[j]) + __popcll(a[1] ^ bplane[nbp + j]) +
                              __popcll(a[2] ^ bplane[2 * nbp + j]) + __popcll(a[3] ^ bplane[3 * nbp + j]);
                if (d < v.b1) {
                    v.b2 = v.b1;
                    v.b1 = d;
                    v.i1 = j;
                } else if (d < v.b2) {
                    v.b2 = d;
                }
            }
            v = wave_best2(v);
            // :228-231
            if (v.b1 <= th_low && (float)v.b1 < nnratio * (float)v.b2)
                result = v.i1;
        }
        if (lane == 0) {
            if (sweep == 0 || m[i] != result)
                n_changed++;
            m[i] = result;
            if (result >= 0)
                atomicMin(&c_wr[result], i);
        }
    }
    if (lane == 0 && n_changed)
        atomicAdd(&chg[sweep], n_changed);
}

// Bounded fallback: exact serial greedy pass by one wave per pair, only if the sweeps did not
// converge.
__global__ __launch_bounds__(64) void k_bf_serial(int cap, const uint8_t *__restrict__ desc_a,
                                                  const uint8_t *__restrict__ valid_a,
                                                  const int *__restrict__ na_p, const uint8_t *__restrict__ desc_b,
                                                  const int *__restrict__ nb_p, int th_low, float nnratio,
                                                  int *__restrict__ match_a, int *__restrict__ claim3,
                                                  int *__restrict__ changed)
{
    const int pair = blockIdx.x;
    int *chg = changed + (size_t)pair * (BF_MAX_SWEEPS + 1);
    if (chg[BF_MAX_SWEEPS - 1] == 0)
        return;
    const int na = min(na_p[pair], cap), nb = min(nb_p[pair], cap);
    const int lane = threadIdx.x;
    int *claimed = claim3 + (size_t)pair * 3 * cap;  // reuse table 0 as 0/1 flags
    for (int j = lane; j < nb; j += 64)
        claimed[j] = 0;
    __syncthreads();
    const uint64_t *ga = reinterpret_cast<const uint64_t *>(desc_a + (size_t)pair * cap * 32);
    const uint64_t *gb = reinterpret_cast<const uint64_t *>(desc_b + (size_t)pair * cap * 32);
    int *m = match_a + (size_t)pair * cap;
    const uint8_t *va = valid_a ? valid_a + (size_t)pair * cap : nullptr;
    for (int i = 0; i < na; i++) {
        int result = -1;
        if (!va || va[i]) {
            uint64_t a[4];
            for (int w = 0; w < 4; w++)
                a[w] = ga[(size_t)i * 4 + w];
            Best2 v{256, INT_MAX, 256};
            for (int j = lane; j < nb; j += 64) {
                if (claimed[j])
                    continue;
                uint64_t b[4];
                for (int w = 0; w < 4; w++)
                    b[w] = gb[(size_t)j * 4 + w];
                const int d = hamming256(a, b);
                if (d < v.b1) {
                    v.b2 = v.b1;
                    v.b1 = d;
                    v.i1 = j;
                } else if (d < v.b2) {
                    v.b2 = d;
                }
            }
            v = wave_best2(v);
            if (v.b1 <= th_low && (float)v.b1 < nnratio * (float)v.b2)
                result = v.i1;
        }
        if (lane == 0) {
            m[i] = result;
            if (result >= 0)
                claimed[result] = 1;
        }
        __syncthreads();  // single wave: orders the flag store before the next row's loads
    }
    if (lane == 0)
        chg[BF_MAX_SWEEPS] = 1;  // diagnostic: fallback used
}

// Scatter A->B matches, rotation-histogram consistency (:262-285), count.  One workgroup per pair.
__global__ __launch_bounds__(256) void k_bf_finish(int cap, const int *__restrict__ na_p,
                                                   const int *__restrict__ nb_p, const int *__restrict__ match_a,
                                                   const uint8_t *__restrict__ angle_a,
                                                   const uint8_t *__restrict__ angle_b, size_t angle_stride,
                                                   int check_orientation, int *__restrict__ match_b,
                                                   int *__restrict__ nmatches, const int *__restrict__ changed,
                                                   int *__restrict__ sweeps_used)
{
    __shared__ int histo[ORBGPU_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    const int pair = blockIdx.x;
    const int na = min(na_p[pair], cap), nb = min(nb_p[pair], cap);
    const int *m = match_a + (size_t)pair * cap;
    int *mb = match_b + (size_t)pair * cap;
    const uint8_t *aa = angle_a + (size_t)pair * cap * angle_stride;
    const uint8_t *ab = angle_b + (size_t)pair * cap * angle_stride;
    for (int j = threadIdx.x; j < cap; j += 256)
        mb[j] = -1;
    if (threadIdx.x < ORBGPU_HISTO_LENGTH)
        histo[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        s_count = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = threadIdx.x; i < na; i += 256) {
        const int j = m[i];
        if (j < 0 || j >= nb)
            continue;
        cnt++;
        if (check_orientation) {
            const float fa = *reinterpret_cast<const float *>(aa + (size_t)i * angle_stride);
            const float fb = *reinterpret_cast<const float *>(ab + (size_t)j * angle_stride);
            atomicAdd(&histo[rot_bin(fa, fb)], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int i1 = -1, i2 = -1, i3 = -1;
        if (check_orientation)
            three_maxima(histo, ORBGPU_HISTO_LENGTH, i1, i2, i3);
        s_keep[0] = i1;
        s_keep[1] = i2;
        s_keep[2] = i3;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < na; i += 256) {
        const int j = m[i];
        if (j < 0 || j >= nb)
            continue;
        bool keep = true;
        if (check_orientation) {
            const float fa = *reinterpret_cast<const float *>(aa + (size_t)i * angle_stride);
            const float fb = *reinterpret_cast<const float *>(ab + (size_t)j * angle_stride);
            const int b = rot_bin(fa, fb);
            keep = (b == s_keep[0] || b == s_keep[1] || b == s_keep[2]);
        }
        if (keep)
            mb[j] = i;
        else
            cnt--;
    }
    cnt = wave_reduce_add(cnt);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&s_count, cnt);
    __syncthreads();
    if (threadIdx.x == 0) {
        nmatches[pair] = s_count;
        const int *chg = changed + (size_t)pair * (BF_MAX_SWEEPS + 1);
        int s = 0;
        while (s < BF_MAX_SWEEPS && chg[s] != 0)
            s++;
        sweeps_used[pair] = chg[BF_MAX_SWEEPS] ? -1 : s + 1;
    }
}

// M0 for n independent pairs
__global__ void k_hamming_pairs(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int n,
                                int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t *pa = reinterpret_cast<const uint64_t *>(a) + (size_t)i * 4;
    const uint64_t *pb = reinterpret_cast<const uint64_t *>(b) + (size_t)i * 4;
    uint64_t x[4] = {pa[0], pa[1], pa[2], pa[3]}, y[4] = {pb[0], pb[1], pb[2], pb[3]};
    out[i] = hamming256(x, y);
}

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_matcher {
    int device_id = 0;
    int max_pairs = 0, cap = 0;
    DevBuf d_match_a, d_claim, d_changed, d_sweeps;
    int last_pairs = 0;
};

extern "C" {

int orbgpu_matcher_create(int32_t device_id, int32_t max_pairs, int32_t cap, orbgpu_matcher **out)
{
    ORBGPU_REQUIRE(out, "null argument");
    ORBGPU_REQUIRE(max_pairs >= 1 && cap >= 1 && cap <= 4096, "max_pairs must be >= 1 and cap in [1,4096]");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_matcher *m = new (std::nothrow) orbgpu_matcher();
    if (!m) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    m->device_id = device_id;
    m->max_pairs = max_pairs;
    m->cap = cap;
    const size_t P = (size_t)max_pairs;
    if ((rc = m->d_match_a.reserve(sizeof(int) * P * cap)) != ORBGPU_OK ||
        (rc = m->d_claim.reserve(sizeof(int) * P * 3 * cap)) != ORBGPU_OK ||
        (rc = m->d_changed.reserve(sizeof(int) * P * (BF_MAX_SWEEPS + 1))) != ORBGPU_OK ||
        (rc = m->d_sweeps.reserve(sizeof(int) * P)) != ORBGPU_OK) {
        orbgpu_matcher_destroy(m);
        return rc;
    }
    hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_sweep),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 4096 + 256);
    if (he != hipSuccess) {
        set_error("hipFuncSetAttribute: %s", hipGetErrorString(he));
        orbgpu_matcher_destroy(m);
        return ORBGPU_EHIP;
    }
    *out = m;
    return ORBGPU_OK;
}

int orbgpu_matcher_destroy(orbgpu_matcher *m)
{
    if (!m)
        return ORBGPU_OK;
    (void)hipSetDevice(m->device_id);
    (void)hipDeviceSynchronize();
    m->d_match_a.release();
    m->d_claim.release();
    m->d_changed.release();
    m->d_sweeps.release();
    delete m;
    return ORBGPU_OK;
}

int orbgpu_match_bf_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                                 const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_na,
                                 const uint8_t *d_desc_b, const void *d_angle_b, const int32_t *d_nb,
                                 size_t angle_stride, int32_t th_low, float nnratio, int32_t check_orientation,
                                 int32_t *d_match_b, int32_t *d_nmatches, void *hip_stream)
{
    ORBGPU_REQUIRE(m && d_desc_a && d_desc_b && d_na && d_nb && d_match_b && d_nmatches, "null argument");
    ORBGPU_REQUIRE(pairs >= 1 && pairs <= m->max_pairs && cap >= 1 && cap <= m->cap,
                   "pairs/cap exceed the matcher's capacity (%d pairs, cap %d)", m->max_pairs, m->cap);
    ORBGPU_REQUIRE(!check_orientation || (d_angle_a && d_angle_b && angle_stride >= 4 && angle_stride % 4 == 0),
                   "orientation check needs 4-byte aligned angle arrays");
    int rc = select_device(m->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    hipStream_t st = (hipStream_t)hip_stream;
    int *match_a = m->d_match_a.as<int>();
    int *claim = m->d_claim.as<int>();
    int *changed = m->d_changed.as<int>();
    // tables 0 and 1 start empty (sweep 0 reads table 0 and writes table 1)
    ORBGPU_HIP_TRY(hipMemsetAsync(claim, 0x7F, sizeof(int) * (size_t)pairs * 3 * cap, st));
    ORBGPU_HIP_TRY(hipMemsetAsync(changed, 0, sizeof(int) * (size_t)pairs * (BF_MAX_SWEEPS + 1), st));
    const int nbp = (cap + 63) & ~63;
    const size_t lds = (size_t)36 * nbp;
    const dim3 grid((cap + BF_ROWS_PER_BLOCK - 1) / BF_ROWS_PER_BLOCK, pairs);
    for (int s = 0; s < BF_MAX_SWEEPS; s++)
        hipLaunchKernelGGL(k_bf_sweep, grid, dim3(256), lds, st, s, cap, d_desc_a, d_valid_a, d_na, d_desc_b, d_nb,
                           th_low, nnratio, match_a, claim, changed);
    hipLaunchKernelGGL(k_bf_serial, dim3(pairs), dim3(64), 0, st, cap, d_desc_a, d_valid_a, d_na, d_desc_b, d_nb,
                       th_low, nnratio, match_a, claim, changed);
    hipLaunchKernelGGL(k_bf_finish, dim3(pairs), dim3(256), 0, st, cap, d_na, d_nb, match_a,
                       reinterpret_cast<const uint8_t *>(d_angle_a), reinterpret_cast<const uint8_t *>(d_angle_b),
                       angle_stride, check_orientation, d_match_b, d_nmatches, changed, m->d_sweeps.as<int>());
    ORBGPU_HIP_TRY(hipGetLastError());
    m->last_pairs = pairs;
    return ORBGPU_OK;
}

int orbgpu_matcher_last_sweeps(orbgpu_matcher *m, int32_t *sweeps)
{
    ORBGPU_REQUIRE(m && sweeps, "null argument");
    ORBGPU_REQUIRE(m->last_pairs > 0, "no call recorded");
    int rc = select_device(m->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipDeviceSynchronize());
    ORBGPU_HIP_TRY(hipMemcpy(sweeps, m->d_sweeps.p, sizeof(int) * m->last_pairs, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}

int orbgpu_match_bf(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, int32_t na,
                    const uint8_t *desc_b, const float *angle_b, int32_t nb, int32_t th_low, float nnratio,
                    int32_t check_orientation, int32_t *match_b, int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE(match_b && nmatches, "null argument");
    ORBGPU_REQUIRE(na >= 0 && nb >= 0 && na <= 4096 && nb <= 4096, "na/nb must be in [0,4096]");
    ORBGPU_REQUIRE((na == 0 || desc_a) && (nb == 0 || desc_b), "null descriptors");
    ORBGPU_REQUIRE(!check_orientation || ((na == 0 || angle_a) && (nb == 0 || angle_b)), "null angles");
    if (na == 0 || nb == 0) {
        int rc = select_device(device_id);  // still no CPU path: fail loudly without a device
        if (rc != ORBGPU_OK)
            return rc;
        for (int j = 0; j < nb; j++)
            match_b[j] = -1;
        *nmatches = 0;
        return ORBGPU_OK;
    }
    const int cap = std::max(na, nb);
    orbgpu_matcher *m = nullptr;
    int rc = orbgpu_matcher_create(device_id, 1, cap, &m);
    if (rc != ORBGPU_OK)
        return rc;
    DevBuf da, db, aa, ab, va, cnt, mb, nm;
    auto cleanup = [&]() {
        for (DevBuf *b : {&da, &db, &aa, &ab, &va, &cnt, &mb, &nm})
            b->release();
        orbgpu_matcher_destroy(m);
    };
#define TRY_RC(x) if ((rc = (x)) != ORBGPU_OK) { cleanup(); return rc; }
#define TRY_HIP(x) { hipError_t e__ = (x); if (e__ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e__)); cleanup(); return ORBGPU_EHIP; } }
    TRY_RC(da.reserve((size_t)cap * 32));
    TRY_RC(db.reserve((size_t)cap * 32));
    TRY_RC(aa.reserve((size_t)cap * 4));
    TRY_RC(ab.reserve((size_t)cap * 4));
    TRY_RC(va.reserve((size_t)cap));
    TRY_RC(cnt.reserve(8));
    TRY_RC(mb.reserve((size_t)cap * 4));
    TRY_RC(nm.reserve(4));
    TRY_HIP(hipMemcpy(da.p, desc_a, (size_t)na * 32, hipMemcpyHostToDevice));
    TRY_HIP(hipMemcpy(db.p, desc_b, (size_t)nb * 32, hipMemcpyHostToDevice));
    if (check_orientation) {
        TRY_HIP(hipMemcpy(aa.p, angle_a, (size_t)na * 4, hipMemcpyHostToDevice));
        TRY_HIP(hipMemcpy(ab.p, angle_b, (size_t)nb * 4, hipMemcpyHostToDevice));
    }
    if (valid_a)
        TRY_HIP(hipMemcpy(va.p, valid_a, (size_t)na, hipMemcpyHostToDevice));
    int counts[2] = {na, nb};
    TRY_HIP(hipMemcpy(cnt.p, counts, 8, hipMemcpyHostToDevice));
    TRY_RC(orbgpu_match_bf_batch_device(m, 1, cap, da.as<uint8_t>(), aa.p, valid_a ? va.as<uint8_t>() : nullptr,
                                        cnt.as<int>(), db.as<uint8_t>(), ab.p, cnt.as<int>() + 1, 4, th_low, nnratio,
                                        check_orientation, mb.as<int>(), nm.as<int>(), nullptr));
    TRY_HIP(hipDeviceSynchronize());
    TRY_HIP(hipMemcpy(match_b, mb.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
    TRY_HIP(hipMemcpy(nmatches, nm.p, 4, hipMemcpyDeviceToHost));
#undef TRY_RC
#undef TRY_HIP
    cleanup();
    return ORBGPU_OK;
}

int orbgpu_hamming256(const uint8_t *a, const uint8_t *b, int32_t n, int32_t *out, int32_t device_id)
{
    ORBGPU_REQUIRE(n >= 0 && (n == 0 || (a && b && out)), "bad arguments");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (n == 0)
        return ORBGPU_OK;
    DevBuf da, db, dout;
    auto cleanup = [&]() {
        da.release();
        db.release();
        dout.release();
    };
    if ((rc = da.reserve((size_t)n * 32)) != ORBGPU_OK || (rc = db.reserve((size_t)n * 32)) != ORBGPU_OK ||
        (rc = dout.reserve((size_t)n * 4)) != ORBGPU_OK) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemcpy(da.p, a, (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(db.p, b, (size_t)n * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, 0, da.as<uint8_t>(), db.as<uint8_t>(), n,
                           dout.as<int>());
        e = hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) {
        set_error("hamming256: %s", hipGetErrorString(e));
        return ORBGPU_EHIP;
    }
    return ORBGPU_OK;
}

} // extern "C"
