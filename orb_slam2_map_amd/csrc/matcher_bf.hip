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
// sweep 0, row i after at most i+1 sweeps; in practice 2-7 sweeps suffice and a sweep that changes
// nothing proves convergence.  The all-pairs Hamming work is done ONCE (k_bf_topk keeps the 4 best B
// rows of every A row); the sweeps (k_bf_resolve, one workgroup per pair, claim tables in LDS) only walk
// those cached lists and fall back to an exact full scan of a row when its list cannot decide.
#include "common.h"
#include "matcher_common.h"
#include "workspace.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <new>
#include <vector>

namespace orbgpu {

constexpr int BF_TOPK = 4;
constexpr uint32_t BF_KEY_NONE = 0xFFFFFFFFu;  // "no further candidate"

// key = distance << 12 | B index: ascending keys = the order in which the reference's sequential scan
// would prefer candidates (smaller distance first, lower index first among equals)
__device__ __forceinline__ uint32_t bf_key(int dist, int j) { return ((uint32_t)dist << 12) | (uint32_t)j; }
__device__ __forceinline__ int bf_key_dist(uint32_t k) { return k == BF_KEY_NONE ? 256 : (int)(k >> 12); }
__device__ __forceinline__ int bf_key_idx(uint32_t k) { return (int)(k & 0xFFFu); }

__device__ __forceinline__ void cswap(uint32_t &a, uint32_t &b)
{
    const uint32_t lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

// insert into a sorted 4-list (keeps the 4 smallest)
__device__ __forceinline__ void top4_insert(uint32_t t[4], uint32_t k)
{
    if (k < t[3]) {
        t[3] = k;
        cswap(t[2], t[3]);
        cswap(t[1], t[2]);
        cswap(t[0], t[1]);
    }
}

// 4 smallest of two sorted 4-lists (bitonic split + sort)
__device__ __forceinline__ void top4_merge(uint32_t t[4], const uint32_t o[4])
{
    uint32_t c0 = min(t[0], o[3]), c1 = min(t[1], o[2]), c2 = min(t[2], o[1]), c3 = min(t[3], o[0]);
    cswap(c0, c2);
    cswap(c1, c3);
    cswap(c0, c1);
    cswap(c2, c3);
    cswap(c1, c2);
    t[0] = c0;
    t[1] = c1;
    t[2] = c2;
    t[3] = c3;
}

// Pass 1: for every A row the BF_TOPK best B rows over ALL B rows (claims are applied in pass 2).
//
// All-pairs Hamming distance is a GEMM: with every bit b mapped to the signed byte 2b-1 (+1 / -1), the dot product
// of two 256-byte rows is 256 - 2 * hamming.  The matrix cores (idle everywhere else on this path) evaluate it with
// v_mfma_i32_32x32x32_i8: a wave owns 32 A rows, whose expanded descriptors stay in 32 VGPRs as the MFMA's B operand
// (N = A row = the accumulator's COLUMN = lane & 31), and walks the B rows in tiles of 32 (the MFMA's A operand,
// M = B row = accumulator ROW), 8 MFMAs per tile.  Each lane then holds 16 distances of ONE A row to 16 different B
// rows and feeds them to its private sorted top-4 (key = distance << 12 | B index); lanes l and l + 32 share an A
// row and merge at the end.  The 8 waves of a workgroup (256 A rows) share every B tile: the workgroup expands the
// tile's 64 x 256 bits to bytes once (one dword per thread -> 32 B) into LDS, double buffered, and each wave reads its
// 16 fragments back with ds_read_b128 (row stride 272 B: conflict free).  Lane maps of the i8 MFMA verified with
// exact integer data: tools/ubench/mfma_i8_probe.hip.
// Only B rows within dmax of the A row are listed, dmax = the largest distance that can change a decision of pass 2:
// a best beyond th_low is rejected whatever it is, and a second best b2 with nnratio * b2 > th_low passes every
// ratio test (ORBmatcher.cc:228-231), so a list that ends early means "nothing else matters" exactly like a list
// that ran out of B rows.
// grid = (ceil(cap / 256), pairs, nsplit): the B rows are split into nsplit contiguous ranges to put enough
// waves in flight (none for a full batch -- every range keeps a top-4 of its own, so two ranges insert almost twice as
// often as one: 0.30 -> 0.28 ms per 512 pairs --, up to BF_MAX_SPLIT for a single pair, where the scan is latency bound);
// k_bf_resolve merges the partial lists (4 smallest of their union).
constexpr int BF_MIN_SPLIT = 1, BF_MAX_SPLIT = 16;
constexpr size_t BF_RESOLVE_MAX_LDS = 150 * 1024;  // of the CU's 160 KB; static LDS of k_bf_resolve is < 1 KB
constexpr int BF_TILE_STRIDE = 272;                // bytes per expanded B row in LDS (256 + 16: bank spread)
constexpr int BF_TILE_ROWS = 64;                   // B rows staged per iteration
constexpr int BF_TOPK_THREADS = 512;               // 8 waves = 256 A rows share every staged B tile
constexpr int BF_TOPK_ROWS = BF_TOPK_THREADS / 64 * 32;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 16 descriptor bits -> 16 signed bytes (bit 1 -> +1, bit 0 -> -1); byte j <- bit j.
// (n * 0x00204081) & 0x01010101 spreads the 4 bits of a nibble to the low bits of 4 bytes (a 24-bit multiply: both
// factors are small; 32-bit integer multiplies are quarter rate), and the 0 / 1 bytes are themselves the v_perm
// selector that picks 0xFF or 0x01 out of the constant 0x000001FF.
// PM = the two byte values: 0x01FF gives +1 / -1, 0x10F0 gives +16 / -16 (the A rows of k_bf_topk, see there).
template <uint32_t PM = 0x000001FFu> __device__ __forceinline__ v4i bf_expand16(uint32_t hw)
{
    v4i r;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t n = (hw >> (4 * q)) & 0xFu;
        const uint32_t w01 = __umul24(n, 0x00204081u) & 0x01010101u;
        r[q] = (int)__builtin_amdgcn_perm(0u, PM, w01);
    }
    return r;
}

// 16 descriptor bits -> 16 bytes 0 / 1 (the staged B rows of k_bf_topk: one v_perm per dword less than +-1 bytes)
__device__ __forceinline__ v4i bf_expand16_01(uint32_t hw)
{
    v4i r;
#pragma unroll
    for (int q = 0; q < 4; q++)
        r[q] = (int)(__umul24((hw >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u);
    return r;
}

// NODES: the node-wise matcher (SearchByBoW on real feature vectors): a B row is a candidate of an A row only if both
// sit under the same vocabulary node (node_a / node_b, [pairs][cap], < 0 = not in the feature vector).
template <bool NODES>
__global__ __launch_bounds__(BF_TOPK_THREADS) void k_bf_topk(int cap, const uint8_t *__restrict__ desc_a,
                                                 const int *__restrict__ na_p, const uint8_t *__restrict__ desc_b,
                                                 const int *__restrict__ nb_p, uint32_t *__restrict__ topk,
                                                 int acc_min, const int *__restrict__ node_a,
                                                 const int *__restrict__ node_b)
{
    __shared__ __align__(16) uint8_t s_tile[2][BF_TILE_ROWS * BF_TILE_STRIDE];
    const int nsplit = gridDim.z;
    const int pair = blockIdx.y;
    const int na = min(na_p[pair], cap);
    const int nb = min(nb_p[pair], cap);
    if ((int)(blockIdx.x * BF_TOPK_ROWS) >= na)
        return;  // the whole workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int i = blockIdx.x * BF_TOPK_ROWS + wave * 32 + r;  // this lane's A row (= accumulator column)
    int my_node = 0;
    const int *nodes_b = nullptr;
    if (NODES) {
        my_node = node_a[(size_t)pair * cap + min(i, na - 1)];
        nodes_b = node_b + (size_t)pair * cap;
    }
    // the A rows of this wave as the MFMA's B operand: fragment s holds k = 32 s + 16 h .. + 15 = halfword 2 s + h
    v4i fa[8];
    {
        const uint32_t *ga = reinterpret_cast<const uint32_t *>(desc_a + ((size_t)pair * cap + min(i, na - 1)) * 32);
#pragma unroll
        for (int s = 0; s < 8; s++)
            fa[s] = bf_expand16<0x000020E0u>(ga[s] >> (16 * h));  // +-32, against B bytes 0 / 1: see the accumulator start below
    }
    const int per = (nb + nsplit - 1) / nsplit;
    const int jbeg = (int)blockIdx.z * per, jend = min(nb, jbeg + per);
    const int ntiles = (jend - jbeg + BF_TILE_ROWS - 1) / BF_TILE_ROWS;
    const uint32_t *gb = reinterpret_cast<const uint32_t *>(desc_b + (size_t)pair * cap * 32);
    // staging: a tile is BF_TILE_ROWS = 64 B rows (two MFMA row blocks, i.e. two independent accumulator chains per
    // iteration and half as many barriers); thread t expands dword (t & 7) of tile row (t >> 3): two
    // fragments = 32 contiguous bytes.  The dword of tile + 2 is requested while tile is multiplied, so its latency
    // never sits in front of an expansion.
    const int sm = tid >> 3, sw = tid & 7;  // tile row 0..63, dword 0..7: one dword per thread
    auto fetch = [&](int tile) {
        const int j = min(jbeg + tile * BF_TILE_ROWS + sm, nb - 1);  // past the end: repeats (never selected)
        return gb[(size_t)j * 8 + sw];
    };
    auto stage = [&](uint32_t d, int buf) {
        v4i *dst = reinterpret_cast<v4i *>(&s_tile[buf][sm * BF_TILE_STRIDE + sw * 32]);
        dst[0] = bf_expand16_01(d);
        dst[1] = bf_expand16_01(d >> 16);
    };
    // Accumulator start.  With a_k = +-1 the A row, b_k = 0 / 1 the B row:  sum 32 a_k b_k = 16 (dot + S_a), dot = sum a_k
    // (2 b_k - 1) = 256 - 2 hamming and S_a = sum a_k = 2 popcount(A row) - 256, a constant of the lane (the A row is the
    // accumulator's column).  Starting the chain from (15 - reg) - 16 S_a leaves acc[reg] = dot << 4 | 15 - reg.
    v16i tag;
    {
        const uint32_t *ga = reinterpret_cast<const uint32_t *>(desc_a + ((size_t)pair * cap + min(i, na - 1)) * 32);
        int pop = 0;
#pragma unroll
        for (int s = 0; s < 8; s++)
            pop += __popc(ga[s]);
        const int sa16 = 16 * (2 * pop - 256);
#pragma unroll
        for (int reg = 0; reg < 16; reg++)
            tag[reg] = (15 - reg) - sa16;
    }
    uint32_t t[4] = {BF_KEY_NONE, BF_KEY_NONE, BF_KEY_NONE, BF_KEY_NONE};
    int lim = acc_min;  // accumulator value a distance must reach to be inserted
    uint32_t d_next = 0;
    if (ntiles > 0)
        stage(fetch(0), 0);
    if (ntiles > 1)
        d_next = fetch(1);
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const uint32_t d_cur = d_next;
        if (tile + 2 < ntiles)
            d_next = fetch(tile + 2);
        if (tile + 1 < ntiles)
            stage(d_cur, (tile + 1) & 1);
        const uint8_t *tb = &s_tile[tile & 1][r * BF_TILE_STRIDE + h * 16];
        // The selection below wants every value tagged with its register index, e = dot << 4 | 15 - reg.  The matrix
        // core does it: the A rows are expanded to +-32 instead of +-1 and the chain starts from the lane's constant
        // accumulator (above), so acc[reg] IS e (32 v_lshl_or per tile and wave less, and the staged B rows stay 0 / 1).
        v16i acc0 = tag, acc1 = tag;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const v4i fb0 = *reinterpret_cast<const v4i *>(tb + s * 32);
            const v4i fb1 = *reinterpret_cast<const v4i *>(tb + 32 * BF_TILE_STRIDE + s * 32);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb0, fa[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb1, fa[s], acc1, 0, 0, 0);
        }
        // acc[reg] >> 4 = 256 - 2 * hamming(A row i, B row j0 + (reg & 3) + 8 (reg >> 2) + 4 h).  Only distances <= dmax
        // can influence a decision (see the host side), i.e. acc >= acc_min: most row blocks hold no such value for any
        // lane of the wave and are dismissed with 8 v_max3 and one branch.
        // Selection.  A lane's 16 values per row block rarely hold anything that beats its list, but SOME lane of the
        // wave nearly always does, so a test-and-branch per value would run every insert body for the whole wave.
        // Instead each value is tagged with its register index (e = acc << 4 | 15 - reg: larger = closer, then lower
        // B row), a lane's best candidate is one v_max3 tree away, and the wave loops only while some lane still has
        // a candidate above its bar: one insert per lane per trip, typically one or two trips per block.
        const int jlast = jend - 1;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const v16i &acc = u ? acc1 : acc0;
            int e[16];
#pragma unroll
            for (int reg = 0; reg < 16; reg++)
                e[reg] = acc[reg];
            const int j0 = jbeg + tile * BF_TILE_ROWS + 32 * u + 4 * h;
            for (;;) {
                int m = max(max(e[0], e[1]), e[2]);
#pragma unroll
                for (int reg = 3; reg < 15; reg += 2)
                    m = max(max(m, e[reg]), e[reg + 1]);
                m = max(m, e[15]);
                const bool hit = (m >> 4) >= lim;
                if (!__any(hit))
                    break;
                if (hit) {
                    const int reg = 15 - (m & 15);
                    const int j = j0 + (reg & 3) + 8 * (reg >> 2);
                    if (j <= jlast && (!NODES || nodes_b[j] == my_node)) {
                        top4_insert(t, (uint32_t)(((256 - (m >> 4)) << 11) + j));  // (2 ham) << 11 = ham << 12
                        if (t[3] != BF_KEY_NONE)
                            lim = max(lim, 256 - 2 * (int)(t[3] >> 12));  // a full list admits distances <= its last
                    }
#pragma unroll
                    for (int r2 = 0; r2 < 16; r2++)
                        e[r2] = e[r2] == m ? INT_MIN : e[r2];  // taken
                }
            }
        }
        __syncthreads();
    }
    // lanes l and l + 32 hold the same A row
    {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            o[k] = __shfl_xor(t[k], 32, 64);
        top4_merge(t, o);
    }
    if (h == 0 && i < na)
        *reinterpret_cast<uint4 *>(topk + (((size_t)pair * cap + i) * nsplit + blockIdx.z) * BF_TOPK) =
            make_uint4(t[0], t[1], t[2], t[3]);
}

// Exact best/second over the B rows visible to A row i: full scan by one wave (lanes stride over the
// B rows; ties resolve to the lower index exactly as the sequential scan does).  Used only when the
// cached top-K list cannot decide.
__device__ __forceinline__ void bf_full_scan_wave(const uint64_t *__restrict__ ga, const uint64_t *gb, int i, int nb,
                                  const int *claim, int &b1, int &i1, int &b2, const int *__restrict__ nodes_b,
                                  int my_node)
{
    const int lane = threadIdx.x & 63;
    uint64_t a[4] = {ga[(size_t)i * 4], ga[(size_t)i * 4 + 1], ga[(size_t)i * 4 + 2], ga[(size_t)i * 4 + 3]};
    int v1 = 256, vi = INT_MAX, v2 = 256;
    for (int j = lane; j < nb; j += 64) {
        if (claim[j] < i)
            continue;
        if (nodes_b && nodes_b[j] != my_node)
            continue;
        uint64_t b[4] = {gb[(size_t)j * 4], gb[(size_t)j * 4 + 1], gb[(size_t)j * 4 + 2], gb[(size_t)j * 4 + 3]};
        const int d = hamming256(a, b);
        if (d < v1) {
            v2 = v1;
            v1 = d;
            vi = j;
        } else if (d < v2) {
            v2 = d;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o1 = __shfl_xor(v1, off, 64), oi = __shfl_xor(vi, off, 64), o2 = __shfl_xor(v2, off, 64);
        const bool other_first = (o1 < v1) || (o1 == v1 && oi < vi);
        const int n2 = min(min(v2, o2), max(v1, o1));
        v1 = other_first ? o1 : v1;
        vi = other_first ? oi : vi;
        v2 = n2;
    }
    b1 = v1;
    i1 = vi == INT_MAX ? -1 : vi;
    b2 = v2;
}

// merged candidate list of row i: the 4 smallest keys over the nsplit partial lists
__device__ __forceinline__ void bf_load_keys(const uint32_t *__restrict__ tk, int i, int nsplit, uint32_t key[4])
{
    const uint4 k4 = *reinterpret_cast<const uint4 *>(tk + (size_t)i * nsplit * BF_TOPK);
    key[0] = k4.x, key[1] = k4.y, key[2] = k4.z, key[3] = k4.w;
    for (int sp = 1; sp < nsplit; sp++) {
        const uint4 o4 = *reinterpret_cast<const uint4 *>(tk + ((size_t)i * nsplit + sp) * BF_TOPK);
        const uint32_t o[4] = {o4.x, o4.y, o4.z, o4.w};
        top4_merge(key, o);
    }
}

// Pass 2: the greedy claim order of the reference as a fixpoint, one workgroup per pair.  Every sweep
// re-decides every A row in parallel from its cached candidate list, hiding the B rows that rows < i
// claimed in the previous sweep (two claim tables in LDS).  Row i is final after at most i+1 sweeps and
// a sweep without changes is the sequential result, so the loop (bounded by na+1) is always exact.
// Then: scatter A->B, rotation-histogram consistency (ORBmatcher.cc:262-285), count.
__global__ __launch_bounds__(1024) void k_bf_resolve(int cap, const uint8_t *__restrict__ desc_a,
                                                     const uint8_t *__restrict__ valid_a,
                                                     const int *__restrict__ na_p,
                                                     const uint8_t *__restrict__ desc_b,
                                                     const int *__restrict__ nb_p,
                                                     const uint32_t *__restrict__ topk, int th_low, float nnratio,
                                                     const uint8_t *__restrict__ angle_a,
                                                     const uint8_t *__restrict__ angle_b, size_t angle_stride,
                                                     int check_orientation, int *__restrict__ match_b,
                                                     int *__restrict__ nmatches, int *__restrict__ sweeps_used,
                                                     int stage_b, int nsplit, const int *__restrict__ node_a,
                                                     const int *__restrict__ node_b)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int histo[ORBGPU_HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count, s_changed, s_nslow;
    const int pair = blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int na = min(na_p[pair], cap), nb = min(nb_p[pair], cap);
    int *claimA = reinterpret_cast<int *>(smem);  // [cap] read
    int *claimB = claimA + cap;                   // [cap] written
    int *match = claimB + cap;                    // [cap]
    int *slow = match + cap;                      // [cap] rows whose cached list cannot decide
    const uint64_t *ga = reinterpret_cast<const uint64_t *>(desc_a + (size_t)pair * cap * 32);
    const uint64_t *gb = reinterpret_cast<const uint64_t *>(desc_b + (size_t)pair * cap * 32);
    const uint32_t *tk = topk + (size_t)pair * cap * nsplit * BF_TOPK;
    const uint8_t *va = valid_a ? valid_a + (size_t)pair * cap : nullptr;
    const int *nda = node_a ? node_a + (size_t)pair * cap : nullptr;
    const int *ndb = node_b ? node_b + (size_t)pair * cap : nullptr;
    int *mb = match_b + (size_t)pair * cap;

    for (int j = tid; j < cap; j += nt) {
        claimA[j] = INT_MAX;
        match[j] = -2;  // "undecided": differs from every possible result
    }
    // the full scans of undecidable rows read every B descriptor: keep them in LDS when they fit
    const uint64_t *scan_b = gb;
    if (stage_b) {
        uint4 *sb = reinterpret_cast<uint4 *>(slow + cap);
        const uint4 *g4 = reinterpret_cast<const uint4 *>(gb);
        for (int x = tid; x < nb * 2; x += nt)
            sb[x] = g4[x];
        scan_b = reinterpret_cast<const uint64_t *>(sb);
    }
    // the candidate list of row `tid` (the only row of this thread unless na > blockDim) stays in registers
    uint32_t key0[4] = {BF_KEY_NONE, BF_KEY_NONE, BF_KEY_NONE, BF_KEY_NONE};
    uint8_t valid0 = 0;
    if (tid < na) {
        bf_load_keys(tk, tid, nsplit, key0);
        valid0 = ((!va || va[tid]) && (!nda || nda[tid] >= 0)) ? 1 : 0;
    }
    __syncthreads();
    int sweeps = 0;
    for (int iter = 0; iter <= na + 1; iter++) {
        for (int j = tid; j < nb; j += nt)
            claimB[j] = INT_MAX;
        if (tid == 0) {
            s_changed = 0;
            s_nslow = 0;
        }
        __syncthreads();
        bool changed = false;
        // (a) one thread per row: decide from the cached candidate list; undecidable rows are queued
        for (int i = tid; i < na; i += nt) {
            int result = -1;
            bool decided = true;
            if (i == tid ? valid0 != 0 : ((!va || va[i]) && (!nda || nda[i] >= 0))) {
                uint32_t key[4] = {key0[0], key0[1], key0[2], key0[3]};
                if (i != tid)
                    bf_load_keys(tk, i, nsplit, key);
                int b1 = 256, i1 = -1, b2 = 256, found = 0;
                bool complete = false;  // list exhausted: every B row has been considered
#pragma unroll
                for (int k = 0; k < BF_TOPK; k++) {
                    if (key[k] == BF_KEY_NONE) {
                        complete = true;
                    } else if (found < 2 && !(claimA[bf_key_idx(key[k])] < i)) {
                        if (found == 0) {
                            b1 = bf_key_dist(key[k]);
                            i1 = bf_key_idx(key[k]);
                        } else {
                            b2 = bf_key_dist(key[k]);
                        }
                        found++;
                    }
                }
                // a visible best that already fails the threshold needs no second distance; neither does one that
                // passes the ratio test against the LAST cached distance: every B row outside the list is at least
                // that far, so the true second distance b2 >= d4 and b1 < r*d4 implies b1 < r*b2 (:228-231)
                decided = found == 2 || complete || (found == 1 && b1 > th_low);
                if (!decided && found == 1 && (float)b1 < nnratio * (float)bf_key_dist(key[BF_TOPK - 1])) {
                    decided = true;
                    b2 = bf_key_dist(key[BF_TOPK - 1]);
                }
                if (decided && b1 <= th_low && (float)b1 < nnratio * (float)b2)  // ORBmatcher.cc:228-231
                    result = i1;
            }
            if (!decided) {
                slow[atomicAdd(&s_nslow, 1)] = i;
                continue;
            }
            if (match[i] != result) {
                changed = true;
                match[i] = result;
            }
            if (result >= 0)
                atomicMin(&claimB[result], i);
        }
        __syncthreads();
        // (b) one wave per queued row: exact full scan
        {
            const int nslow = s_nslow;
            const int wave = tid >> 6, nw = nt >> 6;
            for (int r = wave; r < nslow; r += nw) {
                const int i = slow[r];
                int b1, i1, b2;
                bf_full_scan_wave(ga, scan_b, i, nb, claimA, b1, i1, b2, ndb, nda ? nda[i] : 0);
                if ((tid & 63) == 0) {
                    int result = -1;
                    if (b1 <= th_low && (float)b1 < nnratio * (float)b2)
                        result = i1;
                    if (match[i] != result) {
                        changed = true;
                        match[i] = result;
                    }
                    if (result >= 0)
                        atomicMin(&claimB[result], i);
                }
            }
        }
        if (changed)
            s_changed = 1;
        __syncthreads();
        sweeps++;
        const bool again = s_changed != 0;
        __syncthreads();
        if (!again)
            break;
        int *t = claimA;
        claimA = claimB;
        claimB = t;
    }

    // ---- finish
    for (int j = tid; j < cap; j += nt)
        mb[j] = -1;
    if (tid < ORBGPU_HISTO_LENGTH)
        histo[tid] = 0;
    if (tid == 0)
        s_count = 0;
    __syncthreads();
    const uint8_t *aa = angle_a + (size_t)pair * cap * angle_stride;
    const uint8_t *ab = angle_b + (size_t)pair * cap * angle_stride;
    int cnt = 0;
    for (int i = tid; i < na; i += nt) {
        const int j = match[i];
        if (j < 0)
            continue;
        cnt++;
        if (check_orientation) {
            const float fa = *reinterpret_cast<const float *>(aa + (size_t)i * angle_stride);
            const float fb = *reinterpret_cast<const float *>(ab + (size_t)j * angle_stride);
            atomicAdd(&histo[rot_bin(fa, fb)], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int i1 = -1, i2 = -1, i3 = -1;
        if (check_orientation)
            three_maxima(histo, ORBGPU_HISTO_LENGTH, i1, i2, i3);
        s_keep[0] = i1;
        s_keep[1] = i2;
        s_keep[2] = i3;
    }
    __syncthreads();
    for (int i = tid; i < na; i += nt) {
        const int j = match[i];
        if (j < 0)
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
    if ((tid & 63) == 0)
        atomicAdd(&s_count, cnt);
    __syncthreads();
    if (tid == 0) {
        nmatches[pair] = s_count;
        sweeps_used[pair] = sweeps;
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

// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:242-307): among the n descriptors observing a
// map point pick the one whose MEDIAN Hamming distance to all of them (itself included, distance 0) is least;
// median = sorted row [ (size_t)(0.5 * (n - 1)) ], the first minimum wins.  One workgroup per map point, its
// descriptors in LDS; a thread owns rows i = tid, tid + 256, ... and finds the row's k-th smallest distance by
// bisection on the value range [0, 256] (9 counting passes over the row) instead of sorting it.
constexpr int DD_MAX = 2048;  // descriptors per map point (64 KB of LDS)

__global__ __launch_bounds__(256) void k_distinctive(const int *__restrict__ offsets,
                                                     const uint8_t *__restrict__ desc, int *__restrict__ best_idx)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ unsigned long long s_best;
    uint64_t *sd = reinterpret_cast<uint64_t *>(smem);
    const int g = blockIdx.x, tid = threadIdx.x;
    const int beg = offsets[g], n = offsets[g + 1] - beg;
    if (n <= 0) {
        if (tid == 0)
            best_idx[g] = -1;  // the reference returns without touching mDescriptor
        return;
    }
    const uint64_t *gd = reinterpret_cast<const uint64_t *>(desc) + (size_t)beg * 4;
    for (int x = tid; x < n * 4; x += 256)
        sd[x] = gd[x];
    if (tid == 0)
        s_best = ~0ull;
    __syncthreads();
    const int k = (n - 1) >> 1;  // (size_t)(0.5 * (n - 1))
    unsigned long long mine = ~0ull;
    for (int i = tid; i < n; i += 256) {
        const uint64_t a[4] = {sd[i * 4], sd[i * 4 + 1], sd[i * 4 + 2], sd[i * 4 + 3]};
        int lo = 0, hi = 256;  // smallest v with #{j : d_ij <= v} >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < n; j++) {
                const uint64_t b[4] = {sd[j * 4], sd[j * 4 + 1], sd[j * 4 + 2], sd[j * 4 + 3]};
                cnt += hamming256(a, b) <= mid ? 1 : 0;
            }
            if (cnt >= k + 1)
                hi = mid;
            else
                lo = mid + 1;
        }
        const unsigned long long key = ((unsigned long long)lo << 32) | (unsigned)i;  // least median, then least index
        mine = key < mine ? key : mine;
    }
    atomicMin(&s_best, mine);
    __syncthreads();
    if (tid == 0)
        best_idx[g] = (int)(s_best & 0xFFFFFFFFull);
}

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_matcher {
    int device_id = 0;
    int max_pairs = 0, cap = 0;
    DevBuf d_topk, d_sweeps;
    int last_pairs = 0;
    int resolve_threads = 1024;  // workgroup size of k_bf_resolve (ORBGPU_DEBUG_BF_RESOLVE_THREADS: 256 / 512 / 1024)
};

extern "C" {

int orbgpu_matcher_create(int32_t device_id, int32_t max_pairs, int32_t cap, orbgpu_matcher **out)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
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
    if (const char *q = getenv("ORBGPU_DEBUG_BF_RESOLVE_THREADS")) {
        const int v = atoi(q);
        if (v == 256 || v == 512 || v == 1024)
            m->resolve_threads = v;
    }
    const size_t P = (size_t)max_pairs;
    if ((rc = m->d_topk.reserve(sizeof(uint32_t) * std::max<size_t>(P * BF_MIN_SPLIT, BF_MAX_SPLIT) * cap * BF_TOPK)) != ORBGPU_OK ||
        (rc = m->d_sweeps.reserve(sizeof(int) * P)) != ORBGPU_OK) {
        orbgpu_matcher_destroy(m);
        return rc;
    }
    hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void *>(k_bf_resolve),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, BF_RESOLVE_MAX_LDS);
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
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!m)
        return ORBGPU_OK;
    (void)hipSetDevice(m->device_id);
    (void)hipDeviceSynchronize();
    m->d_topk.release();
    m->d_sweeps.release();
    delete m;
    return ORBGPU_OK;
}

static int match_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                              const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_node_a, const int32_t *d_na,
                              const uint8_t *d_desc_b, const void *d_angle_b, const int32_t *d_node_b, const int32_t *d_nb,
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
    uint32_t *topk = m->d_topk.as<uint32_t>();
    // enough waves to cover the device (~2048) while the partial lists fit the buffer sized at creation
    int nsplit = BF_MIN_SPLIT;
    while (nsplit < BF_MAX_SPLIT && (size_t)pairs * ((cap + BF_TOPK_ROWS - 1) / BF_TOPK_ROWS) * (BF_TOPK_THREADS / 64) * nsplit < 2048 &&
           (size_t)pairs * nsplit * 2 <= std::max<size_t>((size_t)m->max_pairs * BF_MIN_SPLIT, BF_MAX_SPLIT))
        nsplit *= 2;
    const dim3 grid((cap + BF_TOPK_ROWS - 1) / BF_TOPK_ROWS, pairs, nsplit);
    // largest second-best distance that can still fail the ratio test for some best <= th_low (same float
    // arithmetic as the test itself, which is monotone in the second distance)
    int dmax = std::min(std::max(th_low, 0), 256);
    while (dmax < 256 && !(nnratio * (float)(dmax + 1) > (float)th_low))
        dmax++;
    if (d_node_a)
        hipLaunchKernelGGL(k_bf_topk<true>, grid, dim3(BF_TOPK_THREADS), 0, st, cap, d_desc_a, d_na, d_desc_b, d_nb, topk,
                           256 - 2 * dmax, d_node_a, d_node_b);
    else
        hipLaunchKernelGGL(k_bf_topk<false>, grid, dim3(BF_TOPK_THREADS), 0, st, cap, d_desc_a, d_na, d_desc_b, d_nb, topk,
                           256 - 2 * dmax, (const int *)nullptr, (const int *)nullptr);
    // claim / match / queue tables (16 B per row) + the B descriptors (32 B per row) when both fit in LDS
    const int stage_b = (size_t)48 * cap <= BF_RESOLVE_MAX_LDS ? 1 : 0;
    const size_t lds = (size_t)(stage_b ? 48 : 16) * cap;
    hipLaunchKernelGGL(k_bf_resolve, dim3(pairs), dim3(m->resolve_threads), lds, st, cap, d_desc_a, d_valid_a, d_na,
                       d_desc_b, d_nb, topk, th_low, nnratio, reinterpret_cast<const uint8_t *>(d_angle_a),
                       reinterpret_cast<const uint8_t *>(d_angle_b), angle_stride, check_orientation, d_match_b,
                       d_nmatches, m->d_sweeps.as<int>(), stage_b, nsplit, d_node_a, d_node_b);
    ORBGPU_HIP_TRY(hipGetLastError());
    m->last_pairs = pairs;
    return ORBGPU_OK;
}

int orbgpu_match_bf_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                                 const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_na,
                                 const uint8_t *d_desc_b, const void *d_angle_b, const int32_t *d_nb,
                                 size_t angle_stride, int32_t th_low, float nnratio, int32_t check_orientation,
                                 int32_t *d_match_b, int32_t *d_nmatches, void *hip_stream)
{
    return match_batch_device(m, pairs, cap, d_desc_a, d_angle_a, d_valid_a, nullptr, d_na, d_desc_b, d_angle_b, nullptr,
                              d_nb, angle_stride, th_low, nnratio, check_orientation, d_match_b, d_nmatches, hip_stream);
}

int orbgpu_search_by_bow_batch_device(orbgpu_matcher *m, int32_t pairs, int32_t cap, const uint8_t *d_desc_a,
                                      const void *d_angle_a, const uint8_t *d_valid_a, const int32_t *d_node_a,
                                      const int32_t *d_na, const uint8_t *d_desc_b, const void *d_angle_b,
                                      const int32_t *d_node_b, const int32_t *d_nb, size_t angle_stride, int32_t th_low,
                                      float nnratio, int32_t check_orientation, int32_t *d_match_b,
                                      int32_t *d_nmatches, void *hip_stream)
{
    ORBGPU_REQUIRE(d_node_a && d_node_b, "null node arrays");
    return match_batch_device(m, pairs, cap, d_desc_a, d_angle_a, d_valid_a, d_node_a, d_na, d_desc_b, d_angle_b, d_node_b,
                              d_nb, angle_stride, th_low, nnratio, check_orientation, d_match_b, d_nmatches, hip_stream);
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

static int match_host(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, const int32_t *node_a,
                      int32_t na, const uint8_t *desc_b, const float *angle_b, const int32_t *node_b, int32_t nb,
                      int32_t th_low, float nnratio, int32_t check_orientation, int32_t *match_b, int32_t *nmatches,
                      int32_t device_id)
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
    // per-thread workspace (matcher handle + staging buffers) reused across calls: the reference builds an
    // ORBmatcher on the stack per call site, so allocating per call would dominate the kernel time
    struct Ws {
        int device = -1, cap = 0;
        orbgpu_matcher *m = nullptr;
        hipStream_t st = nullptr;
        DevBuf da, db, aa, ab, va, cnt, mb, nm, nda, ndb;
        ~Ws()  // end of the owning thread: release, unless the process is exiting (workspace.h)
        {
            if (device < 0 || process_exiting().load())
                return;
            (void)hipSetDevice(device);
            if (st)
                (void)hipStreamSynchronize(st);
            if (m)
                orbgpu_matcher_destroy(m);
            for (DevBuf *b : {&da, &db, &aa, &ab, &va, &cnt, &mb, &nm, &nda, &ndb})
                b->release();
            if (st)
                (void)hipStreamDestroy(st);
        }
    };
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    Ws &ws = per_device_workspace<Ws>(device_id);  // stream, matcher handle and buffers of THIS device
    const int need = std::max(na, nb);
    if (ws.device != device_id || ws.cap < need) {
        if (ws.m)
            orbgpu_matcher_destroy(ws.m);
        ws.m = nullptr;
        const int cap = std::min(4096, std::max(need, 1024));
        if ((rc = orbgpu_matcher_create(device_id, 1, cap, &ws.m)) != ORBGPU_OK)
            return rc;
        if (!ws.st)
            ORBGPU_HIP_TRY(hipStreamCreateWithFlags(&ws.st, hipStreamNonBlocking));
        if ((rc = ws.da.reserve((size_t)cap * 32)) != ORBGPU_OK || (rc = ws.db.reserve((size_t)cap * 32)) != ORBGPU_OK ||
            (rc = ws.aa.reserve((size_t)cap * 4)) != ORBGPU_OK || (rc = ws.ab.reserve((size_t)cap * 4)) != ORBGPU_OK ||
            (rc = ws.va.reserve((size_t)cap)) != ORBGPU_OK || (rc = ws.cnt.reserve(8)) != ORBGPU_OK ||
            (rc = ws.mb.reserve((size_t)cap * 4)) != ORBGPU_OK || (rc = ws.nm.reserve(4)) != ORBGPU_OK ||
            (rc = ws.nda.reserve((size_t)cap * 4)) != ORBGPU_OK || (rc = ws.ndb.reserve((size_t)cap * 4)) != ORBGPU_OK)
            return rc;
        ws.device = device_id;
        ws.cap = cap;
    }
    const int cap = ws.cap;
    hipStream_t st = ws.st;
    ORBGPU_HIP_TRY(hipMemcpyAsync(ws.da.p, desc_a, (size_t)na * 32, hipMemcpyHostToDevice, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(ws.db.p, desc_b, (size_t)nb * 32, hipMemcpyHostToDevice, st));
    if (check_orientation) {
        ORBGPU_HIP_TRY(hipMemcpyAsync(ws.aa.p, angle_a, (size_t)na * 4, hipMemcpyHostToDevice, st));
        ORBGPU_HIP_TRY(hipMemcpyAsync(ws.ab.p, angle_b, (size_t)nb * 4, hipMemcpyHostToDevice, st));
    }
    if (valid_a)
        ORBGPU_HIP_TRY(hipMemcpyAsync(ws.va.p, valid_a, (size_t)na, hipMemcpyHostToDevice, st));
    if (node_a) {
        ORBGPU_HIP_TRY(hipMemcpyAsync(ws.nda.p, node_a, (size_t)na * 4, hipMemcpyHostToDevice, st));
        ORBGPU_HIP_TRY(hipMemcpyAsync(ws.ndb.p, node_b, (size_t)nb * 4, hipMemcpyHostToDevice, st));
    }
    const int counts[2] = {na, nb};
    ORBGPU_HIP_TRY(hipMemcpyAsync(ws.cnt.p, counts, 8, hipMemcpyHostToDevice, st));
    if ((rc = match_batch_device(ws.m, 1, cap, ws.da.as<uint8_t>(), ws.aa.p, valid_a ? ws.va.as<uint8_t>() : nullptr,
                                 node_a ? ws.nda.as<int>() : nullptr, ws.cnt.as<int>(), ws.db.as<uint8_t>(), ws.ab.p,
                                 node_a ? ws.ndb.as<int>() : nullptr, ws.cnt.as<int>() + 1, 4, th_low, nnratio,
                                 check_orientation, ws.mb.as<int>(), ws.nm.as<int>(), st)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpyAsync(match_b, ws.mb.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(nmatches, ws.nm.p, 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

int orbgpu_match_bf(const uint8_t *desc_a, const float *angle_a, const uint8_t *valid_a, int32_t na,
                    const uint8_t *desc_b, const float *angle_b, int32_t nb, int32_t th_low, float nnratio,
                    int32_t check_orientation, int32_t *match_b, int32_t *nmatches, int32_t device_id)
{
    return match_host(desc_a, angle_a, valid_a, nullptr, na, desc_b, angle_b, nullptr, nb, th_low, nnratio,
                      check_orientation, match_b, nmatches, device_id);
}

int orbgpu_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, const uint8_t *valid_kf,
                         const int32_t *node_kf, int32_t n_kf, const uint8_t *desc_f, const float *angle_f,
                         const int32_t *node_f, int32_t n_f, int32_t th_low, float nnratio, int32_t check_orientation,
                         int32_t *match_f, int32_t *nmatches, int32_t device_id)
{
    ORBGPU_REQUIRE((n_kf == 0 || node_kf) && (n_f == 0 || node_f), "null node arrays");
    return match_host(desc_kf, angle_kf, valid_kf, node_kf, n_kf, desc_f, angle_f, node_f, n_f, th_low, nnratio,
                      check_orientation, match_f, nmatches, device_id);
}

int orbgpu_search_by_bow_keyframes(const uint8_t *desc1, const float *angle1, const uint8_t *valid1,
                                   const int32_t *node1, int32_t n1, const uint8_t *desc2, const float *angle2,
                                   const uint8_t *valid2, const int32_t *node2, int32_t n2, float nnratio,
                                   int32_t check_orientation, int32_t *match12, int32_t *nmatches, int32_t device_id)
{
    // ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:522-655) is the frame version with
    // (a) candidates restricted to key points of key frame 2 that have a good map point (:569-575): they get a node
    // id no valid row carries, (b) a STRICT threshold `bestDist1 < TH_LOW` (:592) = `<= TH_LOW - 1` on integers,
    // (c) the result indexed by key frame 1: matches are one to one (vbMatched2), so the B-indexed table inverts.
    ORBGPU_REQUIRE(match12 && nmatches, "null argument");
    ORBGPU_REQUIRE(n1 >= 0 && n2 >= 0 && (n1 == 0 || node1) && (n2 == 0 || node2), "bad arguments");
    std::vector<int32_t> nd2((size_t)std::max(n2, 1)), mb((size_t)std::max(n2, 1), -1);
    for (int j = 0; j < n2; j++)
        nd2[j] = (!valid2 || valid2[j]) ? node2[j] : -2;
    int rc = match_host(desc1, angle1, valid1, node1, n1, desc2, angle2, nd2.data(), n2, ORBGPU_TH_LOW - 1, nnratio,
                        check_orientation, mb.data(), nmatches, device_id);
    if (rc != ORBGPU_OK)
        return rc;
    for (int i = 0; i < n1; i++)
        match12[i] = -1;
    for (int j = 0; j < n2; j++)
        if (mb[j] >= 0)
            match12[mb[j]] = j;
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

int orbgpu_distinctive_descriptors(int32_t groups, const int32_t *offsets, const uint8_t *desc, int32_t *best_idx,
                                   int32_t device_id)
{
    ORBGPU_REQUIRE(groups >= 0 && (groups == 0 || (offsets && best_idx)), "bad arguments");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK || groups == 0)
        return rc;
    int max_n = 0;
    ORBGPU_REQUIRE(offsets[0] == 0, "offsets[0] must be 0");
    for (int g = 0; g < groups; g++) {
        const int n = offsets[g + 1] - offsets[g];
        ORBGPU_REQUIRE(n >= 0 && n <= DD_MAX, "group %d has %d descriptors (limit %d)", g, n, DD_MAX);
        max_n = std::max(max_n, n);
    }
    const size_t total = (size_t)offsets[groups];
    ORBGPU_REQUIRE(total == 0 || desc, "null descriptors");
    struct Scoped : DevBuf {
        ~Scoped() { release(); }
    } d_off, d_desc, d_out;
    if ((rc = d_off.reserve(sizeof(int) * ((size_t)groups + 1))) != ORBGPU_OK ||
        (rc = d_desc.reserve(std::max<size_t>(total * 32, 32))) != ORBGPU_OK ||
        (rc = d_out.reserve(sizeof(int) * (size_t)groups)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpy(d_off.p, offsets, sizeof(int) * ((size_t)groups + 1), hipMemcpyHostToDevice));
    if (total)
        ORBGPU_HIP_TRY(hipMemcpy(d_desc.p, desc, total * 32, hipMemcpyHostToDevice));
    ORBGPU_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_distinctive),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DD_MAX * 32));
    hipLaunchKernelGGL(k_distinctive, dim3(groups), dim3(256), (size_t)std::max(max_n, 1) * 32, nullptr, d_off.as<int>(),
                       d_desc.as<uint8_t>(), d_out.as<int>());
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpy(best_idx, d_out.p, sizeof(int) * (size_t)groups, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}

} // extern "C"
