// Device-resident MapPoint table and device-resident Frame: the host side of the projection matchers made cheap.
//
// In the reference a map point is a heap object behind a mutex: ORBmatcher::SearchByProjection reads, per point and per
// call, GetDescriptor() (lock + 32-byte clone, MapPoint.cc:309-313), isBad(), Observations() and the mTrack* members
// (ORBmatcher.cc:53-63, 88) -- 10 k scattered objects for C3.  The table keeps what the matchers read of every map point
// (world position, normal, the scale-invariance distances, the distinctive descriptor, bad flag, "has observations")
// in HBM, keyed by MapPoint::mnId (MapPoint.h:84), and is edited where the reference edits the object: the
// constructors (MapPoint.cc:32-71), SetWorldPos (:73-78), AddObservation / EraseObservation (:98-149), SetBadFlag
// (:151-175), Replace (:177-228), ComputeDistinctiveDescriptors (:242-307), UpdateNormalAndDepth (:330-371).
// Per frame the caller then hands over ids, not objects: the id -> row lookup, the gather of the call's local map, the
// translation of the frame's existing associations and the matcher itself all run on the device.
//
// id -> row: an open-addressing hash table (linear probing, multiplicative hash) kept twice -- the host copy hands out
// rows when points are inserted, the device copy (patched slot by slot after every insert) serves the per-frame lookups.
#include "common.h"
#include "proj_internal.h"
#include "id_hash.h"

#include <algorithm>
#include <climits>
#include <cstring>
#include <new>
#include <vector>

namespace orbgpu {

constexpr int64_t MT_EMPTY = ID_HASH_EMPTY;
constexpr int MT_MAX_CALL = 1 << 26;  // ids per call
constexpr int MT_MAX_ROWS = 1 << 27;  // rows of a table (82 B each)
__host__ __device__ __forceinline__ uint32_t mt_hash(int64_t id, int log2cap) { return id_hash_slot(id, log2cap); }

__device__ __forceinline__ int mt_lookup(const int64_t *__restrict__ hkeys, const int32_t *__restrict__ hvals, int log2cap,
                                         int64_t id)
{
    if (id < 0)
        return -1;
    const uint32_t mask = (1u << log2cap) - 1u;
    uint32_t s = mt_hash(id, log2cap);
    for (uint32_t probe = 0; probe <= mask; probe++, s = (s + 1) & mask) {  // (the table is never full: load <= 1/2)
        const int64_t k = hkeys[s];
        if (k == id)
            return hvals[s];
        if (k == MT_EMPTY)
            return -1;
    }
    return -1;
}

struct TableDev {  // device arrays, one row per map point
    float *world_pos, *normal, *min_dist, *max_dist;
    uint8_t *desc, *bad, *obs;
    int64_t *id;
};

struct UpsertDev {  // staged attribute arrays of one upsert call (nullptr: keep what the row has)
    const int32_t *row;
    const uint8_t *is_new;
    const int64_t *id;
    const float *world_pos, *normal, *min_dist, *max_dist;
    const uint8_t *desc;
    const int32_t *n_obs;
};

__global__ __launch_bounds__(256) void k_table_scatter(int n, UpsertDev u, TableDev t)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int r = u.row[i];
    const bool fresh = u.is_new[i] != 0;
    if (fresh) {
        t.id[r] = u.id[i];
        t.bad[r] = 0;
    }
    if (u.world_pos || fresh)
        for (int c = 0; c < 3; c++)
            t.world_pos[3 * r + c] = u.world_pos ? u.world_pos[3 * i + c] : 0.f;
    if (u.normal || fresh)
        for (int c = 0; c < 3; c++)
            t.normal[3 * r + c] = u.normal ? u.normal[3 * i + c] : 0.f;
    if (u.min_dist || fresh)
        t.min_dist[r] = u.min_dist ? u.min_dist[i] : 0.f;
    if (u.max_dist || fresh)
        t.max_dist[r] = u.max_dist ? u.max_dist[i] : 0.f;
    if (u.desc || fresh) {
        uint4 *dst = reinterpret_cast<uint4 *>(t.desc + (size_t)r * 32);
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 *src = u.desc ? reinterpret_cast<const uint4 *>(u.desc + (size_t)i * 32) : nullptr;
        dst[0] = src ? src[0] : z;
        dst[1] = src ? src[1] : z;
    }
    if (u.n_obs || fresh)
        t.obs[r] = u.n_obs ? (u.n_obs[i] > 0 ? 1 : 0) : 1;  // a point inserted without a count is a regular map point
}

__global__ __launch_bounds__(256) void k_hash_apply(int n, const int32_t *__restrict__ slot, const int64_t *__restrict__ key,
                                                    const int32_t *__restrict__ val, int64_t *__restrict__ hkeys,
                                                    int32_t *__restrict__ hvals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        hvals[slot[i]] = val[i];
        hkeys[slot[i]] = key[i];
    }
}

__global__ __launch_bounds__(256) void k_table_flag(int n, const int32_t *__restrict__ row, const int32_t *__restrict__ n_obs,
                                                    uint8_t *__restrict__ bad, uint8_t *__restrict__ obs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || row[i] < 0)
        return;
    if (n_obs)
        obs[row[i]] = n_obs[i] > 0 ? 1 : 0;
    else
        bad[row[i]] = 1;
}

// orbgpu_mappoint_table_retain: row src[i] of the old arrays becomes row i of the new ones
__global__ __launch_bounds__(256) void k_table_compact(int n, const int32_t *__restrict__ src, TableDev o, TableDev d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int r = src[i];
    for (int c = 0; c < 3; c++) {
        d.world_pos[3 * i + c] = o.world_pos[3 * r + c];
        d.normal[3 * i + c] = o.normal[3 * r + c];
    }
    d.min_dist[i] = o.min_dist[r];
    d.max_dist[i] = o.max_dist[r];
    const uint4 *sd = reinterpret_cast<const uint4 *>(o.desc + (size_t)r * 32);
    uint4 *dd = reinterpret_cast<uint4 *>(d.desc + (size_t)i * 32);
    dd[0] = sd[0];
    dd[1] = sd[1];
    d.bad[i] = o.bad[r];
    d.obs[i] = o.obs[r];
    d.id[i] = o.id[r];
}

// The call's local map: thread i < m looks up ids[i] and copies the row into the call's arrays (what
// orbgpu_device_mappoint_table / orbgpu_device_lastframe_view point at).  pos_of_row[row] = first list position of the
// row (for the translation of the frame's existing associations); unknown ids are counted and skipped.
struct GatherOut {
    float *world_pos, *normal, *min_dist, *max_dist;
    uint8_t *desc, *skip, *obs, *has;
    int32_t *pos_of_row;
    int32_t *unknown;  // counter
};

__global__ __launch_bounds__(256) void k_table_gather(int m, const int64_t *__restrict__ ids,
                                                      const uint8_t *__restrict__ skip_in,
                                                      const int64_t *__restrict__ hkeys, const int32_t *__restrict__ hvals,
                                                      int log2cap, TableDev t, GatherOut g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    const int64_t id = ids[i];
    const int r = mt_lookup(hkeys, hvals, log2cap, id);
    if (g.has)
        g.has[i] = r >= 0 ? 1 : 0;
    if (r < 0) {
        if (id >= 0)
            atomicAdd(g.unknown, 1);  // an id the table was never told about: reported to the caller
        g.skip[i] = 1;
        g.obs[i] = 0;
        return;
    }
    for (int c = 0; c < 3; c++)
        g.world_pos[3 * i + c] = t.world_pos[3 * r + c];
    if (g.normal) {
        for (int c = 0; c < 3; c++)
            g.normal[3 * i + c] = t.normal[3 * r + c];
        g.min_dist[i] = t.min_dist[r];
        g.max_dist[i] = t.max_dist[r];
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(t.desc + (size_t)r * 32);
    uint4 *dst = reinterpret_cast<uint4 *>(g.desc + (size_t)i * 32);
    dst[0] = src[0];
    dst[1] = src[1];
    g.skip[i] = ((skip_in && skip_in[i]) || t.bad[r]) ? 1 : 0;
    g.obs[i] = t.obs[r];
    if (g.pos_of_row)
        atomicMin(&g.pos_of_row[r], i);
}

// F.mvpMapPoints as the matchers want it: thread j < cap writes k2m[j] = list position of the point key point j
// holds, -2 if it holds a point outside the list that has observations (ORBmatcher.cc:87-89), -1 otherwise.
__global__ __launch_bounds__(256) void k_table_kp(int n, int cap, int m, const int64_t *__restrict__ kp_ids,
                                                  const int64_t *__restrict__ hkeys, const int32_t *__restrict__ hvals,
                                                  int log2cap, const uint8_t *__restrict__ obs,
                                                  const int32_t *__restrict__ pos_of_row, int32_t *__restrict__ k2m,
                                                  int32_t *__restrict__ unknown)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cap)
        return;
    int v = -1;
    if (j < n && kp_ids[j] >= 0) {
        const int r = mt_lookup(hkeys, hvals, log2cap, kp_ids[j]);
        if (r < 0) {
            atomicAdd(unknown, 1);
            v = -2;
        } else {
            const int pos = pos_of_row ? pos_of_row[r] : INT_MAX;
            v = pos < m ? pos : (obs[r] ? -2 : -1);
        }
    }
    k2m[j] = v;
}

// SoA frame view -> the records of the device frame view (orbgpu_keypoint: x, y, size, angle, response, octave, class_id)
static void pack_keypoints(const orbgpu_frame_view *f, orbgpu_keypoint *out)
{
    for (int i = 0; i < f->n; i++) {
        orbgpu_keypoint k{};
        k.x = f->kp_x[i];
        k.y = f->kp_y[i];
        k.angle = f->kp_angle ? f->kp_angle[i] : 0.f;
        k.octave = f->kp_octave[i];
        k.class_id = -1;
        out[i] = k;
    }
}

struct Pinned {  // grow-only pinned staging buffer
    void *p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n)
    {
        if (n <= bytes)
            return ORBGPU_OK;
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t want = std::max<size_t>(n + n / 2, 1 << 16);
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
            p = nullptr;
            return ORBGPU_ENOMEM;
        }
        bytes = want;
        return ORBGPU_OK;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// carve sub-arrays out of one staging block (host and device blocks share the layout)
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes)
    {
        const size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    }
};

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_mappoint_table {
    int device_id = 0;
    hipStream_t stream = nullptr;
    int rows = 0, cap = 0;
    DevBuf world_pos, normal, min_dist, max_dist, desc, bad, obs, id;
    IdHash hash;                 // host copy of the id -> row hash (id_hash.h)
    std::vector<int32_t> stamp;  // per row: last upsert call that touched it (duplicate ids inside one call are refused)
    int call_no = 0;
    DevBuf d_hkeys, d_hvals;
    Pinned stage;       // host staging of a call's inputs / outputs
    DevBuf d_stage;     // its device twin
    DevBuf g_block;     // the call's gathered local map + translation tables + results
    DevBuf pos_of_row;  // [cap] list position of every table row in the current call
    ProjWorkspace *pws = nullptr;  // the matchers' scratch for calls over this table (they run on `stream`)
    int32_t last_unknown_list = 0, last_unknown_kp = 0;  // ids of the last search call the table had never been told about
    TableDev dev() const
    {
        return TableDev{world_pos.as<float>(), normal.as<float>(), min_dist.as<float>(), max_dist.as<float>(),
                        desc.as<uint8_t>(), bad.as<uint8_t>(), obs.as<uint8_t>(), id.as<int64_t>()};
    }
};

struct orbgpu_frame {
    int device_id = 0;
    hipStream_t stream = nullptr;
    int n = 0, cap = 0, nlevels = 0;
    float scale_factors[ORBGPU_MAX_LEVELS];
    float min_x = 0, max_x = 0, min_y = 0, max_y = 0;
    Pinned stage;
    DevBuf block;
    size_t o_n = 0, o_kps = 0, o_desc = 0, o_ur = 0, o_cs = 0, o_ci = 0;
};

namespace orbgpu {

static int host_find(const orbgpu_mappoint_table *t, int64_t id) { return t->hash.find(id); }
static uint32_t host_insert(orbgpu_mappoint_table *t, int64_t id, int row) { return t->hash.insert(id, row); }

// row capacity `want`, hash capacity >= 2 * want; existing rows are carried over on the device
static int table_grow(orbgpu_mappoint_table *t, int want)
{
    if (want <= t->cap)
        return ORBGPU_OK;
    int ncap = std::max(t->cap * 2, 1024);
    while (ncap < want)
        ncap *= 2;
    ORBGPU_HIP_TRY(hipStreamSynchronize(t->stream));
    struct Item {
        DevBuf *b;
        size_t elt;
    } items[] = {{&t->world_pos, 12}, {&t->normal, 12}, {&t->min_dist, 4}, {&t->max_dist, 4},
                 {&t->desc, 32},      {&t->bad, 1},     {&t->obs, 1},      {&t->id, 8}};
    for (const Item &it : items) {
        DevBuf nb;
        int rc = nb.reserve(it.elt * (size_t)ncap);
        if (rc != ORBGPU_OK)
            return rc;
        hipError_t he = hipMemsetAsync(nb.p, 0, it.elt * (size_t)ncap, t->stream);
        if (he == hipSuccess && t->rows > 0)
            he = hipMemcpyAsync(nb.p, it.b->p, it.elt * (size_t)t->rows, hipMemcpyDeviceToDevice, t->stream);
        if (he == hipSuccess)
            he = hipStreamSynchronize(t->stream);
        if (he != hipSuccess) {  // the array being replaced stays as it was; the fresh one must not leak
            nb.release();
            set_error("MapPoint table growth to %d rows: %s", ncap, hipGetErrorString(he));
            return ORBGPU_EHIP;
        }
        it.b->release();
        *it.b = nb;
    }
    int rc = t->pos_of_row.reserve(sizeof(int32_t) * (size_t)ncap);
    if (rc != ORBGPU_OK)
        return rc;
    t->stamp.resize((size_t)ncap, 0);
    t->cap = ncap;
    // hash: load factor <= 1/2, rebuilt from the rows' ids
    int l2 = 1;
    while ((1 << l2) < 2 * ncap)
        l2++;
    t->hash.rebuild(l2);
    if ((rc = t->d_hkeys.reserve(sizeof(int64_t) << l2)) != ORBGPU_OK || (rc = t->d_hvals.reserve(sizeof(int32_t) << l2)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpyAsync(t->d_hkeys.p, t->hash.keys.data(), sizeof(int64_t) << l2, hipMemcpyHostToDevice, t->stream));
    ORBGPU_HIP_TRY(hipMemcpyAsync(t->d_hvals.p, t->hash.vals.data(), sizeof(int32_t) << l2, hipMemcpyHostToDevice, t->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(t->stream));
    return ORBGPU_OK;
}

static void frame_dev_view(const orbgpu_frame *fr, orbgpu_device_frame_view *v)
{
    uint8_t *b = fr->block.as<uint8_t>();
    v->cap = fr->cap;
    v->n = reinterpret_cast<const int32_t *>(b + fr->o_n);
    v->kps = reinterpret_cast<const orbgpu_keypoint *>(b + fr->o_kps);
    v->desc = b + fr->o_desc;
    v->u_right = reinterpret_cast<const float *>(b + fr->o_ur);
    v->cell_start = reinterpret_cast<const int32_t *>(b + fr->o_cs);
    v->cell_items = reinterpret_cast<const int32_t *>(b + fr->o_ci);
    v->nlevels = fr->nlevels;
    v->scale_factors = fr->scale_factors;
    v->min_x = fr->min_x, v->max_x = fr->max_x, v->min_y = fr->min_y, v->max_y = fr->max_y;
}

// A search call with an empty list (or an empty frame): kp_to_mp still follows the output contract -- a key point holding
// a point WITH observations is -2, one holding a point without (or nothing) is -1 -- so the table is consulted.
static int translate_kp_only(orbgpu_mappoint_table *t, int n, int cap, const int64_t *kp_ids, int32_t *kp_to_mp)
{
    if (n == 0)
        return ORBGPU_OK;
    if (!kp_ids) {
        for (int j = 0; j < n; j++)
            kp_to_mp[j] = -1;
        return ORBGPU_OK;
    }
    Carver c;
    const size_t o_kp = c.take(8 * (size_t)n), o_k2m = c.take(4 * (size_t)cap), o_cnt = c.take(16);
    int rc;
    if ((rc = t->stage.reserve(c.off)) != ORBGPU_OK || (rc = t->d_stage.reserve(c.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)t->stage.p, *d = t->d_stage.as<uint8_t>();
    std::memcpy(h + o_kp, kp_ids, 8 * (size_t)n);
    hipStream_t st = t->stream;
    ORBGPU_HIP_TRY(hipMemcpyAsync(d + o_kp, h + o_kp, 8 * (size_t)n, hipMemcpyHostToDevice, st));
    ORBGPU_HIP_TRY(hipMemsetAsync(d + o_cnt, 0, 16, st));
    hipLaunchKernelGGL(k_table_kp, dim3((cap + 255) / 256), dim3(256), 0, st, n, cap, 0, (const int64_t *)(d + o_kp),
                       t->d_hkeys.as<int64_t>(), t->d_hvals.as<int32_t>(), t->hash.log2cap, t->obs.as<uint8_t>(),
                       (const int32_t *)nullptr, (int32_t *)(d + o_k2m), (int32_t *)(d + o_cnt) + 3);
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpyAsync(h + o_k2m, d + o_k2m, (o_cnt + 16) - o_k2m, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(kp_to_mp, h + o_k2m, 4 * (size_t)n);
    t->last_unknown_kp = ((const int32_t *)(h + o_cnt))[3];
    return ORBGPU_OK;
}

} // namespace orbgpu

extern "C" {

int orbgpu_mappoint_table_last_unknown(const orbgpu_mappoint_table *t, int32_t *list_ids, int32_t *kp_ids)
{
    ORBGPU_REQUIRE(t, "null table");
    if (list_ids)
        *list_ids = t->last_unknown_list;
    if (kp_ids)
        *kp_ids = t->last_unknown_kp;
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// table
// ---------------------------------------------------------------------------------------------------------------------
int orbgpu_mappoint_table_create(int32_t device_id, int32_t initial_rows, orbgpu_mappoint_table **out)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    ORBGPU_REQUIRE(out && initial_rows >= 0 && initial_rows <= (1 << 26), "bad argument");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_mappoint_table *t = new (std::nothrow) orbgpu_mappoint_table();
    if (!t) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    t->device_id = device_id;
    t->pws = proj_workspace_new();
    if (!t->pws) {
        set_error("out of host memory");
        delete t;
        return ORBGPU_ENOMEM;
    }
    hipError_t he = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(he));
        delete t;
        return ORBGPU_EHIP;
    }
    if ((rc = table_grow(t, std::max(initial_rows, 1024))) != ORBGPU_OK) {
        orbgpu_mappoint_table_destroy(t);
        return rc;
    }
    *out = t;
    return ORBGPU_OK;
}

int orbgpu_mappoint_table_destroy(orbgpu_mappoint_table *t)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!t)
        return ORBGPU_OK;
    (void)hipSetDevice(t->device_id);
    if (t->stream) {
        (void)hipStreamSynchronize(t->stream);
        (void)hipStreamDestroy(t->stream);
    }
    DevBuf *bufs[] = {&t->world_pos, &t->normal, &t->min_dist, &t->max_dist, &t->desc,    &t->bad,    &t->obs,
                      &t->id,        &t->d_hkeys, &t->d_hvals, &t->d_stage,  &t->g_block, &t->pos_of_row};
    for (DevBuf *b : bufs)
        b->release();
    t->stage.release();
    proj_workspace_delete(t->pws);
    delete t;
    return ORBGPU_OK;
}

int orbgpu_mappoint_table_rows(const orbgpu_mappoint_table *t, int32_t *rows)
{
    ORBGPU_REQUIRE(t && rows, "null argument");
    *rows = t->rows;
    return ORBGPU_OK;
}

int orbgpu_mappoint_table_upsert(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, const float *world_pos,
                                 const float *normal, const float *min_dist, const float *max_dist, const uint8_t *desc,
                                 const int32_t *n_obs)
{
    ORBGPU_REQUIRE(t && n >= 0 && (n == 0 || ids), "bad argument");
    if (n == 0)
        return ORBGPU_OK;
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(n <= MT_MAX_CALL, "at most %d map points per upsert call", MT_MAX_CALL);
    int64_t unknown = 0;  // ids that need a row (an id listed twice counts twice: the call is refused below anyway)
    for (int i = 0; i < n; i++) {
        ORBGPU_REQUIRE(ids[i] >= 0, "map point id %lld is negative", (long long)ids[i]);
        unknown += host_find(t, ids[i]) < 0;
    }
    ORBGPU_REQUIRE((int64_t)t->rows + unknown <= MT_MAX_ROWS, "the MapPoint table is limited to %d rows", MT_MAX_ROWS);
    if (unknown && (rc = table_grow(t, t->rows + (int)unknown)) != ORBGPU_OK)  // an update of known points never grows the table
        return rc;
    // staging layout
    Carver cv;
    const size_t o_row = cv.take(4 * (size_t)n), o_new = cv.take((size_t)n), o_id = cv.take(8 * (size_t)n);
    const size_t o_wp = world_pos ? cv.take(12 * (size_t)n) : 0, o_nr = normal ? cv.take(12 * (size_t)n) : 0;
    const size_t o_mn = min_dist ? cv.take(4 * (size_t)n) : 0, o_mx = max_dist ? cv.take(4 * (size_t)n) : 0;
    const size_t o_ds = desc ? cv.take(32 * (size_t)n) : 0, o_ob = n_obs ? cv.take(4 * (size_t)n) : 0;
    const size_t o_hs = cv.take(4 * (size_t)n), o_hk = cv.take(8 * (size_t)n), o_hv = cv.take(4 * (size_t)n);
    if ((rc = t->stage.reserve(cv.off)) != ORBGPU_OK || (rc = t->d_stage.reserve(cv.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)t->stage.p;
    int32_t *h_row = (int32_t *)(h + o_row), *h_hs = (int32_t *)(h + o_hs), *h_hv = (int32_t *)(h + o_hv);
    uint8_t *h_new = h + o_new;
    int64_t *h_hk = (int64_t *)(h + o_hk);
    t->call_no++;
    int nnew = 0;
    const int rows_before = t->rows;
    for (int i = 0; i < n; i++) {
        int r = host_find(t, ids[i]);
        h_new[i] = r < 0;
        if (r < 0) {
            r = t->rows++;
            h_hs[nnew] = (int32_t)host_insert(t, ids[i], r);
            h_hk[nnew] = ids[i];
            h_hv[nnew] = r;
            nnew++;
        }
        if (t->stamp[r] == t->call_no) {  // undo nothing: rows handed out stay valid, but the call is refused
            set_error("map point id %lld appears twice in one upsert call", (long long)ids[i]);
            // the new rows of this call were inserted into the host hash only; drop them again
            t->hash.rollback(h_hs, nnew);
            // (linear probing: removing the most recent insertions in reverse order leaves no broken chains, because
            //  nothing was inserted after them)
            t->rows = rows_before;
            return ORBGPU_EINVAL;
        }
        t->stamp[r] = t->call_no;
        h_row[i] = r;
    }
    std::memcpy(h + o_id, ids, 8 * (size_t)n);
    if (world_pos)
        std::memcpy(h + o_wp, world_pos, 12 * (size_t)n);
    if (normal)
        std::memcpy(h + o_nr, normal, 12 * (size_t)n);
    if (min_dist)
        std::memcpy(h + o_mn, min_dist, 4 * (size_t)n);
    if (max_dist)
        std::memcpy(h + o_mx, max_dist, 4 * (size_t)n);
    if (desc)
        std::memcpy(h + o_ds, desc, 32 * (size_t)n);
    if (n_obs)
        std::memcpy(h + o_ob, n_obs, 4 * (size_t)n);
    // from here on a HIP failure must take the rows this call handed out back again: the device hash never got them
    auto undo = [&](hipError_t he, const char *what) {
        set_error("%s: %s (MapPoint table upsert)", what, hipGetErrorString(he));
        t->hash.rollback(h_hs, nnew);
        t->rows = rows_before;
        return ORBGPU_EHIP;
    };
    hipError_t he = hipMemcpyAsync(t->d_stage.p, h, cv.off, hipMemcpyHostToDevice, t->stream);
    if (he != hipSuccess)
        return undo(he, "hipMemcpyAsync");
    const uint8_t *d = t->d_stage.as<uint8_t>();
    UpsertDev u{(const int32_t *)(d + o_row),
                d + o_new,
                (const int64_t *)(d + o_id),
                world_pos ? (const float *)(d + o_wp) : nullptr,
                normal ? (const float *)(d + o_nr) : nullptr,
                min_dist ? (const float *)(d + o_mn) : nullptr,
                max_dist ? (const float *)(d + o_mx) : nullptr,
                desc ? d + o_ds : nullptr,
                n_obs ? (const int32_t *)(d + o_ob) : nullptr};
    hipLaunchKernelGGL(k_table_scatter, dim3((n + 255) / 256), dim3(256), 0, t->stream, n, u, t->dev());
    if (nnew)
        hipLaunchKernelGGL(k_hash_apply, dim3((nnew + 255) / 256), dim3(256), 0, t->stream, nnew, (const int32_t *)(d + o_hs),
                           (const int64_t *)(d + o_hk), (const int32_t *)(d + o_hv), t->d_hkeys.as<int64_t>(),
                           t->d_hvals.as<int32_t>());
    if ((he = hipGetLastError()) != hipSuccess)
        return undo(he, "kernel launch");
    if ((he = hipStreamSynchronize(t->stream)) != hipSuccess)  // the staging block is reused by the next call
        return undo(he, "hipStreamSynchronize");
    return ORBGPU_OK;
}

static int table_flag(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, const int32_t *n_obs, int32_t *known)
{
    ORBGPU_REQUIRE(t && n >= 0 && (n == 0 || ids), "bad argument");
    if (known)
        *known = 0;
    if (n == 0)
        return ORBGPU_OK;
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    Carver cv;
    const size_t o_row = cv.take(4 * (size_t)n), o_ob = n_obs ? cv.take(4 * (size_t)n) : 0;
    if ((rc = t->stage.reserve(cv.off)) != ORBGPU_OK || (rc = t->d_stage.reserve(cv.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)t->stage.p;
    int32_t *h_row = (int32_t *)(h + o_row);
    int nk = 0;
    for (int i = 0; i < n; i++) {
        h_row[i] = host_find(t, ids[i]);
        nk += h_row[i] >= 0;
    }
    if (n_obs)
        std::memcpy(h + o_ob, n_obs, 4 * (size_t)n);
    ORBGPU_HIP_TRY(hipMemcpyAsync(t->d_stage.p, h, cv.off, hipMemcpyHostToDevice, t->stream));
    const uint8_t *d = t->d_stage.as<uint8_t>();
    hipLaunchKernelGGL(k_table_flag, dim3((n + 255) / 256), dim3(256), 0, t->stream, n, (const int32_t *)(d + o_row),
                       n_obs ? (const int32_t *)(d + o_ob) : nullptr, t->bad.as<uint8_t>(), t->obs.as<uint8_t>());
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipStreamSynchronize(t->stream));
    if (known)
        *known = nk;
    return ORBGPU_OK;
}

int orbgpu_mappoint_table_set_bad(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, int32_t *known)
{
    return table_flag(t, n, ids, nullptr, known);
}

int orbgpu_mappoint_table_set_observations(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, const int32_t *n_obs,
                                           int32_t *known)
{
    ORBGPU_REQUIRE(n == 0 || n_obs, "null observation counts");
    return table_flag(t, n, ids, n_obs, known);
}

int orbgpu_mappoint_table_retain(orbgpu_mappoint_table *t, int32_t n, const int64_t *ids, int32_t *dropped)
{
    ORBGPU_REQUIRE(t && n >= 0 && (n == 0 || ids) && n <= MT_MAX_CALL, "bad argument");
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    // the rows that stay, in the order given (an id listed twice or unknown is passed over)
    std::vector<int32_t> src;
    std::vector<int64_t> kept;
    src.reserve((size_t)n);
    kept.reserve((size_t)n);
    t->call_no++;
    for (int i = 0; i < n; i++) {
        const int r = host_find(t, ids[i]);
        if (r < 0 || t->stamp[(size_t)r] == t->call_no)
            continue;
        t->stamp[(size_t)r] = t->call_no;
        src.push_back(r);
        kept.push_back(ids[i]);
    }
    const int m = (int)src.size(), before = t->rows;
    int ncap = 1024;
    while (ncap < m)
        ncap *= 2;
    ORBGPU_HIP_TRY(hipStreamSynchronize(t->stream));
    // new arrays first; the table is only switched over when everything has succeeded
    DevBuf nb[8], nsrc;
    const size_t elt[8] = {12, 12, 4, 4, 32, 1, 1, 8};
    auto drop = [&] {
        for (DevBuf &b : nb)
            b.release();
        nsrc.release();
    };
    for (int k = 0; k < 8; k++)
        if ((rc = nb[k].reserve(elt[k] * (size_t)ncap)) != ORBGPU_OK) {
            drop();
            return rc;
        }
    if ((rc = nsrc.reserve(sizeof(int32_t) * (size_t)std::max(m, 1))) != ORBGPU_OK) {
        drop();
        return rc;
    }
    hipError_t he = hipSuccess;
    for (int k = 0; k < 8 && he == hipSuccess; k++)
        he = hipMemsetAsync(nb[k].p, 0, elt[k] * (size_t)ncap, t->stream);
    if (he == hipSuccess && m > 0)
        he = hipMemcpyAsync(nsrc.p, src.data(), sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, t->stream);
    if (he == hipSuccess && m > 0) {
        TableDev d{nb[0].as<float>(), nb[1].as<float>(), nb[2].as<float>(), nb[3].as<float>(),
                   nb[4].as<uint8_t>(), nb[5].as<uint8_t>(), nb[6].as<uint8_t>(), nb[7].as<int64_t>()};
        hipLaunchKernelGGL(k_table_compact, dim3((m + 255) / 256), dim3(256), 0, t->stream, m, nsrc.as<int32_t>(), t->dev(), d);
        he = hipGetLastError();
    }
    if (he == hipSuccess)
        he = hipStreamSynchronize(t->stream);
    // the id -> row hash of the kept ids, host and device
    int l2 = 1;
    while ((1 << l2) < 2 * ncap)
        l2++;
    IdHash nh;
    nh.rebuild(l2);
    for (int i = 0; i < m; i++)
        nh.insert(kept[(size_t)i], i);
    DevBuf nk, nv;
    if (he == hipSuccess && ((rc = nk.reserve(sizeof(int64_t) << l2)) != ORBGPU_OK || (rc = nv.reserve(sizeof(int32_t) << l2)) != ORBGPU_OK ||
                             (rc = t->pos_of_row.reserve(sizeof(int32_t) * (size_t)ncap)) != ORBGPU_OK)) {
        nk.release(), nv.release();
        drop();
        return rc;
    }
    if (he == hipSuccess)
        he = hipMemcpyAsync(nk.p, nh.keys.data(), sizeof(int64_t) << l2, hipMemcpyHostToDevice, t->stream);
    if (he == hipSuccess)
        he = hipMemcpyAsync(nv.p, nh.vals.data(), sizeof(int32_t) << l2, hipMemcpyHostToDevice, t->stream);
    if (he == hipSuccess)
        he = hipStreamSynchronize(t->stream);
    if (he != hipSuccess) {
        set_error("MapPoint table retain: %s", hipGetErrorString(he));
        nk.release(), nv.release();
        drop();
        return ORBGPU_EHIP;
    }
    DevBuf *cur[8] = {&t->world_pos, &t->normal, &t->min_dist, &t->max_dist, &t->desc, &t->bad, &t->obs, &t->id};
    for (int k = 0; k < 8; k++) {
        cur[k]->release();
        *cur[k] = nb[k];
    }
    nsrc.release();
    t->d_hkeys.release(), t->d_hvals.release();
    t->d_hkeys = nk, t->d_hvals = nv;
    t->hash = nh;
    t->rows = m;
    t->cap = ncap;
    t->stamp.assign((size_t)ncap, 0);
    t->call_no = 0;
    if (dropped)
        *dropped = before - m;
    return ORBGPU_OK;
}

int orbgpu_mappoint_table_read(orbgpu_mappoint_table *t, int64_t id, float *world_pos, float *normal, float *min_dist,
                               float *max_dist, uint8_t *desc, int32_t *has_observations, int32_t *bad)
{
    ORBGPU_REQUIRE(t, "null table");
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const int r = host_find(t, id);
    ORBGPU_REQUIRE(r >= 0, "map point id %lld is not in the table", (long long)id);
    ORBGPU_HIP_TRY(hipStreamSynchronize(t->stream));
    uint8_t b1 = 0, o1 = 0;
    if (world_pos)
        ORBGPU_HIP_TRY(hipMemcpy(world_pos, t->world_pos.as<float>() + 3 * (size_t)r, 12, hipMemcpyDeviceToHost));
    if (normal)
        ORBGPU_HIP_TRY(hipMemcpy(normal, t->normal.as<float>() + 3 * (size_t)r, 12, hipMemcpyDeviceToHost));
    if (min_dist)
        ORBGPU_HIP_TRY(hipMemcpy(min_dist, t->min_dist.as<float>() + r, 4, hipMemcpyDeviceToHost));
    if (max_dist)
        ORBGPU_HIP_TRY(hipMemcpy(max_dist, t->max_dist.as<float>() + r, 4, hipMemcpyDeviceToHost));
    if (desc)
        ORBGPU_HIP_TRY(hipMemcpy(desc, t->desc.as<uint8_t>() + 32 * (size_t)r, 32, hipMemcpyDeviceToHost));
    ORBGPU_HIP_TRY(hipMemcpy(&b1, t->bad.as<uint8_t>() + r, 1, hipMemcpyDeviceToHost));
    ORBGPU_HIP_TRY(hipMemcpy(&o1, t->obs.as<uint8_t>() + r, 1, hipMemcpyDeviceToHost));
    if (has_observations)
        *has_observations = o1;
    if (bad)
        *bad = b1;
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// device-resident frame
// ---------------------------------------------------------------------------------------------------------------------
int orbgpu_frame_create(int32_t device_id, orbgpu_frame **out)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    ORBGPU_REQUIRE(out, "null argument");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    orbgpu_frame *fr = new (std::nothrow) orbgpu_frame();
    if (!fr) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    fr->device_id = device_id;
    hipError_t he = hipStreamCreateWithFlags(&fr->stream, hipStreamNonBlocking);
    if (he != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(he));
        delete fr;
        return ORBGPU_EHIP;
    }
    *out = fr;
    return ORBGPU_OK;
}

int orbgpu_frame_destroy(orbgpu_frame *fr)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!fr)
        return ORBGPU_OK;
    (void)hipSetDevice(fr->device_id);
    if (fr->stream) {
        (void)hipStreamSynchronize(fr->stream);
        (void)hipStreamDestroy(fr->stream);
    }
    fr->block.release();
    fr->stage.release();
    delete fr;
    return ORBGPU_OK;
}

int orbgpu_frame_upload(orbgpu_frame *fr, const orbgpu_frame_view *f)
{
    ORBGPU_REQUIRE(fr, "null frame");
    int rc = validate_frame(f);
    if (rc != ORBGPU_OK)
        return rc;
    ORBGPU_REQUIRE(f->max_x > f->min_x && f->max_y > f->min_y, "empty image bounds");
    if ((rc = select_device(fr->device_id)) != ORBGPU_OK)
        return rc;
    const int n = f->n, cap = std::max(n, 1);
    const int nc = ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS;
    Carver cv;
    fr->o_n = cv.take(4);
    fr->o_kps = cv.take(sizeof(orbgpu_keypoint) * (size_t)cap);
    fr->o_desc = cv.take(32 * (size_t)cap);
    fr->o_ur = cv.take(4 * (size_t)cap);
    fr->o_cs = cv.take(4 * (size_t)(nc + 1));
    fr->o_ci = cv.take(4 * (size_t)cap);
    if ((rc = fr->stage.reserve(cv.off)) != ORBGPU_OK || (rc = fr->block.reserve(cv.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)fr->stage.p;
    *(int32_t *)(h + fr->o_n) = n;
    pack_keypoints(f, (orbgpu_keypoint *)(h + fr->o_kps));
    std::memcpy(h + fr->o_desc, f->desc, 32 * (size_t)n);
    std::memcpy(h + fr->o_ur, f->u_right, 4 * (size_t)n);
    std::memcpy(h + fr->o_cs, f->cell_start, 4 * (size_t)(nc + 1));
    std::memcpy(h + fr->o_ci, f->cell_items, 4 * (size_t)f->cell_start[nc]);
    ORBGPU_HIP_TRY(hipMemcpyAsync(fr->block.p, h, cv.off, hipMemcpyHostToDevice, fr->stream));
    ORBGPU_HIP_TRY(hipStreamSynchronize(fr->stream));
    fr->n = n;
    fr->cap = cap;
    fr->nlevels = f->nlevels;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++)
        fr->scale_factors[l] = l < f->nlevels ? f->scale_factors[l] : 0.f;
    fr->min_x = f->min_x, fr->max_x = f->max_x, fr->min_y = f->min_y, fr->max_y = f->max_y;
    return ORBGPU_OK;
}

int orbgpu_frame_device_view(const orbgpu_frame *fr, orbgpu_device_frame_view *view, int32_t *n)
{
    ORBGPU_REQUIRE(fr && view, "null argument");
    ORBGPU_REQUIRE(fr->block.p, "the frame has not been uploaded yet");
    frame_dev_view(fr, view);
    if (n)
        *n = fr->n;
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// matchers over the table
// ---------------------------------------------------------------------------------------------------------------------
int orbgpu_search_local_points_table(const orbgpu_frame *fr, orbgpu_mappoint_table *t, int32_t m, const int64_t *ids,
                                     const uint8_t *skip, const orbgpu_mappoint_view *scratch, const float *Tcw, float fx,
                                     float fy, float cx, float cy, float mbf, float log_scale_factor, float cos_limit,
                                     float th, float nnratio, const int64_t *kp_ids, int32_t *kp_to_mp, int32_t *nmatches,
                                     orbgpu_track_scratch *track_out)
{
    ORBGPU_REQUIRE(fr && t && kp_to_mp && nmatches && m >= 0 && (m == 0 || ids), "bad argument");
    ORBGPU_REQUIRE(fr->block.p, "the frame has not been uploaded yet");
    ORBGPU_REQUIRE(fr->device_id == t->device_id, "frame and table live on different devices");
    ORBGPU_REQUIRE(scratch || Tcw, "neither tracking scratch nor a pose given");
    ORBGPU_REQUIRE(m < (1 << 20), "bad map point count");
    if (scratch)
        ORBGPU_REQUIRE(scratch->m == m && (m == 0 || (scratch->in_view && scratch->level && scratch->view_cos &&
                                                       scratch->proj_x && scratch->proj_y && scratch->proj_xr)),
                       "tracking scratch incomplete");
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const int n = fr->n, cap = fr->cap;
    *nmatches = 0;
    t->last_unknown_list = t->last_unknown_kp = 0;
    if (m == 0 || n == 0)  // nothing to search; the frame's own associations are still translated through the table
        return translate_kp_only(t, n, cap, kp_ids, kp_to_mp);
    const size_t M = (size_t)m;
    // inputs: one staged block
    Carver ci;
    const size_t i_ids = ci.take(8 * M), i_skip = skip ? ci.take(M) : 0, i_kp = kp_ids ? ci.take(8 * (size_t)n) : 0;
    size_t i_iv = 0, i_lv = 0, i_vc = 0, i_px = 0, i_py = 0, i_pr = 0;
    if (scratch) {
        i_iv = ci.take(M), i_lv = ci.take(4 * M), i_vc = ci.take(4 * M);
        i_px = ci.take(4 * M), i_py = ci.take(4 * M), i_pr = ci.take(4 * M);
    }
    // device-side working set + results: one block; the result part is contiguous for one D2H
    Carver cg;
    const size_t g_wp = cg.take(12 * M), g_nr = cg.take(12 * M), g_mn = cg.take(4 * M), g_mx = cg.take(4 * M);
    const size_t g_ds = cg.take(32 * M), g_sk = cg.take(M), g_ob = cg.take(M);
    const size_t r_beg = cg.off;
    const size_t r_cnt = cg.take(16), r_k2m = cg.take(4 * (size_t)cap);
    const bool want_track = track_out && !scratch;
    size_t r_iv = 0, r_px = 0, r_py = 0, r_pr = 0, r_vc = 0, r_lv = 0;
    if (want_track) {
        r_iv = cg.take(M), r_px = cg.take(4 * M), r_py = cg.take(4 * M), r_pr = cg.take(4 * M);
        r_vc = cg.take(4 * M), r_lv = cg.take(4 * M);
    }
    const size_t r_end = cg.off;
    const size_t host_bytes = std::max(ci.off, r_end - r_beg);
    if ((rc = t->stage.reserve(host_bytes)) != ORBGPU_OK || (rc = t->d_stage.reserve(ci.off)) != ORBGPU_OK ||
        (rc = t->g_block.reserve(cg.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)t->stage.p;
    std::memcpy(h + i_ids, ids, 8 * M);
    if (skip)
        std::memcpy(h + i_skip, skip, M);
    if (kp_ids)
        std::memcpy(h + i_kp, kp_ids, 8 * (size_t)n);
    if (scratch) {
        std::memcpy(h + i_iv, scratch->in_view, M);
        std::memcpy(h + i_lv, scratch->level, 4 * M);
        std::memcpy(h + i_vc, scratch->view_cos, 4 * M);
        std::memcpy(h + i_px, scratch->proj_x, 4 * M);
        std::memcpy(h + i_py, scratch->proj_y, 4 * M);
        std::memcpy(h + i_pr, scratch->proj_xr, 4 * M);
    }
    hipStream_t st = t->stream;
    ORBGPU_HIP_TRY(hipMemcpyAsync(t->d_stage.p, h, ci.off, hipMemcpyHostToDevice, st));
    ORBGPU_HIP_TRY(hipMemsetAsync(t->pos_of_row.p, 0x7F, sizeof(int32_t) * (size_t)t->rows, st));  // 0x7F7F7F7F: "not in the list"
    uint8_t *d = t->d_stage.as<uint8_t>(), *g = t->g_block.as<uint8_t>();
    int32_t *d_cnt = (int32_t *)(g + r_cnt);
    ORBGPU_HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, st));
    GatherOut go{(float *)(g + g_wp), (float *)(g + g_nr), (float *)(g + g_mn), (float *)(g + g_mx), g + g_ds, g + g_sk,
                 g + g_ob,            nullptr,             t->pos_of_row.as<int32_t>(), d_cnt + 2};
    hipLaunchKernelGGL(k_table_gather, dim3((m + 255) / 256), dim3(256), 0, st, m, (const int64_t *)(d + i_ids),
                       skip ? d + i_skip : nullptr, t->d_hkeys.as<int64_t>(), t->d_hvals.as<int32_t>(), t->hash.log2cap, t->dev(),
                       go);
    if (kp_ids)
        hipLaunchKernelGGL(k_table_kp, dim3((cap + 255) / 256), dim3(256), 0, st, n, cap, m, (const int64_t *)(d + i_kp),
                           t->d_hkeys.as<int64_t>(), t->d_hvals.as<int32_t>(), t->hash.log2cap, t->obs.as<uint8_t>(),
                           t->pos_of_row.as<int32_t>(), (int32_t *)(g + r_k2m), d_cnt + 3);
    else
        ORBGPU_HIP_TRY(hipMemsetAsync(g + r_k2m, 0xFF, 4 * (size_t)cap, st));  // every key point free
    orbgpu_device_frame_view fv;
    frame_dev_view(fr, &fv);
    orbgpu_device_mappoint_table tv{m,          (const float *)(g + g_wp), (const float *)(g + g_nr), (const float *)(g + g_mn),
                                    (const float *)(g + g_mx), g + g_ds,   g + g_sk,                  g + g_ob};
    ScratchDev sd{d + i_iv, (const int32_t *)(d + i_lv), (const float *)(d + i_vc), (const float *)(d + i_px),
                  (const float *)(d + i_py), (const float *)(d + i_pr)};
    orbgpu_track_scratch trk{};
    if (want_track)
        trk = orbgpu_track_scratch{g + r_iv,           (float *)(g + r_px), (float *)(g + r_py), (float *)(g + r_pr),
                                   (float *)(g + r_vc), (int32_t *)(g + r_lv)};
    // counts layout: [0] matches, [1] bad levels, [2] unknown list ids, [3] unknown key-point ids
    int32_t *d_counts2 = d_cnt;  // the matcher zeroes and fills [0], [1]; [2], [3] were written before it runs ...
    // ... so they must survive its memset of the first two words only (it clears exactly 2 ints)
    ProjWorkspaceScope own_workspace(t->pws);
    if ((rc = search_local_points_device_impl(&fv, &tv, scratch ? &sd : nullptr, Tcw, fx, fy, cx, cy, mbf, log_scale_factor,
                                              cos_limit, th, nnratio, (int32_t *)(g + r_k2m), d_counts2,
                                              want_track ? &trk : nullptr, t->device_id, st)) != ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpyAsync(h, g + r_beg, r_end - r_beg, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    const int32_t *cnt = (const int32_t *)(h + (r_cnt - r_beg));
    // ids the table has not been told about yet are not an error: LocalMapping publishes a new point to the key frames
    // (LocalMapping.cc:434-440) before its attributes are final, and Tracking::UpdateLocalPoints may list it in that
    // window.  Such a list row is skipped, such a key-point association counts as held; the counts are reported by
    // orbgpu_mappoint_table_last_unknown.
    t->last_unknown_list = cnt[2], t->last_unknown_kp = cnt[3];
    if (scratch && cnt[1]) {
        set_error("%d map point(s) with a predicted level outside [0,%d)", cnt[1], fr->nlevels);
        return ORBGPU_ELEVEL;  // as orbgpu_search_by_projection (H5)
    }
    std::memcpy(kp_to_mp, h + (r_k2m - r_beg), 4 * (size_t)n);
    *nmatches = cnt[0];
    if (track_out && want_track) {
        if (track_out->in_view)
            std::memcpy(track_out->in_view, h + (r_iv - r_beg), M);
        if (track_out->proj_x)
            std::memcpy(track_out->proj_x, h + (r_px - r_beg), 4 * M);
        if (track_out->proj_y)
            std::memcpy(track_out->proj_y, h + (r_py - r_beg), 4 * M);
        if (track_out->proj_xr)
            std::memcpy(track_out->proj_xr, h + (r_pr - r_beg), 4 * M);
        if (track_out->view_cos)
            std::memcpy(track_out->view_cos, h + (r_vc - r_beg), 4 * M);
        if (track_out->level)
            std::memcpy(track_out->level, h + (r_lv - r_beg), 4 * M);
    }
    return ORBGPU_OK;
}

int orbgpu_search_by_projection_last_table(const orbgpu_frame *cur, const float *cur_Tcw, const orbgpu_frame *last,
                                           const float *last_Tcw, orbgpu_mappoint_table *t, const int64_t *last_ids,
                                           const uint8_t *last_outlier, const int64_t *cur_kp_ids, float fx, float fy,
                                           float cx, float cy, float mbf, float mb, float th, int32_t mono,
                                           int32_t check_orientation, int32_t *kp_to_mp, int32_t *nmatches)
{
    ORBGPU_REQUIRE(cur && last && t && cur_Tcw && last_Tcw && kp_to_mp && nmatches, "null argument");
    ORBGPU_REQUIRE(cur->block.p && last->block.p, "a frame has not been uploaded yet");
    ORBGPU_REQUIRE(cur->device_id == t->device_id && last->device_id == t->device_id, "frames and table live on different devices");
    ORBGPU_REQUIRE(last->n == 0 || last_ids, "null last-frame ids");
    int rc = select_device(t->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    const int n = cur->n, cap = cur->cap, nl = last->n;
    *nmatches = 0;
    t->last_unknown_list = t->last_unknown_kp = 0;
    if (nl == 0 || n == 0)
        return translate_kp_only(t, n, cap, cur_kp_ids, kp_to_mp);
    const size_t M = (size_t)nl;
    Carver ci;
    const size_t i_ids = ci.take(8 * M), i_out = last_outlier ? ci.take(M) : 0, i_kp = cur_kp_ids ? ci.take(8 * (size_t)n) : 0;
    Carver cg;
    const size_t g_wp = cg.take(12 * M), g_ds = cg.take(32 * M), g_sk = cg.take(M), g_ob = cg.take(M), g_has = cg.take(M);
    const size_t r_beg = cg.off;
    const size_t r_cnt = cg.take(16), r_k2m = cg.take(4 * (size_t)cap);
    const size_t r_end = cg.off;
    if ((rc = t->stage.reserve(std::max(ci.off, r_end - r_beg))) != ORBGPU_OK || (rc = t->d_stage.reserve(ci.off)) != ORBGPU_OK ||
        (rc = t->g_block.reserve(cg.off)) != ORBGPU_OK)
        return rc;
    uint8_t *h = (uint8_t *)t->stage.p;
    std::memcpy(h + i_ids, last_ids, 8 * M);
    if (last_outlier)
        std::memcpy(h + i_out, last_outlier, M);
    if (cur_kp_ids)
        std::memcpy(h + i_kp, cur_kp_ids, 8 * (size_t)n);
    hipStream_t st = t->stream;
    ORBGPU_HIP_TRY(hipMemcpyAsync(t->d_stage.p, h, ci.off, hipMemcpyHostToDevice, st));
    uint8_t *d = t->d_stage.as<uint8_t>(), *g = t->g_block.as<uint8_t>();
    int32_t *d_cnt = (int32_t *)(g + r_cnt);
    ORBGPU_HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, st));
    // rows = key points of the last frame; a key point without a map point carries id -1 (has = 0).  The matcher skips
    // outliers itself; bad points stay in (the reference does not test isBad() here, ORBmatcher.cc:1351-1357).
    GatherOut go{(float *)(g + g_wp), nullptr, nullptr, nullptr, g + g_ds, g + g_sk, g + g_ob, g + g_has, nullptr, d_cnt + 2};
    hipLaunchKernelGGL(k_table_gather, dim3((nl + 255) / 256), dim3(256), 0, st, nl, (const int64_t *)(d + i_ids),
                       (const uint8_t *)nullptr, t->d_hkeys.as<int64_t>(), t->d_hvals.as<int32_t>(), t->hash.log2cap, t->dev(), go);
    if (cur_kp_ids)  // the current frame's own associations never point into the last frame's rows: held or free
        hipLaunchKernelGGL(k_table_kp, dim3((cap + 255) / 256), dim3(256), 0, st, n, cap, 0, (const int64_t *)(d + i_kp),
                           t->d_hkeys.as<int64_t>(), t->d_hvals.as<int32_t>(), t->hash.log2cap, t->obs.as<uint8_t>(),
                           (const int32_t *)nullptr, (int32_t *)(g + r_k2m), d_cnt + 3);
    else
        ORBGPU_HIP_TRY(hipMemsetAsync(g + r_k2m, 0xFF, 4 * (size_t)cap, st));
    orbgpu_device_frame_view cv_, lv_;
    frame_dev_view(cur, &cv_);
    frame_dev_view(last, &lv_);
    orbgpu_device_lastframe_view lv{last->cap, lv_.n, lv_.kps, g + g_has, last_outlier ? d + i_out : nullptr, g + g_ob,
                                    (const float *)(g + g_wp), g + g_ds};
    ProjWorkspaceScope own_workspace(t->pws);
    if ((rc = orbgpu_search_by_projection_last_device(&cv_, cur_Tcw, &lv, last_Tcw, fx, fy, cx, cy, mbf, mb, th, mono,
                                                      check_orientation, (int32_t *)(g + r_k2m), d_cnt, t->device_id, st)) !=
        ORBGPU_OK)
        return rc;
    ORBGPU_HIP_TRY(hipMemcpyAsync(h, g + r_beg, r_end - r_beg, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));
    const int32_t *cnt = (const int32_t *)(h + (r_cnt - r_beg));
    t->last_unknown_list = cnt[2], t->last_unknown_kp = cnt[3];  // skipped rows / held key points, not an error (see above)
    std::memcpy(kp_to_mp, h + (r_k2m - r_beg), 4 * (size_t)n);
    *nmatches = cnt[0];
    return ORBGPU_OK;
}

} // extern "C"
