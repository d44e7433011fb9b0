// Vocabulary-tree transform (DBoW2) for MI355X (gfx950).
//
// Replaces, for Frame::ComputeBoW (reference src/Frame.cc:395-402, levelsup = 4) and KeyFrame::ComputeBoW:
//   TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
//                                  Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1207
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   :1231-1274
//   FORB::distance                 Thirdparty/DBoW2/DBoW2/FORB.cpp:81-102  (256-bit Hamming)
//
// The descent is the compute: per feature L levels x k Hamming distances against the children of the current node
// (k = 10, L = 6 for ORBvoc: 60 distances, 1.9 KB of node descriptors per feature).  Sixteen lanes own a feature:
// lane c takes child c (children beyond 16 in further rounds), a 4-step xor-shuffle picks the first minimum
// (key = distance << 16 | child position, the reference's strict `<` keeps the earliest child), six dependent
// rounds.  The tree is laid out children-contiguous so a round reads k consecutive 32-byte descriptors.
// The std::map-shaped results (BowVector, FeatureVector) are assembled on the host from the per-feature
// (word, weight, node) triples in the reference's double-precision operation order: they are host containers in
// the reference too, and the node-wise matcher on the device only needs the per-feature node ids.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

namespace orbgpu {

struct VocDev {
    const int *child_first;    // [n_nodes] first child slot
    const int *child_count;    // [n_nodes] 0 = leaf (Node::isLeaf() is children.empty())
    const int *child_node;     // [n_nodes - 1] node id of a child slot
    const uint8_t *child_desc; // [n_nodes - 1][32] descriptor of a child slot
    const int *word_id;        // [n_nodes] -1 for inner nodes
    const double *weight;      // [n_nodes]
    int L;
};

constexpr int BOW_LANES = 16;

// grid = (ceil(cap / 16), batch); desc [batch][cap][32]; n_dev [batch] or nullptr (then n_host rows per frame)
__global__ __launch_bounds__(256) void k_bow_transform(VocDev V, const uint8_t *__restrict__ desc, int cap,
                                                       const int *__restrict__ n_dev, int n_host, int levelsup,
                                                       int *__restrict__ word_out, double *__restrict__ weight_out,
                                                       int *__restrict__ node_out)
{
    const int frame = blockIdx.y;
    const int n = n_dev ? min(max(n_dev[frame], 0), cap) : n_host;
    const int sub = threadIdx.x & (BOW_LANES - 1);
    const int f = blockIdx.x * (256 / BOW_LANES) + (threadIdx.x / BOW_LANES);
    const bool live = f < n;
    const size_t row = (size_t)frame * cap + (live ? f : 0);
    const uint64_t *pd = reinterpret_cast<const uint64_t *>(desc + row * 32);
    const uint64_t a[4] = {pd[0], pd[1], pd[2], pd[3]};
    const int nid_level = V.L - levelsup;
    int node = 0, level = 0, nid = 0;
    bool got = nid_level <= 0;  // root
    // all 16 lanes of a feature walk together (node is uniform within the group); a dead group walks feature 0
    while (true) {
        const int cnt = V.child_count[node];
        if (cnt == 0)
            break;
        ++level;
        const int first = V.child_first[node];
        uint32_t best = 0xFFFFFFFFu;
        for (int c = sub; c < cnt; c += BOW_LANES) {
            const uint64_t *pc = reinterpret_cast<const uint64_t *>(V.child_desc + (size_t)(first + c) * 32);
            const uint64_t b[4] = {pc[0], pc[1], pc[2], pc[3]};
            best = min(best, ((uint32_t)hamming256(a, b) << 16) | (uint32_t)c);
        }
#pragma unroll
        for (int off = BOW_LANES / 2; off > 0; off >>= 1)
            best = min(best, (uint32_t)__shfl_xor((int)best, off, 64));
        node = V.child_node[first + (int)(best & 0xFFFFu)];
        if (level == nid_level) {
            nid = node;
            got = true;
        }
    }
    if (!got)
        nid = node;  // leaf above the requested level: the reference leaves the caller's variable unset
    if (live && sub == 0) {
        word_out[row] = V.word_id[node];
        weight_out[row] = V.weight[node];
        node_out[row] = V.weight[node] > 0 ? nid : -1;  // stopped words (:1170) are not in the FeatureVector
    }
}

} // namespace orbgpu

using namespace orbgpu;

struct orbgpu_vocabulary {
    int device_id = 0, k = 0, L = 0, n_nodes = 0, n_words = 0, weighting = 0, scoring = 0;
    DevBuf child_first, child_count, child_node, child_desc, word_id, weight;
    // host-entry staging
    DevBuf d_desc, d_word, d_weight, d_node;
    hipStream_t stream = nullptr;
    VocDev dev() const
    {
        VocDev v;
        v.child_first = child_first.as<int>();
        v.child_count = child_count.as<int>();
        v.child_node = child_node.as<int>();
        v.child_desc = child_desc.as<uint8_t>();
        v.word_id = word_id.as<int>();
        v.weight = weight.as<double>();
        v.L = L;
        return v;
    }
};

extern "C" {

int orbgpu_vocabulary_create(int32_t k, int32_t L, int32_t n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                             const uint8_t *desc, const double *weight, int32_t weighting, int32_t scoring,
                             int32_t device_id, orbgpu_vocabulary **out)
{
    ORBGPU_REQUIRE(out && parent && is_leaf && desc && weight, "null argument");
    // the limits of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1370)
    ORBGPU_REQUIRE(k >= 1 && k <= 20 && L >= 1 && L <= 10, "k must be in [1,20], L in [1,10]");
    ORBGPU_REQUIRE(weighting >= 0 && weighting <= 3 && scoring >= 0 && scoring <= 5, "bad weighting / scoring type");
    ORBGPU_REQUIRE(n_nodes >= 2 && n_nodes < (1 << 28), "bad node count");
    int rc = select_device(device_id);
    if (rc != ORBGPU_OK)
        return rc;
    std::vector<int> cnt((size_t)n_nodes, 0), first((size_t)n_nodes, 0), wid((size_t)n_nodes, -1);
    for (int i = 1; i < n_nodes; i++) {
        ORBGPU_REQUIRE(parent[i] >= 0 && parent[i] < i, "node %d: parent %d must precede it (loader order)", i, parent[i]);
        cnt[parent[i]]++;
    }
    int acc = 0, nw = 0;
    for (int i = 0; i < n_nodes; i++) {
        ORBGPU_REQUIRE(cnt[i] < 65536, "node %d has too many children", i);
        first[i] = acc;
        acc += cnt[i];
    }
    for (int i = 1; i < n_nodes; i++) {
        ORBGPU_REQUIRE((cnt[i] == 0) == (is_leaf[i] != 0), "node %d: is_leaf disagrees with the tree", i);
        if (is_leaf[i])
            wid[i] = nw++;  // leaves numbered in node order (:1421-1426)
    }
    ORBGPU_REQUIRE(cnt[0] > 0, "the root has no children");
    std::vector<int> pos(first), cnode((size_t)n_nodes - 1);
    std::vector<uint8_t> cdesc(((size_t)n_nodes - 1) * 32);
    for (int i = 1; i < n_nodes; i++) {  // ascending id = the loader's children.push_back order
        const int slot = pos[parent[i]]++;
        cnode[slot] = i;
        memcpy(&cdesc[(size_t)slot * 32], desc + (size_t)i * 32, 32);
    }
    orbgpu_vocabulary *v = new (std::nothrow) orbgpu_vocabulary();
    if (!v) {
        set_error("out of host memory");
        return ORBGPU_ENOMEM;
    }
    v->device_id = device_id;
    v->k = k, v->L = L, v->n_nodes = n_nodes, v->n_words = nw, v->weighting = weighting, v->scoring = scoring;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        int r = b.reserve(bytes);
        if (r != ORBGPU_OK)
            return r;
        ORBGPU_HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        ORBGPU_HIP_TRY(hipStreamSynchronize(nullptr));  // (the kernels that read this run on non-blocking streams)
        return ORBGPU_OK;
    };
    if ((rc = up(v->child_first, first.data(), sizeof(int) * (size_t)n_nodes)) != ORBGPU_OK ||
        (rc = up(v->child_count, cnt.data(), sizeof(int) * (size_t)n_nodes)) != ORBGPU_OK ||
        (rc = up(v->child_node, cnode.data(), sizeof(int) * ((size_t)n_nodes - 1))) != ORBGPU_OK ||
        (rc = up(v->child_desc, cdesc.data(), cdesc.size())) != ORBGPU_OK ||
        (rc = up(v->word_id, wid.data(), sizeof(int) * (size_t)n_nodes)) != ORBGPU_OK ||
        (rc = up(v->weight, weight, sizeof(double) * (size_t)n_nodes)) != ORBGPU_OK) {
        orbgpu_vocabulary_destroy(v);
        return rc;
    }
    hipError_t e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(e));
        orbgpu_vocabulary_destroy(v);
        return ORBGPU_EHIP;
    }
    *out = v;
    return ORBGPU_OK;
}

int orbgpu_vocabulary_destroy(orbgpu_vocabulary *v)
{
    std::lock_guard<std::mutex> lifecycle(orbgpu::lifecycle_mutex());
    if (!v)
        return ORBGPU_OK;
    (void)hipSetDevice(v->device_id);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&v->child_first, &v->child_count, &v->child_node, &v->child_desc, &v->word_id,
                      &v->weight,      &v->d_desc,      &v->d_word,     &v->d_weight,   &v->d_node};
    for (DevBuf *b : bufs)
        b->release();
    if (v->stream)
        (void)hipStreamDestroy(v->stream);
    delete v;
    return ORBGPU_OK;
}

int orbgpu_vocabulary_size(const orbgpu_vocabulary *v, int32_t *n_words)
{
    ORBGPU_REQUIRE(v && n_words, "null argument");
    *n_words = v->n_words;
    return ORBGPU_OK;
}

int orbgpu_bow_transform_batch_device(orbgpu_vocabulary *v, const uint8_t *d_desc, int32_t batch, int32_t cap,
                                      const int32_t *d_n, int32_t levelsup, int32_t *d_word_id, double *d_word_weight,
                                      int32_t *d_node_id, void *hip_stream)
{
    ORBGPU_REQUIRE(v && d_desc && d_n && d_word_id && d_word_weight && d_node_id, "null argument");
    ORBGPU_REQUIRE(batch >= 1 && cap >= 1 && levelsup >= 0, "bad batch / cap / levelsup");
    int rc = select_device(v->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    hipLaunchKernelGGL(k_bow_transform, dim3((cap + 15) / 16, batch), dim3(256), 0, (hipStream_t)hip_stream, v->dev(),
                       d_desc, cap, d_n, 0, levelsup, d_word_id, d_word_weight, d_node_id);
    ORBGPU_HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
}

int orbgpu_bow_transform(orbgpu_vocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup, int32_t *word_id,
                         double *word_weight, int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                         int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_items, int32_t *n_fv)
{
    ORBGPU_REQUIRE(v && n >= 0 && levelsup >= 0 && (n == 0 || (desc && word_id && word_weight && node_id)),
                   "bad arguments");
    int rc = select_device(v->device_id);
    if (rc != ORBGPU_OK)
        return rc;
    if (n_bow)
        *n_bow = 0;
    if (n_fv)
        *n_fv = 0;
    if (fv_start)
        fv_start[0] = 0;
    if (n == 0)
        return ORBGPU_OK;
    if ((rc = v->d_desc.reserve((size_t)n * 32)) != ORBGPU_OK || (rc = v->d_word.reserve((size_t)n * 4)) != ORBGPU_OK ||
        (rc = v->d_weight.reserve((size_t)n * 8)) != ORBGPU_OK || (rc = v->d_node.reserve((size_t)n * 4)) != ORBGPU_OK)
        return rc;
    hipStream_t st = v->stream;
    ORBGPU_HIP_TRY(hipMemcpyAsync(v->d_desc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bow_transform, dim3((n + 15) / 16, 1), dim3(256), 0, st, v->dev(), v->d_desc.as<uint8_t>(), n,
                       (const int *)nullptr, n, levelsup, v->d_word.as<int>(), v->d_weight.as<double>(),
                       v->d_node.as<int>());
    ORBGPU_HIP_TRY(hipGetLastError());
    ORBGPU_HIP_TRY(hipMemcpyAsync(word_id, v->d_word.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(word_weight, v->d_weight.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipMemcpyAsync(node_id, v->d_node.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_TRY(hipStreamSynchronize(st));

    // ---- the std::map containers of the reference, as sorted arrays (host bookkeeping over n triples) ----
    std::vector<std::pair<int, int>> order;  // (key, feature): std::map order, features in insertion order
    order.reserve((size_t)n);
    if (bow_ids && bow_vals && n_bow) {
        for (int i = 0; i < n; i++)
            if (word_weight[i] > 0)  // "not stopped" (:1170 / :1198)
                order.emplace_back(word_id[i], i);
        std::sort(order.begin(), order.end());
        const bool tf = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
        int nb = 0;
        for (size_t p = 0; p < order.size();) {
            size_t q = p;
            double val = 0;
            for (; q < order.size() && order[q].first == order[p].first; q++) {
                const double w = word_weight[order[q].second];
                if (q == p)
                    val = w;
                else if (tf)
                    val += w;
            }
            bow_ids[nb] = order[p].first;
            bow_vals[nb] = val;
            nb++;
            p = q;
        }
        const bool must = v->scoring != 5;  // ScoringObject.h:74-89: all but DOT_PRODUCT normalise
        const bool l2 = v->scoring == 1;
        if (tf && nb > 0 && !must) {
            const double nd = (double)nb;
            for (int i = 0; i < nb; i++)
                bow_vals[i] /= nd;
        }
        if (must) {  // BowVector::normalize (BowVector.cpp:62-88): sequential double sums in ascending word order
            double norm = 0.0;
            if (!l2) {
                for (int i = 0; i < nb; i++)
                    norm += fabs(bow_vals[i]);
            } else {
                for (int i = 0; i < nb; i++)
                    norm += bow_vals[i] * bow_vals[i];
                norm = sqrt(norm);
            }
            if (norm > 0.0)
                for (int i = 0; i < nb; i++)
                    bow_vals[i] /= norm;
        }
        *n_bow = nb;
    }
    if (fv_nodes && fv_start && fv_items && n_fv) {
        order.clear();
        for (int i = 0; i < n; i++)
            if (word_weight[i] > 0)
                order.emplace_back(node_id[i], i);
        std::sort(order.begin(), order.end());
        int nf = 0;
        for (size_t p = 0; p < order.size(); p++) {
            if (p == 0 || order[p - 1].first != order[p].first) {
                fv_nodes[nf] = order[p].first;
                fv_start[nf] = (int)p;
                nf++;
            }
            fv_items[p] = order[p].second;
        }
        fv_start[nf] = (int)order.size();
        *n_fv = nf;
    }
    return ORBGPU_OK;
}

} // extern "C"
