// CPU unit test of the MapPoint table's id -> row hash (orb_slam2_map_amd/csrc/id_hash.h) and of the shim's pointer index
// (orbgpu_shim::PtrIndex): plain g++ with the sanitizers, no HIP, nothing linked.
#include "id_hash.h"
#include "orbgpu_shim.hpp"

#include <algorithm>
#include <cstdio>
#include <map>
#include <random>

#define CHECK(c)                                                  \
    do {                                                          \
        if (!(c)) {                                               \
            std::printf("FAILED %s (line %d)\n", #c, __LINE__);   \
            return 1;                                             \
        }                                                         \
    } while (0)

static int test_id_hash()
{
    std::mt19937_64 rng(7);
    orbgpu::IdHash h;
    std::map<int64_t, int32_t> ref;
    int rows = 0, cap_rows = 0;
    auto ensure = [&](int want) {  // the owner's growth rule: capacity >= 2 * rows
        if (want <= cap_rows)
            return;
        cap_rows = std::max(cap_rows * 2, 16);
        while (cap_rows < want)
            cap_rows *= 2;
        int l2 = 1;
        while ((1 << l2) < 2 * cap_rows)
            l2++;
        h.rebuild(l2);
    };
    CHECK(h.find(5) == -1);  // empty table
    for (int round = 0; round < 200; round++) {
        const int n = 1 + (int)(rng() % 700);
        ensure(rows + n);
        std::vector<int32_t> slots;
        const int rows_before = rows;
        bool dup = false;
        std::vector<int64_t> batch;
        for (int i = 0; i < n; i++) {
            // ids: small dense numbers, multiples of large strides, 40-bit values -- and now and then one of the batch again
            int64_t id = (rng() % 3 == 0) ? (int64_t)(rng() % 5000) : (int64_t)((rng() % 100000) * ((rng() % 2) ? 7919ll : (1ll << 33)) + 3);
            if (!batch.empty() && rng() % 400 == 0)
                id = batch[rng() % batch.size()];
            if (std::find(batch.begin(), batch.end(), id) != batch.end()) {
                dup = true;  // the table refuses the whole call: undo what it inserted
                break;
            }
            batch.push_back(id);
            if (h.find(id) < 0)
                slots.push_back((int32_t)h.insert(id, rows++));
        }
        if (dup) {
            h.rollback(slots.data(), (int)slots.size());
            rows = rows_before;
        } else {
            int r = rows_before;
            for (int64_t id : batch)
                if (!ref.count(id))
                    ref[id] = r++;
            CHECK(r == rows);
        }
        // every id ever accepted is found with its row, rolled-back and never-seen ids are absent
        if (round % 10 == 0 || dup) {
            for (const auto &kv : ref)
                CHECK(h.find(kv.first) == kv.second);
            if (dup)
                for (int64_t id : batch)
                    CHECK(h.find(id) == (ref.count(id) ? ref[id] : -1));
            for (int k = 0; k < 200; k++) {
                const int64_t id = (int64_t)(rng() >> 12);
                CHECK(h.find(id) == (ref.count(id) ? ref[id] : -1));
            }
        }
        // load factor invariant
        size_t used = 0;
        for (int64_t k : h.keys)
            used += k != orbgpu::ID_HASH_EMPTY;
        CHECK(used == ref.size() && 2 * used <= h.capacity());
    }
    CHECK(h.find(-1) == -1 && h.find(-12345) == -1);
    std::printf("id_hash ok: %zu ids, capacity %zu\n", ref.size(), h.capacity());
    return 0;
}

struct Obj {
    int v;
};

static int test_ptr_index()
{
    std::mt19937 rng(11);
    std::vector<Obj> pool(5000);
    for (int round = 0; round < 300; round++) {
        const int m = (int)(rng() % 3000), n = (int)(rng() % 400);
        std::vector<Obj *> list(m), held(n);
        for (auto &p : list)
            p = &pool[rng() % pool.size()];  // duplicates in the list: the FIRST position counts
        for (auto &p : held)
            p = (rng() % 4 == 0) ? nullptr : &pool[rng() % pool.size()];
        orbgpu_shim::PtrIndex<Obj> idx(held.begin(), held.end());
        idx.locate(list.data(), m);
        for (Obj *p : held) {
            int want = -1;
            for (int i = 0; i < m && want < 0 && p; i++)
                if (list[i] == p)
                    want = i;
            CHECK(idx.position(p) == want);
        }
        CHECK(idx.position(nullptr) == -1);
        CHECK(idx.position(&pool[0] + pool.size()) == -1);  // a pointer that was never a key
    }
    std::printf("ptr_index ok\n");
    return 0;
}

int main() { return test_id_hash() || test_ptr_index(); }
