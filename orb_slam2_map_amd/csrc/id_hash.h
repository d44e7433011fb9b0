// id -> row hash of the MapPoint table (map_table.hip), host side.  Pure C++ (tests/id_hash_test.cpp builds it with g++ and
// the sanitizers); the device kernels probe a byte-identical copy of `keys` / `vals` with the same hash function.
//
// Open addressing, linear probing, capacity 2^log2cap with a load factor <= 1/2 (the owner grows it before it fills up),
// multiplicative hash.  Entries are never removed -- except the ones a refused call has just inserted, which are taken
// out again in one go (`rollback`): nothing was inserted after them, so clearing their slots cannot cut a probe chain
// that an older entry depends on.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace orbgpu {

constexpr int64_t ID_HASH_EMPTY = -1;

#if defined(__HIPCC__)
#define ORBGPU_HD __host__ __device__
#else
#define ORBGPU_HD
#endif
ORBGPU_HD inline uint32_t id_hash_slot(int64_t id, int log2cap)
{
    return (uint32_t)(((uint64_t)id * 0x9E3779B97F4A7C15ull) >> (64 - log2cap));
}

struct IdHash {
    std::vector<int64_t> keys;  // ID_HASH_EMPTY = free slot
    std::vector<int32_t> vals;
    int log2cap = 0;

    size_t capacity() const { return keys.size(); }

    // row of `id`, -1 if absent (ids are >= 0)
    int find(int64_t id) const
    {
        if (log2cap == 0 || id < 0)
            return -1;
        const uint32_t mask = (1u << log2cap) - 1u;
        for (uint32_t s = id_hash_slot(id, log2cap);; s = (s + 1) & mask) {
            if (keys[s] == id)
                return vals[s];
            if (keys[s] == ID_HASH_EMPTY)
                return -1;
        }
    }
    // inserts an id that is NOT in the table (the caller has looked it up) and returns its slot; the table must have a
    // free slot (load factor)
    uint32_t insert(int64_t id, int32_t row)
    {
        const uint32_t mask = (1u << log2cap) - 1u;
        uint32_t s = id_hash_slot(id, log2cap);
        while (keys[s] != ID_HASH_EMPTY)
            s = (s + 1) & mask;
        keys[s] = id;
        vals[s] = row;
        return s;
    }
    // undoes the n most recent insertions (their slots, in any order)
    void rollback(const int32_t *slots, int n)
    {
        for (int k = 0; k < n; k++) {
            keys[(size_t)slots[k]] = ID_HASH_EMPTY;
            vals[(size_t)slots[k]] = -1;
        }
    }
    // capacity 2^l2 (>= twice the rows it must hold), existing entries re-inserted
    void rebuild(int l2)
    {
        std::vector<int64_t> old_keys;
        std::vector<int32_t> old_vals;
        old_keys.swap(keys);
        old_vals.swap(vals);
        log2cap = l2;
        keys.assign((size_t)1 << l2, ID_HASH_EMPTY);
        vals.assign((size_t)1 << l2, -1);
        for (size_t s = 0; s < old_keys.size(); s++)
            if (old_keys[s] != ID_HASH_EMPTY)
                insert(old_keys[s], old_vals[s]);
    }
};

} // namespace orbgpu
