// setorder.h -- iteration order of libstdc++'s std::unordered_set<unsigned> (identity hash) after a sequence of
// distinct insertions into a default-constructed set.  metaserver keeps its reader sets in that container
// (metaserver.cpp:23) and prints / sums in its iteration order, so the order is part of bit-exact parity.
//
// Model (libstdc++ _Hashtable with _Prime_rehash_policy, max load factor 1):
//   * one singly linked list of all nodes; the nodes of a bucket are contiguous in it;
//   * inserting key k: if bucket k % B is non-empty the node goes to the FRONT of that bucket's run, otherwise to
//     the front of the whole list;
//   * before the insertion that makes size exceed the current threshold the table is rehashed to the next size of
//     1 -> 13 -> 29 -> 59 -> 127 -> 257 -> 541 (first insertion allocates 13: "initial bucket size of 11" rounded up by
//     _M_next_bkt); rehashing re-inserts every node, in list order, into an empty table by the same rule.
// tests/test_setorder.py checks this model against the real container for random sequences up to 273 keys.
#pragma once
#include <cstdint>

#ifdef __HIPCC__
#define DSM_HD __host__ __device__
#else
#define DSM_HD
#endif

namespace dsm {

DSM_HD inline uint32_t so_place(uint16_t* L, uint32_t s, uint32_t B, uint16_t k) {
    const uint32_t b = k % B;
    uint32_t p = 0;
    for (uint32_t i = 0; i < s; ++i)
        if (L[i] % B == b) { p = i; break; }
    for (uint32_t i = s; i > p; --i) L[i] = L[i - 1];
    L[p] = k;
    return s + 1;
}

DSM_HD inline uint32_t so_next_buckets(uint32_t size_after) {  // bucket count in force once `size_after` keys are in
    return size_after <= 13 ? 13u : size_after <= 29 ? 29u : size_after <= 59 ? 59u : size_after <= 127 ? 127u : size_after <= 257 ? 257u : 541u;
}

// out[0..m) = iteration order after inserting seq[0..m) (distinct keys < 541); tmp needs m entries.
DSM_HD inline void set_iteration_order(const uint16_t* seq, uint32_t m, uint16_t* out, uint16_t* tmp) {
    uint32_t s = 0, B = 1;
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t nb = so_next_buckets(t + 1);
        if (nb != B) {  // rehash before the insertion that crosses the threshold
            for (uint32_t i = 0; i < s; ++i) tmp[i] = out[i];
            uint32_t s2 = 0;
            for (uint32_t i = 0; i < s; ++i) s2 = so_place(out, s2, nb, tmp[i]);
            B = nb;
        }
        s = so_place(out, s, B, seq[t]);
    }
}

}  // namespace dsm
