// setorder.h -- iteration order of libstdc++'s std::unordered_set<unsigned> (identity hash) after a sequence of
// distinct insertions into a default-constructed set.  metaserver keeps its reader sets in that container
// (metaserver.cpp:23) and prints / sums in its iteration order, so the order is part of bit-exact parity.
//
// Model (libstdc++ _Hashtable with _Prime_rehash_policy, max load factor 1), replayed with the container's own data structure:
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

DSM_HD inline uint32_t so_next_buckets(uint32_t size_after) {  // bucket count in force once `size_after` keys are in
    return size_after <= 13 ? 13u : size_after <= 29 ? 29u : size_after <= 59 ? 59u : size_after <= 127 ? 127u : size_after <= 257 ? 257u : 541u;
}

// The container as it is: a singly linked list threaded through nxt[key] plus, per bucket, the node BEFORE the bucket's
// first node (_Hashtable::_M_buckets; SO_HEAD stands for _M_before_begin).  O(1) per insertion, O(size + buckets) per rehash.
// K = key / link type: uint16_t in general, uint8_t when every key is below 253 (less scratch per thread on the device).
template <typename K> struct SoMark {
    static constexpr K HEAD = (K)~(K)0, NIL = (K)(HEAD - 1), EMPTY = (K)(HEAD - 2);
};

// work space of set_iteration_order: nxt[max key + 1] followed by before[so_next_buckets(m)]
DSM_HD inline uint32_t so_work_size(uint32_t key_limit, uint32_t m) { return key_limit + so_next_buckets(m); }

// _M_insert_bucket_begin (hashtable.h): front of the bucket's run, or front of the whole list for an empty bucket
template <typename K>
DSM_HD inline void so_insert(K* nxt, K* before, uint32_t B, K& first, K k) {
    const uint32_t b = k % B;
    const K prev = before[b];
    if (prev != SoMark<K>::EMPTY) {
        K& slot = prev == SoMark<K>::HEAD ? first : nxt[prev];
        nxt[k] = slot;
        slot = k;
    } else {
        nxt[k] = first;
        first = k;
        if (nxt[k] != SoMark<K>::NIL) before[nxt[k] % B] = k;
        before[b] = SoMark<K>::HEAD;
    }
}

// out[0..m) = iteration order after inserting seq[0..m) (distinct keys < key_limit <= 541, and < 253 for K = uint8_t);
// work needs so_work_size(key_limit, m) entries.
template <typename K>
DSM_HD inline void set_iteration_order(const K* seq, uint32_t m, K* out, K* work, uint32_t key_limit) {
    K* nxt = work;
    K* before = work + key_limit;
    K first = SoMark<K>::NIL;
    uint32_t B = 1;
    before[0] = SoMark<K>::EMPTY;
    for (uint32_t t = 0; t < m; ++t) {
        const uint32_t nb = so_next_buckets(t + 1);
        if (nb != B) {  // _M_rehash_aux(unique keys) before the insertion that crosses the threshold: re-insert in list order
            for (uint32_t i = 0; i < nb; ++i) before[i] = SoMark<K>::EMPTY;
            K p = first;
            first = SoMark<K>::NIL;
            while (p != SoMark<K>::NIL) {
                const K nx = nxt[p];
                so_insert<K>(nxt, before, nb, first, p);
                p = nx;
            }
            B = nb;
        }
        so_insert<K>(nxt, before, B, first, seq[t]);
    }
    uint32_t i = 0;
    for (K p = first; p != SoMark<K>::NIL; p = nxt[p]) out[i++] = p;
}

}  // namespace dsm
