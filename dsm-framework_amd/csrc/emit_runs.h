// emit_runs.h -- a chunk of tuples in which a handful failed the entropy test leaves the emitter as the RUNS of kept tuples between
// them, in place: only the offsets of a run are rebased to its first tuple (engine.hip, emit_job).  Host code, no device types: the
// index arithmetic is checked on the CPU (tests/native/emit_runs_check.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dsm {

// runs [seg[2i], seg[2i+1]) of tuples whose verdict is not `drop`, in order
inline void kept_runs(const uint8_t* keep, uint32_t nt, uint8_t drop, std::vector<uint32_t>& seg) {
    seg.clear();
    uint32_t from = 0;
    for (const uint8_t* q = keep; (q = (const uint8_t*)memchr(q, drop, (size_t)(keep + nt - q))) != nullptr; ++q) {
        const uint32_t r = (uint32_t)(q - keep);
        if (r > from) { seg.push_back(from); seg.push_back(r); }
        from = r + 1;
    }
    if (nt > from) { seg.push_back(from); seg.push_back(nt); }
}

// The offset arrays have nt + 1 entries; run i owns the entries seg[2i] .. seg[2i+1] (its closing one included: the entry of the
// dropped tuple behind it, or entry nt).  Entries lo .. hi - 1 are rebased by the bases of the runs that own them (bp / bq: the
// runs' first entries BEFORE any rebasing); entries no run owns are left alone.  Ranges of different callers may be disjoint pieces
// of 0 .. nt in any order.
inline void rebase_runs(uint32_t* rel_path, uint32_t* rel_pair, const std::vector<uint32_t>& seg, const std::vector<uint32_t>& bp,
                        const std::vector<uint32_t>& bq, uint32_t lo, uint32_t hi) {
    const size_t ns = seg.size() / 2;
    size_t i = (size_t)(std::upper_bound(seg.begin(), seg.end(), lo) - seg.begin()) / 2;  // the run that contains lo, or the next one
    if (i > 0 && lo <= seg[2 * (i - 1) + 1]) --i;                                          // (lo is the closing entry of the run before)
    for (; i < ns && seg[2 * i] < hi; ++i) {
        const uint32_t a = seg[2 * i] > lo ? seg[2 * i] : lo, b = seg[2 * i + 1] + 1 < hi ? seg[2 * i + 1] + 1 : hi;
        const uint32_t sp = bp[i], sq = bq[i];
        if (!sp && !sq) continue;
        for (uint32_t r = a; r < b; ++r) { rel_path[r] -= sp; rel_pair[r] -= sq; }
    }
}

}  // namespace dsm
