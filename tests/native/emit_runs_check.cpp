// CPU check of csrc/emit_runs.h: the kept runs of a chunk and the rebasing of their offsets, split over any number of callers, against
// a direct restatement (every run's offsets recomputed from the tuples' own lengths).  usage: emit_runs_check <cases> <seed>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../dsm-framework_amd/csrc/emit_runs.h"

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 2000;
    std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
    long runs_seen = 0;
    for (int it = 0; it < cases; ++it) {
        const uint32_t nt = 1 + (uint32_t)(rng() % (it % 7 == 0 ? 5 : 3000));
        const double pdrop = it % 5 == 0 ? 0.5 : (it % 5 == 1 ? 0.0 : 0.01);
        std::vector<uint8_t> keep(nt);
        std::vector<uint32_t> plen(nt), qlen(nt), rel_path(nt + 1), rel_pair(nt + 1);
        for (uint32_t r = 0; r < nt; ++r) {
            keep[r] = (double)(rng() % 1000000) / 1e6 < pdrop ? 0 : (rng() % 50 == 0 ? 2 : 1);   // 0 = dropped; 1, 2 = kept
            plen[r] = (uint32_t)(rng() % 40);
            qlen[r] = (uint32_t)(rng() % 4);
        }
        if (it % 11 == 0) keep[0] = 0;
        if (it % 13 == 0) keep[nt - 1] = 0;
        rel_path[0] = rel_pair[0] = 0;
        for (uint32_t r = 0; r < nt; ++r) { rel_path[r + 1] = rel_path[r] + plen[r]; rel_pair[r + 1] = rel_pair[r] + qlen[r]; }
        std::vector<uint32_t> seg;
        dsm::kept_runs(keep.data(), nt, 0, seg);
        // the runs: maximal, in order, exactly the kept tuples
        std::vector<uint8_t> covered(nt, 0);
        for (size_t i = 0; i + 1 < seg.size(); i += 2) {
            if (seg[i] >= seg[i + 1] || (i && seg[i] <= seg[i - 1])) { printf("bad run order, case %d\n", it); return 1; }
            if ((seg[i] > 0 && keep[seg[i] - 1] != 0) || (seg[i + 1] < nt && keep[seg[i + 1]] != 0)) { printf("run not maximal, case %d\n", it); return 1; }
            for (uint32_t r = seg[i]; r < seg[i + 1]; ++r) covered[r] = 1;
        }
        for (uint32_t r = 0; r < nt; ++r) if (covered[r] != (keep[r] != 0)) { printf("coverage, case %d tuple %u\n", it, r); return 1; }
        const size_t ns = seg.size() / 2;
        runs_seen += (long)ns;
        std::vector<uint32_t> bp(ns), bq(ns);
        for (size_t i = 0; i < ns; ++i) { bp[i] = rel_path[seg[2 * i]]; bq[i] = rel_pair[seg[2 * i]]; }
        // callers: nth ranges of `per` entries as emit_job cuts them (the last one takes the closing entry nt), in shuffled order
        const unsigned nth = 1 + (unsigned)(rng() % 17);
        const uint32_t per = (nt + nth - 1) / nth;
        std::vector<unsigned> order(nth);
        for (unsigned t = 0; t < nth; ++t) order[t] = t;
        for (unsigned t = nth; t > 1; --t) std::swap(order[t - 1], order[rng() % t]);
        for (unsigned k = 0; k < nth; ++k) {
            const unsigned t = order[k];
            const uint32_t lo = t * per < nt ? t * per : nt, hi = t + 1 == nth ? nt + 1 : (lo + per < nt ? lo + per : nt);
            dsm::rebase_runs(rel_path.data(), rel_pair.data(), seg, bp, bq, lo, hi);
        }
        // every run now reads like a batch of its own: offsets from 0, the tuples' own lengths
        for (size_t i = 0; i < ns; ++i) {
            uint32_t p = 0, q = 0;
            for (uint32_t r = seg[2 * i]; r < seg[2 * i + 1]; ++r) {
                if (rel_path[r] != p || rel_pair[r] != q) { printf("offset, case %d run %zu tuple %u: %u/%u want %u/%u\n", it, i, r, rel_path[r], rel_pair[r], p, q); return 1; }
                p += plen[r]; q += qlen[r];
            }
            if (rel_path[seg[2 * i + 1]] != p || rel_pair[seg[2 * i + 1]] != q) { printf("closing entry, case %d run %zu\n", it, i); return 1; }
        }
    }
    printf("ok %d cases %ld runs\n", cases, runs_seen);
    return 0;
}
