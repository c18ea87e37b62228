// Compares dsm::set_iteration_order with the real std::unordered_set<unsigned> (the container metaserver.cpp:23 uses).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <unordered_set>
#include <vector>
#include "../../dsm-framework_amd/csrc/setorder.h"
int main(int argc, char** argv) {
    int trials = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937 rng(12345);
    long checked = 0;
    for (int t = 0; t < trials; ++t) {
        unsigned d = 1 + rng() % 273;
        unsigned m = 1 + rng() % d;
        std::vector<uint16_t> ids(d);
        for (unsigned i = 0; i < d; ++i) ids[i] = (uint16_t)i;
        if (t % 3) std::shuffle(ids.begin(), ids.end(), rng);          // every third trial inserts 0..m-1 in order (the root set)
        std::vector<uint16_t> seq(ids.begin(), ids.begin() + m), out(m + 1), tmp(dsm::so_work_size(d, m));
        std::unordered_set<unsigned> s;
        for (auto k : seq) s.insert(k);
        dsm::set_iteration_order<uint16_t>(seq.data(), m, out.data(), tmp.data(), d);
        if (d <= 250) {  // the byte-sized instantiation used on the device for up to 64 samples
            std::vector<uint8_t> seq8(seq.begin(), seq.end()), out8(m + 1), tmp8(dsm::so_work_size(d, m));
            dsm::set_iteration_order<uint8_t>(seq8.data(), m, out8.data(), tmp8.data(), d);
            for (unsigned k = 0; k < m; ++k)
                if (out8[k] != out[k]) { printf("U8 MISMATCH trial %d\n", t); return 1; }
        }
        unsigned i = 0;
        for (auto it = s.begin(); it != s.end(); ++it, ++i)
            if (*it != out[i]) { printf("MISMATCH trial %d d=%u m=%u at %u: real %u model %u\n", t, d, m, i, *it, out[i]); return 1; }
        // copy construction / assignment keep the order (metaserver.cpp:322,332)
        std::unordered_set<unsigned> c(s), a;
        a.insert(999);
        a = s;
        i = 0;
        for (auto it = c.begin(), jt = a.begin(); it != c.end(); ++it, ++jt, ++i)
            if (*it != out[i] || *jt != out[i]) { printf("COPY MISMATCH trial %d\n", t); return 1; }
        checked += m;
    }
    printf("ok %ld keys in %d trials\n", checked, trials);
    return 0;
}
