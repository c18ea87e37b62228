// entropy_tables.h -- the exact entropy of metaserver.cpp:366-389 from tables (host).
// (f+1)*log(f+1)/log(2) and log(N)/log(2) are pure functions of small integers: tabulated once with exactly the reference's
// expression (glibc's log, log(2) as the constant the reference's compiler folds), so a lookup returns the very double the inline
// evaluation would.  emit_job (engine.hip) reads them on the host; the text emitter (format.hip) uploads them and evaluates the same
// sum, in the same order, with IEEE double additions and one division on the device: bit-identical (no contraction: -ffp-contract=off).
#pragma once
#include <cmath>
#include <mutex>
#include <vector>

#include "common.h"

namespace dsm {

static const double LN2 = 0x1.62e42fefa39efp-1;  // the reference's log(2), folded by its compiler (metaserver.cpp:379,389)

// (f+1)*log(f+1)/log(2) and log(N)/log(2) are pure functions of small integers: tabulated once with exactly the
// reference's expression (metaserver.cpp:379,389), so a lookup returns the very double the inline evaluation would.
constexpr u32 TERM_TAB = 1u << 16, LOGN_TAB = 1u << 20;
inline const double* term_table() {
    static std::vector<double> t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.resize(TERM_TAB);
        for (u32 f = 0; f < TERM_TAB; ++f) t[f] = (double)((u64)f + 1) * log((double)((u64)f + 1)) / LN2;
    });
    return t.data();
}
inline const double* logn_table() {
    static std::vector<double> t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.resize(LOGN_TAB);
        t[0] = 0;
        for (u32 n = 1; n < LOGN_TAB; ++n) t[n] = log((double)n) / LN2;
    });
    return t.data();
}


// the entropy of one tuple exactly as the reference computes it (frequencies in the reference's print order)
inline double exact_entropy(u32 d, const u64* freqs, u32 n) {
    const double* terms = term_table();
    const double* logn = logn_table();
    u64 sumN = d;
    double sumNlogN = 0;
    for (u32 q = 0; q < n; ++q) {
        const u64 f = freqs[q];
        sumN += f;
        sumNlogN += f < TERM_TAB ? terms[f] : (double)(f + 1) * log((double)(f + 1)) / LN2;
    }
    return (sumN < LOGN_TAB ? logn[sumN] : log((double)sumN) / LN2) - sumNlogN / (double)sumN;
}

}  // namespace dsm
