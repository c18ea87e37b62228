// fmt6.h -- printf("%f") of a double without floating-point arithmetic (shared by format.hip's kernels and the host-side check in
// tests/native/fmt6_check.cpp, which compares it with snprintf).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define DSM_HD __host__ __device__
#else
#define DSM_HD
#endif

namespace dsm {
typedef uint64_t u64;
typedef uint32_t u32;

// ---- printf("%f") of a double, exactly ------------------------------------------------------------------------------------
// |x| * 10^6 rounded to the nearest integer, ties to even, for |x| < 2^40 (so that the result fits 64 bits); ok = false otherwise.
DSM_HD inline u64 fixed6_of(double x, bool* neg, bool* ok) {
    u64 bits;
    memcpy(&bits, &x, 8);
    *neg = (bits >> 63) != 0;   // (-0.0 and negative values that round to zero print their sign, as printf does)
    const u32 ex = (u32)((bits >> 52) & 0x7FFu);
    u64 m = bits & ((1ull << 52) - 1);
    if (ex >= 1023 + 40) { *ok = false; return 0; }  // 2^40 and up, infinities, NaNs
    *ok = true;
    int e;                       // |x| = m * 2^e
    if (ex == 0) e = -1074; else { m |= 1ull << 52; e = (int)ex - 1075; }
    // P = m * 10^6 < 2^73 as (hi, lo)
    const u64 K = 1000000ull;
    const u64 lo = m * K;
#ifdef __HIP_DEVICE_COMPILE__
    const u64 hi = __umul64hi(m, K);
#else
    const u64 hi = (u64)(((unsigned __int128)m * K) >> 64);
#endif
    const int s = -e;            // (e < 0 for every |x| < 2^40 with m >= 2^52; subnormals have s = 1074)
    if (s >= 74) return 0;       // P < 2^73 <= half of 2^s: rounds to zero, and cannot be a tie
    u64 q, rem_hi, rem_lo, half_hi, half_lo;
    if (s < 64) {                // 13 <= s here (m >= 2^52, |x| < 2^40), except for zero and subnormals which left above or have m small
        q = s == 0 ? lo : ((hi << (64 - s)) | (lo >> s));
        rem_hi = 0; rem_lo = s == 0 ? 0 : (lo & ((1ull << s) - 1));
        half_hi = 0; half_lo = s == 0 ? 0 : (1ull << (s - 1));
    } else {
        const int t = s - 64;    // 0..9
        q = hi >> t;
        rem_hi = t == 0 ? 0 : (hi & ((1ull << t) - 1)); rem_lo = lo;
        half_hi = t == 0 ? 0 : (1ull << (t - 1)); half_lo = t == 0 ? (1ull << 63) : 0;
    }
    const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
    const bool tie = rem_hi == half_hi && rem_lo == half_lo && s != 0;
    if (above || (tie && (q & 1ull))) ++q;
    return q;
}
DSM_HD inline u32 dec_digits(u64 v) {  // decimal digits of v (1 for 0)
    u32 d = 1;
    while (v >= 10) { v /= 10; ++d; }
    return d;
}
DSM_HD inline u32 put_dec(char* p, u64 v, u32 nd) {  // nd = dec_digits(v); writes nd bytes
    for (u32 k = nd; k-- > 0;) { p[k] = (char)('0' + (int)(v % 10)); v /= 10; }
    return nd;
}


}  // namespace dsm
