// fmt6_check.cpp -- the integer-only "%f" of csrc/fmt6.h against snprintf (glibc), on the host.
// usage: fmt6_check <count> <seed>; prints "ok <checked>" or the first mismatch.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../dsm-framework_amd/csrc/fmt6.h"

static int check(double x, uint64_t* n) {
    char want[512], got[64];
    snprintf(want, sizeof want, "%f", x);
    bool neg, ok;
    const uint64_t v = dsm::fixed6_of(x, &neg, &ok);
    if (!ok) return 0;  // outside the device's range: such a batch goes to snprintf itself
    char* p = got;
    if (neg) *p++ = '-';
    const uint64_t ip = v / 1000000ull;
    p += dsm::put_dec(p, ip, dsm::dec_digits(ip));
    *p++ = '.';
    uint32_t fr = (uint32_t)(v % 1000000ull);
    for (int k = 5; k >= 0; --k) { p[k] = (char)('0' + (int)(fr % 10u)); fr /= 10u; }
    p += 6;
    *p = 0;
    ++*n;
    if (strcmp(want, got) != 0) {
        uint64_t bits;
        memcpy(&bits, &x, 8);
        printf("mismatch: bits %016" PRIx64 " snprintf '%s' fmt6 '%s'\n", bits, want, got);
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    const uint64_t count = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    uint64_t n = 0;
    // specials and edges
    const double edge[] = {0.0, -0.0, 5e-324, -5e-324, 2.2250738585072014e-308, 4.9999995e-7, 5e-7, 5.0000005e-7, 0.0000005, 0.0000015, 0.0000025,
                           0.5, 1.5, 2.5, 9.9999995, 9.99999949999999, 9.9999996, 8.092747, 1e-300, 1099511627775.9999, 1099511627775.5,
                           0.9999995, 0.99999949999999994, 123456.7890125, 123456.7890135, -1e-10, -0.9999996};
    for (double x : edge) if (check(x, &n) || check(-x, &n)) return 1;
    std::uniform_real_distribution<double> ent(0.0, 8.1), unit(0.0, 1.0);
    for (uint64_t i = 0; i < count; ++i) {
        double x;
        switch (i % 8) {
            case 0: x = ent(rng); break;                                   // entropies
            case 1: x = (double)(rng() % 100000000ull) / 128.0; break;       // k / 2^7: seven decimals ending in 5 -- exact ties
            case 2: x = (double)(rng() % 1000000000ull) / 1024.0 * 1e-3; break;
            case 3: { uint64_t b = rng(); memcpy(&x, &b, 8); break; }      // any bit pattern (mostly out of range, NaNs, tiny)
            case 4: x = unit(rng) * 1e-6; break;                           // around the last printed digit
            case 5: x = (double)(rng() % 2000001ull) * 0.5e-6; break;       // decimal ties that are not binary ties
            case 6: x = -ent(rng) * 1e-9; break;                           // negative noise around zero (one sample: metaserver.cpp:389)
            default: x = ldexp((double)(rng() >> 11), -(int)(rng() % 80)); break;  // every binary exponent down to 2^-27
        }
        if (check(x, &n)) return 1;
    }
    printf("ok %" PRIu64 "\n", n);
    return 0;
}
