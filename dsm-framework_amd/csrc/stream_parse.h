// stream_parse.h -- server-side decode of one client wire stream (host code, no device dependence).
// TrieReader's token rules (TrieReader.h:32-106: '(' sym ... varint(freq) ['R' varint(count)] leftchar ')', checksum R for
// depth <= 6) and ServerSocket's varint (ServerSocket.h:45-58), parsed into level arrays in the order the nodes appear
// = path order inside a level.  Header-only so that the CPU test suite can exercise it without a GPU
// (tests/native/stream_parse_check.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

namespace dsm {
typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
int fail(int code, const std::string& msg);

struct HostTrieLevel {
    std::vector<u64> freq;
    std::vector<u8> pl;
    std::vector<u32> fc;
};

inline int parse_client_stream(const u8* p, size_t n, std::vector<HostTrieLevel>& L, u64* nodes, u64* maxfreq) {
    L.clear();
    L.emplace_back();
    L[0].freq.push_back(0); L[0].pl.push_back(0); L[0].fc.push_back(0);
    std::vector<u32> stack;  // index of the open node at every depth (stack[0] = root)
    stack.push_back(0);
    size_t pos = 0;
    u64 opened = 0, mf = 0;
    auto varint = [&](u64& v) -> bool {  // ServerSocket.h:45-58
        if (pos >= n) return false;
        u8 c = p[pos++];
        if (c >= 0x80) { v = (u64)(c ^ 0x80); return true; }
        if (c > 8 || pos + c > n) return false;
        v = 0;
        for (u8 i = 0; i < c; ++i) v |= (u64)p[pos++] << (8 * i);
        return true;
    };
    while (pos < n) {
        const size_t depth = stack.size() - 1;
        if (p[pos] == '(') {
            if (pos + 1 >= n) return fail(DSM_E_FORMAT, "stream: truncated child");
            const u8 sym = p[pos + 1];
            const int k = sym == 'A' ? 0 : sym == 'C' ? 1 : sym == 'G' ? 2 : sym == 'T' ? 3 : -1;
            if (k < 0) return fail(DSM_E_FORMAT, "stream: expecting dna byte");  // TrieReader.h:58-63
            pos += 2;
            if (L.size() <= depth + 1) L.emplace_back();
            HostTrieLevel& me = L[depth + 1];
            HostTrieLevel& par = L[depth];
            const u32 pi = stack.back();
            if ((par.pl[pi] & 15) == 0) par.fc[pi] = (u32)me.freq.size();
            if ((par.pl[pi] & 15u) >> k) return fail(DSM_E_FORMAT, "stream: children out of order");  // A < C < G < T, each once
            par.pl[pi] |= (u8)(1u << k);
            me.freq.push_back(0); me.pl.push_back(0); me.fc.push_back(0);
            if (me.freq.size() > 0xFFFFFFF0ull) return fail(DSM_E_CAPACITY, "stream: level too wide");
            stack.push_back((u32)(me.freq.size() - 1));
            ++opened;
        } else {
            if (depth == 0) return fail(DSM_E_FORMAT, "stream: unexpected byte at top level");
            u64 f = 0;
            if (!varint(f)) return fail(DSM_E_FORMAT, "stream: bad frequency");
            if (depth <= 6) {  // TrieReader.h:84-106
                if (pos >= n || p[pos] != 'R') return fail(DSM_E_FORMAT, "stream: expecting R byte");
                ++pos;
                u64 chk = 0;
                if (!varint(chk)) return fail(DSM_E_FORMAT, "stream: bad checksum");
                if (chk != opened) return fail(DSM_E_FORMAT, "stream: checksum mismatch");
            }
            if (pos + 2 > n) return fail(DSM_E_FORMAT, "stream: truncated close");
            const u8 lc = p[pos], cl = p[pos + 1];
            pos += 2;
            if (cl != ')') return fail(DSM_E_FORMAT, "stream: expecting ) byte");  // TrieReader.h:75-81
            const int code = lc == '0' ? 0 : lc == 'A' ? 1 : lc == 'C' ? 2 : lc == 'G' ? 3 : lc == 'T' ? 4 : lc == 'N' ? 5 : -1;
            if (code < 0) return fail(DSM_E_FORMAT, "stream: bad left char");
            HostTrieLevel& me = L[depth];
            const u32 mi = stack.back();
            me.freq[mi] = f;
            me.pl[mi] |= (u8)(code << 4);
            mf = f > mf ? f : mf;
            stack.pop_back();
        }
    }
    if (stack.size() != 1) return fail(DSM_E_FORMAT, "stream: unbalanced parentheses");
    *nodes = opened;
    *maxfreq = mf;
    return 0;
}

}  // namespace dsm
