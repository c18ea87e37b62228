// stream_parse.h -- server-side decode of one client wire stream (host code, no device dependence).
// TrieReader's token rules (TrieReader.h:32-106: '(' sym ... varint(freq) ['R' varint(count)] leftchar ')', checksum R for
// depth <= 6) and ServerSocket's varint (ServerSocket.h:45-58), parsed into level arrays in the order the nodes appear
// = path order inside a level.  The decoder is incremental (the reference's server reads its sockets token by token,
// metaserver.cpp:682-728): bytes are fed in pieces of any size, and the entries of a level that can no longer change -- all but
// the node still open at that depth -- may be taken away at any time, so a consumer needs memory for a window of every
// level, not for the stream.  Header-only so that the CPU test suite can exercise it without a GPU
// (tests/native/stream_parse_check.cpp).
#pragma once
#include <cstdint>
#include <cstring>
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

// Merging while receiving (dsm_server_*).  The streams of one server all start with the same enforced path of chain_len symbols
// (metaenumerate's prefix, EnumerateQuery.cpp:240-290): every level down to chain_len holds ONE node.  The nodes of depth unit_depth
// (> chain_len) are the UNITS: when one closes, the decoder notes how far every deeper level has grown -- the levels are in path
// order, so the unit's subtree is the range of each level between the previous unit's mark and this one.  The nodes between the
// path and the units ("top" nodes, depths chain_len + 1 .. unit_depth - 1) and the units close in the stream's post-order; the
// decoder lists those closes (events) and keeps the stream's position -- the open path below the prefix and the last child closed
// under every open node -- from which a consumer tells whether the stream can still produce a given event.
struct UnitEvent {
    std::vector<u8> path;     // symbols (0..3) of the node below the enforced path: 1 .. unit_depth - chain_len of them
    u64 index = 0;            // of the node inside its level
    std::vector<u64> upto;    // units only: upto[k] = entries of level unit_depth + 1 + k when the unit closed
};
// p before q in the post-order of the union trie?  (children before their parent, siblings in symbol order)
inline bool post_before(const std::vector<u8>& p, const std::vector<u8>& q) {
    const size_t m = p.size() < q.size() ? p.size() : q.size();
    for (size_t i = 0; i < m; ++i)
        if (p[i] != q[i]) return p[i] < q[i];
    return p.size() > q.size();   // q is a proper prefix of p: the descendant comes first (equal paths: not before)
}

struct StreamParser {
    u32 unit_depth = 0;              // 0: no bookkeeping
    u32 chain_len = 0;
    std::vector<u8> chain_sym;       // symbols of the nodes of the enforced path (depth 1 .. chain_len), as far as the stream has them
    std::vector<UnitEvent> events;   // closes at the depths chain_len + 1 .. unit_depth, in the order they happened
    std::vector<u8> open_path;       // symbols of the open nodes of those depths
    std::vector<int> last_closed;    // last_closed[k]: symbol of the last child closed under the open node of depth chain_len + k (-1: none)
    // can the stream still produce the event with this path?  (false once it is past it; the caller adds "or the stream has ended")
    bool chain_done = false;         // the node at the end of the enforced path has closed: nothing below the path can come any more
    bool may_produce(const std::vector<u8>& ev) const {
        if (chain_done) return false;                    // climbed back above the enforced path: its subtree is complete
        if (stack.size() - 1 < chain_len) return true;   // still on its way down the enforced path (or not started)
        const size_t m = open_path.size(), n = ev.size();
        const size_t c = m < n ? m : n;
        for (size_t i = 0; i < c; ++i)
            if (open_path[i] != ev[i]) return open_path[i] < ev[i];
        if (m >= n) return true;                          // inside the event's subtree (or at the node itself, still open)
        return (int)ev[m] > last_closed[m];               // at an ancestor: the branch towards the event is still ahead
    }
    // Level l holds the entries [base[l], base[l] + L[l].freq.size()) of that level; earlier ones were taken by the consumer.
    std::vector<HostTrieLevel> L;
    std::vector<u64> base;
    std::vector<u64> stack;   // index (inside its level) of the open node at every depth; stack[0] = the root
    std::vector<u8> pend;     // bytes of an incomplete token at the end of the previous piece
    u64 opened = 0, mf = 0;
    bool failed = false, finished = false;

    StreamParser() {
        L.emplace_back();
        base.push_back(0);
        L[0].freq.push_back(0); L[0].pl.push_back(0); L[0].fc.push_back(0);
        stack.push_back(0);
    }
    u64 count(size_t l) const { return base[l] + L[l].freq.size(); }
    // entries of level l below this index are final (the node open at depth l, if any, is the last one and still collects children)
    // (the root stays open until the stream ends)
    u64 final_count(size_t l) const { return !finished && l < stack.size() ? stack[l] : count(l); }
    // the consumer has copied the first k entries still held of level l
    void drop_front(size_t l, size_t k) {
        HostTrieLevel& v = L[l];
        v.freq.erase(v.freq.begin(), v.freq.begin() + k);
        v.pl.erase(v.pl.begin(), v.pl.begin() + k);
        v.fc.erase(v.fc.begin(), v.fc.begin() + k);
        base[l] += k;
    }

    // Longest token: varint (9) 'R' varint (9) leftchar ')' = 21 bytes.  Without `last` a piece is parsed up to the point where
    // fewer than 24 bytes remain; they wait for the next piece.
    int feed(const u8* b, size_t nb, bool last) {
        if (failed) return fail(DSM_E_FORMAT, "stream: already failed");
        const u8* p = b;
        size_t n = nb;
        if (!pend.empty()) {  // (rare: only when a token straddles two pieces)
            pend.insert(pend.end(), b, b + nb);
            p = pend.data();
            n = pend.size();
        }
        size_t pos = 0;
        const int rc = parse(p, n, pos, last);
        if (rc) { failed = true; return rc; }
        std::vector<u8> rest(p + pos, p + n);
        pend.swap(rest);
        if (last && stack.size() != 1) { failed = true; return fail(DSM_E_FORMAT, "stream: unbalanced parentheses"); }
        if (last) finished = true;
        return 0;
    }

private:
    int parse(const u8* p, size_t n, size_t& pos, bool last) {
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
            if (!last && n - pos < 24) break;
            const size_t depth = stack.size() - 1;
            if (p[pos] == '(') {
                if (pos + 1 >= n) return fail(DSM_E_FORMAT, "stream: truncated child");
                const u8 sym = p[pos + 1];
                const int k = sym == 'A' ? 0 : sym == 'C' ? 1 : sym == 'G' ? 2 : sym == 'T' ? 3 : -1;
                if (k < 0) return fail(DSM_E_FORMAT, "stream: expecting dna byte");  // TrieReader.h:58-63
                pos += 2;
                if (L.size() <= depth + 1) { L.emplace_back(); base.push_back(0); }
                HostTrieLevel& me = L[depth + 1];
                HostTrieLevel& par = L[depth];
                const size_t pi = (size_t)(stack.back() - base[depth]);
                if ((par.pl[pi] & 15) == 0) par.fc[pi] = (u32)count(depth + 1);
                if ((par.pl[pi] & 15u) >> k) return fail(DSM_E_FORMAT, "stream: children out of order");  // A < C < G < T, each once
                par.pl[pi] |= (u8)(1u << k);
                me.freq.push_back(0); me.pl.push_back(0); me.fc.push_back(0);
                if (count(depth + 1) > 0xFFFFFFF0ull) return fail(DSM_E_CAPACITY, "stream: level too wide");
                if (unit_depth) {
                    if (depth + 1 <= chain_len) {
                        if (count(depth + 1) > 1) return fail(DSM_E_FORMAT, "stream: two nodes on the enforced path (the prefix length given to the server is too long for this stream)");
                        chain_sym.push_back((u8)k);
                        if (depth + 1 == chain_len) last_closed.assign(1, -1);
                    } else if (depth + 1 <= unit_depth) {
                        open_path.push_back((u8)k);
                        last_closed.resize(open_path.size() + 1);
                        last_closed[open_path.size()] = -1;
                    }
                }
                stack.push_back(count(depth + 1) - 1);
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
                const size_t mi = (size_t)(stack.back() - base[depth]);
                me.freq[mi] = f;
                me.pl[mi] |= (u8)(code << 4);
                mf = f > mf ? f : mf;
                if (unit_depth && depth > chain_len && depth <= unit_depth) {
                    UnitEvent ev;
                    ev.path = open_path;
                    ev.index = stack.back();
                    if (depth == unit_depth)
                        for (size_t l = depth + 1; l < L.size(); ++l) ev.upto.push_back(count(l));
                    events.push_back(std::move(ev));
                    last_closed[open_path.size() - 1] = (int)open_path.back();
                    open_path.pop_back();
                }
                if (unit_depth && depth == chain_len) chain_done = true;  // (chain_len == 0: the root never closes, the stream's end says so)
                stack.pop_back();
            }
        }
        return 0;
    }
};

// What a server merging while it receives does next (dsm_server, engine.hip): among the streams' next unprocessed events the first one in
// the post-order of the union trie; it is due (`ready`) when every stream either lists it as ITS next event, has ended, or can no
// longer produce it or anything before it.  taken[k]: events of stream k already processed.  (Host logic without a device:
// tests/native/server_sched_check.cpp drives it over the reference streams in random interleavings.)
struct EventPick {
    bool have = false, ready = false;
    std::vector<u8> path;
};
inline EventPick pick_event(const std::vector<const StreamParser*>& sp, const std::vector<size_t>& taken, const std::vector<bool>& ended) {
    EventPick pk;
    for (size_t k = 0; k < sp.size(); ++k)
        if (taken[k] < sp[k]->events.size()) {
            const std::vector<u8>& p = sp[k]->events[taken[k]].path;
            if (!pk.have || post_before(p, pk.path)) { pk.path = p; pk.have = true; }
        }
    if (!pk.have) return pk;
    pk.ready = true;
    for (size_t k = 0; k < sp.size() && pk.ready; ++k) {
        const bool mine = taken[k] < sp[k]->events.size() && sp[k]->events[taken[k]].path == pk.path;
        pk.ready = mine || ended[k] || !sp[k]->may_produce(pk.path);
    }
    return pk;
}

// the whole stream at once: every level stays on the host
inline int parse_client_stream(const u8* p, size_t n, std::vector<HostTrieLevel>& L, u64* nodes, u64* maxfreq) {
    StreamParser sp;
    if (int rc = sp.feed(p, n, true)) return rc;
    L.swap(sp.L);
    *nodes = sp.opened;
    *maxfreq = sp.mf;
    return 0;
}

}  // namespace dsm
