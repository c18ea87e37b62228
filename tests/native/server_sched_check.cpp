// Host-side check of the order in which a server that merges while it receives (dsm_server, csrc/engine.hip) processes the subtrees
// of its streams -- dsm::pick_event over the streams' decoders (csrc/stream_parse.h) -- without a GPU:
//   server_sched_check <prefix length> <unit depth> <seed> <stream file>...   ->  "ok events=<n> units=<u> early=<e>"  or  "error: ..."
// The streams (client bytes without the 'S' name '.' header) are fed in pieces of random sizes in a random interleaving; after every
// piece the events that are due are taken.  Checked: they come in the post-order of the union trie, each exactly once, all of them;
// and SAFETY -- when an event is taken, no stream produces it, or one before it, later (known from a first, complete parse).
// early = events taken before the last stream had ended (the point of merging while receiving).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../dsm-framework_amd/csrc/stream_parse.h"

static std::string g_err;
namespace dsm {
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace dsm

using namespace dsm;

static std::vector<unsigned char> slurp(const char* path) {
    std::vector<unsigned char> buf;
    FILE* f = fopen(path, "rb");
    if (!f) return buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return buf;
}
static void setup(StreamParser& sp, u32 K, u32 U) {
    sp.unit_depth = U;
    sp.chain_len = K;
    if (K == 0) sp.last_closed.assign(1, -1);
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const u32 K = (u32)atol(argv[1]), U = (u32)atol(argv[2]);
    unsigned long long rng = strtoull(argv[3], nullptr, 10) * 2654435761ull + 12345;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    const int d = argc - 4;
    std::vector<std::vector<unsigned char>> body(d);
    for (int k = 0; k < d; ++k) body[k] = slurp(argv[4 + k]);
    // the complete event lists (ground truth)
    std::vector<std::vector<std::vector<u8>>> all(d);
    std::set<std::vector<u8>> uni;
    for (int k = 0; k < d; ++k) {
        StreamParser sp;
        setup(sp, K, U);
        if (sp.feed(body[k].data(), body[k].size(), true)) { printf("error: %s\n", g_err.c_str()); return 1; }
        for (auto& e : sp.events) { all[k].push_back(e.path); uni.insert(e.path); }
    }
    std::vector<StreamParser> sp(d);
    std::vector<size_t> pos(d, 0), taken(d, 0);
    std::vector<bool> ended(d, false);
    for (int k = 0; k < d; ++k) setup(sp[k], K, U);
    std::vector<std::vector<u8>> order;
    size_t early = 0;
    auto drain = [&]() -> int {
        for (;;) {
            std::vector<const StreamParser*> ps;
            for (auto& p : sp) ps.push_back(&p);
            const EventPick pk = pick_event(ps, taken, ended);
            if (!pk.have || !pk.ready) return 0;
            // safety: every event of every stream that is this one or before it has been produced already
            for (int k = 0; k < d; ++k)
                for (size_t q = sp[k].events.size(); q < all[k].size(); ++q)
                    if (all[k][q] == pk.path || post_before(all[k][q], pk.path)) { g_err = "an event was taken before stream " + std::to_string(k) + " produced it or one before it"; return 1; }
            if (!order.empty() && !post_before(order.back(), pk.path)) { g_err = "events out of post-order"; return 1; }
            order.push_back(pk.path);
            bool all_ended = true;
            for (int k = 0; k < d; ++k) all_ended = all_ended && ended[k];
            if (!all_ended) ++early;
            for (int k = 0; k < d; ++k)
                if (taken[k] < sp[k].events.size() && sp[k].events[taken[k]].path == pk.path) ++taken[k];
        }
    };
    for (;;) {
        std::vector<int> live;
        for (int k = 0; k < d; ++k) if (!ended[k]) live.push_back(k);
        if (live.empty()) break;
        const int k = live[rnd() % live.size()];
        if (pos[k] >= body[k].size()) {
            if (sp[k].feed(nullptr, 0, true)) { printf("error: %s\n", g_err.c_str()); return 1; }
            ended[k] = true;
        } else {
            static const size_t sizes[] = {1, 7, 24, 100, 1000, 5000, 40000};
            size_t n = sizes[rnd() % 7];
            if (n > body[k].size() - pos[k]) n = body[k].size() - pos[k];
            if (sp[k].feed(body[k].data() + pos[k], n, false)) { printf("error: %s\n", g_err.c_str()); return 1; }
            pos[k] += n;
        }
        if (drain()) { printf("error: %s\n", g_err.c_str()); return 1; }
    }
    if (drain()) { printf("error: %s\n", g_err.c_str()); return 1; }
    if (order.size() != uni.size()) { printf("error: %zu events taken, the union has %zu\n", order.size(), uni.size()); return 1; }
    for (int k = 0; k < d; ++k)
        if (taken[k] != sp[k].events.size()) { printf("error: stream %d has events left\n", k); return 1; }
    size_t units = 0;
    for (auto& p : order) units += p.size() == U - K;
    printf("ok events=%zu units=%zu early=%zu\n", order.size(), units, early);
    return 0;
}
