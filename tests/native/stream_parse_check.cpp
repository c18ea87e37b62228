// Host-side check of the server-side stream decoder (dsm-framework_amd/csrc/stream_parse.h) without a GPU:
//   stream_parse_check <file> [piece]  ->  "ok nodes=<n> maxfreq=<f> levels=<l> sig=<hash of the level arrays>"  or  "error: <message>"
//   stream_parse_check <file> <piece> <unit depth>  ->  the same plus " chain=<symbols> units=<symbol:nodes of the subtree,...>" from the marks
//   the decoder leaves when a node of the unit depth closes (what dsm_server takes the subtrees out of the stream by)
// The file holds a client stream WITHOUT the 'S' name '.' header (what dsm_trie_parse receives).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../dsm-framework_amd/csrc/stream_parse.h"

static std::string g_err, g_units;
namespace dsm {
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace dsm

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    std::vector<dsm::HostTrieLevel> L;
    dsm::u64 nodes = 0, maxfreq = 0;
    int rc = 0;
    const size_t piece = argc > 2 ? (size_t)atol(argv[2]) : 0;
    if (piece == 0) {
        rc = dsm::parse_client_stream(buf.data(), buf.size(), L, &nodes, &maxfreq);
    } else {
        // the incremental decoder fed `piece` bytes at a time, with a consumer that takes every final entry away after each piece
        // (the way dsm_trie_stream moves them to the card): the collected levels must equal the one-shot parse
        dsm::StreamParser sp;
        if (argc > 3) {  // unit depth [prefix length, default unit depth - 1]
            sp.unit_depth = (dsm::u32)atol(argv[3]);
            sp.chain_len = argc > 4 ? (dsm::u32)atol(argv[4]) : sp.unit_depth - 1;
            if (sp.chain_len == 0) sp.last_closed.assign(1, -1);
        }
        auto take = [&]() {
            if (L.size() < sp.L.size()) L.resize(sp.L.size());
            for (size_t l = 0; l < sp.L.size(); ++l) {
                const size_t k = (size_t)(sp.final_count(l) - sp.base[l]);
                L[l].freq.insert(L[l].freq.end(), sp.L[l].freq.begin(), sp.L[l].freq.begin() + k);
                L[l].pl.insert(L[l].pl.end(), sp.L[l].pl.begin(), sp.L[l].pl.begin() + k);
                L[l].fc.insert(L[l].fc.end(), sp.L[l].fc.begin(), sp.L[l].fc.begin() + k);
                sp.drop_front(l, k);
            }
        };
        for (size_t o = 0; o < buf.size() && !rc; o += piece) {
            rc = sp.feed(buf.data() + o, buf.size() - o < piece ? buf.size() - o : piece, false);
            if (!rc) take();
        }
        if (!rc) rc = sp.feed(nullptr, 0, true);
        if (!rc) take();
        nodes = sp.opened;
        maxfreq = sp.mf;
        if (!rc && sp.unit_depth) {
            g_units = " chain=";
            for (auto c : sp.chain_sym) g_units += "ACGT"[c];
            g_units += " units=";
            std::vector<dsm::u64> from;
            bool first = true;
            for (const dsm::UnitEvent& ev : sp.events) {
                std::string path;
                for (auto c : ev.path) path += "ACGT"[c];
                dsm::u64 sz = 1;
                if (ev.path.size() == sp.unit_depth - sp.chain_len) {  // a unit: nodes of its subtree from the marks
                    if (from.size() < ev.upto.size()) from.resize(ev.upto.size(), 0);
                    for (size_t k = 0; k < ev.upto.size(); ++k) { sz += ev.upto[k] - from[k]; from[k] = ev.upto[k]; }
                } else {
                    sz = 0;                                            // a node between the path and the units
                }
                g_units += (first ? "" : ",") + path + ":" + std::to_string(sz) + ":" + std::to_string(L[sp.chain_len + ev.path.size()].freq[ev.index]);
                first = false;
            }
        }
    }
    if (rc) { printf("error: %s\n", g_err.c_str()); return 1; }
    unsigned long long sig = 1469598103934665603ull;  // FNV-1a over (freq, pl, fc) of every node, level by level
    auto mixin = [&](unsigned long long v) { for (int i = 0; i < 8; ++i) { sig ^= (v >> (8 * i)) & 0xFF; sig *= 1099511628211ull; } };
    unsigned long long per_level_nodes = 0;
    for (auto& lv : L) {
        per_level_nodes += lv.freq.size();
        for (size_t i = 0; i < lv.freq.size(); ++i) { mixin(lv.freq[i]); mixin(lv.pl[i]); mixin(lv.fc[i]); }
    }
    printf("ok nodes=%llu maxfreq=%llu levels=%zu stored=%llu sig=%llx%s\n", (unsigned long long)nodes, (unsigned long long)maxfreq, L.size(), per_level_nodes, sig, g_units.c_str());
    return 0;
}
