// Host-side check of the server-side stream decoder (dsm-framework_amd/csrc/stream_parse.h) without a GPU:
//   stream_parse_check <file>  ->  "ok nodes=<n> maxfreq=<f> levels=<l> sig=<hash of the level arrays>"  or  "error: <message>"
// The file holds a client stream WITHOUT the 'S' name '.' header (what dsm_trie_parse receives).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../dsm-framework_amd/csrc/stream_parse.h"

static std::string g_err;
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
    int rc = dsm::parse_client_stream(buf.data(), buf.size(), L, &nodes, &maxfreq);
    if (rc) { printf("error: %s\n", g_err.c_str()); return 1; }
    unsigned long long sig = 1469598103934665603ull;  // FNV-1a over (freq, pl, fc) of every node, level by level
    auto mixin = [&](unsigned long long v) { for (int i = 0; i < 8; ++i) { sig ^= (v >> (8 * i)) & 0xFF; sig *= 1099511628211ull; } };
    unsigned long long per_level_nodes = 0;
    for (auto& lv : L) {
        per_level_nodes += lv.freq.size();
        for (size_t i = 0; i < lv.freq.size(); ++i) { mixin(lv.freq[i]); mixin(lv.pl[i]); mixin(lv.fc[i]); }
    }
    printf("ok nodes=%llu maxfreq=%llu levels=%zu stored=%llu sig=%llx\n", (unsigned long long)nodes, (unsigned long long)maxfreq, L.size(), per_level_nodes, sig);
    return 0;
}
