// dsm_oracle.cpp -- CPU restatement of the reference's substring-enumeration hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (dsm-framework_amd/, include/) links,
// imports, calls or executes this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it, and only as the checker / reported baseline.
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
// (a) .fmi files written by the unmodified reference builder, (b) raw client byte streams
// captured from the unmodified reference metaenumerate, (c) stdout of the unmodified reference
// metaserver -- all produced by tests/golden/make_golden.py from oracle/_ref binaries.
//
// Each function cites the reference file:line it follows (paths relative to /root/reference).
// The data layout is deliberately the reference's (three arrays per bitvector: data/Rs/Rb,
// one bitvector per Huffman-shaped wavelet-tree node) so that timing it is a fair CPU baseline.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_set>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef uint8_t u8;

namespace {

thread_local std::string g_err;

// ---------------------------------------------------------------------------------------------
// BitRank: plain bitvector + superblock (256 bit) + block (64 bit) counts.  BitRank.cpp:111-195
// ---------------------------------------------------------------------------------------------
struct BitRank {
    u64 n = 0, integers = 0;
    uint32_t b = 0, s = 0;
    std::vector<u64> data, Rs;
    std::vector<u8> Rb;
    // BitRank.cpp:191-195 -- ones in [0..i]; i = (u64)-1 wraps to 0 and yields 0.
    inline u64 rank(u64 i) const {
        ++i;
        return Rs[i >> 8] + Rb[i >> 6] + (u64)__builtin_popcountll(data[i >> 6] & ((1ull << (i & 63)) - 1));
    }
    // BitRank.cpp:338-340
    inline bool bit(u64 i) const { return (data[i >> 6] >> (i & 63)) & 1; }
};

struct Reader {
    const u8* p;
    size_t n, pos = 0;
    bool ok = true;
    Reader(const u8* p_, size_t n_) : p(p_), n(n_) {}
    void get(void* dst, size_t k) {
        if (pos + k > n) { ok = false; memset(dst, 0, k); pos = n; return; }
        memcpy(dst, p + pos, k);
        pos += k;
    }
    template <class T> T rd() { T v; get(&v, sizeof(T)); return v; }
};

// HuffWT node, pointer-free.  HuffWT.h:49-54, load order HuffWT.cpp:57-71 (pre-order).
struct WTNode {
    bool leaf = true;
    u8 ch = 0;
    int left = -1, right = -1, br = -1;
};

struct CodeEntry { u64 count; uint32_t bits, code; };  // HuffWT.h:13-46

struct Index {
    u64 n = 0;
    uint32_t samplerate = 0;
    u64 C[256];
    u64 bwtEndPos = 0;
    CodeEntry ct[256];
    std::vector<WTNode> nodes;
    std::vector<BitRank> brs;
    uint32_t numberOfTexts = 0;
    u64 maxTextLength = 0;
    u8 version = 0;

    int load_node(Reader& r) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        u8 leaf = r.rd<u8>();
        u8 ch = r.rd<u8>();
        nodes[id].leaf = leaf != 0;
        nodes[id].ch = ch;
        if (!r.ok) return id;
        if (!leaf) {
            BitRank b;
            b.n = r.rd<u64>();
            b.integers = r.rd<u64>();
            b.b = r.rd<uint32_t>();
            b.s = r.rd<uint32_t>();
            if (!r.ok || b.b != 64 || b.s != 256 || b.integers > (r.n / 8) + 1) { r.ok = false; return id; }
            b.data.resize(b.integers);
            r.get(b.data.data(), 8 * b.integers);
            b.Rs.resize(b.n / b.s + 1);
            r.get(b.Rs.data(), 8 * b.Rs.size());
            b.Rb.resize(b.n / b.b + 1);
            r.get(b.Rb.data(), b.Rb.size());
            nodes[id].br = (int)brs.size();
            brs.push_back(std::move(b));
            if (!r.ok) return id;
            int l = load_node(r);
            nodes[id].left = l;
            if (!r.ok) return id;
            int rr = load_node(r);
            nodes[id].right = rr;
        }
        return id;
    }

    // FMIndex.cpp:245-357 (loader), layout of FMIndex.cpp:155-217 (writer).
    bool load(const u8* buf, size_t len) {
        Reader r(buf, len);
        version = r.rd<u8>();
        if (version != 17 && version != 16 && version != 15 && version != 14) {
            g_err = "FMIndex::FMIndex(): invalid save file version.";
            return false;
        }
        n = r.rd<u64>();
        samplerate = r.rd<uint32_t>();
        if (version == 14) for (int i = 0; i < 256; ++i) C[i] = r.rd<uint32_t>();
        else r.get(C, sizeof(C));
        bwtEndPos = r.rd<u64>();
        for (int i = 0; i < 256; ++i) {  // HuffWT.h:21-37
            if (version < 16) ct[i].count = r.rd<uint32_t>(); else ct[i].count = r.rd<u64>();
            ct[i].bits = r.rd<uint32_t>();
            ct[i].code = r.rd<uint32_t>();
        }
        load_node(r);
        numberOfTexts = r.rd<uint32_t>();
        maxTextLength = r.rd<u64>();
        u8 nameFlag = r.rd<u8>();
        u8 tsFlag = r.rd<u8>();
        (void)r.rd<u8>();        // colorCoded
        (void)r.rd<uint32_t>();  // rotationLength
        if (!r.ok) { g_err = "file read error (truncated .fmi)"; return false; }
        if (nameFlag || tsFlag) { g_err = ".fmi with name/text storage is not produced by builder; unsupported"; return false; }
        // FMIndex.cpp:346-356: a C[] with truncated (32-bit, version 14) values is recounted through the wavelet tree
        // (recomputeC, FMIndex.cpp:219-237); the reference also saves the repaired index as <name>.reC, which a checker does not.
        for (int i = 1; i < 256; ++i) {
            if (C[i] < C[i - 1]) {
                for (int j = 0; j < 256; ++j) C[j] = 0;
                for (u64 k = 0; k < n; ++k) C[(int)access(k, nullptr)]++;
                u64 prev = C[0], temp;
                C[0] = 0;
                for (int j = 1; j < 256; ++j) { temp = C[j]; C[j] = C[j - 1] + prev; prev = temp; }
                break;
            }
        }
        return true;
    }

    // HuffWT.h:66-83
    inline u64 wt_rank(u8 c, u64 i, u64* rank_ops) const {
        if (ct[c].count == 0) return 0;
        int t = 0;
        unsigned level = 0;
        uint32_t code = ct[c].code;
        while (!nodes[t].leaf) {
            const BitRank& b = brs[nodes[t].br];
            if (rank_ops) ++*rank_ops;
            if ((code & (1u << level)) == 0) { i = i - b.rank(i); t = nodes[t].left; }
            else { i = b.rank(i) - 1; t = nodes[t].right; }
            ++level;
        }
        return i + 1;
    }
    // HuffWT.h:126-140
    inline u8 access(u64 i, u64* rank_ops) const {
        int t = 0;
        while (!nodes[t].leaf) {
            const BitRank& b = brs[nodes[t].br];
            if (rank_ops) ++*rank_ops;
            if (b.bit(i)) { i = b.rank(i) - 1; t = nodes[t].right; }
            else { i = i - b.rank(i); t = nodes[t].left; }
        }
        return nodes[t].ch;
    }
    // FMIndex.h:84-90
    inline u64 LF(u8 c, u64 i, u64* lf_steps, u64* rank_ops) const {
        if (lf_steps) ++*lf_steps;
        if (C[(int)c + 1] - C[(int)c] == 0) return C[(int)c];
        return C[(int)c] + wt_rank(c, i, rank_ops);
    }
};

// ---------------------------------------------------------------------------------------------
// Client side: EnumerateQuery.  Query.h:37-51, EnumerateQuery.cpp:9-290, ClientSocket.h:12-46
// ---------------------------------------------------------------------------------------------
struct Enumerator {
    const Index& tc;
    std::string enforcepath;
    unsigned fmin, maxdepth;
    std::vector<u8>& out;
    u64 reported = 0, lf_steps = 0, rank_ops = 0;
    std::vector<u64> smin, smax;
    std::vector<u64> extmin[4], extmax[4];
    std::vector<u8> match;
    static constexpr const char* ALPHABET = "ACGT";  // Query.cpp:3

    Enumerator(const Index& t, const std::string& ep, unsigned fm, unsigned md, std::vector<u8>& o)
        : tc(t), enforcepath(ep), fmin(fm), maxdepth(md), out(o) {}

    inline u64 LF(u8 c, u64 i) { return tc.LF(c, i, &lf_steps, &rank_ops); }
    inline void putc(u8 c) { out.push_back(c); }
    // ClientSocket.h:20-39
    inline void putulong(u64 u) {
        if (u < (1u << 7)) { putc((u8)((u & 0xFF) | 0x80)); return; }
        u8 l = 0;
        u64 tmp = u;
        do { ++l; } while ((u >>= 8));
        putc(l);
        u = tmp;
        do { putc((u8)(u & 0xFF)); } while ((u >>= 8));
    }
    // Query.h:37-45 + EnumerateQuery.cpp:39-58
    bool pushChar(u8 c) {
        u64 nmin = LF(c, smin.back() - 1);
        u64 nmax = LF(c, smax.back()) - 1;
        if (nmin > nmax) return false;
        smin.push_back(nmin);
        smax.push_back(nmax);
        match.push_back(c);
        for (unsigned i = 0; i < 4; ++i) {
            u64 emin = extmin[i].back(), emax = extmax[i].back();
            if (emin <= emax) {
                emin = LF(c, emin - 1);
                emax = LF(c, emax) - 1;
            }
            extmin[i].push_back(emin);
            extmax[i].push_back(emax);
        }
        return true;
    }
    void popChar() {  // Query.h:47-51 + EnumerateQuery.cpp:60-68
        smin.pop_back(); smax.pop_back(); match.pop_back();
        for (unsigned i = 0; i < 4; ++i) { extmin[i].pop_back(); extmax[i].pop_back(); }
    }
    u8 leftChar() {  // EnumerateQuery.cpp:77-103
        bool matches = false, any = false;
        unsigned c = 255;
        for (unsigned i = 0; i < 4; ++i) {
            u64 a = extmin[i].back(), b = extmax[i].back();
            if (a <= b) {
                any = true;
                c = i;
                if (a == smin.back() && b == smax.back()) matches = true;
            }
        }
        if (matches) return (u8)ALPHABET[c];
        if (any) return 'N';
        return '0';
    }
    void closeNode() {  // EnumerateQuery.cpp:213-222 (same tail at :138-147, :279-288)
        putulong(smax.back() - smin.back() + 1);
        if (match.size() <= 6) { putc('R'); putulong(reported); }
        putc(leftChar());
        putc(')');
        popChar();
    }
    void followOneBranch() {  // EnumerateQuery.cpp:105-149
        unsigned i = 0;
        u8 c = tc.access(smin.back(), &rank_ops);
        while (c == 'A' || c == 'C' || c == 'G' || c == 'T') {
            if (match.size() >= maxdepth) break;
            ++i;
            if (!pushChar(c)) { fprintf(stderr, "oracle: followOneBranch pushChar failed\n"); abort(); }
            putc('('); putc(c);
            ++reported;
            c = tc.access(smin.back(), &rank_ops);
        }
        while (i > 0) { --i; closeNode(); }  // freq is 1 here: smax-smin+1 == 1
    }
    void nextSymbol() {  // EnumerateQuery.cpp:151-238
        if (match.size() >= maxdepth) return;
        if (smax.back() - smin.back() == 0) { followOneBranch(); return; }
        for (int k = 0; k < 4; ++k) {
            u8 c = (u8)ALPHABET[k];
            if (!pushChar(c)) continue;
            if (smax.back() - smin.back() + 1 < fmin) { popChar(); continue; }
            putc('('); putc(match.back());
            ++reported;
            nextSymbol();
            closeNode();
        }
    }
    void nextEnforced() {  // EnumerateQuery.cpp:240-290
        u8 c = (u8)enforcepath[match.size()];
        if (!pushChar(c)) return;
        if (smax.back() - smin.back() + 1 < fmin) { popChar(); return; }
        putc('('); putc(match.back());
        ++reported;
        if (enforcepath.size() > match.size()) nextEnforced(); else nextSymbol();
        closeNode();
    }
    void enumerate() {  // EnumerateQuery.cpp:9-37
        smin.assign(1, 0);
        smax.assign(1, tc.n - 1);
        for (unsigned i = 0; i < 4; ++i) {
            extmin[i].assign(1, LF((u8)ALPHABET[i], (u64)-1));
            extmax[i].assign(1, LF((u8)ALPHABET[i], tc.n - 1) - 1);
        }
        if (enforcepath.empty()) nextSymbol(); else nextEnforced();
    }
};

// ---------------------------------------------------------------------------------------------
// Server side: TrieReader.h:32-106, ServerSocket.h:25-83, metaserver.cpp:147-486
// ---------------------------------------------------------------------------------------------
struct Stream {
    const u8* p; size_t n; size_t pos = 0;
    bool eof = false;
    u64 cnt = 0, occs = 0;  // TrieReader::n, ::occs
    int id = -1;
    std::string name;
    u8 getc() { if (pos >= n) { eof = true; return 0; } return p[pos++]; }
    u8 peek() { if (pos >= n) { eof = true; return 0; } return p[pos]; }
    bool good() const { return !eof; }
    u64 getulong() {  // ServerSocket.h:45-58
        u8 c = getc();
        if (c >= 0x80) return (u64)(c ^ 0x80);
        u64 u = 0;
        for (u8 i = 0; i < c; ++i) { u64 j = getc(); u |= (j & 0xFF) << (8 * i); }
        return u;
    }
    bool hasChild() {  // TrieReader.h:32-50
        if (!good()) return false;
        u8 c = peek();
        if (!good()) return false;
        return c == '(';
    }
};

struct ServerParams {
    unsigned pmin = 2, pmax = 0, mindepth = 0;
    double emin = 0.0, emax = -1.0;
};

typedef std::unordered_set<unsigned> readerset;  // metaserver.cpp:23 -- iteration order is part of parity

struct Server {
    std::vector<Stream> rd;
    ServerParams P;
    std::string path, out, err;
    u64 total_paths = 0, total_output = 0, total_occs = 0;
    bool failed = false;

    void fail(const std::string& m) { if (!failed) { failed = true; err = m; } }

    int readChildSym(Stream& s) {  // TrieReader.h:51-67
        u8 c = s.getc();
        if (c != '(') { fail("expecting ( byte at reader " + s.name); return -1; }
        c = s.getc();
        int k = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
        if (k < 0) { fail("expecting dna byte at reader " + s.name); return -1; }
        ++s.cnt;
        return k;
    }
    void checkR(Stream& s) {  // TrieReader.h:84-106
        u8 c = s.getc();
        if (c != 'R') { fail("expecting R byte at reader " + s.name); return; }
        u64 checksum = s.getulong();
        if (checksum != s.cnt) fail("checksum mismatch at reader " + s.name);
    }
    u8 readClose(Stream& s) {  // TrieReader.h:75-81
        u8 l = s.getc();
        u8 c = s.getc();
        if (c != ')') fail("expecting ) byte at reader " + s.name);
        return l;
    }
    void traverseOne(unsigned r) {  // metaserver.cpp:211-226 (does NOT consume R: reference quirk)
        if (failed) return;
        Stream& s = rd[r];
        while (s.hasChild()) {
            if (readChildSym(s) < 0) return;
            traverseOne(r);
            if (failed) return;
        }
        s.occs = s.getulong();
        readClose(s);
        ++total_paths;
    }
    void traverse(const readerset& treaders) {  // metaserver.cpp:269-486
        if (failed) return;
        if (treaders.size() == 1 && P.pmin > 1) { traverseOne(*treaders.begin()); return; }
        readerset atr = treaders;
        std::vector<readerset> children(4);
        unsigned numberOfChildren = 0;
        for (;;) {
            // readChildren(), metaserver.cpp:159-189
            for (readerset::const_iterator it = atr.begin(); it != atr.end(); ++it) {
                Stream& s = rd[*it];
                if (s.hasChild()) {
                    int k = readChildSym(s);
                    if (k < 0) return;
                    children[k].insert(*it);
                }
            }
            int i = 0;
            while (i < 4 && children[i].size() == 0) ++i;
            if (i == 4) break;
            atr = children[i];
            ++numberOfChildren;
            path.push_back("ACGT"[i]);
            traverse(children[i]);
            path.resize(path.size() - 1);
            children[i].clear();
            if (failed) return;
        }
        if (path.empty()) return;
        u8 leftChar = 0;
        u64 sumN = rd.size();
        double sumNlogN = 0;
        for (readerset::const_iterator it = treaders.begin(); it != treaders.end(); ++it) {
            Stream& s = rd[*it];
            u64 freq = s.occs = s.getulong();
            sumN += freq;
            sumNlogN += (double)(freq + 1) * log(freq + 1) / log(2);  // metaserver.cpp:379
            if (path.size() <= 6) checkR(s);
            u8 l = readClose(s);
            if (failed) return;
            if (leftChar == 0) leftChar = l;
            else if (leftChar != l) leftChar = 'N';
        }
        double entropy = log(sumN) / log(2) - sumNlogN / (double)sumN;  // metaserver.cpp:389
        bool output = true;  // metaserver.cpp:406-419
        if (path.size() < P.mindepth) output = false;
        if (P.pmax != 0 && treaders.size() > P.pmax) output = false;
        if (treaders.size() < P.pmin) output = false;
        if (P.emax > 0 && (entropy < P.emin || entropy > P.emax)) output = false;
        if (numberOfChildren == 1 && treaders.size() == atr.size()) output = false;
        if (leftChar == 'A' || leftChar == 'C' || leftChar == 'G' || leftChar == 'T') output = false;
        ++total_paths;
        if (output) {  // metaserver.cpp:467-485
            ++total_output;
            char buf[64];
            out += path;
            snprintf(buf, sizeof buf, " %f", entropy);
            out += buf;
            for (readerset::const_iterator it = treaders.begin(); it != treaders.end(); ++it) {
                snprintf(buf, sizeof buf, " %d:%lu", rd[*it].id, (unsigned long)rd[*it].occs);
                out += buf;
                ++total_occs;
            }
            out += '\n';
        }
    }
};

char* dup_out(const std::string& s, size_t* len) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    if (len) *len = s.size();
    return p;
}

}  // namespace

// =================================================================================================
// C entry points used by the tests (ctypes) and by bench.py's cpu_baseline leg.
// =================================================================================================
extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_index_load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { g_err = "file not found"; return nullptr; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<u8> buf((size_t)sz);
    if (fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); g_err = "read error"; return nullptr; }
    fclose(f);
    Index* ix = new Index();
    if (!ix->load(buf.data(), buf.size())) { delete ix; return nullptr; }
    return ix;
}
void orc_index_free(void* h) { delete (Index*)h; }
u64 orc_index_length(void* h) { return ((Index*)h)->n; }  // FMIndex.h:68-69
void orc_index_meta(void* h, u64* C, u64* count, uint32_t* bits, uint32_t* code) {
    Index* ix = (Index*)h;
    for (int i = 0; i < 256; ++i) { C[i] = ix->C[i]; count[i] = ix->ct[i].count; bits[i] = ix->ct[i].bits; code[i] = ix->ct[i].code; }
}
u64 orc_lf(void* h, unsigned c, u64 i) { return ((Index*)h)->LF((u8)c, i, nullptr, nullptr); }
void orc_lf_batch(void* h, const u8* c, const u64* i, u64* out, size_t k) {
    Index* ix = (Index*)h;
    for (size_t j = 0; j < k; ++j) out[j] = ix->LF(c[j], i[j], nullptr, nullptr);
}
unsigned orc_getL(void* h, u64 i) { return ((Index*)h)->access(i, nullptr); }  // FMIndex.h:99-102
void orc_bwt(void* h, u8* out) { Index* ix = (Index*)h; for (u64 i = 0; i < ix->n; ++i) out[i] = ix->access(i, nullptr); }

// One client connection: 'S' name '.' then the node grammar (metaenumerate.cpp:283-286 + EnumerateQuery).
// stats[0]=reported stats[1]=lf_steps stats[2]=rank_ops.  Returned buffer is malloc'd.
u8* orc_enumerate(void* h, const char* name, const char* prefix, unsigned fmin, unsigned maxdepth,
                  size_t* len, u64* stats) {
    Index* ix = (Index*)h;
    std::vector<u8> out;
    if (name) { out.push_back('S'); for (const char* p = name; *p; ++p) out.push_back((u8)*p); out.push_back('.'); }
    Enumerator e(*ix, prefix ? prefix : "", fmin, maxdepth, out);
    e.enumerate();
    if (stats) { stats[0] = e.reported; stats[1] = e.lf_steps; stats[2] = e.rank_ops; }
    u8* p = (u8*)malloc(out.size() + 1);
    memcpy(p, out.data(), out.size());
    *len = out.size();
    return p;
}

// metaserver for one prefix: names (id = position), one captured stream per client.
// Returns stdout text (malloc'd) or NULL on a fatal server error (orc_last_error()).
// stats[0]=total_paths stats[1]=total_output stats[2]=total_occs
char* orc_server(int nnames, const char** names, int nstreams, const u8** streams, const size_t* lens,
                 unsigned pmin, unsigned pmax, unsigned mindepth, double emin, double emax,
                 size_t* outlen, u64* stats) {
    Server sv;
    sv.P.pmin = pmin; sv.P.pmax = pmax; sv.P.mindepth = mindepth; sv.P.emin = emin; sv.P.emax = emax;
    std::map<std::string, int> libtoid;  // metaserver.cpp:606-653
    for (int i = 0; i < nnames; ++i) {
        if (libtoid.count(names[i])) { g_err = "DUPLICATE CLIENT NAME"; return nullptr; }
        int id = (int)libtoid.size();
        libtoid[names[i]] = id;
    }
    sv.rd.resize(libtoid.size());
    std::vector<bool> seen(libtoid.size(), false);
    if ((size_t)nstreams != libtoid.size()) { g_err = "expected one stream per name"; return nullptr; }
    for (int k = 0; k < nstreams; ++k) {  // metaserver.cpp:682-728
        Stream s;
        s.p = streams[k]; s.n = lens[k];
        if (s.getc() != 'S') { g_err = "received invalid start byte"; return nullptr; }
        std::string nm;
        for (u8 c = s.getc(); c != '.'; c = s.getc()) { if (!s.good()) { g_err = "bad header"; return nullptr; } nm += (char)c; }
        auto f = libtoid.find(nm);
        if (f == libtoid.end()) { g_err = "received invalid libname: " + nm; return nullptr; }
        if (seen[f->second]) { g_err = "DUPLICATE CONNECTING CLIENT"; return nullptr; }
        seen[f->second] = true;
        s.id = f->second; s.name = nm;
        sv.rd[f->second] = s;
    }
    readerset rb;
    for (size_t i = 0; i < sv.rd.size(); ++i) rb.insert((unsigned)i);  // metaserver.cpp:736-739
    sv.traverse(rb);
    if (sv.failed) { g_err = sv.err; return nullptr; }
    if (stats) { stats[0] = sv.total_paths; stats[1] = sv.total_output; stats[2] = sv.total_occs; }
    return dup_out(sv.out, outlen);
}

// In-process client+server for a list of prefixes: the reference pipeline without sockets.
// With threads>1 prefixes run on OpenMP threads (the reference's one-thread-per-prefix client,
// metaenumerate.cpp:268, and one server process per prefix).  Output = concatenation in prefix order.
// stats: [0]=reported(sum over samples) [1]=lf_steps [2]=rank_ops [3]=total_paths [4]=total_output [5]=total_occs
char* orc_mine(int nidx, void** idx, const char** names, int nprefix, const char** prefixes,
               unsigned fmin, unsigned maxdepth, unsigned pmin, unsigned pmax, unsigned mindepth,
               double emin, double emax, int threads, size_t* outlen, u64* stats) {
    std::vector<std::string> outs(nprefix);
    std::vector<std::string> errs(nprefix);
    u64 acc[6] = {0, 0, 0, 0, 0, 0};
    bool bad = false;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int p = 0; p < nprefix; ++p) {
        std::vector<u8*> bufs(nidx);
        std::vector<size_t> lens(nidx);
        u64 loc[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < nidx; ++s) {
            u64 st[3];
            bufs[s] = orc_enumerate(idx[s], names[s], prefixes[p], fmin, maxdepth, &lens[s], st);
            loc[0] += st[0]; loc[1] += st[1]; loc[2] += st[2];
        }
        size_t ol = 0;
        u64 st[3] = {0, 0, 0};
        char* o = orc_server(nidx, names, nidx, (const u8**)bufs.data(), lens.data(), pmin, pmax, mindepth, emin, emax, &ol, st);
        if (!o) {
#ifdef _OPENMP
#pragma omp critical
#endif
            { bad = true; errs[p] = g_err; }
        } else { if (outlen) outs[p].assign(o, ol); free(o); }  // outlen == NULL: timing leg, text discarded
        loc[3] = st[0]; loc[4] = st[1]; loc[5] = st[2];
        for (int s = 0; s < nidx; ++s) free(bufs[s]);
#ifdef _OPENMP
#pragma omp critical
#endif
        for (int k = 0; k < 6; ++k) acc[k] += loc[k];
    }
    if (bad) { for (auto& e : errs) if (!e.empty()) { g_err = e; break; } return nullptr; }
    std::string all;
    for (auto& o : outs) all += o;
    if (stats) for (int k = 0; k < 6; ++k) stats[k] = acc[k];
    return dup_out(all, outlen);
}

// Client-only timing leg for the CPU baseline: enumerate every prefix of one index on `threads`
// OpenMP threads (one prefix per thread at a time), discarding the stream like a TCP sink would.
// stats as orc_enumerate, summed.
void orc_enumerate_prefixes(void* h, int nprefix, const char** prefixes, unsigned fmin, unsigned maxdepth,
                            int threads, u64* stats, u64* bytes) {
    u64 acc[3] = {0, 0, 0}, nb = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int p = 0; p < nprefix; ++p) {
        u64 st[3];
        size_t len = 0;
        u8* b = orc_enumerate(h, "x", prefixes[p], fmin, maxdepth, &len, st);
        free(b);
#ifdef _OPENMP
#pragma omp critical
#endif
        { acc[0] += st[0]; acc[1] += st[1]; acc[2] += st[2]; nb += len; }
    }
    for (int k = 0; k < 3; ++k) stats[k] = acc[k];
    if (bytes) *bytes = nb;
}

void orc_free(void* p) { free(p); }
int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
