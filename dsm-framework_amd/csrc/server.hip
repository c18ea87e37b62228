// server.hip -- the server side of the drop-in surface (metaserver.cpp:159-486, 682-739; TrieReader.h:32-106): client byte streams parsed
// into device tries (dsm_trie_*), merged at once (dsm_merge) or while they arrive (dsm_server_*).  Host code around the engine's runs
// (engine_api.h) and one small kernel.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "stream_parse.h"
#include "engine_api.h"

namespace dsm {
static inline dim3 grid_for(u64 n, int t = 256) { return dim3((unsigned)((n + t - 1) / t)); }
__global__ void fc_rebase_kernel(u32* fc, size_t n, u32 delta) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fc[i] -= delta;
}
}  // namespace dsm

using namespace dsm;

extern "C" {

int dsm_trie_parse(const uint8_t* bytes, size_t n, int device, dsm_trie** out) {
    if ((!bytes && n) || !out) return fail(DSM_E_INVAL, "dsm_trie_parse: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_trie_parse: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_trie_parse: bad device ordinal");
    std::vector<HostTrieLevel> L;
    u64 nodes = 0, mf = 0;
    if (int rc = parse_client_stream(bytes, n, L, &nodes, &mf)) return rc;
    std::unique_ptr<dsm_trie, void (*)(dsm_trie*)> t(new dsm_trie(), dsm_trie_free);  // error paths release the device arrays too
    t->device = device;
    t->nodes = nodes;
    t->maxfreq = mf;
    u64 tot = 0;
    for (auto& l : L) { t->level_off.push_back(tot); tot += l.freq.size(); }
    t->level_off.push_back(tot);
    DSM_HIP(hipSetDevice(device));
    DSM_HIP(hipMalloc((void**)&t->d_freq, tot * sizeof(u64)));
    DSM_HIP(hipMalloc((void**)&t->d_pl, tot));
    DSM_HIP(hipMalloc((void**)&t->d_fc, tot * sizeof(u32)));
    for (size_t l = 0; l < L.size(); ++l) {
        const u64 o = t->level_off[l], k = L[l].freq.size();
        DSM_HIP(hipMemcpy(t->d_freq + o, L[l].freq.data(), k * sizeof(u64), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_pl + o, L[l].pl.data(), k, hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_fc + o, L[l].fc.data(), k * sizeof(u32), hipMemcpyHostToDevice));
    }
    *out = t.release();
    return DSM_OK;
}
// Incremental form of dsm_trie_parse: the bytes of a connection are fed as they arrive; the entries of a level that can no
// longer change go to the card in windows, so the host holds a window per level instead of the stream (and never the parsed trie).
struct dsm_trie_stream {
    struct DevLevel {
        dsm::u64* freq = nullptr;
        dsm::u8* pl = nullptr;
        dsm::u32* fc = nullptr;
        size_t n = 0, cap = 0;
    };
    dsm::StreamParser sp;
    int device = 0;
    std::vector<DevLevel> dl;
    size_t WINDOW = 1u << 16;  // entries of a level collected on the host before they are uploaded (DSM_TRIE_WINDOW: tests use small ones)

    ~dsm_trie_stream() {
        (void)hipSetDevice(device);
        for (DevLevel& v : dl) {
            if (v.freq) (void)hipFree(v.freq);
            if (v.pl) (void)hipFree(v.pl);
            if (v.fc) (void)hipFree(v.fc);
        }
    }
    int grow(DevLevel& v, size_t need) {
        using namespace dsm;
        if (need <= v.cap) return 0;
        size_t cap = v.cap ? v.cap * 2 : WINDOW;
        if (cap < need) cap = need;
        u64* f = nullptr; u8* p = nullptr; u32* c = nullptr;
        hipError_t e = hipMalloc((void**)&f, cap * sizeof(u64));
        if (e == hipSuccess) e = hipMalloc((void**)&p, cap);
        if (e == hipSuccess) e = hipMalloc((void**)&c, cap * sizeof(u32));
        if (e == hipSuccess && v.n) {
            e = hipMemcpy(f, v.freq, v.n * sizeof(u64), hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(p, v.pl, v.n, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(c, v.fc, v.n * sizeof(u32), hipMemcpyDeviceToDevice);
        }
        if (e != hipSuccess) {
            if (f) (void)hipFree(f);
            if (p) (void)hipFree(p);
            if (c) (void)hipFree(c);
            return fail(DSM_E_NOMEM, std::string("dsm_trie_stream: ") + hipGetErrorString(e));
        }
        if (v.freq) (void)hipFree(v.freq);
        if (v.pl) (void)hipFree(v.pl);
        if (v.fc) (void)hipFree(v.fc);
        v.freq = f; v.pl = p; v.fc = c; v.cap = cap;
        return 0;
    }
    // upload what is final of every level (all of it at the end, whole windows otherwise)
    int flush(bool all) {
        using namespace dsm;
        if (dl.size() < sp.L.size()) dl.resize(sp.L.size());
        for (size_t l = 0; l < sp.L.size(); ++l) {
            const u64 fin = sp.final_count(l);
            const size_t k = (size_t)(fin - sp.base[l]);
            if (k == 0 || (!all && k < WINDOW)) continue;
            DevLevel& v = dl[l];
            if (int rc = grow(v, v.n + k)) return rc;
            const HostTrieLevel& h = sp.L[l];
            DSM_HIP(hipMemcpy(v.freq + v.n, h.freq.data(), k * sizeof(u64), hipMemcpyHostToDevice));
            DSM_HIP(hipMemcpy(v.pl + v.n, h.pl.data(), k, hipMemcpyHostToDevice));
            DSM_HIP(hipMemcpy(v.fc + v.n, h.fc.data(), k * sizeof(u32), hipMemcpyHostToDevice));
            v.n += k;
            sp.drop_front(l, k);
        }
        return 0;
    }
};

int dsm_trie_stream_begin(int device, dsm_trie_stream** out) {
    if (!out) return fail(DSM_E_INVAL, "dsm_trie_stream_begin: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_trie_stream_begin: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_trie_stream_begin: bad device ordinal");
    dsm_trie_stream* s = new dsm_trie_stream();
    s->device = device;
    if (const char* e = getenv("DSM_TRIE_WINDOW")) { const long w = atol(e); if (w > 0) s->WINDOW = (size_t)w; }
    *out = s;
    return DSM_OK;
}
int dsm_trie_stream_feed(dsm_trie_stream* s, const uint8_t* bytes, size_t n) {
    if (!s || (!bytes && n)) return fail(DSM_E_INVAL, "dsm_trie_stream_feed: null argument");
    if (int rc = s->sp.feed(bytes, n, false)) return rc;
    DSM_HIP(hipSetDevice(s->device));
    return s->flush(false);
}
void dsm_trie_stream_abort(dsm_trie_stream* s) { delete s; }
int dsm_trie_stream_end(dsm_trie_stream* s, dsm_trie** out) {
    if (!s || !out) { delete s; return fail(DSM_E_INVAL, "dsm_trie_stream_end: null argument"); }
    *out = nullptr;
    std::unique_ptr<dsm_trie_stream> guard(s);
    if (int rc = s->sp.feed(nullptr, 0, true)) return rc;
    DSM_HIP(hipSetDevice(s->device));
    if (int rc = s->flush(true)) return rc;
    std::unique_ptr<dsm_trie, void (*)(dsm_trie*)> t(new dsm_trie(), dsm_trie_free);
    t->device = s->device;
    t->nodes = s->sp.opened;
    t->maxfreq = s->sp.mf;
    u64 tot = 0;
    for (auto& v : s->dl) { t->level_off.push_back(tot); tot += v.n; }
    t->level_off.push_back(tot);
    // the levels move into the one allocation the merge reads (device to device; a level's buffers are released as soon as it has moved)
    DSM_HIP(hipMalloc((void**)&t->d_freq, tot * sizeof(u64)));
    DSM_HIP(hipMalloc((void**)&t->d_pl, tot));
    DSM_HIP(hipMalloc((void**)&t->d_fc, tot * sizeof(u32)));
    for (size_t l = 0; l < s->dl.size(); ++l) {
        auto& v = s->dl[l];
        const u64 o = t->level_off[l];
        if (v.n) {
            DSM_HIP(hipMemcpy(t->d_freq + o, v.freq, v.n * sizeof(u64), hipMemcpyDeviceToDevice));
            DSM_HIP(hipMemcpy(t->d_pl + o, v.pl, v.n, hipMemcpyDeviceToDevice));
            DSM_HIP(hipMemcpy(t->d_fc + o, v.fc, v.n * sizeof(u32), hipMemcpyDeviceToDevice));
        }
        if (v.freq) (void)hipFree(v.freq);
        if (v.pl) (void)hipFree(v.pl);
        if (v.fc) (void)hipFree(v.fc);
        v.freq = nullptr; v.pl = nullptr; v.fc = nullptr; v.n = v.cap = 0;
    }
    *out = t.release();
    return DSM_OK;
}

void dsm_trie_free(dsm_trie* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->d_freq) (void)hipFree(t->d_freq);
    if (t->d_pl) (void)hipFree(t->d_pl);
    if (t->d_fc) (void)hipFree(t->d_fc);
    delete t;
}
uint64_t dsm_trie_nodes(const dsm_trie* t) { return t ? t->nodes : 0; }

int dsm_merge(dsm_trie* const* tries, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (!tries || n <= 0 || !p) return fail(DSM_E_INVAL, "dsm_merge: bad arguments");
    bool wide = p->wide != 0;
    for (int k = 0; k < n; ++k) {
        if (!tries[k]) return fail(DSM_E_INVAL, "dsm_merge: null trie");
        if (tries[k]->maxfreq >= 0xFFFFFFF0ull) wide = true;
    }
    dsm_params q = *p;
    q.world_size = 1;  // one process holds every stream, like one metaserver
    q.rank = 0;
    q.fmin = 0;        // the clients already applied --fmin / --maxdepth
    q.maxdepth = ~0u;
    return merge_once(wide, tries, n, &q, sink, ctx, stats);
}

// ---------------------------------------------------------------------------------------------------------------------------
// dsm_server: one metaserver -- d connections whose streams are merged WHILE they arrive (metaserver.cpp:682-739 reads its
// sockets token by token inside traverse(), so it holds no stream and prints a node as soon as every client has closed it).
// A level-synchronous merge needs whole subtrees, so the unit here is the subtree of a node of depth prefix_len + 1: every client
// of one server enforces the same prefix (metaenumerate.cpp:268-309, EnumerateQuery.cpp:240-290), above that depth a stream is a
// single path, and a unit is final in a stream once the stream has closed it, gone past it, or ended.  As soon as that holds for
// every connection the unit's entries -- a range of every deeper level, the levels being in path order -- leave the streams'
// device windows as compact tries and are merged and printed (run_auto with the enforced path down to the unit and, as the
// reader-set order of its root, the one a shallow pass over the nodes above gives: it depends on earlier siblings only,
// metaserver.cpp:322-339) while the later units are still being received; the nodes of the enforced path, which close last, follow
// at the end.  The card holds the units in flight, not d complete tries.  prefix_len < 0: the classic way (merge after the last
// stream has ended; also what a stream that is not a single path above the unit depth would need -- such a stream is refused).
// ---------------------------------------------------------------------------------------------------------------------------
}  // extern "C"

extern "C" {
struct dsm_server {
    struct DevLevel {
        dsm::u64* freq = nullptr;
        dsm::u8* pl = nullptr;
        dsm::u32* fc = nullptr;
        size_t n = 0, cap = 0;
        size_t head = 0;    // entries at the front that left with a unit (their room is reclaimed when the window is next copied)
        dsm::u64 abs0 = 0;  // index, inside its level of the stream, of the first entry still held (= entry `head` of the arrays)
        void release() {
            if (freq) (void)hipFree(freq);
            if (pl) (void)hipFree(pl);
            if (fc) (void)hipFree(fc);
            freq = nullptr; pl = nullptr; fc = nullptr; n = cap = 0; head = 0;
        }
    };
    struct Stream {
        dsm::StreamParser sp;
        std::mutex mu;                 // the connection's reader thread (feed) against the merger (taking a unit out)
        std::vector<DevLevel> dl;      // levels below the unit depth: what has been uploaded and not yet left with a unit
        std::vector<dsm::u64> from;    // first entry, per level below the unit depth, of the next unit to leave
        size_t taken = 0;              // units of this stream that have left
        bool ended = false;
        dsm_trie_stream* classic = nullptr;  // prefix_len < 0: the whole stream, merged at the end
        dsm_trie* whole = nullptr;
    };
    int d = 0, device = 0;
    int K = -1;          // length of the enforced prefix; units are the nodes of depth U = K + 1
    dsm::u32 U = 0;
    size_t WINDOW = 1u << 16;
    dsm_params prm;
    dsm_tuple_sink sink = nullptr;
    void* ctx = nullptr;
    std::vector<std::unique_ptr<Stream>> s;
    std::mutex mu;
    std::condition_variable cv;
    std::thread merger;
    bool quit = false, done = false;
    int rc = 0;
    std::string err;
    dsm_stats stats;
    std::atomic<dsm::u64> units_merged{0}, peak_unit_nodes{0};  // (written by the merger thread, read by dsm_server_units)

    ~dsm_server() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (merger.joinable()) merger.join();
        (void)hipSetDevice(device);
        for (auto& st : s) {
            for (DevLevel& v : st->dl) v.release();
            if (st->classic) dsm_trie_stream_abort(st->classic);
            if (st->whole) dsm_trie_free(st->whole);
        }
        dsm::server_engines_destroy(engines);
    }
    // room for `need` live entries (v.n of them are held, behind v.head dead ones): a new allocation takes the live ones only
    int grow(DevLevel& v, size_t need) {
        using namespace dsm;
        if (v.head + need <= v.cap) return 0;
        size_t cap = v.cap ? v.cap * 2 : WINDOW;
        if (need <= v.cap / 2) cap = v.cap;  // (mostly dead entries: the same size will do)
        if (cap < need) cap = need;
        DevLevel w;
        hipError_t e = hipMalloc((void**)&w.freq, cap * sizeof(u64));
        if (e == hipSuccess) e = hipMalloc((void**)&w.pl, cap);
        if (e == hipSuccess) e = hipMalloc((void**)&w.fc, cap * sizeof(u32));
        if (e == hipSuccess && v.n) {
            e = hipMemcpy(w.freq, v.freq + v.head, v.n * sizeof(u64), hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(w.pl, v.pl + v.head, v.n, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(w.fc, v.fc + v.head, v.n * sizeof(u32), hipMemcpyDeviceToDevice);
        }
        if (e != hipSuccess) { w.release(); return fail(DSM_E_NOMEM, std::string("dsm_server: ") + hipGetErrorString(e)); }
        w.n = v.n; w.cap = cap; w.abs0 = v.abs0; w.head = 0;
        v.release();
        v = w;
        return 0;
    }
    // final entries of the levels below the unit depth go to the card (whole windows, or everything); caller holds st.mu
    int upload(Stream& st, bool all) {
        using namespace dsm;
        if (st.dl.size() < st.sp.L.size()) st.dl.resize(st.sp.L.size());
        for (size_t l = (size_t)U + 1; l < st.sp.L.size(); ++l) {
            const size_t k = (size_t)(st.sp.final_count(l) - st.sp.base[l]);
            if (k == 0 || (!all && k < WINDOW)) continue;
            DevLevel& v = st.dl[l];
            if (int r = grow(v, v.n + k)) return r;
            const HostTrieLevel& h = st.sp.L[l];
            DSM_HIP(hipMemcpy(v.freq + v.head + v.n, h.freq.data(), k * sizeof(u64), hipMemcpyHostToDevice));
            DSM_HIP(hipMemcpy(v.pl + v.head + v.n, h.pl.data(), k, hipMemcpyHostToDevice));
            DSM_HIP(hipMemcpy(v.fc + v.head + v.n, h.fc.data(), k * sizeof(u32), hipMemcpyHostToDevice));
            v.n += k;
            st.sp.drop_front(l, k);
        }
        return 0;
    }
    // a trie of the levels down to the unit depth as the stream has them now: the enforced path, the nodes between it and the units,
    // the unit roots.  Nodes that are still open carry a place holder for their frequency (nothing prints them).  Caller holds st.mu
    int hollow(Stream& st, dsm_trie** out) {
        using namespace dsm;
        std::vector<u64> f; std::vector<u8> pl; std::vector<u32> fc;
        std::unique_ptr<dsm_trie, void (*)(dsm_trie*)> t(new dsm_trie(), dsm_trie_free);
        t->device = device;
        const size_t nl = st.sp.L.size() < (size_t)U + 1 ? st.sp.L.size() : (size_t)U + 1;
        for (size_t l = 0; l < nl; ++l) {
            const HostTrieLevel& h = st.sp.L[l];   // (these levels are never taken away from the host: a handful of entries)
            if (h.freq.empty()) break;
            t->level_off.push_back(f.size());
            for (size_t i = 0; i < h.freq.size(); ++i) {
                const bool open = !st.sp.finished && l < st.sp.stack.size() && st.sp.stack[l] == st.sp.base[l] + i;
                f.push_back(l == 0 ? 0 : (open || h.freq[i] == 0 ? 1 : h.freq[i]));
                pl.push_back(h.pl[i]);
                fc.push_back(h.fc[i]);
                if (h.freq[i] > t->maxfreq) t->maxfreq = h.freq[i];
            }
        }
        t->level_off.push_back(f.size());
        t->nodes = f.size();
        DSM_HIP(hipMalloc((void**)&t->d_freq, f.size() * sizeof(u64)));
        DSM_HIP(hipMalloc((void**)&t->d_pl, f.size()));
        DSM_HIP(hipMalloc((void**)&t->d_fc, f.size() * sizeof(u32)));
        DSM_HIP(hipMemcpy(t->d_freq, f.data(), f.size() * sizeof(u64), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_pl, pl.data(), f.size(), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_fc, fc.data(), f.size() * sizeof(u32), hipMemcpyHostToDevice));
        *out = t.release();
        return 0;
    }
    // the same cut down to the single path towards `path` (no siblings): the top of a sample that lacks a unit
    int path_only(Stream& st, const std::vector<dsm::u8>& path, dsm_trie** out) {
        using namespace dsm;
        std::vector<u64> f; std::vector<u8> pl; std::vector<u32> fc;
        std::unique_ptr<dsm_trie, void (*)(dsm_trie*)> t(new dsm_trie(), dsm_trie_free);
        t->device = device;
        // follow the path as far as the stream has it
        size_t idx = 0;
        for (size_t l = 0; l < st.sp.L.size() && l <= (size_t)K + path.size(); ++l) {
            const HostTrieLevel& h = st.sp.L[l];
            if (h.freq.empty() || idx >= h.freq.size()) break;
            t->level_off.push_back(f.size());
            int next = -1;  // symbol of the next node of the path
            if (l < (size_t)K) next = l < st.sp.chain_sym.size() ? (int)st.sp.chain_sym[l] : -1;
            else if (l - (size_t)K + 1 < path.size()) next = (int)path[l - (size_t)K];   // (the unit itself is not part of it)
            const u32 kids = h.pl[idx] & 15u;
            const bool has = next >= 0 && ((kids >> next) & 1u);
            f.push_back(l ? 1 : 0); pl.push_back((u8)(has ? 1u << next : 0u)); fc.push_back(0);
            if (!has) break;
            idx = (size_t)h.fc[idx] + (size_t)__builtin_popcount(kids & ((1u << next) - 1u)) - (size_t)st.sp.base[l + 1];
        }
        if (t->level_off.empty()) { t->level_off.push_back(0); f.push_back(0); pl.push_back(0); fc.push_back(0); }
        t->level_off.push_back(f.size());
        t->nodes = f.size();
        DSM_HIP(hipMalloc((void**)&t->d_freq, f.size() * sizeof(u64)));
        DSM_HIP(hipMalloc((void**)&t->d_pl, f.size()));
        DSM_HIP(hipMalloc((void**)&t->d_fc, f.size() * sizeof(u32)));
        DSM_HIP(hipMemcpy(t->d_freq, f.data(), f.size() * sizeof(u64), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_pl, pl.data(), f.size(), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_fc, fc.data(), f.size() * sizeof(u32), hipMemcpyHostToDevice));
        *out = t.release();
        return 0;
    }
    // the unit of the stream's next event leaves its windows as a compact trie: path, root, and its range of every deeper level.  Caller holds st.mu
    int take_unit(Stream& st, const dsm::UnitEvent& ev, dsm_trie** out) {
        using namespace dsm;
        if (int r = upload(st, true)) return r;
        std::unique_ptr<dsm_trie, void (*)(dsm_trie*)> t(new dsm_trie(), dsm_trie_free);
        t->device = device;
        const size_t nl = ev.upto.size();  // levels below the unit depth that existed when the unit closed
        if (st.from.size() < nl) st.from.resize(nl, 0);
        u64 tot = (u64)U + 1;
        for (size_t k = 0; k < nl; ++k) {
            if (ev.upto[k] < st.from[k]) return fail(DSM_E_HIP, "dsm_server: unit marks out of order");
            tot += ev.upto[k] - st.from[k];
        }
        DSM_HIP(hipMalloc((void**)&t->d_freq, tot * sizeof(u64)));
        DSM_HIP(hipMalloc((void**)&t->d_pl, tot));
        DSM_HIP(hipMalloc((void**)&t->d_fc, tot * sizeof(u32)));
        const HostTrieLevel& hu = st.sp.L[U];
        const size_t ri = (size_t)(ev.index - st.sp.base[U]);
        std::vector<u64> f; std::vector<u8> pl; std::vector<u32> fc;
        for (u32 l = 0; l <= U; ++l) {
            t->level_off.push_back(l);
            if (l == U) { f.push_back(hu.freq[ri]); pl.push_back(hu.pl[ri]); fc.push_back(0); }
            else { f.push_back(l ? 1 : 0); pl.push_back((u8)(1u << (l < (u32)K ? st.sp.chain_sym[l] : ev.path[l - (u32)K]))); fc.push_back(0); }
        }
        DSM_HIP(hipMemcpy(t->d_freq, f.data(), f.size() * sizeof(u64), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_pl, pl.data(), f.size(), hipMemcpyHostToDevice));
        DSM_HIP(hipMemcpy(t->d_fc, fc.data(), f.size() * sizeof(u32), hipMemcpyHostToDevice));
        u64 o = (u64)U + 1;
        std::vector<size_t> reclaim;
        for (size_t k = 0; k < nl; ++k) {
            const size_t l = (size_t)U + 1 + k;
            const u64 a = st.from[k], b = ev.upto[k], cnt = b - a;
            if (cnt == 0) break;  // (a level without entries of this unit: none deeper either)
            t->level_off.push_back(o);
            DevLevel& v = st.dl[l];
            if (v.abs0 != a || v.abs0 + v.n < b) return fail(DSM_E_HIP, "dsm_server: a unit's entries are not in the device window");
            DSM_HIP(hipMemcpyAsync(t->d_freq + o, v.freq + v.head, cnt * sizeof(u64), hipMemcpyDeviceToDevice, 0));
            DSM_HIP(hipMemcpyAsync(t->d_pl + o, v.pl + v.head, cnt, hipMemcpyDeviceToDevice, 0));
            DSM_HIP(hipMemcpyAsync(t->d_fc + o, v.fc + v.head, cnt * sizeof(u32), hipMemcpyDeviceToDevice, 0));
            const u64 child_from = k + 1 < nl ? st.from[k + 1] : 0;  // entries of the next level count from the unit's first one
            if (child_from) hipLaunchKernelGGL(fc_rebase_kernel, grid_for(cnt), dim3(256), 0, 0, t->d_fc + o, (size_t)cnt, (u32)child_from);
            // the unit's entries stay where they are, dead: the window gives their room back when it is next copied (grow)
            v.head += (size_t)cnt;
            v.n -= (size_t)cnt;
            v.abs0 = b;
            o += cnt;
            if (v.n == 0 || v.head > 2 * v.n) reclaim.push_back(l);
        }
        DSM_HIP(hipDeviceSynchronize());
        for (size_t l : reclaim) {  // windows that are mostly dead now give the room back (one wait for all of them, above)
            DevLevel& v = st.dl[l];
            if (v.n == 0) { if (v.cap > 4 * WINDOW) v.release(); else v.head = 0; }
            else {
                DevLevel w;
                if (int r = grow(w, v.n > WINDOW ? v.n : WINDOW)) return r;
                DSM_HIP(hipMemcpy(w.freq, v.freq + v.head, v.n * sizeof(u64), hipMemcpyDeviceToDevice));
                DSM_HIP(hipMemcpy(w.pl, v.pl + v.head, v.n, hipMemcpyDeviceToDevice));
                DSM_HIP(hipMemcpy(w.fc, v.fc + v.head, v.n * sizeof(u32), hipMemcpyDeviceToDevice));
                w.n = v.n; w.abs0 = v.abs0;
                v.release();
                v = w;
            }
        }
        t->level_off.push_back(o);
        t->nodes = o;
        t->maxfreq = st.sp.mf;
        for (size_t k = 0; k < nl; ++k) st.from[k] = ev.upto[k];
        *out = t.release();
        return 0;
    }
    void add_stats(const dsm_stats& a, dsm::u64 nodes) {
        stats.tuples += a.tuples; stats.pairs += a.pairs; stats.candidates += a.candidates;
        stats.union_nodes += nodes;
        stats.levels += a.levels; stats.device_ms += a.device_ms; stats.host_ms += a.host_ms; stats.splits += a.splits;
        if (a.max_frontier > stats.max_frontier) stats.max_frontier = a.max_frontier;
        stats.pair_order_exact = a.pair_order_exact;
    }
    std::string text_of(const std::vector<dsm::u8>& p) const {
        std::string t;
        for (dsm::u8 c : p) t += "ACGT"[c];
        return t;
    }
    // Reader-set iteration order of a node below the enforced path (metaserver.cpp:322-339): it follows from its parent's order and
    // the reader sets of its EARLIER siblings, all final once every stream is past the node -- a shallow pass over the parent's
    // children with the parent's order as its seed, remembered (a node's order never changes afterwards).
    std::map<std::vector<dsm::u8>, std::vector<dsm::u16>> orders;
    dsm::ServerEngines* engines = dsm::server_engines_create();  // (released by the destructor)
    int order_of(const std::vector<dsm::u8>& path, dsm_trie* const* hol, bool wide, const std::string& chain, std::vector<dsm::u16>* out) {
        using namespace dsm;
        auto f = orders.find(path);
        if (f != orders.end()) { *out = f->second; return 0; }
        const std::vector<u8> parent(path.begin(), path.end() - 1);
        ServerOrder cap, seed;
        cap.depth = (u32)K + (u32)path.size();
        const ServerOrder* sp = nullptr;
        if (!parent.empty()) {
            std::vector<u16> po;
            if (int r = order_of(parent, hol, wide, chain, &po)) return r;
            seed.depth = (u32)K + (u32)parent.size();
            seed.ord.push_back(po);
            sp = &seed;
        }
        if (int r = server_run(wide, hol, d, prm, chain + text_of(parent), sink, ctx, false, 1, ~0u, cap.depth, sp, &cap, nullptr, engines, 1)) return r;
        size_t q = 0;
        while (q < cap.sym.size() && cap.sym[q] != (u32)path.back()) ++q;
        if (q == cap.sym.size()) return fail(DSM_E_HIP, "dsm_server: the shallow pass did not find the node " + chain + text_of(path));
        orders[path] = cap.ord[q];
        *out = cap.ord[q];
        return 0;
    }
    // One event -- a unit, or a node between the enforced path and the units -- that every stream is past.  A unit leaves the
    // windows, its root's reader-set order comes from a shallow pass, it is merged and printed; a node between is printed by a
    // run over the top of the streams that shows it its children and emits its depth only.
    int merge_event(const std::vector<dsm::u8>& path) {
        using namespace dsm;
        DSM_HIP(hipSetDevice(device));
        const bool is_unit = path.size() == (size_t)(U - (u32)K);
        std::vector<dsm_trie*> unit(d, nullptr), hol(d, nullptr);
        auto cleanup = [&] { for (auto* t : unit) if (t) dsm_trie_free(t); for (auto* t : hol) if (t) dsm_trie_free(t); };
        std::string chain;
        bool wide = prm.wide != 0;
        u64 nodes = 0;
        int r = 0;
        for (int k = 0; k < d && !r; ++k) {
            Stream& st = *s[k];
            std::lock_guard<std::mutex> lk(st.mu);
            if (st.sp.chain_sym.size() == (size_t)K) {
                const std::string mine = text_of(st.sp.chain_sym);
                if (chain.empty()) chain = mine;
                else if (chain != mine) r = fail(DSM_E_FORMAT, "dsm_server: the connections enforce different prefixes (" + chain + " / " + mine + ")");
            }
            if (r) break;
            if (st.sp.mf >= 0xFFFFFFF0ull) wide = true;
            const bool has = st.taken < st.sp.events.size() && st.sp.events[st.taken].path == path;
            r = hollow(st, &hol[k]);
            if (!r && is_unit) {
                r = has ? take_unit(st, st.sp.events[st.taken], &unit[k]) : path_only(st, path, &unit[k]);
                if (!r) nodes += unit[k]->nodes;
            }
            if (!r && has) ++st.taken;
        }
        if (r) { cleanup(); return r; }
        const std::string full = chain + text_of(path);
        const u32 depth = (u32)full.size();
        dsm_stats a;
        memset(&a, 0, sizeof a);
        ServerOrder seed;
        seed.depth = depth;
        seed.ord.emplace_back();
        r = order_of(path, hol.data(), wide, chain, &seed.ord[0]);
        if (!r && is_unit) {
            // (small units share one engine, whose buffers then stay; a large unit gets an engine of its size that goes with it)
            u64 biggest = 0;
            for (auto* t : unit) biggest = t->nodes > biggest ? t->nodes : biggest;
            r = server_run(wide, unit.data(), d, prm, full, sink, ctx, true, U, ~0u, ~0u, &seed, nullptr, &a, engines, biggest <= (1u << 25) ? 2 : 0);
            if (!r) { add_stats(a, a.union_nodes >= (u64)(U - 1) ? a.union_nodes - (u64)(U - 1) : 0); ++units_merged; if (nodes > peak_unit_nodes) peak_unit_nodes = nodes; }
        } else if (!r) {
            r = server_run(wide, hol.data(), d, prm, full, sink, ctx, true, depth, depth, depth + 1, &seed, nullptr, &a, engines, 1);
            if (!r) add_stats(a, 1);
        }
        cleanup();
        return r;
    }
    // the nodes of the enforced path, after every event and every stream's end
    int merge_path() {
        using namespace dsm;
        if (K < 1) return 0;
        DSM_HIP(hipSetDevice(device));
        std::vector<dsm_trie*> hol(d, nullptr);
        std::string chain;
        bool wide = prm.wide != 0;
        int r = 0;
        for (int k = 0; k < d && !r; ++k) {
            Stream& st = *s[k];
            std::lock_guard<std::mutex> lk(st.mu);
            if (st.sp.chain_sym.size() == (size_t)K && chain.empty()) chain = text_of(st.sp.chain_sym);
            if (st.sp.mf >= 0xFFFFFFF0ull) wide = true;
            r = hollow(st, &hol[k]);
        }
        dsm_stats a;
        memset(&a, 0, sizeof a);
        if (!r && !chain.empty()) {
            r = server_run(wide, hol.data(), d, prm, chain, sink, ctx, true, 1, (u32)K, (u32)K + 1, nullptr, nullptr, &a, engines, 1);
            if (!r) add_stats(a, (u64)K);
        }
        for (auto* t : hol) if (t) dsm_trie_free(t);
        return r;
    }
    void merger_main() {
        for (;;) {
            std::vector<dsm::u8> ev;
            bool have = false, all_ended = true;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (quit || rc) return;
                    // the first event, in the post-order of the union trie, among the streams' next ones -- ready when no stream can still
                    // produce it or one before it (pick_event, stream_parse.h); every stream is held while they are looked at together
                    {
                        std::vector<std::unique_lock<std::mutex>> held;
                        std::vector<const dsm::StreamParser*> sps;
                        std::vector<size_t> taken;
                        std::vector<bool> ended;
                        all_ended = true;
                        for (int k = 0; k < d; ++k) {
                            held.emplace_back(s[k]->mu);
                            sps.push_back(&s[k]->sp); taken.push_back(s[k]->taken); ended.push_back(s[k]->ended);
                            all_ended = all_ended && s[k]->ended;
                        }
                        const dsm::EventPick pk = dsm::pick_event(sps, taken, ended);
                        have = pk.have;
                        if (pk.have && pk.ready) { ev = pk.path; break; }
                        if (!pk.have && all_ended) break;
                    }
                    cv.wait(lk);
                }
            }
            const int r = have ? merge_event(ev) : merge_path();
            std::lock_guard<std::mutex> lk(mu);
            if (r) { rc = r; err = dsm_last_error(); cv.notify_all(); return; }
            if (!have) { done = true; cv.notify_all(); return; }
        }
    }
};

int dsm_server_create(int nsamples, int device, int prefix_len, int unit_extra, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_server** out) {
    if (!out || !p || nsamples <= 0 || !sink) return fail(DSM_E_INVAL, "dsm_server_create: bad arguments");
    *out = nullptr;
    if (nsamples > 273) return fail(DSM_E_INVAL, "too many samples (MAX_READERS 273, metaserver.cpp:19)");
    if (prefix_len > 32 || unit_extra < 0 || unit_extra > 4) return fail(DSM_E_INVAL, "dsm_server_create: prefix_len > 32 or unit_extra outside 0..4");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_server_create: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_server_create: bad device ordinal");
    std::unique_ptr<dsm_server> sv(new dsm_server());
    sv->d = nsamples; sv->device = device; sv->K = prefix_len; sv->U = prefix_len >= 0 ? (u32)(prefix_len + 1 + unit_extra) : 0;
    sv->prm = *p;
    sv->prm.world_size = 1; sv->prm.rank = 0; sv->prm.fmin = 0; sv->prm.maxdepth = ~0u; sv->prm.prefix = "";  // as dsm_merge
    sv->sink = sink; sv->ctx = ctx;
    memset(&sv->stats, 0, sizeof sv->stats);
    if (const char* e = getenv("DSM_TRIE_WINDOW")) { const long w = atol(e); if (w > 0) sv->WINDOW = (size_t)w; }
    for (int k = 0; k < nsamples; ++k) {
        sv->s.emplace_back(new dsm_server::Stream());
        if (prefix_len >= 0) {
            dsm::StreamParser& sp = sv->s.back()->sp;
            sp.unit_depth = sv->U; sp.chain_len = (u32)prefix_len;
            if (prefix_len == 0) sp.last_closed.assign(1, -1);
        }
        else if (int rc = dsm_trie_stream_begin(device, &sv->s.back()->classic)) return rc;
    }
    if (prefix_len >= 0) { dsm_server* raw = sv.get(); sv->merger = std::thread([raw] { raw->merger_main(); }); }
    *out = sv.release();
    return DSM_OK;
}
int dsm_server_feed(dsm_server* sv, int sample, const uint8_t* bytes, size_t n) {
    if (!sv || sample < 0 || sample >= sv->d || (!bytes && n)) return fail(DSM_E_INVAL, "dsm_server_feed: bad arguments");
    dsm_server::Stream& st = *sv->s[sample];
    if (st.classic) return dsm_trie_stream_feed(st.classic, bytes, n);
    {
        std::lock_guard<std::mutex> lk(sv->mu);
        if (sv->rc) return fail(sv->rc, sv->err);
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(st.mu);
        if (st.ended) return fail(DSM_E_INVAL, "dsm_server_feed: the connection has ended");
        rc = st.sp.feed(bytes, n, false);
        if (!rc) { DSM_HIP(hipSetDevice(sv->device)); rc = sv->upload(st, false); }
    }
    {   // (the merger is either before its look at the streams or already waiting.)  A stream that cannot be parsed fails the whole
        // server: the merger stops, the other connections' feeds and finish() return this error instead of waiting for an end that
        // will not come.
        std::lock_guard<std::mutex> lk(sv->mu);
        if (rc && !sv->rc) { sv->rc = rc; sv->err = dsm_last_error(); }
    }
    sv->cv.notify_all();
    return rc;
}
int dsm_server_end(dsm_server* sv, int sample) {
    if (!sv || sample < 0 || sample >= sv->d) return fail(DSM_E_INVAL, "dsm_server_end: bad arguments");
    dsm_server::Stream& st = *sv->s[sample];
    if (st.classic) {
        dsm_trie_stream* ts = st.classic;
        st.classic = nullptr;
        const int rc = dsm_trie_stream_end(ts, &st.whole);
        if (!rc) st.ended = true;
        return rc;
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(st.mu);
        if (st.ended) return DSM_OK;
        rc = st.sp.feed(nullptr, 0, true);
        if (!rc) st.ended = true;
    }
    { std::lock_guard<std::mutex> lk(sv->mu); }
    sv->cv.notify_all();
    return rc;
}
int dsm_server_finish(dsm_server* sv, dsm_stats* stats) {
    if (!sv) return fail(DSM_E_INVAL, "dsm_server_finish: null server");
    for (auto& st : sv->s)
        if (!st->ended) return fail(DSM_E_INVAL, "dsm_server_finish: a connection has not ended");
    if (sv->K < 0) {
        std::vector<dsm_trie*> tr;
        for (auto& st : sv->s) tr.push_back(st->whole);
        const int rc = dsm_merge(tr.data(), sv->d, &sv->prm, sv->sink, sv->ctx, &sv->stats);
        if (stats) *stats = sv->stats;
        return rc;
    }
    std::unique_lock<std::mutex> lk(sv->mu);
    sv->cv.wait(lk, [&] { return sv->done || sv->rc; });
    if (sv->rc) return fail(sv->rc, sv->err);
    if (stats) *stats = sv->stats;
    return DSM_OK;
}
uint64_t dsm_server_units(const dsm_server* sv, uint64_t* peak_unit_nodes) {
    if (!sv) return 0;
    if (peak_unit_nodes) *peak_unit_nodes = sv->peak_unit_nodes.load();
    return sv->units_merged.load();
}
void dsm_server_destroy(dsm_server* sv) { delete sv; }


}  // extern "C"
