// engine.hip -- frontier engine: FM-index backward-search enumeration + cross-sample merge (gfx950).
//
// The reference walks each sample's suffix trie depth first, one LF call at a time
// (EnumerateQuery::nextSymbol, EnumerateQuery.cpp:151-238) and merges the per-sample streams in a
// single-threaded server (metaserver.cpp:269-486).  Here the trie of one prefix is expanded level by
// level.  Per level:
//   expand_kernel   one thread per frontier node: the four child intervals, the left-extension intervals
//                   of every child, the fmin test and the left-char code, fused (the LF-step kernel).
//                   Child records go to a compact buffer (block-wise allocation), the [4F] frequency
//                   column and left-char codes go to the exchange buffer.
//   (exchange)      one all-gather of the columns across ranks (RCCL through the host's callback).
//   advance_*       union frontier of the next level: alive flags from the exchanged columns, one scan,
//                   node links / reader counts / merged left chars / record handles in the down-sweep.
//   order_kernel    iteration order of the reference's reader sets (only when d > 1).
//   filter_kernel   metaserver's output predicates for the nodes whose children are now known.
// The DFS order of the reference's output (post-order tuples, nested wire stream) is recovered at the end
// of a prefix from subtree aggregates over the retained levels (prefix sums, no sorting).
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <type_traits>

#include "common.h"
#include "scan.h"
#include "setorder.h"
#include "stream_parse.h"
#include "lfstep.h"
#include "entropy_tables.h"
#include "textemit.h"
#include "engine_api.h"
#include "emit_runs.h"

namespace dsm {

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}


// Children directory of a level ("kids"): per 64 nodes (one wave) four bit planes -- bit j of plane c: node 64w + j has a child
// with symbol c in the union trie -- and four counts cum[c] = index, in the next level, of the first such child of the wave.
//   child c of node v = cum[c] + popcount(plane[c] & below(v & 63))
// 48 bytes per 64 nodes; the lanes of a wave read the same words.
struct Kids {
    const u64* plane;  // [4 * waves]
    const u32* cum;    // [4 * waves]
};
__device__ __forceinline__ u32 kid_mask(const Kids& k, u32 v) {
    const u64* p = k.plane + (size_t)(v >> 6) * 4;
    const u32 s = v & 63;
    return (u32)((p[0] >> s) & 1) | ((u32)((p[1] >> s) & 1) << 1) | ((u32)((p[2] >> s) & 1) << 2) | ((u32)((p[3] >> s) & 1) << 3);
}
__device__ __forceinline__ u32 kid_index(const Kids& k, u32 v, u32 c) {
    const size_t e = (size_t)(v >> 6) * 4 + c;
    return k.cum[e] + (u32)__popcll(k.plane[e] & ((1ull << (v & 63)) - 1));
}

// The directory words of a wave's 64 nodes, fetched once per wave (w is wave-uniform: scalar loads) instead of once per lane.
struct KidWave {
    u64 p[4];
    u32 c[4];
};
__device__ __forceinline__ void kid_wave(const Kids& k, u32 w, KidWave& o) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { o.p[c] = k.plane[(size_t)w * 4 + c]; o.c[c] = k.cum[(size_t)w * 4 + c]; }
}
__device__ __forceinline__ u32 bits_below_lane(u64 p) {  // set bits of p below this lane's position
    return __builtin_amdgcn_mbcnt_hi((u32)(p >> 32), __builtin_amdgcn_mbcnt_lo((u32)p, 0u));
}

// Workgroups are dealt round-robin to the eight XCDs, each with its own L2.  The sweeps below write dense arrays in runs that are
// shorter than a cache line and not aligned to one, so neighbouring workgroups finish each other's lines: block b takes the place
// xcd_block() in the sweep, which gives every XCD one contiguous range of the level -- neighbouring runs meet in ONE L2 and leave
// it as whole lines (otherwise two L2s each write back a partial line).
constexpr u32 NUM_XCD = 8;
__device__ __forceinline__ u32 xcd_block() {
    const u32 b = blockIdx.x, G = gridDim.x;
    const u32 x = b % NUM_XCD, i = b / NUM_XCD, per = G / NUM_XCD, rem = G % NUM_XCD;
    return x * per + (x < rem ? x : rem) + i;
}

// Thread-per-node kernels do little per node; a thread takes NPT nodes a grid-width apart (coalescing is kept) and issues
// all their loads before using any, so a wave has several lines in flight and the grid is NPT times smaller.
constexpr int NPT = 4;
static inline dim3 grid_npt(u64 n) { return dim3((unsigned)((n + 256ull * NPT - 1) / (256ull * NPT))); }


// DSM_TIMELINE=1: host time stamps of a mining call on stderr (milliseconds since the first stamp), a debugging aid
static inline void timeline(const char* what, const char* arg = "") {
    static const bool on = getenv("DSM_TIMELINE") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "dsm timeline %9.3f ms  %s %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what, arg);
}

constexpr u32 BC_OK = 0x4f4b4f4bu, BC_CAPACITY = 0x46554c4cu, BC_ABORT = 0x41424f52u;  // first word of the owner's broadcast: go on / split this prefix / the owner failed
constexpr u32 XHDR = 16;  // every rank's message starts with the largest child frequency it saw (u64) and 8 spare bytes
// Several engines of one process (prefix lanes on their own streams) share the device.  Their expand launches are chained:
// each waits for the previously issued expand of the process on that device, so two LF-step kernels never run side by
// side (their per-launch durations stay meaningful) while everything else of one lane overlaps the other lane's expand.
struct ExpandChain {
    struct Link { hipEvent_t last = nullptr; const void* owner = nullptr; };  // completion of the most recently issued expand launch
    std::mutex mu;
    std::map<int, Link> dev;           // per device ordinal
};
static ExpandChain g_expand_chain;

// view of one exchange buffer: rank-major; inside a rank [nlocal][F] P (frequency of node v in the sample, 0 = absent)
// then [nlocal][F] u8 (bits 0-3: surviving children of v in the sample, bits 4-6: left-char code of v)
struct Xchg {
    const u8* base;
    u64 bpr;      // bytes per rank
    u32 nlocal;
    u32 d;        // total samples = world * nlocal
    u64 F;        // nodes of the level
    u32 fb;       // bytes per frequency entry: 2 when every frequency of the level is below 65535, else sizeof(P); 1 = packed: one
                  // 16-bit word per node and nothing else (frequency below 512 in bits 0-8, the flag byte in bits 9-15)
    u32 nm;       // packed levels of several local samples: node-major inside a rank -- the words of node v in the rank's samples are
                  // adjacent ([F][nlocal] instead of [nlocal][F]): whoever looks at a node across the samples (reduce, predicates,
                  // reader-set orders, candidate store) reads one 2 * nlocal-byte piece instead of nlocal strided 2-byte entries
};
// place of the packed word of node v in local sample l of a rank
__device__ __forceinline__ u64 x_word_index(const Xchg& x, u32 l, u64 v) { return x.nm ? v * x.nlocal + l : (u64)l * x.F + v; }
__device__ __forceinline__ void x_split(const Xchg& x, u32 g, u32& r, u32& l) {
    if (x.nlocal == 1) { r = g; l = 0; }          // one sample per rank (multi-GPU runs)
    else if (x.d == x.nlocal) { r = 0; l = g; }   // single process
    else { r = g / x.nlocal; l = g % x.nlocal; }
}
template <typename P>
__device__ __forceinline__ P x_freq(const Xchg& x, u32 g, u64 v) {
    u32 r, l;
    x_split(x, g, r, l);
    const u8* rb = x.base + (u64)r * x.bpr + XHDR;
    if (x.fb == 1) return (P)(reinterpret_cast<const u16*>(rb)[x_word_index(x, l, v)] & (PACK_FMAX - 1));
    if (x.fb == 2) return (P)reinterpret_cast<const u16*>(rb)[(u64)l * x.F + v];
    return reinterpret_cast<const P*>(rb)[(u64)l * x.F + v];
}
template <typename P>
__device__ __forceinline__ u32 x_pl(const Xchg& x, u32 g, u64 v) {
    u32 r, l;
    x_split(x, g, r, l);
    if (x.fb == 1) return (u32)reinterpret_cast<const u16*>(x.base + (u64)r * x.bpr + XHDR)[x_word_index(x, l, v)] >> 9;
    const u8* rb = x.base + (u64)r * x.bpr + XHDR + (u64)x.nlocal * x.F * x.fb;
    return rb[(u64)l * x.F + v];
}

// eight local samples of a single process on a packed node-major level: the node's eight words as one 16-byte piece
__device__ __forceinline__ bool x_is_nm8(const Xchg& x) { return x.nm && x.fb == 1 && x.nlocal == 8 && x.d == 8; }
__device__ __forceinline__ void x_words8(const Xchg& x, u64 v, u32 w[8]) {
    const uint4 q = reinterpret_cast<const uint4*>(x.base + XHDR)[v];
    w[0] = q.x & 0xFFFFu; w[1] = q.x >> 16; w[2] = q.y & 0xFFFFu; w[3] = q.y >> 16;
    w[4] = q.z & 0xFFFFu; w[5] = q.z >> 16; w[6] = q.w & 0xFFFFu; w[7] = q.w >> 16;
}

// ---------------------------------------------------------------------------------------------
// advance: union frontier of the next level (colex order = stable partition of the children by symbol).
//   reduce (more than one sample): per parent, how many samples keep each child -> union bit planes, tile counts
//   scan over the [4][tiles] counts: entry (c, t) = index of the first child with symbol c of tile t
//   down: node links, reader counts, record handles of the new level, the per-wave directory counts
// ---------------------------------------------------------------------------------------------
constexpr int MAX_LOCAL = 273;  // local samples per process (MAX_READERS, metaserver.cpp:19)

// how many samples keep each child of parent u (0 = the slot is not a union node)
template <typename P>
__device__ __forceinline__ void parent_eval(const Xchg& x, u32 u, u32 nT4[4]) {
    const u32 world = x.d / x.nlocal;
#pragma unroll
    for (int c = 0; c < 4; ++c) nT4[c] = 0;
    for (u32 r = 0; r < world; ++r) {
        if (x.fb == 1) {  // packed columns: the children nibble sits in bits 9-12 of the node's word
            const u16* pw = reinterpret_cast<const u16*>(x.base + (u64)r * x.bpr + XHDR);
            if (x.nm && x.nlocal == 8) {  // eight samples per rank: the node's words are one 16-byte piece
                const uint4 q = reinterpret_cast<const uint4*>(pw)[u];
                const u32 w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32 m0 = (w4[k] >> 9) & 15u, m1 = w4[k] >> 25;
#pragma unroll
                    for (int c = 0; c < 4; ++c) nT4[c] += ((m0 >> c) & 1u) + ((m1 >> c) & 1u);
                }
                continue;
            }
            for (u32 l = 0; l < x.nlocal; ++l) {
                const u32 m = (u32)pw[x_word_index(x, l, u)] >> 9;
#pragma unroll
                for (int c = 0; c < 4; ++c) nT4[c] += (m >> c) & 1u;
            }
            continue;
        }
        const u8* pb = x.base + (u64)r * x.bpr + XHDR + (u64)x.nlocal * x.F * x.fb;
        for (u32 l = 0; l < x.nlocal; ++l) {
            const u32 m = pb[(u64)l * x.F + u];
#pragma unroll
            for (int c = 0; c < 4; ++c) nT4[c] += (m >> c) & 1u;
        }
    }
}

// sinfo[4u + c] = number of samples that keep child c of parent u, so that the down-sweep does not re-read d columns
template <typename P>
__global__ __launch_bounds__(256) void advance_reduce_kernel(Xchg x, u16* __restrict__ sinfo, u64* __restrict__ kplane, u32* __restrict__ cnt4, u32 nbp) {
    __shared__ u32 wtot[4][4];
    const u32 u = blockIdx.x * TILE + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 nT4[4] = {0, 0, 0, 0};
    if (u < x.F) {
        parent_eval<P>(x, u, nT4);
        uint2 q;
        q.x = nT4[0] | (nT4[1] << 16);
        q.y = nT4[2] | (nT4[3] << 16);
        *reinterpret_cast<uint2*>(sinfo + (size_t)u * 4) = q;
    }
    u64 bal[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bal[c] = __ballot(nT4[c] != 0);
    if (lane < 4) {
        const u64 b = DSM_PICK(bal, lane);
        wtot[w][lane] = (u32)__popcll(b);
        kplane[((size_t)blockIdx.x * 4 + w) * 4 + lane] = b;
    }
    __syncthreads();
    if (threadIdx.x < 4) cnt4[(size_t)threadIdx.x * nbp + blockIdx.x] = wtot[0][threadIdx.x] + wtot[1][threadIdx.x] + wtot[2][threadIdx.x] + wtot[3][threadIdx.x];
}

struct FilterArgs {
    u32 F;
    u32 depth;
    u32 d;
    u32 pmin, pmax, mindepth;
    double emin, emax;
    u32 exact_order;  // 0: d == 1; 1: nibble-packed order[] (d <= 13); 2: u16 order arrays (d > 13)
};

// output predicates of metaserver.cpp:406-419; the entropy test is decided here only when it is not
// within ENT_MARGIN of a threshold -- everything kept is re-tested on the host with glibc's log (bit-exact).
// The device value uses the hardware log2 (v_log_f32, 1 ulp: at most 2^-18 absolute for arguments below 2^64), so it
// is off by less than 1e-5; the margin leaves a factor of ten.
constexpr double ENT_MARGIN = 1e-4;

// the output predicates of one node (metaserver.cpp:406-419) given its number of readers t, of children nc and whether its single
// child carries every reader; merged left char (metaserver.cpp:383-387) and entropy over the samples that hold the node
template <typename P>
__device__ __forceinline__ bool filter_node(const FilterArgs& a, const Xchg& x, u32 v, u32 t, u32 nc, u32 same) {
    if (a.depth < a.mindepth) return false;
    if (a.pmax != 0 && t > a.pmax) return false;
    if (t < a.pmin) return false;
    if (nc == 1 && same) return false;
    u64 sumN = a.d;
    double s = 0;
    u32 l = 0xFF;
    if (x_is_nm8(x)) {
        u32 w[8];
        x_words8(x, v, w);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u64 f = w[g] & (PACK_FMAX - 1);
            if (f) {
                const u32 lg = w[g] >> 13;
                l = l == 0xFF ? lg : (l == lg ? l : 5u);
                sumN += f;
                if (a.emax > 0) s += (double)(f + 1) * (double)__log2f((float)(f + 1));
            }
        }
    } else {
        for (u32 g = 0; g < a.d; ++g) {
            u64 f = (u64)x_freq<P>(x, g, v);
            if (f) {
                u32 lg = x_pl<P>(x, g, v) >> 4;
                l = l == 0xFF ? lg : (l == lg ? l : 5u);
                sumN += f;
                if (a.emax > 0) s += (double)(f + 1) * (double)__log2f((float)(f + 1));
            }
        }
    }
    if (l >= 1 && l <= 4) return false;
    if (a.emax > 0) {
        double e = (double)__log2f((float)sumN) - s / (double)sumN;
        if (e < a.emin - ENT_MARGIN || e > a.emax + ENT_MARGIN) return false;
    }
    return true;
}

// Path words (mining): every node keeps the last (up to 16) symbols of its path, two bits each, and the index of its ancestor at the
// level where that chunk starts -- level 16 * ((l - 1) / 16) for a node of level l.  A child's word follows from its parent's in the
// advance sweep (the lane of parent u reads word u, coalesced), and the path of a candidate of level l is put together from
// (l + 15) / 16 words instead of l parent links: one random 64-byte line per 16 symbols instead of one per symbol.
constexpr u32 PW_CHUNK = 16;
constexpr u32 SREC_FIRST = 4u, SREC_LAST = 8u;  // flag bits of a stream record's second word (see keep_kernel)
__device__ __forceinline__ uint2 child_path_word(const uint2 parent, u32 u, u32 plevel, u32 c) {
    const u32 r = plevel & (PW_CHUNK - 1);
    return r == 0 ? make_uint2(u, c) : make_uint2(parent.x, parent.y | (c << (2 * r)));
}

struct AdvanceOut {
    u32* slot;        // retained: 4*parent + sym of every new node (null: nobody reads the links -- mining without derived handles)
    uint2* sa;        // stream mode: per new node its parent index and symbol / first / last flags (keep_kernel completes the record)
    uint2* pw;        // retained path words of the new nodes (null: stream mode); pw_after_slot: they follow the slot array, whose length
    const uint2* parent_pw;  // (the new level's width) only the device knows when this sweep runs -- *width, or the single tile's total
    const u32* width;
    u32 pw_after_slot;
    u32 plevel;       // level of the parents
    u32 kshift;       // log2 of the words per wave in kplane: 2, or 3 when the expand kernel of a single sample wrote whole lines
    u32 cand_copy;    // ... whose words 4 and 5 are the wave's candidate bits and counts: copied to candbits / wsum here
    u16* nT;          // per new node
    u8* samechild;    // per parent: single child that carries every reader (metaserver.cpp:416-417)
    const u16* parent_nT;
    const u64* kplane;   // union planes of the parents' level (from the reduce pass, or the sample's own planes when d == 1)
    u64* kplane_w;       // where the level's retained directory keeps them (null: kplane already is that array)
    u32* kcum;           // retained directory counts, per wave (written here)
    const u32* cnt4;     // scanned tile offsets [4][nbp] (unused by a single-tile level)
    u32* cnt_clear;      // single sample: the raw tile counts the expand kernel accumulated (cleared here for the next level)
    u32 nbp;
    const u16* sinfo;    // per-slot sample counts from the reduce pass (null: d == 1, or the single tile evaluates the columns itself)
    u32 eval;            // single tile, more than one sample: evaluate the union from the exchanged columns here
    u32 single;          // one sample in all (index mode): no reader counts, its planes are the union's planes
    // per local sample record handles
    u32* const* rp;             // device table of nlocal pointers: handles of the new level
    u32* rp0;                   // the same for the single sample of a one-sample run (no table to read first)
    const u64* const* splane;   // per local sample: the planes its expand kernel wrote (index mode)
    u32 rp_index;               // index mode, several samples: 1 = write the samples' handle tables (0: the expand kernels derive them)
    const u32* const* tpos;     // trie mode: handle of the first allowed child in the parsed stream
    u32 nlocal, rank;
    u32 seg;           // handles per symbol segment of the record buffers
    u32 cap;           // entries of the new level's arrays: a wider level is reported through the total, not written
    u32* h_totals;       // single-tile levels: [0] = nodes of the new level (larger levels: the grand total of the scan)
    // the output predicates of this level's nodes ride along in the wave sweep (its lanes hold the node's child and reader counts)
    u32 filter_on;
    FilterArgs fa;
    u64* candbits;       // per wave of 64 nodes: which of them are candidates
    u64* wsum;           // per wave: candidates | pairs << 32
    // one sample, advance_single_kernel: the candidates' records {node, 0, frequency} are stored by the sweep itself -- their places come
    // from the fifth row of the tile counts (candidates per tile, scanned along with the child counts)
    uint4* crec;         // null: the candidate store is a kernel of its own (cand_store_kernel)
    u32 crec_cap;        // records the block holds (a level with more is stored the old way)
};

// The few words the host reads after a level travel as ONE 16-byte store to pinned host memory: {sequence number, width of
// the new level | wide-frequency flag << 31, candidates, pairs}.  A naturally aligned 16-byte store is one write on the bus, so
// the host that sees the sequence number sees the rest -- no system-scope fence whose completion the kernel would wait for.
// (A large kernel that wrote to host memory itself would end with a system-scope release of everything it left dirty in L2.)
struct PublishArgs {
    const u32* total;     // width of the new level
    const u64* cand;      // candidates | pairs << 32 of the level just filtered (null: not filtered)
    const u8* cmax_base;  // exchange buffer of the level: rank r's message starts with the largest child frequency it saw
    u64 cmax_bpr;
    u32 cmax_world;
    u32* clear;           // header of the message the next level's expand kernels will fill (4 words), may be null
    u32* next;            // device copy of the new level's width and frequency class for an expand launch queued ahead (may be null)
    uint4* packet;        // pinned, 16-byte aligned
    u32 seq;
    uint4* bc_header;     // owner mode: the same packet at the head of the broadcast message (device), may be null
    const u32* grand;     // (candidate counts scanned behind the child counts) grand total of that scan: candidates = *grand - *total, one pair each
};
__global__ void publish_kernel(PublishArgs a) {
    __shared__ u32 wide;  // bit 0: some child frequency of the level is 65535 or more, bit 1: 512 or more
    if (threadIdx.x == 0) wide = 0;
    __syncthreads();
    for (u32 r = threadIdx.x; r < a.cmax_world; r += blockDim.x) {
        const u64 m = *reinterpret_cast<const u64*>(a.cmax_base + (u64)r * a.cmax_bpr);
        if (m >= PACK_FMAX) atomicOr(&wide, m >= 65535 ? 3u : 2u);
    }
    const u32 tot = threadIdx.x == 0 ? *a.total : 0u;
    u64 cand = threadIdx.x == 0 && a.cand ? *a.cand : 0ull;
    if (threadIdx.x == 0 && a.grand) { const u64 c = (u64)(*a.grand - tot); cand = c | (c << 32); }
    __syncthreads();
    if (a.clear && threadIdx.x < 4) a.clear[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        if (a.next) { a.next[0] = tot; a.next[1] = wide; }
        const uint4 pk = make_uint4(a.seq, tot | ((wide & 1u) << 31) | ((wide >> 1) << 30), (u32)cand, (u32)(cand >> 32));
        if (a.bc_header) *a.bc_header = make_uint4(BC_OK, pk.y, 0u, 0u);  // what the clients need: width and frequency class of the new level
        *a.packet = pk;  // one global_store_dwordx4
    }
}

// Down-sweep, one thread per parent, one tile of 256 parents per block (the tiles of the expand kernel).  A child's index in
// the new level is a prefix count over its symbol's plane, so the lanes that write children of one symbol write neighbouring
// entries.
template <typename P>
__global__ __launch_bounds__(256) void advance_down_kernel(Xchg x, AdvanceOut o) {
    __shared__ u32 wcnt[4][4];
    const u32 F = (u32)x.F;
    const u32 u = blockIdx.x * TILE + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t wv = (size_t)blockIdx.x * 4 + w;  // wave index inside the level
    const u64 lt = (1ull << lane) - 1;
    u32 nT4[4] = {0, 0, 0, 0};
    u64 up[4];
    if (o.eval) {
        if (u < F) parent_eval<P>(x, u, nT4);
#pragma unroll
        for (int c = 0; c < 4; ++c) up[c] = __ballot(nT4[c] != 0);
    } else {
        const bool inside = wv * 64 < F;  // the expand kernel writes planes for the waves that hold nodes only
#pragma unroll
        for (int c = 0; c < 4; ++c) up[c] = inside ? o.kplane[(wv << o.kshift) + c] : 0ull;
        if (o.sinfo) {
            if (u < F) {
                const uint2 q = *reinterpret_cast<const uint2*>(o.sinfo + (size_t)u * 4);
                nT4[0] = q.x & 0xFFFFu; nT4[1] = q.x >> 16; nT4[2] = q.y & 0xFFFFu; nT4[3] = q.y >> 16;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) nT4[c] = (u32)((up[c] >> lane) & 1);
        }
    }
    if (lane < 4) {
        const u64 b = DSM_PICK(up, lane);
        wcnt[w][lane] = (u32)__popcll(b);
        if (o.kplane_w) o.kplane_w[wv * 4 + lane] = b;
    }
    __syncthreads();
    u32 cum[4];  // index of the first child with symbol c of this wave
    u32 total = 0;
    {
        u32 base = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32 before = 0, all = 0;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) { const u32 t = wcnt[q][c]; before += q < w ? t : 0u; all += t; }
            cum[c] = (gridDim.x == 1 ? base : o.cnt4[(size_t)c * o.nbp + blockIdx.x]) + before;
            base += all;
        }
        total = base;
    }
    if (lane < 4) o.kcum[wv * 4 + lane] = DSM_PICK(cum, lane);
    if (o.cnt_clear && threadIdx.x < 4) o.cnt_clear[(size_t)threadIdx.x * o.nbp + blockIdx.x] = 0;  // the expand kernels of the next level add into it
    uint2* pw = o.pw;
    if (o.pw_after_slot) pw = reinterpret_cast<uint2*>(reinterpret_cast<u8*>(o.slot) + (((size_t)total * 4 + 255) & ~(size_t)255));
    if (u < F) {
        u32 pres = 0, lastT = 0, vj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vj[c] = cum[c] + (u32)__popcll(up[c] & lt);
            if (nT4[c]) { pres |= 1u << c; lastT = nT4[c]; }
        }
        const u32 nc = __popc(pres);
        if (!o.single) o.samechild[u] = (nc == 1 && lastT == o.parent_nT[u]) ? 1 : 0;  // (a single sample: always 1 reader, nobody reads these)
        uint2 mypw = make_uint2(0u, 0u);
        if (pw) mypw = o.parent_pw[u];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (((pres >> c) & 1u) && vj[c] < o.cap) {  // a level wider than its arrays is reported through the total, not written
                if (o.slot) o.slot[vj[c]] = 4u * u + (u32)c;
                if (pw) pw[vj[c]] = child_path_word(mypw, u, o.plevel, (u32)c);
                if (o.sa) o.sa[vj[c]] = make_uint2(u, (u32)c | ((pres & ((1u << c) - 1u)) ? 0u : SREC_FIRST) | ((pres >> (c + 1)) ? 0u : SREC_LAST));
                if (!o.single) o.nT[vj[c]] = (u16)nT4[c];
            }
        }
        for (u32 sl = 0; sl < o.nlocal && pres && (o.tpos || o.single || o.rp_index); ++sl) {  // record handles of this parent's children in every local sample
            u32* rp = o.rp[sl];
            if (o.tpos) {  // parsed stream: the children of a node are consecutive in the stream's next level
                const u32 m = x_pl<P>(x, o.rank * o.nlocal + sl, u) & 15u;
                const u32 h = m ? o.tpos[sl][u] : 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (((pres >> c) & 1u) && vj[c] < o.cap) rp[vj[c]] = ((m >> c) & 1u) ? h + (u32)__popc(m & ((1u << c) - 1u)) : DEAD;
            } else {       // index: the place the sample's expand kernel gave the child (see the record layout)
                if (o.single) {  // single sample: its planes are the union's, a wave's tile is the round
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (!((pres >> c) & 1u) || vj[c] >= o.cap) continue;
                        rp[vj[c]] = ((up[c] >> lane) & 1) ? (u32)c * o.seg + (u32)wv * 64u + (u32)__popcll(up[c] & lt) : DEAD;
                    }
                } else {
                    const u64* sp = o.splane[sl] + (size_t)wv * 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (!((pres >> c) & 1u) || vj[c] >= o.cap) continue;
                        const u64 mine = sp[c];
                        rp[vj[c]] = ((mine >> lane) & 1) ? (u32)c * o.seg + (u32)wv * 64u + (u32)__popcll(mine & lt) : DEAD;
                    }
                }
            }
        }
    }
    if (threadIdx.x == 0) o.h_totals[0] = total;  // (this kernel runs single-tile levels only)
}

// The same down-sweep for levels of more than one tile: one block per tile of 256 parents, one wave per 64.  Everything a wave needs
// from memory is requested before anything is used -- its planes, the tile's scanned counts, the parents' path words, reader
// counts and (one sample) column entries -- so a wave waits for one round trip; the waves of the tile then exchange their child
// counts through LDS (one barrier) and write.
template <typename P>
__global__ __launch_bounds__(256) void advance_wave_kernel(Xchg x, AdvanceOut o) {
    __shared__ u32 wcnt[4][4];
    const u32 F = (u32)x.F;
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6;
    const u32 tile = xcd_block();
    const u32 wi = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 w = tile * 4 + wi;
    const bool wave_in = w < nw;   // (the last tile of a level may end before its fourth wave)
    const u32 wc = wave_in ? w : 0u;
    const u64 lt = (1ull << lane) - 1;
    const u32 u = w * 64 + lane;
    const bool in = wave_in && u < F;
    const u32 uc = in ? u : 0u;
    // ---- requests ----
    u64 up[4];
    u32 base[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        up[c] = o.kplane[((size_t)wc << o.kshift) + c];
        base[c] = o.cnt4[(size_t)c * o.nbp + tile];
    }
    uint2 mypw = make_uint2(0u, 0u);
    uint2* pw = o.pw;
    u32 width = 0;
    if (o.pw_after_slot) width = o.width[0];
    if (pw) mypw = o.parent_pw[uc];
    uint2 sq = make_uint2(0u, 0u);
    if (o.sinfo) sq = *reinterpret_cast<const uint2*>(o.sinfo + (size_t)uc * 4);
    u32 myT = 1u;
    if (!o.single) myT = (u32)o.parent_nT[uc];
    // one sample: the node's column entry (frequency, flags) for the output predicates, whatever they will decide
    u64 candw = 0;  // (lanes 4 and 5: the wave's candidate bits and counts as the expand kernel left them)
    if (o.cand_copy && (lane == 4 || lane == 5)) candw = o.kplane[((size_t)wc << 3) + lane];
    u64 f1 = 0;
    u32 l1 = 0;
    if (o.single && o.filter_on) { f1 = (u64)x_freq<P>(x, 0, uc); l1 = x_pl<P>(x, 0, uc) >> 4; }
    // ---- the tile's child counts per wave ----
    if (!wave_in) {
#pragma unroll
        for (int c = 0; c < 4; ++c) up[c] = 0;
    }
    if (lane < 4) wcnt[wi][lane] = (u32)__popcll(DSM_PICK(up, lane));
    __syncthreads();
    if (!wave_in) return;
    if (o.pw_after_slot) pw = reinterpret_cast<uint2*>(reinterpret_cast<u8*>(o.slot) + (((size_t)width * 4 + 255) & ~(size_t)255));
    u32 cum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32 before = 0;
#pragma unroll
        for (u32 q = 0; q < 3; ++q) before += q < wi ? wcnt[q][c] : 0u;
        cum[c] = base[c] + before;
    }
    u32 nT4[4] = {0, 0, 0, 0};
    if (o.sinfo) {
        if (in) { nT4[0] = sq.x & 0xFFFFu; nT4[1] = sq.x >> 16; nT4[2] = sq.y & 0xFFFFu; nT4[3] = sq.y >> 16; }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) nT4[c] = (u32)((up[c] >> lane) & 1);
    }
    if (lane < 4) {
        o.kcum[(size_t)w * 4 + lane] = DSM_PICK(cum, lane);
        if (o.kplane_w) o.kplane_w[(size_t)w * 4 + lane] = DSM_PICK(up, lane);
        if (o.cnt_clear && wi == 0) o.cnt_clear[(size_t)lane * o.nbp + tile] = 0;  // the expand kernels of the next level add into it
    }
    if (o.cand_copy) {
        if (lane == 4) o.candbits[w] = candw;
        if (lane == 5) o.wsum[w] = candw;
    }
    u32 pres = 0, lastT = 0, vj[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        vj[c] = cum[c] + (u32)__popcll(up[c] & lt);
        if (nT4[c]) { pres |= 1u << c; lastT = nT4[c]; }
    }
    const u32 nc = __popc(pres);
    if (!in) myT = 0;
    const u32 same = o.single ? 1u : ((nc == 1 && lastT == myT) ? 1u : 0u);
    if (o.filter_on) {
        bool cand;
        if (o.single) {
            // metaserver.cpp:406-419 with one reader: the entropy of a single frequency is 0 up to rounding (the host decides it
            // exactly), so the thresholds are tested against 0 with the margin
            const FilterArgs& a = o.fa;
            cand = in && a.depth >= a.mindepth && 1u >= a.pmin && nc != 1 && !(f1 != 0 && l1 >= 1 && l1 <= 4);  // (one reader never exceeds pmax)
            if (cand && a.emax > 0) {
                const u64 sumN = 1ull + f1;
                const double sl = f1 ? (double)(f1 + 1) * (double)__log2f((float)(f1 + 1)) : 0.0;
                const double e = (double)__log2f((float)sumN) - sl / (double)sumN;
                if (e < a.emin - ENT_MARGIN || e > a.emax + ENT_MARGIN) cand = false;
            }
        } else {
            cand = in && filter_node<P>(o.fa, x, u, myT, nc, same);
        }
        const u64 bits = __ballot(cand);
        u64 pairs = (u64)__popcll(bits);
        if (!o.single) pairs = wave_sum_u64(cand ? (u64)myT : 0ull);
        if (lane == 0) { o.candbits[w] = bits; o.wsum[w] = (u64)__popcll(bits) | (pairs << 32); }
    }
    if (!in) return;
    if (!o.single) o.samechild[u] = (u8)same;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (((pres >> c) & 1u) && vj[c] < o.cap) {
            if (o.slot) o.slot[vj[c]] = 4u * u + (u32)c;
            if (pw) pw[vj[c]] = child_path_word(mypw, u, o.plevel, (u32)c);
            if (o.sa) o.sa[vj[c]] = make_uint2(u, (u32)c | ((pres & ((1u << c) - 1u)) ? 0u : SREC_FIRST) | ((pres >> (c + 1)) ? 0u : SREC_LAST));
            if (!o.single) o.nT[vj[c]] = (u16)nT4[c];
        }
    }
    if (o.single) {  // one sample: its planes are the union's, its handle table is passed directly
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (((pres >> c) & 1u) && vj[c] < o.cap) o.rp0[vj[c]] = (u32)c * o.seg + w * 64u + (u32)__popcll(up[c] & lt);
        return;
    }
    for (u32 sl = 0; sl < o.nlocal && pres && (o.tpos || o.rp_index); ++sl) {
        u32* rp = o.rp[sl];
        if (o.tpos) {
            const u32 m = x_pl<P>(x, o.rank * o.nlocal + sl, u) & 15u;
            const u32 h = m ? o.tpos[sl][u] : 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (((pres >> c) & 1u) && vj[c] < o.cap) rp[vj[c]] = ((m >> c) & 1u) ? h + (u32)__popc(m & ((1u << c) - 1u)) : DEAD;
        } else {
            const u64* sp = o.splane[sl] + (size_t)w * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!((pres >> c) & 1u) || vj[c] >= o.cap) continue;
                const u64 mine = sp[c];
                rp[vj[c]] = ((mine >> lane) & 1) ? (u32)c * o.seg + w * 64u + (u32)__popcll(mine & lt) : DEAD;
            }
        }
    }
}

// The down-sweep of ONE sample that is mined (no links, no reader counts, no stream records: path words and record handles only), the
// case BASELINE's metric is quoted on.  advance_wave_kernel handles every mode with run-time flags and spends ~420 instructions per
// wave of 64 parents -- more scalar than vector ones: the sweep is bound by their issue, not by the 21 bytes per node it moves.  Here a
// WAVE takes a whole tile of 256 parents: the tile's four plane lines arrive through scalar loads, the counts before each of its four
// rounds are scalar sums, no LDS and no barrier; the parents' path words of all four rounds are requested before the first is used.
// (the line of a round, written by the LF-step kernel: {plane[4], candidate bits, candidates | pairs << 32, -, -}, ExpandArgs::symbol_phase)
template <typename P>
__global__ __launch_bounds__(256) void advance_single_kernel(Xchg x, AdvanceOut o) {
    const int lane = threadIdx.x & 63;
    const u32 F = (u32)x.F;
    const u32 nw = (F + 63) >> 6;
    const u32 tile = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    if (tile >= o.nbp) return;
    const u64 lt = (1ull << lane) - 1;
    const u64* __restrict__ line = o.kplane + (size_t)tile * 32;
    // ---- requests: the parents' path words, the lines as the lanes will copy them, the tile's scanned counts ----
    uint2 mypw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32 u = (tile * 4 + (u32)q) * 64 + lane;
        mypw[q] = o.parent_pw[u < F ? u : 0u];
    }
    u64 copyw = 0;
    if (lane < 32) copyw = line[lane];
    u32 cum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cum[c] = o.cnt4[(size_t)c * o.nbp + tile];
    u64 up[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) up[q][c] = tile * 4 + (u32)q < nw ? line[q * 8 + c] : 0ull;   // (planes exist for the waves that hold nodes only)
    // ---- the retained directory of the level: planes and first-child indexes per round; the candidate words to their arrays ----
    {
        const u32 q = (u32)lane >> 3, j = (u32)lane & 7u, w = tile * 4 + q;
        if (lane < 32 && w < nw) {
            if (j < 4) o.kplane_w[(size_t)w * 4 + j] = copyw;
            else if (o.cand_copy && j == 4) o.candbits[w] = copyw;
            else if (o.cand_copy && j == 5) o.wsum[w] = copyw;
        }
    }
    const u32 r = o.plevel & (PW_CHUNK - 1);
    u32 kc = 0;   // lane 4q + c: index of the first child c of round q
    u32 cbase = 0;  // (candidate store) record of the round's first candidate
    u64 cbq[4] = {0, 0, 0, 0};   // candidate bits of the four rounds, their counts, this lane's frequencies where it is one
    u32 ncq[4] = {0, 0, 0, 0};
    u64 fq[4] = {0, 0, 0, 0};
    if (o.crec) {
        cbase = o.cnt4[(size_t)4 * o.nbp + tile] - o.cnt4[(size_t)4 * o.nbp];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (tile * 4 + (u32)q < nw) { cbq[q] = line[q * 8 + 4]; ncq[q] = (u32)line[q * 8 + 5]; }
            const u32 u = (tile * 4 + (u32)q) * 64 + lane;
            if ((cbq[q] >> lane) & 1ull) fq[q] = (u64)x_freq<P>(x, 0, u);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32 w = tile * 4 + (u32)q, u = w * 64 + lane;
        const uint2 ppw = mypw[q];
        if (o.crec && cbq[q]) {   // this round's candidates: node, frequency (metaserver.cpp:472-484 prints id:frequency; one sample: id 0)
            const u32 k = cbase + bits_below_lane(cbq[q]);
            if (((cbq[q] >> lane) & 1ull) && k < o.crec_cap) o.crec[k] = make_uint4(u, 0u, (u32)fq[q], (u32)(fq[q] >> 32));
        }
        cbase += ncq[q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u64 m = up[q][c];
            if (lane == q * 4 + c) kc = cum[c];
            const u32 below = bits_below_lane(m);
            const u32 vj = cum[c] + below;
            if (((m >> lane) & 1ull) && vj < o.cap) {
                o.rp0[vj] = (u32)c * o.seg + w * 64u + below;
                o.pw[vj] = r == 0 ? make_uint2(u, (u32)c) : make_uint2(ppw.x, ppw.y | ((u32)c << (2 * r)));
            }
            cum[c] += (u32)__popcll(m);
        }
    }
    if (lane < 16 && tile * 4 + ((u32)lane >> 2) < nw) o.kcum[(size_t)tile * 16 + lane] = kc;
}

// ---------------------------------------------------------------------------------------------
// Owner mode, on the ranks that do NOT merge the prefix: the owner's broadcast carries the union's child planes of the level
// (4 x 64 bits per 64 parents); from them a client only needs the next level's links (4 * parent + symbol), from which its
// LF-step kernel derives the handles of its own records (expand_tile, SELF).  Tile counts, one scan, links.
// ---------------------------------------------------------------------------------------------
// (kshift: log2 of the words a wave's entry takes in kplane -- 2, or 3 for the whole lines a single sample's LF-step kernel writes)
// cand_row (kshift 3: the lines a single sample's LF-step kernel writes): a fifth row, the tile's candidates (word 5 of a line)
__global__ __launch_bounds__(256) void lite_count_kernel(const u64* __restrict__ kplane, u32 nw, u32* __restrict__ cnt4, u32 nbp, u32 kshift, u32 cand_row = 0) {
    const u32 tile = blockIdx.x * 64 + (threadIdx.x >> 2), c = threadIdx.x & 3;  // one thread per (tile, symbol)
    if (tile >= nbp) return;
    u32 s = 0, nc = 0;
    if (cand_row && tile * 4 + c < nw) nc = (u32)kplane[((size_t)(tile * 4 + c) << 3) + 5];   // thread c: the candidates of the tile's wave c
#pragma unroll
    for (u32 q = 0; q < 4; ++q) {
        const u32 w = tile * 4 + q;
        if (w < nw) s += (u32)__popcll(kplane[((size_t)w << kshift) + c]);
    }
    cnt4[(size_t)c * nbp + tile] = s;
    if (cand_row) {   // (the four threads of a tile are neighbouring lanes)
        nc += __shfl_xor(nc, 1, 64);
        nc += __shfl_xor(nc, 2, 64);
        if (c == 0) cnt4[(size_t)4 * nbp + tile] = nc;
    }
}
__global__ __launch_bounds__(256) void lite_slot_kernel(const u64* __restrict__ kplane, u32 F, const u32* __restrict__ cnt4, u32 nbp, u32 single_tile,
                                                        u32* __restrict__ slot, u32 cap) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6;
    const u32 tile = xcd_block();
    const u32 wi = threadIdx.x >> 6, w = tile * 4 + wi;
    if (w >= nw) return;
    const u32 u = w * 64 + lane;
    u32 base = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32 before = 0, all = 0;
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
            const u32 wq = tile * 4 + q;
            const u32 t = wq < nw ? (u32)__popcll(kplane[(size_t)wq * 4 + c]) : 0u;
            before += q < wi ? t : 0u;
            all += t;
        }
        const u64 up = kplane[(size_t)w * 4 + c];
        const u32 first = (single_tile ? base : cnt4[(size_t)c * nbp + tile]) + before;
        base += all;
        const u32 vj = first + (u32)__popcll(up & ((1ull << lane) - 1));
        if (u < F && ((up >> lane) & 1) && vj < cap) slot[vj] = 4u * u + (u32)c;
    }
}

// ---------------------------------------------------------------------------------------------
// Iteration order of the reference's std::unordered_set<unsigned> reader sets (metaserver.cpp:23).
// With at most 13 samples libstdc++ keeps 13 buckets and every id its own bucket, so a set iterates in
// REVERSE insertion order; the insertion sequence into children[c] follows readChildren() round by
// round (metaserver.cpp:159-189, 322-339).  Orders are nibble-packed, first iterated id in bits 0-3.
// ---------------------------------------------------------------------------------------------
// Most nodes have ONE child in the union trie: all readers that go on enter it in the parent's order, so its set iterates in the
// reverse of that.  Those nodes are done in the first phase; the nodes with several children -- a few per wave, but nearly every
// wave has one -- are queued in LDS and replayed round by round afterwards by as many lanes as there are such nodes, instead of
// every wave running the replay for its few.
template <typename P>
__global__ __launch_bounds__(256) void order_kernel(u32 F, Xchg x, const u16* __restrict__ nT, const u64* __restrict__ order, Kids kids,
                                                    u64* __restrict__ order_next) {
    __shared__ u32 qn;
    __shared__ u32 qu[256];
    __shared__ u64 qm[256], qo[256];
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < F) {
        const u32 cnt = nT[u];
        const u64 ord = order[u];
        u64 mbyid = 0;  // 4 presence bits per reader id (ids < 16)
        u32 um = 0;     // children of the node in the union
        if (x_is_nm8(x)) {  // (a sample that does not hold the node has an empty word: no children)
            u32 w[8];
            x_words8(x, u, w);
#pragma unroll
            for (int g = 0; g < 8; ++g) { const u32 m = (w[g] >> 9) & 15u; mbyid |= (u64)m << (4 * g); um |= m; }
        } else {
            for (u32 k = 0; k < cnt; ++k) {
                const u32 r = (u32)((ord >> (4 * k)) & 15);
                const u32 m = x_pl<P>(x, r, u) & 15u;  // which children this reader continues into
                mbyid |= (u64)m << (4 * r);
                um |= m;
            }
        }
        if (um && !(um & (um - 1))) {  // one child: the readers that enter it, last one first
            u64 rev = 0;
            for (u32 k = 0; k < cnt; ++k) {
                const u32 r = (u32)((ord >> (4 * k)) & 15);
                if ((mbyid >> (4 * r)) & 15) rev = (rev << 4) | r;
            }
            order_next[kid_index(kids, u, (u32)(__ffs((int)um) - 1))] = rev;
        } else if (um) {
            const u32 q = atomicAdd(&qn, 1u);
            qu[q] = u; qm[q] = mbyid; qo[q] = ord;
        }
    }
    __syncthreads();
    for (u32 q = threadIdx.x; q < qn; q += blockDim.x) {
        const u32 v = qu[q];
        const u64 mbyid = qm[q], ord = qo[q];
        const u32 cnt = nT[v];
        u64 ins[4] = {0, 0, 0, 0};
        u32 icnt[4] = {0, 0, 0, 0};
        // round 1: every reader of the parent reads its first child
        for (u32 k = 0; k < cnt; ++k) {
            u32 r = (u32)((ord >> (4 * k)) & 15);
            u32 m = (u32)((mbyid >> (4 * r)) & 15);
            if (m) {
                int f = __ffs(m) - 1;
                ins[f] |= (u64)r << (4 * icnt[f]);
                ++icnt[f];
            }
        }
        for (int i = 0; i < 4; ++i) {
            if (!icnt[i]) continue;
            // iteration order of children[i] = reverse insertion order
            u64 rev = 0;
            for (u32 k = 0; k < icnt[i]; ++k) rev |= ((ins[i] >> (4 * k)) & 15) << (4 * (icnt[i] - 1 - k));
            order_next[kid_index(kids, v, (u32)i)] = rev;
            // next round: the readers of this child (in its iteration order) read their next child
            for (u32 k = 0; k < icnt[i]; ++k) {
                u32 r = (u32)((rev >> (4 * k)) & 15);
                u32 m = (u32)((mbyid >> (4 * r)) & 15) & ~((2u << i) - 1);
                if (m) {
                    int g = __ffs(m) - 1;
                    ins[g] |= (u64)r << (4 * icnt[g]);
                    ++icnt[g];
                }
            }
        }
    }
}

// More than 13 samples: the sets rehash (13 -> 29 -> 59 -> ...) and ids share buckets, so the order is replayed with the
// container model of setorder.h.  Orders are u16 arrays, d entries per node.
template <typename P, int MAXD>
__global__ void order_big_kernel(u32 F, Xchg x, const u16* __restrict__ nT, const u16* __restrict__ order, Kids kids,
                                 u16* __restrict__ order_next) {
    // per-thread working set in scratch: byte-sized ids when they fit (at most 64 samples: 575 B instead of 1.1 KB)
    typedef typename std::conditional<(MAXD <= 250), u8, u16>::type K;
    u32 u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= F) return;
    const u32 cnt = nT[u];
    const u16* ord = order + (size_t)u * x.d;
    u8 mask[MAXD];
    K ins[4][MAXD];
    K loc[MAXD];
    K work[MAXD + (MAXD <= 127 ? 127 : 541)];  // so_work_size(MAXD, MAXD): next pointers by id, bucket table
    u32 icnt[4] = {0, 0, 0, 0};
    for (u32 k = 0; k < cnt; ++k) {  // round 1: every reader of the parent reads its first child
        const u32 r = ord[k];
        u32 m = x_pl<P>(x, r, u) & 15u;
        mask[r] = (u8)m;
        if (m) { int f = __ffs(m) - 1; ins[f][icnt[f]++] = (K)r; }
    }
    for (int i = 0; i < 4; ++i) {
        if (!icnt[i]) continue;
        set_iteration_order<K>(ins[i], icnt[i], loc, work, (u32)MAXD);
        u16* dst = order_next + (size_t)kid_index(kids, u, (u32)i) * x.d;
        for (u32 k = 0; k < icnt[i]; ++k) dst[k] = (u16)loc[k];
        for (u32 k = 0; k < icnt[i]; ++k) {  // next round: this child's readers, in its iteration order
            const u32 r = loc[k];
            u32 m = mask[r] & ~((2u << i) - 1);
            if (m) { int g = __ffs(m) - 1; ins[g][icnt[g]++] = (K)r; }
        }
    }
}

// Per wave of 64 nodes: candbits[w] = which of them are candidates, wsum[w] = their number (low word) and their number of
// (id, freq) pairs (high word); a scan over the waves gives every wave the place of its first candidate.
template <typename P>
__global__ __launch_bounds__(256) void filter_kernel(FilterArgs a, Xchg x, const u16* __restrict__ nT, Kids kids,
                                                     const u8* __restrict__ samechild, u64* __restrict__ candbits, u64* __restrict__ wsum) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (a.F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    const bool one = a.d == 1;  // a single sample: every node has one reader, a single child always carries it
    u32 t[NPT];
    bool out[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {  // the cheap predicates of all NPT nodes first: their loads are in flight together
        const u32 w = w0 + (u32)i * stride;
        const u32 v = w * 64 + lane, vc = v < a.F ? v : 0u;
        t[i] = one ? 1u : (u32)nT[vc];
        u32 nc = 0;
        if (w < nw) {
            KidWave kw;
            kid_wave(kids, w, kw);
            nc = (u32)((kw.p[0] >> lane) & 1) + (u32)((kw.p[1] >> lane) & 1) + (u32)((kw.p[2] >> lane) & 1) + (u32)((kw.p[3] >> lane) & 1);
        }
        const u32 same = one ? 1u : (u32)samechild[vc];
        bool o = v < a.F;
        if (a.depth < a.mindepth) o = false;
        if (a.pmax != 0 && t[i] > a.pmax) o = false;
        if (t[i] < a.pmin) o = false;
        if (nc == 1 && same) o = false;
        out[i] = o;
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride;
        const u32 v = w * 64 + lane;
        if (out[i]) {  // merged left char (metaserver.cpp:383-387) and entropy over the samples that hold the node
            u64 sumN = a.d;
            double s = 0;
            u32 l = 0xFF;
            for (u32 g = 0; g < a.d; ++g) {
                u64 f = (u64)x_freq<P>(x, g, v);
                if (f) {
                    u32 lg = x_pl<P>(x, g, v) >> 4;
                    l = l == 0xFF ? lg : (l == lg ? l : 5u);
                    sumN += f;
                    if (a.emax > 0) s += (double)(f + 1) * (double)__log2f((float)(f + 1));
                }
            }
            if (l >= 1 && l <= 4) out[i] = false;
            if (out[i] && a.emax > 0) {
                double e = (double)__log2f((float)sumN) - s / (double)sumN;
                if (e < a.emin - ENT_MARGIN || e > a.emax + ENT_MARGIN) out[i] = false;
            }
        }
        if (w < nw) {
            const u64 bits = __ballot(out[i]);
            u64 pairs = (u64)__popcll(bits);
            if (!one) pairs = wave_sum_u64(out[i] ? (u64)t[i] : 0ull);
            if (lane == 0) { candbits[w] = bits; wsum[w] = (u64)__popcll(bits) | (pairs << 32); }
        }
    }
}

// store the candidates of a level: node index and (id, freq) pairs in the reference's iteration order
template <typename P>
__global__ __launch_bounds__(256) void cand_store_kernel(FilterArgs a, Xchg x, const u16* __restrict__ nT, const u64* __restrict__ order,
                                                         const u16* __restrict__ order16, const u64* __restrict__ candbits,
                                                         const u64* __restrict__ wscan, u32* __restrict__ cand_node, u32* __restrict__ cand_poff,
                                                         u32* __restrict__ ids, u64* __restrict__ freqs) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (a.F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    // (few nodes are candidates: most waves only look at their words.  The words of all of a wave's groups are requested first.)
    u64 bitsv[NPT], basev[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, wc = w < nw ? w : 0u;
        bitsv[i] = w < nw ? candbits[wc] : 0ull;
        basev[i] = wscan[wc];
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride;
        const u64 bits = bitsv[i];
        if (!bits) continue;
        const u64 base = basev[i];
        const bool mine = (bits >> lane) & 1;
        const u32 v = w * 64 + lane;
        const u32 cnt = mine ? (a.d > 1 ? (u32)nT[v] : 1u) : 0u;  // pairs of this lane's candidate = samples that hold the node
        const u32 k = (u32)(base & 0xFFFFFFFFu) + bits_below_lane(bits);
        u32 o = k;
        if (a.d > 1) o = (u32)(base >> 32) + (wave_inclusive_scan<u32>(cnt) - cnt);  // pairs of the candidates in lower lanes
        if (!mine) continue;
        cand_node[k] = v;
        cand_poff[k] = o;
        const u64 j = v;
        if (a.exact_order == 1) {
            const u64 ord = order[v];
            for (u32 q = 0; q < cnt; ++q) {
                u32 g = (u32)((ord >> (4 * q)) & 15);
                ids[o] = g; freqs[o] = (u64)x_freq<P>(x, g, j); ++o;
            }
        } else if (a.exact_order == 2) {
            const u16* ord = order16 + (size_t)v * a.d;
            for (u32 q = 0; q < cnt; ++q) {
                u32 g = ord[q];
                ids[o] = g; freqs[o] = (u64)x_freq<P>(x, g, j); ++o;
            }
        } else {
            for (u32 g = 0; g < a.d; ++g) {
                u64 f = (u64)x_freq<P>(x, g, j);
                if (f) { ids[o] = g; freqs[o] = f; ++o; }
            }
        }
    }
}

// stream mode (one sample): a level retains one 16-byte record per node for the wire stream:
//   .x parent index, .y symbol | first-child << 2 | last-child << 3 (from the advance sweep of the parent's level, which leaves
//   them in a dense 8-byte array of its own: both kernels then write whole lines),
//   .z/.w frequency | left-char code << 61 (known when the node's own level has been expanded: here)
template <typename P>
__global__ void keep_kernel(u32 F, Xchg x, const uint2* __restrict__ sa, uint4* __restrict__ srec, u8* __restrict__ clen) {
    u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= F) return;
    const uint2 a = sa[v];
    const u64 f = (u64)x_freq<P>(x, 0, v);
    const u64 w = f | ((u64)(x_pl<P>(x, 0, v) >> 4) << 61);
    srec[v] = make_uint4(a.x, a.y, (u32)w, (u32)(w >> 32));
    // bytes of the node's closing token without an 'R' part: varint(freq) left ')' (ClientSocket.h:20-39); the top-down sweep of
    // the stream reads this byte instead of the record
    clen[v] = (u8)((f < 128 ? 1u : 1u + (u32)((64 - __clzll((long long)f) + 7) >> 3)) + 2u);
}

// ---- subtree aggregates over the retained levels ------------------------------------------------
// A wave takes 64 consecutive nodes (and NPT such groups a grid-width apart): the directory words are wave-uniform.  The sweeps
// are latency chains -- directory words, then the gathers they address, then the store -- so a wave issues the loads of one
// kind for ALL its groups before it uses any (no early exit between them: a group beyond the level reads group 0 and stores
// nothing, a lane without that child reads entry 0): two round trips per wave instead of two per group.
// bottom-up: agg[v] = own[v] + sum over children agg_child
template <typename T, typename OwnT>
__global__ __launch_bounds__(256) void up_kernel(u32 F, const OwnT* __restrict__ own, Kids kids, const T* __restrict__ child_agg,
                                                 T* __restrict__ agg) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    KidWave kw[NPT];
    T s[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, wc = w < nw ? w : 0u;
        const u32 v = wc * 64 + lane;
        s[i] = v < F ? (own ? (T)own[v] : (T)1) : (T)0;
        if (child_agg) kid_wave(kids, wc, kw[i]);
    }
    if (child_agg) {
        T g[NPT][4];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) g[i][c] = child_agg[((kw[i].p[c] >> lane) & 1) ? kw[i].c[c] + bits_below_lane(kw[i].p[c]) : 0u];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[i] += ((kw[i].p[c] >> lane) & 1) ? g[i][c] : (T)0;
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, v = w * 64 + lane;
        if (w < nw && v < F) agg[v] = s[i];
    }
}
// the same with own[v] = bit v of a word array (null: 0 everywhere): candidates in the subtree
template <typename T>
__global__ __launch_bounds__(256) void up_bits_kernel(u32 F, const u64* __restrict__ ownbits, Kids kids, const T* __restrict__ child_agg,
                                                      T* __restrict__ agg) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    KidWave kw[NPT];
    T s[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, wc = w < nw ? w : 0u;
        s[i] = ownbits ? (T)((ownbits[wc] >> lane) & 1) : (T)0;
        if (child_agg) kid_wave(kids, wc, kw[i]);
    }
    if (child_agg) {
        T g[NPT][4];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) g[i][c] = child_agg[((kw[i].p[c] >> lane) & 1) ? kw[i].c[c] + bits_below_lane(kw[i].p[c]) : 0u];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[i] += ((kw[i].p[c] >> lane) & 1) ? g[i][c] : (T)0;
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, v = w * 64 + lane;
        if (w < nw && v < F) agg[v] = s[i];
    }
}
// top-down: start[child_k] = start[v] + lead + sum_{j<k} agg[child_j]   (children in A,C,G,T order)
template <typename T>
__global__ __launch_bounds__(256) void down_kernel(u32 F, const T* __restrict__ start, T lead, Kids kids,
                                                   const T* __restrict__ child_agg, T* __restrict__ child_start) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    KidWave kw[NPT];
    T s[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, wc = w < nw ? w : 0u;
        const u32 v = wc * 64 + lane;
        kid_wave(kids, wc, kw[i]);
        s[i] = start[v < F ? v : 0u] + lead;
    }
    // a child's subtree size is needed only when a later sibling exists: most nodes have one child, and their lanes read entry 0
    T g[NPT][3];
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool need = ((kw[i].p[c] >> lane) & 1) && ((((kw[i].p[1] >> lane) & 1) << 1 | ((kw[i].p[2] >> lane) & 1) << 2 | ((kw[i].p[3] >> lane) & 1) << 3) >> (c + 1)) != 0;
            g[i][c] = child_agg[need ? kw[i].c[c] + bits_below_lane(kw[i].p[c]) : 0u];
        }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride;
        if (w >= nw) continue;
        const u32 m = (u32)((kw[i].p[0] >> lane) & 1) | ((u32)((kw[i].p[1] >> lane) & 1) << 1) | ((u32)((kw[i].p[2] >> lane) & 1) << 2) |
                      ((u32)((kw[i].p[3] >> lane) & 1) << 3);
        T acc = s[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((m >> c) & 1u) {
                child_start[kw[i].c[c] + bits_below_lane(kw[i].p[c])] = acc;
                if (c < 3 && (m >> (c + 1))) acc += g[i][c];
            }
        }
    }
}

// candidate k of a level gets its post-order rank among all candidates of the prefix; the tuple's sizes go to their place
// in output order (the ranks of neighbouring candidates are far apart: the levels are in colex order, the output in trie order)
// (crec: the records the advance sweep of a single sample stored -- node, 0, frequency -- instead of the four arrays: one pair each)
__global__ void cand_rank_kernel(u32 ncand, const u32* __restrict__ cand_node, const u32* __restrict__ cand_poff, u32 npairs,
                                 const u32* __restrict__ start, const u32* __restrict__ sub, u32 level, u32* __restrict__ crank,
                                 u32* __restrict__ plen, u32* __restrict__ npair, const uint4* __restrict__ crec) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncand) return;
    u32 v = crec ? crec[k].x : cand_node[k];
    u32 r = start[v] + sub[v] - 1;
    crank[k] = r;
    plen[r] = level;
    npair[r] = crec ? 1u : (k + 1 < ncand ? cand_poff[k + 1] : npairs) - cand_poff[k];
}

struct LevelDev {
    const uint2* pw;    // path words of the level's nodes (see AdvanceOut)
    const u32* cand_node;
    const u32* cand_poff;
    const uint4* crec;  // (one sample, stored by the advance sweep: {node, 0, frequency} per candidate instead of the arrays around it)
    const u32* crank;   // post-order rank of candidate k
    const u32* ids;
    const u64* freqs;
    u32 ncand;
    u32 npairs;
    u32 cbase;          // candidates of the shallower levels
};

// offsets of the chunk boundaries (path bytes, pairs) for the host: out[2c], out[2c+1] for boundary tuple tb[c]
constexpr int EMIT_MAX_CHUNKS = 8;
struct ChunkBounds { u32 tb[EMIT_MAX_CHUNKS + 1]; int n; };
__global__ void chunk_bounds_kernel(ChunkBounds cbs, const u32* __restrict__ path_off, const u32* __restrict__ pair_off, u32* __restrict__ out) {
    const int c = threadIdx.x;
    if (c <= cbs.n) { out[2 * c] = path_off[cbs.tb[c]]; out[2 * c + 1] = pair_off[cbs.tb[c]]; }
}

// Paths and pairs of all tuples.  Threads take the candidates level by level in node order.  A path is put together from the path
// words of the node and of its ancestors at the chunk boundaries (levels 16, 32, ...): 16 symbols per dependent load.
// What the host emitter used to compute per tuple (emit_job's first pass, 3.4 ms of sixteen threads per chunk of four million tuples,
// in the open at the end of a pass): the exact entropy (metaserver.cpp:366-389 from the uploaded tables: the same entries added in the
// same order, one division, one subtraction -- IEEE double, bit-identical to the host's), the emin / emax verdict (:413) and the offsets
// relative to the chunk (a batch's offsets start at 0).  The thread that fills a tuple has its frequencies in registers.  A frequency or
// a total beyond the tables (a few hundred nodes at the top of a pass) leaves the tuple to the host (EV_HOST).
constexpr u8 EV_DROP = 0, EV_KEEP = 1, EV_HOST = 2;
struct FillVerdict {
    double* ent;          // per tuple (output rank); null: no verdicts (text mode computes its own)
    u8* keep;
    u32* rel_path;        // offsets relative to the chunk, chunk c at [rank_lo + c, rank_hi + c]
    u32* rel_pair;
    u32* counts;          // of the chunk: [0] dropped, [1] left to the host
    const double* terms;  // device copies of term_table() / logn_table()
    const double* logn;
    u32 chunk, d;
    u32 pb0, qb0;         // path byte / pair at which the chunk starts
    double emin, emax;
};
__global__ __launch_bounds__(256) void tuple_fill_kernel(u32 nt, u32 nlev, const LevelDev* __restrict__ lv, const u32* __restrict__ path_off,
                                                         const u32* __restrict__ pair_off, char* __restrict__ paths, u32* __restrict__ ids,
                                                         u64* __restrict__ freqs, u32 rank_lo, u32 rank_hi, FillVerdict fv) {
    // the per-level arrays and candidate bases are read by every walk: kept in LDS, with the text of every byte of four symbols
    constexpr u32 LDS_LEVELS = 1024;
    __shared__ const uint2* s_pw[LDS_LEVELS];
    __shared__ u32 s_cbase[LDS_LEVELS];
    __shared__ u32 s_text[256];
    const u32 nl = nlev < LDS_LEVELS ? nlev : LDS_LEVELS;
    for (u32 q = threadIdx.x; q < nl; q += blockDim.x) { s_pw[q] = lv[q].pw; s_cbase[q] = lv[q].cbase; }
    {
        const u32 b = threadIdx.x;  // (256 threads: one table entry each) "ACGT"[sym], the first symbol in the lowest byte
        s_text[b] = ((0x54474341u >> (8 * (b & 3))) & 0xFFu) | (((0x54474341u >> (8 * ((b >> 2) & 3))) & 0xFFu) << 8) |
                    (((0x54474341u >> (8 * ((b >> 4) & 3))) & 0xFFu) << 16) | (((0x54474341u >> (8 * ((b >> 6) & 3))) & 0xFFu) << 24);
    }
    __syncthreads();
    const u32 f = blockIdx.x * blockDim.x + threadIdx.x;  // candidates in level-major order
    if (f >= nt) return;
    u32 lo = 1, hi = nlev - 1;  // the level whose candidates include f: largest l with cbase[l] <= f among the levels that have any
    while (lo < hi) {
        const u32 mid = (lo + hi + 1) >> 1;
        if ((mid < LDS_LEVELS ? s_cbase[mid] : lv[mid].cbase) <= f) lo = mid; else hi = mid - 1;
    }
    const u32 lvl = lo;
    const LevelDev L = lv[lvl];
    const u32 k = f - L.cbase;
    const u32 r = L.crank[k];
    if (r < rank_lo || r >= rank_hi) return;  // (a launch fills one chunk of consecutive output ranks)
    const bool one = L.crec != nullptr;
    const uint4 cr = one ? L.crec[k] : make_uint4(0u, 0u, 0u, 0u);
    const u32 b = one ? k : L.cand_poff[k];
    const u32 e = one ? k + 1 : (k + 1 < L.ncand ? L.cand_poff[k + 1] : L.npairs);
    u32 o = pair_off[r];
    const u32 o_first = o, p_first = path_off[r];
    u64 sumN = fv.d;
    double sl = 0;
    bool beyond = false;
    for (u32 q = b; q < e; ++q, ++o) {
        const u64 fq = one ? (((u64)cr.w << 32) | cr.z) : L.freqs[q];
        ids[o] = one ? 0u : L.ids[q]; freqs[o] = fq;
        sumN += fq;
        if (fv.ent && !beyond) {
            if (fq < TERM_TAB) sl += fv.terms[fq]; else beyond = true;
        }
    }
    if (fv.ent) {
        beyond = beyond || sumN >= LOGN_TAB;
        u8 verdict = EV_HOST;
        if (!beyond) {
            const double en = fv.logn[sumN] - sl / (double)sumN;
            fv.ent[r] = en;
            verdict = (fv.emax > 0 && (en < fv.emin || en > fv.emax)) ? EV_DROP : EV_KEEP;
        }
        fv.keep[r] = verdict;
        fv.rel_path[r + fv.chunk] = p_first - fv.pb0;
        fv.rel_pair[r + fv.chunk] = o_first - fv.qb0;
        if (r + 1 == rank_hi) {  // the chunk's closing entries
            fv.rel_path[r + 1 + fv.chunk] = path_off[r + 1] - fv.pb0;
            fv.rel_pair[r + 1 + fv.chunk] = o - fv.qb0;
        }
        const u64 md = __ballot(verdict == EV_DROP), mh = __ballot(verdict == EV_HOST);
        if ((md | mh) && (threadIdx.x & 63) == (u32)(__ffsll((long long)__ballot(1)) - 1)) {
            if (md) atomicAdd(fv.counts, (u32)__popcll(md));
            if (mh) atomicAdd(fv.counts + 1, (u32)__popcll(mh));
        }
    }
    u32 v = one ? cr.x : L.cand_node[k];
    char* dst = paths + p_first;
    u32 l = lvl;
    while (l > 0) {
        const u32 first = ((l - 1) / PW_CHUNK) * PW_CHUNK;  // path position of the chunk's first symbol = level of the ancestor it hangs from
        const uint2 w = (l < LDS_LEVELS ? s_pw[l] : lv[l].pw)[v];
        const u32 cnt = l - first;  // 1..16 symbols; only a path's last chunk is a partial one
        const u32 q0 = s_text[w.y & 0xFFu], q1 = s_text[(w.y >> 8) & 0xFFu], q2 = s_text[(w.y >> 16) & 0xFFu], q3 = s_text[w.y >> 24];
        char* d = dst + first;
        if (cnt == PW_CHUNK) {
            __builtin_memcpy(d, &q0, 4); __builtin_memcpy(d + 4, &q1, 4); __builtin_memcpy(d + 8, &q2, 4); __builtin_memcpy(d + 12, &q3, 4);
        } else {
            const u32 full = cnt >> 2;
            if (full > 0) __builtin_memcpy(d, &q0, 4);
            if (full > 1) __builtin_memcpy(d + 4, &q1, 4);
            if (full > 2) __builtin_memcpy(d + 8, &q2, 4);
            const u32 tail = full == 0 ? q0 : (full == 1 ? q1 : (full == 2 ? q2 : q3));
            for (u32 q = 0; q < (cnt & 3u); ++q) d[4 * full + q] = (char)(tail >> (8 * q));
        }
        v = w.x;
        l = first;
    }
}

// ---- wire stream (ClientSocket.h:20-39) ---------------------------------------------------------
// The stream is the depth-first serialisation  node := '(' sym node* varint(freq) ['R' varint(count)]{depth<=6} leftchar ')'.
// Cut at the leaves it is a sequence of CHUNKS, one per leaf in trie order: the opening tokens of the leaf's "open run" (the leaf
// and its ancestors as long as each is the FIRST child of its parent: their '(' sym tokens are adjacent in the stream), then the
// closing tokens of its "close run" (the leaf and its ancestors as long as each is the LAST child).  Every node is in exactly one
// open run and one close run.  So: leaves per subtree bottom-up, then top-down the rank of every leaf and the sizes of its two
// runs, a scan over the leaves for the chunk offsets, and one thread per leaf that walks up its runs -- one dependent read per
// node -- and writes its chunk front to back: neighbouring threads write neighbouring bytes.  (A node-per-thread scatter of the
// two tokens wrote two partial lines per node into the multi-GB buffer.)
__device__ __forceinline__ u32 varint_len(u64 u) {
    if (u < 128) return 1;
    return 1 + (u32)((64 - __clzll((long long)u) + 7) >> 3);
}
__device__ __forceinline__ u32 put_varint(u8* p, u64 u) {
    if (u < 128) { p[0] = (u8)(u | 0x80); return 1; }
    u32 l = (u32)((64 - __clzll((long long)u) + 7) >> 3);
    p[0] = (u8)l;
    for (u32 k = 0; k < l; ++k) p[1 + k] = (u8)(u >> (8 * k));
    return 1 + l;
}
constexpr u64 SREC_FREQ_MASK = (1ull << 61) - 1;
__device__ __forceinline__ u64 srec_word(const uint4& r) { return ((u64)r.w << 32) | r.z; }

// bottom-up: leaves in the subtree (a node without children is one)
__global__ __launch_bounds__(256) void stream_leaves_kernel(u32 F, Kids kids, const u32* __restrict__ child_lf, u32* __restrict__ lf) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (F + 63) >> 6, stride = gridDim.x * 4;
    const u32 w0 = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    KidWave kw[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, wc = w < nw ? w : 0u;
        if (child_lf) kid_wave(kids, wc, kw[i]);
    }
    u32 s[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) s[i] = 0;
    if (child_lf) {
        u32 g[NPT][4];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) g[i][c] = child_lf[((kw[i].p[c] >> lane) & 1) ? kw[i].c[c] + bits_below_lane(kw[i].p[c]) : 0u];
#pragma unroll
        for (int i = 0; i < NPT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) s[i] += ((kw[i].p[c] >> lane) & 1) ? g[i][c] : 0u;
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const u32 w = w0 + (u32)i * stride, v = w * 64 + lane;
        if (w < nw && v < F) lf[v] = s[i] ? s[i] : 1u;
    }
}

// What travels top-down: per node the rank of its first leaf, and what it inherits from its parent -- the length of the open run
// above it (0 unless it is a first child) and the bytes of the close run above it (0 unless it is a last child).
struct StreamDown {
    u32 F, level;
    const u32* rank;      // this level: [F]
    const u32* kin;
    const u32* cin;
    u32* rank_n;          // next level
    u32* kin_n;
    u32* cin_n;
    const u8* clen;       // this level's closing-token sizes (null at level 0: the root is not a node of the stream)
    const u32* child_lf;  // leaves per subtree of the next level (null: this is the last level)
    Kids kids;
    u64* leaf_id;         // per leaf rank: level << 32 | node
    u32* chunk;           // bytes of the leaf's chunk ('R' tokens of the nodes above depth 6 are added later)
    u32* kop;             // length of its open run
    u32* top_rank;        // levels 1..6: rank of the node's first leaf and its number of leaves (for the 'R' tokens), may be null
    u32* top_lf;
    const u32* lf;        // this level's leaves per subtree
};
__global__ __launch_bounds__(256) void stream_down_kernel(StreamDown a) {
    const int lane = threadIdx.x & 63;
    const u32 nw = (a.F + 63) >> 6;
    const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)(xcd_block() * 4 + (threadIdx.x >> 6)));
    if (w >= nw) return;
    const u32 v = w * 64 + lane;
    const bool in = v < a.F;
    const u32 vc = in ? v : 0u;
    KidWave kw;
    u32 m = 0;
    if (a.child_lf) {
        kid_wave(a.kids, w, kw);
        m = (u32)((kw.p[0] >> lane) & 1) | ((u32)((kw.p[1] >> lane) & 1) << 1) | ((u32)((kw.p[2] >> lane) & 1) << 2) | ((u32)((kw.p[3] >> lane) & 1) << 3);
    }
    const u32 rank = a.rank[vc], kin = a.kin[vc], cin = a.cin[vc];
    u32 mylen = 0;
    if (a.clen) mylen = a.clen[vc];
    u32 g[3] = {0, 0, 0};
    if (a.child_lf) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool need = ((m >> c) & 1u) && (m >> (c + 1)) != 0;  // a child's leaves matter only when a later sibling exists
            g[c] = a.child_lf[need ? kw.c[c] + bits_below_lane(kw.p[c]) : 0u];
        }
    }
    if (!in) return;
    const u32 kopen = a.level >= 1 ? kin + 1 : 0u;                                                   // open run down to and including this node
    const u32 cb = a.level >= 1 ? cin + mylen : 0u;                                                  // close run: varint(freq) left ')'
    if (a.top_rank) { a.top_rank[v] = rank; a.top_lf[v] = a.lf[v]; }
    if (!m) {  // a leaf: its chunk
        if (a.level >= 1) {
            a.leaf_id[rank] = ((u64)a.level << 32) | v;
            a.chunk[rank] = 2u * kopen + cb;
            a.kop[rank] = kopen;
        }
        return;
    }
    const u32 firstc = (u32)__ffs((int)m) - 1u, lastc = 31u - (u32)__clz((int)m);
    u32 r = rank;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if ((m >> c) & 1u) {
            const u32 ch = kw.c[c] + bits_below_lane(kw.p[c]);
            a.rank_n[ch] = r;
            a.kin_n[ch] = (u32)c == firstc ? kopen : 0u;
            a.cin_n[ch] = (u32)c == lastc ? cb : 0u;
            if (c < 3) r += g[c];
        }
    }
}

// 'R' tokens (EnumerateQuery.cpp:214-218): a node of depth <= 6 sends the number of nodes reported when its subtree is done =
// the nodes opened by the chunks up to its last leaf (every node's '(' is in exactly one chunk, and chunks are in stream order).
// cumk[r] = open-run lengths of the leaves before r.  The token belongs to the chunk of the node's LAST leaf.
__global__ void stream_rtok_kernel(u32 F, u64 rbase, const u32* __restrict__ top_rank, const u32* __restrict__ top_lf, const u64* __restrict__ cumk,
                                   const u32* __restrict__ kop, u32 nleaf, u64* __restrict__ rval, u32* __restrict__ chunk) {
    u32 v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= F) return;
    const u32 last = top_rank[v] + top_lf[v] - 1;
    const u64 val = rbase + cumk[last] + kop[last];
    rval[v] = val;
    atomicAdd(&chunk[last], 1u + varint_len(val));
}

// one thread per leaf, in trie order: the chunk, front to back
constexpr u32 STREAM_LDS_LEVELS = 1024;
__global__ __launch_bounds__(256) void stream_chunk_kernel(u32 rank0, u32 nleaf, u32 nlev, const uint4* const* __restrict__ srec_tab, const u64* const* __restrict__ rval_tab,
                                                           const u64* __restrict__ leaf_id, const u64* __restrict__ chunk_off, const u32* __restrict__ kop,
                                                           u8* __restrict__ out) {
    __shared__ const uint4* s_rec[STREAM_LDS_LEVELS];
    const u32 nl = nlev < STREAM_LDS_LEVELS ? nlev : STREAM_LDS_LEVELS;
    for (u32 q = threadIdx.x; q < nl; q += blockDim.x) s_rec[q] = srec_tab[q];
    __syncthreads();
    const u32 r = rank0 + blockIdx.x * blockDim.x + threadIdx.x;  // (leaves rank0 .. nleaf - 1: a slice of the stream)
    if (r >= nleaf) return;
    const u64 id = leaf_id[r];
    u32 l = (u32)(id >> 32), v = (u32)id;
    const u32 ko = kop[r];
    u8* po = out + chunk_off[r] + 2ull * ko;  // the opening tokens end here (written back to front: the walk goes up)
    u8* pc = po;                              // the closing tokens start here
    bool inopen = true, inclose = true;
    while (l >= 1 && (inopen || inclose)) {
        const uint4 rec = (l < STREAM_LDS_LEVELS ? s_rec[l] : srec_tab[l])[v];
        if (inopen) {
            po -= 2;
            const unsigned short head = (unsigned short)('(' | ((0x54474341u >> (8 * (rec.y & 3u))) & 0xFFu) << 8);
            __builtin_memcpy(po, &head, 2);
            if (!(rec.y & SREC_FIRST)) inopen = false;
        }
        if (inclose) {
            const u64 fw = srec_word(rec);
            const u64 f = fw & SREC_FREQ_MASK;
            const u32 lc = (0x4E54474341300000ull >> (8 * ((u32)(fw >> 61) + 2))) & 0xFFu;  // "0ACGTN"[left]
            if (l > 6 && f < (1ull << 32)) {  // at most eight bytes: put together in a register, stored as 4 + 2 + 1 byte pieces
                u64 tok;
                u32 n;
                if (f < 128) { tok = f | 0x80; n = 1; }
                else { const u32 ln = (u32)((64 - __clzll((long long)f) + 7) >> 3); tok = (u64)ln | (f << 8); n = 1 + ln; }
                tok |= ((u64)lc | ((u64)')' << 8)) << (8 * n);
                n += 2;
                if (n & 4) { const u32 wv = (u32)tok; __builtin_memcpy(pc, &wv, 4); pc += 4; tok >>= 32; }
                if (n & 2) { const unsigned short wv = (unsigned short)tok; __builtin_memcpy(pc, &wv, 2); pc += 2; tok >>= 16; }
                if (n & 1) { *pc = (u8)tok; pc += 1; }
            } else {
                pc += put_varint(pc, f);
                if (l <= 6) { *pc++ = 'R'; pc += put_varint(pc, rval_tab[l][v]); }
                *pc++ = (u8)lc;
                *pc++ = ')';
            }
            if (!(rec.y & SREC_LAST)) inclose = false;
        }
        v = rec.x;
        --l;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Arena {
    u8* base = nullptr;
    size_t cap = 0, off = 0;
    template <class T> T* get(size_t n) {
        size_t bytes = ((n ? n : 1) * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) return nullptr;
        T* p = reinterpret_cast<T*>(base + off);
        off += bytes;
        return p;
    }
};

struct LevelHost {
    u32 n = 0;
    u32* slot = nullptr;        // 4 * parent + sym (stream mode, derived handles; null otherwise)
    uint2* pw = nullptr;        // path words (mining)
    u64* kplane = nullptr;      // children directory (struct Kids): 4 planes and 4 counts per 64 nodes, whole tiles
    u32* kcum = nullptr;
    Kids kids() const { Kids k; k.plane = kplane; k.cum = kcum; return k; }
    // mine
    u64* cand_bits = nullptr;   // bit j of word w: node 64w + j is a candidate (one word per wave of 64 nodes)
    u32 ncand = 0, npairs = 0;
    u32* cand_node = nullptr;
    u32* cand_poff = nullptr;
    uint4* crec = nullptr;      // (one sample: the candidates as the advance sweep stored them, see AdvanceOut::crec)
    u32* ids = nullptr;
    u64* freqs = nullptr;
    u32* sub = nullptr;
    // stream
    uint4* srec = nullptr;      // per node: parent, symbol and first / last flags, frequency and left char (keep_kernel)
    u8* clen = nullptr;         // bytes of the node's closing token (without an 'R' part)
    u32* lf = nullptr;          // leaves in the subtree
    u32* top_rank = nullptr;    // levels 1..6: rank of the first leaf, leaves, value of the 'R' token
    u32* top_lf = nullptr;
    u64* rval = nullptr;
};

static inline dim3 grid_for(u64 n, int t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

#define ARENA_GET(var, T, n)                                                                        \
    do {                                                                                            \
        (var) = arena.get<T>(n);                                                                    \
        if (!(var)) return fail(DSM_E_CAPACITY, "device arena exhausted: use a longer prefix or a larger arena_bytes"); \
    } while (0)

static unsigned host_threads() {
    static unsigned n = 0;
    if (!n) {
        const char* e = getenv("DSM_HOST_THREADS");
        long v = e ? atol(e) : 0;
        if (v <= 0) {
            cpu_set_t cs;
            v = sched_getaffinity(0, sizeof cs, &cs) == 0 ? CPU_COUNT(&cs) : 1;
            if (v > 16) v = 16;
        }
        n = (unsigned)(v < 1 ? 1 : v);
    }
    return n;
}

// One emit job: the candidates of one prefix, already in pinned host memory.  Exact entropy
// (metaserver.cpp:366-389), the emin/emax test (:413), order-preserving compaction, delivery.
struct RawBuf {
    void* p = nullptr;
    size_t cap = 0;
    void* ensure(size_t n) {
        if (n > cap) {
            free(p);
            cap = n + n / 4 + 4096;
            p = malloc(cap);
        }
        return p;
    }
    ~RawBuf() { free(p); }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want);
        if (e != hipSuccess) return fail(DSM_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        cap = want;
        return 0;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};
struct DevGrow {  // device buffer that only grows
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(DSM_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        cap = want;
        return 0;
    }
    ~DevGrow() { if (p) (void)hipFree(p); }
};
struct EmitSet {
    DevGrow dev[10];  // the five arrays on the device (they outlive the arena while the copy stream drains them), then what the fill
                      // adds for the host: [5] entropies, [6] verdicts, [7] / [8] offsets relative to the chunks, [9] per-chunk counts
    hipEvent_t ready = nullptr;  // recorded on the copy stream after the last device-to-host copy
    // A large set travels in chunks of consecutive tuples: chunk c is filled, copied and handed to the sink while the
    // following ones are still on their way (the tail of a prefix is one chunk of host work, not the whole set).
    static constexpr int MAX_CHUNKS = EMIT_MAX_CHUNKS;
    int nchunk = 1;
    u32 cb[MAX_CHUNKS + 1] = {0};            // tuple boundaries
    hipEvent_t cready[MAX_CHUNKS] = {nullptr};  // chunk c has landed in pinned memory
    int device = 0;
    PinBuf pin[10];  // [0] / [1] offsets relative to the chunks (chunk c at [cb[c] + c, cb[c + 1] + c]), [2] ids, [3] freqs, [4] paths (device order
                     // = post-order rank), [5] entropies, [6] verdicts, [9] per-chunk counts {dropped, left to the host}
    u64 pb[MAX_CHUNKS + 1] = {0}, qb[MAX_CHUNKS + 1] = {0};  // path byte / pair at which chunk c starts
    RawBuf out[6];   // o_path, o_pair, ent, paths, ids, freqs (kept tuples only)
    u32 nt = 0;
    bool busy = false;
    ~EmitSet() {
        if (ready) (void)hipEventDestroy(ready);
        for (auto e : cready) if (e) (void)hipEventDestroy(e);
    }
};

// Persistent helpers of the emitter thread: a pass over a chunk is split into nth ranges; creating threads per pass would
// put ~30 clone()/mmap() calls per prefix in competition with the HIP runtime's own address-space work.
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::function<void(unsigned)> job;
    unsigned gen = 0, pending = 0;
    bool stop = false;
    void ensure(unsigned n) {  // n - 1 workers, worker k runs range k + 1
        while (th.size() + 1 < n) {
            const unsigned id = (unsigned)th.size() + 1;
            th.emplace_back([this, id] {
                unsigned seen = 0;
                for (;;) {
                    std::function<void(unsigned)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        if (id >= active) { continue; }
                        f = job;
                    }
                    f(id);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) done_cv.notify_all();
                    }
                }
            });
        }
    }
    unsigned active = 0;
    void run(unsigned n, const std::function<void(unsigned)>& f) {  // f(0) .. f(n-1), f(0) on the caller
        ensure(n);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = f; active = n; pending = n - 1; ++gen;
        }
        cv.notify_all();
        f(0u);
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

// One chunk of a prefix's tuples, landed in pinned memory, to the sink.  The entropies, the emin / emax verdicts and the offsets relative
// to the chunk arrived with it (tuple_fill_kernel).  What is left for the host: the exact entropy of the few tuples whose frequencies lie
// beyond the device's tables, with libm as the reference does it, and -- only when tuples were dropped -- moving the kept ones together.
// With nothing dropped (always so with one sample: its LF-step kernel applied the verdict already, see KEEP_FREQS) the batch IS the
// pinned arrays and no pass over the tuples runs here at all.
static int emit_job(HostPool& pool, EmitSet& E, int c, u32 d, double emin, double emax, dsm_tuple_sink sink, void* ctx, u64* n_tuples, u64* n_pairs, double* ms) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const u32 t_lo = E.cb[c], t_hi = E.cb[c + 1];
    const u32 nt = t_hi - t_lo;  // tuples [t_lo, t_hi) of the set
    u32* rel_path = (u32*)E.pin[0].p + t_lo + c;   // nt + 1 entries, the first one 0
    u32* rel_pair = (u32*)E.pin[1].p + t_lo + c;
    u32* ids = (u32*)E.pin[2].p + E.qb[c];
    u64* freqs = (u64*)E.pin[3].p + E.qb[c];
    char* paths = (char*)E.pin[4].p + E.pb[c];
    double* ent = (double*)E.pin[5].p + t_lo;
    u8* keep = (u8*)E.pin[6].p + t_lo;
    const u32* counts = (const u32*)E.pin[9].p + 2 * c;
    u64 ndrop = counts[0];
    if (counts[1]) {  // frequencies of 65536 and more, or a total of 2^20 and more (a few nodes at the top of a pass)
        for (u8* q = keep; (q = (u8*)memchr(q, EV_HOST, (size_t)(keep + nt - q))) != nullptr; ++q) {  // (memchr: a byte loop over four million verdicts is 4 ms)
            const u32 r = (u32)(q - keep);
            const double e = exact_entropy(d, freqs + rel_pair[r], rel_pair[r + 1] - rel_pair[r]);
            ent[r] = e;
            const bool k = !(emax > 0 && (e < emin || e > emax));
            keep[r] = k ? EV_KEEP : EV_DROP;
            if (!k) ++ndrop;
        }
    }
    static const bool tl_on = getenv("DSM_TIMELINE") != nullptr;
    if (tl_on) { char msg[96]; snprintf(msg, sizeof msg, "nt=%u dropped=%llu host=%u", nt, (unsigned long long)ndrop, counts[1]); timeline("  emitter: verdicts", msg); }
    u64 W = nt, QW = rel_pair[nt];
    u32* o_path = rel_path;
    u32* o_pair = rel_pair;
    double* o_ent = ent;
    char* o_paths = paths;
    u32* o_ids = ids;
    u64* o_freqs = freqs;
    constexpr u64 FEW_DROPS = 4096;
    if (ndrop && ndrop <= FEW_DROPS && ndrop * 64 < nt) {
        // A handful of tuples dropped among millions (one sample: the few nodes of four million occurrences and more, which only the
        // host's libm can decide): the runs of kept tuples between them leave as batches of their own, in place -- only the offsets of
        // a run are rebased to its first tuple (8 bytes per tuple instead of moving the ~55 bytes of every tuple behind the first drop).
        std::vector<u32> seg;   // runs [seg[2i], seg[2i+1]) of kept tuples (emit_runs.h)
        kept_runs(keep, nt, EV_DROP, seg);
        const size_t ns = seg.size() / 2;
        std::vector<u32> bp(ns), bq(ns);
        for (size_t i = 0; i < ns; ++i) { bp[i] = rel_path[seg[2 * i]]; bq[i] = rel_pair[seg[2 * i]]; }
        unsigned nth = host_threads();
        if (nt < 65536) nth = 1;
        const u32 per = (nt + nth - 1) / nth;
        auto rebase = [&](unsigned t) {   // entries lo .. hi - 1 of this thread (the last one takes the closing entry nt)
            const u32 lo = t * per < nt ? t * per : nt, hi = t + 1 == nth ? nt + 1 : (lo + per < nt ? lo + per : nt);
            rebase_runs(rel_path, rel_pair, seg, bp, bq, lo, hi);
        };
        pool.run(nth, rebase);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        *ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        for (size_t i = 0; i < ns; ++i) {
            const u32 a = seg[2 * i], n = seg[2 * i + 1] - a;
            *n_tuples += n;
            *n_pairs += rel_pair[a + n];
            if (!sink) continue;
            dsm_tuple_batch bt;
            bt.ntuples = n;
            bt.path_off = rel_path + a; bt.path_bytes = paths + bp[i]; bt.entropy = ent + a;
            bt.pair_off = rel_pair + a; bt.ids = ids + bq[i]; bt.freqs = freqs + bq[i];
            if (sink(ctx, &bt)) return 1;
        }
        return 0;
    }
    if (ndrop) {  // the kept tuples move together (ranges of the chunk by the pool's threads: count, then move)
        unsigned nth = host_threads();
        if (nt < 65536) nth = 1;
        const u32 per = (nt + nth - 1) / nth;
        std::vector<u64> cnt_t(nth + 1, 0), cnt_p(nth + 1, 0), cnt_q(nth + 1, 0);
        auto range = [&](unsigned t, u32& lo, u32& hi) { lo = t * per < nt ? t * per : nt; hi = lo + per < nt ? lo + per : nt; };
        auto pass1 = [&](unsigned t) {
            u32 lo, hi;
            range(t, lo, hi);
            u64 kt = 0, kp = 0, kq = 0;
            for (u32 r = lo; r < hi; ++r)
                if (keep[r] == EV_KEEP) { ++kt; kp += rel_path[r + 1] - rel_path[r]; kq += rel_pair[r + 1] - rel_pair[r]; }
            cnt_t[t + 1] = kt; cnt_p[t + 1] = kp; cnt_q[t + 1] = kq;
        };
        pool.run(nth, pass1);
        for (unsigned t = 0; t < nth; ++t) { cnt_t[t + 1] += cnt_t[t]; cnt_p[t + 1] += cnt_p[t]; cnt_q[t + 1] += cnt_q[t]; }
        W = cnt_t[nth]; QW = cnt_q[nth];
        const u64 PW = cnt_p[nth];
        o_path = (u32*)E.out[0].ensure((W + 1) * 4);
        o_pair = (u32*)E.out[1].ensure((W + 1) * 4);
        o_ent = (double*)E.out[2].ensure(W * 8 + 8);
        o_paths = (char*)E.out[3].ensure(PW + 1);
        o_ids = (u32*)E.out[4].ensure(QW * 4 + 4);
        o_freqs = (u64*)E.out[5].ensure(QW * 8 + 8);
        auto pass2 = [&](unsigned t) {
            u32 lo, hi;
            range(t, lo, hi);
            u64 w = cnt_t[t], pw = cnt_p[t], qw = cnt_q[t];
            for (u32 r = lo; r < hi; ++r) {
                if (keep[r] != EV_KEEP) continue;
                const u32 pb = rel_path[r], pl = rel_path[r + 1] - pb, qb = rel_pair[r], ql = rel_pair[r + 1] - qb;
                o_path[w] = (u32)pw; o_pair[w] = (u32)qw; o_ent[w] = ent[r];
                memcpy(o_paths + pw, paths + pb, pl);
                memcpy(o_ids + qw, ids + qb, (size_t)ql * 4);
                memcpy(o_freqs + qw, freqs + qb, (size_t)ql * 8);
                ++w; pw += pl; qw += ql;
            }
        };
        pool.run(nth, pass2);
        o_path[W] = (u32)PW;
        o_pair[W] = (u32)QW;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    *n_tuples += W;
    *n_pairs += QW;
    if (W && sink) {
        dsm_tuple_batch b;
        b.ntuples = W;
        b.path_off = o_path; b.path_bytes = o_paths; b.entropy = o_ent; b.pair_off = o_pair; b.ids = o_ids; b.freqs = o_freqs;
        if (sink(ctx, &b)) return 1;
    }
    return 0;
}

// Host worker of the wire-stream path: the bytes of prefix k cross PCIe and reach the sink while the GPU already works on prefix
// k+1.  Two device buffers alternate between the prefixes; pieces go through two pinned staging buffers.
struct StreamOut {
    // bytes [off, off + len) of the buffer, written when `ev` has happened; last: the prefix ends here; release: the buffer is free
    // after this job (a prefix's stream leaves in slices: the first ones cross the bus while the last ones are still being written)
    struct Job { int k; u64 total; int tag; u64 off; u64 len; bool last; hipEvent_t ev; bool release; };
    static constexpr int MAX_SLICES = 8;
    hipEvent_t sev[2][MAX_SLICES] = {{nullptr}, {nullptr}};
    u8* buf[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    bool busy[2] = {false, false};
    hipEvent_t ready[2] = {nullptr, nullptr};  // the write kernels of the buffer's prefix have finished
    PinBuf pin[2];
    hipStream_t copy_stream = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    bool stop = false, started = false;
    int err = 0;  // 1: sink failed, 2: HIP error
    int device = 0, next = 0;
    int inflight = 0;  // jobs submitted and not yet delivered (an end-of-prefix job without bytes holds no buffer, but its sink call counts)
    dsm_byte_sink sink = nullptr;
    dsm_prefix_byte_sink psink = nullptr;
    void* ctx = nullptr;
    u64 delivered = 0;
    static constexpr size_t PIECE = 64u << 20;

    int deliver(const Job& j) {
        (void)hipSetDevice(device);
        if (j.total) {
            if (hipStreamWaitEvent(copy_stream, j.ev ? j.ev : ready[j.k], 0) != hipSuccess) return 2;
            const u8* src = buf[j.k] + j.off;
            const u64 np = (j.len + PIECE - 1) / PIECE;
            auto bytes = [&](u64 i) { return (size_t)((j.len - i * PIECE) < PIECE ? (j.len - i * PIECE) : PIECE); };
            if (np && hipMemcpyAsync(pin[0].p, src, bytes(0), hipMemcpyDeviceToHost, copy_stream) != hipSuccess) return 2;
            for (u64 i = 0; i < np; ++i) {
                if (hipStreamSynchronize(copy_stream) != hipSuccess) return 2;  // piece i has landed
                if (i + 1 < np && hipMemcpyAsync(pin[(i + 1) & 1].p, src + (i + 1) * PIECE, bytes(i + 1), hipMemcpyDeviceToHost, copy_stream) != hipSuccess)
                    return 2;
                const u8* d = (const u8*)pin[i & 1].p;
                const int rc = psink ? psink(ctx, j.tag, d, bytes(i)) : (sink ? sink(ctx, d, bytes(i)) : 0);
                if (rc) { (void)hipStreamSynchronize(copy_stream); return 1; }
            }
        }
        if (j.last && psink && psink(ctx, j.tag, nullptr, 0)) return 1;  // end of this prefix
        return 0;
    }
    void loop() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front();
                q.pop_front();
            }
            int rc = err ? 0 : deliver(j);  // after a failure the remaining prefixes are dropped
            {
                std::lock_guard<std::mutex> lk(mu);
                if (rc && !err) err = rc;
                if (!rc) delivered += j.len;
                if (j.total && j.release) busy[j.k] = false;
                --inflight;
            }
            cv.notify_all();
        }
    }
    // a free device buffer of at least `total` bytes (blocks while both are still crossing the bus)
    int acquire(u64 total, int dev, int* k_out) {
        device = dev;
        if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) return fail(DSM_E_HIP, "hipStreamCreate failed");
        if (int rc = pin[0].ensure(PIECE)) return rc;
        if (int rc = pin[1].ensure(PIECE)) return rc;
        std::unique_lock<std::mutex> lk(mu);
        const int k = next;
        cv.wait(lk, [&] { return !busy[k]; });
        lk.unlock();
        if (!ready[k] && hipEventCreateWithFlags(&ready[k], hipEventDisableTiming) != hipSuccess) return fail(DSM_E_HIP, "hipEventCreate failed");
        if (cap[k] < total) {
            if (buf[k]) (void)hipFree(buf[k]);
            buf[k] = nullptr;
            cap[k] = 0;
            const size_t want = (size_t)total + (size_t)(total / 8) + 4096;
            hipError_t e = hipMalloc((void**)&buf[k], want);
            if (e != hipSuccess) return fail(DSM_E_NOMEM, std::string("hipMalloc (wire stream): ") + hipGetErrorString(e));
            cap[k] = want;
        }
        *k_out = k;
        return 0;
    }
    void submit(int k, u64 total, int tag, u64 off, u64 len, bool last, hipEvent_t ev = nullptr, bool release = true) {  // total == 0: nothing below the root, only the end-of-prefix call
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!started) { started = true; th = std::thread([this] { loop(); }); }
            if (total) { busy[k] = true; next = k ^ 1; }
            ++inflight;
            q.push_back(Job{k, total, tag, off, len, last, ev, release});
        }
        cv.notify_all();
    }
    hipEvent_t slice_event(int k, int j) {
        if (!sev[k][j] && hipEventCreateWithFlags(&sev[k][j], hipEventDisableTiming) != hipSuccess) return nullptr;
        return sev[k][j];
    }
    int drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return inflight == 0; });
        const int e = err;
        err = 0;
        return e;
    }
    ~StreamOut() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
        for (int k = 0; k < 2; ++k) {
            if (buf[k]) (void)hipFree(buf[k]);
            if (ready[k]) (void)hipEventDestroy(ready[k]);
            for (auto e : sev[k]) if (e) (void)hipEventDestroy(e);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

// Host worker: emits prefix k while the GPU already expands prefix k+1 (two pinned sets).
struct Emitter {
    EmitSet set[2];
    HostPool pool;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> q;
    bool stop = false, started = false;
    int sink_err = 0;
    u32 d = 1;
    double emin = 0, emax = 0;
    dsm_tuple_sink sink = nullptr;
    void* ctx = nullptr;
    // text mode (dsm_miner_mine_text): the chunks leave the card as the reference server's lines (textemit.h); nothing binary is copied
    dsm_text_sink text_sink = nullptr;
    TextEmit* te = nullptr;
    std::string text_err;
    u64 tuples = 0, pairs = 0;
    double ms = 0;
    int next = 0;

    int text_job(EmitSet& E, u32 t0, u32 t1, u64* n_tuples, u64* n_pairs, double* ms_) {
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        if (!te) te = text_emit_create(E.device);
        if (!te) return 1;
        const char* text = nullptr;
        size_t len = 0;
        int rc = text_emit_chunk(te, t0, t1, (const u32*)E.dev[0].p, (const u32*)E.dev[1].p, (const u32*)E.dev[2].p, (const u64*)E.dev[3].p, (const char*)E.dev[4].p, d,
                                 emin, emax, &text, &len, n_tuples, n_pairs);
        clock_gettime(CLOCK_MONOTONIC, &b);
        *ms_ += (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
        if (rc) { text_err = dsm_last_error(); return 2; }
        if (len && text_sink(ctx, text, len)) return 1;
        return 0;
    }

    void loop() {
        for (;;) {
            int k;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                k = q.front();
                q.pop_front();
            }
            u64 t = 0, pq = 0;
            double m = 0;
            (void)hipSetDevice(set[k].device);
            int rc = 0;
            for (int c = 0; c < set[k].nchunk && !rc; ++c) {
                rc = set[k].cready[c] && hipEventSynchronize(set[k].cready[c]) != hipSuccess ? 1 : 0;  // the chunk has landed (text mode: has been filled)
                timeline("  emitter: chunk landed");
                if (!rc && set[k].cb[c + 1] > set[k].cb[c]) {
                    if (text_sink) rc = text_job(set[k], set[k].cb[c], set[k].cb[c + 1], &t, &pq, &m);
                    else rc = emit_job(pool, set[k], c, d, emin, emax, sink, ctx, &t, &pq, &m);
                }
                timeline("  emitter: chunk through the sink");
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                tuples += t; pairs += pq; ms += m;
                if (rc) sink_err = rc;
                set[k].busy = false;
            }
            cv.notify_all();
        }
    }
    EmitSet& acquire() {  // a free set (blocks while both are being emitted)
        std::unique_lock<std::mutex> lk(mu);
        int k = next;
        cv.wait(lk, [&] { return !set[k].busy; });
        return set[k];
    }
    void submit() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!started) { started = true; th = std::thread([this] { loop(); }); }
            set[next].busy = true;
            q.push_back(next);
            next ^= 1;
        }
        cv.notify_all();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.empty() && !set[0].busy && !set[1].busy; });
    }
    ~Emitter() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
        if (te) text_emit_destroy(te);
    }
};

// ---------------------------------------------------------------------------------------------
// A sample given as an already enumerated trie: the byte stream an (unmodified reference) client sent to the server.
// Parsed on the host with TrieReader's token rules (TrieReader.h:32-106: '(' sym ... varint(freq) ['R' varint(count)]
// leftchar ')', checksum R for depth <= 6) into level arrays, in the order the nodes appear = path order inside a level.
// ---------------------------------------------------------------------------------------------


template <typename P>
__global__ void trie_expand_kernel(u32 F, const u32* __restrict__ rp, const u64* __restrict__ freq, const u8* __restrict__ plv,
                                   const u32* __restrict__ fc, P* __restrict__ valf, u8* __restrict__ pl, u32* __restrict__ tpos,
                                   u32 allowed) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const u32 r = rp[i];
    if (r == DEAD) { valf[i] = 0; pl[i] = 0; return; }
    const u32 full = plv[r];
    valf[i] = (P)freq[r];
    pl[i] = (u8)((full & 0xF0u) | (full & allowed & 15u));  // allowed: 15, one enforced child, or 0 (see ExpandArgs)
    const u32 low = allowed & (0u - allowed);
    tpos[i] = fc[r] + __popc(full & 15u & (low - 1));       // children before the first allowed one keep their slots
}

// emission-side allocations (candidates, tuple assembly) come from their own arena in multi-rank runs, so that the structure
// arena is used identically on every rank and only the emission side can differ (owner-only emission)
#define EARENA_GET(var, T, n)                                                                       \
    do {                                                                                            \
        (var) = ea->template get<T>(n);                                                             \
        if (!(var)) return fail(DSM_E_CAPACITY, "device arena exhausted: use a longer prefix or a larger arena_bytes"); \
    } while (0)

template <typename P>
class Engine {
  public:
    std::vector<const dsm_index*> idx;
    std::vector<const dsm_trie*> tries;  // trie mode: samples are parsed client streams instead of indexes
    bool trie_mode = false;
    dsm_params prm;
    bool stream_mode = false;
    int nlocal = 0, world = 1, rank = 0;
    u32 d = 1;
    hipStream_t st = 0;
    int device = 0;

    // frontier buffers
    u32 Fcap = 0;
    bool multi = false;       // the level exchange goes through the host's all-gather (world > 1, or forced for rehearsals)
    bool owner_mode = false;  // ... or: columns go to the prefix's owner only, which sends back the union's child planes (dsm_gather_fn)
    bool is_owner = true;     // this rank merges the prefix (always, unless owner_mode)
    int owner = 0;
    u8* bc_buf = nullptr;     // owner mode: the broadcast message, 16-byte header + 32 bytes per 64 parents
    u32* lite_slot[2] = {nullptr, nullptr};  // owner mode, clients: links of the levels (ping-pong)
    uint2* sa[2] = {nullptr, nullptr};       // stream mode: parent / symbol / flags of the level being built (ping-pong, see keep_kernel)
    u32 pub_seq = 0;          // sequence number of the last publish kernel
    u32 Seg = 0;              // handles per symbol segment of a record buffer: Fcap rounded up to whole tiles
    u32 Rcap = 0;             // handles of a record buffer = 4 * Seg
    // A level whose records are WIDE (the few at the top of a prefix) numbers its handles with a smaller segment: a wide record is 81
    // bytes with 64-bit positions, the compact one 32 and complete in itself, and the buffers are sized for Fcap compact records -- a
    // level as wide as the capacity is always compact.  A wide level of more than FcapW nodes splits the prefix like any level that
    // does not fit.  (32-bit positions: the 16-byte compact record keeps slots 2, 3 in the wide fields; one segment size.)
    u32 FcapW = 0, SegW = 0, RcapW = 0;
    u32 seg_of(bool compact) const { return compact ? Seg : SegW; }
    u32* d_pub_tot = nullptr; u64* d_pub_cmax = nullptr;  // device side of what publish_kernel hands over
    std::vector<P*> rec[2];     // child records, ping-pong by level
    std::vector<u32*> rp[2];    // record handle per frontier node, ping-pong
    std::vector<u32*> tpos;     // trie mode: handle of the first child of every node of the level being expanded
    std::vector<u64*> splane;   // index mode: per wave of the level being expanded, which children the sample keeps
    u32** d_rp_tab[2] = {nullptr, nullptr};  // device copies of rp[k][*], tpos[*] and splane[*] for the advance kernel
    u32** d_tpos_tab = nullptr;
    u64** d_splane_tab = nullptr;
    bool spec_mode = false;     // single sample: the next level's LF-step launch is queued before the host has seen the level (see ExpandArgs::dyn)
    u32* d_dyn = nullptr;       // [0] width [1] frequency class of the level the last publish kernel announced
    // Packed columns of several local samples are node-major (the words of a node side by side: the merge reads a node's eight words with
    // one 16-byte load) -- up to eight samples: a sample's LF-step kernel then stores its words 2 * nlocal bytes apart, and with 64
    // samples that is one 128-byte line per 2-byte store (measured, 64 samples of 10^6 reads: 2171 ms per pass node-major, 1756 sample-major).
    bool node_major(bool w9) const { return w9 && nlocal > 1 && nlocal <= 8 && !trie_mode; }
    bool dense_mode = true;     // DSM_DENSE=0: the sparse sweep on every level (A/B runs)
    u32 dense_min = 1u << 18;   // DSM_DENSE_MIN: narrowest level the dense sweep takes
    bool pack_columns = true;   // levels whose frequencies are all below 512: one 16-bit column word per node (DSM_PACK=0 turns it off)
    bool batch_mode = true;     // several samples: one launch per level for up to BATCH_MAX of this process's, handles derived in the kernel
    bool self_mode = false;     // = several samples, index mode, batch_mode: no handle tables (see expand_tile, SELF)
    std::vector<u64*> splane2;  // self_mode: second plane buffer per sample (a level reads its parent level's planes while writing its own)
    u8* xsend = nullptr;
    u8* xrecv[2] = {nullptr, nullptr};
    u64 bpr_cap = 0;
    u32 *cnt4 = nullptr, *scan_tmp = nullptr;  // [4][tiles] child counts of the level being advanced -> scanned offsets
    u32* cntraw = nullptr;                     // single sample: the counts as the expand kernel accumulates them (kept zero between levels)
    LfGeometry lfgeo{256, 16};                 // resident workgroups of the LF-step kernel and their waves (expand.hip)
    u16* sinfo = nullptr;
    u16* nT[2] = {nullptr, nullptr};
    u8* samechild = nullptr;
    u64* order[2] = {nullptr, nullptr};
    u16* order16[2] = {nullptr, nullptr};  // d > 13
    u64 *cand_wsum = nullptr, *cand_wscan = nullptr, *scan_tmp64 = nullptr;  // per wave of 64 nodes: candidates | pairs << 32, and their scan
    u64* d_counters = nullptr;
    u32* d_totals = nullptr;
    u64* d_totals64 = nullptr;
    u32* h_totals = nullptr;  // pinned: [0..7] u32 totals, [8..8+MAX_LOCAL) record allocations, [300..] u64 totals
    u64* h_childmax = nullptr;  // pinned: one per rank
    std::vector<void*> owned;
    size_t owned_bytes = 0;   // device bytes behind `owned` (page-rounded)
    Arena arena;
    Arena carena;          // one sample: the candidate records the advance sweep stores (a block that is cut to size after the level)
    bool fold_cands = true;   // DSM_FOLD_CANDS=0: cand_store_kernel on every level (A/B runs)
    Arena earena;          // multi-rank: emission-side allocations
    Arena* ea = nullptr;   // &earena, or &arena in single-process runs
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evpool;
    dsm_stats stats;
    u32 splits = 0;
    Emitter emitter;
    dsm_text_sink text_sink_ = nullptr;  // set by run_many for dsm_miner_mine_text: the tuples leave as text (textemit.h)

    ~Engine() {
        for (void* p : owned) (void)hipFree(p);
        if (h_totals) (void)hipHostFree(h_totals);
        if (h_childmax) (void)hipHostFree(h_childmax);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        {   // nobody may wait on an event of this engine any more
            std::lock_guard<std::mutex> lk(g_expand_chain.mu);
            for (auto& kv : g_expand_chain.dev)
                if (kv.second.owner == this) kv.second = ExpandChain::Link();
        }
        for (hipEvent_t e : evpool) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (fill_done) (void)hipEventDestroy(fill_done);
        for (auto e : chunk_filled) if (e) (void)hipEventDestroy(e);
    }

    template <class T> int dalloc(T*& p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) return fail(DSM_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        owned.push_back(q);
        owned_bytes += ((n ? n : 1) * sizeof(T) + 4095) & ~(size_t)4095;
        p = (T*)q;
        return 0;
    }

    int init(dsm_index* const* ix, int n, const dsm_params& p, bool stream) {
        device = ix[0]->device;
        u64 nsum = 0, nmax = 0;
        for (int k = 0; k < n; ++k) {
            if (ix[k]->device != device) return fail(DSM_E_INVAL, "all local indexes must live on one device");
            idx.push_back(ix[k]);
            nsum += ix[k]->meta.n;
            nmax = ix[k]->meta.n > nmax ? ix[k]->meta.n : nmax;
        }
        return init_common(n, p, stream, nsum, nmax);
    }
    int init_tries(dsm_trie* const* tr, int n, const dsm_params& p) {
        device = tr[0]->device;
        trie_mode = true;
        u64 nsum = 0, nmax = 0;
        for (int k = 0; k < n; ++k) {
            if (tr[k]->device != device) return fail(DSM_E_INVAL, "all tries must live on one device");
            tries.push_back(tr[k]);
            nsum += tr[k]->nodes + 16;
            nmax = tr[k]->nodes + 16 > nmax ? tr[k]->nodes + 16 : nmax;
        }
        return init_common(n, p, false, nsum, nmax);
    }
    // nsum / nmax: total and largest sample size (indexed symbols, or trie nodes): bounds of a frontier level
    int init_common(int n, const dsm_params& p, bool stream, u64 nsum, u64 nmax) {
        memset(&stats, 0, sizeof stats);
        prm = p;
        stream_mode = stream;
        nlocal = n;
        world = p.world_size > 1 ? (int)p.world_size : 1;
        rank = world > 1 ? (int)p.rank : 0;
        // rehearsal aid: a single rank that still drives the whole exchange path (send buffer, callback, status words)
        multi = world > 1 || (p.allgather && getenv("DSM_FORCE_EXCHANGE"));
        if (world > 1 && !p.allgather) return fail(DSM_E_INVAL, "world_size > 1 needs an allgather callback");
        owner_mode = multi && p.owner_mode != 0;
        if (owner_mode) {
            if (!p.gather || !p.bcast) return fail(DSM_E_INVAL, "owner_mode needs the gather and bcast callbacks");
            if ((int)p.owner_rank >= world) return fail(DSM_E_INVAL, "owner_rank >= world_size");
            if (stream || trie_mode) return fail(DSM_E_INVAL, "owner_mode is for mining indexes");
            owner = (int)p.owner_rank;
            is_owner = rank == owner;
        }
        if (rank >= world) return fail(DSM_E_INVAL, "rank >= world_size");
        if (n > MAX_LOCAL) return fail(DSM_E_INVAL, "at most 273 local samples per process");
        d = (u32)(world * nlocal);
        if (d > 273) return fail(DSM_E_INVAL, "too many samples (MAX_READERS 273, metaserver.cpp:19)");
        st = (hipStream_t)p.stream;
        DSM_HIP(hipSetDevice(device));
        size_t free_b = 0, total_b = 0;
        DSM_HIP(hipMemGetInfo(&free_b, &total_b));
        if (const char* e = getenv("DSM_BATCH")) batch_mode = atoi(e) != 0;
        if (const char* e = getenv("DSM_PACK")) pack_columns = atoi(e) != 0;
        if (const char* e = getenv("DSM_DENSE")) dense_mode = atoi(e) != 0;
        if (const char* e = getenv("DSM_DENSE_MIN")) dense_min = (u32)atol(e);
        spec_mode = d == 1 && !trie_mode && !multi;
        if (const char* e = getenv("DSM_SPEC")) spec_mode = spec_mode && atoi(e) != 0;
        if (spec_mode) { if (int rc = dalloc(d_dyn, (size_t)4)) return rc; }
        self_mode = d > 1 && !trie_mode && batch_mode;
        if (owner_mode && !self_mode) return fail(DSM_E_INVAL, "owner_mode needs handles derived in the LF-step kernel (several samples, DSM_BATCH not 0)");
        // A frontier level holds disjoint suffix intervals, so it is never wider than the indexed text; the
        // union over d samples is bounded by the sum.  Size the default budget from that, not from the card.
        const u64 fbound = (world > 1 ? (u64)d * nmax : nsum) + 16;
        u64 budget = p.arena_bytes ? p.arena_bytes : (u64)(free_b * 0.7);
        if (!p.arena_bytes) {
            // 32 bytes per indexed symbol: at 20x coverage a one-letter prefix needs about 12 (frontier buffers for 1.2 % of n
            // nodes, retained level arrays for 0.35 n nodes); a prefix that does not fit is split automatically.  Very large
            // allocations are slow to create (a 190 GB arena takes 4-7 s, 40 GB no measurable time).
            u64 want = (256ull << 20) + 32ull * nsum * (u64)world;
            if (want < budget) budget = want;
        }
        if (budget > free_b) budget = (u64)(free_b * 0.9);
        // bytes per unit of frontier capacity
        const bool small_rec = sizeof(P) == 8 && !trie_mode;   // record buffers sized by the compact record (see FcapW)
        const u64 rec_b = small_rec ? 32 : REC_FIELDS * sizeof(P) + 1;
        u64 perF = (u64)nlocal * (2 * rec_b * 4 + 2 * 4 + 4 + 1)   // rec x2 (four symbol segments each), rp x2, tpos, planes
                   + (u64)nlocal * (sizeof(P) + 1)                             // send
                   + 2ull * d * (sizeof(P) + 1)                                // recv x2
                   + 2 * (2 + 1 + 8) + 1 + 16 + 64 + (d > 13 ? 4ull * d : 0) + (d > 1 ? 8 : 0);
        // Share of the frontier buffers: a third of the budget for one sample (the retained levels need the rest); with several
        // samples the per-slot cost is dominated by the samples' record buffers while a prefix retains about as much as with one
        // sample, so two thirds go to the frontier -- eight 1-Gbase samples then take a one-letter prefix without splitting it.
        // (the default sizing only: an explicit budget is split as before)
        u64 fc = budget * (nlocal > 1 && !p.arena_bytes ? 2 : 1) / 3 / perF;
        if (fc > (1u << 28) - TILE) fc = (1u << 28) - TILE;
        if (fc < 512) return fail(DSM_E_NOMEM, "not enough device memory for the frontier buffers");
        if (fc > fbound) fc = fbound < 1024 ? 1024 : fbound;
        Fcap = (u32)fc;
        bpr_cap = (((u64)nlocal * Fcap * (sizeof(P) + 1) + 15) & ~15ull) + 16;
        if (p.exchange_send && p.exchange_recv && multi) {
            if (p.exchange_bytes < 1024) return fail(DSM_E_INVAL, "exchange buffers too small");
            // caller-owned buffers bound the frontier as well; recv holds 2 * world * exchange_bytes, used as two halves
            u64 cap_slots = (p.exchange_bytes - 32) / ((u64)nlocal * (sizeof(P) + 1));
            if (cap_slots < Fcap) Fcap = (u32)cap_slots;
            bpr_cap = p.exchange_bytes;
            xsend = (u8*)p.exchange_send;
            xrecv[0] = (u8*)p.exchange_recv;
            xrecv[1] = (u8*)p.exchange_recv + (size_t)world * bpr_cap;  // second half: levels alternate
        } else {
            if (int rc = dalloc(xrecv[0], (size_t)world * bpr_cap)) return rc;
            if (int rc = dalloc(xrecv[1], (size_t)world * bpr_cap)) return rc;
            if (multi) { if (int rc = dalloc(xsend, (size_t)bpr_cap)) return rc; }
        }
        // every rank must take the same capacity decisions (a prefix that overflows is split on all ranks or on none):
        // agree on the smallest frontier capacity through the host's all-gather
        if (multi) {
            u64 mine = Fcap, agreed = 0;
            if (int rc = agree_min(mine, &agreed)) return rc;
            Fcap = (u32)agreed;
        }
        // Record handles: four symbol segments of Seg handles, a tile of 256 parents owns 256 handles in each (see the record layout)
        Seg = (Fcap + TILE - 1) / TILE * TILE;
        Rcap = 4 * Seg;
        SegW = Seg; FcapW = Fcap;
        if (small_rec) {
            SegW = (u32)((u64)Seg * 32 / (REC_FIELDS * sizeof(P) + 1)) / TILE * TILE;
            if (SegW < TILE) SegW = TILE;
            FcapW = SegW < Fcap ? SegW : Fcap;
        }
        RcapW = 4 * SegW;
        // (elements of P: the wide layout over RcapW handles, or 32-byte compact records over Rcap)
        const size_t rec_n = std::max(rec_elems<P>(RcapW), small_rec ? (size_t)Rcap * 32 / sizeof(P) : (size_t)0);
        const size_t ntile = Seg / TILE, nwave = ntile * 4;
        const u64 slots = (u64)Fcap * 4;
        for (int s = 0; s < nlocal; ++s) {
            P *a = nullptr, *b = nullptr;
            u32 *r0, *r1, *tp = nullptr;
            u64* pln = nullptr;
            if (!trie_mode) {
                if (int rc = dalloc(a, rec_n)) return rc;
                if (int rc = dalloc(b, rec_n)) return rc;
                if (int rc = dalloc(pln, nwave * (d == 1 ? 8 : 4))) return rc;  // (one sample: whole lines per tile, see ExpandArgs::symbol_phase)
            } else {
                if (int rc = dalloc(tp, (size_t)Fcap)) return rc;
            }
            if (int rc = dalloc(r0, (size_t)Fcap)) return rc;
            if (int rc = dalloc(r1, (size_t)Fcap)) return rc;
            u64* pln2 = nullptr;
            if (self_mode) { if (int rc = dalloc(pln2, nwave * 4)) return rc; }
            splane2.push_back(pln2);
            rec[0].push_back(a); rec[1].push_back(b); rp[0].push_back(r0); rp[1].push_back(r1); tpos.push_back(tp); splane.push_back(pln);
        }
        for (int k = 0; k < 2; ++k) {
            if (int rc = dalloc(d_rp_tab[k], (size_t)nlocal)) return rc;
            DSM_HIP(hipMemcpy(d_rp_tab[k], rp[k].data(), (size_t)nlocal * sizeof(u32*), hipMemcpyHostToDevice));
        }
        if (int rc = dalloc(d_tpos_tab, (size_t)nlocal)) return rc;
        DSM_HIP(hipMemcpy(d_tpos_tab, tpos.data(), (size_t)nlocal * sizeof(u32*), hipMemcpyHostToDevice));
        if (int rc = dalloc(d_splane_tab, (size_t)nlocal)) return rc;
        DSM_HIP(hipMemcpy(d_splane_tab, splane.data(), (size_t)nlocal * sizeof(u64*), hipMemcpyHostToDevice));
        if (int rc = dalloc(cnt4, 5 * ntile + 8)) return rc;   // (fifth row: candidates per tile, one sample)
        if (stream_mode)
            for (int k = 0; k < 2; ++k) if (int rc = dalloc(sa[k], (size_t)Fcap + 64)) return rc;
        if (owner_mode) {
            if (int rc = dalloc(bc_buf, 16 + 32 * nwave + 64)) return rc;
            for (int k = 0; k < 2; ++k) if (int rc = dalloc(lite_slot[k], (size_t)Fcap + 64)) return rc;
        }
        if (d == 1 && !trie_mode) {
            if (int rc = dalloc(cntraw, 5 * ntile + 8)) return rc;
            DSM_HIP(hipMemset(cntraw, 0, (5 * ntile + 8) * sizeof(u32)));
        }
        if (int rc = lf_step_geometry(sizeof(P) == 8, device, &lfgeo)) return rc;  // (all waves of an LF-step launch are resident)
        if (d > 1 || trie_mode) { if (int rc = dalloc(sinfo, (size_t)slots)) return rc; }
        if (int rc = dalloc(scan_tmp, scan_tmp_elems(5 * ntile) + 8)) return rc;
        for (int k = 0; k < 2; ++k) {
            if (int rc = dalloc(nT[k], Fcap)) return rc;
            if (int rc = dalloc(order[k], Fcap)) return rc;
        }
        if (d > 13)
            for (int k = 0; k < 2; ++k)
                if (int rc = dalloc(order16[k], (size_t)Fcap * d)) return rc;
        if (int rc = dalloc(samechild, Fcap)) return rc;
        if (int rc = dalloc(cand_wsum, nwave + 8)) return rc;
        if (int rc = dalloc(cand_wscan, nwave + 8)) return rc;
        if (int rc = dalloc(scan_tmp64, scan_tmp_elems(nwave) + 8)) return rc;
        if (int rc = dalloc(d_counters, (size_t)COUNTER_SHARDS * 8 + KEEP_FREQS / 64)) return rc;
        {   // the keep table behind the counters (see KEEP_FREQS): the emitter's own expression (emit_job), for d = 1
            std::vector<u32> kt(KEEP_FREQS / 32, ~0u);
            if (d == 1 && !trie_mode && prm.emax > 0) {
                const double* terms = term_table();
                const double* logn = logn_table();
                for (u32 f = 0; f < KEEP_FREQS; ++f) {
                    const u64 sumN = 1ull + f;
                    const double sl = f < TERM_TAB ? terms[f] : (double)((u64)f + 1) * log((double)((u64)f + 1)) / LN2;
                    const double e = (sumN < LOGN_TAB ? logn[sumN] : log((double)sumN) / LN2) - sl / (double)sumN;
                    if (e < prm.emin || e > prm.emax) kt[f >> 5] &= ~(1u << (f & 31));
                }
            }
            DSM_HIP(hipMemcpy(d_counters + (size_t)COUNTER_SHARDS * 8, kt.data(), kt.size() * sizeof(u32), hipMemcpyHostToDevice));
        }
        if (int rc = dalloc(d_totals, 8)) return rc;
        if (int rc = dalloc(d_totals64, 4)) return rc;
        if (int rc = dalloc(d_pub_tot, 8)) return rc;
        if (int rc = dalloc(d_pub_cmax, (size_t)(world > 0 ? world : 1))) return rc;
        DSM_HIP(hipHostMalloc((void**)&h_totals, 320 * sizeof(u32)));
        DSM_HIP(hipHostMalloc((void**)&h_childmax, (size_t)(world > 0 ? world : 1) * sizeof(u64)));
        // what this miner took so far (its own count: the card's free memory also moves with the other lanes and ranks of a card)
        const size_t used = owned_bytes;
        u64 arena_b = budget > used ? budget - used : 0;
        const u64 floor_b = p.arena_bytes ? (1u << 20) : (64u << 20);  // an explicit budget is honoured down to 1 MiB
        if (arena_b < floor_b) arena_b = floor_b;
        if (multi) {
            u64 agreed = 0;
            if (int rc = agree_min(arena_b, &agreed)) return rc;
            arena_b = agreed;
        }
        if (int rc = dalloc(arena.base, arena_b)) return rc;
        arena.cap = arena_b;
        ea = &arena;
        if (multi && !stream_mode) {  // 60 % structure (identical on every rank), 40 % emission
            arena.cap = (size_t)(arena_b * 0.6) & ~(size_t)255;
            earena.base = arena.base + arena.cap;
            earena.cap = arena_b - arena.cap;
            ea = &earena;
        }
        if (const char* e = getenv("DSM_FOLD_CANDS")) fold_cands = atoi(e) != 0;
        if (d == 1 && !multi && !stream_mode && !trie_mode && fold_cands) {
            // one sample mined: a sixteenth of the arena for the candidate records (16 B each; at the benchmark size one node in forty is a
            // candidate and a level retains ≈12 B per node: the share is generous, and a level that does not fit is stored the old way)
            carena.cap = (arena.cap / 16) & ~(size_t)255;
            arena.cap -= carena.cap;
            carena.base = arena.base + arena.cap;
            if (const char* e = getenv("DSM_CAND_ARENA")) {   // test hook: a block so small that levels overflow it (they are stored the old way)
                const size_t want = (size_t)atol(e) & ~(size_t)255;
                if (want < carena.cap) carena.cap = want;
            }
        }
        // (timing events without the system-scope fence a default event carries: the records bracket every LF-step launch, and a
        // fence there would write the L2 back twice per level)
        DSM_HIP(hipEventCreateWithFlags(&ev0, hipEventDisableSystemFence));
        DSM_HIP(hipEventCreateWithFlags(&ev1, hipEventDisableSystemFence));
        return 0;
    }

    // min over ranks of one u64, through the exchange buffers and the host's all-gather callback
    int agree_min(u64 mine, u64* out) {
        u8* send = xsend;
        DSM_HIP(hipMemcpyAsync(send, &mine, sizeof(u64), hipMemcpyHostToDevice, st));
        if (prm.allgather(prm.allgather_ctx, send, xrecv[0], sizeof(u64), (void*)st)) return fail(DSM_E_SINK, "allgather callback failed");
        std::vector<u64> all((size_t)world);
        DSM_HIP(hipMemcpyAsync(all.data(), xrecv[0], (size_t)world * sizeof(u64), hipMemcpyDeviceToHost, st));
        DSM_HIP(hipStreamSynchronize(st));
        u64 m = all[0];
        for (u64 v : all) m = v < m ? v : m;
        *out = m;
        return 0;
    }

    Xchg xview(int which, u64 F_, u64 bpr) const {
        Xchg x;
        x.base = xrecv[which];
        x.bpr = bpr;
        x.nlocal = (u32)nlocal;
        x.d = d;
        x.F = F_;
        x.fb = 0;
        x.nm = 0;
        return x;
    }

    // children directory of a level: whole tiles, written by the advance down-sweep of that level
    int alloc_kids(LevelHost& lv) {
        const size_t nw = ((size_t)lv.n + TILE - 1) / TILE * 4;
        ARENA_GET(lv.kplane, u64, 4 * nw);
        ARENA_GET(lv.kcum, u32, 4 * nw);
        return 0;
    }

    hipEvent_t pool_event(size_t k) {
        while (evpool.size() <= k) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
            evpool.push_back(e);
        }
        return evpool[k];
    }

    // Runs one prefix.  mine: tuples to `tsink` (through the emitter thread); stream: wire bytes to `bsink`.
    // emit_lo / emit_hi: only nodes with emit_lo <= depth <= emit_hi are filtered and emitted; expand_cap: nodes at that
    // depth or deeper are not expanded.  Used when a prefix is split because a level did not fit (see MinerT::run_auto).
    // seed / capture: reader-set iteration orders depend on the whole sibling structure above a node, so a sub-prefix
    // run must start from the order its root had in the unsplit trie (captured by the shallow pass of the parent).
    struct NodeOrder {
        u32 depth = 0;
        std::vector<u32> sym;               // capture: symbol of each node at `depth`
        std::vector<std::vector<u16>> ord;  // capture: their orders; seed: ord[0]
    };
    // Owner mode: between a level's gather and the broadcast that answers it the clients of the prefix wait for the owner.  Whatever
    // makes the owner leave in between (a failed call, a sink error, a limit) must still answer, or the clients would wait for ever:
    // the guard broadcasts BC_ABORT -- the same message size as the answer that was due -- when it goes out of scope armed.
    // (A CLIENT that fails outside this path cannot tell the owner: its host must tear the communicator or the process down, see dsmhip.h.)
    struct BcastGuard {
        Engine* e;
        size_t gather_bytes = 0;  // the clients have sent (or will send) their columns of the next level: take them first
        void* gather_into = nullptr;
        size_t bytes = 0;         // the broadcast they wait for; 0: nothing owed
        explicit BcastGuard(Engine* e_) : e(e_) {}
        void arm(size_t b) { gather_bytes = 0; bytes = b; }
        void arm_next(size_t g, void* into, size_t b) { gather_bytes = g; gather_into = into; bytes = b; }
        void disarm() { gather_bytes = 0; bytes = 0; }
        ~BcastGuard() {
            if (!bytes) return;
            if (gather_bytes) (void)e->prm.gather(e->prm.owner_ctx, e->owner, e->xsend, gather_into, gather_bytes, (void*)e->st);
            const u32 hdr[4] = {BC_ABORT, 0, 0, 0};
            if (hipMemcpyAsync(e->bc_buf, hdr, 16, hipMemcpyHostToDevice, e->st) == hipSuccess) (void)hipStreamSynchronize(e->st);
            (void)e->prm.bcast(e->prm.owner_ctx, e->owner, e->bc_buf, bytes, (void*)e->st);
        }
    };
    int run(const char* prefix_c, dsm_tuple_sink tsink, dsm_byte_sink bsink, void* ctx, bool emit = true, u32 emit_lo = 1,
            u32 emit_hi = ~0u, u32 expand_cap = ~0u, const NodeOrder* seed = nullptr, NodeOrder* capture = nullptr, bool count = true) {
        const std::string prefix = prefix_c ? prefix_c : "";
        for (char ch : prefix)
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T')
                return fail(DSM_E_INVAL, "prefix must be over A,C,G,T");  // anything else has an empty LF interval: nothing to send
        DSM_HIP(hipSetDevice(device));
        for (int s_ = 0; s_ < nlocal && !trie_mode; ++s_)
            if (!idx[s_]->dev.blk) return fail(DSM_E_INVAL, "an index of this miner is offloaded: dsm_index_reload first");
        arena.off = 0;
        earena.off = 0;
        carena.off = 0;
        bool emitting = emit;       // cleared when this rank's emission side runs out of memory in a multi-rank run
        bool emit_failed = false;
        std::vector<LevelHost> L;
        L.reserve(512);
        DSM_HIP(hipMemsetAsync(d_counters, 0, (size_t)COUNTER_SHARDS * 8 * sizeof(u64), st));
        if (cntraw) DSM_HIP(hipMemsetAsync(cntraw, 0, (4 * (size_t)(Seg / TILE) + 8) * sizeof(u32), st));  // a run that failed mid-level may have left counts
        DSM_HIP(hipEventRecord(ev0, st));
        size_t nev = 0;

        // ---- level 0: the root (EnumerateQuery::enumerate, EnumerateQuery.cpp:9-37) --------------
        const char* bases = "ACGT";
        for (int s = 0; s < nlocal && trie_mode; ++s) {  // the root of a parsed stream is its node 0
            const u32 zero = 0;
            DSM_HIP(hipMemcpyAsync(rp[0][s], &zero, sizeof(u32), hipMemcpyHostToDevice, st));
            stats.reported += tries[s]->nodes;
        }
        for (int s = 0; s < nlocal && !trie_mode; ++s) {
            const IndexMeta& m = idx[s]->meta;
            P h[REC_FIELDS];
            for (int f = 0; f < REC_FIELDS; ++f) h[f] = 0;
            h[0] = 0;
            h[1] = (P)(m.n - 1);
            u8 hmask = 0;
            int slot = 0;
            for (int a = 0; a < 4; ++a) {
                u64 lo = m.C[(int)bases[a]], cnt = m.codes[(int)bases[a]].count;  // LF(a,-1), LF(a,n-1)
                if (cnt) { h[2 + 2 * slot] = (P)lo; h[3 + 2 * slot] = (P)(lo + cnt - 1); ++slot; hmask |= (u8)(1u << a); }
            }
            for (int f = 0; f < REC_FIELDS; ++f)
                DSM_HIP(hipMemcpyAsync(rec[0][s] + (size_t)f * RcapW, &h[f], sizeof(P), hipMemcpyHostToDevice, st));   // (the root's record is wide)
            DSM_HIP(hipMemcpyAsync(rec[0][s] + (size_t)REC_FIELDS * RcapW, &hmask, 1, hipMemcpyHostToDevice, st));
            const u32 zero = 0;
            DSM_HIP(hipMemcpyAsync(rp[0][s], &zero, sizeof(u32), hipMemcpyHostToDevice, st));
            if (self_mode) {  // the root as "child A of parent 0" of a level above it: slot 0, plane bit 0, handle 0
                const u64 one = 1;
                DSM_HIP(hipMemcpyAsync(splane2[s], &one, sizeof(u64), hipMemcpyHostToDevice, st));
            }
            stats.lf_steps += 8;
            for (int a = 0; a < 4; ++a) stats.rank_ops += 2 * m.lfcost[a];
        }
        {
            LevelHost root;
            root.n = 1;
            ARENA_GET(root.slot, u32, 1);
            ARENA_GET(root.pw, uint2, 1);
            if (int rc = alloc_kids(root)) return rc;
            if (self_mode) DSM_HIP(hipMemsetAsync(root.slot, 0, sizeof(u32), st));
            DSM_HIP(hipMemsetAsync(root.pw, 0, sizeof(uint2), st));
            L.push_back(root);
            u16 rootT = (u16)d;
            DSM_HIP(hipMemcpyAsync(nT[0], &rootT, sizeof(u16), hipMemcpyHostToDevice, st));
            // root reader set: ids inserted 0..d-1 (metaserver.cpp:736-739)
            std::vector<u16> seq(d), ro(d + 1), tmp(so_work_size(d, d));
            for (u32 k = 0; k < d; ++k) seq[k] = (u16)k;
            set_iteration_order<u16>(seq.data(), d, ro.data(), tmp.data(), d);
            if (d <= 13) {
                u64 ord = 0;
                for (u32 k = 0; k < d; ++k) ord |= (u64)ro[k] << (4 * k);
                DSM_HIP(hipMemcpyAsync(order[0], &ord, sizeof(u64), hipMemcpyHostToDevice, st));
            } else {
                DSM_HIP(hipMemcpyAsync(order16[0], ro.data(), (size_t)d * sizeof(u16), hipMemcpyHostToDevice, st));
            }
        }
        // pmax == 1 (sample-specific substrings, BASELINE configs[4]): every printed node has one reader, its single pair needs no
        // order, and nothing else depends on the reader sets' iteration orders -- the order kernels are skipped altogether
        const u32 order_mode = (d < 2 || prm.pmax == 1) ? 0u : (d <= 13 ? 1u : 2u);  // (declared before the level loop: used by seed/capture)
        stats.pair_order_exact = 1;

        // Width of the frequency column of the level about to be exchanged.  Every rank derives it from the same number: the
        // largest frequency any sample has at that level, carried in the previous level's exchange (the root level is wide).
        bool w16 = false;
        bool w9 = false;   // ... and below 512: one 16-bit word per node holds frequency and flags
        int cur = 0;      // ping-pong index of the current level (rec, rp, nT, order)
        int xcur = 0;     // exchange buffer that will receive the current level's children
        u32 F = 1;
        u32 depth = 0;
        // The expand launch of a level is queued as early as possible: for level L+1 right after the synchronisation of
        // level L, ahead of that level's remaining small launches (order, candidate store), so the GPU does not wait for
        // the host to get through them.
        const bool trace_levels = getenv("DSM_TRACE_LEVELS") != nullptr;  // debugging aid: widths of the levels on stderr
        bool fmt_in = false;  // format of the records of the level about to be expanded (the root's record is wide)
        // dynamic: F is not known yet (0 is passed): the kernel takes the width from d_dyn and runs only if the level's class is (w16, w9)
        // (dynamic launches: Fmax bounds the level -- four children per node of the level before it -- so a small level is not
        // swept by the whole resident grid)
        auto launch_expand = [&](u32 F, u32 depth, int cur, int xcur, bool w16, bool w9, const u32* lslot, bool fmt_in, bool dynamic, u64 Fmax = ~0ull) -> int {
            // ---- expand ---------------------------------------------------------------------------
            const u64 slots = (u64)F * 4;
            const u32 fb = w9 ? 1u : (w16 ? 2u : (u32)sizeof(P));
            const u32 colb = w9 ? 2u : fb + 1;  // column bytes per node
            // per rank: 16-byte header (largest child frequency of this level), [nlocal][F] frequencies, [nlocal][F] bytes, padding
            // (packed levels: [nlocal][F] 16-bit words)
            const u64 bpr = (((u64)nlocal * F * colb + 15) & ~15ull) + 16;
            const int nxt = cur ^ 1;
            u8* send = multi ? xsend : xrecv[xcur];
            ExpandArgs ea;
            memset(&ea, 0, sizeof ea);
            ea.F = F; ea.nbp = (F + TILE - 1) / TILE; ea.fmin = prm.fmin; ea.w16 = w9 ? 2u : (w16 ? 1u : 0u);
            // handle spaces: the children's records are compact iff this level is narrow (w16), this level's iff its parent level was (fmt_in)
            ea.seg = seg_of(w16 && !trie_mode); ea.cap = 4 * ea.seg;
            ea.seg_in = seg_of(fmt_in); ea.cap_in = 4 * ea.seg_in;
            const bool nm = node_major(w9);  // node-major packed columns (Xchg::nm)
            ea.cstride = nm ? (u32)nlocal : 1u;
            if (dynamic) {
                ea.dyn = d_dyn;
                ea.dyn_mask = pack_columns ? 3u : 1u;
                ea.dyn_expect = ((w16 ? 0u : 1u) | (w9 ? 0u : 2u)) & ea.dyn_mask;
                ea.fcap = w16 ? Fcap : FcapW;
            }
            unsigned long long* d_childmax = reinterpret_cast<unsigned long long*>(send);  // header, cleared by the previous level's publish kernel
            if (depth < prefix.size()) {
                const char* q = strchr(bases, prefix[depth]);
                ea.allowed = 1u << (q - bases);
                ea.symbol_phase = 0;
            } else {
                ea.allowed = (depth >= prm.maxdepth || depth >= expand_cap) ? 0u : 15u;  // EnumerateQuery.cpp:153
                ea.symbol_phase = 1;
            }
            // `reported` counts a node once however a prefix was split: a run counts the depths it is responsible for
            if (count && depth + 1 >= emit_lo && depth + 1 <= emit_hi) ea.symbol_phase |= 2u;
            if (d == 1 && !trie_mode) {  // one sample: the output predicates of this level's nodes are evaluated by its LF-step kernel
                ea.symbol_phase |= 8u;
                const bool emit_level = !stream_mode && emit && depth >= 1 && depth >= emit_lo && depth <= emit_hi;
                const bool ent_ok = !(prm.emax > 0 && (0.0 < prm.emin - ENT_MARGIN || 0.0 > prm.emax + ENT_MARGIN));
                if (emit_level && depth >= prm.mindepth && prm.pmin <= 1 && ent_ok) ea.symbol_phase |= 4u;
            }
            ea.probe_slot = 1u + 3u * ((u32)(nev / 2) % (u32)((COUNTER_SHARDS - 4) / 3));  // (read by DSM_CLOCK_PROBE builds of expand.hip only)
            hipEvent_t ea0 = pool_event(nev++), ea1 = pool_event(nev++);
            if (!ea0 || !ea1) return fail(DSM_E_HIP, "hipEventCreate failed");
            std::unique_lock<std::mutex> chain_lock(g_expand_chain.mu);
            ExpandChain::Link& link = g_expand_chain.dev[device];
            if (link.last && link.owner != this) DSM_HIP(hipStreamWaitEvent(st, link.last, 0));
            DSM_HIP(hipEventRecord(ea0, st));
            for (int s = 0; s < nlocal && trie_mode; ++s) {  // children, frequency and left char come from the parsed stream
                const dsm_trie* t = tries[s];
                P* cf = reinterpret_cast<P*>(send + XHDR + (size_t)s * F * fb);
                u8* cl = send + XHDR + (size_t)nlocal * F * fb + (size_t)s * F;
                if (depth + 1 < t->level_off.size()) {
                    const u64 o = t->level_off[depth];
                    hipLaunchKernelGGL((trie_expand_kernel<P>), grid_for(F), dim3(256), 0, st, F, rp[cur][s], t->d_freq + o, t->d_pl + o, t->d_fc + o, cf, cl, tpos[s], ea.allowed);
                } else {  // deeper than this sample's trie: it holds none of these nodes
                    DSM_HIP(hipMemsetAsync(cf, 0, (size_t)F * fb, st));
                    DSM_HIP(hipMemsetAsync(cl, 0, (size_t)F, st));
                }
            }
            ExpandBatch eb;
            int nb = 0;
            bool all_one_sb = true;
            for (int s = 0; s < nlocal && !trie_mode; ++s) all_one_sb = all_one_sb && (idx[s]->meta.n >> SB_SHIFT) == 0;
            for (int s = 0; s < nlocal && !trie_mode; ++s) {
                const IndexMeta& m = idx[s]->meta;
                ExpandSample& es = eb.s[nb];
                es.ix = idx[s]->dev;
                es.rp = rp[cur][s]; es.rec = rec[cur][s]; es.out = rec[nxt][s]; es.splane = splane[s];
                es.pplane = nullptr;
                if (self_mode) {  // handles from the level's slots and the planes of the parent level (the two plane buffers alternate)
                    es.rp = lslot;
                    es.splane = (depth & 1) ? splane2[s] : splane[s];
                    es.pplane = (depth & 1) ? splane[s] : splane2[s];
                }
                es.valf = send + XHDR + (nm ? (size_t)s * 2 : (size_t)s * F * (w9 ? 2u : fb));  // this sample's frequency column (packed: its words)
                es.pl = send + XHDR + (size_t)nlocal * F * fb + (size_t)s * F;   // children nibble | left char << 4 (not used when packed)
                for (int c = 0; c < 4; ++c) es.cost[c] = m.lfcost[c];
                for (int c = 0; c < 4; ++c) es.sb.sb0[c] = m.C[(int)(unsigned char)bases[c]];  // superblock 0: nothing before it
                es.costsum_lo = es.costsum_hi = 0;
                for (u32 set = 0; set < 16; ++set) {
                    u64 sum = 0;
                    for (int c = 0; c < 4; ++c) sum += ((set >> c) & 1u) ? m.lfcost[c] : 0u;
                    if (sum > 63) return fail(DSM_E_UNSUPPORTED, "Huffman codes too long for the cost table");
                    if (set < 10) es.costsum_lo |= sum << (6 * set); else es.costsum_hi |= sum << (6 * (set - 10));
                }
                es.access_pack = 0; es.pad = 0;
                for (int c = 0; c < 8; ++c) {
                    const u32 bits = c < m.ncodes ? m.codes[m.code2byte[c]].bits : 0;
                    if (bits > 15) return fail(DSM_E_UNSUPPORTED, "Huffman code longer than 15 bits");
                    es.access_pack |= bits << (4 * c);
                }
                ++nb;
                stats.expand_slots += F;
                stats.expand_column_bytes += (u64)F * colb;
                LfConfig lc;
                lc.wide_pos = sizeof(P) == 8; lc.fmt_in = fmt_in; lc.fmt_out = w16;
                // several samples: a sample holds a fraction of the union level (half with eight 1-Gbase samples, a fifth with 64): on the
                // levels wide enough to keep every wave busy with items, each sample's own nodes are packed into full tiles (expand.hip)
                lc.dense = dense_mode && self_mode && fmt_in && F >= dense_min;
                // record formats: this level's records are compact iff its parent level was narrow (fmt_in), the children's iff this one is
                if (self_mode) {  // several samples: one launch for up to BATCH_MAX of this process's
                    if (nb < BATCH_MAX && s + 1 < nlocal) continue;
                    static const u32 grid_factor = getenv("DSM_BATCH_GRID_FACTOR") ? (u32)atoi(getenv("DSM_BATCH_GRID_FACTOR")) : 1u;
                    lc.one_sb = all_one_sb;
                    lf_step_launch_batch(lc, lfgeo, grid_factor ? grid_factor : 1u, nb, st, eb, ea, d_counters, d_childmax);
                    nb = 0;
                    ++stats.expand_launches;
                    continue;
                }
                nb = 0;
                ea.sb = es.sb;
                for (int c = 0; c < 4; ++c) ea.cost[c] = es.cost[c];
                ea.access_pack = es.access_pack; ea.costsum_lo = es.costsum_lo; ea.costsum_hi = es.costsum_hi;
                lc.one_sb = (m.n >> SB_SHIFT) == 0;
                u32* ecnt = nullptr;  // (round 4: the tile counts of a single sample are counted from its planes by lite_count_kernel, not by atomics here)
                // (a launch queued ahead: Fmax bounds the level -- four children per node of the level before it)
                const u64 tiles_bound = dynamic ? (Fmax == ~0ull ? ~0ull >> 8 : (Fmax + 63) / 64) : ((u64)F + 63) / 64;
                lf_step_launch(lc, lfgeo, tiles_bound, st, idx[s]->dev, rp[cur][s], rec[cur][s], rec[nxt][s], splane[s], ecnt, es.valf, es.pl, ea, d_counters,
                               d_childmax);
                ++stats.expand_launches;
            }
            DSM_HIP(hipEventRecord(ea1, st));
            link.last = ea1; link.owner = this;
            chain_lock.unlock();
            DSM_HIP(hipGetLastError());
            return 0;
        };
        DSM_HIP(hipMemsetAsync(multi ? xsend : xrecv[xcur], 0, XHDR, st));  // later levels: cleared by publish_kernel
        if (int rc = launch_expand(F, depth, cur, xcur, w16, w9, L[0].slot, fmt_in, false)) return rc;
        fmt_in = w16 && !trie_mode;  // the next level's records are compact iff this level is narrow
        BcastGuard owed(this);  // (owner mode, on the owner: the answer to the level's gather, or the final verdict, is still due)
        static const int test_owner_fail = getenv("DSM_TEST_OWNER_FAIL") ? atoi(getenv("DSM_TEST_OWNER_FAIL")) : -1;  // test hook: the owner fails at this depth
        while (true) {
            const u64 slots = (u64)F * 4;
            const u32 fb = w9 ? 1u : (w16 ? 2u : (u32)sizeof(P));
            const u64 bpr = (((u64)nlocal * F * (w9 ? 2u : fb + 1) + 15) & ~15ull) + 16;
            const int nxt = cur ^ 1;
            // ---- exchange: one all-gather per level, or (owner mode) columns to the owner and the union's child planes back -------
            if (multi && !owner_mode) {
                int rc = prm.allgather(prm.allgather_ctx, xsend, xrecv[xcur], (size_t)bpr, (void*)st);
                if (rc) return fail(DSM_E_SINK, "allgather callback failed");
                stats.exchange_bytes_sent += bpr;
                stats.exchange_bytes_received += bpr * (u64)(world - 1);
            }
            const u32 nwv = (F + 63) >> 6;
            const size_t bc_bytes = 16 + (size_t)nwv * 32;
            if (owner_mode) {
                owed.disarm();  // (the gather that was owed is this one; if it fails the communicator is gone and nobody can be told)
                if (prm.gather(prm.owner_ctx, owner, xsend, xrecv[xcur], (size_t)bpr, (void*)st)) return fail(DSM_E_SINK, "gather callback failed");
                if (!is_owner) {
                    // ---- a client of this prefix: wait for the owner's verdict on the level, then only the links of the next one ----
                    stats.exchange_bytes_sent += bpr;
                    if (prm.bcast(prm.owner_ctx, owner, bc_buf, bc_bytes, (void*)st)) return fail(DSM_E_SINK, "bcast callback failed");
                    stats.exchange_bytes_received += bc_bytes;
                    u32 hdr[4] = {0, 0, 0, 0};
                    DSM_HIP(hipMemcpyAsync(hdr, bc_buf, 16, hipMemcpyDeviceToHost, st));
                    DSM_HIP(hipStreamSynchronize(st));
                    if (hdr[0] == BC_CAPACITY) return fail(DSM_E_CAPACITY, "the prefix does not fit the owner's device arena: use a longer prefix or a larger arena_bytes");
                    if (hdr[0] == BC_ABORT) return fail(DSM_E_SINK, "the prefix's owner failed and aborted the prefix (its own error says why)");
                    if (hdr[0] != BC_OK) return fail(DSM_E_HIP, "malformed broadcast from the prefix's owner");
                    const u32 Fn = hdr[1] & 0x3FFFFFFFu;
                    w16 = !(hdr[1] >> 31);
                    w9 = w16 && !((hdr[1] >> 30) & 1u) && pack_columns;
                    stats.union_nodes += depth >= 1 ? F : 0;
                    if (F > stats.max_frontier) stats.max_frontier = F;
                    ++stats.levels;
                    if (trace_levels) fprintf(stderr, "dsm level prefix=%s depth=%u F=%u (client of rank %d)\n", prefix.c_str(), depth, F, owner);
                    if (Fn > (w16 ? Fcap : FcapW)) return fail(DSM_E_CAPACITY, "frontier wider than the device buffers: use a longer prefix or a larger arena_bytes");
                    if (!Fn) break;
                    const u64* planes = reinterpret_cast<const u64*>(bc_buf + 16);
                    const u32 nbp = (F + TILE - 1) / TILE;
                    if (nbp > 1) {
                        hipLaunchKernelGGL(lite_count_kernel, dim3((nbp + 63) / 64), dim3(256), 0, st, planes, nwv, cnt4, nbp, 2u);
                        exclusive_scan<u32, u32>(cnt4, cnt4, (size_t)4 * nbp, scan_tmp, d_totals, st);
                    }
                    hipLaunchKernelGGL(lite_slot_kernel, dim3(nbp), dim3(256), 0, st, planes, F, cnt4, nbp, nbp == 1 ? 1u : 0u, lite_slot[nxt], Fcap);
                    DSM_HIP(hipGetLastError());
                    if (capture && depth + 1 == capture->depth) {  // (which bases continue the prefix: every rank enumerates the same sub-prefixes)
                        std::vector<u32> hs(Fn);
                        DSM_HIP(hipMemcpyAsync(hs.data(), lite_slot[nxt], (size_t)Fn * sizeof(u32), hipMemcpyDeviceToHost, st));
                        DSM_HIP(hipStreamSynchronize(st));
                        capture->sym.clear();
                        capture->ord.clear();
                        for (u32 v = 0; v < Fn; ++v) { capture->sym.push_back(hs[v] & 3u); capture->ord.push_back(std::vector<u16>(1, 0)); }
                    }
                    if (int rc = launch_expand(Fn, depth + 1, nxt, xcur ^ 1, w16, w9, lite_slot[nxt], fmt_in, false)) return rc;
                    fmt_in = w16 && !trie_mode;
                    cur = nxt;
                    xcur ^= 1;
                    F = Fn;
                    ++depth;
                    continue;
                }
                stats.exchange_bytes_received += bpr * (u64)(world - 1);
                owed.arm(bc_bytes);  // from here to the broadcast below every way out answers the clients (BC_ABORT)
                if (test_owner_fail >= 0 && (int)depth == test_owner_fail) return fail(DSM_E_SINK, "injected owner failure (DSM_TEST_OWNER_FAIL)");
                // the owner: whatever this level may allocate must fit before anything is sent back -- a failure later would leave the
                // clients waiting.  (Emission-side allocations only flag a failure, see emit_failed.)
                const size_t wc = (size_t)F * 4 < Fcap ? (size_t)F * 4 : Fcap;
                const size_t need = wc * 12 + 512 + ((wc + TILE - 1) / TILE * 4) * 48 + 4096;
                if (arena.off + need > arena.cap) {
                    const u32 hdr[4] = {BC_CAPACITY, 0, 0, 0};
                    DSM_HIP(hipMemcpyAsync(bc_buf, hdr, 16, hipMemcpyHostToDevice, st));
                    DSM_HIP(hipStreamSynchronize(st));
                    owed.disarm();
                    if (prm.bcast(prm.owner_ctx, owner, bc_buf, bc_bytes, (void*)st)) return fail(DSM_E_SINK, "bcast callback failed");
                    return fail(DSM_E_CAPACITY, "device arena exhausted: use a longer prefix or a larger arena_bytes");
                }
            }
            Xchg x = xview(xcur, F, bpr);
            x.fb = fb;
            x.nm = node_major(w9) ? 1u : 0u;
            // ---- union frontier of the next level -------------------------------------------------
            LevelHost& me = L[depth];
            LevelHost child;
            // the arena hands out memory past `off`; the new level's arrays are claimed after Fn is known, so
            // the down-sweep writes into a provisional window that is then committed
            if (stream_mode && depth >= 1) {  // this level's own frequencies / left chars complete its records for the wire stream
                ARENA_GET(me.srec, uint4, F);
                ARENA_GET(me.clen, u8, F);
                hipLaunchKernelGGL((keep_kernel<P>), grid_for(F), dim3(256), 0, st, F, x, sa[cur], me.srec, me.clen);
            }
            const bool emit_here = !stream_mode && emitting && depth >= 1 && depth >= emit_lo && depth <= emit_hi;
            bool filtered = false;
            if (emit_here) {
                int erc = emit_alloc(me, F);
                if (erc == DSM_E_CAPACITY && multi) { emit_failed = true; emitting = false; }  // agreed on at the end of the prefix
                else if (erc) return erc;
                else filtered = true;
            }
            const size_t mark2 = arena.off;
            // What a level retains per node: the links (4 * parent + symbol) where something reads them -- the wire stream, handles
            // derived inside the LF-step kernel -- and the path words when tuples are mined.  With both, the words follow the links,
            // whose length only the device knows when the sweep runs: the window is sized for the widest level possible here.
            const bool keep_slot = self_mode, keep_pw = !stream_mode;  // (stream mode: records, completed at the node's own level)
            const size_t wcap = (size_t)F * 4 < Fcap ? (size_t)F * 4 : Fcap;
            u8* window = arena.get<u8>(wcap * ((keep_slot ? 4 : 0) + (keep_pw ? 8 : 0)) + 512);
            if (!window) return fail(DSM_E_CAPACITY, "device arena exhausted: use a longer prefix or a larger arena_bytes");
            u32* new_slot2 = keep_slot ? reinterpret_cast<u32*>(window) : nullptr;
            const u32 nbp = (F + TILE - 1) / TILE;  // tiles of this level
            AdvanceOut ao;
            memset(&ao, 0, sizeof ao);
            ao.slot = new_slot2; ao.nT = nT[nxt]; ao.samechild = samechild;
            if (stream_mode) ao.sa = sa[nxt];
            if (keep_pw) {
                ao.pw = reinterpret_cast<uint2*>(window);  // (behind the links: the kernels place it, see pw_after_slot)
                ao.pw_after_slot = keep_slot ? 1u : 0u;
                ao.parent_pw = me.pw;
                ao.plevel = depth;
            }
            ao.parent_nT = nT[cur]; ao.nlocal = (u32)nlocal; ao.rank = (u32)rank;
            ao.cap = (u32)((size_t)F * 4 < Fcap ? (size_t)F * 4 : Fcap);
            ao.seg = seg_of(w16 && !trie_mode);  // (of the records this level's LF-step launch wrote: the children's)
            ao.h_totals = d_pub_tot;
            ao.rp = d_rp_tab[nxt];
            ao.rp0 = rp[nxt][0];
            ao.tpos = trie_mode ? d_tpos_tab : nullptr;
            ao.splane = d_splane_tab;
            ao.rp_index = self_mode ? 0u : 1u;
            ao.kcum = me.kcum; ao.cnt4 = cnt4; ao.nbp = nbp; ao.kshift = 2;
            ao.width = d_totals;  // (levels of several tiles: the scan's grand total)
            const bool merged = d > 1 || trie_mode;  // the union of several columns (a parsed stream is treated alike)
            if (merged && nbp == 1) {  // a single tile evaluates the columns itself
                ao.eval = 1; ao.kplane_w = me.kplane;
            } else if (merged) {
                hipLaunchKernelGGL((advance_reduce_kernel<P>), dim3(nbp), dim3(256), 0, st, x, sinfo, me.kplane, cnt4, nbp);
                ao.kplane = me.kplane; ao.sinfo = sinfo;
            } else {                   // one sample: the union trie is its trie, the expand kernel wrote planes and tile counts
                ao.kplane = splane[0]; ao.kplane_w = me.kplane; ao.single = 1;
                ao.kshift = 3;
            }
            static const bool lean_sweep = !(getenv("DSM_LEAN_ADVANCE") && atoi(getenv("DSM_LEAN_ADVANCE")) == 0);
            const bool lean = lean_sweep && !merged && nbp > 1 && keep_pw && !keep_slot && !stream_mode;   // advance_single_kernel takes the level
            // One sample: the level's candidates are stored by the sweep itself.  Their number is known only after it: the records go
            // to a block of their own arena, taken as large as the level could need and cut to size after the synchronisation.
            bool fold = false;
            u32 crec_cap = 0;
            if (lean && filtered && carena.cap) {
                const size_t room = (carena.cap - carena.off) / sizeof(uint4);
                crec_cap = (u32)(room < (size_t)F ? room : (size_t)F);
                fold = crec_cap > 0;
            }
            if (nbp > 1) {
                // one sample: the tile counts from the planes its LF-step kernel wrote (that kernel added them up with four atomics per
                // tile of 64 nodes before: 470 K memory-side atomics per launch of the wide levels); with them the candidates per tile
                if (!merged) hipLaunchKernelGGL(lite_count_kernel, dim3((nbp + 63) / 64), dim3(256), 0, st, splane[0], (F + 63) >> 6, cntraw, nbp, 3u, fold ? 1u : 0u);
                exclusive_scan<u32, u32>(merged ? cnt4 : cntraw, cnt4, (size_t)(fold ? 5 : 4) * nbp, scan_tmp, d_totals, st);
            }
            // ---- the output predicates for the nodes of THIS level (their children are known now) ride in the wave sweep; the scan of
            // the candidate counts is queued ahead of the wait ----
            const bool fused_filter = filtered && nbp > 1;
            if (fused_filter) {
                ao.fa = filter_args(F, depth, order_mode); ao.candbits = me.cand_bits; ao.wsum = cand_wsum;
                if (ao.single) ao.cand_copy = 1;  // (the level's LF-step kernel decided: the sweep only moves the words into place)
                else ao.filter_on = 1;
            }
            if (fold) { ao.crec = reinterpret_cast<uint4*>(carena.base + carena.off); ao.crec_cap = crec_cap; }
            if (nbp == 1) hipLaunchKernelGGL((advance_down_kernel<P>), dim3(1), dim3(256), 0, st, x, ao);
            else if (lean) hipLaunchKernelGGL((advance_single_kernel<P>), dim3((nbp + 3) / 4), dim3(256), 0, st, x, ao);
            else hipLaunchKernelGGL((advance_wave_kernel<P>), dim3(nbp), dim3(256), 0, st, x, ao);
            if (filtered) {
                if (int erc = emit_filter(me, F, depth, x, cur, order_mode, !fused_filter, fold)) return erc;
            }
            {
                PublishArgs pa;
                memset(&pa, 0, sizeof pa);
                pa.total = nbp == 1 ? d_pub_tot : d_totals;  // new level's width: from the single tile, or the scan's total
                pa.cmax_base = x.base; pa.cmax_bpr = x.bpr; pa.cmax_world = (u32)world;
                if (filtered) pa.cand = d_totals64;
                if (fold) {  // the scan ran over the child counts and, behind them, the candidate counts: the width is what it had reached there
                    pa.total = cnt4 + (size_t)4 * nbp;
                    pa.grand = d_totals;
                    pa.cand = nullptr;
                }
                pa.clear = reinterpret_cast<u32*>(multi ? xsend : xrecv[xcur ^ 1]);  // where the next level's expand reports its child maximum
                pa.packet = reinterpret_cast<uint4*>(h_totals + 304); pa.seq = ++pub_seq;
                pa.next = spec_mode ? d_dyn : nullptr;
                pa.bc_header = owner_mode ? reinterpret_cast<uint4*>(bc_buf) : nullptr;
                hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, pa);
            }
            if (owner_mode) {  // the union's child planes of this level follow the header: what a client needs to go on
                DSM_HIP(hipMemcpyAsync(bc_buf + 16, me.kplane, (size_t)nwv * 32, hipMemcpyDeviceToDevice, st));
                owed.disarm();
                if (prm.bcast(prm.owner_ctx, owner, bc_buf, bc_bytes, (void*)st)) return fail(DSM_E_SINK, "bcast callback failed");
                stats.exchange_bytes_sent += bc_bytes * (u64)(world - 1);
            }
            // ---- the next level's LF-step launch goes out now, sized on the device, assuming the level is of this level's frequency
            // class (the largest frequency only falls with depth: a prefix changes class twice); the host catches up below ----
            const bool spec = spec_mode;
            const bool spec_w16 = w16, spec_w9 = w9;
            if (spec) { if (int rc = launch_expand(0, depth + 1, nxt, xcur ^ 1, spec_w16, spec_w9, new_slot2, fmt_in, true, (u64)F * 4)) return rc; }
            u32 pk[4];
            {   // the publish kernel is the last work queued: its packet in pinned memory is this level's completion.  Spinning on
                // it returns a few microseconds after the store; a stream synchronisation wakes the thread later.
                volatile u32* fl = h_totals + 304;
                u32 spins = 0;
                while (*fl != pub_seq) {
                    if ((++spins & 0xFFFFu) == 0 && hipStreamQuery(st) != hipErrorNotReady) {  // finished (or failed) without the packet?
                        DSM_HIP(hipStreamSynchronize(st));
                        if (*fl != pub_seq) return fail(DSM_E_HIP, "publish kernel did not report");
                        break;
                    }
                    __builtin_ia32_pause();
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int q = 0; q < 4; ++q) pk[q] = fl[q];
            }
            const u32 Fn = pk[1] & 0x3FFFFFFFu;
            w16 = !(pk[1] >> 31) && !trie_mode;  // the next level's frequencies all fit 16 bits (parsed streams stay wide)
            w9 = w16 && !((pk[1] >> 30) & 1u) && pack_columns;  // ... and nine: frequency and flags share a 16-bit word
            // did the launch queued ahead run?  (the kernel tested the same two words the packet carries)
            const bool spec_hit = spec && Fn > 0 && Fn <= (w16 ? Fcap : FcapW) && w16 == spec_w16 && (!pack_columns || ((pk[1] >> 30) & 1u) == (spec_w9 ? 0u : 1u));
            if (spec && !spec_hit) --stats.expand_launches;  // (the launch queued ahead found no level, or another class, and did nothing)
            h_totals[300] = pk[2]; h_totals[301] = pk[3];  // candidate totals of this level (read by emit_store)
            // (a level of wide records has the smaller capacity FcapW; every rank, and in owner mode every client, sees the same two numbers)
            if (Fn > (w16 ? Fcap : FcapW)) return fail(DSM_E_CAPACITY, "frontier wider than the device buffers: use a longer prefix or a larger arena_bytes");
            if (owner_mode) {  // what the clients do next: send the next level's columns and wait for its answer, or wait for the final verdict
                if (!Fn) owed.arm(16);
                else owed.arm_next((size_t)((((u64)nlocal * Fn * (w9 ? 2u : (w16 ? 3u : (u32)sizeof(P) + 1)) + 15) & ~15ull) + 16), xrecv[xcur ^ 1], 16 + (size_t)((Fn + 63) >> 6) * 32);
            }
            // commit the provisional window at its real size
            arena.off = mark2;
            child.n = Fn;
            if (Fn) {
                if (keep_slot) child.slot = arena.get<u32>(Fn);   // same address as new_slot2
                if (keep_pw) child.pw = arena.get<uint2>(Fn);     // the window's start, or right behind the links (256-byte granules, as the kernels assume)
                if (int rc = alloc_kids(child)) return rc;
                if (spec_hit) {
                    stats.expand_slots += Fn;
                    stats.expand_column_bytes += (u64)Fn * (w9 ? 2u : (w16 ? 3u : (u32)sizeof(P) + 1));
                } else {
                    if (int rc = launch_expand(Fn, depth + 1, nxt, xcur ^ 1, w16, w9, child.slot, fmt_in, false)) return rc;  // w16, w9 already describe the next level
                }
                fmt_in = w16 && !trie_mode;
                // orders are only needed by a rank that emits this prefix (and by the shallow pass that captures them)
                if (!(emit || capture)) {}
                else if (order_mode == 1)
                    hipLaunchKernelGGL((order_kernel<P>), grid_for(F), dim3(256), 0, st, F, x, nT[cur], order[cur], me.kids(), order[nxt]);
                else if (order_mode == 2 && d <= 64)
                    hipLaunchKernelGGL((order_big_kernel<P, 64>), grid_for(F, 64), dim3(64), 0, st, F, x, nT[cur], order16[cur], me.kids(), order16[nxt]);
                else if (order_mode == 2)
                    hipLaunchKernelGGL((order_big_kernel<P, 273>), grid_for(F, 64), dim3(64), 0, st, F, x, nT[cur], order16[cur], me.kids(), order16[nxt]);
            }
            if (Fn && order_mode && seed && depth + 1 == seed->depth) {  // the sub-prefix root keeps its order from the unsplit trie
                const std::vector<u16>& so = seed->ord[0];
                if (order_mode == 1) {
                    u64 ord = 0;
                    for (size_t k = 0; k < so.size(); ++k) ord |= (u64)so[k] << (4 * k);
                    DSM_HIP(hipMemcpyAsync(order[nxt], &ord, sizeof(u64), hipMemcpyHostToDevice, st));
                } else {
                    DSM_HIP(hipMemcpyAsync(order16[nxt], so.data(), so.size() * sizeof(u16), hipMemcpyHostToDevice, st));
                }
            }
            if (Fn && capture && depth + 1 == capture->depth) {
                std::vector<u32> hs(Fn);
                std::vector<u16> hn(Fn);
                if (child.slot) DSM_HIP(hipMemcpyAsync(hs.data(), child.slot, (size_t)Fn * sizeof(u32), hipMemcpyDeviceToHost, st));
                else if (stream_mode) {  // the symbol sits in the second word of the sweep's (parent, flags) pair
                    std::vector<uint2> hr(Fn);
                    DSM_HIP(hipMemcpyAsync(hr.data(), sa[nxt], (size_t)Fn * sizeof(uint2), hipMemcpyDeviceToHost, st));
                    DSM_HIP(hipStreamSynchronize(st));
                    for (u32 v = 0; v < Fn; ++v) hs[v] = hr[v].y & 3u;
                } else {  // the last symbol of the node's path word
                    std::vector<uint2> hp(Fn);
                    DSM_HIP(hipMemcpyAsync(hp.data(), child.pw, (size_t)Fn * sizeof(uint2), hipMemcpyDeviceToHost, st));
                    DSM_HIP(hipStreamSynchronize(st));
                    for (u32 v = 0; v < Fn; ++v) hs[v] = (hp[v].y >> (2 * (depth % PW_CHUNK))) & 3u;
                }
                DSM_HIP(hipMemcpyAsync(hn.data(), nT[nxt], (size_t)Fn * sizeof(u16), hipMemcpyDeviceToHost, st));
                DSM_HIP(hipStreamSynchronize(st));
                capture->sym.clear();
                capture->ord.clear();
                for (u32 v = 0; v < Fn; ++v) {
                    capture->sym.push_back(hs[v] & 3);
                    std::vector<u16> o(d == 1 ? 1 : hn[v]);  // (a single sample keeps no reader counts)
                    if (order_mode == 1) {
                        u64 ord = 0;
                        DSM_HIP(hipMemcpy(&ord, order[nxt] + v, sizeof(u64), hipMemcpyDeviceToHost));
                        for (u32 k = 0; k < hn[v]; ++k) o[k] = (u16)((ord >> (4 * k)) & 15);
                    } else if (order_mode == 2) {
                        DSM_HIP(hipMemcpy(o.data(), order16[nxt] + (size_t)v * d, (size_t)hn[v] * sizeof(u16), hipMemcpyDeviceToHost));
                    }
                    capture->ord.push_back(o);
                }
            }
            if (filtered) {  // the candidates of this level: totals arrived with the synchronisation above
                int erc = emit_store(me, F, depth, x, cur, order_mode, fold ? crec_cap : 0u);
                if (erc == DSM_E_CAPACITY && multi) { emit_failed = true; emitting = false; }
                else if (erc) return erc;
            }
            DSM_HIP(hipGetLastError());
            if (trace_levels) fprintf(stderr, "dsm level prefix=%s depth=%u F=%u\n", prefix.c_str(), depth, F);
            if (pend.submit && (F >= 200000u || depth >= 24)) { if (int rc = flush_pending()) return rc; }  // the previous prefix's tuples may leave now
            stats.union_nodes += depth >= 1 ? F : 0;
            if (F > stats.max_frontier) stats.max_frontier = F;
            ++stats.levels;
            if (!Fn) break;
            L.push_back(child);
            cur = nxt;
            xcur ^= 1;
            F = Fn;
            ++depth;
            if (L.size() > 60000) return fail(DSM_E_CAPACITY, "trie deeper than 60000 levels");
        }
        const u32 nlev = (u32)L.size();  // levels 0..nlev-1, level l holds the nodes of depth l
        timeline("levels done", prefix.c_str());
        if (int rc = flush_pending()) return rc;  // (a prefix that never got wide or deep)

        bool ready = false;
        if (stream_mode) {
            if (emit) { if (int rc = finish_stream(L, nlev, bsink, ctx)) return rc; }
        } else {
            if (emitting) {
                int rc = finish_mine(L, nlev, tsink, ctx, &ready);
                if (rc == DSM_E_CAPACITY && multi) { emit_failed = true; ready = false; }
                else if (rc) return rc;
            }
            if (owner_mode) {  // only the owner emits: its word on the emission side reaches the clients with one last broadcast
                u32 hdr[4] = {emit_failed ? BC_CAPACITY : BC_OK, 0, 0, 0};
                if (is_owner) {
                    DSM_HIP(hipMemcpyAsync(bc_buf, hdr, 16, hipMemcpyHostToDevice, st));
                    DSM_HIP(hipStreamSynchronize(st));
                }
                owed.disarm();
                if (prm.bcast(prm.owner_ctx, owner, bc_buf, 16, (void*)st)) return fail(DSM_E_SINK, "bcast callback failed");
                if (!is_owner) {
                    DSM_HIP(hipMemcpyAsync(hdr, bc_buf, 16, hipMemcpyDeviceToHost, st));
                    DSM_HIP(hipStreamSynchronize(st));
                }
                if (hdr[0] == BC_ABORT) return fail(DSM_E_SINK, "the prefix's owner failed and aborted the prefix (its own error says why)");
                if (hdr[0] != BC_OK) return fail(DSM_E_CAPACITY, "device arena exhausted on the prefix's owner: use a longer prefix or a larger arena_bytes");
            } else if (multi) {  // every rank learns whether some rank's emission side overflowed: split together or not at all
                u64 ok = emit_failed ? 0 : 1, all_ok = 0;
                if (int rc = agree_min(ok, &all_ok)) return rc;
                if (!all_ok) return fail(DSM_E_CAPACITY, "device arena exhausted on a rank: use a longer prefix or a larger arena_bytes");
            }
            if (ready) pend.submit = true;   // (the set is prepared; flush_pending hands it over once the next prefix is under way)
            else pend.E = nullptr;
            static const bool defer = !(getenv("DSM_DEFER_EMIT") && atoi(getenv("DSM_DEFER_EMIT")) == 0);  // (0: at once, for A/B runs)
            if (!defer) { if (int rc = flush_pending()) return rc; }
        }

        DSM_HIP(hipEventRecord(ev1, st));
        timeline("finish queued", prefix.c_str());
        DSM_HIP(hipStreamSynchronize(st));
        timeline("device done", prefix.c_str());
        float ms = 0;
        DSM_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        stats.device_ms += ms;
        float expand_ms = 0;
        for (size_t k = 0; k + 1 < nev; k += 2) {
            float t = 0;
            DSM_HIP(hipEventElapsedTime(&t, evpool[k], evpool[k + 1]));
            expand_ms += t;
        }
        stats.expand_ms += expand_ms;
        u64 hc[NCOUNTERS] = {0, 0, 0, 0, 0, 0};
        {
            std::vector<u64> sh((size_t)COUNTER_SHARDS * 8);
            DSM_HIP(hipMemcpy(sh.data(), d_counters, sh.size() * sizeof(u64), hipMemcpyDeviceToHost));
            for (int k = 0; k < COUNTER_SHARDS; ++k)
                for (int c = 0; c < NCOUNTERS; ++c) hc[c] += sh[(size_t)k * 8 + c];
#ifdef DSM_CLOCK_PROBE
            if (sh[7]) {  // (expand.hip built with the same flag: per launch the earliest / latest wave start and end, and their sums)
                double dur = 0, ramp = 0, tail = 0, meanbusy = 0;
                int nl = 0;
                const double W = (double)lfgeo.blocks * lfgeo.waves_per_block;
                for (int k = 1; k + 2 < COUNTER_SHARDS; k += 3) {
                    const u64 s0 = ~sh[(size_t)k * 8 + 6], s9 = sh[(size_t)k * 8 + 7], e0 = ~sh[(size_t)(k + 1) * 8 + 6], e9 = sh[(size_t)(k + 1) * 8 + 7];
                    if (!s9 || e9 - s0 < 5000) continue;   // launches of 50 us or more
                    const double ms = (double)sh[(size_t)(k + 2) * 8 + 6] / W, me = (double)sh[(size_t)(k + 2) * 8 + 7] / W;  // (low 32 bits of the ticks, summed)
                    dur += (double)(e9 - s0); ramp += (double)(s9 - s0); tail += (double)(e9 - e0); meanbusy += me - ms; ++nl;
                }
                fprintf(stderr, "clock probe: %.1f MHz; %d launches >= 50 us: duration %.2f ms, first-to-last wave start %.2f ms, first-to-last wave end %.2f ms, mean wave busy %.2f ms\n",
                        (double)sh[6] / (double)sh[7] * 100.0, nl, dur * 1e-5, ramp * 1e-5, tail * 1e-5, meanbusy * 1e-5);
            }
#endif
        }
        stats.reported += hc[0];
        stats.lf_steps += hc[1];
        stats.rank_ops += hc[2];
        stats.index_lines += hc[3];
        stats.record_bytes += hc[4];
        stats.records_read += hc[5];
        return 0;
    }

    // ---- output predicates and candidate store for the nodes of one level (their children are known) ----
    // Two halves: the filter and its scan are queued before the host waits for the width of the next level, so the one
    // synchronisation per level also returns the candidate totals; the store follows once they are known.
    FilterArgs filter_args(u32 F, u32 depth, u32 order_mode) const {
        FilterArgs fa;
        fa.F = F; fa.depth = depth; fa.d = d; fa.pmin = prm.pmin; fa.pmax = prm.pmax; fa.mindepth = prm.mindepth;
        fa.emin = prm.emin; fa.emax = prm.emax; fa.exact_order = order_mode;
        return fa;
    }
    int emit_alloc(LevelHost& me, u32 F) {  // before the provisional window of the next level is taken from the arena
        EARENA_GET(me.cand_bits, u64, ((size_t)F + 63) / 64);
        return 0;
    }
    int emit_filter(LevelHost& me, u32 F, u32 depth, const Xchg& xp, int cur, u32 order_mode, bool run_kernel, bool stored = false) {
        const FilterArgs fa = filter_args(F, depth, order_mode);
        if (run_kernel)  // (single-tile levels; larger ones evaluate the predicates inside the advance sweep)
            hipLaunchKernelGGL((filter_kernel<P>), grid_npt(F), dim3(256), 0, st, fa, xp, nT[cur], me.kids(), samechild, me.cand_bits, cand_wsum);
        // (stored: the sweep stored the candidates itself, their places from the scan of the tile counts)
        if (!stored) exclusive_scan<u64, u64>(cand_wsum, cand_wscan, ((size_t)F + 63) / 64, scan_tmp64, d_totals64, st);
        return 0;
    }
    // stored_cap > 0: the advance sweep stored the level's candidates as records at the top of the candidate arena, up to that many
    int emit_store(LevelHost& me, u32 F, u32 depth, const Xchg& xp, int cur, u32 order_mode, u32 stored_cap = 0) {  // after the level's synchronisation
        const FilterArgs fa = filter_args(F, depth, order_mode);
        u64 tot = 0;
        memcpy(&tot, h_totals + 300, sizeof tot);
        me.ncand = (u32)(tot & 0xFFFFFFFFu);
        me.npairs = (u32)(tot >> 32);
        if (stored_cap && me.ncand <= stored_cap) {   // the block is cut to the records the level has
            if (me.ncand) {
                me.crec = carena.get<uint4>(me.ncand);   // (cap and offsets are multiples of the allocation granule: what fitted the block fits here)
                if (!me.crec) return fail(DSM_E_HIP, "candidate arena: the block that held the records cannot be claimed");
            }
            stats.candidates += me.ncand;
            return 0;
        }
        if (stored_cap)   // more candidates than the block could take (its records beyond were not written): the old way, from the words
            exclusive_scan<u64, u64>(cand_wsum, cand_wscan, ((size_t)F + 63) / 64, scan_tmp64, d_totals64, st);
        if (me.ncand) {
            u32 nc = me.ncand;
            me.ncand = 0;  // stays 0 if the store does not fit: the level then has no usable candidates
            EARENA_GET(me.cand_node, u32, nc);
            EARENA_GET(me.cand_poff, u32, nc);
            EARENA_GET(me.ids, u32, me.npairs);
            EARENA_GET(me.freqs, u64, me.npairs);
            me.ncand = nc;
            // (tried on a side stream next to the following level's LF-step launch -- nothing after it reads the store: the event
            // waits between the streams cost more than the overlap returned, device time +7 ms per pass)
            hipLaunchKernelGGL((cand_store_kernel<P>), grid_npt(F), dim3(256), 0, st, fa, xp, nT[cur], order[cur], order16[cur], me.cand_bits, cand_wscan,
                               me.cand_node, me.cand_poff, me.ids, me.freqs);
        }
        stats.candidates += me.ncand;
        return 0;
    }

    // ---- mine: post-order ranks of the candidates, tuple assembly, exact entropy on the host ------
    // Prepares everything up to the copies into the pinned set; the caller submits the set to the emitter (*ready).
    int finish_mine(std::vector<LevelHost>& L, u32 nlev, dsm_tuple_sink sink, void* ctx, bool* ready) {
        u64 ncand_total = 0;
        for (u32 l = 1; l < nlev; ++l) ncand_total += L[l].ncand;
        if (ncand_total == 0) return 0;
        if (ncand_total > 0xFFFFFFF0ull) return fail(DSM_E_CAPACITY, "more than 2^32 candidate tuples in one prefix: use a longer prefix");
        const u32 nt = (u32)ncand_total;
        // bottom-up: candidates in subtree
        for (u32 l = nlev; l-- > 1;) {
            EARENA_GET(L[l].sub, u32, L[l].n);
            const u32* child_sub = l + 1 < nlev ? L[l + 1].sub : nullptr;
            // (a level outside the emitted depth range has no candidates of its own: no word array)
            hipLaunchKernelGGL((up_bits_kernel<u32>), grid_npt(L[l].n), dim3(256), 0, st, L[l].n, L[l].cand_bits, L[l].kids(), child_sub, L[l].sub);
        }
        // top-down: start offsets (two rolling arrays); every candidate's post-order rank, and its sizes at that rank
        u32 *plen, *npair;
        EARENA_GET(plen, u32, nt);
        EARENA_GET(npair, u32, nt);
        u32 maxn = 1;
        for (u32 l = 0; l < nlev; ++l) maxn = L[l].n > maxn ? L[l].n : maxn;
        u32* startbuf[2];
        EARENA_GET(startbuf[0], u32, maxn);
        EARENA_GET(startbuf[1], u32, maxn);
        std::vector<u32*> crank(nlev, nullptr);
        for (u32 l = 1; l < nlev; ++l)
            if (L[l].ncand) EARENA_GET(crank[l], u32, L[l].ncand);
        DSM_HIP(hipMemsetAsync(startbuf[0], 0, sizeof(u32), st));
        for (u32 l = 0; l + 1 < nlev; ++l) {
            u32* s_cur = startbuf[l & 1];
            u32* s_next = startbuf[(l + 1) & 1];
            hipLaunchKernelGGL((down_kernel<u32>), grid_npt(L[l].n), dim3(256), 0, st, L[l].n, s_cur, 0u, L[l].kids(), L[l + 1].sub, s_next);
            if (L[l + 1].ncand)
                hipLaunchKernelGGL(cand_rank_kernel, grid_for(L[l + 1].ncand), dim3(256), 0, st, L[l + 1].ncand, L[l + 1].cand_node, L[l + 1].cand_poff,
                                   L[l + 1].npairs, s_next, L[l + 1].sub, l + 1, crank[l + 1], plen, npair, L[l + 1].crec);
        }
        // tuple sizes -> offsets
        std::vector<LevelDev> lv(nlev);
        u32 cb = 0;
        for (u32 l = 0; l < nlev; ++l) {
            lv[l].pw = L[l].pw; lv[l].cand_node = L[l].cand_node; lv[l].cand_poff = L[l].cand_poff; lv[l].crec = L[l].crec; lv[l].crank = crank[l];
            lv[l].ids = L[l].ids; lv[l].freqs = L[l].freqs; lv[l].ncand = l ? L[l].ncand : 0; lv[l].npairs = L[l].npairs;
            lv[l].cbase = cb;
            cb += lv[l].ncand;
        }
        LevelDev* d_lv;
        EARENA_GET(d_lv, LevelDev, nlev);
        DSM_HIP(hipMemcpyAsync(d_lv, lv.data(), nlev * sizeof(LevelDev), hipMemcpyHostToDevice, st));
        // The tuple arrays live in the emit set (not the arena): the copy stream drains them to pinned memory while the
        // compute stream already expands the next prefix.
        EmitSet& E = emitter.acquire();
        E.device = device;
        if (!E.ready) DSM_HIP(hipEventCreateWithFlags(&E.ready, hipEventDisableTiming));
        if (!copy_stream) DSM_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        if (int rc = E.dev[0].ensure(((size_t)nt + 1) * 4)) return rc;
        if (int rc = E.dev[1].ensure(((size_t)nt + 1) * 4)) return rc;
        u32* path_off = (u32*)E.dev[0].p;
        u32* pair_off = (u32*)E.dev[1].p;
        u32* stmp;
        EARENA_GET(stmp, u32, scan_tmp_elems(nt) + 8);
        exclusive_scan<u32, u32>(plen, path_off, nt, stmp, d_totals, st);
        exclusive_scan<u32, u32>(npair, pair_off, nt, stmp, d_totals + 1, st);
        DSM_HIP(hipMemcpyAsync(path_off + nt, d_totals, sizeof(u32), hipMemcpyDeviceToDevice, st));
        DSM_HIP(hipMemcpyAsync(pair_off + nt, d_totals + 1, sizeof(u32), hipMemcpyDeviceToDevice, st));
        // chunk boundaries: consecutive tuple ranges of at least a million tuples
        ChunkBounds cbs;
        cbs.n = nt >= (4u << 20) ? 4 : (nt >= (2u << 20) ? 2 : 1);
        for (int c = 0; c <= cbs.n; ++c) cbs.tb[c] = (u32)((u64)nt * c / cbs.n);
        u32* d_bounds;
        EARENA_GET(d_bounds, u32, 2 * (EmitSet::MAX_CHUNKS + 1));
        hipLaunchKernelGGL(chunk_bounds_kernel, dim3(1), dim3(64), 0, st, cbs, path_off, pair_off, d_bounds);  // after the two sentinel copies
        DSM_HIP(hipMemcpyAsync(h_totals + 16, d_bounds, 2 * (cbs.n + 1) * sizeof(u32), hipMemcpyDeviceToHost, st));
        DSM_HIP(hipMemcpyAsync(h_totals, d_totals, 2 * sizeof(u32), hipMemcpyDeviceToHost, st));
        DSM_HIP(hipStreamSynchronize(st));
        const u64 path_bytes = h_totals[0], npairs = h_totals[1];
        if (int rc = E.dev[2].ensure((size_t)npairs * 4)) return rc;
        if (int rc = E.dev[3].ensure((size_t)npairs * 8)) return rc;
        if (int rc = E.dev[4].ensure((size_t)path_bytes)) return rc;
        u32* d_ids = (u32*)E.dev[2].p;
        u64* d_freqs = (u64*)E.dev[3].p;
        char* d_paths = (char*)E.dev[4].p;
        emitter.text_sink = text_sink_;
        const bool text_mode = text_sink_ != nullptr;  // the emitter formats the chunks on the card: no binary copies, no pinned arrays
        FillVerdict fv;
        memset(&fv, 0, sizeof fv);
        if (!text_mode) {
            const size_t nrel = (size_t)nt + EmitSet::MAX_CHUNKS + 1;
            if (int rc = E.pin[0].ensure(nrel * 4)) return rc;
            if (int rc = E.pin[1].ensure(nrel * 4)) return rc;
            if (int rc = E.pin[2].ensure((size_t)npairs * 4)) return rc;
            if (int rc = E.pin[3].ensure((size_t)npairs * 8)) return rc;
            if (int rc = E.pin[4].ensure((size_t)path_bytes)) return rc;
            if (int rc = E.pin[5].ensure((size_t)nt * 8)) return rc;
            if (int rc = E.pin[6].ensure((size_t)nt)) return rc;
            if (int rc = E.pin[9].ensure(2 * EmitSet::MAX_CHUNKS * sizeof(u32))) return rc;
            if (int rc = E.dev[5].ensure((size_t)nt * 8)) return rc;
            if (int rc = E.dev[6].ensure((size_t)nt)) return rc;
            if (int rc = E.dev[7].ensure(nrel * 4)) return rc;
            if (int rc = E.dev[8].ensure(nrel * 4)) return rc;
            if (int rc = E.dev[9].ensure(2 * EmitSet::MAX_CHUNKS * sizeof(u32))) return rc;
            if (!d_terms) {  // the tables of the exact entropy, once per miner (8.5 MB)
                if (int rc = dalloc(d_terms, (size_t)TERM_TAB)) return rc;
                if (int rc = dalloc(d_logn, (size_t)LOGN_TAB)) return rc;
                DSM_HIP(hipMemcpyAsync(d_terms, term_table(), (size_t)TERM_TAB * 8, hipMemcpyHostToDevice, st));
                DSM_HIP(hipMemcpyAsync(d_logn, logn_table(), (size_t)LOGN_TAB * 8, hipMemcpyHostToDevice, st));
            }
            DSM_HIP(hipMemsetAsync(E.dev[9].p, 0, 2 * EmitSet::MAX_CHUNKS * sizeof(u32), st));
            fv.ent = (double*)E.dev[5].p; fv.keep = (u8*)E.dev[6].p; fv.rel_path = (u32*)E.dev[7].p; fv.rel_pair = (u32*)E.dev[8].p;
            fv.terms = d_terms; fv.logn = d_logn; fv.d = d; fv.emin = prm.emin; fv.emax = prm.emax;
        }
        E.nchunk = cbs.n;
        // One fill per chunk of output ranks (its threads follow the levels, not the output order: every launch looks at all candidates and
        // keeps the ones of its chunk), so that a chunk is on its way to the host while the next one is being filled -- what shows at
        // the end of a pass, where nothing else hides the last prefix's copy (1 GB, 20 ms, behind a 14 ms fill).
        if (!copy_stream) DSM_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        // The fills go out now.  The copies to the host -- blit kernels that spread over the compute units and stay for the length of a PCIe
        // transfer -- and the emitter's work are HELD BACK until the next prefix is a few levels deep (flush_pending): the first levels of
        // a prefix are a dozen dependent launches of a workgroup or two each, and beside the previous prefix's copies every one of them
        // waited for a compute unit (measured: 300-1500 us for a one-workgroup LF-step launch that takes 10 us alone; 9 ms per pass).
        for (int c = 0; c < cbs.n; ++c) {
            const u32 t0 = cbs.tb[c], t1 = cbs.tb[c + 1];
            fv.chunk = (u32)c; fv.pb0 = h_totals[16 + 2 * c]; fv.qb0 = h_totals[17 + 2 * c];
            fv.counts = text_mode ? nullptr : (u32*)E.dev[9].p + 2 * c;
            hipLaunchKernelGGL(tuple_fill_kernel, grid_for(nt), dim3(256), 0, st, nt, nlev, d_lv, path_off, pair_off, d_paths, d_ids, d_freqs, t0, t1, fv);
            DSM_HIP(hipGetLastError());
            E.cb[c] = t0; E.cb[c + 1] = t1;
            if (!E.cready[c]) DSM_HIP(hipEventCreateWithFlags(&E.cready[c], hipEventDisableTiming));
            if (!chunk_filled[c]) DSM_HIP(hipEventCreateWithFlags(&chunk_filled[c], hipEventDisableTiming));
            DSM_HIP(hipEventRecord(chunk_filled[c], st));
            pend.off[2 * c] = h_totals[16 + 2 * c]; pend.off[2 * c + 1] = h_totals[17 + 2 * c];
        }
        pend.off[2 * cbs.n] = h_totals[16 + 2 * cbs.n]; pend.off[2 * cbs.n + 1] = h_totals[17 + 2 * cbs.n];
        pend.E = &E; pend.nchunk = cbs.n; pend.text = text_mode;
        E.nt = nt;
        emitter.d = d; emitter.emin = prm.emin; emitter.emax = prm.emax; emitter.sink = sink; emitter.ctx = ctx;
        *ready = true;
        return 0;
    }
    // what finish_mine held back (see there): the chunks' copies, then the emitter
    struct PendingEmit {
        EmitSet* E = nullptr;
        int nchunk = 0;
        bool text = false, submit = false;
        u64 off[2 * (EmitSet::MAX_CHUNKS + 1)];  // path / pair offsets of the chunk boundaries
    } pend;
    int flush_pending() {
        if (!pend.E || !pend.submit) return 0;
        EmitSet& E = *pend.E;
        pend.E = nullptr;
        pend.submit = false;
        DSM_HIP(hipSetDevice(device));
        const u32* d_ids = (const u32*)E.dev[2].p;
        const u64* d_freqs = (const u64*)E.dev[3].p;
        const char* d_paths = (const char*)E.dev[4].p;
        for (int c = 0; c < pend.nchunk; ++c) {
            const u32 t0 = E.cb[c], t1 = E.cb[c + 1];
            if (pend.text) {  // the emitter thread takes the chunk from the card, on its own stream
                DSM_HIP(hipStreamWaitEvent(copy_stream, chunk_filled[c], 0));
                DSM_HIP(hipEventRecord(E.cready[c], copy_stream));
                continue;
            }
            DSM_HIP(hipStreamWaitEvent(copy_stream, chunk_filled[c], 0));
            const u64 pb0 = pend.off[2 * c], qb0 = pend.off[2 * c + 1], pb1 = pend.off[2 * (c + 1)], qb1 = pend.off[2 * (c + 1) + 1];
            // offsets relative to the chunk (its own closing entry included), entropies, verdicts, the chunk's counts
            DSM_HIP(hipMemcpyAsync((u32*)E.pin[0].p + t0 + c, (const u32*)E.dev[7].p + t0 + c, ((size_t)(t1 - t0) + 1) * 4, hipMemcpyDeviceToHost, copy_stream));
            DSM_HIP(hipMemcpyAsync((u32*)E.pin[1].p + t0 + c, (const u32*)E.dev[8].p + t0 + c, ((size_t)(t1 - t0) + 1) * 4, hipMemcpyDeviceToHost, copy_stream));
            if (t1 > t0) {
                DSM_HIP(hipMemcpyAsync((double*)E.pin[5].p + t0, (const double*)E.dev[5].p + t0, (size_t)(t1 - t0) * 8, hipMemcpyDeviceToHost, copy_stream));
                DSM_HIP(hipMemcpyAsync((u8*)E.pin[6].p + t0, (const u8*)E.dev[6].p + t0, (size_t)(t1 - t0), hipMemcpyDeviceToHost, copy_stream));
            }
            DSM_HIP(hipMemcpyAsync((u32*)E.pin[9].p + 2 * c, (const u32*)E.dev[9].p + 2 * c, 2 * sizeof(u32), hipMemcpyDeviceToHost, copy_stream));
            E.pb[c] = pb0; E.qb[c] = qb0;
            if (qb1 > qb0) {
                DSM_HIP(hipMemcpyAsync((u32*)E.pin[2].p + qb0, d_ids + qb0, (qb1 - qb0) * 4, hipMemcpyDeviceToHost, copy_stream));
                DSM_HIP(hipMemcpyAsync((u64*)E.pin[3].p + qb0, d_freqs + qb0, (qb1 - qb0) * 8, hipMemcpyDeviceToHost, copy_stream));
            }
            if (pb1 > pb0) DSM_HIP(hipMemcpyAsync((char*)E.pin[4].p + pb0, d_paths + pb0, pb1 - pb0, hipMemcpyDeviceToHost, copy_stream));
            DSM_HIP(hipEventRecord(E.cready[c], copy_stream));
        }
        DSM_HIP(hipEventRecord(E.ready, copy_stream));
        emitter.submit();
        return 0;
    }
    hipStream_t copy_stream = nullptr;
    double* d_terms = nullptr;   // device copies of the entropy tables (entropy_tables.h), made when the first tuples are filled
    double* d_logn = nullptr;
    hipEvent_t fill_done = nullptr;
    hipEvent_t chunk_filled[EmitSet::MAX_CHUNKS] = {nullptr};

    // wait for the emitter and fold its counters into stats
    int finish_emits() {
        timeline("finish_emits");
        if (int rc = flush_pending()) return rc;
        timeline("last set submitted");
        emitter.drain();
        timeline("emitter drained");
        std::lock_guard<std::mutex> lk(emitter.mu);
        stats.tuples += emitter.tuples; stats.pairs += emitter.pairs; stats.host_ms += emitter.ms;
        emitter.tuples = emitter.pairs = 0;
        emitter.ms = 0;
        if (emitter.sink_err) {
            const int e = emitter.sink_err;
            emitter.sink_err = 0;
            if (e == 2) return fail(DSM_E_HIP, "text emitter: " + emitter.text_err);
            return fail(DSM_E_SINK, "tuple sink failed");
        }
        return 0;
    }

    // ---- stream: chunks of the depth-first serialisation, one per leaf (see the kernels) ------------------------------
    int finish_stream(std::vector<LevelHost>& L, u32 nlev, dsm_byte_sink sink, void* ctx) {
        if (nlev < 2) {  // nothing below the root: the client sends only its handshake
            sout.submit(0, 0, stream_tag, 0, 0, stream_last);
            return 0;
        }
        // one connection per prefix: reported starts at 0 (EnumerateQuery.h:19-21); a sub-run of a split prefix continues the count
        const u64 rbase = stream_rbase;
        const u32 ntop = nlev - 1 < 6 ? nlev - 1 : 6;  // levels whose nodes send an 'R' token
        u32 maxn = 1;
        for (u32 l = 0; l < nlev; ++l) {
            ARENA_GET(L[l].lf, u32, L[l].n);
            maxn = L[l].n > maxn ? L[l].n : maxn;
        }
        for (u32 l = 1; l <= ntop; ++l) {
            ARENA_GET(L[l].top_rank, u32, L[l].n);
            ARENA_GET(L[l].top_lf, u32, L[l].n);
            ARENA_GET(L[l].rval, u64, L[l].n);
        }
        for (u32 l = nlev; l-- > 0;)  // leaves per subtree
            hipLaunchKernelGGL(stream_leaves_kernel, grid_npt(L[l].n), dim3(256), 0, st, L[l].n, L[l].kids(), l + 1 < nlev ? L[l + 1].lf : (const u32*)nullptr,
                               L[l].lf);
        u32 nleaf = 0;
        DSM_HIP(hipMemcpyAsync(&nleaf, L[0].lf, sizeof(u32), hipMemcpyDeviceToHost, st));
        DSM_HIP(hipStreamSynchronize(st));
        if (nleaf == 0) return fail(DSM_E_HIP, "wire stream: no leaves");
        u64 *leaf_id, *cumk, *chunk_off, *stmp;
        u32 *chunk, *kop, *trip[2][3];
        ARENA_GET(leaf_id, u64, nleaf);
        ARENA_GET(chunk, u32, nleaf);
        ARENA_GET(kop, u32, nleaf);
        ARENA_GET(cumk, u64, (size_t)nleaf + 1);
        ARENA_GET(chunk_off, u64, (size_t)nleaf + 1);
        ARENA_GET(stmp, u64, scan_tmp_elems(nleaf) + 8);
        for (int k = 0; k < 2; ++k)
            for (int q = 0; q < 3; ++q) ARENA_GET(trip[k][q], u32, maxn);
        for (int q = 0; q < 3; ++q) DSM_HIP(hipMemsetAsync(trip[0][q], 0, sizeof(u32), st));  // the root: first leaf 0, nothing inherited
        for (u32 l = 0; l < nlev; ++l) {  // top-down: leaf ranks and run sizes; the leaves' chunks
            StreamDown a;
            memset(&a, 0, sizeof a);
            a.F = L[l].n; a.level = l;
            a.rank = trip[l & 1][0]; a.kin = trip[l & 1][1]; a.cin = trip[l & 1][2];
            a.rank_n = trip[(l + 1) & 1][0]; a.kin_n = trip[(l + 1) & 1][1]; a.cin_n = trip[(l + 1) & 1][2];
            a.clen = l >= 1 ? L[l].clen : nullptr;
            a.child_lf = l + 1 < nlev ? L[l + 1].lf : nullptr;
            a.kids = L[l].kids();
            a.leaf_id = leaf_id; a.chunk = chunk; a.kop = kop;
            a.top_rank = (l >= 1 && l <= ntop) ? L[l].top_rank : nullptr;
            a.top_lf = (l >= 1 && l <= ntop) ? L[l].top_lf : nullptr;
            a.lf = L[l].lf;
            hipLaunchKernelGGL(stream_down_kernel, dim3((L[l].n + TILE - 1) / TILE), dim3(256), 0, st, a);
        }
        // nodes opened before each chunk -> the values of the 'R' tokens, whose bytes join the chunk of the node's last leaf
        exclusive_scan<u32, u64>(kop, cumk, nleaf, stmp, (u64*)nullptr, st);
        for (u32 l = 1; l <= ntop; ++l)
            hipLaunchKernelGGL(stream_rtok_kernel, grid_for(L[l].n), dim3(256), 0, st, L[l].n, rbase, L[l].top_rank, L[l].top_lf, cumk, kop, nleaf, L[l].rval,
                               chunk);
        exclusive_scan<u32, u64>(chunk, chunk_off, nleaf, stmp, d_totals64 + 2, st);
        u64 total = 0;
        DSM_HIP(hipMemcpyAsync(&total, d_totals64 + 2, 8, hipMemcpyDeviceToHost, st));
        // The stream leaves the card in slices of consecutive leaves: slice j crosses the bus while the later ones are still being
        // written, so the part of a pass that nothing overlaps is the last slice of its last prefix, not a whole prefix.
        const int nslice = nleaf >= (1u << 22) ? 4 : 1;
        u32 srank[StreamOut::MAX_SLICES + 1];
        u64 sbyte[StreamOut::MAX_SLICES + 1];
        for (int j = 0; j <= nslice; ++j) srank[j] = (u32)((u64)nleaf * j / nslice);
        sbyte[0] = 0;
        for (int j = 1; j < nslice; ++j) DSM_HIP(hipMemcpyAsync(&sbyte[j], chunk_off + srank[j], 8, hipMemcpyDeviceToHost, st));
        // a sub-run of a split prefix leaves out the closing tokens of the first stream_tail_levels nodes of its enforced path:
        // node 0 of the levels 1, 2, ... (one node per level there)
        const size_t ntail = stream_tail_levels < nlev ? stream_tail_levels : nlev - 1;
        std::vector<uint4> tail_rec(ntail);
        std::vector<u64> tail_rval(ntail, 0);
        for (size_t l = 0; l < ntail; ++l) {
            DSM_HIP(hipMemcpyAsync(&tail_rec[l], L[l + 1].srec, sizeof(uint4), hipMemcpyDeviceToHost, st));
            if (l + 1 <= ntop) DSM_HIP(hipMemcpyAsync(&tail_rval[l], L[l + 1].rval, 8, hipMemcpyDeviceToHost, st));
        }
        DSM_HIP(hipStreamSynchronize(st));
        auto vlen = [](u64 u) { u32 n = 1; if (u >= 128) { u32 bits = 0; for (u64 t = u; t; t >>= 1) ++bits; n = 1 + (bits + 7) / 8; } return (u64)n; };
        u64 tail_bytes = 0;
        for (size_t l = 0; l < ntail; ++l) {
            const u64 fw = ((u64)tail_rec[l].w << 32) | tail_rec[l].z;
            tail_bytes += vlen(fw & ((1ull << 61) - 1)) + 2 + (l + 1 <= ntop ? 1 + vlen(tail_rval[l]) : 0);
        }
        if (stream_head_skip + tail_bytes > total) return fail(DSM_E_HIP, "split stream: slice larger than the stream");
        int k = 0;
        if (int rc = sout.acquire(total, device, &k)) return rc;  // (waits for the prefix before the previous one to have left the card)
        u8* d_out = sout.buf[k];
        // the per-level tables of the writer
        std::vector<const uint4*> h_rec(nlev, nullptr);
        std::vector<const u64*> h_rval(nlev, nullptr);
        for (u32 l = 1; l < nlev; ++l) { h_rec[l] = L[l].srec; h_rval[l] = L[l].rval; }
        const uint4** d_rec;
        const u64** d_rv;
        ARENA_GET(d_rec, const uint4*, nlev);
        ARENA_GET(d_rv, const u64*, nlev);
        DSM_HIP(hipMemcpyAsync(d_rec, h_rec.data(), nlev * sizeof(void*), hipMemcpyHostToDevice, st));
        DSM_HIP(hipMemcpyAsync(d_rv, h_rval.data(), nlev * sizeof(void*), hipMemcpyHostToDevice, st));
        sbyte[nslice] = total;
        const u64 lo = stream_head_skip, hi = total - tail_bytes;  // what of the buffer is sent
        DSM_HIP(hipStreamSynchronize(st));  // (the tables on the host stack above must outlive their copies)
        for (int j = 0; j < nslice; ++j) {
            if (srank[j + 1] > srank[j])
                hipLaunchKernelGGL(stream_chunk_kernel, grid_for(srank[j + 1] - srank[j]), dim3(256), 0, st, srank[j], srank[j + 1], nlev, d_rec, d_rv, leaf_id,
                                   chunk_off, kop, d_out);
            hipEvent_t ev = sout.slice_event(k, j);
            if (!ev) return fail(DSM_E_HIP, "hipEventCreate failed");
            DSM_HIP(hipEventRecord(ev, st));
            const u64 a = sbyte[j] > lo ? sbyte[j] : lo, b = sbyte[j + 1] < hi ? sbyte[j + 1] : hi;
            const bool final_slice = j + 1 == nslice;
            sout.submit(k, total, stream_tag, a, b > a ? b - a : 0, stream_last && final_slice, ev, final_slice);
        }
        DSM_HIP(hipGetLastError());
        return 0;
    }
    StreamOut sout;
    int stream_tag = 0;          // index of the prefix being enumerated (dsm_miner_enumerate_many)
    // a prefix too large for the buffers goes out as the concatenation of slices of its sub-prefixes' streams (MinerT::stream_auto)
    u64 stream_rbase = 0;        // nodes reported before this run's subtree in the unsplit stream
    u64 stream_head_skip = 0;    // leading bytes to leave out (the opening tokens of the enforced path, sent by an earlier sub-run)
    u32 stream_tail_levels = 0;  // enforced nodes whose closing tokens a later sub-run sends
    bool stream_last = true;     // the prefix ends with this run
};

bool need_wide(dsm_index* const* idx, int n, const dsm_params* p) {
    if (p && p->wide) return true;
    for (int k = 0; k < n; ++k)
        if (idx[k]->meta.n >= 0xFFFFFFF0ull) return true;
    return false;
}

template <typename P>
struct MinerT : MinerBase {
    Engine<P> e;
    bool stream_mode() const override { return e.stream_mode; }
    int run(const char* prefix, dsm_tuple_sink ts, dsm_byte_sink bs, void* ctx, dsm_stats* out) override {
        const char* one[1] = {prefix};
        return run_many(one, 1, ts, bs, ctx, out);
    }
    // A prefix whose trie does not fit the device buffers is split the way the reference scales: by longer prefixes.
    // The four sub-prefixes emit everything deeper than the prefix itself; a shallow pass (children of the prefix only)
    // then emits the prefix node and its ancestors, whose predicates need all four children (metaserver.cpp:416-417).
    // Post-order is preserved: descendants first, in A,C,G,T order.  Collective-safe: every rank sees the same failure
    // only if capacities agree, so ranks must use equal arena sizes.  (`reported` counts every node once: a run counts the depths it
    // emits, the shallow pass none; union_nodes and the cost counters still count the enforced path
    // once per sub-run in that case.)
    typedef typename Engine<P>::NodeOrder NodeOrder;
    int run_auto(const std::string& prefix, dsm_tuple_sink ts, void* ctx, bool emit, u32 lo, const NodeOrder* seed) {
        int rc = e.run(prefix.c_str(), ts, nullptr, ctx, emit, lo, ~0u, ~0u, seed);
        if (rc != DSM_E_CAPACITY || prefix.size() >= 32) return rc;
        ++e.splits;
        if (getenv("DSM_TRACE_SPLITS")) fprintf(stderr, "dsm split rank=%d prefix=%s Fcap=%u: %s\n", e.rank, prefix.c_str(), e.Fcap, dsm_last_error());
        const u32 k = (u32)prefix.size();
        NodeOrder cap;  // shallow pass: the children of the prefix node with all four siblings visible -> their orders
        cap.depth = k + 1;
        rc = e.run(prefix.c_str(), ts, nullptr, ctx, false, lo, ~0u, k + 1, seed, &cap, false);
        if (rc) return rc;
        for (size_t q = 0; q < cap.sym.size(); ++q) {
            NodeOrder sub;
            sub.depth = k + 1;
            sub.ord.push_back(cap.ord[q]);
            rc = run_auto(prefix + "ACGT"[cap.sym[q]], ts, ctx, emit, k + 1, &sub);
            if (rc) return rc;
        }
        return e.run(prefix.c_str(), ts, nullptr, ctx, emit, lo, k, k + 1, seed);
    }
    // The wire stream of a prefix whose trie does not fit the device buffers, as the concatenation of slices of its sub-prefixes'
    // streams.  With p = p1..pk and c1 < .. < cm the bases that continue p,
    //   stream(p) = open(p1..pk)  [ '(' ci  subtree(p ci)  close(p ci) ]i=1..m  close(pk..p1)
    // and the run with the enforced path p ci sends open(p1..pk) '(' ci subtree close(p ci) close(pk..p1): the first sub-run is
    // cut before its closing tokens of pk..p1, the middle ones on both sides, the last one after its 2k opening bytes.  The 'R'
    // values (nodes reported so far, EnumerateQuery.cpp:214-218) come out right when a sub-run starts counting at the number of
    // nodes in the subtrees before it -- which also makes the last sub-run's closing tokens of pk..p1 the unsplit stream's.
    // head_skip / tail_levels: what an enclosing split wants cut from this prefix's stream; *below: nodes of the subtree of p's node.
    int stream_auto(const std::string& prefix, void* ctx, u64 rbase, u64 head_skip, u32 tail_levels, bool last, u64* below) {
        const u32 k = (u32)prefix.size();
        e.stream_rbase = rbase; e.stream_head_skip = head_skip; e.stream_tail_levels = tail_levels; e.stream_last = last;
        const u64 before = e.stats.reported;
        int rc = e.run(prefix.c_str(), nullptr, nullptr, ctx, true);
        if (rc != DSM_E_CAPACITY || k >= 32) {
            const u64 rep = e.stats.reported - before;            // the enforced path's k nodes and everything below p's node
            if (below) *below = rep >= k ? rep - k + (k ? 1 : 0) : 0;  // (the root of the empty prefix is not a node of the stream)
            return rc;
        }
        ++e.splits;
        if (getenv("DSM_TRACE_SPLITS")) fprintf(stderr, "dsm split rank=%d prefix=%s Fcap=%u: %s\n", e.rank, prefix.c_str(), e.Fcap, dsm_last_error());
        NodeOrder cap;  // shallow pass: which bases continue p
        cap.depth = k + 1;
        rc = e.run(prefix.c_str(), nullptr, nullptr, ctx, false, 1, ~0u, k + 1, nullptr, &cap);
        if (rc) return rc;
        u64 sum = 0;
        const size_t m = cap.sym.size();
        for (size_t q = 0; q < m; ++q) {
            u64 sub = 0;
            rc = stream_auto(prefix + "ACGT"[cap.sym[q]], ctx, rbase + sum, q == 0 ? head_skip : 2ull * k, q + 1 == m ? tail_levels : k,
                             last && q + 1 == m, &sub);
            if (rc) return rc;
            sum += sub;
        }
        if (m == 0) return fail(DSM_E_CAPACITY, "device arena exhausted: use a larger arena_bytes");  // (cannot happen: a level overflowed)
        if (below) *below = sum + (k ? 1 : 0);
        e.stats.reported = before + k + sum;  // what the unsplit run would have counted: the enforced path once, every node below it once
        return 0;
    }
    // prefixes one after the other on the GPU; the host emits prefix k while prefix k+1 is being expanded
    int run_many(const char* const* prefixes, int n, dsm_tuple_sink ts, dsm_byte_sink bs, void* ctx, dsm_stats* out,
                 dsm_prefix_byte_sink ps = nullptr, dsm_text_sink xs = nullptr) override {
        memset(&e.stats, 0, sizeof e.stats);
        e.text_sink_ = xs;
        int rc = 0;
        timeline("call begins");
        if (e.stream_mode) { e.sout.sink = bs; e.sout.psink = ps; e.sout.ctx = ctx; }
        for (int k = 0; k < n && !rc; ++k) {
            const bool mine = e.owner_mode ? e.is_owner : (!e.prm.emit_owner_only || e.world <= 1 || (k % e.world) == e.rank);
            e.stream_tag = k;
            if (e.stream_mode) rc = stream_auto(prefixes[k] ? prefixes[k] : "", ctx, 0, 0, 0, true, nullptr);
            else rc = run_auto(prefixes[k] ? prefixes[k] : "", ts, ctx, mine, 1, nullptr);
        }
        int rc2 = e.finish_emits();
        if (e.stream_mode) {  // the last prefixes may still be on their way to the sink
            const int se = e.sout.drain();
            if (se && !rc2) rc2 = fail(se == 1 ? DSM_E_SINK : DSM_E_HIP, se == 1 ? "byte sink failed" : "copying the wire stream to the host failed");
        }
        e.stats.splits = e.splits;
        e.splits = 0;
        if (out) *out = e.stats;
        return rc ? rc : rc2;
    }
};

template <typename P>
static int mine_impl(dsm_index* const* idx, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    std::unique_ptr<MinerT<P>> m(new MinerT<P>());
    int rc = m->e.init(idx, n, *p, false);
    if (rc) return rc;
    return m->run(p->prefix, sink, nullptr, ctx, stats);
}
template <typename P>
static int enum_impl(const dsm_index* idx, const char* prefix, u32 fmin, u32 maxdepth, dsm_byte_sink sink, void* ctx, dsm_stats* stats) {
    dsm_params p;
    dsm_params_default(&p);
    p.prefix = prefix;
    p.fmin = fmin;
    p.maxdepth = maxdepth;
    std::unique_ptr<MinerT<P>> m(new MinerT<P>());
    dsm_index* one = const_cast<dsm_index*>(idx);
    int rc = m->e.init(&one, 1, p, true);
    if (rc) return rc;
    return m->run(prefix, nullptr, sink, ctx, stats);  // (a prefix that does not fit the buffers is sent as slices of its sub-prefixes' streams)
}

template <typename P>
static int merge_impl(dsm_trie* const* tr, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    std::unique_ptr<MinerT<P>> m(new MinerT<P>());
    int rc = m->e.init_tries(tr, n, *p);
    if (rc) return rc;
    return m->run("", sink, nullptr, ctx, stats);
}

// one engine run over the given tries (sample id = position): capture (shallow pass, nothing emitted), a unit (run_auto: a unit that
// does not fit the buffers splits like any prefix) or the closing pass over the depths lo..hi
// An engine a server keeps: created for tries of up to `cap` nodes each (place holders size its buffers) and pointed at the tries of
// every run that fits; a run that does not fit gets a new, larger one.  (Creating an engine costs ~50 ms of allocations; a server
// runs dozens of small passes over the tops of its streams and one pass per unit.)
template <typename P>
struct KeptEngine {
    std::unique_ptr<MinerT<P>> m;
    u64 cap = 0;
};
template <typename P>
static int server_run_t(dsm_trie* const* tr, int n, const dsm_params& q, const std::string& prefix, dsm_tuple_sink sink, void* ctx, bool emit,
                        u32 lo, u32 hi, u32 expand_cap, const ServerOrder* seed, ServerOrder* capture, dsm_stats* out, KeptEngine<P>* keep, u64 min_cap) {
    std::unique_ptr<MinerT<P>> own;
    MinerT<P>* m = nullptr;
    u64 need = 0;
    for (int k = 0; k < n; ++k) need = tr[k]->nodes > need ? tr[k]->nodes : need;
    if (keep && keep->m && need <= keep->cap) {
        m = keep->m.get();
        memset(&m->e.stats, 0, sizeof m->e.stats);
    } else {
        own.reset(new MinerT<P>());
        m = own.get();
        int rc;
        if (keep) {
            const u64 cap = need + need / 8 > min_cap ? need + need / 8 : min_cap;  // (a little room: the next units are about this size)
            std::vector<dsm_trie> ph(n);
            std::vector<dsm_trie*> pp(n);
            for (int k = 0; k < n; ++k) { ph[k].device = tr[k]->device; ph[k].nodes = cap; pp[k] = &ph[k]; }
            keep->m.reset();  // (the old one's buffers go first)
            rc = m->e.init_tries(pp.data(), n, q);
            if (rc) return rc;
            keep->m = std::move(own);
            keep->cap = cap;
        } else {
            rc = m->e.init_tries(tr, n, q);
            if (rc) return rc;
        }
    }
    if (keep)
        for (int k = 0; k < n; ++k) m->e.tries[k] = tr[k];
    typename Engine<P>::NodeOrder sd, cp;
    if (seed) { sd.depth = seed->depth; sd.sym = seed->sym; sd.ord = seed->ord; }
    if (capture) cp.depth = capture->depth;
    int rc;
    if (capture || hi != ~0u) rc = m->e.run(prefix.c_str(), sink, nullptr, ctx, emit, lo, hi, expand_cap, seed ? &sd : nullptr, capture ? &cp : nullptr, capture == nullptr);
    else rc = m->run_auto(prefix, sink, ctx, emit, lo, seed ? &sd : nullptr);
    const int rc2 = m->e.finish_emits();
    if (capture) { capture->sym = cp.sym; capture->ord = cp.ord; }
    if (out) *out = m->e.stats;
    return rc ? rc : rc2;
}
struct ServerEngines {  // for the passes over the tops of the streams (top) and for the units (unit)
    KeptEngine<u32> top32, unit32;
    KeptEngine<u64> top64, unit64;
};
// which: 0 = an engine for this run only, 1 = the kept engine for tops, 2 = the kept engine for units
int server_run(bool wide, dsm_trie* const* tr, int n, const dsm_params& q, const std::string& prefix, dsm_tuple_sink sink, void* ctx,
               bool emit, u32 lo, u32 hi, u32 expand_cap, const ServerOrder* seed, ServerOrder* capture, dsm_stats* out, ServerEngines* keep, int which) {
    const u64 min_cap = which == 1 ? 65536 : (1u << 20);
    if (wide) return server_run_t<u64>(tr, n, q, prefix, sink, ctx, emit, lo, hi, expand_cap, seed, capture, out, !keep || !which ? nullptr : (which == 1 ? &keep->top64 : &keep->unit64), min_cap);
    return server_run_t<u32>(tr, n, q, prefix, sink, ctx, emit, lo, hi, expand_cap, seed, capture, out, !keep || !which ? nullptr : (which == 1 ? &keep->top32 : &keep->unit32), min_cap);
}

// ---- what the C entry points (abi.hip) and the server side (server.hip) use of the engine: engine_api.h -------------------------------
MinerBase* miner_create(dsm_index* const* idx, int n, const dsm_params& p, bool stream_mode, int* rc) {
    MinerBase* m;
    if (need_wide(idx, n, &p)) { auto* t = new MinerT<u64>(); *rc = t->e.init(idx, n, p, stream_mode); m = t; }
    else { auto* t = new MinerT<u32>(); *rc = t->e.init(idx, n, p, stream_mode); m = t; }
    if (*rc) { delete m; return nullptr; }
    return m;
}
int mine_once(dsm_index* const* idx, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (need_wide(idx, n, p)) return mine_impl<u64>(idx, n, p, sink, ctx, stats);
    return mine_impl<u32>(idx, n, p, sink, ctx, stats);
}
int enumerate_once(const dsm_index* idx, const char* prefix, u32 fmin, u32 maxdepth, dsm_byte_sink sink, void* ctx, dsm_stats* stats) {
    dsm_index* one = const_cast<dsm_index*>(idx);
    if (need_wide(&one, 1, nullptr)) return enum_impl<u64>(idx, prefix, fmin, maxdepth, sink, ctx, stats);
    return enum_impl<u32>(idx, prefix, fmin, maxdepth, sink, ctx, stats);
}
int merge_once(bool wide, dsm_trie* const* tr, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (wide) return merge_impl<u64>(tr, n, p, sink, ctx, stats);
    return merge_impl<u32>(tr, n, p, sink, ctx, stats);
}
ServerEngines* server_engines_create() { return new ServerEngines(); }
void server_engines_destroy(ServerEngines* e) { delete e; }
}  // namespace dsm
