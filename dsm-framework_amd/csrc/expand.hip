// expand.hip -- the LF-step kernel (gfx950): one sweep of a sample over a frontier level.
//
// The reference walks each sample's suffix trie depth first, one LF call at a time (EnumerateQuery::nextSymbol,
// EnumerateQuery.cpp:151-238; Query::pushChar, Query.h:37-45; EnumerateQuery::pushChar / leftChar, EnumerateQuery.cpp:39-103).
// Here one lane takes one frontier node: the four child intervals, the left-extension intervals of every child, the fmin test and
// the left-char code, fused.  Child records go to the wave's own handles of the next record buffer, the frequency column and the
// left-char codes to the exchange buffer, the child planes to the advance sweep.  See lfstep.h for the record layout and the order
// of a level.
//
// Work distribution (round 4).  A workgroup is a whole CU: as many waves as fit at the kernel's register footprint (16 with 32-bit
// positions, 12 with 64-bit ones), and its waves take their tiles of 64 nodes from a counter in LDS instead of a fixed stride.  A SIMD
// issues from its oldest wave first, so with a fixed share per wave the first wave of every SIMD was done at 0.6 of a launch and the
// last one ran its tail alone, without anybody to hide its memory waits behind; with the counter the waves of a CU end within a tile
// of each other.  A CU's tiles are rows of consecutive tiles, G rows apart (G = workgroups of the launch): the launch still sweeps
// the level -- and with it the sample's records and the index -- front to back.
#include "lfstep.h"

namespace dsm {

#ifdef DSM_LF_STATIC
constexpr bool LF_DYNAMIC = false;   // round 3's distribution: 256-thread workgroups, a fixed stride of tiles per wave (kept for A/B runs)
#else
constexpr bool LF_DYNAMIC = true;
#endif
// (INC: the level's records are compact.  The wide levels -- the few at the top of a prefix, a handful of tiles each -- run with two
// waves per SIMD: their interval-major child loop keeps four children's state in registers.)
// (DENSE: the sweep of one sample among several, its live nodes packed into full tiles -- see expand_sweep; three waves per SIMD: its
// queue and plane words take the LDS a fourth wave's staging area would)
template <typename P, bool INC = true, bool DENSE = false>
struct LfShape {
    static constexpr int WAVES_PER_SIMD = !INC ? 2 : (DENSE ? 3 : (sizeof(P) == 4 ? 4 : 3));   // 256 / 128 / 168 vector registers
#ifdef DSM_LF_WPB
    static constexpr int WPB = DSM_LF_WPB;
#else
    static constexpr int WPB = LF_DYNAMIC ? 4 * WAVES_PER_SIMD : 4;  // waves per workgroup
#endif
};

// ---------------------------------------------------------------------------------------------
// rank on the bit-plane blocks
// ---------------------------------------------------------------------------------------------
// occurrences of A,C,G,T among the first `off` symbols of a block with four population counts:
// |base|, |base & p1| = G+T, |base & p0| = C+T, |base & p1 & p0| = T
__device__ __forceinline__ void blk_counts4(const Blk16& r, u32 off, u32 out[4]) {
    u64 ma = off >= 64 ? ~0ull : ((1ull << off) - 1);
    u64 mb = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
    u64 ba = ma & ~r.p2a, bb = mb & ~r.p2b;
    u64 h1a = ba & r.p1a, h1b = bb & r.p1b;
    u32 tot = __popcll(ba) + __popcll(bb);
    u32 s1 = __popcll(h1a) + __popcll(h1b);
    u32 s0 = __popcll(ba & r.p0a) + __popcll(bb & r.p0b);
    u32 s3 = __popcll(h1a & r.p0a) + __popcll(h1b & r.p0b);
    out[3] = s3; out[2] = s1 - s3; out[1] = s0 - s3; out[0] = tot - s1 - s0 + s3;
}

// LF(c, x-1) for c = A,C,G,T at once: out[c] = C[c] + occurrences of c in BWT[0, x).
template <typename P, bool ONESB>
__device__ __forceinline__ void rank4_blk(const SbArgs& sa, const u64* sbl, const Blk16& r, P x, P out[4]) {
    u32 c4[4];
    blk_counts4(r, (u32)(x & (BLK_SYMS - 1)), c4);
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = (ONESB ? (P)sa.sb0[c] : (P)sbl[(size_t)((u64)x >> SB_SHIFT) * 4 + c]) + (P)(r.cnt[c] + c4[c]);
}

// LF(c, x-1) for one base chosen per lane (c = 0..3), on the plane words of a block; cntc = the block's count of that base.
// A plane is taken as it is or inverted, so the base needs no branch.
template <typename P, bool ONESB>
__device__ __forceinline__ P rank_one(const SbArgs& sa, const u64* sbl, u64 p0a, u64 p0b, u64 p1a, u64 p1b, u64 p2a, u64 p2b, u32 cntc, P x, u32 c) {
    const u32 off = (u32)(x & (BLK_SYMS - 1));
    const u64 ma = off >= 64 ? ~0ull : ((1ull << off) - 1);
    const u64 mb = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
    const u64 i0 = (c & 1u) ? 0ull : ~0ull, i1 = (c & 2u) ? 0ull : ~0ull;
    const u64 xa = ma & ~p2a & (p1a ^ i1) & (p0a ^ i0);
    const u64 xb = mb & ~p2b & (p1b ^ i1) & (p0b ^ i0);
    const P base = ONESB ? (P)DSM_PICK(sa.sb0, c) : (P)sbl[(size_t)((u64)x >> SB_SHIFT) * 4 + c];
    return base + (P)(cntc + (u32)(__popcll(xa) + __popcll(xb)));
}

// The index blocks of a tile are staged in LDS.  The intervals of a tile's nodes are disjoint and increase along the lanes, so
// the blocks the lanes need (block of sp, block of ep + 1) form a non-decreasing sequence: the distinct ones -- about 0.7 per node
// in the wide levels, where neighbouring nodes share blocks -- are numbered by two ballots, their numbers listed in LDS, and
// fetched by groups of four lanes, one 16-byte quarter each: a block is ONE 64-byte request of one load instruction instead of
// four quarter requests in four instructions, repeated by every lane that shares it.  (The kernel is bound by the requests its
// waves issue, not by bytes.)  Word w of the staging area = quarter w & 3 of distinct block w >> 2; lanes then read the blocks
// they need -- for the four-base ranks and again for the left-extension ranks -- from there.
constexpr u32 STAGE_BLOCKS = 128;                        // distinct blocks of a tile: at most two per lane
constexpr u32 WAVE_LDS_WORDS = STAGE_BLOCKS * 4 + 32;    // uint4 per wave: the staged blocks, then the list of their numbers (8.5 KB)
__device__ __forceinline__ void staged_blk(const uint4* wl, u32 idx, Blk16& r) {
    const uint4 h = wl[idx * 4 + 0], a = wl[idx * 4 + 1], c = wl[idx * 4 + 2], d = wl[idx * 4 + 3];
    r.cnt[0] = h.x; r.cnt[1] = h.y; r.cnt[2] = h.z; r.cnt[3] = h.w;
    r.p0a = ((u64)a.y << 32) | a.x; r.p0b = ((u64)a.w << 32) | a.z;
    r.p1a = ((u64)c.y << 32) | c.x; r.p1b = ((u64)c.w << 32) | c.z;
    r.p2a = ((u64)d.y << 32) | d.x; r.p2b = ((u64)d.w << 32) | d.z;
}
template <typename P, bool ONESB>
__device__ __forceinline__ P rank_staged(const SbArgs& sa, const u64* sbl, const uint4* wl, u32 idx, P x, u32 c) {
    const uint4 q1 = wl[idx * 4 + 1], q2 = wl[idx * 4 + 2], q3 = wl[idx * 4 + 3];
    const u32 cntc = reinterpret_cast<const u32*>(wl + idx * 4)[c];
    return rank_one<P, ONESB>(sa, sbl, ((u64)q1.y << 32) | q1.x, ((u64)q1.w << 32) | q1.z, ((u64)q2.y << 32) | q2.x, ((u64)q2.w << 32) | q2.z,
                              ((u64)q3.y << 32) | q3.x, ((u64)q3.w << 32) | q3.z, cntc, x, c);
}
// the same from memory (positions outside the two blocks a lane holds: intervals over more than two blocks only)
template <typename P, bool ONESB>
__device__ __forceinline__ P rank_load(const DevIndex& ix, const SbArgs& sa, const u64* sbl, P x, u32 c) {
    Blk16 b;
    load_blk(ix.blk, (u64)(x >> BLK_SHIFT), b);
    return rank_one<P, ONESB>(sa, sbl, b.p0a, b.p0b, b.p1a, b.p1b, b.p2a, b.p2b, DSM_PICK(b.cnt, c), x, c);
}

__device__ __forceinline__ u64 lf_wave_sum_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ u32 lf_bits_below_lane(u64 p) {  // set bits of p below this lane's position
    return __builtin_amdgcn_mbcnt_hi((u32)(p >> 32), __builtin_amdgcn_mbcnt_lo((u32)p, 0u));
}

// BitRank::rank calls per LF on A,C,G,T in the reference, four bits each (the argument block's cost[] in one scalar register), and
// their sum over a set of bases
__device__ __forceinline__ u32 cost_of(u32 cost_pack, u32 c) { return (cost_pack >> (4 * c)) & 15u; }
__device__ __forceinline__ u32 costsum(u32 cost_pack, u32 set) {
    return ((set & 1u) ? cost_pack & 15u : 0u) + ((set & 2u) ? (cost_pack >> 4) & 15u : 0u) + ((set & 4u) ? (cost_pack >> 8) & 15u : 0u) +
           ((set & 8u) ? (cost_pack >> 12) & 15u : 0u);
}
// the set bits of a 4-bit mask in increasing order, two bits each (masks 0-7 in LO, 8-15 in HI, eight bits per mask)
__device__ __forceinline__ u32 bit_list(u32 m) {
    return (u32)(((m & 8u) ? 0xe439380e340d0c03ull : 0x2409080204010000ull) >> (8 * (m & 7u))) & 0xFFu;
}

// The LF-step kernel.  A wave takes tiles of 64 consecutive nodes of the (colex-ordered) union level.  While a tile is being
// ranked, the record heads of the wave's next tile and the handles of the one after are already on their way (two-stage software
// pipeline), so a tile waits for one memory round trip -- its index blocks -- instead of three dependent ones.  Waves never
// synchronise with each other: a wave's children with symbol c go to the 64 handles  c * seg + 64 * tile ..  of the next record
// buffer, ranked by ballot.
// splane receives, per tile, the four bit planes "this sample keeps child c of node j" (the advance kernel derives the
// children's record handles from them); cnt (single sample only: the union trie is the sample's trie) accumulates the child
// counts per symbol and 256-node tile for the scan.
constexpr u32 CREC_WORDS(size_t psize) { return psize == 4 ? 1u : 2u; }  // uint4 per compact record

template <typename P, bool INC>
struct RecHead;
template <typename P>
struct RecHead<P, false> {   // what a lane needs of its record before anything can be ranked (wide format: decoded fields, all four slots)
    P sp, ep, e0min, e0max, e1min, e1max, e2min, e2max, e3min, e3max;
    u32 flags;     // bits 0-3: mask of the non-empty left-extension intervals, bit 8: the node is present in this sample
    u32 r;         // the record's handle (the few nodes with more than two intervals read the others through it)
};
template <typename P>
struct RecHead<P, true> {    // compact format: the raw words, decoded when the tile is worked on
    uint4 w[sizeof(P) == 4 ? 1 : 2];
    u32 r;         // handle, DEAD for an absent node
};
// Every load of the pipeline is unconditional (absent nodes read record 0 and discard it; the first two interval slots are
// read whether or not they are in use): a load behind a branch makes the compiler wait for ALL outstanding loads at the join --
// the prefetches would stop being prefetches -- and one node in ten has two intervals, i.e. nearly every wave has such a lane.
template <typename P>
__device__ __forceinline__ void load_head(const P* __restrict__ rec, size_t cap, u32 r, RecHead<P, false>& h) {
    const bool live = r != DEAD;
    const u32 rr = live ? r : 0u;
    const P sp = rec[rr], ep = rec[cap + rr];
    h.e0min = rec[2 * cap + rr]; h.e0max = rec[3 * cap + rr];
    h.e1min = rec[4 * cap + rr]; h.e1max = rec[5 * cap + rr];
    h.e2min = rec[6 * cap + rr]; h.e2max = rec[7 * cap + rr];   // (the wide levels are the few at the top of a prefix: their nodes have all
    h.e3min = rec[8 * cap + rr]; h.e3max = rec[9 * cap + rr];   // four intervals as a rule, and reading them later costs a round trip each)
    const u32 m = reinterpret_cast<const u8*>(rec + (size_t)REC_FIELDS * cap)[rr];
    h.sp = live ? sp : (P)1; h.ep = live ? ep : (P)0;
    h.flags = live ? (m | 0x100u) : 0u;
    h.r = rr;
}
template <typename P>
__device__ __forceinline__ void load_head(const P* __restrict__ rec, size_t cap, u32 r, RecHead<P, true>& h) {
    const uint4* c = reinterpret_cast<const uint4*>(rec) + (size_t)(r != DEAD ? r : 0u) * CREC_WORDS(sizeof(P));
    h.w[0] = c[0];
    if (sizeof(P) == 8) h.w[sizeof(P) == 4 ? 0 : 1] = c[sizeof(P) == 4 ? 0 : 1];
    h.r = r;
}
// decoded view of a head
template <typename P>
struct NodeIn {
    P sp, ep, e0min, e0max, e1min, e1max;
    P e2min = 0, e2max = 0, e3min = 0, e3max = 0;   // wide format only
    u32 emask, r;
    bool live;
};
template <typename P>
__device__ __forceinline__ void decode_head(const RecHead<P, false>& h, NodeIn<P>& n) {
    n.sp = h.sp; n.ep = h.ep; n.e0min = h.e0min; n.e0max = h.e0max; n.e1min = h.e1min; n.e1max = h.e1max;
    n.e2min = h.e2min; n.e2max = h.e2max; n.e3min = h.e3min; n.e3max = h.e3max;
    n.emask = h.flags & 31u; n.live = (h.flags & 0x100u) != 0; n.r = h.r;
}
__device__ __forceinline__ void decode_head(const RecHead<u32, true>& h, NodeIn<u32>& n) {
    const uint4 v = h.w[0];
    n.live = h.r != DEAD; n.r = n.live ? h.r : 0u;
    const u32 sp = v.x;
    n.sp = n.live ? sp : 1u;
    n.ep = n.live ? sp + (v.y & 0xFFFFu) : 0u;
    n.e0min = sp + (v.y >> 16); n.e0max = sp + (v.z & 0xFFFFu);
    n.e1min = sp + (v.z >> 16); n.e1max = sp + (v.w & 0xFFFFu);
    n.emask = n.live ? (v.w >> 16) & 31u : 0u;
}
__device__ __forceinline__ void decode_head(const RecHead<u64, true>& h, NodeIn<u64>& n) {
    const uint4 v = h.w[0], x = h.w[1];
    n.live = h.r != DEAD; n.r = n.live ? h.r : 0u;
    const u64 sp = ((u64)v.y << 32) | v.x;
    n.sp = n.live ? sp : 1ull;
    n.ep = n.live ? sp + (v.z & 0xFFFFu) : 0ull;
    n.e0min = sp + (v.z >> 16); n.e0max = sp + (v.w & 0xFFFFu);
    n.e1min = sp + (v.w >> 16); n.e1max = sp + (x.x & 0xFFFFu);
    n.e2min = sp + (x.y & 0xFFFFu); n.e2max = sp + (x.y >> 16);   // (the 32-byte record holds all four slots)
    n.e3min = sp + (x.z & 0xFFFFu); n.e3max = sp + (x.z >> 16);
    n.emask = n.live ? (x.x >> 16) & 31u : 0u;
}
// A finished child: interval [nsp, nep], kept intervals 0 and 1 as absolute positions.  Slots 2, 3: written to the wide fields already,
// except for the compact record of 64-bit positions, which takes them as packed 16-bit offsets (o2, o3: min | max << 16).
template <typename P>
constexpr bool slots_inline(bool compact) { return compact && sizeof(P) == 8; }
template <typename P, bool OUTC>
__device__ __forceinline__ void store_child(P* __restrict__ out, size_t cap, u32 q, P nsp, P nep, P l0, P h0, P l1, P h1, u32 cn, u32 cm,
                                            u32 o2 = 0, u32 o3 = 0) {
    if (OUTC) {
        uint4* c = reinterpret_cast<uint4*>(out) + (size_t)q * CREC_WORDS(sizeof(P));
        const u32 len = (u32)(nep - nsp), a0 = cn > 0 ? (u32)(l0 - nsp) : 0u, b0 = cn > 0 ? (u32)(h0 - nsp) : 0u,
                  a1 = cn > 1 ? (u32)(l1 - nsp) : 0u, b1 = cn > 1 ? (u32)(h1 - nsp) : 0u;
        if (sizeof(P) == 4) {
            c[0] = make_uint4((u32)nsp, len | (a0 << 16), b0 | (a1 << 16), b1 | (cm << 16));
        } else {
            c[0] = make_uint4((u32)nsp, (u32)((u64)nsp >> 32), len | (a0 << 16), b0 | (a1 << 16));
            c[sizeof(P) == 4 ? 0 : 1] = make_uint4(b1 | (cm << 16), cn > 2 ? o2 : 0u, cn > 3 ? o3 : 0u, 0u);
        }
    } else {
        out[q] = nsp;
        out[cap + q] = nep;
        if (cn > 0) { out[2 * cap + q] = l0; out[3 * cap + q] = h0; }
        if (cn > 1) { out[4 * cap + q] = l1; out[5 * cap + q] = h1; }
        reinterpret_cast<u8*>(out + (size_t)REC_FIELDS * cap)[q] = (u8)cm;
    }
}

// Counters of a wave's sweep.  They are sums over the lanes, and every term is a population count of a lane mask the tile computes
// anyway (children per symbol, nodes with at least i left-extension intervals, ...): the sums are formed by the scalar unit, per wave,
// and need neither vector registers nor a reduction at the end (round 4; per-lane accumulators before).
struct ExpandAcc {
    u32 kne = 0;        // children emitted (when they count as reported)
    u32 live = 0;       // records read
    u32 lines = 0;      // index blocks fetched
    u32 lf = 0, rank = 0;  // what the reference would have spent: LF calls, BitRank::rank calls
    u32 rbytes = 0;     // bytes of records read and written (wave-uniform part)
    u32 rb_lane = 0;    // ... per lane: wide child records only, whose size depends on the intervals a child keeps
    u32 wide = 0;       // bit 0: some surviving child has a frequency of 512 or more, bit 1: of 65535 or more (the next level's column format)
};
__device__ __forceinline__ u32 mask_count(u64 m) { return (u32)__popcll(m); }

// Which tile a wave takes next.  Dynamic: the workgroup's waves share a counter in LDS; draw k of workgroup b is tile
// ((k / WPB) * G + b) * WPB + k % WPB -- rows of WPB consecutive tiles, G rows apart (G = workgroups of the launch).  Static (round 3):
// wave gw of nwaves takes gw, gw + nwaves, ...  A draw past the level's last tile means "no tile"; the draws of a wave increase, so
// every later one is past it too.
// (Measured and not kept: handing the rows out through a counter in global memory as well, so that the CUs that are done early take
// more -- the CUs' ends then coincide, the launch does not get shorter (77.4-77.9 against 77.3-77.7 ms per pass), and the bookkeeping
// costs registers the kernel does not have: profiles/r04_experiments.)
template <int WPB>
struct TileSeq {
    u32* ctr;     // dynamic: the workgroup's counter (LDS)
    u32 G, b;     // workgroups of the launch, this one
    u32 last;     // static: the tile drawn last
    u32 stride;   // static: waves of the launch
    __device__ __forceinline__ u32 issue() {  // the draw is an LDS atomic: issued here, its value is taken by take()
        if (!LF_DYNAMIC) return 0u;
        u32 k = 0;
        if ((threadIdx.x & 63) == 0) k = atomicAdd(ctr, 1u);
        return k;
    }
    __device__ __forceinline__ u32 take(u32 issued) {
        if (!LF_DYNAMIC) {
            const u32 far = ~last < stride ? ~0u : last + stride;  // (saturates: "no tile" stays "no tile")
            last = far;
            return far;
        }
        const u32 k = (u32)__builtin_amdgcn_readfirstlane((int)issued);
        return ((k / (u32)WPB) * G + b) * (u32)WPB + k % (u32)WPB;
    }
    __device__ __forceinline__ u32 next() { return take(issue()); }
};

// One tile of 64 nodes.  hc: the heads of this tile (requested one tile ago); hn: receives the heads of the wave's next tile,
// whose handles are in rn (requested one tile ago); rn then receives the handles of the tile drawn here (tfar: the wave's tile after
// next).
// SELF (several samples, no handle table): rp is the level's slot array (4 * parent + base per node, shared by all samples) and a
// node's handle follows from the planes the sample wrote at the parent level (pplane): child c of the parent at round T, lane j has
// the handle c * seg + 64 * T + (set bits of plane[T][c] below j), and no handle when the bit is clear.  The pipeline is one stage
// deeper: slots of the tile three ahead, plane words of the tile two ahead, heads of the next tile.
struct SelfState {
    u32 s1 = DEAD, s2 = DEAD;  // slots of the lane's nodes one and two tiles ahead
    u64 pn = 0;                // plane word for s1
};
__device__ __forceinline__ u32 self_handle(u32 slot, u64 plane, u32 seg) {
    const u32 j = (slot >> 2) & 63u;
    const bool has = slot != DEAD && ((plane >> j) & 1ull);
    return has ? (slot & 3u) * seg + ((slot >> 8) << 6) + (u32)__popcll(plane & ((1ull << j) - 1)) : DEAD;
}
__device__ __forceinline__ size_t self_plane_index(u32 slot) { return slot != DEAD ? (size_t)(slot >> 8) * 4 + (slot & 3u) : (size_t)0; }

// DENSE (one sample among several): the 64 lanes hold 64 consecutive nodes THIS SAMPLE HAS, wherever they sit in the union level -- a
// union tile's nodes may be spread over two dense tiles, a dense tile may span many union tiles.  What depended on "lane j = node j of
// union tile t" is carried per lane: the node's place in the union level (its column entry), and, for the children's handles
// c * seg + 64 * T + (parents with a child c before it in union tile T), the lanes of the same union tile (they are neighbours) plus
// what the previous dense tile held of the first one.  The planes of the union tiles are put together in LDS, bit by bit.
#ifndef DSM_DENSE_STEP32
#define DSM_DENSE_STEP32 4   // union tiles per producer step, 32-bit positions (A/B builds: -DDSM_DENSE_STEP32=6)
#endif
constexpr u32 DENSE_Q = 512;       // queue entries per wave (a ring: a tile being prefetched, a tile in waiting, a producer step of up to four union tiles)
constexpr u32 DENSE_ITEM_MAX = 32; // union tiles per work item (ExpandArgs::item_tiles: smaller on smaller levels)
constexpr u32 DENSE_LDS_WORDS = DENSE_Q + DENSE_Q / 2 + DENSE_ITEM_MAX * 4 * 2;  // u32 per wave: queue (handles; places inside the item, 16 bits), the item's plane words
struct DenseTile {
    u32 iu;         // per lane: the node's place in the union level
    u32 rnext;      // per lane: handle of the lane's node in the wave's next dense tile (DEAD: none)
    u64* lpl;       // LDS: plane words of the item's union tiles, [tile - T0][4]
    u32 T0;         // first union tile of the item
    u32 carry[4];   // in / out (wave-uniform): children c among the nodes of union tile lastT that earlier dense tiles held
    u32 lastT;      // in / out: union tile of the last node of the previous dense tile (~0: none)
};

template <typename P, bool ONESB, bool INC, bool OUTC, bool SELF, int WPB, bool DENSE = false>
__device__ __forceinline__ void expand_tile(const DevIndex& ix, const u64* sbl, uint4* wl, const u32* __restrict__ rp, const P* __restrict__ rec,
                                            P* __restrict__ out, u64* __restrict__ splane, u32* __restrict__ cnt, P* __restrict__ valf,
                                            u8* __restrict__ pl, const ExpandArgs& a, const u32 t, TileSeq<WPB>& seq, u32& tfar, const u32 ntile,
                                            const RecHead<P, INC>& hc, RecHead<P, INC>& hn, u32& rn, ExpandAcc& acc,
                                            const u64* __restrict__ pplane, SelfState* ss, const u32* __restrict__ keeptab, const u32 cost_pack,
                                            DenseTile* dt = nullptr) {
    const int lane = threadIdx.x & 63;
    const u64 lt = (1ull << lane) - 1;
    const size_t cap = a.cap, capi = a.cap_in;   // handle spaces of the children's records and of this level's
    const u32 i = DENSE ? dt->iu : t * 64 + lane;
    const u32 drawn = DENSE ? 0u : seq.issue();  // (the wave's tile after next, or three ahead: taken where its handles are requested)
    NodeIn<P> nd;
    decode_head(hc, nd);
    const bool live = nd.live;
    const P sp = nd.sp, ep = nd.ep;
    // (bit 4 of a record's mask: the node's parent had one occurrence, i.e. the reference reached it inside followOneBranch, which reads
    // the node's BWT symbol even when the node lies at maxdepth: EnumerateQuery.cpp:105-149 against :151-153)
    const u32 emask = nd.emask & 15u;
    const bool by_branch = (nd.emask >> 4) & 1u;
    const u32 ne = __popc(emask);
    u32 keepw = ~0u;  // the word of the keep table that holds this node's frequency (requested here, used at the candidate ballot)
    const P freq1 = ep - sp + 1;   // (positions stay of type P: no 64-bit arithmetic on a 32-bit index)
    if (!SELF) keepw = keeptab[(freq1 < (P)KEEP_FREQS ? (u32)freq1 : 0u) >> 5];
    const P ep1 = ep + 1;
    const P b0 = sp >> BLK_SHIFT, b1 = ep1 >> BLK_SHIFT;  // blocks of the two interval ends
    P Rsp[4], Rep[4];
    u32 present = 0;  // bit c: child c is emitted
    u32 idx0, idx1;   // numbers of this lane's two blocks among the tile's distinct blocks
    {
        // ---- the distinct blocks of the tile (see the staging note above) ----
        u32* list = reinterpret_cast<u32*>(wl + STAGE_BLOCKS * 4);
        u32 pm = live ? (u32)b1 + 1u : 0u;  // 1 + last block of the lane; running maximum over the lanes below = the last block listed
        if (!__all(live)) {  // absent nodes in between (several samples, or the last tile): the maximum is carried across them
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(pm, dd, 64); if (lane >= dd) pm = o > pm ? o : pm; }
        }
        u32 prev = __shfl_up(pm, 1, 64);  // (every lane present: the blocks increase along the lanes, the lane below holds the maximum)
        if (lane == 0) prev = 0;
        const bool f0 = live && (u32)b0 + 1u > prev, f1 = live && b1 != b0;
        const u64 m0 = __ballot(f0), m1 = __ballot(f1);
        const u32 D = (u32)__popcll(m0) + (u32)__popcll(m1);
        const u32 before = lf_bits_below_lane(m0) + lf_bits_below_lane(m1);
        idx0 = f0 ? before : (before ? before - 1u : 0u);
        idx1 = f1 ? idx0 + 1u : idx0;
        if (!live) { idx0 = 0; idx1 = 0; }
        acc.lines += D;
        if (f0) list[idx0] = (u32)b0;
        if (f1) list[idx1] = (u32)b1;
        if (D == 0 && lane == 0) list[0] = 0;
        const u32 Dm1 = D ? D - 1u : 0u;
        const u32 g = (u32)lane >> 2, qq = (u32)lane & 3u;
        // four lanes per block, sixteen blocks per instruction; lanes beyond the last block repeat it
        uint4 v0, v1, v2, v3;
#if defined(DSM_LF_SENS_GATHER) || defined(DSM_LF_SENS_STREAM)
        uint4 sens0 = make_uint4(0, 0, 0, 0), sens1 = sens0, sens2 = sens0, sens3 = sens0;
#endif
        {
            const u32 d0 = g < Dm1 ? g : Dm1, d1 = g + 16 < Dm1 ? g + 16 : Dm1, d2 = g + 32 < Dm1 ? g + 32 : Dm1, d3 = g + 48 < Dm1 ? g + 48 : Dm1;
            const u32 bb0 = list[d0], bb1 = list[d1], bb2 = list[d2], bb3 = list[d3];
            const uint4* base = reinterpret_cast<const uint4*>(ix.blk);
            v0 = base[(size_t)bb0 * 4 + qq]; v1 = base[(size_t)bb1 * 4 + qq]; v2 = base[(size_t)bb2 * 4 + qq]; v3 = base[(size_t)bb3 * 4 + qq];
#ifdef DSM_LF_SENS_GATHER  // sensitivity probe: every distinct block is fetched a second time from an unrelated place of the index
            {
                const u32 nb = (u32)ix.nblk, o = nb / 2 + 12345u;
                const u32 c0 = (bb0 + o) % nb, c1 = (bb1 + o) % nb, c2 = (bb2 + o) % nb, c3 = (bb3 + o) % nb;
                sens0 = base[(size_t)c0 * 4 + qq]; sens1 = base[(size_t)(d1 == d0 ? c0 : c1) * 4 + qq]; sens2 = base[(size_t)(d2 == d1 ? c0 : c2) * 4 + qq]; sens3 = base[(size_t)(d3 == d2 ? c0 : c3) * 4 + qq];
            }
#endif
        }
        if (D > 64) {  // (wave-uniform) the rarer second half
            const u32 d0 = g + 64 < Dm1 ? g + 64 : Dm1, d1 = g + 80 < Dm1 ? g + 80 : Dm1, d2 = g + 96 < Dm1 ? g + 96 : Dm1, d3 = g + 112 < Dm1 ? g + 112 : Dm1;
            const u32 bb0 = list[d0], bb1 = list[d1], bb2 = list[d2], bb3 = list[d3];
            const uint4* base = reinterpret_cast<const uint4*>(ix.blk);
            const uint4 w0 = base[(size_t)bb0 * 4 + qq], w1 = base[(size_t)bb1 * 4 + qq], w2 = base[(size_t)bb2 * 4 + qq], w3 = base[(size_t)bb3 * 4 + qq];
            wl[256 + lane] = w0; wl[320 + lane] = w1; wl[384 + lane] = w2; wl[448 + lane] = w3;
        }
        // ---- the pipeline: heads of the next tile, handles of the one after (younger than the block loads, so waiting for
        // the blocks leaves them in flight) ----
        if (DENSE) load_head<P>(rec, capi, dt->rnext, hn);
        else {
            tfar = seq.take(drawn);
            const u32 ifar = tfar * 64 + lane;
            const bool infar = tfar < ntile && ifar < a.F;
            if (SELF) {
                load_head<P>(rec, capi, self_handle(ss->s1, ss->pn, a.seg_in), hn);
                ss->pn = pplane[self_plane_index(ss->s2)];
                ss->s1 = ss->s2;
                const u32 v = rp[infar ? ifar : 0u];
                ss->s2 = infar ? v : DEAD;
            } else {
                load_head<P>(rec, capi, rn, hn);
#ifdef DSM_LF_SENS_STREAM  // sensitivity probe: 16 more bytes per node streamed in (the lines the children's records will be written to)
                sens0 = reinterpret_cast<const uint4*>(out)[(size_t)(rn != DEAD ? rn : 0u)];
#endif
                const u32 v = rp[infar ? ifar : 0u];
                rn = infar ? v : DEAD;
            }
        }
        wl[lane] = v0; wl[64 + lane] = v1; wl[128 + lane] = v2; wl[192 + lane] = v3;
        asm volatile("" ::: "memory");  // the blocks are read back from LDS (other lanes' words among them)
        // No branch on `live` around the ranks: an absent node holds the empty interval [1, 0], every child of which is empty.
        Blk16 r0;
        staged_blk(wl, idx0, r0);
        rank4_blk<P, ONESB>(a.sb, sbl, r0, sp, Rsp);  // LF(c, sp-1)
        // BWT[sp], for the size-1 path.  (A node of frequency 1 exists only when fmin is 1: with fmin >= 2 every node of a level has at
        // least fmin occurrences, the root all of them -- the test below is then never true and the symbol is not looked up.)
        const bool may_single = (a.symbol_phase & 1u) && a.fmin <= 1u;
        u32 lcode = 0;
        if (may_single) lcode = blk_code_at(r0, (u32)sp & (BLK_SYMS - 1));
        staged_blk(wl, idx1, r0);
        rank4_blk<P, ONESB>(a.sb, sbl, r0, ep1, Rep);  // LF(c, ep)
#ifdef DSM_LF_SENS_VALU  // sensitivity probe: the two four-base ranks a second time (about 110 vector instructions per tile more)
        {
            u32 i0b = idx0, i1b = idx1;
            asm volatile("" : "+v"(i0b), "+v"(i1b));
            P R2[4], R3[4];
            Blk16 rr;
            staged_blk(wl, i0b, rr);
            rank4_blk<P, ONESB>(a.sb, sbl, rr, sp, R2);
            staged_blk(wl, i1b, rr);
            rank4_blk<P, ONESB>(a.sb, sbl, rr, ep1, R3);
#pragma unroll
            for (int c = 0; c < 4; ++c) if (__any(R2[c] != Rsp[c] || R3[c] != Rep[c])) acc.wide |= 4u;
        }
#endif
#if defined(DSM_LF_SENS_GATHER) || defined(DSM_LF_SENS_STREAM)
        if (__any((sens0.x ^ sens1.y ^ sens2.z ^ sens3.w) == 0x9e3779b9u && sens0.w == 0x7f4a7c15u)) acc.wide |= 4u;  // (never: the loads must stay)
#endif
        // ---- the children: child c of a node is the interval [Rsp[c], Rep[c] - 1], empty when the two ranks agree (an absent node
        // holds the empty interval [1, 0]); it is emitted when its frequency reaches fmin (EnumerateQuery.cpp:186).  No branches:
        // the comparisons ARE the wave's child masks.
        u64 nemask[4];    // lanes whose child c is non-empty (what the reference pays left-extension ranks for)
        P fmax = 0;       // largest frequency among this lane's emitted children
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const P f = Rep[c] - Rsp[c];
            const bool okc = ((a.allowed >> c) & 1u) != 0 && live;
            const bool ne_c = okc && f != 0;
            const bool pr_c = ne_c && f >= (P)a.fmin;
            nemask[c] = __ballot(ne_c);
            present |= pr_c ? (1u << c) : 0u;
            fmax = pr_c && f > fmax ? f : fmax;
        }
        {
            const u64 w1 = __ballot(fmax >= (P)PACK_FMAX), w2 = __ballot(fmax >= (P)65535);
            acc.wide |= (w1 ? 1u : 0u) | (w2 ? 3u : 0u);
        }
        // What the reference would have spent on this node: two LF per attempted base (Query::pushChar, Query.h:37-45) and two per
        // left-extension interval for every base whose interval is non-empty; BitRank::rank calls = LF calls weighted by the
        // base's code length.  Summed over the wave: (lanes with at least i intervals) x (lanes whose child c is non-empty), as masks.
        const bool single = may_single && sp == ep && live;  // followOneBranch, EnumerateQuery.cpp:105-149
        if (!may_single || !__any(single)) {
            const u64 lv = __ballot(live);
            const u64 m1 = __ballot(ne >= 1), m2 = __ballot(ne >= 2);
            u32 s_lf = 0, s_rank = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32 sc = mask_count(m1 & nemask[c]) + mask_count(m2 & nemask[c]);
                s_lf += sc; s_rank += sc * cost_of(cost_pack, c);
            }
            if (__any(ne > 2)) {  // (well under one node in a hundred)
                const u64 m3 = __ballot(ne >= 3), m4 = __ballot(ne >= 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 sc = mask_count(m3 & nemask[c]) + mask_count(m4 & nemask[c]);
                    s_lf += sc; s_rank += sc * cost_of(cost_pack, c);
                }
            }
            const u32 nl = mask_count(lv);
            acc.lf += 2u * ((u32)__popc(a.allowed) * nl + s_lf);
            acc.rank += 2u * (costsum(cost_pack, a.allowed) * nl + s_rank);
        } else {  // nodes of frequency 1 follow one branch by getL instead (only reachable with fmin = 1): per lane, then summed
            u32 nonempty = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) nonempty |= ((nemask[c] >> lane) & 1ull) ? (1u << c) : 0u;
            u32 n_lf = 2 * (u32)__popc(a.allowed) + 2 * ne * (u32)__popc(nonempty);
            u32 n_rank = 2 * costsum(cost_pack, a.allowed) + 2 * ne * costsum(cost_pack, nonempty);
            if (single) {
                const bool go = a.allowed && ((nonempty >> lcode) & 1u) && lcode < 4;
                n_lf = go ? 2 * ne + 2 : 0u;
                n_rank = ((a.allowed || by_branch) ? (a.access_pack >> (4 * lcode)) & 15u : 0u) + (go ? (2 * ne + 2) * cost_of(cost_pack, lcode & 3u) : 0u);
            }
            if (!live) { n_lf = 0; n_rank = 0; }
            acc.lf += (u32)lf_wave_sum_u64(n_lf);
            acc.rank += (u32)lf_wave_sum_u64(n_rank);
        }
    }
    // ---- places of the child records: per symbol, rank of the parent inside the wave's tile ----
    const u32 k = __popc(present);
    u64 bal[4];
    u32 qa[4];  // handle of this lane's child with symbol c
#pragma unroll
    for (int c = 0; c < 4; ++c) bal[c] = __ballot((present >> c) & 1u);
    if constexpr (DENSE) {
        const u64 lv = __ballot(live);                       // (the live lanes are the first ones: only a wave's last tile of an item is short)
        const u32 T = live ? (i >> 6) : 0xFFFFFFFFu;
        const u32 Tp = __shfl_up(T, 1, 64);
        const u64 bnd = __ballot(lane == 0 || T != Tp);      // first lanes of the runs of one union tile
        const u32 gs = 63u - (u32)__clzll((long long)(bnd & (lt | (1ull << lane))));   // first lane of this lane's run
        const u64 gmask = lt & ~((1ull << gs) - 1ull);       // the lanes of the run before this one
        const u32 Tfirst = (u32)__builtin_amdgcn_readfirstlane((int)T);
        const bool cont = dt->lastT == Tfirst;               // the first run continues a union tile the previous dense tile ended in
        const u32 nl = (u32)__popcll(lv);
        const u32 last = nl ? nl - 1u : 0u;
        const u32 Tlast = (u32)__builtin_amdgcn_readlane((int)T, (int)last);
        const u32 gs_last = (u32)__builtin_amdgcn_readlane((int)gs, (int)last);
        const u64 lastrun = lv & ~((1ull << gs_last) - 1ull);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32 cin = cont ? dt->carry[c] : 0u;
            qa[c] = (u32)c * a.seg + (i & ~63u) + (u32)__popcll(bal[c] & gmask) + (T == Tfirst ? cin : 0u);
            if ((present >> c) & 1u) atomicOr((unsigned long long*)&dt->lpl[(size_t)((i >> 6) - dt->T0) * 4 + c], 1ull << (i & 63u));
            dt->carry[c] = (u32)__popcll(bal[c] & lastrun) + (Tlast == Tfirst ? cin : 0u);
        }
        if (nl) dt->lastT = Tlast;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) qa[c] = (u32)c * a.seg + t * 64 + (u32)__popcll(bal[c] & lt);
        if (lane < 4) {
            const u64 b = DSM_PICK(bal, lane);
            if (!(a.symbol_phase & 8u)) splane[(size_t)t * 4 + lane] = b;
            const u32 nb = (u32)__popcll(b);
            if (cnt && nb) atomicAdd(cnt + (size_t)lane * a.nbp + (t >> 2), nb);
        }
    }
    // EnumerateQuery::leftChar, EnumerateQuery.cpp:77-103: 0='0' 1..4=A,C,G,T 5='N' (the letter is the LAST non-empty base)
    bool matches = (ne > 0 && nd.e0min == sp && nd.e0max == ep) || (ne > 1 && nd.e1min == sp && nd.e1max == ep);
    if (!INC || slots_inline<P>(INC)) {
        matches = matches || (ne > 2 && nd.e2min == sp && nd.e2max == ep) || (ne > 3 && nd.e3min == sp && nd.e3max == ep);
    } else if (__any(ne > 2)) {  // third / fourth interval of a node (well under one node in a hundred): read here and again by its pairs
        if (ne > 2) {
#pragma unroll
            for (int e = 2; e < 4; ++e)
                if ((u32)e < ne && rec[(size_t)(2 + 2 * e) * capi + nd.r] == sp && rec[(size_t)(3 + 2 * e) * capi + nd.r] == ep) matches = true;
        }
    }
    const u32 mycode = !live ? 0u : (matches ? 1u + (31u - (u32)__clz((int)emask)) : (ne ? 5u : 0u));
    if (a.symbol_phase & 8u) {  // one sample: planes and the level's candidates in one line per tile (the advance sweep reads it)
        // metaserver.cpp:406-419 for a node with one reader: not a single child (416-417), no single left char (383-387, 418)
        const bool ekeep = SELF || freq1 >= (P)KEEP_FREQS || ((keepw >> ((u32)freq1 & 31u)) & 1u);  // the exact entropy verdict for this frequency
        const u64 cb = __ballot(live && (a.symbol_phase & 4u) && k != 1u && !(mycode >= 1u && mycode <= 4u) && ekeep);
        if (lane < 6) {
            const u64 nc = (u64)__popcll(cb);
            const u64 word = lane < 4 ? DSM_PICK(bal, lane) : (lane == 4 ? cb : (nc | (nc << 32)));
            splane[(size_t)t * 8 + lane] = word;
        }
    }
    // (a node with one occurrence follows one branch: its child's record says so, see by_branch)
    const u32 branch_bit = ((a.symbol_phase & 1u) && a.fmin <= 1u && sp == ep && live) ? 16u : 0u;
    // ---- child records.  A lane's work is the list of its (child, left-extension interval) pairs, child-major: most lanes have
    // one pair, one in ten has two, so a wave runs about two rounds instead of (most children) x (most intervals).  Per pair: LF
    // with the child's base at both ends of the parent's interval (EnumerateQuery.cpp:44-55) -- an end that coincides with sp or
    // ep + 1 is the child's own end -- and the child keeps the interval if it stays non-empty, compacted into its first slots.
    // The ends lie inside [sp, ep + 1], i.e. in one of the two parked blocks unless the interval spans more than two blocks.
    if constexpr (!INC) {
        // ---- wide levels (the few at the top of a prefix): interval-major.  Their nodes have up to four children and four intervals
        // whose ends lie anywhere in a huge interval -- blocks far from the two staged ones.  The pair list below would fetch a block per
        // (child, interval end) -- up to 32 dependent round trips per node, 300 us per launch for a level of a few thousand nodes, a tenth
        // of a pass in 36 launches.  Here an interval's two ends are ranked ONCE, for all four bases (two independent block loads issued
        // together), and every child takes its base's pair of ranks: at most eight loads per node, four rounds.
        u32 cn4 = 0, cm4 = 0;   // per child: intervals kept so far (3 bits each), their mask (4 bits each)
        P kl0[4] = {0, 0, 0, 0}, kh0[4] = {0, 0, 0, 0}, kl1[4] = {0, 0, 0, 0}, kh1[4] = {0, 0, 0, 0};  // (compact children only: their first two)
        u32 ko2[4] = {0, 0, 0, 0}, ko3[4] = {0, 0, 0, 0};   // (... of 64-bit positions: the other two as offsets, see store_child)
        const u32 kkpack = bit_list(emask);
        P prevx = 0, prevR[4] = {0, 0, 0, 0};
#pragma nounroll
        for (u32 e = 0; __any(e < ne); ++e) {
            const bool act = live && e < ne;
            const u32 kk = (kkpack >> (2 * e)) & 3u;
            const P xmin = e == 0 ? nd.e0min : (e == 1 ? nd.e1min : (e == 2 ? nd.e2min : nd.e3min));
            const P xmax = e == 0 ? nd.e0max : (e == 1 ? nd.e1max : (e == 2 ? nd.e2max : nd.e3max));
            const P xl = act ? xmin : sp, xh = act ? xmax + 1 : ep1;
            const P bl = xl >> BLK_SHIFT, bh = xh >> BLK_SHIFT;
            const bool l_sp = xl == sp, l_prev = e > 0 && xl == prevx, h_ep = xh == ep1;
            const bool l_stag = bl == b0 || bl == b1, h_stag = bh == b0 || bh == b1;
            const bool farl = act && !l_sp && !l_prev && !l_stag, farh = act && !h_ep && !h_stag;
            Blk16 Bl, Bh;
            if (__any(farl || farh)) {  // both requests before either is used
                load_blk(ix.blk, (u64)(farl ? bl : (P)0), Bl);
                load_blk(ix.blk, (u64)(farh ? bh : (P)0), Bh);
                acc.lines += mask_count(__ballot(farl)) + mask_count(__ballot(farh));
            }
            P Rl[4], Rh[4];
            {
                Blk16 t;
                staged_blk(wl, bl != b0 ? idx1 : idx0, t);
                if (farl) t = Bl;
                rank4_blk<P, ONESB>(a.sb, sbl, t, xl, Rl);
                staged_blk(wl, bh != b0 ? idx1 : idx0, t);
                if (farh) t = Bh;
                rank4_blk<P, ONESB>(a.sb, sbl, t, xh, Rh);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (l_sp) Rl[c] = Rsp[c];
                if (l_prev) Rl[c] = prevR[c];   // adjacent intervals share an end
                if (h_ep) Rh[c] = Rep[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const P l = Rl[c], h = Rh[c];
                if (act && ((present >> c) & 1u) && l <= h - 1) {  // EnumerateQuery.cpp:44-55: the child keeps the interval if it stays non-empty
                    const u32 cnc = (cn4 >> (3 * c)) & 7u;
                    if (OUTC && cnc == 0) { kl0[c] = l; kh0[c] = h - 1; }
                    else if (OUTC && cnc == 1) { kl1[c] = l; kh1[c] = h - 1; }
                    else if (slots_inline<P>(OUTC)) {
                        const u32 o = (u32)(l - Rsp[c]) | ((u32)(h - 1 - Rsp[c]) << 16);
                        if (cnc == 2) ko2[c] = o; else ko3[c] = o;
                    }
                    else { out[(size_t)(2 + 2 * cnc) * cap + qa[c]] = l; out[(size_t)(3 + 2 * cnc) * cap + qa[c]] = h - 1; }
                    cn4 += 1u << (3 * c);
                    cm4 |= 1u << (4 * c + kk);
                }
                prevR[c] = Rh[c];
            }
            prevx = xh;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((present >> c) & 1u) {
                const u32 cnc = (cn4 >> (3 * c)) & 7u, cmc = (cm4 >> (4 * c)) & 15u;
                if (OUTC) store_child<P, true>(out, cap, qa[c], Rsp[c], Rep[c] - 1, kl0[c], kh0[c], kl1[c], kh1[c], cnc, cmc | branch_bit, ko2[c], ko3[c]);
                else {
                    out[qa[c]] = Rsp[c];
                    out[cap + qa[c]] = Rep[c] - 1;
                    reinterpret_cast<u8*>(out + (size_t)REC_FIELDS * cap)[qa[c]] = (u8)(cmc | branch_bit);
                }
                if (!OUTC) acc.rb_lane += (cnc < 2 ? cnc : 2u) * 2u * (u32)sizeof(P);
                if (cnc > 2 && !slots_inline<P>(OUTC)) acc.rb_lane += (cnc - 2) * 2u * (u32)sizeof(P);
            }
        }
    } else
    {
        const u32 ne1 = ne ? ne : 1u;
        const u32 npair = k * ne1;
        const u32 cjpack = bit_list(present), kkpack = bit_list(emask);  // two bits per slot: bases of the children / of the intervals, in order
        u32 cn = 0, cm = 0;               // number and mask of the intervals the current child has kept
        P kl0 = 0, kh0 = 0, kl1 = 0, kh1 = 0;  // the first two of them
        u32 ko2 = 0, ko3 = 0;                  // (64-bit positions, compact children: the other two as offsets, see store_child)
        P prevx = 0, prevh = 0;                // upper end of the previous pair's interval and its rank (same child when e > 0)
        u32 j = 0, e = 0;  // the pair at hand: child j (in base order), interval e of the parent
#pragma nounroll
        for (u32 p = 0; __any(p < npair); ++p) {
            const bool act = p < npair;
            const u32 c = (cjpack >> (2 * j)) & 3u, kk = (kkpack >> (2 * e)) & 3u;
            const u32 q = DSM_PICK(qa, c);
            const P nsp = DSM_PICK(Rsp, c), nep1 = DSM_PICK(Rep, c);  // the child's interval is [nsp, nep1 - 1]
            if (e == 0) { cn = 0; cm = 0; }
            const bool hasext = act && ne > 0;
            P xmin = e == 0 ? nd.e0min : nd.e1min, xmax = e == 0 ? nd.e0max : nd.e1max;
            if constexpr (slots_inline<P>(INC)) {
                if (e > 1) { xmin = e == 2 ? nd.e2min : nd.e3min; xmax = e == 2 ? nd.e2max : nd.e3max; }
            } else if (__any(hasext && e > 1)) {
                if (hasext && e > 1) {
                    xmin = rec[(size_t)(2 + 2 * e) * capi + nd.r];
                    xmax = rec[(size_t)(3 + 2 * e) * capi + nd.r];
                }
            }
            const P xl = hasext ? xmin : sp, xh = hasext ? xmax + 1 : ep1;
            P l = nsp, h = nep1;
            bool needl = xl != sp;
            const bool needh = xh != ep1;
            if (e > 0 && xl == prevx) { l = prevh; needl = false; }  // adjacent intervals share an end: the rank is the previous pair's
            if (__any(needl)) {
                if (needl) {
                    const P bl = xl >> BLK_SHIFT;
                    if (bl == b0 || bl == b1) l = rank_staged<P, ONESB>(a.sb, sbl, wl, bl != b0 ? idx1 : idx0, xl, c);
                    else l = rank_load<P, ONESB>(ix, a.sb, sbl, xl, c);
                }
                acc.lines += mask_count(__ballot(needl && (xl >> BLK_SHIFT) != b0 && (xl >> BLK_SHIFT) != b1));
            }
            if (__any(needh)) {
                if (needh) {
                    const P bh = xh >> BLK_SHIFT;
                    if (bh == b0 || bh == b1) h = rank_staged<P, ONESB>(a.sb, sbl, wl, bh != b0 ? idx1 : idx0, xh, c);
                    else h = rank_load<P, ONESB>(ix, a.sb, sbl, xh, c);
                }
                acc.lines += mask_count(__ballot(needh && (xh >> BLK_SHIFT) != b0 && (xh >> BLK_SHIFT) != b1));
            }
            prevx = xh; prevh = h;
            if (hasext && l <= h - 1) {
                if (cn == 0) { kl0 = l; kh0 = h - 1; }
                else if (cn == 1) { kl1 = l; kh1 = h - 1; }
                else if (slots_inline<P>(OUTC)) {
                    const u32 o = (u32)(l - nsp) | ((u32)(h - 1 - nsp) << 16);
                    if (cn == 2) ko2 = o; else ko3 = o;
                }
                else { out[(size_t)(2 + 2 * cn) * cap + q] = l; out[(size_t)(3 + 2 * cn) * cap + q] = h - 1; }  // slots 2, 3: wide fields
                ++cn;
                cm |= 1u << kk;
            }
            if (act && e == ne1 - 1) {
                store_child<P, OUTC>(out, cap, q, nsp, nep1 - 1, kl0, kh0, kl1, kh1, cn, cm | branch_bit, ko2, ko3);
                // (bytes of the child's record beyond the compact word / the fixed fields: rare or wide levels only)
                if (!OUTC) acc.rb_lane += (cn < 2 ? cn : 2u) * 2u * (u32)sizeof(P);
                if (cn > 2 && !slots_inline<P>(OUTC)) acc.rb_lane += (cn - 2) * 2u * (u32)sizeof(P);
            }
            ++e;
            if (e >= ne1) { e = 0; ++j; }
        }
    }
    if (DENSE ? live : i < a.F) {   // (DENSE: the entries of the nodes the sample does not have were cleared by the sweep's producer)
        // this node's column entry: its frequency in this sample (0 = absent), which children survive, its left char
        if (a.w16 == 2) {
            reinterpret_cast<u16*>(valf)[(size_t)i * a.cstride] = live ? (u16)((u32)(ep - sp + 1) | ((present | (mycode << 4)) << 9)) : (u16)0;
        } else {
            if (a.w16) reinterpret_cast<u16*>(valf)[i] = live ? (u16)(ep - sp + 1) : (u16)0;
            else valf[i] = live ? (P)(ep - sp + 1) : (P)0;
            pl[i] = (u8)(present | (mycode << 4));
        }
    }
    {   // the wave's sums (scalar): children, records read, record bytes
        const u32 nkids = mask_count(bal[0]) + mask_count(bal[1]) + mask_count(bal[2]) + mask_count(bal[3]);
        const u32 nl = mask_count(__ballot(live));
        acc.kne += (a.symbol_phase & 2u) ? nkids : 0u;
        acc.live += nl;
        // its own record (compact word, or sp, ep, mask and the two slots the head always reads; slots 2, 3 when in use) and its
        // children's (compact word each, or their fixed fields; what depends on the kept intervals is in rb_lane)
        acc.rbytes += nl * (INC ? 16u * CREC_WORDS(sizeof(P)) : (u32)(6 * sizeof(P) + 1)) + nkids * (OUTC ? 16u * CREC_WORDS(sizeof(P)) : (u32)(2 * sizeof(P) + 1));
        if (!slots_inline<P>(INC) && __any(ne > 2)) acc.rbytes += (mask_count(__ballot(ne > 2)) + mask_count(__ballot(ne > 3))) * 2u * (u32)sizeof(P);
    }
}

// One sample's LF-step sweep over a level: the body of expand_kernel (one sample per launch) and of expand_batch_kernel.
// tile_ctr: the workgroup's tile counter in LDS, zero when the sweep starts (the caller's barrier).
// DENSE (SELF, compact records): with d samples a sample holds a fraction of the union level's nodes -- half of them with eight
// 1-Gbase samples, a fifth with 64 -- and a sweep over the union tiles runs its instruction stream for the absent ones as well.  Here
// a wave takes an ITEM of consecutive union tiles, a PRODUCER step turns two of them at a time into handles (slots and parent
// planes, as the sparse sweep does), clears the column entries of the absent nodes and appends the present ones -- handle and place
// in the union level -- to a queue in LDS; whenever the queue holds two tiles' worth (or the item is used up) the wave runs the
// tile body on the next 64 queued nodes, with the heads of the 64 after them requested inside it as before.  An item's plane words
// are put together in LDS and written when it is done.  Nothing crosses items: their last tile is the only short one.
template <typename P, bool ONESB, bool INC, bool OUTC, bool SELF, bool DENSE = false>
__device__ __forceinline__ void expand_sweep(const DevIndex& ix, u64* sbl, uint4* parked, u32* tile_ctr, const u32* __restrict__ rp, const P* __restrict__ rec,
                                             P* __restrict__ out, u64* __restrict__ splane, u32* __restrict__ cnt, P* __restrict__ valf,
                                             u8* __restrict__ pl, const ExpandArgs& a, u64* __restrict__ counters,
                                             unsigned long long* __restrict__ childmax, const u64* __restrict__ pplane, u32* dense_lds = nullptr) {
    constexpr int WPB = LfShape<P, INC, DENSE>::WPB;
    uint4* wl = parked + (threadIdx.x >> 6) * WAVE_LDS_WORDS;
    if (!ONESB) {
        const u32 nsb4 = (u32)((ix.n >> SB_SHIFT) + 1) * 4;
        for (u32 q = threadIdx.x; q < nsb4 && q < SB_LDS_MAX * 4; q += blockDim.x) sbl[q] = ix.sbase[q];
    }
    if (LF_DYNAMIC && threadIdx.x == 0) *tile_ctr = 0;
    if (!ONESB || LF_DYNAMIC) __syncthreads();
    const int lane = threadIdx.x & 63;
    const u32 gw = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * WPB + (threadIdx.x >> 6)));
    const u32 ntile = (a.F + 63) >> 6;
    const u32* keeptab = reinterpret_cast<const u32*>(counters + (size_t)COUNTER_SHARDS * 8);  // (see KEEP_FREQS)
    ExpandAcc acc;
    const u32 cost_pack = (a.cost[0] & 15u) | ((a.cost[1] & 15u) << 4) | ((a.cost[2] & 15u) << 8) | ((a.cost[3] & 15u) << 12);
#ifdef DSM_CLOCK_PROBE
    const u64 probe_c0 = __builtin_readcyclecounter(), probe_r0 = wall_clock64();
#endif
    TileSeq<WPB> seq;
    seq.ctr = tile_ctr; seq.G = gridDim.x; seq.b = blockIdx.x; seq.stride = gridDim.x * WPB;
    u32 tA = ~0u, tB = ~0u, tC = ~0u;
    if constexpr (DENSE) {
        static_assert(SELF && INC && LF_DYNAMIC, "the dense sweep is the several-sample sweep over compact records");
        u32* qh = dense_lds + (threadIdx.x >> 6) * DENSE_LDS_WORDS;   // queue: handles
        unsigned short* qi = reinterpret_cast<unsigned short*>(qh + DENSE_Q);   //  places in the union level, relative to the item's first
        u64* lpl = reinterpret_cast<u64*>(qh + DENSE_Q + DENSE_Q / 2);          // plane words of the item's union tiles
        const u32 item = a.item_tiles;
        const u32 nitem = (ntile + item - 1) / item;
        u32 dummy_t = 0, dummy_r = DEAD;
        DenseTile dt;
        dt.lpl = lpl;
        for (u32 it = seq.next(); it < nitem; it = seq.next()) {
            const u32 T0 = it * item, T1 = T0 + item < ntile ? T0 + item : ntile;
            tA = 0;  // (DSM_CLOCK_PROBE: the wave had work)
            for (u32 w = (u32)lane; w < item * 4; w += 64) lpl[w] = 0;
            dt.T0 = T0; dt.lastT = ~0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) dt.carry[c] = 0;
            u32 qhead = 0, qtail = 0;
            bool have_heads = false;
            RecHead<P, INC> hA, hB;
            // The producer is a pipeline of its own, two steps deep, so that a tile body hides both of its round trips: the slots of
            // step k + 2 are requested when the parent planes of step k + 1 are (their slots have arrived) and step k is queued (its
            // planes have).  chunkS: first union tile of the next slot request; chunkQ: of the step that is queued next.
            // (Measured and not kept: starting the NEXT item's first requests while the last tiles of an item are worked on, so that an
            // item start costs one round trip instead of four -- the state that has to live outside the item loop takes 16 more
            // registers, and the LF-step sweeps of eight samples went from 795-808 to 846-860 ms per pass.)
            constexpr u32 STEP = sizeof(P) == 4 ? (u32)DSM_DENSE_STEP32 : 2u;
            u32 s1[STEP], s2[STEP];   // slots of the step whose slots / planes are on their way
            u64 p2[STEP];             // parent planes of the second
            u32 chunkS = T0, chunkQ = T0;
            auto request_slots = [&](u32* sl) {
#pragma unroll
                for (u32 b = 0; b < STEP; ++b) {
                    const u32 iu = (chunkS + b) * 64 + lane;
                    const bool in = chunkS + b < T1 && iu < a.F;
                    const u32 v = rp[in ? iu : 0u];
                    sl[b] = in ? v : DEAD;
                }
                chunkS += STEP;
            };
            auto advance = [&]() {
                // queue the step whose planes have arrived
#pragma unroll
                for (u32 b = 0; b < STEP; ++b) {
                    const u32 iu = (chunkQ + b) * 64 + lane;
                    const bool in = chunkQ + b < T1 && iu < a.F;
                    const u32 h = self_handle(s2[b], p2[b], a.seg_in);
                    const bool has = h != DEAD;
                    if (in && !has) {   // the sample does not hold the node: an empty column entry
                        if (a.w16 == 2) reinterpret_cast<u16*>(valf)[(size_t)iu * a.cstride] = (u16)0;
                        else {
                            if (a.w16) reinterpret_cast<u16*>(valf)[iu] = (u16)0;
                            else valf[iu] = (P)0;
                            pl[iu] = (u8)0;
                        }
                    }
                    const u64 m = __ballot(has);
                    if (has) {
                        const u32 pos = (qtail + lf_bits_below_lane(m)) & (DENSE_Q - 1);
                        qh[pos] = h; qi[pos] = (unsigned short)(iu - T0 * 64);
                    }
                    qtail += (u32)__popcll(m);
                }
                chunkQ += STEP;
                // the next step's planes (its slots have arrived), the slots of the one after
#pragma unroll
                for (u32 b = 0; b < STEP; ++b) { s2[b] = s1[b]; p2[b] = pplane[self_plane_index(s1[b])]; }
                request_slots(s1);
            };
            // prime: step 0's planes and step 1's slots on their way
            request_slots(s2);
#pragma unroll
            for (u32 b = 0; b < STEP; ++b) p2[b] = pplane[self_plane_index(s2[b])];
            request_slots(s1);
            // one step: the queue is topped up, then one dense tile is worked on with its heads in hc (the next one's go to hn)
            auto step = [&](RecHead<P, INC>& hc, RecHead<P, INC>& hn) -> bool {
                while (qtail - qhead < 128u && chunkQ < T1) advance();
                const u32 qc = qtail - qhead;
                if (qc == 0) return false;
                const u32 n = qc < 64u ? qc : 64u, nn = qc - n < 64u ? qc - n : 64u;
                asm volatile("" ::: "memory");  // (the queue entries other lanes wrote)
                dt.iu = (u32)lane < n ? T0 * 64 + (u32)qi[(qhead + lane) & (DENSE_Q - 1)] : 0u;
                if (!have_heads) {   // the item's first tile: its heads were not requested by a tile before it
                    const u32 hv = (u32)lane < n ? qh[(qhead + lane) & (DENSE_Q - 1)] : DEAD;
                    load_head<P>(rec, a.cap_in, hv, hc);
                    have_heads = true;
                }
                dt.rnext = (u32)lane < nn ? qh[(qhead + n + lane) & (DENSE_Q - 1)] : DEAD;
                expand_tile<P, ONESB, INC, OUTC, SELF, WPB, true>(ix, sbl, wl, rp, rec, out, splane, cnt, valf, pl, a, 0u, seq, dummy_t, ntile, hc, hn, dummy_r, acc, pplane,
                                                                  nullptr, keeptab, cost_pack, &dt);
                qhead += n;
                return true;
            };
            while (step(hA, hB) && step(hB, hA)) {}
            asm volatile("" ::: "memory");
            for (u32 w = (u32)lane; w < (T1 - T0) * 4; w += 64) splane[(size_t)T0 * 4 + w] = lpl[w];
        }
    } else {
    if (LF_DYNAMIC) { tA = seq.next(); tB = seq.next(); if (SELF) tC = seq.next(); }
    else {
        tA = gw; tB = ~gw < seq.stride ? ~0u : gw + seq.stride;
        seq.last = tB;
        if (SELF) tC = seq.next();
    }
    {
        // prologue of the pipeline: heads of the wave's first tile, handles of its second
        u32 rn = DEAD;
        u32 r0 = DEAD;
        const u32 i0 = tA * 64 + lane, i1 = tB * 64 + lane;
        if (tA < ntile && i0 < a.F) r0 = rp[i0];
        if (tB < ntile && i1 < a.F) rn = rp[i1];
        SelfState ss;
        if (SELF) {  // r0, rn hold slots here: handles of the first tile now, the deeper stages primed
            const u32 i2 = tC * 64 + lane;
            ss.s1 = rn;
            if (tC < ntile && i2 < a.F) ss.s2 = rp[i2];
            const u64 p0 = pplane[self_plane_index(r0)];
            ss.pn = pplane[self_plane_index(ss.s1)];
            r0 = self_handle(r0, p0, a.seg_in);
        }
        RecHead<P, INC> hA, hB;
        load_head<P>(rec, a.cap_in, r0, hA);
        // two tiles per trip, the two head sets swapping roles: no register that a load is still filling is ever copied.
        // t0 / t1 / t2: the wave's current tile and the ones whose heads / slots are on their way (t2: SELF only).
        u32 t0 = tA, t1 = tB, t2 = tC, tf = ~0u;
        while (t0 < ntile) {
            expand_tile<P, ONESB, INC, OUTC, SELF, WPB>(ix, sbl, wl, rp, rec, out, splane, cnt, valf, pl, a, t0, seq, tf, ntile, hA, hB, rn, acc, pplane, &ss, keeptab, cost_pack);
            t0 = t1;
            if (SELF) { t1 = t2; t2 = tf; } else t1 = tf;
#ifdef DSM_LF_ONE_COPY   // experiment: one copy of the tile's code, the heads moved between the trips
            hA = hB;
#else
            if (t0 >= ntile) break;
            expand_tile<P, ONESB, INC, OUTC, SELF, WPB>(ix, sbl, wl, rp, rec, out, splane, cnt, valf, pl, a, t0, seq, tf, ntile, hB, hA, rn, acc, pplane, &ss, keeptab, cost_pack);
            t0 = t1;
            if (SELF) { t1 = t2; t2 = tf; } else t1 = tf;
#endif
        }
    }
    }
#ifdef DSM_CLOCK_PROBE
    if (gw == 0 && lane == 0) {  // shader clocks and 100 MHz ticks of this wave's sweep (build-time probe: the clock the kernel runs at)
        atomicAdd((unsigned long long*)&counters[6], (unsigned long long)(__builtin_readcyclecounter() - probe_c0));
        atomicAdd((unsigned long long*)&counters[7], (unsigned long long)(wall_clock64() - probe_r0));
    }
    if (lane == 0 && tA < ntile) {  // every wave with tiles: its start and end (100 MHz ticks): earliest, latest and sum of both, per launch (three shards)
        const unsigned long long e1 = (unsigned long long)wall_clock64(), s1 = (unsigned long long)probe_r0;
        unsigned long long* q = (unsigned long long*)&counters[(size_t)a.probe_slot * 8];
        atomicMax(q + 6, ~s1); atomicMax(q + 7, s1);
        atomicMax(q + 8 + 6, ~e1); atomicMax(q + 8 + 7, e1);
        atomicAdd(q + 16 + 6, s1 & 0xFFFFFFFFull); atomicAdd(q + 16 + 7, e1 & 0xFFFFFFFFull);
    }
#endif
    // ---- counters (exact; the block lines include the ones the ext pass fetched): one reduction per wave and launch ----
    if ((acc.wide & 3u) != 0 && lane == 0) atomicMax(childmax, (acc.wide & 2u) ? 65535ull : (unsigned long long)PACK_FMAX);  // only the class matters
    {
        const u32 extra_bytes = (u32)lf_wave_sum_u64(acc.rb_lane);  // (record bytes that depend on the intervals a lane's children kept: rare)
        const u32 v[NCOUNTERS] = {acc.kne, acc.lf, acc.rank, acc.lines, acc.rbytes + extra_bytes, acc.live};
        if (lane < NCOUNTERS) {
            u32 mine = v[0];
#pragma unroll
            for (int q = 1; q < NCOUNTERS; ++q) mine = lane == q ? v[q] : mine;
            if (mine) atomicAdd((unsigned long long*)&counters[(size_t)(gw & (COUNTER_SHARDS - 1)) * 8 + lane], (unsigned long long)mine);
        }
    }
}

// (32-bit positions fit four waves per SIMD without spilling when the allocator is told to aim for it; 64-bit positions take three)
template <typename P, bool ONESB, bool INC, bool OUTC>
__global__ __launch_bounds__((LfShape<P, INC>::WPB * 64)) __attribute__((amdgpu_waves_per_eu((LfShape<P, INC>::WAVES_PER_SIMD))))
void expand_kernel(DevIndex ix, const u32* __restrict__ rp, const P* __restrict__ rec, P* __restrict__ out, u64* __restrict__ splane, u32* __restrict__ cnt,
                   P* __restrict__ valf, u8* __restrict__ pl, ExpandArgs a, u64* __restrict__ counters, unsigned long long* __restrict__ childmax) {
    __shared__ u64 sbl[ONESB ? 1 : SB_LDS_MAX * 4];
    __shared__ uint4 parked[LfShape<P, INC>::WPB * WAVE_LDS_WORDS];
    __shared__ u32 tile_ctr;
    if (a.dyn) {  // (uniform over the grid: every block takes the same way out)
        const u32 F = a.dyn[0], cls = a.dyn[1];
        if ((cls & a.dyn_mask) != a.dyn_expect || F > a.fcap || F == 0) return;
        a.F = F;
        a.nbp = (F + TILE - 1) / TILE;
        if (a.w16 != 2) pl = reinterpret_cast<u8*>(valf) + (size_t)F * (a.w16 ? 2u : (u32)sizeof(P));  // one sample: the flag bytes follow its frequencies
        if (a.nbp <= 1) cnt = nullptr;
    }
    expand_sweep<P, ONESB, INC, OUTC, false>(ix, sbl, parked, &tile_ctr, rp, rec, out, splane, cnt, valf, pl, a, counters, childmax, nullptr);
}

template <typename P, bool ONESB, bool INC, bool OUTC>
__global__ __launch_bounds__((LfShape<P, INC>::WPB * 64)) __attribute__((amdgpu_waves_per_eu((LfShape<P, INC>::WAVES_PER_SIMD))))
void expand_batch_kernel(ExpandBatch b, ExpandArgs a, u64* __restrict__ counters, unsigned long long* __restrict__ childmax) {
    __shared__ u64 sbl[ONESB ? 1 : SB_LDS_MAX * 4];
    __shared__ uint4 parked[LfShape<P, INC>::WPB * WAVE_LDS_WORDS];
    __shared__ u32 tile_ctr;
    const ExpandSample& S = b.s[blockIdx.y];
    a.sb = S.sb;
#pragma unroll
    for (int c = 0; c < 4; ++c) a.cost[c] = S.cost[c];
    a.access_pack = S.access_pack;
    a.costsum_lo = S.costsum_lo;
    a.costsum_hi = S.costsum_hi;
    expand_sweep<P, ONESB, INC, OUTC, true>(S.ix, sbl, parked, &tile_ctr, S.rp, (const P*)S.rec, (P*)S.out, S.splane, nullptr, (P*)S.valf, S.pl, a, counters, childmax, S.pplane);
}

// the same, every sample's own nodes packed into full tiles (compact records only: LfConfig::dense)
template <typename P, bool ONESB, bool OUTC>
__global__ __launch_bounds__((LfShape<P, true, true>::WPB * 64)) __attribute__((amdgpu_waves_per_eu((LfShape<P, true, true>::WAVES_PER_SIMD))))
void expand_dense_kernel(ExpandBatch b, ExpandArgs a, u64* __restrict__ counters, unsigned long long* __restrict__ childmax) {
    __shared__ u64 sbl[ONESB ? 1 : SB_LDS_MAX * 4];
    __shared__ uint4 parked[LfShape<P, true, true>::WPB * WAVE_LDS_WORDS];
    __shared__ u32 dense_lds[LfShape<P, true, true>::WPB * DENSE_LDS_WORDS];
    __shared__ u32 tile_ctr;
    const ExpandSample& S = b.s[blockIdx.y];
    a.sb = S.sb;
#pragma unroll
    for (int c = 0; c < 4; ++c) a.cost[c] = S.cost[c];
    a.access_pack = S.access_pack;
    a.costsum_lo = S.costsum_lo;
    a.costsum_hi = S.costsum_hi;
    expand_sweep<P, ONESB, true, OUTC, true, true>(S.ix, sbl, parked, &tile_ctr, S.rp, (const P*)S.rec, (P*)S.out, S.splane, nullptr, (P*)S.valf, S.pl, a, counters, childmax,
                                                   S.pplane, dense_lds);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename P>
static int geometry_t(int device, LfGeometry* g) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return fail(DSM_E_HIP, "hipDeviceGetAttribute failed");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (expand_kernel<P, true, true, true>), LfShape<P>::WPB * 64, 0) != hipSuccess)
        return fail(DSM_E_HIP, "hipOccupancyMaxActiveBlocksPerMultiprocessor failed");
    if (const char* e = getenv("DSM_EXPAND_BLOCKS_PER_CU")) per = atoi(e);
    if (per < 1) per = 1;
    g->blocks = (u32)(cus > 0 ? cus : 256) * (u32)per;
    g->waves_per_block = (u32)LfShape<P>::WPB;
    return 0;
}
int lf_step_geometry(bool wide_pos, int device, LfGeometry* g) { return wide_pos ? geometry_t<u64>(device, g) : geometry_t<u32>(device, g); }

template <typename P, bool SB, bool IC, bool OC>
static void launch1(dim3 grid, hipStream_t st, const DevIndex& ix, const u32* rp, const void* rec, void* out, u64* splane, u32* cnt, void* valf, u8* pl,
                    const ExpandArgs& a, u64* counters, unsigned long long* childmax) {
    hipLaunchKernelGGL((expand_kernel<P, SB, IC, OC>), grid, dim3(LfShape<P, IC>::WPB * 64), 0, st, ix, rp, (const P*)rec, (P*)out, splane, cnt, (P*)valf, pl, a, counters,
                       childmax);
}
template <typename P, bool SB, bool IC, bool OC>
static void launchb(dim3 grid, hipStream_t st, const ExpandBatch& b, const ExpandArgs& a, u64* counters, unsigned long long* childmax) {
    hipLaunchKernelGGL((expand_batch_kernel<P, SB, IC, OC>), grid, dim3(LfShape<P, IC>::WPB * 64), 0, st, b, a, counters, childmax);
}
// (the formats are template parameters: eight instantiations per position type)
#define DSM_LF_DISPATCH(FN, P, ...)                                                             \
    do {                                                                                        \
        const int key_ = (c.one_sb ? 4 : 0) | (c.fmt_in ? 2 : 0) | (c.fmt_out ? 1 : 0);         \
        switch (key_) {                                                                         \
            case 0: FN<P, false, false, false>(__VA_ARGS__); break;                             \
            case 1: FN<P, false, false, true>(__VA_ARGS__); break;                              \
            case 2: FN<P, false, true, false>(__VA_ARGS__); break;                              \
            case 3: FN<P, false, true, true>(__VA_ARGS__); break;                               \
            case 4: FN<P, true, false, false>(__VA_ARGS__); break;                              \
            case 5: FN<P, true, false, true>(__VA_ARGS__); break;                               \
            case 6: FN<P, true, true, false>(__VA_ARGS__); break;                               \
            default: FN<P, true, true, true>(__VA_ARGS__); break;                               \
        }                                                                                       \
    } while (0)

void lf_step_launch(const LfConfig& c, const LfGeometry& g, u64 tiles_bound, hipStream_t st, const DevIndex& ix, const u32* rp, const void* rec, void* out,
                    u64* splane, u32* cnt, void* valf, u8* pl, const ExpandArgs& a, u64* counters, unsigned long long* childmax) {
    const u32 wpb = c.fmt_in ? g.waves_per_block : (u32)(c.wide_pos ? LfShape<u64, false>::WPB : LfShape<u32, false>::WPB);  // (wide levels: smaller workgroups)
    u64 need = (tiles_bound + wpb - 1) / wpb;  // workgroups that get a tile at all
    if (need < 1) need = 1;
    const dim3 grid((u32)(need < g.blocks ? need : g.blocks));
    if (c.wide_pos) DSM_LF_DISPATCH(launch1, u64, grid, st, ix, rp, rec, out, splane, cnt, valf, pl, a, counters, childmax);
    else DSM_LF_DISPATCH(launch1, u32, grid, st, ix, rp, rec, out, splane, cnt, valf, pl, a, counters, childmax);
}

void lf_step_launch_batch(const LfConfig& c, const LfGeometry& g, u32 grid_factor, int nb, hipStream_t st, const ExpandBatch& b, const ExpandArgs& a, u64* counters,
                          unsigned long long* childmax) {
    // The samples of the launch share the card: grid_factor x (resident workgroups / samples) workgroups each.  Measured with eight
    // full-size samples and whole-CU workgroups: factor 1 -> 878, 2 -> 893, 4 -> 903, 8 -> 940 ms of LF-step sweeps per pass (every further
    // workgroup is another start-up, and a workgroup's waves balance their tiles among themselves anyway); round 3's workgroups of four
    // waves wanted 8 (972 ms).
    const u32 ntile = (a.F + 63) >> 6;
    u32 gx = (grid_factor ? grid_factor : 1u) * g.blocks / (u32)nb;
    if (gx < 1) gx = 1;
    if (c.dense && c.fmt_in) {
        // items of union tiles: as large as leaves every wave of the sample's share of the card a few of them (a short last tile per item)
        const u32 wpbd = (u32)LfShape<u32, true, true>::WPB;
        ExpandArgs ad = a;
        u32 item = DENSE_ITEM_MAX;
        while (item > 4 && (u64)ntile < (u64)item * gx * wpbd * 4) item >>= 1;
        ad.item_tiles = item;
        const u32 nitem = (ntile + item - 1) / item;
        u32 needd = (nitem + wpbd - 1) / wpbd;
        if (needd < 1) needd = 1;
        const dim3 gridd(needd < gx ? needd : gx, (u32)nb);
        const dim3 blk(wpbd * 64);
        if (c.wide_pos) {
            if (c.one_sb) { if (c.fmt_out) hipLaunchKernelGGL((expand_dense_kernel<u64, true, true>), gridd, blk, 0, st, b, ad, counters, childmax);
                            else hipLaunchKernelGGL((expand_dense_kernel<u64, true, false>), gridd, blk, 0, st, b, ad, counters, childmax); }
            else          { if (c.fmt_out) hipLaunchKernelGGL((expand_dense_kernel<u64, false, true>), gridd, blk, 0, st, b, ad, counters, childmax);
                            else hipLaunchKernelGGL((expand_dense_kernel<u64, false, false>), gridd, blk, 0, st, b, ad, counters, childmax); }
        } else {
            if (c.one_sb) { if (c.fmt_out) hipLaunchKernelGGL((expand_dense_kernel<u32, true, true>), gridd, blk, 0, st, b, ad, counters, childmax);
                            else hipLaunchKernelGGL((expand_dense_kernel<u32, true, false>), gridd, blk, 0, st, b, ad, counters, childmax); }
            else          { if (c.fmt_out) hipLaunchKernelGGL((expand_dense_kernel<u32, false, true>), gridd, blk, 0, st, b, ad, counters, childmax);
                            else hipLaunchKernelGGL((expand_dense_kernel<u32, false, false>), gridd, blk, 0, st, b, ad, counters, childmax); }
        }
        return;
    }
    const u32 wpb = c.fmt_in ? g.waves_per_block : (u32)(c.wide_pos ? LfShape<u64, false>::WPB : LfShape<u32, false>::WPB);
    u32 need = (ntile + wpb - 1) / wpb;
    if (need < 1) need = 1;
    const dim3 grid(need < gx ? need : gx, (u32)nb);
    if (c.wide_pos) DSM_LF_DISPATCH(launchb, u64, grid, st, b, a, counters, childmax);
    else DSM_LF_DISPATCH(launchb, u32, grid, st, b, a, counters, childmax);
}

}  // namespace dsm
