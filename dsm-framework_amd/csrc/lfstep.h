// lfstep.h -- the LF-step kernel's interface to the rest of the engine (gfx950 only).
//
// expand.hip holds the kernels (EnumerateQuery::pushChar / leftChar / nextSymbol on a whole frontier level, EnumerateQuery.cpp:39-238)
// and their launchers; the engine (engine.hip) fills an ExpandArgs per level and calls lf_step_launch / lf_step_launch_batch.  The
// kernels are one object file: a variant is tried by swapping expand.o.
#pragma once
#include "common.h"

namespace dsm {

#define DSM_PICK(a, c) ((c) == 0 ? (a)[0] : ((c) == 1 ? (a)[1] : ((c) == 2 ? (a)[2] : (a)[3])))

// Superblock bases (C[c] + occurrences of c before the superblock).  An index below 2^31 symbols has one superblock:
// its four bases travel as kernel arguments (scalar registers).  Larger indexes keep the table in LDS (sbl): a read of it
// is an LDS access, which does not queue behind the global prefetches the way a global load would.  The choice is a template
// parameter (ONESB), not a run-time select: selecting between the argument block and memory turns every access into a flat
// load, whose wait also drains every prefetch in flight.
struct SbArgs {
    u64 sb0[4];
};
constexpr u32 SB_LDS_MAX = 64;  // superblocks the LDS copy holds (2^37 symbols)

// Records (EnumerateQuery.h:44-45, Query.h:110-111): the interval [sp, ep] of a node in the sample and its non-empty
// left-extension intervals, kept in base order in the first popcount(mask) of four slots (mask bit a: the interval of base a
// is non-empty).  Records are addressed through a handle per frontier node (rp[v]); DEAD = the node is absent from this sample.
// Two formats, chosen per level:
//   wide (struct of arrays)  field f of record r lives at rec[f * cap + r]: 0 sp, 1 ep, 2 + 2e / 3 + 2e = min / max of slot e,
//                            followed by one mask byte per record.  Used while frequencies may reach 65535 (the top few levels
//                            of a prefix) and, in every level, for slots 2 and 3 (fewer than one node in a hundred has them).
//   compact                  one 16-byte word per record (32 bytes with 64-bit positions): sp, then 16-bit ep - sp and the
//                            offsets from sp of the ends of slots 0 and 1, then the mask.  A level is compact when every frequency
//                            of its parent level is below 65535.  One load / one store per record instead of seven: the LF-step
//                            kernel is bound by the memory transactions it issues, not by their bytes.  The words share the
//                            memory of fields 0-3 of the wide format (a level has one format).  The 32-byte record of 64-bit
//                            positions has room for the offsets of slots 2 and 3 as well and is complete in itself; the 16-byte one
//                            keeps them in the wide fields 6-9.
//
// Order of a level.  The nodes of a level are kept in COLEX order of their paths (sorted by the reversed substring), not in
// trie order.  The index holds reversed reads, so the suffix-array interval of a substring P is ordered by reverse(P):
// colex order of the union level IS increasing sp order in every sample.  Neighbouring lanes of the LF-step kernel therefore
// read neighbouring records and neighbouring (often the same) index blocks, and write neighbouring column entries.
// LF is monotone, so the children with symbol c of colex-ordered parents are colex-ordered among themselves and every
// c = A child precedes every c = C child ...: the next level is the stable 4-way partition A|C|G|T of the children, and a
// child's place is a prefix count over its symbol's bit plane -- no atomics, no allocation that can overflow.
//   record handle of child (u, c) in a sample = c * seg + 64 * (u / 64) + rank of u among its wave's parents with a child c
// (every wave of 64 parents owns 64 handles per symbol: at most one child per symbol and parent).  The trie order the
// reference prints in is recovered at the end of a prefix from the retained parent links, as before.
constexpr int REC_FIELDS = 10;
template <typename P>
__host__ __device__ constexpr size_t rec_elems(size_t cap) { return (size_t)REC_FIELDS * cap + (cap + sizeof(P) - 1) / sizeof(P); }
constexpr int COUNTER_SHARDS = 1024;  // power of two; each shard is one 64-byte line
// One sample: the exact entropy test of the reference (metaserver.cpp:379-413) depends on the node's frequency alone -- with one reader
// the entropy is rounding noise around 0, and whether it is below emin decides -- so the host tabulates its verdict,
// computed with its own libm expression, for every frequency below KEEP_FREQS, one bit each, BEHIND the counters (no extra kernel
// argument), and the LF-step kernel's candidate ballot reads it: what reaches the host is final, nothing is dropped there any more
// and the tuples go to the sink from the pinned buffers they arrived in.  (Larger frequencies -- a few nodes at the top -- are kept
// and decided by the host as before.)
constexpr u32 KEEP_FREQS = 1u << 22;
constexpr int NCOUNTERS = 6;          // [0]=reported [1]=lf_steps [2]=rank_ops [3]=index lines fetched [4]=record bytes read + written [5]=records read
constexpr u32 DEAD = 0xFFFFFFFFu;
constexpr u32 PACK_FMAX = 512;        // a level whose frequencies are all below this packs frequency and flags of a node into 16 bits
constexpr u32 TILE = 256;             // parents per block of the advance kernels = the unit of the tile counts (four waves of 64)

struct ExpandArgs {
    u32 F;            // frontier width
    u32 cap;          // handle space of the record buffer the children are written to (stride of its wide fields) = 4 * seg
    u32 seg;          // ... handles per symbol segment (>= F rounded up to a tile)
    u32 cap_in;       // the same two numbers of the buffer this level's records were written to: a level picks its handle space by the
    u32 seg_in;       // format of the records (a compact record is a fraction of a wide one, so compact levels may be wider: engine.hip)
    u32 nbp;          // tiles of the level = stride of cnt4
    u32 allowed;      // bit c set: child c may be tried (enforced prefix / maxdepth)
    u32 fmin;
    u32 symbol_phase; // bit 0: node is handled by nextSymbol (size-1 nodes take followOneBranch); bit 1: the children count as reported;
                      // bit 2 (one sample): the level's nodes are tested for output here -- everything in metaserver.cpp:406-419 that does
                      // not depend on the node (depth, pmin, the entropy thresholds against the 0 a single frequency gives) holds;
                      // bit 3 (one sample): a tile's planes form a 64-byte line {plane[4], candidate bits, candidates | pairs << 32, -, -}
    u32 cstride;      // packed column: distance, in words, between the entries of consecutive nodes (1, or the number of local samples
                      // when the level is node-major, see Xchg::nm)
    u32 w16;          // this level's column: 0 = frequencies as P plus a flag byte; 1 = 16-bit frequencies plus a flag byte (every
                      // frequency of the level is below 65535); 2 = ONE 16-bit word per node, frequency in bits 0-8 and the flags
                      // in bits 9-15 (every frequency below 512: all but the top levels of a prefix)
    SbArgs sb;        // superblock bases of this sample's index
    u32 cost[4];      // BitRank::rank calls per LF on A,C,G,T in the reference
    u32 access_pack;  // BitRank::rank calls of getL by 3-bit code, four bits each (a table in the argument block would be a load)
    u64 costsum_lo, costsum_hi;  // sum of cost[c] over the bases of a 4-bit set, six bits per set: sets 0-9, sets 10-15
    // A launch queued before the host knows the level (single sample): width and frequency class of the level come from two device
    // words the previous level's publish kernel wrote; the launch does nothing when the class is not the one the host assumed
    // (formats are launch-time choices) or the level does not fit -- the host then sees the same words and launches again.
    const u32* dyn;    // null: F, nbp and the formats above are final
    u32 dyn_expect, dyn_mask, fcap;
    u32 item_tiles;    // dense sweeps (one sample among several, expand.hip): union tiles a wave takes at a time, a power of two <= 32
    u32 probe_slot;    // DSM_CLOCK_PROBE builds: counter shard that collects this launch's wave times
};

// Several samples of one process in one launch: blockIdx.y picks the sample (its index, record buffers, columns and code costs come
// from the batch block), so a level's launches are not eight short ones with eight tails but one wide one.
struct ExpandSample {
    DevIndex ix;
    const u32* rp;
    const void* rec;
    void* out;
    u64* splane;
    const u64* pplane;  // the planes this sample wrote at the parent level (handles are derived from them, see expand_tile)
    void* valf;
    u8* pl;
    SbArgs sb;
    u32 cost[4];
    u32 access_pack, pad;
    u64 costsum_lo, costsum_hi;
};
constexpr int BATCH_MAX = 8;
struct ExpandBatch {
    ExpandSample s[BATCH_MAX];
};

// ---- launchers (expand.hip) ---------------------------------------------------------------------------------------------
struct LfConfig {
    bool wide_pos;   // 64-bit positions (P = u64)
    bool one_sb;     // every index of the launch has one superblock
    bool fmt_in;     // this level's records are compact
    bool fmt_out;    // the children's records are compact
    bool dense;      // lf_step_launch_batch: pack every sample's own nodes into full tiles (compact levels that are wide enough)
};
// What a launch looks like on this device: the workgroups that are resident at once and how many tiles of 64 nodes one of them takes
// per round (its waves).  The engine sizes small launches with it.
struct LfGeometry {
    u32 blocks;           // resident workgroups of the one-sample kernel (its waves walk the level with a block-wide tile counter)
    u32 waves_per_block;
};
int lf_step_geometry(bool wide_pos, int device, LfGeometry* g);
// One sample.  tiles_bound: an upper bound of the level's tiles of 64 nodes (the level itself, or four times the level before it
// for a launch queued ahead): the grid is no wider than that needs.
void lf_step_launch(const LfConfig& c, const LfGeometry& g, u64 tiles_bound, hipStream_t st, const DevIndex& ix, const u32* rp, const void* rec,
                    void* out, u64* splane, u32* cnt, void* valf, u8* pl, const ExpandArgs& a, u64* counters, unsigned long long* childmax);
// nb samples of one process (handles derived in the kernel from the level's slots and the samples' parent planes)
void lf_step_launch_batch(const LfConfig& c, const LfGeometry& g, u32 grid_factor, int nb, hipStream_t st, const ExpandBatch& b, const ExpandArgs& a,
                          u64* counters, unsigned long long* childmax);

}  // namespace dsm
