// common.h -- internal declarations shared by the translation units of libdsmhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

namespace dsm {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef uint16_t u16;

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define DSM_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return ::dsm::fail(DSM_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

// ----------------------------------------------------------------------------------------------
// Device layout of the BWT: one 64-byte block per 128 symbols.
//   cnt[c]  (c = A,C,G,T)  occurrences of c before the block, relative to the block's superblock
//   pl[k][w] bit-plane k of the 3-bit symbol codes, word w covers symbols 64w..64w+63 (LSB first)
// codes: A=0 C=1 G=2 T=3, 4..7 = the index's other symbols in increasing byte order ('\0','-','N',..)
// One 64-byte fetch answers rank for all four bases at a position (the reference needs 2-3 dependent
// BitRank::rank per base, three cache lines each: BitRank.cpp:191-195, HuffWT.h:66-83).
// ----------------------------------------------------------------------------------------------
constexpr int BLK_SHIFT = 7;
constexpr u32 BLK_SYMS = 1u << BLK_SHIFT;
constexpr int SB_SHIFT = 31;  // superblock = 2^31 symbols; cnt[] fits 32 bits
struct __attribute__((aligned(64))) Blk {
    u32 cnt[4];
    u64 pl[3][2];
};
static_assert(sizeof(Blk) == 64, "block must be one 64-byte line");
constexpr int RARE_SAMPLE_SHIFT = 6;  // absolute counts of codes 4..7 every 64 blocks

// The file's wavelet tree in the reference 3-array layout, flattened (DSM_OPEN_KEEP_WT / load path).
struct WtNodeDev {
    int leaf;
    int ch;
    int left, right;
    u64 data_off, rs_off, rb_off;  // offsets into the raw blob: u64 data[], u64 Rs[], u8 Rb[]
    u64 nbits;
};

struct DevIndex {
    const Blk* blk;     // nblk = (n >> 7) + 1 blocks
    const u64* sbase;   // [nsb][4]: C[c] + occurrences of base c before superblock
    const u64* rare;    // [(nblk >> 6) + 1][4]: occurrences of codes 4..7 before block 64k
    u64 n;
    u64 nblk;
    // raw wavelet tree (may be null)
    const WtNodeDev* wt_nodes;
    const u8* wt_blob;
    int wt_nnodes;
};

struct IndexMeta {
    u64 n;
    u64 C[256];
    dsm_code codes[256];
    int byte2code[256];  // -1 = symbol absent
    u8 code2byte[8];
    int ncodes;          // number of 3-bit codes in use (4 + others)
    u32 lfcost[4];       // BitRank::rank calls per LF on A,C,G,T = bits(c), 0 when absent
};

}  // namespace dsm

struct dsm_index {
    dsm::IndexMeta meta;
    dsm::DevIndex dev;
    int device;
    std::string name;
    dsm::u64 device_bytes;
    void* d_blk;
    void* d_sbase;
    void* d_rare;
    void* d_wt_nodes;
    void* d_wt_blob;
    // residency (dsm_index_offload / dsm_index_reload): the blocks' pinned host copy, made on the first offload
    void* h_blk = nullptr;
    dsm::u64 blk_bytes = 0;
};

namespace dsm {

#ifdef __HIPCC__
// ---- device-side rank primitives on the plane layout -------------------------------------------
struct Blk16 {  // a block held in registers
    u32 cnt[4];
    u64 p0a, p0b, p1a, p1b, p2a, p2b;
};

__device__ __forceinline__ void load_blk(const Blk* __restrict__ b, u64 bi, Blk16& r) {
    const uint4* q = reinterpret_cast<const uint4*>(b + bi);
    uint4 h = q[0], a = q[1], c = q[2], d = q[3];
    r.cnt[0] = h.x; r.cnt[1] = h.y; r.cnt[2] = h.z; r.cnt[3] = h.w;
    r.p0a = ((u64)a.y << 32) | a.x; r.p0b = ((u64)a.w << 32) | a.z;
    r.p1a = ((u64)c.y << 32) | c.x; r.p1b = ((u64)c.w << 32) | c.z;
    r.p2a = ((u64)d.y << 32) | d.x; r.p2b = ((u64)d.w << 32) | d.z;
}

// occurrences of A,C,G,T among the first `off` (0..127) symbols of the block
__device__ __forceinline__ void blk_counts(const Blk16& r, u32 off, u32 out[4]) {
    u64 ma = off >= 64 ? ~0ull : ((1ull << off) - 1);
    u64 mb = off > 64 ? ((1ull << (off - 64)) - 1) : 0ull;
    u64 ba = ma & ~r.p2a, bb = mb & ~r.p2b;  // base symbols only
    out[0] = __popcll(ba & ~r.p1a & ~r.p0a) + __popcll(bb & ~r.p1b & ~r.p0b);
    out[1] = __popcll(ba & ~r.p1a & r.p0a) + __popcll(bb & ~r.p1b & r.p0b);
    out[2] = __popcll(ba & r.p1a & ~r.p0a) + __popcll(bb & r.p1b & ~r.p0b);
    out[3] = __popcll(ba & r.p1a & r.p0a) + __popcll(bb & r.p1b & r.p0b);
}

__device__ __forceinline__ u32 blk_code_at(const Blk16& r, u32 off) {
    u64 p0 = off < 64 ? r.p0a : r.p0b, p1 = off < 64 ? r.p1a : r.p1b, p2 = off < 64 ? r.p2a : r.p2b;
    u32 s = off & 63;
    return (u32)((p0 >> s) & 1) | ((u32)((p1 >> s) & 1) << 1) | ((u32)((p2 >> s) & 1) << 2);
}
#endif

// host-side helpers implemented in index.hip
int index_check_device(const dsm_index* idx);

}  // namespace dsm
