// fmiwrite.hip -- index construction behind the C ABI: FASTA -> texts -> BWT (bwt.hip) -> Huffman-shaped
// wavelet tree with the reference's BitRank directories -> .fmi v17, byte-compatible with the reference builder's file.
//   input rules        builder.cpp:60-104 (normalize), :183-201 (transform), :203-262 (records)
//   Huffman shape      HuffWT.cpp:133-171 (std::priority_queue with greater<node>: the tie order is the container's)
//   tree / bit vectors HuffWT.cpp:5-55 (partition by code bit, level = bit index), HuffWT.cpp:73-86 (pre-order save)
//   rank directories   BitRank.cpp:154-187 (Rs per 256 bits, Rb per word relative to the superblock), :134-151 (save)
//   container          FMIndex.cpp:155-217 (version 17)
// The bit vector of a tree node is produced straight from the BWT: the node with code prefix p at level l holds, in BWT order, the
// symbols whose code starts with p, its bit is code bit l.  One pass over the BWT per internal node (six for DNA reads).
#include <algorithm>
#include <cstring>
#include <fstream>
#include <queue>
#include <string>
#include <vector>

#include "common.h"
#include "scan.h"

namespace dsm {

constexpr u32 FW_CHUNK = 4096;  // BWT symbols per block of the selection passes

// histogram and first occurrence of every byte value
__global__ __launch_bounds__(256) void fw_hist_kernel(const u8* __restrict__ bwt, u64 n, unsigned long long* __restrict__ counts,
                                                      unsigned long long* __restrict__ first) {
    __shared__ u32 h[256];
    __shared__ unsigned long long f[256];
    h[threadIdx.x] = 0;
    f[threadIdx.x] = ~0ull;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * FW_CHUNK * 16;
    for (u32 k = threadIdx.x; k < FW_CHUNK * 16; k += 256) {
        const u64 i = base + k;
        if (i < n) {
            const u32 c = bwt[i];
            atomicAdd(&h[c], 1u);
            atomicMin(&f[c], (unsigned long long)i);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) {
        atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
        atomicMin(&first[threadIdx.x], f[threadIdx.x]);
    }
}

// member[c] bit 0: symbol c belongs to the node, bit 1: its bit in the node.  Pass 1: members per chunk.
__global__ __launch_bounds__(256) void fw_count_kernel(const u8* __restrict__ bwt, u64 n, const u8* __restrict__ member, u64* __restrict__ chunk_cnt) {
    __shared__ u8 mem[256];
    __shared__ u32 tot;
    mem[threadIdx.x] = member[threadIdx.x];
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * FW_CHUNK;
    u32 c = 0;
    for (u32 k = threadIdx.x; k < FW_CHUNK; k += 256) {
        const u64 i = base + k;
        if (i < n) c += mem[bwt[i]] & 1u;
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&tot, c);
    __syncthreads();
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = tot;
}

// Pass 2: the members of a chunk, in order, append their bits at the chunk's offset of the node's bit vector.  A wave takes 64
// consecutive symbols at a time: the members among them are ranked by a ballot, their bits packed by a second one.
__global__ __launch_bounds__(64) void fw_pack_kernel(const u8* __restrict__ bwt, u64 n, const u8* __restrict__ member, const u64* __restrict__ chunk_off,
                                                     unsigned long long* __restrict__ words) {
    __shared__ u8 mem[256];
    for (u32 k = threadIdx.x; k < 256; k += 64) mem[k] = member[k];
    __syncthreads();
    const int lane = threadIdx.x;
    const u64 base = (u64)blockIdx.x * FW_CHUNK;
    u64 pos = chunk_off[blockIdx.x];  // bit position of the chunk's first member (wave-uniform)
    for (u32 k = 0; k < FW_CHUNK; k += 64) {
        const u64 i = base + k + lane;
        const u32 m = i < n ? mem[bwt[i]] : 0u;
        const u64 isin = __ballot(m & 1u);
        if (!isin) continue;
        // lane r fetches the bit of the r-th member (the position of the r-th set bit of isin: halving steps on population counts),
        // and a ballot packs the fetched bits
        u32 src = 0;
        {
            u32 want = (u32)lane;
            const u32 lo = (u32)isin, hi = (u32)(isin >> 32);
            const u32 pl = (u32)__popc(lo);
            u32 word = lo;
            if (want >= pl) { want -= pl; word = hi; src = 32; }
#pragma unroll
            for (int b = 16; b >= 1; b >>= 1) {
                const u32 low = word & ((1u << b) - 1u);
                const u32 c = (u32)__popc(low);
                if (want >= c) { want -= c; word >>= b; src += (u32)b; } else word = low;
            }
        }
        const u32 cnt = (u32)__popcll(isin);
        const u32 mbit = (u32)__shfl((int)((m >> 1) & 1u), (int)(src & 63u), 64);
        const u64 packed = __ballot(lane < (int)cnt && mbit);
        if (lane == 0) {
            const u64 w = pos >> 6;
            const u32 sh = (u32)(pos & 63);
            atomicOr(&words[w], (unsigned long long)(packed << sh));
            if (sh && sh + cnt > 64) atomicOr(&words[w + 1], (unsigned long long)(packed >> (64 - sh)));
        }
        pos += cnt;
    }
}

__global__ void fw_popc_kernel(const unsigned long long* __restrict__ words, u64 nwords, u64* __restrict__ pc) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nwords) pc[i] = (u64)__popcll(words[i]);
}
// cum[i] = ones in words [0, i), cum has nwords + 1 entries.  Rs[j] = cum[min(4j, nwords)], Rb[k] = ones in words [4 (k/4), k)
__global__ void fw_dirs_kernel(const u64* __restrict__ cum, u64 nwords, u64 nRs, u64 nRb, u64* __restrict__ Rs, u8* __restrict__ Rb) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nRs) { const u64 j = 4 * i < nwords ? 4 * i : nwords; Rs[i] = cum[j]; }
    if (i < nRb) {
        const u64 lo = (i / 4) * 4 < nwords ? (i / 4) * 4 : nwords;
        const u64 hi = i < nwords ? i : nwords;
        Rb[i] = (u8)(cum[hi] - cum[lo]);
    }
}

struct HuffNode {
    u64 weight;
    int id;  // index into the node table
};
struct HuffGreater {
    bool operator()(const HuffNode& a, const HuffNode& b) const { return a.weight > b.weight; }  // node::operator>, HuffWT.cpp:123-125
};
struct HuffTree {
    struct N { int c0 = -1, c1 = -1; int value = 0; };
    std::vector<N> nodes;
    int root = -1;
    u32 bits[256], code[256];
};

static void huff_table(const HuffTree& t, int nd, u32 code, u32 bits, u32* obits, u32* ocode) {  // node::maketable, HuffWT.cpp:156-171
    const HuffTree::N& x = t.nodes[(size_t)nd];
    if (x.c0 >= 0) {
        huff_table(t, x.c0, code, bits + 1, obits, ocode);
        huff_table(t, x.c1, code | (1u << bits), bits + 1, obits, ocode);
    } else {
        ocode[x.value] = code;
        obits[x.value] = bits;
    }
}

static void huffman(const u64 counts[256], HuffTree& t) {  // node::makecodetable, HuffWT.cpp:133-154
    std::priority_queue<HuffNode, std::vector<HuffNode>, HuffGreater> q;
    for (int i = 0; i < 256; ++i) {
        t.bits[i] = 0; t.code[i] = 0;
        if (counts[i]) {
            HuffTree::N leaf;
            leaf.value = i;
            t.nodes.push_back(leaf);
            q.push(HuffNode{counts[i], (int)t.nodes.size() - 1});
        }
    }
    if (q.empty()) return;
    while (q.size() > 1) {
        const HuffNode a = q.top(); q.pop();
        const HuffNode b = q.top(); q.pop();
        HuffTree::N in;
        in.c0 = a.id; in.c1 = b.id;
        t.nodes.push_back(in);
        q.push(HuffNode{a.weight + b.weight, (int)t.nodes.size() - 1});
    }
    t.root = q.top().id;
    huff_table(t, t.root, 0u, 0u, t.bits, t.code);
}

struct FmiWriter {
    const u8* d_bwt;
    u64 n;
    hipStream_t st;
    FILE* f;
    u64 counts[256], first[256];
    u32 bits[256], code[256];
    u8* d_member = nullptr;
    u64 *d_chunk = nullptr, *d_tmp = nullptr;
    unsigned long long* d_words = nullptr;
    u64* d_cum = nullptr;
    u64* d_Rs = nullptr;
    u8* d_Rb = nullptr;
    std::vector<u8> host;
    u64 nchunk = 0;
    std::string err;

    bool put(const void* p, size_t k) { return fwrite(p, 1, k, f) == k; }

    // pre-order: {leaf, ch} then, for an internal node, its BitRank, the zero side, the one side (HuffWT.cpp:73-86)
    int node(u32 prefix, u32 level) {
        // the symbols of this node: codes that continue `prefix` (level low bits)
        u8 member[256];
        u64 size = 0, ones = 0, firstpos = ~0ull;
        int ch = 0;
        for (int c = 0; c < 256; ++c) {
            member[c] = 0;
            if (!counts[c] || bits[c] < level || (code[c] & ((1u << level) - 1u)) != prefix) continue;
            const u32 b = bits[c] > level ? (code[c] >> level) & 1u : 0u;
            member[c] = (u8)(1u | (b << 1));
            size += counts[c];
            ones += b ? counts[c] : 0;
            if (first[c] < firstpos) { firstpos = first[c]; ch = c; }  // ch = the node's first symbol (HuffWT.cpp:9)
        }
        const bool leaf = ones == 0 || ones == size;
        const u8 head[2] = {(u8)(leaf ? 1 : 0), (u8)ch};
        if (!put(head, 2)) return fail(DSM_E_IO, "writing the index file failed");
        if (leaf) return 0;
        // ---- the bit vector ----
        const u64 integers = (size + 1 + 63) / 64;  // BitRank.cpp:16-17: n + 1 bits are allocated
        const u64 nRs = size / 256 + 1, nRb = size / 64 + 1;
        DSM_HIP(hipMemcpyAsync(d_member, member, 256, hipMemcpyHostToDevice, st));
        DSM_HIP(hipMemsetAsync(d_words, 0, (integers + 1) * 8, st));
        hipLaunchKernelGGL(fw_count_kernel, dim3((unsigned)nchunk), dim3(256), 0, st, d_bwt, n, d_member, d_chunk);
        exclusive_scan<u64, u64>(d_chunk, d_chunk, (size_t)nchunk, d_tmp, (u64*)nullptr, st);
        hipLaunchKernelGGL(fw_pack_kernel, dim3((unsigned)nchunk), dim3(64), 0, st, d_bwt, n, d_member, d_chunk, d_words);
        // ---- the rank directories ----
        hipLaunchKernelGGL(fw_popc_kernel, dim3((unsigned)((integers + 255) / 256)), dim3(256), 0, st, d_words, integers, d_cum);
        exclusive_scan<u64, u64>(d_cum, d_cum, (size_t)integers, d_tmp, d_cum + integers, st);
        const u64 nd = nRs > nRb ? nRs : nRb;
        hipLaunchKernelGGL(fw_dirs_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, d_cum, integers, nRs, nRb, d_Rs, d_Rb);
        DSM_HIP(hipGetLastError());
        // ---- BitRank::save (BitRank.cpp:134-151): n, integers, then data, Rs, Rb; b and s precede the data (W = 64, superFactor 4) ----
        const u32 b = 64, s = 256;
        if (!put(&size, 8) || !put(&integers, 8) || !put(&b, 4) || !put(&s, 4)) return fail(DSM_E_IO, "writing the index file failed");
        const size_t total = (size_t)(integers * 8 + nRs * 8 + nRb);
        host.resize(total);
        DSM_HIP(hipMemcpyAsync(host.data(), d_words, integers * 8, hipMemcpyDeviceToHost, st));
        DSM_HIP(hipMemcpyAsync(host.data() + integers * 8, d_Rs, nRs * 8, hipMemcpyDeviceToHost, st));
        DSM_HIP(hipMemcpyAsync(host.data() + integers * 8 + nRs * 8, d_Rb, nRb, hipMemcpyDeviceToHost, st));
        DSM_HIP(hipStreamSynchronize(st));
        if (!put(host.data(), total)) return fail(DSM_E_IO, "writing the index file failed");
        if (int rc = node(prefix, level + 1)) return rc;
        return node(prefix | (1u << level), level + 1);
    }
};

static int fmi_write_impl(const u8* d_bwt, u64 n, u32 ntexts, u64 maxlen, u32 samplerate, int device, const char* path) {
    if (!d_bwt || !path || n == 0) return fail(DSM_E_INVAL, "dsm_fmi_write: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_fmi_write: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_fmi_write: bad device ordinal");
    DSM_HIP(hipSetDevice(device));
    FmiWriter w;
    w.d_bwt = d_bwt; w.n = n; w.st = 0;
    unsigned long long *d_counts = nullptr, *d_first = nullptr;
    DSM_HIP(hipMalloc((void**)&d_counts, 256 * 8));
    DSM_HIP(hipMalloc((void**)&d_first, 256 * 8));
    DSM_HIP(hipMemset(d_counts, 0, 256 * 8));
    DSM_HIP(hipMemset(d_first, 0xFF, 256 * 8));
    hipLaunchKernelGGL(fw_hist_kernel, dim3((unsigned)((n + FW_CHUNK * 16 - 1) / (FW_CHUNK * 16))), dim3(256), 0, 0, d_bwt, n, d_counts, d_first);
    DSM_HIP(hipMemcpy(w.counts, d_counts, 256 * 8, hipMemcpyDeviceToHost));
    DSM_HIP(hipMemcpy(w.first, d_first, 256 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(d_counts);
    (void)hipFree(d_first);
    HuffTree ht;
    huffman(w.counts, ht);
    memcpy(w.bits, ht.bits, sizeof w.bits);
    memcpy(w.code, ht.code, sizeof w.code);
    w.nchunk = (n + FW_CHUNK - 1) / FW_CHUNK;
    const u64 maxwords = (n + 1 + 63) / 64 + 2;
    int rc = 0;
    auto dal = [&](void** p, size_t bytes) { if (!rc && hipMalloc(p, bytes) != hipSuccess) rc = fail(DSM_E_NOMEM, "dsm_fmi_write: out of device memory"); };
    dal((void**)&w.d_member, 256);
    dal((void**)&w.d_chunk, (w.nchunk + 8) * 8);
    dal((void**)&w.d_tmp, (scan_tmp_elems(std::max<u64>(w.nchunk, maxwords)) + 8) * 8);
    dal((void**)&w.d_words, maxwords * 8);
    dal((void**)&w.d_cum, (maxwords + 1) * 8);
    dal((void**)&w.d_Rs, (n / 256 + 2) * 8);
    dal((void**)&w.d_Rb, n / 64 + 2);
    if (!rc) {
        w.f = fopen(path, "wb");
        if (!w.f) rc = fail(DSM_E_IO, std::string("cannot create ") + path);
    }
    if (!rc) {
        // FMIndex::save, FMIndex.cpp:155-217: version, n, samplerate, C[256], bwtEndPos, the code table, the tree, then the
        // collection's counters (no samples, no names, no text storage: enumeration reads none of them)
        const u8 version = 17;
        u64 C[256];
        u64 acc = 0;
        for (int c = 0; c < 256; ++c) { C[c] = acc; acc += w.counts[c]; }  // FMIndex::makewavelet, FMIndex.cpp:395-410
        const u64 zero = 0;
        bool ok = w.put(&version, 1) && w.put(&n, 8) && w.put(&samplerate, 4) && w.put(C, sizeof C) && w.put(&zero, 8);
        for (int c = 0; c < 256 && ok; ++c) ok = w.put(&w.counts[c], 8) && w.put(&w.bits[c], 4) && w.put(&w.code[c], 4);
        if (!ok) rc = fail(DSM_E_IO, "writing the index file failed");
        if (!rc) rc = w.node(0u, 0u);
        if (!rc) {
            const u8 z1 = 0;
            const u32 z4 = 0;
            ok = w.put(&ntexts, 4) && w.put(&maxlen, 8) && w.put(&z1, 1) && w.put(&z1, 1) && w.put(&z1, 1) && w.put(&z4, 4);
            if (!ok) rc = fail(DSM_E_IO, "writing the index file failed");
        }
        if (fclose(w.f) != 0 && !rc) rc = fail(DSM_E_IO, "closing the index file failed");
        if (rc) remove(path);
    }
    for (void* p : {(void*)w.d_member, (void*)w.d_chunk, (void*)w.d_tmp, (void*)w.d_words, (void*)w.d_cum, (void*)w.d_Rs, (void*)w.d_Rb})
        if (p) (void)hipFree(p);
    return rc;
}

// builder.cpp:60-104: upper case; A,C,G,T,N and the colour-space symbols 0-3 and '.' pass, everything else becomes N
static inline u8 norm_sym(u8 c) {
    switch (c) {
        case 'a': return 'A'; case 'c': return 'C'; case 'g': return 'G'; case 't': return 'T'; case 'n': return 'N';
        case 'A': case 'C': case 'G': case 'T': case 'N': case '0': case '1': case '2': case '3': case '.': return c;
        default: return 'N';
    }
}
static inline u8 comp_sym(u8 c) {  // complement(), builder.cpp:36-57: A<->T, C<->G, the rest unchanged
    switch (c) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; default: return c; }
}

}  // namespace dsm

using namespace dsm;

extern "C" {

int dsm_fmi_write(const uint8_t* d_bwt, uint64_t n, uint32_t number_of_texts, uint64_t max_text_length, uint32_t samplerate, int device,
                  const char* path) {
    return fmi_write_impl(d_bwt, n, number_of_texts, max_text_length, samplerate, device, path);
}

int dsm_build_fasta(const char* fasta_path, const char* out_path, uint32_t samplerate, int device, dsm_build_info* info) {
    if (!fasta_path || !out_path) return fail(DSM_E_INVAL, "dsm_build_fasta: null argument");
    std::ifstream in(fasta_path);
    if (!in.good()) return fail(DSM_E_IO, std::string("unable to read input file ") + fasta_path);
    // builder.cpp:203-262: a '>' row starts a record, its sequence rows are joined; a row counts only when its newline was read
    // (getline(...).good()), empty records are skipped.  Text = reverse(read + '-' + revcomp(read)) (builder.cpp:183-201) + '\0'.
    std::vector<u8> text;
    std::string row, cur;
    u64 ntexts = 0, maxlen = 0;
    auto flush = [&]() {
        if (cur.empty()) return;
        const size_t L = cur.size();
        const size_t base = text.size();
        text.resize(base + 2 * L + 2);
        u8* t = text.data() + base;
        // forward = norm(read) '-' revcomp(norm(read)); stored reversed: comp(read) forward, '-', read reversed
        for (size_t k = 0; k < L; ++k) {
            const u8 c = norm_sym((u8)cur[k]);
            t[k] = comp_sym(c);
            t[2 * L - k] = c;
        }
        t[L] = '-';
        t[2 * L + 1] = 0;
        ++ntexts;
        if (2 * L + 2 > maxlen) maxlen = 2 * L + 2;
        cur.clear();
    };
    while (std::getline(in, row).good()) {
        if (!row.empty() && row[0] == '>') flush();
        else cur.append(row);
    }
    flush();
    if (text.empty()) return fail(DSM_E_INVAL, "no sequences in the input");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_build_fasta: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_build_fasta: bad device ordinal");
    DSM_HIP(hipSetDevice(device));
    const u64 n = text.size();
    u8 *d_text = nullptr, *d_bwt = nullptr;
    DSM_HIP(hipMalloc((void**)&d_text, n));
    if (hipMalloc((void**)&d_bwt, n) != hipSuccess) { (void)hipFree(d_text); return fail(DSM_E_NOMEM, "dsm_build_fasta: out of device memory"); }
    int rc = 0;
    if (hipMemcpy(d_text, text.data(), n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(DSM_E_HIP, "dsm_build_fasta: upload failed");
    if (!rc) rc = dsm_bwt_build(d_text, n, d_bwt, device, nullptr);
    (void)hipFree(d_text);
    if (!rc) rc = fmi_write_impl(d_bwt, n, (u32)ntexts, maxlen, samplerate, device, out_path);
    (void)hipFree(d_bwt);
    if (!rc && info) { info->n = n; info->number_of_texts = ntexts; info->max_text_length = maxlen; }
    return rc;
}

}  // extern "C"
