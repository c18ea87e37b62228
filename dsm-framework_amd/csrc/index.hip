// index.hip -- .fmi v14..v17 reader and the HBM-resident index (gfx950).
//
// Load path: the file's Huffman-shaped wavelet tree is uploaded in the reference's own 3-array
// layout (data / Rs / Rb per node: BitRank.cpp:111-132, HuffWT.cpp:57-71), decoded on the GPU with
// HuffWT::access semantics (HuffWT.h:126-140) and transcoded into the interleaved bit-plane blocks
// of common.h.  The raw tree is dropped afterwards unless DSM_OPEN_KEEP_WT asks to keep it for the
// node-by-node LF kernel (HuffWT::rank semantics, HuffWT.h:66-83).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>

#include "common.h"
#include "scan.h"

namespace dsm {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(int code, const std::string& m) { g_err = m; return code; }

// ---------------------------------------------------------------------------------------------
// device: the reference's bitvector rank on its own layout
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 wt_bitrank(const u8* __restrict__ blob, const WtNodeDev& nd, u64 i) {
    // BitRank::rank, BitRank.cpp:191-195 (i = -1 wraps to 0)
    ++i;
    const u64* data = reinterpret_cast<const u64*>(blob + nd.data_off);
    const u64* Rs = reinterpret_cast<const u64*>(blob + nd.rs_off);
    const u8* Rb = blob + nd.rb_off;
    return Rs[i >> 8] + Rb[i >> 6] + (u64)__popcll(data[i >> 6] & ((1ull << (i & 63)) - 1));
}
__device__ __forceinline__ bool wt_bit(const u8* __restrict__ blob, const WtNodeDev& nd, u64 i) {
    const u64* data = reinterpret_cast<const u64*>(blob + nd.data_off);
    return (data[i >> 6] >> (i & 63)) & 1;  // BitRank::IsBitSet, BitRank.cpp:338-340
}

// HuffWT::access, HuffWT.h:126-140 -> 3-bit code of BWT[i]
__global__ void wt_decode_kernel(DevIndex ix, const int* __restrict__ byte2code, u8* __restrict__ codes, u64 base) {
    u64 i = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ix.n) return;
    int t = 0;
    u64 p = i;
    while (!ix.wt_nodes[t].leaf) {
        const WtNodeDev nd = ix.wt_nodes[t];
        if (wt_bit(ix.wt_blob, nd, p)) { p = wt_bitrank(ix.wt_blob, nd, p) - 1; t = nd.right; }
        else { p = p - wt_bitrank(ix.wt_blob, nd, p); t = nd.left; }
    }
    codes[i] = (u8)byte2code[ix.wt_nodes[t].ch];
}

// one wave packs 64 symbols into three plane words with ballots
__global__ __launch_bounds__(256) void pack_planes_kernel(const u8* __restrict__ codes, u64 n, u64 nblk, Blk* __restrict__ blk, u64 base) {
    u64 i = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk * BLK_SYMS) return;  // grid is sized to whole blocks, so full waves reach the ballots
    u32 c = i < n ? codes[i] : 4u;     // padding past n is a non-base code: never counted
    u64 b0 = __ballot(c & 1), b1 = __ballot(c & 2), b2 = __ballot(c & 4);
    if ((threadIdx.x & 63) == 0) {
        Blk* b = blk + (i >> BLK_SHIFT);
        int w = (int)((i >> 6) & 1);
        b->pl[0][w] = b0; b->pl[1][w] = b1; b->pl[2][w] = b2;
    }
}

// per-block histogram of the 8 codes -> cnt8[code][blk]
__global__ void block_hist_kernel(const Blk* __restrict__ blk, u64 nblk, u32* __restrict__ cnt8) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const Blk& k = blk[b];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        u32 s = 0;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            u64 m = ~0ull;
            m &= (c & 1) ? k.pl[0][w] : ~k.pl[0][w];
            m &= (c & 2) ? k.pl[1][w] : ~k.pl[1][w];
            m &= (c & 4) ? k.pl[2][w] : ~k.pl[2][w];
            s += __popcll(m);
        }
        cnt8[(u64)c * nblk + b] = s;
    }
}

// pre8[code][blk] = occurrences before block (absolute).  Fill block headers (relative to superblock),
// superblock bases and the sampled rare-code table.
__global__ void finish_blocks_kernel(Blk* __restrict__ blk, u64 nblk, const u64* __restrict__ pre8, u64* __restrict__ sbase_cnt,
                                     u64* __restrict__ rare) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const u64 per_sb = 1ull << (SB_SHIFT - BLK_SHIFT);
    u64 sb = b / per_sb, first = sb * per_sb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u64 base = pre8[(u64)c * nblk + first];
        blk[b].cnt[c] = (u32)(pre8[(u64)c * nblk + b] - base);
        if (b == first) sbase_cnt[sb * 4 + c] = base;
    }
    if ((b & ((1u << RARE_SAMPLE_SHIFT) - 1)) == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) rare[(b >> RARE_SAMPLE_SHIFT) * 4 + c] = pre8[(u64)(4 + c) * nblk + b];
    }
}

// ---------------------------------------------------------------------------------------------
// LF kernels
// ---------------------------------------------------------------------------------------------
struct LfTables {
    u64 C[256];
};

// occurrences of code `code` (0..7) in BWT[0, x)
__device__ __forceinline__ u64 planes_count(const DevIndex& ix, u32 code, u64 x) {
    u64 bi = x >> BLK_SHIFT;
    u32 off = (u32)(x & (BLK_SYMS - 1));
    Blk16 r;
    load_blk(ix.blk, bi, r);
    if (code < 4) {
        u32 c4[4];
        blk_counts(r, off, c4);
        return (ix.sbase[(x >> SB_SHIFT) * 4 + code]) + r.cnt[code] + c4[code];  // sbase includes C[]: caller subtracts
    }
    // rare code: sampled absolute count + scan of the blocks since the sample
    u64 s = bi >> RARE_SAMPLE_SHIFT;
    u64 cnt = ix.rare[s * 4 + (code - 4)];
    for (u64 b = s << RARE_SAMPLE_SHIFT; b <= bi; ++b) {
        const Blk& k = ix.blk[b];
        u32 lim = b == bi ? off : BLK_SYMS;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            u32 lo = w * 64;
            u64 mask = lim >= lo + 64 ? ~0ull : (lim > lo ? ((1ull << (lim - lo)) - 1) : 0ull);
            u64 m = mask & k.pl[2][w];
            m &= (code & 1) ? k.pl[0][w] : ~k.pl[0][w];
            m &= (code & 2) ? k.pl[1][w] : ~k.pl[1][w];
            cnt += __popcll(m);
        }
    }
    return cnt;
}

__global__ void lf_planes_kernel(DevIndex ix, const LfTables* __restrict__ tb, const int* __restrict__ byte2code,
                                 const u8* __restrict__ c, const u64* __restrict__ pos, u64* __restrict__ out, size_t k) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    u32 ch = c[j];
    int code = byte2code[ch];
    u64 Cc = tb->C[ch];
    if (code < 0) { out[j] = Cc; return; }  // FMIndex.h:86-87: absent symbol
    u64 x = pos[j] + 1;                      // exclusive end; -1 wraps to 0
    if (x > ix.n) x = ix.n;
    u64 v = planes_count(ix, (u32)code, x);
    out[j] = code < 4 ? v : Cc + v;          // sbase already holds C[c] for the bases
}

// HuffWT::rank walked node by node on the reference layout (HuffWT.h:66-83)
__global__ void lf_wt_kernel(DevIndex ix, const LfTables* __restrict__ tb, const dsm_code* __restrict__ codes,
                             const u8* __restrict__ c, const u64* __restrict__ pos, u64* __restrict__ out, size_t k) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    u32 ch = c[j];
    u64 Cc = tb->C[ch];
    if (codes[ch].count == 0) { out[j] = Cc; return; }
    u64 i = pos[j];
    int t = 0;
    u32 level = 0, code = codes[ch].code;
    while (!ix.wt_nodes[t].leaf) {
        const WtNodeDev nd = ix.wt_nodes[t];
        if ((code & (1u << level)) == 0) { i = i - wt_bitrank(ix.wt_blob, nd, i); t = nd.left; }
        else { i = wt_bitrank(ix.wt_blob, nd, i) - 1; t = nd.right; }
        ++level;
    }
    out[j] = Cc + i + 1;
}

__global__ void getl_kernel(DevIndex ix, const u8* __restrict__ code2byte, const u64* __restrict__ pos, u8* __restrict__ out, size_t k) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    u64 i = pos[j];
    if (i >= ix.n) { out[j] = 0; return; }
    Blk16 r;
    load_blk(ix.blk, i >> BLK_SHIFT, r);
    out[j] = code2byte[blk_code_at(r, (u32)(i & (BLK_SYMS - 1)))];
}

// ---------------------------------------------------------------------------------------------
// host: .fmi parser
// ---------------------------------------------------------------------------------------------
struct HostNode {
    bool leaf;
    u8 ch;
    int left, right;
    u64 nbits, integers;
    size_t data_pos, rs_pos, rb_pos;  // byte offsets in the file
};

struct Parser {
    const u8* p;
    size_t n, pos = 0;
    bool ok = true;
    template <class T> T rd() {
        T v;
        if (pos + sizeof(T) > n) { ok = false; pos = n; memset(&v, 0, sizeof(T)); return v; }
        memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    size_t skip(size_t k) {
        size_t at = pos;
        if (k > n - pos) { ok = false; pos = n; return at; }
        pos += k;
        return at;
    }
};

struct MappedFile {
    const u8* p = nullptr;
    size_t n = 0;
    int open(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return fail(DSM_E_NOENT, std::string("file not found: ") + path);
        struct stat sb;
        if (fstat(fd, &sb) != 0) { ::close(fd); return fail(DSM_E_IO, "fstat failed"); }
        n = (size_t)sb.st_size;
        if (n == 0) { ::close(fd); return fail(DSM_E_IO, "truncated or corrupt .fmi"); }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) { n = 0; return fail(DSM_E_IO, "mmap failed"); }
        (void)madvise(m, n, MADV_SEQUENTIAL);
        p = (const u8*)m;
        return 0;
    }
    void close() {
        if (p) munmap((void*)p, n);
        p = nullptr;
        n = 0;
    }
    ~MappedFile() { close(); }
};

// Host ranges -> device, chunk by chunk through pinned staging buffers, on a few threads: a thread copies a chunk out of the
// mapping (which is what faults the pages in) while its previous chunk is on the bus.  The buffers are kept for the next index.
struct UploadPiece { u8* dst; const u8* src; size_t n; };
struct StagePool {
    static constexpr int THREADS = 4;
    static constexpr size_t CHUNK = 8u << 20;
    std::mutex mu;  // one upload at a time uses the pool
    void* buf[THREADS][2] = {{nullptr}};
    int ensure() {
        for (int t = 0; t < THREADS; ++t)
            for (int k = 0; k < 2; ++k)
                if (!buf[t][k] && hipHostMalloc(&buf[t][k], CHUNK, hipHostMallocPortable) != hipSuccess) return fail(DSM_E_NOMEM, "hipHostMalloc (index staging) failed");
        return 0;
    }
};
static StagePool g_stage;
static int staged_upload(const std::vector<UploadPiece>& pieces, int device) {
    struct Chunk { u8* dst; const u8* src; size_t n; };
    std::vector<Chunk> chunks;
    for (const UploadPiece& q : pieces)
        for (size_t o = 0; o < q.n; o += StagePool::CHUNK) chunks.push_back(Chunk{q.dst + o, q.src + o, q.n - o < StagePool::CHUNK ? q.n - o : StagePool::CHUNK});
    std::lock_guard<std::mutex> lk(g_stage.mu);
    if (int rc = g_stage.ensure()) return rc;
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { bad = 1; return; }
        hipStream_t st;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { bad = 1; return; }
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool used[2] = {false, false};
        if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) bad = 1;
        for (int k = 0; !bad; k ^= 1) {
            const size_t i = next.fetch_add(1);
            if (i >= chunks.size()) break;
            if (used[k] && hipEventSynchronize(ev[k]) != hipSuccess) { bad = 1; break; }  // the buffer's previous chunk has left
            memcpy(g_stage.buf[t][k], chunks[i].src, chunks[i].n);
            if (hipMemcpyAsync(chunks[i].dst, g_stage.buf[t][k], chunks[i].n, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { bad = 1; break; }
            used[k] = true;
        }
        if (hipStreamSynchronize(st) != hipSuccess) bad = 1;
        for (int k = 0; k < 2; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    const int nth = chunks.size() < (size_t)StagePool::THREADS ? (int)chunks.size() : StagePool::THREADS;
    for (int t = 1; t < nth; ++t) th.emplace_back(work, t);
    if (nth > 0) work(0);
    for (auto& x : th) x.join();
    if (bad) return fail(DSM_E_HIP, "uploading the index failed");
    return 0;
}

static int parse_node(Parser& r, std::vector<HostNode>& nodes, int depth) {
    int id = (int)nodes.size();
    nodes.emplace_back();
    HostNode h{};
    h.leaf = r.rd<u8>() != 0;
    h.ch = r.rd<u8>();
    h.left = h.right = -1;
    if (!r.ok || depth > 300) { r.ok = false; nodes[id] = h; return id; }
    if (!h.leaf) {
        h.nbits = r.rd<u64>();
        h.integers = r.rd<u64>();
        u32 b = r.rd<u32>(), s = r.rd<u32>();
        if (!r.ok || b != 64 || s != 256 || h.integers > r.n / 8 + 1 || h.integers != (h.nbits + 1 + 63) / 64) { r.ok = false; nodes[id] = h; return id; }
        h.data_pos = r.skip(8 * h.integers);
        h.rs_pos = r.skip(8 * (h.nbits / 256 + 1));
        h.rb_pos = r.skip(h.nbits / 64 + 1);
        nodes[id] = h;
        if (!r.ok) return id;
        int l = parse_node(r, nodes, depth + 1);
        nodes[id].left = l;
        if (!r.ok) return id;
        int rr = parse_node(r, nodes, depth + 1);
        nodes[id].right = rr;
    } else {
        nodes[id] = h;
    }
    return id;
}

static std::string libname(const std::string& full) {  // metaenumerate.cpp:79-88
    size_t f = full.find_last_of("/\\");
    std::string t = f == std::string::npos ? full : full.substr(f + 1);
    return t.substr(0, t.find_first_of('.'));
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
    T* release() { T* q = p; p = nullptr; return q; }
};

// The file's header, code table and wavelet-tree shape, parsed in place and checked the way FMIndex::load / metaenumerate do
// (FMIndex.cpp:305-372, metaenumerate.cpp:243-247).  Host work only: dsm_index_probe runs it without a device.
static int parse_header(const MappedFile& file, IndexMeta& m, std::vector<HostNode>& nodes) {
    Parser r{file.p, file.n};
    u8 ver = r.rd<u8>();
    if (!r.ok || (ver != 17 && ver != 16 && ver != 15 && ver != 14))
        return fail(DSM_E_FORMAT, "FMIndex: invalid save file version (expected 14..17)");
    m.n = r.rd<u64>();
    (void)r.rd<u32>();  // samplerate
    for (int i = 0; i < 256; ++i) m.C[i] = ver == 14 ? (u64)r.rd<u32>() : r.rd<u64>();
    (void)r.rd<u64>();  // bwtEndPos
    for (int i = 0; i < 256; ++i) {
        m.codes[i].count = ver < 16 ? (u64)r.rd<u32>() : r.rd<u64>();
        m.codes[i].bits = r.rd<u32>();
        m.codes[i].code = r.rd<u32>();
    }
    if (r.ok) parse_node(r, nodes, 0);
    (void)r.rd<u32>();  // numberOfTexts
    (void)r.rd<u64>();  // maxTextLength
    u8 nameFlag = r.rd<u8>(), tsFlag = r.rd<u8>();
    u8 color = r.rd<u8>();
    (void)r.rd<u32>();
    if (!r.ok) return fail(DSM_E_IO, "truncated or corrupt .fmi");
    if (nameFlag || tsFlag) return fail(DSM_E_UNSUPPORTED, ".fmi carries name/text storage (not written by builder)");
    if (color) return fail(DSM_E_UNSUPPORTED, "index cannot be color coded (metaenumerate.cpp:243-247)");
    if (m.n == 0) return fail(DSM_E_FORMAT, "empty index");
    {
        u64 tot = 0;
        for (int i = 0; i < 256; ++i) tot += m.codes[i].count;
        if (tot != m.n) return fail(DSM_E_FORMAT, "code table counts do not sum to n");
        if (!nodes.empty() && !nodes[0].leaf && nodes[0].nbits != m.n) return fail(DSM_E_FORMAT, "root bitvector length != n");
    }

    // A version-14 file keeps C[] in 32 bits: past 2^32 symbols the values wrapped.  The reference notices a C[] that is not
    // monotone and recounts the symbols through the wavelet tree (FMIndex.cpp:346-356 -> recomputeC, :219-237: C[i] = symbols smaller
    // than i).  The same numbers follow from the code table's counts, which were just checked to add up to n (counts that wrapped as
    // well fail that check above and the file is refused).  (The reference also saves the repaired index as <name>.reC; this library
    // never writes next to its inputs.)
    for (int i = 1; i < 256; ++i) {
        if (m.C[i] < m.C[i - 1]) {
            u64 run = 0;
            for (int j = 0; j < 256; ++j) { m.C[j] = run; run += m.codes[j].count; }
            break;
        }
    }

    // symbol -> 3-bit code
    for (int i = 0; i < 256; ++i) m.byte2code[i] = -1;
    const char* bases = "ACGT";
    for (int k = 0; k < 4; ++k) {
        m.code2byte[k] = (u8)bases[k];
        if (m.codes[(int)bases[k]].count) m.byte2code[(int)bases[k]] = k;
        m.lfcost[k] = m.codes[(int)bases[k]].count ? m.codes[(int)bases[k]].bits : 0;
    }
    m.ncodes = 4;
    for (int i = 0; i < 256; ++i) {
        if (!m.codes[i].count || i == 'A' || i == 'C' || i == 'G' || i == 'T') continue;
        if (m.ncodes >= 8)
            return fail(DSM_E_UNSUPPORTED, "alphabet has more than 4 symbols besides A,C,G,T; the builder emits at most \\0,'-','N'");
        m.byte2code[i] = m.ncodes;
        m.code2byte[m.ncodes] = (u8)i;
        ++m.ncodes;
    }
    for (int k = m.ncodes; k < 8; ++k) m.code2byte[k] = 0;
    return 0;
}

static int open_impl(const char* path, int device, unsigned flags, dsm_index** out) {
    if (!path || !out) return fail(DSM_E_INVAL, "dsm_index_open: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DSM_E_NODEV, "dsm_index_open: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_index_open: bad device ordinal");
    DSM_HIP(hipSetDevice(device));

    // The file is mapped, not read: the header is parsed in place and the bit vectors go from the page cache through pinned
    // staging buffers to the card on several threads (a vector + fread + pageable hipMemcpy moved every byte three times on one).
    MappedFile file;
    if (int rc = file.open(path)) return rc;
    std::unique_ptr<dsm_index> ix(new dsm_index());
    IndexMeta& m = ix->meta;
    std::vector<HostNode> nodes;
    if (int rc = parse_header(file, m, nodes)) return rc;

    // ---- upload the raw wavelet tree ----------------------------------------------------------
    std::vector<WtNodeDev> wn(nodes.size());
    size_t blob_bytes = 0;
    for (size_t k = 0; k < nodes.size(); ++k) {
        const HostNode& h = nodes[k];
        WtNodeDev& d = wn[k];
        d.leaf = h.leaf; d.ch = h.ch; d.left = h.left; d.right = h.right; d.nbits = h.nbits;
        if (!h.leaf) {
            d.data_off = blob_bytes; blob_bytes += 8 * h.integers;
            d.rs_off = blob_bytes; blob_bytes += 8 * (h.nbits / 256 + 1);
            d.rb_off = blob_bytes; blob_bytes += (h.nbits / 64 + 1 + 7) & ~(size_t)7;
        }
    }
    DevBuf<u8> d_blob;
    DevBuf<WtNodeDev> d_nodes;
    DSM_HIP(d_blob.alloc(blob_bytes));
    DSM_HIP(d_nodes.alloc(wn.size()));
    {
        std::vector<UploadPiece> pieces;
        for (size_t k = 0; k < nodes.size(); ++k) {
            const HostNode& h = nodes[k];
            if (h.leaf) continue;
            pieces.push_back(UploadPiece{d_blob.p + wn[k].data_off, file.p + h.data_pos, (size_t)8 * h.integers});
            pieces.push_back(UploadPiece{d_blob.p + wn[k].rs_off, file.p + h.rs_pos, (size_t)8 * (h.nbits / 256 + 1)});
            pieces.push_back(UploadPiece{d_blob.p + wn[k].rb_off, file.p + h.rb_pos, (size_t)(h.nbits / 64 + 1)});
        }
        if (int rc = staged_upload(pieces, device)) return rc;
    }
    DSM_HIP(hipMemcpy(d_nodes.p, wn.data(), wn.size() * sizeof(WtNodeDev), hipMemcpyHostToDevice));
    file.close();

    const u64 n = m.n;
    const u64 nblk = (n >> BLK_SHIFT) + 1;
    const u64 nsb = (n >> SB_SHIFT) + 1;
    const u64 nrare = (nblk >> RARE_SAMPLE_SHIFT) + 1;
    DevIndex& dv = ix->dev;
    dv.n = n; dv.nblk = nblk;
    dv.wt_nodes = d_nodes.p; dv.wt_blob = d_blob.p; dv.wt_nnodes = (int)wn.size();

    DevBuf<Blk> d_blk;
    DevBuf<u64> d_sbase, d_rare, d_pre8, d_tmp;
    DevBuf<u32> d_cnt8;
    DevBuf<u8> d_codes;
    DevBuf<int> d_b2c;
    DSM_HIP(d_blk.alloc(nblk));
    DSM_HIP(d_sbase.alloc(nsb * 4));
    DSM_HIP(d_rare.alloc(nrare * 4));
    DSM_HIP(d_codes.alloc(n));
    DSM_HIP(d_b2c.alloc(256));
    DSM_HIP(d_cnt8.alloc(8 * nblk));
    DSM_HIP(d_pre8.alloc(8 * nblk));
    DSM_HIP(d_tmp.alloc(scan_tmp_elems(nblk) + 8));
    DSM_HIP(hipMemcpy(d_b2c.p, m.byte2code, sizeof(int) * 256, hipMemcpyHostToDevice));
    DSM_HIP(hipMemset(d_blk.p, 0, nblk * sizeof(Blk)));

    hipStream_t st = 0;
    {
        const int T = 256;
        // a launch holds fewer than 2^32 threads: one thread per symbol goes out in slices of 2^31 (indexes beyond 2^32 symbols)
        const u64 SLICE = 1ull << 31;
        for (u64 base = 0; base < n; base += SLICE) {
            const u64 m_ = n - base < SLICE ? n - base : SLICE;
            hipLaunchKernelGGL(wt_decode_kernel, dim3((unsigned)((m_ + T - 1) / T)), dim3(T), 0, st, dv, d_b2c.p, d_codes.p, base);
        }
        u64 padded = nblk * BLK_SYMS;
        for (u64 base = 0; base < padded; base += SLICE) {
            const u64 m_ = padded - base < SLICE ? padded - base : SLICE;
            hipLaunchKernelGGL(pack_planes_kernel, dim3((unsigned)((m_ + T - 1) / T)), dim3(T), 0, st, d_codes.p, n, nblk, d_blk.p, base);
        }
        hipLaunchKernelGGL(block_hist_kernel, dim3((unsigned)((nblk + T - 1) / T)), dim3(T), 0, st, d_blk.p, nblk, d_cnt8.p);
        for (int c = 0; c < 8; ++c)
            exclusive_scan<u32, u64>(d_cnt8.p + (u64)c * nblk, d_pre8.p + (u64)c * nblk, nblk, d_tmp.p, (u64*)nullptr, st);
        hipLaunchKernelGGL(finish_blocks_kernel, dim3((unsigned)((nblk + T - 1) / T)), dim3(T), 0, st, d_blk.p, nblk, d_pre8.p, d_sbase.p, d_rare.p);
        DSM_HIP(hipGetLastError());
        DSM_HIP(hipStreamSynchronize(st));
    }
    // fold C[] into the superblock bases: LF(c, x-1) = sbase[x>>31][c] + cnt + popcount
    {
        std::vector<u64> sbh(nsb * 4);
        DSM_HIP(hipMemcpy(sbh.data(), d_sbase.p, sbh.size() * 8, hipMemcpyDeviceToHost));
        for (u64 s = 0; s < nsb; ++s)
            for (int c = 0; c < 4; ++c) sbh[s * 4 + c] += m.C[(int)"ACGT"[c]];
        DSM_HIP(hipMemcpy(d_sbase.p, sbh.data(), sbh.size() * 8, hipMemcpyHostToDevice));
    }
    ix->device = device;
    ix->name = libname(path);
    ix->device_bytes = nblk * sizeof(Blk) + nsb * 32 + nrare * 32;
    ix->blk_bytes = nblk * sizeof(Blk);
    dv.blk = d_blk.p; dv.sbase = d_sbase.p; dv.rare = d_rare.p;
    ix->d_blk = d_blk.release();
    ix->d_sbase = d_sbase.release();
    ix->d_rare = d_rare.release();
    if (flags & DSM_OPEN_KEEP_WT) {
        ix->d_wt_nodes = d_nodes.release();
        ix->d_wt_blob = d_blob.release();
        ix->device_bytes += blob_bytes + wn.size() * sizeof(WtNodeDev);
    } else {
        ix->d_wt_nodes = nullptr; ix->d_wt_blob = nullptr;
        dv.wt_nodes = nullptr; dv.wt_blob = nullptr;
    }
    *out = ix.release();
    return DSM_OK;
}

struct TmpDev {
    std::vector<void*> ptrs;
    ~TmpDev() { for (void* p : ptrs) hipFree(p); }
    template <class T> T* get(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T*)p;
    }
};

static int lf_dev(const dsm_index* idx, const u8* d_c, const u64* d_i, u64* d_out, size_t k, unsigned layout, hipStream_t st) {
    if (!idx || (k && (!d_c || !d_i || !d_out))) return fail(DSM_E_INVAL, "dsm_lf_batch: null argument");
    if (k == 0) return DSM_OK;
    if (!idx->dev.blk) return fail(DSM_E_INVAL, "the index is offloaded: dsm_index_reload first");
    if (k >= (1ull << 32) - 256) return fail(DSM_E_INVAL, "dsm_lf_batch: at most 2^32 - 257 queries per call");
    DSM_HIP(hipSetDevice(idx->device));
    TmpDev tmp;
    LfTables* d_tb = tmp.get<LfTables>(1);
    if (!d_tb) return fail(DSM_E_NOMEM, "hipMalloc failed");
    DSM_HIP(hipMemcpyAsync(d_tb, idx->meta.C, sizeof(LfTables), hipMemcpyHostToDevice, st));
    const int T = 256;
    dim3 grid((unsigned)((k + T - 1) / T));
    if (layout == DSM_LAYOUT_WT) {
        if (!idx->dev.wt_nodes) return fail(DSM_E_UNSUPPORTED, "index was opened without DSM_OPEN_KEEP_WT");
        dsm_code* d_codes = tmp.get<dsm_code>(256);
        if (!d_codes) return fail(DSM_E_NOMEM, "hipMalloc failed");
        DSM_HIP(hipMemcpyAsync(d_codes, idx->meta.codes, sizeof(dsm_code) * 256, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lf_wt_kernel, grid, dim3(T), 0, st, idx->dev, d_tb, d_codes, d_c, d_i, d_out, k);
    } else if (layout == DSM_LAYOUT_PLANES) {
        int* d_b2c = tmp.get<int>(256);
        if (!d_b2c) return fail(DSM_E_NOMEM, "hipMalloc failed");
        DSM_HIP(hipMemcpyAsync(d_b2c, idx->meta.byte2code, sizeof(int) * 256, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lf_planes_kernel, grid, dim3(T), 0, st, idx->dev, d_tb, d_b2c, d_c, d_i, d_out, k);
    } else {
        return fail(DSM_E_INVAL, "unknown layout");
    }
    DSM_HIP(hipGetLastError());
    DSM_HIP(hipStreamSynchronize(st));  // temporaries die here
    return DSM_OK;
}

}  // namespace dsm

using namespace dsm;

extern "C" {

const char* dsm_last_error(void) { return dsm::g_err.c_str(); }
int dsm_abi_version(void) { return DSM_ABI_VERSION; }

int dsm_index_open(const char* p, int device, dsm_index** out) { return open_impl(p, device, 0, out); }
int dsm_index_probe(const char* path, uint64_t* n_out) {
    if (!path) return fail(DSM_E_INVAL, "dsm_index_probe: null argument");
    MappedFile file;
    if (int rc = file.open(path)) return rc;
    IndexMeta m;
    std::vector<HostNode> nodes;
    if (int rc = parse_header(file, m, nodes)) return rc;
    for (const HostNode& h : nodes)  // every bit vector the header promises lies inside the file
        if (!h.leaf && (h.data_pos > file.n || 8 * h.integers > file.n - h.data_pos)) return fail(DSM_E_IO, "truncated or corrupt .fmi");
    if (n_out) *n_out = m.n;
    return 0;
}
int dsm_index_open_ex(const char* p, int device, unsigned flags, dsm_index** out) { return open_impl(p, device, flags, out); }

void dsm_index_close(dsm_index* ix) {
    if (!ix) return;
    hipSetDevice(ix->device);
    if (ix->d_blk) hipFree(ix->d_blk);
    hipFree(ix->d_sbase); hipFree(ix->d_rare);
    if (ix->h_blk) (void)hipHostFree(ix->h_blk);
    if (ix->d_wt_nodes) hipFree(ix->d_wt_nodes);
    if (ix->d_wt_blob) hipFree(ix->d_wt_blob);
    delete ix;
}
uint64_t dsm_index_length(const dsm_index* ix) { return ix ? ix->meta.n : 0; }
int dsm_index_meta(const dsm_index* ix, uint64_t C[256], dsm_code codes[256]) {
    if (!ix) return fail(DSM_E_INVAL, "null index");
    if (C) memcpy(C, ix->meta.C, sizeof(u64) * 256);
    if (codes) memcpy(codes, ix->meta.codes, sizeof(dsm_code) * 256);
    return DSM_OK;
}
const char* dsm_index_name(const dsm_index* ix) { return ix ? ix->name.c_str() : ""; }
int dsm_index_device(const dsm_index* ix) { return ix ? ix->device : -1; }
uint64_t dsm_index_device_bytes(const dsm_index* ix) { return ix ? (ix->d_blk ? ix->device_bytes : ix->device_bytes - ix->blk_bytes) : 0; }

// Residency: an index that is not needed for a while gives its blocks' HBM back and keeps them in pinned
// host memory; reloading is one asynchronous host-to-device copy (the index is immutable, so the pinned copy is made once).
int dsm_index_resident(const dsm_index* ix) { return ix && ix->d_blk ? 1 : 0; }
int dsm_index_offload(dsm_index* ix) {
    if (!ix) return fail(DSM_E_INVAL, "null index");
    if (!ix->d_blk) return DSM_OK;
    if (ix->d_wt_blob) return fail(DSM_E_UNSUPPORTED, "an index opened with DSM_OPEN_KEEP_WT cannot be offloaded");
    DSM_HIP(hipSetDevice(ix->device));
    if (!ix->h_blk) {
        DSM_HIP(hipHostMalloc(&ix->h_blk, ix->blk_bytes));
        DSM_HIP(hipMemcpy(ix->h_blk, ix->d_blk, ix->blk_bytes, hipMemcpyDeviceToHost));
    }
    DSM_HIP(hipDeviceSynchronize());  // nothing may still read the blocks
    DSM_HIP(hipFree(ix->d_blk));
    ix->d_blk = nullptr;
    ix->dev.blk = nullptr;
    return DSM_OK;
}
int dsm_index_reload(dsm_index* ix, void* stream) {
    if (!ix) return fail(DSM_E_INVAL, "null index");
    if (ix->d_blk) return DSM_OK;
    DSM_HIP(hipSetDevice(ix->device));
    void* p = nullptr;
    if (hipMalloc(&p, ix->blk_bytes) != hipSuccess) return fail(DSM_E_NOMEM, "dsm_index_reload: hipMalloc failed");
    hipError_t e = hipMemcpyAsync(p, ix->h_blk, ix->blk_bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipFree(p); return fail(DSM_E_HIP, std::string("dsm_index_reload: ") + hipGetErrorString(e)); }
    ix->d_blk = p;
    ix->dev.blk = (const Blk*)p;  // work queued on `stream` after this call sees the blocks; other streams must wait for it
    return DSM_OK;
}

int dsm_lf_batch_dev(const dsm_index* ix, const uint8_t* c, const uint64_t* i, uint64_t* out, size_t k, unsigned layout, void* stream) {
    return lf_dev(ix, c, i, out, k, layout, (hipStream_t)stream);
}

int dsm_lf_batch(const dsm_index* ix, const uint8_t* c, const uint64_t* i, uint64_t* out, size_t k, void* stream) {
    if (!ix || (k && (!c || !i || !out))) return fail(DSM_E_INVAL, "dsm_lf_batch: null argument");
    if (k == 0) return DSM_OK;
    DSM_HIP(hipSetDevice(ix->device));
    TmpDev tmp;
    u8* dc = tmp.get<u8>(k);
    u64* di = tmp.get<u64>(k);
    u64* dout = tmp.get<u64>(k);
    if (!dc || !di || !dout) return fail(DSM_E_NOMEM, "hipMalloc failed");
    hipStream_t st = (hipStream_t)stream;
    DSM_HIP(hipMemcpyAsync(dc, c, k, hipMemcpyHostToDevice, st));
    DSM_HIP(hipMemcpyAsync(di, i, k * 8, hipMemcpyHostToDevice, st));
    int rc = lf_dev(ix, dc, di, dout, k, DSM_LAYOUT_PLANES, st);
    if (rc) return rc;
    DSM_HIP(hipMemcpy(out, dout, k * 8, hipMemcpyDeviceToHost));
    return DSM_OK;
}

int dsm_getl_batch(const dsm_index* ix, const uint64_t* i, uint8_t* out, size_t k, void* stream) {
    if (!ix || (k && (!i || !out))) return fail(DSM_E_INVAL, "dsm_getl_batch: null argument");
    if (k == 0) return DSM_OK;
    if (!ix->dev.blk) return fail(DSM_E_INVAL, "the index is offloaded: dsm_index_reload first");
    DSM_HIP(hipSetDevice(ix->device));
    TmpDev tmp;
    u64* di = tmp.get<u64>(k);
    u8* dout = tmp.get<u8>(k);
    u8* dc2b = tmp.get<u8>(8);
    if (!di || !dout || !dc2b) return fail(DSM_E_NOMEM, "hipMalloc failed");
    hipStream_t st = (hipStream_t)stream;
    DSM_HIP(hipMemcpyAsync(di, i, k * 8, hipMemcpyHostToDevice, st));
    DSM_HIP(hipMemcpyAsync(dc2b, ix->meta.code2byte, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(getl_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, ix->dev, dc2b, di, dout, k);
    DSM_HIP(hipGetLastError());
    DSM_HIP(hipStreamSynchronize(st));
    DSM_HIP(hipMemcpy(out, dout, k, hipMemcpyDeviceToHost));
    return DSM_OK;
}

int dsm_index_check(const dsm_index* ix, uint64_t* total) {  // metaenumerate.cpp:93-127
    if (!ix || !total) return fail(DSM_E_INVAL, "null argument");
    std::vector<u8> c;
    std::vector<u64> pos;
    for (int ch = 0; ch < 255; ++ch) {
        c.push_back((u8)ch); pos.push_back((u64)0 - 1);
        c.push_back((u8)ch); pos.push_back(ix->meta.n - 1);
    }
    std::vector<u64> out(c.size());
    int rc = dsm_lf_batch(ix, c.data(), pos.data(), out.data(), c.size(), nullptr);
    if (rc) return rc;
    u64 tot = 0;
    for (size_t k = 0; k < out.size(); k += 2) {
        u64 nmin = out[k], nmax = out[k + 1] - 1;
        if (nmax >= nmin) tot += nmax - nmin + 1;
    }
    *total = tot;
    return DSM_OK;
}

void dsm_free(void* p) { free(p); }

}  // extern "C"
