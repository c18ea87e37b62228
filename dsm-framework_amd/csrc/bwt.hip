// bwt.hip -- multi-string BWT on the GPU for the index writer, sized for configs[3]: n = 8.08e9 symbols.
//
// The reference builder stores every read as reverse(read + '-' + revcomp(read)) + '\0' (builder.cpp:183-201,
// TextCollectionBuilder.cpp:65-92) and builds the BWT of the collection with the terminators ordered by insertion,
// $_0 < $_1 < ... < every other byte (incbwt/misc/utils.cpp:362-367, incbwt/rlcsa_builder.cpp:35-78,165-179).  Strings are
// short (a read of 100 bp is 202 symbols), so the suffix order is a bounded-depth sort:
//   * symbols are re-coded in 4 bits (0 = terminator, then the bytes present in increasing order: at most 15);
//   * suffixes are bucketed by their first 3 symbols; buckets are processed in batches of at most BATCH suffixes, collected in
//     text order (= terminator order: equal strings must come out in text order, and every sort below is stable);
//   * a batch is sorted by its first 15 symbols (one 60-bit key), then the runs that are still tied AND have not reached their
//     terminator are refined, 16 symbols per round, by a stable sort on (run, next key); a key that holds a terminator ends the
//     comparison (digits after it are 0), so a run of such keys is complete and stays in text order;
//   * BWT[rank] = the byte before the suffix (0 at the start of a string, FMIndex.cpp / TextCollectionBuilder semantics).
// Sorting primitive: rocPRIM's stable radix sort (a plain library sort, off the enumeration path); everything else is kernels
// of this file.  Memory: the text, the BWT and about 60 bytes per suffix of a batch.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>

#include "common.h"
#include "scan.h"

namespace dsm {

struct CodeTab {
    u8 c[256];
};

// 4-bit code of the symbol at p (0 for a terminator; positions at or beyond n read as terminators)
__device__ __forceinline__ u32 code_at(const u8* __restrict__ t, u64 n, const CodeTab& tab, u64 p) { return p < n ? tab.c[t[p]] : 0u; }

// `nsym` code digits starting at p, most significant first; digits after a terminator are 0
__device__ __forceinline__ u64 pack_digits(const u8* __restrict__ t, u64 n, const CodeTab& tab, u64 p, int nsym) {
    u64 k = 0;
    bool ended = false;
    for (int j = 0; j < nsym; ++j) {
        u32 c = ended ? 0u : code_at(t, n, tab, p + j);
        if (c == 0) ended = true;
        k = (k << 4) | c;
    }
    return k;
}

__global__ void byte_hist_kernel(const u8* __restrict__ t, u64 n, unsigned long long* __restrict__ hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (u64)gridDim.x * blockDim.x) atomicAdd(&h[t[p]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// suffixes per 3-symbol bucket (4096 buckets)
__global__ void bucket_hist_kernel(const u8* __restrict__ t, u64 n, CodeTab tab, unsigned long long* __restrict__ hist) {
    __shared__ u32 h[4096];
    for (int q = threadIdx.x; q < 4096; q += blockDim.x) h[q] = 0;
    __syncthreads();
    for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (u64)gridDim.x * blockDim.x)
        atomicAdd(&h[(u32)pack_digits(t, n, tab, p, 3)], 1u);
    __syncthreads();
    for (int q = threadIdx.x; q < 4096; q += blockDim.x)
        if (h[q]) atomicAdd(&hist[q], (unsigned long long)h[q]);
}

// collection of the suffixes of a bucket range [k0, k1), in text order: counts per block of 2048 positions, then the positions
constexpr u32 COLLECT_TILE = 2048;
__global__ __launch_bounds__(256) void collect_count_kernel(const u8* __restrict__ t, u64 n, CodeTab tab, u64 p0, u64 np, u32 k0, u32 k1,
                                                            u32* __restrict__ counts) {
    const u64 base = p0 + (u64)blockIdx.x * COLLECT_TILE;
    u32 c = 0;
    for (u32 j = threadIdx.x; j < COLLECT_TILE; j += 256) {
        const u64 p = base + j;
        if (p < p0 + np) {
            const u32 k = (u32)pack_digits(t, n, tab, p, 3);
            c += k >= k0 && k < k1;
        }
    }
    u32 tot;
    block_exclusive_scan<u32>(c, &tot);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void collect_write_kernel(const u8* __restrict__ t, u64 n, CodeTab tab, u64 p0, u64 np, u32 k0, u32 k1,
                                                            const u64* __restrict__ offs, u64 obase, u64* __restrict__ pos,
                                                            u64* __restrict__ key) {
    // a thread owns 8 consecutive positions so that the output keeps text order
    const u64 base = p0 + (u64)blockIdx.x * COLLECT_TILE + (u64)threadIdx.x * 8;
    u32 c = 0;
    u32 ks[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 p = base + j;
        ks[j] = 0xFFFFFFFFu;
        if (p < p0 + np) {
            const u32 k = (u32)pack_digits(t, n, tab, p, 3);
            if (k >= k0 && k < k1) { ks[j] = k; ++c; }
        }
    }
    u32 tot;
    u32 ex = block_exclusive_scan<u32>(c, &tot);
    u64 o = obase + offs[blockIdx.x] + ex;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (ks[j] != 0xFFFFFFFFu) {
            const u64 p = base + j;
            pos[o] = p;
            key[o] = pack_digits(t, n, tab, p, 15);  // the first 15 symbols (in one go: a terminator among the first three ends the key)
            ++o;
        }
    }
}

// round keys: 16 symbols at offset `off` of the suffix
__global__ void round_key_kernel(const u8* __restrict__ t, u64 n, CodeTab tab, const u64* __restrict__ pos, u64 m, u32 off, u64* __restrict__ key) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) key[i] = pack_digits(t, n, tab, pos[i] + off, 16);
}

// head[i] = 1 when element i starts a new run (run ids given: a run never spans two old runs)
__global__ void head_kernel(const u64* __restrict__ key, const u32* __restrict__ run, u64 m, u32* __restrict__ head) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) head[i] = (i == 0 || key[i] != key[i - 1] || (run && run[i] != run[i - 1])) ? 1u : 0u;
}
// need[i] = 1 when element i is still tied with a neighbour of its run and its key holds no terminator (last digit non-zero)
__global__ void need_kernel(const u64* __restrict__ key, const u32* __restrict__ head, u64 m, u32* __restrict__ need) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const bool tied = !head[i] || (i + 1 < m && !head[i + 1]);
    need[i] = (tied && (key[i] & 15ull)) ? 1u : 0u;
}
// compaction of the tied elements: their slot in the batch (or, from the second round on, the slot their predecessor recorded),
// their position and the id of their run (= inclusive count of heads - 1)
__global__ void compact_kernel(const u32* __restrict__ need, const u64* __restrict__ noff, const u64* __restrict__ hoff, const u32* __restrict__ head,
                               const u64* __restrict__ pos, const u64* __restrict__ slot_in, u64 m, u64* __restrict__ slot_out,
                               u64* __restrict__ pos_out, u32* __restrict__ run_out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || !need[i]) return;
    const u64 o = noff[i];
    slot_out[o] = slot_in ? slot_in[i] : i;
    pos_out[o] = pos[i];
    run_out[o] = (u32)(hoff[i] + head[i] - 1);  // heads up to and including i, minus one
}
__global__ void iota_kernel(u32* __restrict__ v, u64 m) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) v[i] = (u32)i;
}
template <typename T>
__global__ void gather_kernel(const T* __restrict__ in, const u32* __restrict__ perm, u64 m, T* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = in[perm[i]];
}
__global__ void scatter_pos_kernel(const u64* __restrict__ slot, const u64* __restrict__ pos, u64 m, u64* __restrict__ main_pos) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) main_pos[slot[i]] = pos[i];
}
__global__ void emit_bwt_kernel(const u8* __restrict__ t, const u64* __restrict__ pos, u64 m, u8* __restrict__ bwt) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const u64 p = pos[i];
    bwt[i] = p == 0 ? (u8)0 : t[p - 1];  // the byte before a string's first symbol is the previous terminator: 0 either way
}

static inline dim3 g1(u64 n) { return dim3((unsigned)((n + 255) / 256)); }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) return fail(DSM_E_NOMEM, "dsm_bwt_build: hipMalloc failed");
        cap = bytes;
        return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <typename K, typename V>
static int sort_pairs(DevBuf& tmp, const K* kin, K* kout, const V* vin, V* vout, u64 m, unsigned begin_bit, unsigned end_bit, hipStream_t st) {
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)m, begin_bit, end_bit, st) != hipSuccess)
        return fail(DSM_E_HIP, "rocprim::radix_sort_pairs (size query) failed");
    if (int rc = tmp.ensure(bytes + 256)) return rc;
    if (rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, (size_t)m, begin_bit, end_bit, st) != hipSuccess)
        return fail(DSM_E_HIP, "rocprim::radix_sort_pairs failed");
    return 0;
}

}  // namespace dsm

using namespace dsm;

extern "C" int dsm_bwt_build(const uint8_t* d_text, uint64_t n, uint8_t* d_bwt, int device, void* stream) {
    if (!d_text || !d_bwt || n == 0) return fail(DSM_E_INVAL, "dsm_bwt_build: null argument");
    hipStream_t st = (hipStream_t)stream;
    DSM_HIP(hipSetDevice(device));
    // ---- alphabet ----
    DevBuf dh;
    if (int rc = dh.ensure(4096 * sizeof(unsigned long long))) return rc;
    unsigned long long* d_hist = (unsigned long long*)dh.p;
    DSM_HIP(hipMemsetAsync(d_hist, 0, 256 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(byte_hist_kernel, dim3(4096), dim3(256), 0, st, d_text, n, d_hist);
    std::vector<unsigned long long> hb(256);
    DSM_HIP(hipMemcpyAsync(hb.data(), d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    DSM_HIP(hipStreamSynchronize(st));
    CodeTab tab;
    memset(&tab, 0, sizeof tab);
    u32 ncodes = 1;
    for (int b = 1; b < 256; ++b)
        if (hb[b]) {
            if (ncodes > 15) return fail(DSM_E_UNSUPPORTED, "dsm_bwt_build: more than 15 distinct symbols");
            tab.c[b] = (u8)ncodes++;
        }
    {   // the last byte of the text must be a terminator (every string ends in one)
        u8 last = 1;
        DSM_HIP(hipMemcpy(&last, d_text + n - 1, 1, hipMemcpyDeviceToHost));
        if (last != 0) return fail(DSM_E_INVAL, "dsm_bwt_build: the text does not end in a terminator");
    }
    // ---- buckets of the first three symbols ----
    DSM_HIP(hipMemsetAsync(d_hist, 0, 4096 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(bucket_hist_kernel, dim3(2048), dim3(256), 0, st, d_text, n, tab, d_hist);
    std::vector<unsigned long long> hk(4096);
    DSM_HIP(hipMemcpyAsync(hk.data(), d_hist, 4096 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    DSM_HIP(hipStreamSynchronize(st));
    u64 batch_cap = 1ull << 27;
    if (const char* e = getenv("DSM_BWT_BATCH")) batch_cap = strtoull(e, nullptr, 10);
    for (u32 k = 0; k < 4096; ++k)
        if (hk[k] > 0xFFFFFF00ull) return fail(DSM_E_CAPACITY, "dsm_bwt_build: a 3-symbol bucket holds more than 2^32 suffixes");
    // ---- batches ----
    DevBuf b_pos[2], b_key[2], b_cnt, b_off, b_scan, b_sort, b_head, b_need, b_noff, b_hoff;
    struct RoundSet { DevBuf slot, posraw, possorted, run, keysorted; } sets[2];  // a round reads the previous round's set and writes its own
    DevBuf t_ka, t_kb, t_perm_a, t_perm_b, t_perm_c, t_run_sorted, t_run_out;
    const u64 CHUNK = 1ull << 30;  // positions per collection launch
    u64 out_base = 0;
    u32 k0 = 0;
    while (k0 < 4096) {
        u64 m = 0;
        u32 k1 = k0;
        while (k1 < 4096 && (m == 0 || m + hk[k1] <= batch_cap)) m += hk[k1++];
        if (m == 0) { k0 = k1; continue; }
        for (int q = 0; q < 2; ++q) {
            if (int rc = b_pos[q].ensure(m * 8)) return rc;
            if (int rc = b_key[q].ensure(m * 8)) return rc;
        }
        u64* pos = (u64*)b_pos[0].p;
        u64* key = (u64*)b_key[0].p;
        // collect in text order
        u64 got = 0;
        for (u64 p0 = 0; p0 < n; p0 += CHUNK) {
            const u64 np = std::min(CHUNK, n - p0);
            const u64 nb = (np + COLLECT_TILE - 1) / COLLECT_TILE;
            if (int rc = b_cnt.ensure(nb * 4)) return rc;
            if (int rc = b_off.ensure(nb * 8)) return rc;
            if (int rc = b_scan.ensure((scan_tmp_elems(nb) + 8) * 8)) return rc;
            if (int rc = dh.ensure(4096 * 8)) return rc;
            hipLaunchKernelGGL(collect_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_text, n, tab, p0, np, k0, k1, (u32*)b_cnt.p);
            u64* d_tot = (u64*)dh.p;
            exclusive_scan<u32, u64>((const u32*)b_cnt.p, (u64*)b_off.p, nb, (u64*)b_scan.p, d_tot, st);
            hipLaunchKernelGGL(collect_write_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_text, n, tab, p0, np, k0, k1, (const u64*)b_off.p, got, pos, key);
            u64 tot = 0;
            DSM_HIP(hipMemcpyAsync(&tot, d_tot, 8, hipMemcpyDeviceToHost, st));
            DSM_HIP(hipStreamSynchronize(st));
            got += tot;
        }
        if (got != m) return fail(DSM_E_HIP, "dsm_bwt_build: bucket counts and collection disagree");
        // round 0: the first 15 symbols
        if (int rc = sort_pairs<u64, u64>(b_sort, key, (u64*)b_key[1].p, pos, (u64*)b_pos[1].p, m, 0, 60, st)) return rc;
        pos = (u64*)b_pos[1].p;
        key = (u64*)b_key[1].p;
        // refinement rounds on the elements that are still tied
        if (int rc = b_head.ensure(m * 4)) return rc;
        if (int rc = b_need.ensure(m * 4)) return rc;
        if (int rc = b_noff.ensure(m * 8)) return rc;
        if (int rc = b_hoff.ensure(m * 8)) return rc;
        if (int rc = b_scan.ensure((scan_tmp_elems(m) + 8) * 8)) return rc;
        u64 cm = m;                // elements of the current (compacted) working set
        const u64* ckey = key;     // their keys, positions, runs, slots in the batch
        const u64* cpos = pos;
        const u32* crun = nullptr;
        const u64* cslot = nullptr;
        u32 off = 15;
        for (int round = 0;; ++round) {
            hipLaunchKernelGGL(head_kernel, g1(cm), dim3(256), 0, st, ckey, crun, cm, (u32*)b_head.p);
            hipLaunchKernelGGL(need_kernel, g1(cm), dim3(256), 0, st, ckey, (const u32*)b_head.p, cm, (u32*)b_need.p);
            u64* d_tot = (u64*)dh.p;
            exclusive_scan<u32, u64>((const u32*)b_need.p, (u64*)b_noff.p, cm, (u64*)b_scan.p, d_tot, st);
            exclusive_scan<u32, u64>((const u32*)b_head.p, (u64*)b_hoff.p, cm, (u64*)b_scan.p, d_tot + 1, st);
            u64 nm = 0;
            DSM_HIP(hipMemcpyAsync(&nm, d_tot, 8, hipMemcpyDeviceToHost, st));
            DSM_HIP(hipStreamSynchronize(st));
            if (nm == 0) break;
            if (off > 1000000) return fail(DSM_E_UNSUPPORTED, "dsm_bwt_build: strings longer than a million symbols");
            RoundSet& S = sets[round & 1];
            if (int rc = S.slot.ensure(nm * 8)) return rc;
            if (int rc = S.posraw.ensure(nm * 8)) return rc;
            if (int rc = S.possorted.ensure(nm * 8)) return rc;
            if (int rc = S.run.ensure(nm * 4)) return rc;
            if (int rc = S.keysorted.ensure(nm * 8)) return rc;
            if (int rc = t_ka.ensure(nm * 8)) return rc;
            if (int rc = t_kb.ensure(nm * 8)) return rc;
            if (int rc = t_perm_a.ensure(nm * 4)) return rc;
            if (int rc = t_perm_b.ensure(nm * 4)) return rc;
            if (int rc = t_perm_c.ensure(nm * 4)) return rc;
            if (int rc = t_run_sorted.ensure(nm * 4)) return rc;
            if (int rc = t_run_out.ensure(nm * 4)) return rc;
            hipLaunchKernelGGL(compact_kernel, g1(cm), dim3(256), 0, st, (const u32*)b_need.p, (const u64*)b_noff.p, (const u64*)b_hoff.p,
                               (const u32*)b_head.p, cpos, cslot, cm, (u64*)S.slot.p, (u64*)S.posraw.p, (u32*)S.run.p);
            // next 16 symbols; stable sort by key, then by run: the result is ordered by (run, key), ties in text order
            hipLaunchKernelGGL(round_key_kernel, g1(nm), dim3(256), 0, st, d_text, n, tab, (const u64*)S.posraw.p, nm, off, (u64*)t_ka.p);
            hipLaunchKernelGGL(iota_kernel, g1(nm), dim3(256), 0, st, (u32*)t_perm_a.p, nm);
            if (int rc = sort_pairs<u64, u32>(b_sort, (const u64*)t_ka.p, (u64*)t_kb.p, (const u32*)t_perm_a.p, (u32*)t_perm_b.p, nm, 0, 64, st)) return rc;
            hipLaunchKernelGGL((gather_kernel<u32>), g1(nm), dim3(256), 0, st, (const u32*)S.run.p, (const u32*)t_perm_b.p, nm, (u32*)t_run_sorted.p);
            if (int rc = sort_pairs<u32, u32>(b_sort, (const u32*)t_run_sorted.p, (u32*)t_run_out.p, (const u32*)t_perm_b.p, (u32*)t_perm_c.p, nm, 0, 32, st)) return rc;
            // apply: positions and keys in the new order; slots and run ids keep their (ascending) order
            hipLaunchKernelGGL((gather_kernel<u64>), g1(nm), dim3(256), 0, st, (const u64*)S.posraw.p, (const u32*)t_perm_c.p, nm, (u64*)S.possorted.p);
            hipLaunchKernelGGL((gather_kernel<u64>), g1(nm), dim3(256), 0, st, (const u64*)t_ka.p, (const u32*)t_perm_c.p, nm, (u64*)S.keysorted.p);
            hipLaunchKernelGGL(scatter_pos_kernel, g1(nm), dim3(256), 0, st, (const u64*)S.slot.p, (const u64*)S.possorted.p, nm, pos);
            ckey = (const u64*)S.keysorted.p;
            cpos = (const u64*)S.possorted.p;
            crun = (const u32*)S.run.p;
            cslot = (const u64*)S.slot.p;
            cm = nm;
            off += 16;
        }
        hipLaunchKernelGGL(emit_bwt_kernel, g1(m), dim3(256), 0, st, d_text, (const u64*)pos, m, d_bwt + out_base);
        DSM_HIP(hipGetLastError());
        DSM_HIP(hipStreamSynchronize(st));
        out_base += m;
        k0 = k1;
    }
    if (out_base != n) return fail(DSM_E_HIP, "dsm_bwt_build: suffix count mismatch");
    return DSM_OK;
}
