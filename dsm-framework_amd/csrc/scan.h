// scan.h -- device-wide exclusive prefix sum (reduce / scan-of-sums / downsweep), wave64 shuffles.
// Used for frontier compaction, candidate offsets and the per-block symbol counts of the index.
#pragma once
#include <cstdint>

#include "common.h"

namespace dsm {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns exclusive prefix, total in *tot
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* tot) {
    __shared__ T wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = wave_inclusive_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    T base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / 64; ++k) {
        T s = wsum[k];
        if (k < w) base += s;
        total += s;
    }
    __syncthreads();
    *tot = total;
    return base + inc - v;
}

// A thread owns SCAN_ITEMS consecutive elements.  With VEC (both arrays 16-byte aligned) a full group moves as one or two
// wide accesses per thread; element-wise accesses at a stride of SCAN_ITEMS elements would touch every line of the tile
// SCAN_ITEMS times.
template <typename T>
struct alignas(sizeof(T) * SCAN_ITEMS >= 16 ? 16 : sizeof(T) * SCAN_ITEMS) ScanGroup { T v[SCAN_ITEMS]; };

template <bool VEC, typename InT, typename OutT>
__device__ __forceinline__ void scan_load(const InT* __restrict__ in, size_t base, size_t n, OutT v[SCAN_ITEMS]) {
    if (VEC && base + SCAN_ITEMS <= n) {
        const ScanGroup<InT> g = *reinterpret_cast<const ScanGroup<InT>*>(in + base);
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = (OutT)g.v[k];
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = base + k < n ? (OutT)in[base + k] : (OutT)0;
    }
}

template <bool VEC, typename InT, typename OutT>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const InT* __restrict__ in, size_t n, OutT* __restrict__ sums) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    OutT v[SCAN_ITEMS];
    scan_load<VEC, InT, OutT>(in, base, n, v);
    OutT s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
    OutT tot;
    block_exclusive_scan<OutT>(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// out[i] = offsets[block] + exclusive prefix inside the tile.  in/out may alias.
// RAW: `offsets` holds the tiles' SUMS (what scan_reduce_kernel left), not their scan: a block adds up the sums of the tiles before
// it by itself -- at most SCAN_TILE of them, a few KB that sit in L2 -- which saves the single-block launch that would scan them.
template <bool VEC, typename InT, typename OutT, bool RAW = false>
__global__ __launch_bounds__(SCAN_THREADS) void scan_down_kernel(const InT* in, OutT* out, size_t n, const OutT* __restrict__ offsets,
                                                                OutT* __restrict__ total) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    OutT v[SCAN_ITEMS];
    scan_load<VEC, InT, OutT>(in, base, n, v);
    OutT s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) s += v[k];
    OutT off = 0;
    if (RAW) {
        OutT part = 0;
        for (unsigned q = threadIdx.x; q < blockIdx.x; q += SCAN_THREADS) part += offsets[q];
        OutT before;
        block_exclusive_scan<OutT>(part, &before);
        off = before;
    } else if (offsets) {
        off = offsets[blockIdx.x];
    }
    OutT tot;
    OutT ex = block_exclusive_scan<OutT>(s, &tot);
    ex += off;
    if (VEC && base + SCAN_ITEMS <= n) {
        ScanGroup<OutT> g;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) { g.v[k] = ex; ex += v[k]; }
        *reinterpret_cast<ScanGroup<OutT>*>(out + base) = g;
    } else {
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (base + k < n) out[base + k] = ex;
            ex += v[k];
        }
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = off + tot;
}

inline size_t scan_tmp_elems(size_t n) {
    size_t t = 0;
    while (n > (size_t)SCAN_TILE) {
        n = (n + SCAN_TILE - 1) / SCAN_TILE;
        t += n;
    }
    return t + 1;
}

// Exclusive scan of in[0..n) into out[0..n) (may alias when InT == OutT); grand total to *d_total (device, may be null).
// tmp must hold scan_tmp_elems(n) OutT values.
template <typename InT, typename OutT>
inline void exclusive_scan(const InT* in, OutT* out, size_t n, OutT* tmp, OutT* d_total, hipStream_t st) {
    if (n == 0) {
        if (d_total) (void)hipMemsetAsync(d_total, 0, sizeof(OutT), st);
        return;
    }
    const bool vec = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nb == 1) {
        if (vec) hipLaunchKernelGGL((scan_down_kernel<true, InT, OutT>), dim3(1), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)nullptr, d_total);
        else hipLaunchKernelGGL((scan_down_kernel<false, InT, OutT>), dim3(1), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)nullptr, d_total);
        return;
    }
    if (vec) hipLaunchKernelGGL((scan_reduce_kernel<true, InT, OutT>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, tmp);
    else hipLaunchKernelGGL((scan_reduce_kernel<false, InT, OutT>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, tmp);
    if (nb <= (size_t)SCAN_TILE) {  // two launches: every block of the down-sweep adds up the sums before it (see RAW)
        if (vec) hipLaunchKernelGGL((scan_down_kernel<true, InT, OutT, true>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)tmp, d_total);
        else hipLaunchKernelGGL((scan_down_kernel<false, InT, OutT, true>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)tmp, d_total);
        return;
    }
    exclusive_scan<OutT, OutT>(tmp, tmp, nb, tmp + nb, d_total, st);
    if (vec) hipLaunchKernelGGL((scan_down_kernel<true, InT, OutT>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)tmp, (OutT*)nullptr);
    else hipLaunchKernelGGL((scan_down_kernel<false, InT, OutT>), dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, out, n, (const OutT*)tmp, (OutT*)nullptr);
}

}  // namespace dsm
