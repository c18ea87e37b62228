// mix_calib2.hip -- which property of the expand kernel costs throughput?  Variants of the access mix of mix_calib.hip:
//   DEP   the block index comes from a loaded record field (dependent round trip) instead of a hash of the thread id
//   OCC   LDS padding limits the CU to 5 waves per SIMD like the 82-VGPR kernel
//   VALU  extra integer work per thread (~1000 instructions per wave)
//   BAR   a block-wide barrier + one atomicAdd by thread 0 + barrier between the loads and the stores
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
template <bool DEP, int LDSPAD, int VALU, int BAR>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ tab, uint64_t nblk, const uint32_t* __restrict__ rec,
                                                  uint32_t* __restrict__ outrec, uint64_t nq, uint32_t p2, uint32_t* __restrict__ alloc) {
    __shared__ uint32_t pad[LDSPAD > 0 ? LDSPAD : 1];
    __shared__ uint32_t sb;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (LDSPAD > 0 && threadIdx.x == 0) pad[blockIdx.x % LDSPAD] = 1;
    const bool ok = i < nq;
    uint32_t acc = 0, f0 = 0;
    if (ok) {
        f0 = rec[i];
#pragma unroll
        for (int f = 1; f < 11; ++f) acc += rec[(uint64_t)f * nq + i];
    }
    uint64_t h = DEP ? mix((uint64_t)f0 * 0x9E3779B97F4A7C15ull + 777) : mix(i * 0x9E3779B97F4A7C15ull + 12345);
    uint64_t b = h % (nblk - 1);
    const bool two = ((h >> 40) & 1023) < p2;
    const uint4* p = tab + b * 4;
    uint4 v[4], w[4];
    if (ok) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p[k];
        if (two) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = p[4 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += __popc(v[k].x) + __popc(v[k].y) + __popc(v[k].z) + __popc(v[k].w);
        if (two) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += __popc(w[k].x) + __popc(w[k].y) + __popc(w[k].z) + __popc(w[k].w);
        }
#pragma unroll 8
        for (int r = 0; r < VALU; ++r) acc = acc * 1664525u + (acc >> 7) + r;
    }
    uint64_t o = i;
    if (BAR == 1) {         // barrier, one atomic per block on one hot word, barrier
        __syncthreads();
        if (threadIdx.x == 0) sb = atomicAdd(alloc, 256u);
        __syncthreads();
        o = (uint64_t)sb + threadIdx.x;
    } else if (BAR == 2) {  // the two barriers without the atomic
        __syncthreads();
        if (threadIdx.x == 0) sb = blockIdx.x * 256u;
        __syncthreads();
        o = (uint64_t)sb + threadIdx.x;
    } else if (BAR == 3) {  // one atomic per wave, no barrier
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(alloc, 64u);
        base = __shfl(base, 0, 64);
        o = (uint64_t)base + (threadIdx.x & 63);
    } else if (BAR == 4) {  // barrier, one atomic per block on one of 32 words (own lines), barrier
        __syncthreads();
        if (threadIdx.x == 0) sb = (blockIdx.x & 31) * (uint32_t)(nq / 32) + atomicAdd(alloc + (blockIdx.x & 31) * 32, 256u);
        __syncthreads();
        o = (uint64_t)sb + threadIdx.x;
    } else if (BAR == 5) {  // one atomic per wave on one of 32 words, no barrier
        uint32_t base = 0;
        const uint32_t sh = (blockIdx.x * 4 + (threadIdx.x >> 6)) & 31;
        if ((threadIdx.x & 63) == 0) base = sh * (uint32_t)(nq / 32) + atomicAdd(alloc + sh * 32, 64u);
        base = __shfl(base, 0, 64);
        o = (uint64_t)base + (threadIdx.x & 63);
    }
    if (ok && o < nq) {
#pragma unroll
        for (int f = 0; f < 12; ++f) outrec[(uint64_t)f * nq + o] = acc + f;
    }
    if (LDSPAD > 0 && pad[0] == 12345) outrec[0] = 1;
}
template <bool DEP, int LDSPAD, int VALU, int BAR>
static int run(const char* name, const uint4* tab, uint64_t bytes, const uint32_t* rec, uint32_t* outrec, uint64_t nq, uint32_t* alloc) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(alloc, 0, 4096));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((mix_kernel<DEP, LDSPAD, VALU, BAR>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / 64, rec, outrec, nq, 133u, alloc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("%-44s %.3f ms, %.2f G nodes/s\n", name, ms, nq / ms / 1e6);
    }
    return 0;
}
__global__ void fill(uint32_t* r, uint64_t n) { uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) r[i] = (uint32_t)mix(i + 99); }
int main(int argc, char** argv) {
    uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1024;
    uint64_t nq = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 26);
    uint64_t bytes = mib << 20;
    uint4* tab; uint32_t *rec, *outrec, *alloc;
    CK(hipMalloc(&tab, bytes + 256)); CK(hipMalloc(&rec, nq * 44)); CK(hipMalloc(&outrec, nq * 48)); CK(hipMalloc(&alloc, 4096));
    CK(hipMemset(tab, 1, bytes + 256));
    hipLaunchKernelGGL(fill, dim3((unsigned)((nq * 11 + 255) / 256)), dim3(256), 0, 0, rec, nq * 11);
    CK(hipDeviceSynchronize());
    // LDS: 160 KB per CU; 5 waves/SIMD = 20 waves = 5 blocks of 256 -> 32 KB per block = 8192 words
    run<true, 7800, 250, 0>("no allocation", tab, bytes, rec, outrec, nq, alloc);
    run<true, 7800, 250, 1>("barrier + hot atomic per block", tab, bytes, rec, outrec, nq, alloc);
    run<true, 7800, 250, 2>("barriers only", tab, bytes, rec, outrec, nq, alloc);
    run<true, 7800, 250, 3>("hot atomic per wave, no barrier", tab, bytes, rec, outrec, nq, alloc);
    run<true, 7800, 250, 4>("barrier + 32-way sharded atomic per block", tab, bytes, rec, outrec, nq, alloc);
    run<true, 7800, 250, 5>("32-way sharded atomic per wave, no barrier", tab, bytes, rec, outrec, nq, alloc);
    return 0;
}
