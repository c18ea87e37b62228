// mix_calib.hip -- the expand kernel's access MIX as a microbenchmark: per thread 44 B of coalesced record reads, one random
// index block (plus the adjacent one with probability p2), 47 B of coalesced writes.  Compares the 64-byte block layout
// (128 symbols: 4 x u32 counts + 3 x 128-bit planes) with a 32-byte block layout (64 symbols: 4 x u16 counts + 3 x 64-bit
// planes, counts relative to a 16-byte mid entry per 65536 symbols that stays L2 resident).
//   ./mix_calib [table_MiB=1024] [nodes=2^26]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
template <int BB, bool MID, bool REC, int NF = 11>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ tab, uint64_t nblk, const uint4* __restrict__ mid, const uint32_t* __restrict__ rec,
                                                  uint32_t* __restrict__ outrec, uint64_t nq, uint32_t p2) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    uint32_t acc = 0;
    if (REC) {
#pragma unroll
        for (int f = 0; f < NF; ++f) acc += rec[(uint64_t)f * nq + i];
    }
    uint64_t h = mix(i * 0x9E3779B97F4A7C15ull + 12345);
    uint64_t b = h % (nblk - 1);
    const bool two = ((h >> 40) & 1023) < p2;
    const uint4* p = tab + b * (BB / 16);
    uint4 v[BB / 16], w[BB / 16];
#pragma unroll
    for (int k = 0; k < BB / 16; ++k) v[k] = p[k];
    if (two) {
#pragma unroll
        for (int k = 0; k < BB / 16; ++k) w[k] = p[BB / 16 + k];
    }
    if (MID) { uint4 m = mid[(b * (BB * 2)) >> 16]; acc += m.x + m.y + m.z + m.w; if (two) { uint4 m2 = mid[((b + 1) * (BB * 2)) >> 16]; acc += m2.x; } }
#pragma unroll
    for (int k = 0; k < BB / 16; ++k) acc += __popc(v[k].x) + __popc(v[k].y) + __popc(v[k].z) + __popc(v[k].w);
    if (two) {
#pragma unroll
        for (int k = 0; k < BB / 16; ++k) acc += __popc(w[k].x) + __popc(w[k].y) + __popc(w[k].z) + __popc(w[k].w);
    }
    if (REC) {
#pragma unroll
        for (int f = 0; f < NF + 1; ++f) outrec[(uint64_t)f * nq + i] = acc + f;
    } else if (acc == 0x12345678u) outrec[0] = acc;
}
template <int BB, bool MID, bool REC, int NF = 11>
static void run(const char* name, const uint4* tab, uint64_t bytes, const uint4* mid, const uint32_t* rec, uint32_t* outrec, uint64_t nq, uint32_t p2) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((mix_kernel<BB, MID, REC, NF>), dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / BB, mid, rec, outrec, nq, p2);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) printf("%-34s p2=%4u/1024: %.3f ms, %.2f G nodes/s\n", name, p2, ms, nq / ms / 1e6);
    }
}
int main(int argc, char** argv) {
    uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1024;
    uint64_t nq = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 26);
    uint64_t bytes = mib << 20;
    uint4 *tab, *mid; uint32_t *rec, *outrec;
    if (hipMalloc(&tab, bytes + 256) != hipSuccess || hipMalloc(&mid, 4 << 20) != hipSuccess || hipMalloc(&rec, nq * 44) != hipSuccess ||
        hipMalloc(&outrec, nq * 48) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 1, bytes + 256); hipMemset(mid, 1, 4 << 20); hipMemset(rec, 1, nq * 44);
    run<64, false, false>("64-B blocks, gather only", tab, bytes, mid, rec, outrec, nq, 133);
    run<32, true, false>("32-B blocks + mid, gather only", tab, bytes, mid, rec, outrec, nq, 256);
    run<32, false, false>("32-B blocks, no mid, gather only", tab, bytes, mid, rec, outrec, nq, 256);
    run<64, false, true>("64-B blocks, with records", tab, bytes, mid, rec, outrec, nq, 133);
    run<32, true, true>("32-B blocks + mid, with records", tab, bytes, mid, rec, outrec, nq, 256);
    run<32, true, true>("32-B blocks + mid, with records", tab, bytes, mid, rec, outrec, nq, 133);
    run<64, false, true, 6>("64-B blocks, 6-field records", tab, bytes, mid, rec, outrec, nq, 133);
    run<64, false, true, 3>("64-B blocks, 3-field records", tab, bytes, mid, rec, outrec, nq, 133);
    run<64, false, true, 0>("64-B blocks, 1 store only", tab, bytes, mid, rec, outrec, nq, 133);
    return 0;
}
