// gather_calib.hip -- calibration of rocprofv3 FETCH_SIZE / TCC_EA0_RDREQ on THIS access pattern:
// every thread reads one random 64-byte block (4 x dwordx4) of a table far larger than L2 + Infinity Cache,
// exactly like load_blk() in the expand kernel.  Known bytes = nthreads * 64 (or *128 with -b 128).
// Also prints the achieved random-line rate (the practical HBM roofline for this path).
//   ./gather_calib [table_MiB=4096] [nqueries=2^28] [block_bytes=64]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
template <int BB>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, uint64_t nblk, uint64_t nq, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    uint64_t b = mix(i * 0x9E3779B97F4A7C15ull + 12345) % nblk;
    const uint4* p = tab + b * (BB / 16);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < BB / 16; ++k) { uint4 v = p[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;  // keep the loads alive
}
__global__ void stream_kernel(const uint4* __restrict__ tab, uint64_t n16, uint32_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = tab[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
    uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 4096;
    uint64_t nq = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 28);
    int bb = argc > 3 ? atoi(argv[3]) : 64;
    uint64_t bytes = mib << 20;
    uint4* tab; uint32_t* out;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (bb == 32) hipLaunchKernelGGL(gather_kernel<32>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / 32, nq, out);
        else if (bb == 16) hipLaunchKernelGGL(gather_kernel<16>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / 16, nq, out);
        else if (bb == 128) hipLaunchKernelGGL(gather_kernel<128>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / 128, nq, out);
        else hipLaunchKernelGGL(gather_kernel<64>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, tab, bytes / 64, nq, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("gather%d: %llu queries, known bytes %.3f GB, %.3f ms, %.2f G lines/s, %.1f GB/s\n", bb, (unsigned long long)nq, nq * (double)bb / 1e9, ms,
               nq / ms / 1e6, nq * (double)bb / ms / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 32), dim3(256), 0, 0, tab, bytes / 16, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("stream: known bytes %.3f GB, %.3f ms, %.1f GB/s\n", bytes / 1e9, ms, bytes / ms / 1e6);
    }
    return 0;
}
