// distmat.hip -- distance matrices of the tuple stream on the GPU.
// Replaces the accumulation loop of wrapper-distance-matrix/smtxt2entropy.c (add(), :167-197; bucket choice :690-703;
// cumulative print :722-752).  Per tuple the host evaluates the normalised entropy with the reference's own expression
// and libm (so the bucket is exactly the tool's) and the device adds, for every pair of samples with at least one
// non-zero frequency, the three squared-distance terms, plus the co-occurrence counts.
//
// Device mapping: a group of G lanes (G = power of two >= max(pairs, samples), at most 64) owns one tuple; the tuple's
// dense frequency / presence vectors live in LDS, lane p owns sample pair p.  Blocks accumulate into LDS copies of the
// matrices (when nmaxent * samples^2 cells fit) and flush once with global atomics.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace dsm {
int fail(int code, const std::string& m);  // index.hip

constexpr int DM_THREADS = 256;
constexpr u32 DM_LDS_CELLS = 1024;  // cells (count u32 + three 128-bit sums = 52 B) kept in LDS per block: 52 KB

// The three sums of squared distances are kept as 128-bit fixed-point numbers (64 integer bits, 64 fraction bits, two's complement)
// and added with integer atomics: integer addition is associative, so a cell's sum does not depend on the order in which lanes,
// blocks or batches reach it -- the matrices are the same from run to run (round 4; round 3 added doubles in whatever order the
// atomics arrived, and the last printed digit could differ).  A term is a double; its bits below 2^-64 are dropped (the terms are
// squares of differences of logs and roots of frequencies, and lgamma values: magnitudes between 1e-10 and 1e12), the sum itself
// is exact, and it is rounded to a double once, at the end.
struct Fix128 { unsigned long long lo, hi; };
__device__ __forceinline__ void fix_add(Fix128* cell, double x) {
    const double fl = floor(x);
    const unsigned long long hi = (unsigned long long)(long long)fl;               // (two's complement of a negative integer part)
    const unsigned long long lo = __double2ull_rz((x - fl) * 18446744073709551616.0);  // the fraction, in [0, 1), scaled by 2^64
    const unsigned long long old = atomicAdd(&cell->lo, lo);
    atomicAdd(&cell->hi, hi + ((old + lo) < old ? 1ull : 0ull));                    // carries commute with everything else
}
__device__ __forceinline__ void fix_merge(Fix128* cell, const Fix128& v) {
    if (!v.lo && !v.hi) return;
    const unsigned long long old = atomicAdd(&cell->lo, v.lo);
    atomicAdd(&cell->hi, v.hi + ((old + v.lo) < old ? 1ull : 0ull));
}
static inline double fix_to_double(const Fix128& v) {   // (host) hi is signed, lo the fraction
    return (double)(long long)v.hi + (double)v.lo * (1.0 / 18446744073709551616.0);
}

struct DmArgs {
    u32 nt;            // tuples of the batch
    u32 s;             // samples
    u32 nm;            // buckets
    u32 G;             // lanes per tuple
    u32 npairs;        // s * (s - 1) / 2
    u32 use_lds;
    const u32* pair_off;
    const u32* ids;
    const u32* freqs;
    const signed char* bucket;  // per tuple, -1 = none
    u32* count;        // [nm][s][s]
    Fix128* mlog;
    Fix128* msqrt;
    Fix128* mlgamma;
    const double* nfactor;  // -N: 1 / dataset size per sample (null: raw frequencies)
};

__device__ __forceinline__ void dm_pair_of(u32 p, u32 s, u32& j, u32& k) {  // p-th pair (j < k) in row-major order
    // row j starts at j*s - j*(j+1)/2
    u32 jj = 0, start = 0;
    while (true) {
        const u32 len = s - 1 - jj;
        if (p < start + len) break;
        start += len;
        ++jj;
    }
    j = jj;
    k = jj + 1 + (p - start);
}

__global__ __launch_bounds__(DM_THREADS) void distmat_kernel(DmArgs a) {
    extern __shared__ unsigned char smem[];
    // layout: [groups per block][s] u32 freq, [groups][s] u8 present (padded), then optional accumulators
    const u32 groups = DM_THREADS / a.G;
    u32* s_freq = reinterpret_cast<u32*>(smem);
    u32* s_pres = s_freq + groups * a.s;
    const u32 cells = a.nm * a.s * a.s;
    Fix128* l_log = reinterpret_cast<Fix128*>(s_pres + groups * a.s);  // (groups is a power of two >= 4: 2 * groups * s words = a multiple of 8 bytes)
    Fix128* l_sqrt = l_log + (a.use_lds ? cells : 0);
    Fix128* l_lgam = l_sqrt + (a.use_lds ? cells : 0);
    u32* l_cnt = reinterpret_cast<u32*>(l_lgam + (a.use_lds ? cells : 0));
    if (a.use_lds)
        for (u32 c = threadIdx.x; c < cells; c += DM_THREADS) { l_log[c] = Fix128{0, 0}; l_sqrt[c] = Fix128{0, 0}; l_lgam[c] = Fix128{0, 0}; l_cnt[c] = 0; }
    const u32 g = threadIdx.x / a.G, lane = threadIdx.x % a.G;
    u32* fq = s_freq + g * a.s;
    u32* pr = s_pres + g * a.s;
    // the pair owned by this lane does not depend on the tuple
    u32 pj = 0, pk = 0;
    const bool has_pair = lane < a.npairs;
    if (has_pair) dm_pair_of(lane, a.s, pj, pk);
    __syncthreads();
    const u32 stride = gridDim.x * groups;
    const u32 rounds = (a.nt + stride - 1) / stride;
    for (u32 it = 0; it < rounds; ++it) {
        const u32 t = it * stride + blockIdx.x * groups + g;
        const bool valid = t < a.nt;
        const int b = valid ? (int)a.bucket[t] : -1;
        for (u32 x = lane; x < a.s; x += a.G) { fq[x] = 0; pr[x] = 0; }
        __syncthreads();
        if (b >= 0) {
            const u32 q0 = a.pair_off[t], q1 = a.pair_off[t + 1];
            for (u32 q = q0 + lane; q < q1; q += a.G) { const u32 id = a.ids[q]; fq[id] = a.freqs[q]; pr[id] = 1; }
        }
        __syncthreads();
        if (b >= 0) {
            const size_t base = (size_t)b * a.s * a.s;
            if (lane < a.s && pr[lane]) {  // diagonal count, add() :170-172 with j == k
                if (a.use_lds) atomicAdd(&l_cnt[base + (size_t)lane * a.s + lane], 1u);
                else atomicAdd(&a.count[base + (size_t)lane * a.s + lane], 1u);
            }
            for (u32 p = lane; p < a.npairs; p += a.G) {
                u32 j = pj, k = pk;
                if (p != lane) dm_pair_of(p, a.s, j, k);
                const u32 fj = fq[j], fk = fq[k];
                const size_t cell = base + (size_t)j * a.s + k;
                if (pr[j] && pr[k]) {
                    if (a.use_lds) atomicAdd(&l_cnt[cell], 1u);
                    else atomicAdd(&a.count[cell], 1u);
                }
                if (fj || fk) {  // add() :174-196
                    double dl, ds, lg = 0;
                    if (a.nfactor) {  // add_normalized(), smtxt2entropy.c:199-228: the lgamma term is disabled there
                        const double xj = (double)fj * a.nfactor[j], xk = (double)fk * a.nfactor[k];
                        dl = log(xj + 1.0) - log(xk + 1.0);
                        ds = sqrt(xj) - sqrt(xk);
                    } else {
                        dl = log((double)fj + 1.0) - log((double)fk + 1.0);
                        ds = sqrt((double)fj) - sqrt((double)fk);
                        lg = lgamma((double)fj + (double)fk + 1.0) - lgamma((double)fj + 1.0) - lgamma((double)fk + 1.0) - ((double)fj + (double)fk + 1.0);
                    }
                    if (a.use_lds) { fix_add(&l_log[cell], dl * dl); fix_add(&l_sqrt[cell], ds * ds); if (!a.nfactor) fix_add(&l_lgam[cell], lg); }
                    else { fix_add(&a.mlog[cell], dl * dl); fix_add(&a.msqrt[cell], ds * ds); if (!a.nfactor) fix_add(&a.mlgamma[cell], lg); }
                }
            }
        }
        __syncthreads();
    }
    if (a.use_lds) {
        for (u32 c = threadIdx.x; c < cells; c += DM_THREADS) {
            if (l_cnt[c]) atomicAdd(&a.count[c], l_cnt[c]);
            fix_merge(&a.mlog[c], l_log[c]);
            fix_merge(&a.msqrt[c], l_sqrt[c]);
            fix_merge(&a.mlgamma[c], l_lgam[c]);
        }
    }
}

}  // namespace dsm

using namespace dsm;

struct dsm_distmat {
    int device = 0;
    u32 s = 0, nm = 0, minfreq = 0;
    std::vector<double> maxent;       // sorted descending (smtxt2entropy.c:71-78, :542)
    std::vector<u32> noutput;         // per bucket, not yet cumulative
    u32 runs = 0;                     // ids accepted in the input (== s without a mapping)
    std::vector<int> run_to_sample;   // -S
    std::vector<double> nfactor;      // -N: 1 / size
    double* d_nfactor = nullptr;
    u32* d_count = nullptr;
    dsm::Fix128 *d_log = nullptr, *d_sqrt = nullptr, *d_lgamma = nullptr;  // 128-bit fixed-point sums (see Fix128)
    // upload staging (grown on demand)
    void *d_pair_off = nullptr, *d_ids = nullptr, *d_freqs = nullptr, *d_bucket = nullptr;
    size_t cap_t = 0, cap_p = 0;
    bool finished = false;
};

#define DM_HIP(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) return fail(DSM_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int dsm_distmat_steps(double step, double* out, int cap) {  // smtxt2entropy.c:258-285
    if (!(step > 0.0) || step > 1.0 || !out) return fail(DSM_E_INVAL, "entstep must be in (0, 1]");
    int n = (int)round(1 / step + 0.5);
    if ((n - 1) * step < 1.0) ++n;
    if (n > cap) return fail(DSM_E_INVAL, "entstep: too many steps");
    double sum = 0;
    int i = 0;
    while (i < n - 1) { out[i] = sum; sum += step; ++i; }
    out[i] = 1.0;
    return n;
}

int dsm_distmat_create(int device, uint32_t samples, const double* maxent, uint32_t nmaxent, uint32_t minfreq, dsm_distmat** out) {
    return dsm_distmat_create_ex(device, samples, maxent, nmaxent, minfreq, nullptr, 0, nullptr, out);
}

int dsm_distmat_create_ex(int device, uint32_t samples, const double* maxent, uint32_t nmaxent, uint32_t minfreq,
                          const int32_t* run_to_sample, uint32_t runs, const double* sizes, dsm_distmat** out) {
    if (!out || !maxent || nmaxent < 1 || nmaxent > 127) return fail(DSM_E_INVAL, "dsm_distmat_create: bad argument");
    if (samples < 2) return fail(DSM_E_INVAL, "the number of samples must be at least 2 (smtxt2entropy.c:560)");
    if (samples > 273) return fail(DSM_E_INVAL, "too many samples (MAX_READERS 273, metaserver.cpp:19)");
    for (u32 i = 0; i < nmaxent; ++i)
        if (!(maxent[i] >= 0.0 && maxent[i] <= 1.0)) return fail(DSM_E_INVAL, "maxent values must be between 0 and 1 (smtxt2entropy.c:300-305)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(DSM_E_NODEV, "no HIP device: the distance matrices run on the GPU only");
    if (device < 0 || device >= ndev) return fail(DSM_E_INVAL, "bad device");
    DM_HIP(hipSetDevice(device));
    auto* m = new dsm_distmat();
    m->device = device; m->s = samples; m->nm = nmaxent; m->minfreq = minfreq;
    m->maxent.assign(maxent, maxent + nmaxent);
    std::sort(m->maxent.begin(), m->maxent.end(), [](double x, double y) { return x > y; });
    m->noutput.assign(nmaxent, 0);
    m->runs = samples;
    if (run_to_sample) {
        if (runs < samples) { delete m; return fail(DSM_E_INVAL, "unable to use the run-to-sample mapping: fewer runs than samples (smtxt2entropy.c:571-572)"); }
        for (u32 i = 0; i < runs; ++i)
            if (run_to_sample[i] < 0 || (u32)run_to_sample[i] >= samples) { delete m; return fail(DSM_E_INVAL, "run-to-sample mapping: sample id out of range"); }
        m->runs = runs;
        m->run_to_sample.assign(run_to_sample, run_to_sample + runs);
    }
    if (sizes) {
        for (u32 i = 0; i < samples; ++i)
            if (!(sizes[i] != 0)) { delete m; return fail(DSM_E_INVAL, "dataset sizes must be non-zero (smtxt2entropy.c:593)"); }
        m->nfactor.resize(samples);
        for (u32 i = 0; i < samples; ++i) m->nfactor[i] = (double)1 / sizes[i];
    }
    const size_t cells = (size_t)nmaxent * samples * samples;
    if (hipMalloc(&m->d_count, cells * 4) != hipSuccess || hipMalloc(&m->d_log, cells * 16) != hipSuccess ||
        hipMalloc(&m->d_sqrt, cells * 16) != hipSuccess || hipMalloc(&m->d_lgamma, cells * 16) != hipSuccess) {
        dsm_distmat_destroy(m);
        return fail(DSM_E_NOMEM, "hipMalloc failed");
    }
    if (!m->nfactor.empty()) {
        if (hipMalloc(&m->d_nfactor, samples * 8) != hipSuccess) { dsm_distmat_destroy(m); return fail(DSM_E_NOMEM, "hipMalloc failed"); }
        (void)hipMemcpy(m->d_nfactor, m->nfactor.data(), samples * 8, hipMemcpyHostToDevice);
    }
    (void)hipMemset(m->d_count, 0, cells * 4); (void)hipMemset(m->d_log, 0, cells * 16);
    (void)hipMemset(m->d_sqrt, 0, cells * 16); (void)hipMemset(m->d_lgamma, 0, cells * 16);
    *out = m;
    return DSM_OK;
}

void dsm_distmat_destroy(dsm_distmat* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (void* p : {(void*)m->d_count, (void*)m->d_log, (void*)m->d_sqrt, (void*)m->d_lgamma, m->d_pair_off, m->d_ids, m->d_freqs, m->d_bucket, (void*)m->d_nfactor})
        if (p) (void)hipFree(p);
    delete m;
}

// host side of one batch: minfreq filter, duplicate ids (last one wins, :104-121), exact bucket per tuple
static int dm_accumulate(dsm_distmat* m, size_t nt, const u32* pair_off, const u32* ids, const u64* freqs) {
    if (m->finished) return fail(DSM_E_INVAL, "dsm_distmat: already finished");
    if (nt == 0) return DSM_OK;
    if (nt > 0xFFFFFFF0ull) return fail(DSM_E_INVAL, "dsm_distmat_add: batch too large");
    const u32 s = m->s, nm = m->nm;
    // host pass in parallel over tuple ranges: every worker cleans its tuples (minfreq, duplicates, sorted ids) into its own
    // vectors and picks the buckets; the pieces are concatenated afterwards
    unsigned nth = std::thread::hardware_concurrency();
    if (const char* e = getenv("DSM_HOST_THREADS")) nth = (unsigned)atoi(e);
    if (nth > 16) nth = 16;
    if (nth < 1 || nt < 65536) nth = 1;
    struct Piece { std::vector<u32> off, ids, fr; std::vector<u32> nout; int err = 0; };
    std::vector<Piece> pc(nth);
    std::vector<signed char> bucket(nt);
    const double LOG2 = log(2), LOGS = log((int)s);
    auto work = [&](unsigned w) {
        const size_t lo = nt * w / nth, hi = nt * (w + 1) / nth;
        Piece& P = pc[w];
        P.nout.assign(nm, 0);
        P.off.reserve(hi - lo + 1);
        P.ids.reserve(pair_off[hi] - pair_off[lo]);
        P.fr.reserve(pair_off[hi] - pair_off[lo]);
        std::vector<u32> fq(s, 0), seen;
        for (size_t t = lo; t < hi; ++t) {
            P.off.push_back((u32)P.ids.size());
            seen.clear();
            for (u32 q = pair_off[t]; q < pair_off[t + 1]; ++q) {
                u32 run = ids[q];
                const unsigned frq = (unsigned)freqs[q];  // the tool reads frequencies with atoi into unsigned
                if (run >= m->runs) { P.err = 1; return; }
                if (frq < m->minfreq) continue;
                if (!m->run_to_sample.empty()) run = (u32)m->run_to_sample[run];  // several runs of one sample: the last one wins
                bool dup = false;
                for (u32 x : seen) dup |= x == run;
                if (!dup) seen.push_back(run);
                fq[run] = frq;
            }
            std::sort(seen.begin(), seen.end());
            double entr;
            if (!m->nfactor.empty()) {  // normalized_entropy(), smtxt2entropy.c:147-165
                double sumN = (double)(int)s, sumNlogN = 0;
                for (u32 x : seen) {
                    const double frq = (double)fq[x] * m->nfactor[x];
                    sumN += frq;
                    sumNlogN += (frq + 1) * log(frq + 1) / LOG2;
                }
                const double entropy = (log(sumN) / LOG2 - sumNlogN / sumN);
                entr = LOG2 * entropy / LOGS;
            } else {  // entropy(), smtxt2entropy.c:128-145: unsigned 32-bit sumN, term-by-term log(x)/log(2)
                unsigned sumN = s;
                double sumNlogN = 0;
                for (u32 x : seen) {
                    const unsigned frq = fq[x];
                    sumN += frq;
                    sumNlogN += (double)(frq + 1) * log(frq + 1) / LOG2;
                }
                const double entropy = (log(sumN) / LOG2 - sumNlogN / (double)sumN);
                entr = LOG2 * entropy / LOGS;
            }
            int b = -1;
            for (int i = (int)nm; i > 0;) {
                --i;
                if (entr <= m->maxent[i]) { b = i; break; }
            }
            bucket[t] = (signed char)b;
            if (b >= 0) {
                ++P.nout[b];
                for (u32 x : seen) { P.ids.push_back(x); P.fr.push_back(fq[x]); }
            }
            for (u32 x : seen) fq[x] = 0;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned w = 1; w < nth; ++w) th.emplace_back(work, w);
        work(0);
        for (auto& t : th) t.join();
    }
    std::vector<u32> o_off(nt + 1), o_ids, o_fr;
    {
        size_t tot = 0;
        for (auto& P : pc) { if (P.err) return fail(DSM_E_INVAL, "dsm_distmat: run id out of range (smtxt2entropy.c:96-101)"); tot += P.ids.size(); }
        o_ids.reserve(tot); o_fr.reserve(tot);
        size_t t = 0;
        for (unsigned w = 0; w < nth; ++w) {
            const u32 base = (u32)o_ids.size();
            for (u32 v : pc[w].off) o_off[t++] = base + v;
            o_ids.insert(o_ids.end(), pc[w].ids.begin(), pc[w].ids.end());
            o_fr.insert(o_fr.end(), pc[w].fr.begin(), pc[w].fr.end());
            for (u32 i = 0; i < nm; ++i) m->noutput[i] += pc[w].nout[i];
        }
        o_off[nt] = (u32)o_ids.size();
    }
    DM_HIP(hipSetDevice(m->device));
    const size_t np = o_ids.size();
    if (nt + 1 > m->cap_t) {
        if (m->d_pair_off) (void)hipFree(m->d_pair_off);
        if (m->d_bucket) (void)hipFree(m->d_bucket);
        m->d_pair_off = m->d_bucket = nullptr;
        m->cap_t = (nt + 1) + (nt + 1) / 4;
        DM_HIP(hipMalloc(&m->d_pair_off, m->cap_t * 4));
        DM_HIP(hipMalloc(&m->d_bucket, m->cap_t));
    }
    if (np + 1 > m->cap_p) {
        if (m->d_ids) (void)hipFree(m->d_ids);
        if (m->d_freqs) (void)hipFree(m->d_freqs);
        m->d_ids = m->d_freqs = nullptr;
        m->cap_p = (np + 1) + (np + 1) / 4;
        DM_HIP(hipMalloc(&m->d_ids, m->cap_p * 4));
        DM_HIP(hipMalloc(&m->d_freqs, m->cap_p * 4));
    }
    DM_HIP(hipMemcpy(m->d_pair_off, o_off.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
    DM_HIP(hipMemcpy(m->d_bucket, bucket.data(), nt, hipMemcpyHostToDevice));
    if (np) {
        DM_HIP(hipMemcpy(m->d_ids, o_ids.data(), np * 4, hipMemcpyHostToDevice));
        DM_HIP(hipMemcpy(m->d_freqs, o_fr.data(), np * 4, hipMemcpyHostToDevice));
    }
    DmArgs a;
    memset(&a, 0, sizeof a);
    a.nt = (u32)nt; a.s = s; a.nm = nm;
    a.npairs = s * (s - 1) / 2;
    u32 need = a.npairs > s ? a.npairs : s, G = 1;
    while (G < need && G < 64) G <<= 1;
    a.G = G;
    const u32 cells = nm * s * s;
    a.use_lds = cells <= DM_LDS_CELLS ? 1u : 0u;
    a.pair_off = (const u32*)m->d_pair_off; a.ids = (const u32*)m->d_ids; a.freqs = (const u32*)m->d_freqs;
    a.bucket = (const signed char*)m->d_bucket;
    a.count = m->d_count; a.mlog = m->d_log; a.msqrt = m->d_sqrt; a.mlgamma = m->d_lgamma; a.nfactor = m->d_nfactor;
    const u32 groups = DM_THREADS / G;
    size_t shm = (size_t)groups * s * 8 + 24 + (a.use_lds ? (size_t)cells * 52 : 0);
    u64 blocks = (nt + groups - 1) / groups;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(distmat_kernel, dim3((unsigned)blocks), dim3(DM_THREADS), shm, 0, a);
    DM_HIP(hipGetLastError());
    DM_HIP(hipDeviceSynchronize());
    return DSM_OK;
}

int dsm_distmat_add(dsm_distmat* m, const dsm_tuple_batch* b) {
    if (!m || !b) return fail(DSM_E_INVAL, "dsm_distmat_add: null argument");
    return dm_accumulate(m, (size_t)b->ntuples, b->pair_off, b->ids, b->freqs);
}

int dsm_distmat_add_text(dsm_distmat* m, const char* text, size_t len) {
    if (!m || (!text && len)) return fail(DSM_E_INVAL, "dsm_distmat_add_text: null argument");
    std::vector<u32> off, ids;
    std::vector<u64> fr;
    off.push_back(0);
    size_t pos = 0;
    int parsep = -1;
    while (pos < len) {
        size_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        size_t p = pos;
        while (p < e && text[p] != ' ') ++p;
        if (p >= e) return fail(DSM_E_FORMAT, "tuple line without a space (smtxt2entropy.c:662-664)");
        if (parsep < 0) {  // the first row decides whether an entropy column is present (:666-674)
            parsep = 0;
            for (size_t t = p; t < e; ++t) if (text[t] == '.') { parsep = 1; break; }
        }
        if (parsep) { ++p; while (p < e && text[p] != ' ') ++p; }
        while (p < e) {
            while (p < e && text[p] == ' ') ++p;
            if (p >= e) break;
            const u32 run = (u32)atoi(std::string(text + p, std::min<size_t>(e - p, 24)).c_str());
            while (p < e && text[p] != ':') ++p;
            if (p >= e) return fail(DSM_E_FORMAT, "tuple line: id without ':' (smtxt2entropy.c:92-93)");
            const u64 f = (u64)(unsigned)atoi(std::string(text + p + 1, std::min<size_t>(e - p - 1, 24)).c_str());
            while (p < e && text[p] != ' ') ++p;
            ids.push_back(run);
            fr.push_back(f);
        }
        off.push_back((u32)ids.size());
        pos = e + 1;
    }
    return dm_accumulate(m, off.size() - 1, off.data(), ids.data(), fr.data());
}

int dsm_distmat_finish(dsm_distmat* m, double* maxent_sorted, uint32_t* noutput, uint32_t* count, double* mlog, double* msqrt,
                       double* mlgamma) {
    if (!m) return fail(DSM_E_INVAL, "dsm_distmat_finish: null argument");
    if (m->finished) return fail(DSM_E_INVAL, "dsm_distmat: already finished");
    m->finished = true;
    DM_HIP(hipSetDevice(m->device));
    const size_t per = (size_t)m->s * m->s, cells = per * m->nm;
    std::vector<u32> c(cells);
    std::vector<double> l(cells), q(cells), g(cells);
    DM_HIP(hipMemcpy(c.data(), m->d_count, cells * 4, hipMemcpyDeviceToHost));
    {
        std::vector<Fix128> f(cells);
        DM_HIP(hipMemcpy(f.data(), m->d_log, cells * 16, hipMemcpyDeviceToHost));
        for (size_t x = 0; x < cells; ++x) l[x] = fix_to_double(f[x]);
        DM_HIP(hipMemcpy(f.data(), m->d_sqrt, cells * 16, hipMemcpyDeviceToHost));
        for (size_t x = 0; x < cells; ++x) q[x] = fix_to_double(f[x]);
        DM_HIP(hipMemcpy(f.data(), m->d_lgamma, cells * 16, hipMemcpyDeviceToHost));
        for (size_t x = 0; x < cells; ++x) g[x] = fix_to_double(f[x]);
    }
    std::vector<u32> nout = m->noutput;
    for (u32 i = m->nm; i > 1;) {  // accumulate(i -> i-1), smtxt2entropy.c:230-242, in the print loop's order (:746-750)
        --i;
        nout[i - 1] += nout[i];
        for (size_t x = 0; x < per; ++x) {
            c[(i - 1) * per + x] += c[i * per + x];
            l[(i - 1) * per + x] += l[i * per + x];
            q[(i - 1) * per + x] += q[i * per + x];
            g[(i - 1) * per + x] += g[i * per + x];
        }
    }
    if (maxent_sorted) memcpy(maxent_sorted, m->maxent.data(), m->nm * 8);
    if (noutput) memcpy(noutput, nout.data(), m->nm * 4);
    if (count) memcpy(count, c.data(), cells * 4);
    if (mlog) memcpy(mlog, l.data(), cells * 8);
    if (msqrt) memcpy(msqrt, q.data(), cells * 8);
    if (mlgamma) memcpy(mlgamma, g.data(), cells * 8);
    return DSM_OK;
}

int dsm_distmat_format(uint32_t s, uint32_t nm, const double* maxent_sorted, const uint32_t* noutput, const uint32_t* count,
                       const double* mlog, const double* msqrt, const double* mlgamma, char* text[4]) {
    if (!maxent_sorted || !noutput || !count || !mlog || !msqrt || !mlgamma || !text) return fail(DSM_E_INVAL, "dsm_distmat_format: null argument");
    std::string out[4];
    char tmp[256];
    for (u32 i = nm; i > 0;) {  // smtxt2entropy.c:722-744
        --i;
        snprintf(tmp, sizeof tmp, "Matrix for <max_entropy>=<%f> was computed from %u substrings: \n", maxent_sorted[i], noutput[i]);
        for (auto& o : out) o += tmp;
        for (u32 j = 0; j < s; ++j) {
            for (u32 k = 0; k < s; ++k) {
                const size_t x = ((size_t)i * s + j) * s + k;
                snprintf(tmp, sizeof tmp, " %u", count[x]); out[0] += tmp;
                snprintf(tmp, sizeof tmp, " %f", mlog[x]); out[1] += tmp;
                snprintf(tmp, sizeof tmp, " %f", msqrt[x]); out[2] += tmp;
                snprintf(tmp, sizeof tmp, " %f", mlgamma[x]); out[3] += tmp;
            }
            for (auto& o : out) o += "\n";
        }
    }
    for (int f = 0; f < 4; ++f) {
        text[f] = (char*)malloc(out[f].size() + 1);
        if (!text[f]) return fail(DSM_E_NOMEM, "malloc failed");
        memcpy(text[f], out[f].c_str(), out[f].size() + 1);
    }
    return DSM_OK;
}

}  // extern "C"
