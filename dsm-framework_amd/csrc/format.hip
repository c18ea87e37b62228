// format.hip -- metaserver's output text (metaserver.cpp:472-484: "path %f id:freq ...\n" per printed node) from tuple batches.
//
// dsm_format_batch        the reference's loop as it is: snprintf on the host, several threads.  The checker of the device path.
// dsm_formatter_* / dsm_format_batch_dev   the same bytes from the GPU (round 4): one thread per tuple computes its line's length
//                         -- which needs the rounded entropy already: 9.9999996 prints as "10.000000" -- a scan places the lines,
//                         one thread per tuple writes its line.  "%f" is printf's: the exact binary value of the double, rounded to six
//                         decimals, ties to even (glibc rounds the exact decimal expansion in the current rounding mode).  Here: the
//                         53-bit significand times 10^6 as a 128-bit integer, shifted right by the binary exponent, with the
//                         remainder compared against one half.  No floating-point operation touches the value, so there is nothing
//                         that could round differently from the host.  Doubles of 2^40 or more, infinities and NaNs (never an
//                         entropy: that is at most log2 of 273 samples) send the batch to the host formatter.
#include <sched.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "scan.h"
#include "fmt6.h"
#include "entropy_tables.h"
#include "textemit.h"

namespace dsm {

static unsigned fmt_host_threads() {
    static unsigned n = 0;
    if (!n) {
        const char* e = getenv("DSM_HOST_THREADS");
        long v = e ? atol(e) : 0;
        if (v <= 0) {
            cpu_set_t cs;
            v = sched_getaffinity(0, sizeof cs, &cs) == 0 ? CPU_COUNT(&cs) : 1;
            if (v > 16) v = 16;
        }
        n = (unsigned)(v < 1 ? 1 : v);
    }
    return n;
}

// length of a tuple's line; 0 marks a value the device does not print (the batch then goes to the host)
__global__ __launch_bounds__(256) void fmt_len_kernel(u64 nt, const u32* __restrict__ path_off, const double* __restrict__ ent, const u32* __restrict__ pair_off,
                                                      const u32* __restrict__ ids, const u64* __restrict__ freqs, u32* __restrict__ len, u32* __restrict__ bad) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nt) return;
    bool neg, ok;
    const u64 v = fixed6_of(ent[r], &neg, &ok);
    if (!ok) { len[r] = 0; atomicOr(bad, 1u); return; }
    u32 n = (path_off[r + 1] - path_off[r]) + 1u + (neg ? 1u : 0u) + dec_digits(v / 1000000ull) + 7u;  // path, ' ', sign, integer digits, '.', six decimals
    for (u32 q = pair_off[r]; q < pair_off[r + 1]; ++q) n += 2u + dec_digits(ids[q]) + dec_digits(freqs[q]);  // " id:freq"
    len[r] = n + 1u;  // '\n'
}
__global__ __launch_bounds__(256) void fmt_write_kernel(u64 nt, const u32* __restrict__ path_off, const char* __restrict__ paths, const double* __restrict__ ent,
                                                        const u32* __restrict__ pair_off, const u32* __restrict__ ids, const u64* __restrict__ freqs,
                                                        const u64* __restrict__ off, char* __restrict__ out) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nt) return;
    char* p = out + off[r];
    const u32 pb = path_off[r], pl = path_off[r + 1] - pb;
    for (u32 k = 0; k < pl; ++k) p[k] = paths[pb + k];
    p += pl;
    *p++ = ' ';
    bool neg, ok;
    const u64 v = fixed6_of(ent[r], &neg, &ok);
    if (neg) *p++ = '-';
    const u64 ip = v / 1000000ull;
    u32 fr = (u32)(v % 1000000ull);
    p += put_dec(p, ip, dec_digits(ip));
    *p++ = '.';
    for (int k = 5; k >= 0; --k) { p[k] = (char)('0' + (int)(fr % 10u)); fr /= 10u; }
    p += 6;
    for (u32 q = pair_off[r]; q < pair_off[r + 1]; ++q) {
        *p++ = ' ';
        const u64 id = ids[q], f = freqs[q];
        p += put_dec(p, id, dec_digits(id));
        *p++ = ':';
        p += put_dec(p, f, dec_digits(f));
    }
    *p = '\n';
}

struct GrowDev {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) return fail(DSM_E_NOMEM, "dsm_formatter: hipMalloc failed");
        cap = want;
        return 0;
    }
    ~GrowDev() { if (p) (void)hipFree(p); }
};
struct GrowPin {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 4096;
        if (hipHostMalloc(&p, want) != hipSuccess) return fail(DSM_E_NOMEM, "dsm_formatter: hipHostMalloc failed");
        cap = want;
        return 0;
    }
    ~GrowPin() { if (p) (void)hipHostFree(p); }
};


// ---- text mode of the emitter (textemit.h) ---------------------------------------------------------------------------------
// entropy of tuple r exactly as emit_job computes it on the host: the same table entries added in the same order, one division, one
// subtraction (IEEE double, no contraction).  A frequency or a total outside the tables makes the tuple an exception: the host
// computes those with libm (a few hundred nodes at the top of a pass) and writes them back.
constexpr u8 TE_DROP = 0, TE_KEEP = 1, TE_HOST = 2;
__global__ __launch_bounds__(256) void te_entropy_kernel(u32 t0, u32 t1, const u32* __restrict__ pair_off, const u64* __restrict__ freqs, u32 d, double emin,
                                                         double emax, const double* __restrict__ terms, const double* __restrict__ logn, double* __restrict__ ent,
                                                         u8* __restrict__ keep, u32* __restrict__ nexc, u32* __restrict__ exc, u32 exc_cap) {
    const u32 r = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t1) return;
    u64 sumN = d;
    double s = 0;
    bool host = false;
    for (u32 q = pair_off[r]; q < pair_off[r + 1]; ++q) {
        const u64 f = freqs[q];
        sumN += f;
        if (f >= TERM_TAB) { host = true; break; }
        s += terms[f];
    }
    if (sumN >= LOGN_TAB) host = true;
    if (host) {
        const u32 k = atomicAdd(nexc, 1u);
        if (k < exc_cap) exc[k] = r;
        keep[r - t0] = TE_HOST;
        return;
    }
    const double e = logn[sumN] - s / (double)sumN;
    ent[r - t0] = e;
    keep[r - t0] = (emax > 0 && (e < emin || e > emax)) ? TE_DROP : TE_KEEP;
}
// the frequencies of the exceptions, for the host: row i = count, then the frequencies in order
constexpr u32 TE_ROW = 280;  // >= 1 + MAX_READERS (273)
__global__ void te_gather_kernel(u32 n, const u32* __restrict__ exc, const u32* __restrict__ pair_off, const u64* __restrict__ freqs, u64* __restrict__ rows) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = exc[i], q0 = pair_off[r], q1 = pair_off[r + 1];
    u32 c = q1 - q0;
    if (c > TE_ROW - 1) c = TE_ROW - 1;
    rows[(size_t)i * TE_ROW] = c;
    for (u32 k = 0; k < c; ++k) rows[(size_t)i * TE_ROW + 1 + k] = freqs[q0 + k];
}
__global__ void te_scatter_kernel(u32 n, u32 t0, const u32* __restrict__ exc, const double* __restrict__ val, double emin, double emax, double* __restrict__ ent,
                                  u8* __restrict__ keep) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = exc[i];
    const double e = val[i];
    ent[r - t0] = e;
    keep[r - t0] = (emax > 0 && (e < emin || e > emax)) ? TE_DROP : TE_KEEP;
}
// length of a kept tuple's line (0: dropped); the kept tuples and pairs are counted
__global__ __launch_bounds__(256) void te_len_kernel(u32 t0, u32 t1, const u32* __restrict__ path_off, const u32* __restrict__ pair_off, const u32* __restrict__ ids,
                                                     const u64* __restrict__ freqs, const double* __restrict__ ent, const u8* __restrict__ keep,
                                                     u32* __restrict__ len, unsigned long long* __restrict__ counts, u32* __restrict__ bad) {
    const u32 r = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t1) return;
    if (keep[r - t0] != TE_KEEP) { len[r - t0] = 0; return; }
    bool neg, ok;
    const u64 v = fixed6_of(ent[r - t0], &neg, &ok);
    if (!ok) { atomicOr(bad, 1u); len[r - t0] = 0; return; }  // (an entropy is at most log2(273): cannot happen)
    u32 n = (path_off[r + 1] - path_off[r]) + 1u + (neg ? 1u : 0u) + dec_digits(v / 1000000ull) + 7u;
    for (u32 q = pair_off[r]; q < pair_off[r + 1]; ++q) n += 2u + dec_digits(ids[q]) + dec_digits(freqs[q]);
    len[r - t0] = n + 1u;
    atomicAdd(counts, 1ull);
    atomicAdd(counts + 1, (unsigned long long)(pair_off[r + 1] - pair_off[r]));
}
__global__ __launch_bounds__(256) void te_write_kernel(u32 t0, u32 t1, const u32* __restrict__ path_off, const char* __restrict__ paths, const u32* __restrict__ pair_off,
                                                       const u32* __restrict__ ids, const u64* __restrict__ freqs, const double* __restrict__ ent,
                                                       const u32* __restrict__ len, const u64* __restrict__ off, char* __restrict__ out) {
    const u32 r = t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= t1 || len[r - t0] == 0) return;
    char* p = out + off[r - t0];
    const u32 pb = path_off[r], pl = path_off[r + 1] - pb;
    for (u32 k = 0; k < pl; ++k) p[k] = paths[pb + k];
    p += pl;
    *p++ = ' ';
    bool neg, ok;
    const u64 v = fixed6_of(ent[r - t0], &neg, &ok);
    if (neg) *p++ = '-';
    const u64 ip = v / 1000000ull;
    u32 fr = (u32)(v % 1000000ull);
    p += put_dec(p, ip, dec_digits(ip));
    *p++ = '.';
    for (int k = 5; k >= 0; --k) { p[k] = (char)('0' + (int)(fr % 10u)); fr /= 10u; }
    p += 6;
    for (u32 q = pair_off[r]; q < pair_off[r + 1]; ++q) {
        *p++ = ' ';
        const u64 id = ids[q], f = freqs[q];
        p += put_dec(p, id, dec_digits(id));
        *p++ = ':';
        p += put_dec(p, f, dec_digits(f));
    }
    *p = '\n';
}

struct TextEmit {
    int device = 0;
    hipStream_t st = nullptr;
    double* d_terms = nullptr;
    double* d_logn = nullptr;
    GrowDev ent, keep, len, off, tmp, misc, exc, rows, val, out;
    GrowPin h_out, h_rows, h_val;
    ~TextEmit() {
        if (d_terms) (void)hipFree(d_terms);
        if (d_logn) (void)hipFree(d_logn);
        if (st) (void)hipStreamDestroy(st);
    }
};
TextEmit* text_emit_create(int device) {
    if (hipSetDevice(device) != hipSuccess) { (void)fail(DSM_E_HIP, "text emitter: hipSetDevice failed"); return nullptr; }
    TextEmit* t = new TextEmit();
    t->device = device;
    bool ok = hipStreamCreateWithFlags(&t->st, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc((void**)&t->d_terms, (size_t)TERM_TAB * 8) == hipSuccess && hipMalloc((void**)&t->d_logn, (size_t)LOGN_TAB * 8) == hipSuccess;
    ok = ok && hipMemcpy(t->d_terms, term_table(), (size_t)TERM_TAB * 8, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(t->d_logn, logn_table(), (size_t)LOGN_TAB * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { (void)fail(DSM_E_NOMEM, "text emitter: cannot create the device tables"); delete t; return nullptr; }
    return t;
}
void text_emit_destroy(TextEmit* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    delete t;
}

#define TE_HIP(x)                                                                     \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) return fail(DSM_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

int text_emit_chunk(TextEmit* t, u32 t0, u32 t1, const u32* path_off, const u32* pair_off, const u32* ids, const u64* freqs, const char* paths, u32 d,
                    double emin, double emax, const char** text, size_t* len, u64* kept_tuples, u64* kept_pairs) {
    *text = "";
    *len = 0;
    if (t1 <= t0) return 0;
    const u32 nt = t1 - t0;
    TE_HIP(hipSetDevice(t->device));
    hipStream_t st = t->st;
    const u32 exc_cap = 1u << 16;
    if (int rc = t->ent.ensure((size_t)nt * 8)) return rc;
    if (int rc = t->keep.ensure(nt)) return rc;
    if (int rc = t->len.ensure((size_t)nt * 4)) return rc;
    if (int rc = t->off.ensure(((size_t)nt + 1) * 8)) return rc;
    if (int rc = t->tmp.ensure((scan_tmp_elems(nt) + 8) * 8)) return rc;
    if (int rc = t->misc.ensure(64)) return rc;
    if (int rc = t->exc.ensure((size_t)exc_cap * 4)) return rc;
    // misc: [0] exceptions, [1] bad value flag, [2..3] pad, then u64 [2] total bytes, [3] kept tuples, [4] kept pairs
    u32* d_nexc = (u32*)t->misc.p;
    u32* d_bad = d_nexc + 1;
    unsigned long long* d_total = (unsigned long long*)t->misc.p + 2;
    unsigned long long* d_counts = d_total + 1;
    TE_HIP(hipMemsetAsync(t->misc.p, 0, 64, st));
    const dim3 grid((nt + 255) / 256);
    hipLaunchKernelGGL(te_entropy_kernel, grid, dim3(256), 0, st, t0, t1, pair_off, freqs, d, emin, emax, (const double*)t->d_terms, (const double*)t->d_logn,
                       (double*)t->ent.p, (u8*)t->keep.p, d_nexc, (u32*)t->exc.p, exc_cap);
    u32 h4[4] = {0, 0, 0, 0};
    TE_HIP(hipMemcpyAsync(h4, t->misc.p, 16, hipMemcpyDeviceToHost, st));
    TE_HIP(hipStreamSynchronize(st));
    const u32 nexc = h4[0];
    if (nexc > exc_cap) return fail(DSM_E_CAPACITY, "text emitter: more than 65536 tuples of a chunk have frequencies beyond the entropy tables");
    if (nexc) {  // frequencies of 65536 and more, or a total of 2^20 and more: the host's libm, as emit_job does for the binary batches
        if (int rc = t->rows.ensure((size_t)nexc * TE_ROW * 8)) return rc;
        if (int rc = t->val.ensure((size_t)nexc * 8)) return rc;
        if (int rc = t->h_rows.ensure((size_t)nexc * TE_ROW * 8)) return rc;
        if (int rc = t->h_val.ensure((size_t)nexc * 8)) return rc;
        hipLaunchKernelGGL(te_gather_kernel, dim3((nexc + 255) / 256), dim3(256), 0, st, nexc, (const u32*)t->exc.p, pair_off, freqs, (u64*)t->rows.p);
        TE_HIP(hipMemcpyAsync(t->h_rows.p, t->rows.p, (size_t)nexc * TE_ROW * 8, hipMemcpyDeviceToHost, st));
        TE_HIP(hipStreamSynchronize(st));
        const u64* rows = (const u64*)t->h_rows.p;
        double* val = (double*)t->h_val.p;
        for (u32 i = 0; i < nexc; ++i) val[i] = exact_entropy(d, rows + (size_t)i * TE_ROW + 1, (u32)rows[(size_t)i * TE_ROW]);
        TE_HIP(hipMemcpyAsync(t->val.p, val, (size_t)nexc * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(te_scatter_kernel, dim3((nexc + 255) / 256), dim3(256), 0, st, nexc, t0, (const u32*)t->exc.p, (const double*)t->val.p, emin, emax,
                           (double*)t->ent.p, (u8*)t->keep.p);
    }
    hipLaunchKernelGGL(te_len_kernel, grid, dim3(256), 0, st, t0, t1, path_off, pair_off, ids, freqs, (const double*)t->ent.p, (const u8*)t->keep.p, (u32*)t->len.p,
                       d_counts, d_bad);
    exclusive_scan<u32, u64>((const u32*)t->len.p, (u64*)t->off.p, nt, (u64*)t->tmp.p, (u64*)d_total, st);
    u64 h8[8];
    TE_HIP(hipMemcpyAsync(h8, t->misc.p, 64, hipMemcpyDeviceToHost, st));
    TE_HIP(hipStreamSynchronize(st));
    if (((const u32*)h8)[1]) return fail(DSM_E_HIP, "text emitter: an entropy outside the printable range");
    const u64 total = h8[2];
    *kept_tuples += h8[3];
    *kept_pairs += h8[4];
    if (!total) return 0;
    if (int rc = t->out.ensure(total + 1)) return rc;
    if (int rc = t->h_out.ensure(total + 1)) return rc;
    hipLaunchKernelGGL(te_write_kernel, grid, dim3(256), 0, st, t0, t1, path_off, paths, pair_off, ids, freqs, (const double*)t->ent.p, (const u32*)t->len.p,
                       (const u64*)t->off.p, (char*)t->out.p);
    TE_HIP(hipGetLastError());
    TE_HIP(hipMemcpyAsync(t->h_out.p, t->out.p, total, hipMemcpyDeviceToHost, st));
    TE_HIP(hipStreamSynchronize(st));
    *text = (const char*)t->h_out.p;
    *len = total;
    return 0;
}

}  // namespace dsm

using namespace dsm;

struct dsm_formatter {
    int device = 0;
    hipStream_t st = nullptr;
    GrowDev d_path_off, d_paths, d_ent, d_pair_off, d_ids, d_freqs, d_len, d_off, d_tmp, d_out, d_misc;
    GrowPin h_out;
    char* fallback = nullptr;  // text of a batch the host formatted (values outside the device's range)
    ~dsm_formatter() {
        if (fallback) free(fallback);
        if (st) (void)hipStreamDestroy(st);
    }
};

#define FMT_HIP(x)                                                                    \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) return fail(DSM_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int dsm_format_batch(const dsm_tuple_batch* b, char** text, size_t* len) {  // metaserver.cpp:472-484
    if (!b || !text || !len) return fail(DSM_E_INVAL, "dsm_format_batch: null argument");
    // printf's "%f" per tuple is the expensive part (one thread: 6 M lines/s, the mining delivers 100 M tuples/s): the tuples
    // are formatted in ranges by several threads, each into its own worst-case window of the output, then moved together.
    const u64 nt = b->ntuples;
    unsigned nth = fmt_host_threads();
    if (nt < 65536) nth = 1;
    std::vector<size_t> cap(nth + 1, 0), used(nth, 0);
    auto range = [&](unsigned t, u64& lo, u64& hi) { lo = nt * t / nth; hi = nt * (t + 1) / nth; };
    for (unsigned t = 0; t < nth; ++t) {
        u64 lo, hi;
        range(t, lo, hi);
        size_t c = 0;
        if (hi > lo) c = (size_t)(b->path_off[hi] - b->path_off[lo]) + (size_t)(hi - lo) * 48 + (size_t)(b->pair_off[hi] - b->pair_off[lo]) * 34;
        cap[t + 1] = cap[t] + c;
    }
    char* out = (char*)malloc(cap[nth] + 64);
    if (!out) return fail(DSM_E_NOMEM, "malloc failed");
    std::atomic<int> bad{0};  // a caller-made batch whose numbers do not fit the windows (an entropy of 1e300 prints 300 digits)
    auto work = [&](unsigned t) {
        u64 lo, hi;
        range(t, lo, hi);
        char* o = out + cap[t];
        const size_t room = cap[t + 1] - cap[t] + (t + 1 == nth ? 64 : 0);
        size_t w = 0;
        for (u64 r = lo; r < hi && !bad; ++r) {
            size_t pl = b->path_off[r + 1] - b->path_off[r];
            memcpy(o + w, b->path_bytes + b->path_off[r], pl);
            w += pl;
            int k = snprintf(o + w, room - w, " %f", b->entropy[r]);
            if (k < 0 || (size_t)k >= room - w) { bad = 1; break; }
            w += (size_t)k;
            for (u32 q = b->pair_off[r]; q < b->pair_off[r + 1] && !bad; ++q) {
                k = snprintf(o + w, room - w, " %d:%lu", (int)b->ids[q], (unsigned long)b->freqs[q]);
                if (k < 0 || (size_t)k + 1 >= room - w) { bad = 1; break; }
                w += (size_t)k;
            }
            if (bad) break;
            o[w++] = '\n';
        }
        used[t] = w;
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nth; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
    }
    if (bad) { free(out); return fail(DSM_E_INVAL, "dsm_format_batch: a value does not fit its text window (entropy out of range?)"); }
    size_t w = used[0];
    for (unsigned t = 1; t < nth; ++t) {  // close the gaps between the windows
        memmove(out + w, out + cap[t], used[t]);
        w += used[t];
    }
    out[w] = 0;
    *text = out;
    *len = w;
    return DSM_OK;
}

int dsm_formatter_create(int device, dsm_formatter** out) {
    if (!out) return fail(DSM_E_INVAL, "dsm_formatter_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSM_E_NODEV, "dsm_formatter_create: no HIP device");
    if (device < 0 || device >= ndev) return fail(DSM_E_NODEV, "dsm_formatter_create: bad device ordinal");
    FMT_HIP(hipSetDevice(device));
    dsm_formatter* f = new dsm_formatter();
    f->device = device;
    if (hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking) != hipSuccess) { delete f; return fail(DSM_E_HIP, "hipStreamCreate failed"); }
    *out = f;
    return DSM_OK;
}
void dsm_formatter_destroy(dsm_formatter* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    delete f;
}

int dsm_formatter_format(dsm_formatter* f, const dsm_tuple_batch* b, const char** text, size_t* len) {
    if (!f || !b || !text || !len) return fail(DSM_E_INVAL, "dsm_formatter_format: null argument");
    *text = "";
    *len = 0;
    const u64 nt = b->ntuples;
    if (nt == 0) return DSM_OK;
    if (nt > 0xFFFFFFF0ull) return fail(DSM_E_INVAL, "dsm_formatter_format: batch too large");
    FMT_HIP(hipSetDevice(f->device));
    const size_t pbytes = b->path_off[nt], npairs = b->pair_off[nt];
    if (int rc = f->d_path_off.ensure((nt + 1) * 4)) return rc;
    if (int rc = f->d_pair_off.ensure((nt + 1) * 4)) return rc;
    if (int rc = f->d_paths.ensure(pbytes + 1)) return rc;
    if (int rc = f->d_ent.ensure(nt * 8)) return rc;
    if (int rc = f->d_ids.ensure(npairs * 4 + 4)) return rc;
    if (int rc = f->d_freqs.ensure(npairs * 8 + 8)) return rc;
    if (int rc = f->d_len.ensure(nt * 4)) return rc;
    if (int rc = f->d_off.ensure((nt + 1) * 8)) return rc;
    if (int rc = f->d_tmp.ensure((scan_tmp_elems(nt) + 8) * 8)) return rc;
    if (int rc = f->d_misc.ensure(64)) return rc;
    hipStream_t st = f->st;
    FMT_HIP(hipMemcpyAsync(f->d_path_off.p, b->path_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
    FMT_HIP(hipMemcpyAsync(f->d_pair_off.p, b->pair_off, (nt + 1) * 4, hipMemcpyHostToDevice, st));
    FMT_HIP(hipMemcpyAsync(f->d_ent.p, b->entropy, nt * 8, hipMemcpyHostToDevice, st));
    if (pbytes) FMT_HIP(hipMemcpyAsync(f->d_paths.p, b->path_bytes, pbytes, hipMemcpyHostToDevice, st));
    if (npairs) {
        FMT_HIP(hipMemcpyAsync(f->d_ids.p, b->ids, npairs * 4, hipMemcpyHostToDevice, st));
        FMT_HIP(hipMemcpyAsync(f->d_freqs.p, b->freqs, npairs * 8, hipMemcpyHostToDevice, st));
    }
    u32* d_bad = (u32*)f->d_misc.p;
    u64* d_total = (u64*)f->d_misc.p + 1;
    FMT_HIP(hipMemsetAsync(d_bad, 0, 16, st));
    const dim3 grid((unsigned)((nt + 255) / 256));
    hipLaunchKernelGGL(fmt_len_kernel, grid, dim3(256), 0, st, nt, (const u32*)f->d_path_off.p, (const double*)f->d_ent.p, (const u32*)f->d_pair_off.p,
                       (const u32*)f->d_ids.p, (const u64*)f->d_freqs.p, (u32*)f->d_len.p, d_bad);
    exclusive_scan<u32, u64>((const u32*)f->d_len.p, (u64*)f->d_off.p, nt, (u64*)f->d_tmp.p, d_total, st);
    u64 h[2] = {0, 0};
    FMT_HIP(hipMemcpyAsync(h, f->d_misc.p, 16, hipMemcpyDeviceToHost, st));
    FMT_HIP(hipStreamSynchronize(st));
    if ((u32)h[0]) {  // a value the device does not print: the reference's own loop on the host
        if (f->fallback) { free(f->fallback); f->fallback = nullptr; }
        size_t n = 0;
        if (int rc = dsm_format_batch(b, &f->fallback, &n)) return rc;
        *text = f->fallback;
        *len = n;
        return DSM_OK;
    }
    const u64 total = h[1];
    if (int rc = f->d_out.ensure(total + 1)) return rc;
    if (int rc = f->h_out.ensure(total + 1)) return rc;
    hipLaunchKernelGGL(fmt_write_kernel, grid, dim3(256), 0, st, nt, (const u32*)f->d_path_off.p, (const char*)f->d_paths.p, (const double*)f->d_ent.p,
                       (const u32*)f->d_pair_off.p, (const u32*)f->d_ids.p, (const u64*)f->d_freqs.p, (const u64*)f->d_off.p, (char*)f->d_out.p);
    FMT_HIP(hipGetLastError());
    FMT_HIP(hipMemcpyAsync(f->h_out.p, f->d_out.p, total, hipMemcpyDeviceToHost, st));
    FMT_HIP(hipStreamSynchronize(st));
    ((char*)f->h_out.p)[total] = 0;
    *text = (const char*)f->h_out.p;
    *len = total;
    return DSM_OK;
}

int dsm_format_batch_dev(const dsm_tuple_batch* b, int device, char** text, size_t* len) {
    if (!b || !text || !len) return fail(DSM_E_INVAL, "dsm_format_batch_dev: null argument");
    dsm_formatter* f = nullptr;
    if (int rc = dsm_formatter_create(device, &f)) return rc;
    const char* t = nullptr;
    size_t n = 0;
    int rc = dsm_formatter_format(f, b, &t, &n);
    if (!rc) {
        char* o = (char*)malloc(n + 1);
        if (!o) rc = fail(DSM_E_NOMEM, "malloc failed");
        else { memcpy(o, t, n); o[n] = 0; *text = o; *len = n; }
    }
    dsm_formatter_destroy(f);
    return rc;
}

}  // extern "C"
