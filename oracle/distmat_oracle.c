/* distmat_oracle.c -- CPU restatement of the reference's distance-matrix accumulation:
 * wrapper-distance-matrix/smtxt2entropy.c.  TEST INFRASTRUCTURE ONLY: used by tests/ to check the GPU path
 * (dsm_distmat_*); nothing in the product links or calls it.  Parity PINNED: tests/test_distmat.py compares
 * its text output byte for byte with tests/golden/<set>/distmat.* produced by the unmodified reference tool
 * (oracle/_ref/smtxt2entropy, tests/golden/make_golden_distmat.py).
 *
 * Restated: the default mode, the -S run-to-sample mapping (:101-104, 385-423) and the -N normalisation
 * (normalized_entropy :147-165, add_normalized :199-228, factors and tables :585-611; its lgamma matrix stays 0).
 *   line parsing            smtxt2entropy.c:84-125, 656-680   (first token = path, second = entropy text when it
 *                                                             contains '.', then id:freq pairs; -M drops pairs)
 *   normalised entropy      smtxt2entropy.c:128-145           (unsigned 32-bit sumN, log(x)/log(2) term by term)
 *   bucket choice           smtxt2entropy.c:542 (qsort descending), 690-703 (smallest maxent >= entropy)
 *   add()                   smtxt2entropy.c:167-197           (tables for freq < 100000, direct calls above)
 *   accumulate + print      smtxt2entropy.c:230-242, 722-752  ("%f", cumulative from the smallest bucket up)
 *   -e steps                smtxt2entropy.c:258-285
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PRECMP 100000
#define MAXS 512

static double prelog[PRECMP], presqrt[PRECMP], prelgamma[PRECMP];
static int tables_ready = 0;
static void tables(void) {
    if (tables_ready) return;
    for (int i = 0; i < PRECMP; ++i) { prelog[i] = log(i + 1); presqrt[i] = sqrt(i); prelgamma[i] = lgamma(i + 1); }
    tables_ready = 1;
}

typedef struct { unsigned count; double lg, sq, lgam; } cell;
typedef struct { double maxent; unsigned noutput; } param;

static int paramcmp(const void* a, const void* b) {  /* smtxt2entropy.c:71-78: descending, never 0 */
    const param* x = (const param*)a; const param* y = (const param*)b;
    return x->maxent < y->maxent ? +1 : -1;
}
static int ucmp(const void* a, const void* b) { return (int)(*(const unsigned*)a - *(const unsigned*)b); }

/* -e step list, smtxt2entropy.c:258-285 */
int orc_distmat_steps(double step, double* out, int cap) {
    int n = (int)round(1 / step + 0.5);
    if ((n - 1) * step < 1.0) ++n;
    if (n > cap) return -1;
    double sum = 0;
    int i = 0;
    while (i < n - 1) { out[i] = sum; sum += step; ++i; }
    out[i] = 1.0;
    return n;
}

typedef struct {
    int smpls, nm;
    param* par;      /* sorted descending */
    cell* m;         /* [nm][smpls][smpls] */
    unsigned minfreq;
    unsigned long rows;
    int runs;        /* ids accepted in the input; == smpls without -S */
    int* runtosmpl;  /* NULL without -S */
    double* nfactor; /* NULL without -N: 1 / dataset size per sample */
    double** prenormlog;
    double** prenormsqrt;
} dm;

void* orc_distmat_new(int smpls, const double* maxent, int nmaxent, unsigned minfreq) {
    tables();
    if (smpls < 2 || smpls > MAXS || nmaxent < 1) return NULL;
    dm* d = (dm*)calloc(1, sizeof(dm));
    d->smpls = smpls; d->nm = nmaxent; d->minfreq = minfreq; d->runs = smpls;
    d->par = (param*)calloc((size_t)nmaxent, sizeof(param));
    for (int i = 0; i < nmaxent; ++i) { d->par[i].maxent = maxent[i]; d->par[i].noutput = 0; }
    qsort(d->par, (size_t)nmaxent, sizeof(param), paramcmp);
    d->m = (cell*)calloc((size_t)nmaxent * smpls * smpls, sizeof(cell));
    return d;
}
void orc_distmat_free(void* h) {
    dm* d = (dm*)h;
    if (!d) return;
    if (d->prenormlog) for (int i = 0; i < d->smpls; ++i) { free(d->prenormlog[i]); free(d->prenormsqrt[i]); }
    free(d->prenormlog); free(d->prenormsqrt); free(d->nfactor); free(d->runtosmpl);
    free(d->par); free(d->m); free(d);
}
/* -S: runtosmpl[runs] (sample of every run id); -N: sizes[smpls] (dataset sizes).  Either may be NULL. */
int orc_distmat_options(void* h, const int* runtosmpl, int runs, const double* sizes) {
    dm* d = (dm*)h;
    if (runtosmpl) {
        d->runs = runs;
        d->runtosmpl = (int*)malloc(sizeof(int) * (size_t)runs);
        memcpy(d->runtosmpl, runtosmpl, sizeof(int) * (size_t)runs);
    }
    if (sizes) {   /* :585-611 */
        d->nfactor = (double*)malloc(sizeof(double) * (size_t)d->smpls);
        d->prenormlog = (double**)calloc((size_t)d->smpls, sizeof(double*));
        d->prenormsqrt = (double**)calloc((size_t)d->smpls, sizeof(double*));
        for (int i = 0; i < d->smpls; ++i) {
            d->nfactor[i] = (double)1 / sizes[i];
            d->prenormlog[i] = (double*)malloc(PRECMP * sizeof(double));
            d->prenormsqrt[i] = (double*)malloc(PRECMP * sizeof(double));
            for (int j = 0; j < PRECMP; ++j) {
                d->prenormlog[i][j] = log((double)j * d->nfactor[i] + 1);
                d->prenormsqrt[i][j] = sqrt((double)j * d->nfactor[i]);
            }
        }
    }
    return 0;
}

#define OFF(d, x, y, z) ((size_t)(x) * (d)->smpls * (d)->smpls + (size_t)(y) * (d)->smpls + (z))

/* one tuple given as (id, freq) pairs in line order; returns the bucket it went to, -1 = none */
int orc_distmat_add(void* h, const unsigned* ids, const unsigned long long* freqs, unsigned npairs) {
    dm* d = (dm*)h;
    static __thread unsigned samples[MAXS], freq[MAXS];
    const int smpls = d->smpls;
    unsigned l = 0;
    memset(freq, 0, sizeof(unsigned) * (size_t)smpls);
    for (unsigned q = 0; q < npairs; ++q) {           /* parse(), :84-125 */
        unsigned run = ids[q], frq = (unsigned)freqs[q];
        if (run >= (unsigned)d->runs) return -2;
        if (frq < d->minfreq) continue;
        if (d->runtosmpl) run = (unsigned)d->runtosmpl[run];
        samples[l++] = run;
        freq[run] = frq;
    }
    ++d->rows;
    unsigned uniq = 0;
    if (l) {
        qsort(samples, l, sizeof(unsigned), ucmp);
        unsigned j = 0, k = 1;
        while (k < l) {
            while (k < l && samples[k] == samples[j]) ++k;
            if (k < l) samples[++j] = samples[k];
        }
        uniq = j + 1;
    }
    double entr;
    if (d->nfactor) {   /* normalized_entropy(), :147-165 */
        double sumN = (double)smpls, sumNlogN = 0;
        for (unsigned i = 0; i < uniq; ++i) {
            double frq = (double)freq[samples[i]] * d->nfactor[samples[i]];
            sumN += frq;
            sumNlogN += (frq + 1) * log(frq + 1) / log(2);
        }
        double entropy = (log(sumN) / log(2) - sumNlogN / sumN);
        entr = log(2) * entropy / log(smpls);
    } else {            /* entropy(), :128-145 */
        unsigned sumN = (unsigned)smpls;
        double sumNlogN = 0;
        for (unsigned i = 0; i < uniq; ++i) {
            unsigned frq = freq[samples[i]];
            sumN += frq;
            sumNlogN += (double)(frq + 1) * log(frq + 1) / log(2);
        }
        double entropy = (log(sumN) / log(2) - sumNlogN / (double)sumN);
        entr = log(2) * entropy / log(smpls);
    }
    int bucket = -1;
    for (int i = d->nm; i > 0;) {                      /* :690-703 */
        --i;
        if (entr <= d->par[i].maxent) { bucket = i; break; }
    }
    if (bucket >= 0) {
        d->par[bucket].noutput++;
        cell* M = d->m;
        for (unsigned j = 0; j < uniq; ++j)              /* add(), :167-197 */
            for (unsigned k = j; k < uniq; ++k) M[OFF(d, bucket, samples[j], samples[k])].count++;
        for (int j = 0; j < smpls; ++j)
            for (int k = j + 1; k < smpls; ++k)
                if (freq[j] || freq[k]) {
                    cell* c = &M[OFF(d, bucket, j, k)];
                    if (d->nfactor) {   /* add_normalized(), :199-228 (lgamma disabled there) */
                        if (freq[j] < PRECMP && freq[k] < PRECMP) {
                            c->lg += pow(d->prenormlog[j][freq[j]] - d->prenormlog[k][freq[k]], 2);
                            c->sq += pow(d->prenormsqrt[j][freq[j]] - d->prenormsqrt[k][freq[k]], 2);
                        } else {
                            c->lg += pow(log(1 + (double)freq[j] * d->nfactor[j]) - log(1 + (double)freq[k] * d->nfactor[k]), 2);
                            c->sq += pow(sqrt((double)freq[j] * d->nfactor[j]) - sqrt((double)freq[k] * d->nfactor[k]), 2);
                        }
                    } else
                    if (freq[j] < PRECMP && freq[k] < PRECMP) {
                        c->lg += pow(prelog[freq[j]] - prelog[freq[k]], 2);
                        c->sq += pow(presqrt[freq[j]] - presqrt[freq[k]], 2);
                        c->lgam += (freq[j] + freq[k] < PRECMP ? prelgamma[freq[j] + freq[k]] : lgamma(freq[j] + freq[k] + 1)) -
                                   prelgamma[freq[j]] - prelgamma[freq[k]] - (freq[j] + freq[k] + 1);
                    } else {
                        c->lg += pow(log(1 + freq[j]) - log(1 + freq[k]), 2);
                        c->sq += pow(sqrt(freq[j]) - sqrt(freq[k]), 2);
                        c->lgam += lgamma(freq[j] + freq[k] + 1) - lgamma(freq[j] + 1) - lgamma(freq[k] + 1) - (freq[j] + freq[k] + 1);
                    }
                }
    }
    return bucket;
}

/* server text: "path entropy id:freq id:freq ...\n" per line (the entropy column is skipped when the first row has a '.') */
long orc_distmat_add_text(void* h, const char* text, size_t len) {
    size_t pos = 0;
    long rows = 0;
    int parsep = -1;
    static __thread unsigned ids[MAXS * 4];
    static __thread unsigned long long fr[MAXS * 4];
    while (pos < len) {
        size_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        size_t p = pos;
        while (p < e && text[p] != ' ') ++p;
        if (parsep < 0) { parsep = 0; for (size_t t = p; t < e; ++t) if (text[t] == '.') { parsep = 1; break; } }
        if (parsep) { ++p; while (p < e && text[p] != ' ') ++p; }
        unsigned n = 0;
        while (p < e) {
            while (p < e && text[p] == ' ') ++p;
            if (p >= e) break;
            unsigned run = (unsigned)atoi(text + p);
            while (p < e && text[p] != ':') ++p;
            if (p >= e) return -1;
            unsigned long long f = (unsigned long long)(unsigned)atoi(text + p + 1);
            while (p < e && text[p] != ' ') ++p;
            if (n < MAXS * 4) { ids[n] = run; fr[n] = f; ++n; }
        }
        if (orc_distmat_add(h, ids, fr, n) == -2) return -2;
        ++rows;
        pos = e + 1;
    }
    return rows;
}

/* cumulative matrices and counts exactly as printed (:722-752): returns the four files' text, malloc'ed */
static void app(char** buf, size_t* len, size_t* cap, const char* s) {
    size_t n = strlen(s);
    if (*len + n + 1 > *cap) { *cap = (*cap + n) * 2 + 64; *buf = (char*)realloc(*buf, *cap); }
    memcpy(*buf + *len, s, n + 1);
    *len += n;
}
int orc_distmat_finish(void* h, char** o_count, char** o_log, char** o_sqrt, char** o_lgamma, unsigned* noutput,
                       unsigned* count, double* mlog, double* msqrt, double* mlgamma) {
    dm* d = (dm*)h;
    char* b[4] = {0, 0, 0, 0};
    size_t len[4] = {0, 0, 0, 0}, cap[4] = {0, 0, 0, 0};
    char tmp[256];
    const int s = d->smpls;
    for (int i = d->nm; i > 0;) {
        --i;
        snprintf(tmp, sizeof tmp, "Matrix for <max_entropy>=<%f> was computed from %u substrings: \n", d->par[i].maxent, d->par[i].noutput);
        for (int f = 0; f < 4; ++f) app(&b[f], &len[f], &cap[f], tmp);
        for (int j = 0; j < s; ++j) {
            for (int k = 0; k < s; ++k) {
                const cell* c = &d->m[OFF(d, i, j, k)];
                snprintf(tmp, sizeof tmp, " %u", c->count); app(&b[0], &len[0], &cap[0], tmp);
                snprintf(tmp, sizeof tmp, " %f", c->lg); app(&b[1], &len[1], &cap[1], tmp);
                snprintf(tmp, sizeof tmp, " %f", c->sq); app(&b[2], &len[2], &cap[2], tmp);
                snprintf(tmp, sizeof tmp, " %f", c->lgam); app(&b[3], &len[3], &cap[3], tmp);
                if (count) { size_t o = OFF(d, i, j, k); count[o] = c->count; mlog[o] = c->lg; msqrt[o] = c->sq; mlgamma[o] = c->lgam; }
            }
            for (int f = 0; f < 4; ++f) app(&b[f], &len[f], &cap[f], "\n");
        }
        if (noutput) noutput[i] = d->par[i].noutput;
        if (i) {                                        /* accumulate(), :230-242 */
            d->par[i - 1].noutput += d->par[i].noutput;
            for (int hh = 0; hh < s; ++hh)
                for (int k = 0; k < s; ++k) {
                    cell* dst = &d->m[OFF(d, i - 1, hh, k)];
                    const cell* src = &d->m[OFF(d, i, hh, k)];
                    dst->count += src->count; dst->lg += src->lg; dst->sq += src->sq; dst->lgam += src->lgam;
                }
        }
    }
    *o_count = b[0]; *o_log = b[1]; *o_sqrt = b[2]; *o_lgamma = b[3];
    return 0;
}
void orc_distmat_sorted(void* h, double* maxent_sorted) { dm* d = (dm*)h; for (int i = 0; i < d->nm; ++i) maxent_sorted[i] = d->par[i].maxent; }
void orc_free_text(char* p) { free(p); }
