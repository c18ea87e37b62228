// smtxt2entropy_hip -- drop-in for wrapper-distance-matrix/smtxt2entropy.c (default mode): reads metaserver tuple
// lines from stdin, accumulates the distance matrices on the GPU through libdsmhip's C ABI and writes the tool's four
// files count.<suffix>, log.<suffix>, sqrt.<suffix>, lgamma.<suffix>.  Options as the reference (smtxt2entropy.c:482-556):
//   -s,--samples N | -S,--samplefile f   -m,--maxent a,b,...  |  -e,--entstep x   -F,--file suffix
//   [-M,--minfreq n] [-N,--normalize sizes] [-v]
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

static void die(const char* msg, const char* name) {
    fprintf(stderr, "\nerror: %s\nPlease see `%s --help'\n", msg, name);
    exit(1);
}

int main(int argc, char** argv) {
    int smpls = -1, verbose = 0;
    unsigned minfreq = 0;
    std::vector<double> maxent;
    std::string suffix, smplsfile, normfile;
    static struct option lo[] = {{"samples", required_argument, 0, 's'}, {"samplefile", required_argument, 0, 'S'},
                                 {"maxent", required_argument, 0, 'm'},  {"entstep", required_argument, 0, 'e'},
                                 {"file", required_argument, 0, 'F'},    {"normalize", required_argument, 0, 'N'},
                                 {"minfreq", required_argument, 0, 'M'}, {"verbose", no_argument, 0, 'v'},
                                 {"help", no_argument, 0, 'h'},          {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "s:S:m:e:F:N:M:vh", lo, &oi)) != -1) {
        switch (c) {
        case 's': smpls = atoi(optarg); break;
        case 'm': {
            if (!maxent.empty()) die("give either -m,--maxent or -e,--entstep, but not both.", argv[0]);
            std::string v = optarg;
            for (char* p = strtok(&v[0], " ,-"); p; p = strtok(nullptr, " ,-")) {  // parse_doubles, smtxt2entropy.c:287-316
                double x = atof(p);
                if (x < 0.0 || x > 1.0) die("arguments of -m,--maxent must be between 0 and 1", argv[0]);
                maxent.push_back(x);
            }
            break;
        }
        case 'e': {
            if (!maxent.empty()) die("give either -m,--maxent or -e,--entstep, but not both.", argv[0]);
            double buf[4096];
            int n = dsm_distmat_steps(atof(optarg), buf, 4096);
            if (n < 0) die(dsm_last_error(), argv[0]);
            maxent.assign(buf, buf + n);
            break;
        }
        case 'F': suffix = optarg; break;
        case 'M': minfreq = (unsigned)atoi(optarg); break;
        case 'v': verbose = 1; break;
        case 'S': smplsfile = optarg; break;
        case 'N': normfile = optarg; break;
        default: fprintf(stderr, "usage: %s -s smpls (-m a,b,.. | -e step) -F suffix [-M minfreq] [-v] < tuples\n", argv[0]); return 1;
        }
    }
    if ((smpls == -1) == smplsfile.empty()) die("give either -s,--samples or -S,--samplefile, but not both.", argv[0]);
    std::vector<int32_t> mapping;
    if (!smplsfile.empty()) {  // parse_samples_file, smtxt2entropy.c:385-423: one sample id per line, line number = run id
        FILE* f = fopen(smplsfile.c_str(), "rt");
        if (!f) { fprintf(stderr, "error: sample file -S,--samplefile %s does not exist. Aborting!\n", smplsfile.c_str()); return 1; }
        int v;
        smpls = 0;
        while (fscanf(f, "%d\n", &v) == 1) { mapping.push_back(v); if (v + 1 > smpls) smpls = v + 1; }
        fclose(f);
        if (smpls < 2 || (int)mapping.size() < smpls) die("unable to parse the samples file in the argument -S,--samplefile.", argv[0]);
    }
    std::vector<double> sizes;
    if (!normfile.empty()) {   // parse_size_file, smtxt2entropy.c:425-455: "name<TAB>size" per sample
        FILE* f = fopen(normfile.c_str(), "rt");
        if (!f) { fprintf(stderr, "error: size file for -N,--normalize %s does not exist. Aborting!\n", normfile.c_str()); return 1; }
        char name[128];
        double x;
        while ((int)sizes.size() < smpls && fscanf(f, "%100s\t%lf\n", name, &x) == 2) sizes.push_back(x);
        fclose(f);
        if ((int)sizes.size() != smpls) { fprintf(stderr, "error: unable to parse the file -N,--normalize %s.\n", normfile.c_str()); return 1; }
    }
    if (smpls < 2) die("the argument -s,--samples must be at least 2.", argv[0]);
    if (maxent.empty()) die("the argument -m,--maxent is mandatory.", argv[0]);
    const char* kinds[4] = {"count", "log", "sqrt", "lgamma"};
    for (const char* k : kinds) {  // open_output_file, smtxt2entropy.c:364-383: existing files are refused
        std::string fn = std::string(k) + "." + suffix;
        if (FILE* f = fopen(fn.c_str(), "r")) { fclose(f); fprintf(stderr, "error: output file %s already exists. Aborting!\n", fn.c_str()); return 1; }
    }
    dsm_distmat* dm = nullptr;
    if (dsm_distmat_create_ex(0, (uint32_t)smpls, maxent.data(), (uint32_t)maxent.size(), minfreq, mapping.empty() ? nullptr : mapping.data(),
                              (uint32_t)mapping.size(), sizes.empty() ? nullptr : sizes.data(), &dm)) {
        fprintf(stderr, "error: %s\n", dsm_last_error());
        return 1;
    }
    std::vector<char> buf(64u << 20);
    std::string carry;
    unsigned long rows = 0;
    for (;;) {  // whole lines per batch
        size_t n = fread(buf.data(), 1, buf.size(), stdin);
        if (n == 0) break;
        carry.append(buf.data(), n);
        size_t last = carry.rfind('\n');
        if (last == std::string::npos) continue;
        if (dsm_distmat_add_text(dm, carry.data(), last + 1)) { fprintf(stderr, "error: %s\n", dsm_last_error()); return 1; }
        for (size_t i = 0; i <= last; ++i) rows += carry[i] == '\n';
        carry.erase(0, last + 1);
    }
    if (!carry.empty()) {
        carry.push_back('\n');
        if (dsm_distmat_add_text(dm, carry.data(), carry.size())) { fprintf(stderr, "error: %s\n", dsm_last_error()); return 1; }
        ++rows;
    }
    const size_t nm = maxent.size(), cells = nm * (size_t)smpls * smpls;
    std::vector<double> ms(nm), l(cells), q(cells), g(cells);
    std::vector<uint32_t> nout(nm), cnt(cells);
    if (dsm_distmat_finish(dm, ms.data(), nout.data(), cnt.data(), l.data(), q.data(), g.data())) { fprintf(stderr, "error: %s\n", dsm_last_error()); return 1; }
    char* text[4];
    if (dsm_distmat_format((uint32_t)smpls, (uint32_t)nm, ms.data(), nout.data(), cnt.data(), l.data(), q.data(), g.data(), text)) {
        fprintf(stderr, "error: %s\n", dsm_last_error());
        return 1;
    }
    for (int k = 0; k < 4; ++k) {
        std::string fn = std::string(kinds[k]) + "." + suffix;
        FILE* f = fopen(fn.c_str(), "w");
        if (!f) { fprintf(stderr, "error: could not open output file %s. Aborting!\n", fn.c_str()); return 1; }
        fputs(text[k], f);
        fclose(f);
        dsm_free(text[k]);
    }
    if (verbose) fprintf(stderr, "Number of lines processed: %lu\n", rows);
    dsm_distmat_destroy(dm);
    return 0;
}
