// dsm_node -- fused single-process driver: all samples of one node, enumeration + merge + entropy filter on the
// GPU(s) of this process, reference-format tuples out (what N metaenumerate clients + one metaserver per prefix
// produce, metaserver.cpp:467-485).  Sample ids follow the order of the index files (the server's names order).
//   dsm_node -E emax [-e emin] [-P pmin] [--pmax N] [-m mindepth] [-f fmin] [-M maxdepth] [--device D]
//            [--out-prefix path.] -p PREFIX[,PREFIX...] a.fmi b.fmi ...
// With --out-prefix every prefix goes to <out-prefix><PREFIX>.txt (server-wrapper.sh:35 naming), else to stdout.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

static int to_file(void* ctx, const dsm_tuple_batch* b) {
    char* text = nullptr;
    size_t len = 0;
    if (dsm_format_batch(b, &text, &len)) return 1;
    size_t w = fwrite(text, 1, len, (FILE*)ctx);
    dsm_free(text);
    return w != len;
}

int main(int argc, char** argv) {
    dsm_params p;
    dsm_params_default(&p);
    std::string prefixes = "", outprefix = "";
    int device = 0;
    static option long_options[] = {{"pmin", required_argument, 0, 'P'},     {"pmax", required_argument, 0, 258},
                                    {"mindepth", required_argument, 0, 'm'}, {"emin", required_argument, 0, 'e'},
                                    {"emax", required_argument, 0, 'E'},     {"fmin", required_argument, 0, 'f'},
                                    {"maxdepth", required_argument, 0, 'M'}, {"prefix", required_argument, 0, 'p'},
                                    {"device", required_argument, 0, 257},   {"out-prefix", required_argument, 0, 259},
                                    {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "P:m:e:E:f:M:p:", long_options, &oi)) != -1) {
        switch (c) {
            case 'P': p.pmin = (unsigned)atoi(optarg); break;
            case 258: p.pmax = (unsigned)atoi(optarg); break;
            case 'm': p.mindepth = (unsigned)atoi(optarg); break;
            case 'e': p.emin = atof(optarg); break;
            case 'E': p.emax = atof(optarg); break;
            case 'f': p.fmin = (unsigned)atoi(optarg); break;
            case 'M': p.maxdepth = (unsigned)atoi(optarg); break;
            case 'p': prefixes = optarg; break;
            case 257: device = atoi(optarg); break;
            case 259: outprefix = optarg; break;
            default: std::cerr << "usage: dsm_node -E emax [options] -p PREFIX[,PREFIX..] a.fmi b.fmi ..." << std::endl; return 1;
        }
    }
    if (p.emax < 0) { std::cerr << argv[0] << ": error: expecting parameter --emax" << std::endl; return 1; }  // metaserver.cpp:582-586
    if (p.emin > p.emax) { std::cerr << argv[0] << ": error: -e <double> must be smaller than or equal to -E <double>" << std::endl; return 1; }
    if (prefixes.empty() || optind >= argc) { std::cerr << "usage: dsm_node -E emax [options] -p PREFIX[,PREFIX..] a.fmi b.fmi ..." << std::endl; return 1; }
    std::vector<dsm_index*> idx;
    for (int k = optind; k < argc; ++k) {
        dsm_index* ix = nullptr;
        if (dsm_index_open(argv[k], device, &ix)) { std::cerr << argv[k] << ": " << dsm_last_error() << std::endl; return 1; }
        idx.push_back(ix);
    }
    dsm_miner* m = nullptr;
    if (dsm_miner_create(idx.data(), (int)idx.size(), &p, 0, &m)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
    std::stringstream ss(prefixes);
    std::string pre;
    while (std::getline(ss, pre, ',')) {
        FILE* out = stdout;
        if (!outprefix.empty()) {
            out = fopen((outprefix + pre + ".txt").c_str(), "w");
            if (!out) { std::cerr << "cannot open output for prefix " << pre << std::endl; return 1; }
        }
        dsm_stats st;
        if (dsm_miner_mine(m, pre.c_str(), to_file, out, &st)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
        if (out != stdout) fclose(out);
        std::cerr << "prefix " << pre << ": " << st.reported << " nodes, " << st.union_nodes << " paths, " << st.tuples << " reported" << std::endl;
    }
    dsm_miner_destroy(m);
    for (auto* ix : idx) dsm_index_close(ix);
    return 0;
}
