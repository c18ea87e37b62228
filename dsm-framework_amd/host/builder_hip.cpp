// builder_hip -- drop-in for the reference `builder` (builder.cpp:329-472): FASTA in, <input>.fmi out, on the GPU.
//   builder_hip [options] <input> [output]
//     -s <int>, --sample-rate <int>   stored in the file (the enumeration path never reads samples); default 124
//     -v, --verbose                   progress on stderr
//     -h, --help
//     --device D                      HIP device ordinal (default 0)
// The index is written to <output>.fmi, or <input>.fmi without an output name (TextCollection::save adds the suffix,
// builder.cpp:425-426).  Input "-" (stdin) is spooled to a temporary file first.  The colour-space (-c), rotation (-R) and
// reverse-index (-F) variants of the reference are read-aligner leftovers outside this product's path and are refused.
#include <getopt.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <string>

#include "../../include/dsmhip.h"

static void usage(const char* name) {
    std::cerr << "usage: " << name << " [options] <input> [output]" << std::endl
              << "Check README or `" << name << " --help' for more information." << std::endl;
}

int main(int argc, char** argv) {
    unsigned samplerate = 124;  // TEXTCOLLECTION_DEFAULT_SAMPLERATE, TextCollectionBuilder.h:30
    int device = 0;
    bool verbose = false;
    static option long_options[] = {{"sample-rate", required_argument, 0, 's'}, {"help", no_argument, 0, 'h'}, {"verbose", no_argument, 0, 'v'},
                                    {"device", required_argument, 0, 256},     {"color", no_argument, 0, 'c'}, {"rotation", required_argument, 0, 'R'},
                                    {"reverse", no_argument, 0, 'F'},          {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "cR:s:Fhv", long_options, &oi)) != -1) {
        switch (c) {
            case 's': {
                char* end = nullptr;
                long v = strtol(optarg, &end, 10);
                if (!end || *end || v < 1) {
                    std::cerr << argv[0] << ": argument of -s, --sample-rate must be of type <int>, and greater than or equal to 1" << std::endl;
                    return 1;
                }
                samplerate = (unsigned)v;
                break;
            }
            case 'v': verbose = true; break;
            case 256: device = atoi(optarg); break;
            case 'h':
                std::cerr << "usage: " << argv[0] << " [options] <input> [output]" << std::endl
                          << "<input> is the input filename (FASTA).  If no output filename is given, the index is stored as <input>.fmi" << std::endl
                          << " -s <int>, --sample-rate <int> Sampling rate stored in the index (default: 124)." << std::endl
                          << " -h, --help                    Display command line options." << std::endl
                          << " -v, --verbose                 Print progress information." << std::endl
                          << " --device <int>                GPU to build on (default: 0)." << std::endl;
                return 0;
            case 'c': case 'R': case 'F':
                std::cerr << argv[0] << ": the colour-space / rotation / reverse-index variants are not part of this build" << std::endl;
                return 1;
            default: usage(argv[0]); return 1;
        }
    }
    if (argc - optind < 1) { usage(argv[0]); return 1; }
    if (argc - optind > 2) std::cerr << "Warning: too many filenames given! Ignoring all but first two." << std::endl;
    std::string input = argv[optind++];
    std::string output = optind != argc ? argv[optind++] : "";
    std::string spool;
    if (input == "-") {
        if (output.empty()) { std::cerr << argv[0] << ": reading from stdin needs an output name" << std::endl; return 1; }
        char tmpl[] = "/tmp/builder_hip.XXXXXX";
        int fd = mkstemp(tmpl);
        if (fd < 0) { std::cerr << argv[0] << ": cannot create a temporary file" << std::endl; return 1; }
        spool = tmpl;
        char buf[1 << 16];
        size_t k;
        while ((k = fread(buf, 1, sizeof buf, stdin)) > 0)
            if (write(fd, buf, k) != (ssize_t)k) { std::cerr << argv[0] << ": writing the temporary file failed" << std::endl; close(fd); unlink(tmpl); return 1; }
        close(fd);
        input = spool;
    }
    if (output.empty()) output = input;
    output += ".fmi";
    time_t t0 = time(NULL);
    if (verbose) std::cerr << "Building the forward index:" << std::endl;
    dsm_build_info info;
    int rc = dsm_build_fasta(input.c_str(), output.c_str(), samplerate, device, &info);
    if (!spool.empty()) unlink(spool.c_str());
    if (rc) {
        std::cerr << "builder: " << dsm_last_error() << std::endl;
        return 1;
    }
    if (verbose)
        std::cerr << "Creating new index with " << info.number_of_texts << " sequences, n = " << info.n << std::endl
                  << "Saving to file " << output << std::endl
                  << "Save complete. (total wall-clock time " << difftime(time(NULL), t0) << " s)" << std::endl;
    return 0;
}
