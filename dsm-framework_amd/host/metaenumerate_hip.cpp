// metaenumerate_hip -- drop-in for the reference client (metaenumerate.cpp:130-323): same command line, same
// hostinfo on stdin, same bytes on the wire, enumeration on an MI355X through libdsmhip.so.
//   metaenumerate_hip [-f/--fmin N] [-M/--maxdepth N] [-C/--check] [-v] [--device D] <index.fmi> < hostinfo.txt
// The reference opens every connection at start-up and walks the prefixes on one OpenMP thread each; here the
// connections are also opened up front (servers wait for every expected client before they start merging,
// metaserver.cpp:682-728) and the prefixes are enumerated one after the other on the GPU, each one's bytes leaving the card and
// entering its socket while the next is being enumerated.
#include <getopt.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

struct HostInfo { std::string name; int port; std::string prefix; int fd; };

static int connect_to(const std::string& host, int port) {  // ClientSocket::ClientSocket, ClientSocket.cpp:19-48
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) { std::cerr << "error: ERROR opening socket" << std::endl; exit(1); }
    hostent* server = gethostbyname(host.c_str());
    if (!server) { std::cerr << "error: ERROR, no such host" << std::endl; exit(1); }
    sockaddr_in addr;
    memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET;
    memcpy(&addr.sin_addr.s_addr, server->h_addr, server->h_length);
    addr.sin_port = htons(port);
    if (connect(fd, (sockaddr*)&addr, sizeof addr) < 0) { std::cerr << "error: ERROR connecting" << std::endl; exit(1); }
    return fd;
}

static int send_all(void* ctx, const uint8_t* p, size_t n) {  // writen(), ClientSocket.cpp:50-74
    int fd = *(int*)ctx;
    while (n > 0) {
        ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
        if (w <= 0) {
            if (w < 0 && errno == EINTR) continue;
            return 1;
        }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}

// sink of dsm_miner_enumerate_many: pieces of prefix k go to its connection, which is closed when the prefix is complete
static int send_prefix(void* ctx, int k, const uint8_t* p, size_t n) {
    std::vector<int*>& fds = *(std::vector<int*>*)ctx;
    if (!p) { close(*fds[k]); return 0; }
    return send_all(fds[k], p, n);
}

int main(int argc, char** argv) {
    if (argc <= 1) { std::cerr << "usage: " << argv[0] << " [options] <index> < hostinfo.txt" << std::endl; return 0; }
    unsigned fmin = 10, maxdepth = ~0u;  // metaenumerate.cpp:141-142
    bool checkonly = false, verbose = false;
    int device = 0;
    static option long_options[] = {{"fmin", required_argument, 0, 'f'}, {"maxdepth", required_argument, 0, 'M'},
                                    {"check", no_argument, 0, 'C'},      {"verbose", no_argument, 0, 'v'},
                                    {"help", no_argument, 0, 'h'},       {"debug", no_argument, 0, 256},
                                    {"device", required_argument, 0, 257}, {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "f:M:Cvh", long_options, &oi)) != -1) {
        switch (c) {
            case 'f': fmin = (unsigned)atoi(optarg); if (fmin < 1) { std::cerr << "argument of -f, --fmin must be >= 1" << std::endl; return 1; } break;
            case 'M': maxdepth = (unsigned)atoi(optarg); if (maxdepth < 1) { std::cerr << "argument of -M, --maxdepth must be >= 1" << std::endl; return 1; } break;
            case 'C': checkonly = true; break;
            case 'v': verbose = true; break;
            case 256: break;
            case 257: device = atoi(optarg); break;
            default: std::cerr << "usage: " << argv[0] << " [options] <index> < hostinfo.txt" << std::endl; return 1;
        }
    }
    if (argc - optind != 1) { std::cerr << argv[0] << ": expecting index filename" << std::endl; return 1; }
    std::string indexfile = argv[optind];
    std::vector<HostInfo> hosts;  // metaenumerate.cpp:201-231
    while (std::cin.good()) {
        HostInfo hi;
        hi.fd = -1;
        std::cin >> hi.name;
        if (hi.name.empty()) break;
        std::cin >> hi.port;
        if (hi.port < 1024) { std::cerr << "error: invalid port number: " << hi.port << std::endl; abort(); }
        std::cin >> hi.prefix;
        if (hi.prefix.empty()) { std::cerr << "error: invalid enforced path: " << hi.prefix << std::endl; abort(); }
        hosts.push_back(hi);
    }
    if (hosts.empty()) { std::cerr << "error: empty host info" << std::endl; abort(); }

    dsm_index* idx = nullptr;
    if (dsm_index_open(indexfile.c_str(), device, &idx)) {
        std::cerr << argv[0] << ": could not read index file " << indexfile << ": " << dsm_last_error() << std::endl;
        return 1;
    }
    if (checkonly) {  // metaenumerate.cpp:93-127
        uint64_t total = 0;
        if (dsm_index_check(idx, &total)) { std::cerr << dsm_last_error() << std::endl; return 1; }
        std::cerr << indexfile << ": " << (total == dsm_index_length(idx) ? "OK     " : "FAILED *********** ") << "n = " << dsm_index_length(idx)
                  << ", total = " << total << std::endl;
        return 0;
    }
    const std::string name = dsm_index_name(idx);
    for (auto& h : hosts) {  // handshake: 'S' libname '.'  (metaenumerate.cpp:283-286)
        h.fd = connect_to(h.name, h.port);
        std::string hello = "S" + name + ".";
        if (send_all(&h.fd, (const uint8_t*)hello.data(), hello.size())) { std::cerr << "error: ERROR writing the output" << std::endl; return 1; }
    }
    dsm_params p;
    dsm_params_default(&p);
    p.fmin = fmin;
    p.maxdepth = maxdepth;
    dsm_miner* m = nullptr;
    dsm_index* one[1] = {idx};
    if (dsm_miner_create(one, 1, &p, 1, &m)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
    // the reference pops hostinfo lines from the back (metaenumerate.cpp:274-275); all prefixes go down in one call, so the bytes of
    // one prefix cross PCIe and enter its socket while the GPU enumerates the next
    std::vector<const char*> order;
    std::vector<int*> fds;
    for (size_t k = hosts.size(); k-- > 0;) { order.push_back(hosts[k].prefix.c_str()); fds.push_back(&hosts[k].fd); }
    dsm_stats st;
    if (dsm_miner_enumerate_many(m, order.data(), (int)order.size(), send_prefix, &fds, &st)) {
        std::cerr << "error: " << dsm_last_error() << std::endl;
        return 1;
    }
    const uint64_t total_occs = st.reported;
    if (verbose) std::cerr << order.size() << " prefixes: " << st.reported << " nodes, expand " << st.expand_ms << " ms" << std::endl;
    if (verbose) std::cerr << "Number of reported alignments: " << total_occs << std::endl;
    dsm_miner_destroy(m);
    dsm_index_close(idx);
    return 0;
}
