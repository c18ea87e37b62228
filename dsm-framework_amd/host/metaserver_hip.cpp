// metaserver_hip -- drop-in for the reference server (metaserver.cpp:488-815): same options, the expected sample names
// on stdin, one TCP connection per sample carrying the reference wire protocol, reference-format tuples on stdout.
// Unmodified reference clients (metaenumerate) can feed it.  A reader thread per connection hands the bytes to the library as
// they arrive (dsm_server_feed: the reference's token rules and R checksums; what is decoded moves to the card in windows, so the
// host never holds a stream).  With --prefix-len K (the length of the prefix this server's clients enforce: one server per prefix,
// wrapper-SLURM/example-server.sh:27-41) the streams are merged WHILE they arrive, as the reference's traverse() does
// (metaserver.cpp:682-739): the subtree of every node of depth K + 1 + X (--unit-extra X, default 1: up to 16 subtrees) is merged on
// the GPU and printed as soon as every connection is past it, and leaves the card.  Without --prefix-len the merge starts when the
// last stream has ended.
//   metaserver_hip -E emax [-e emin] [-P pmin] [--pmax N] [-p port] [-m mindepth] [-v] [--device D] [--prefix-len K [--unit-extra X]] < names.txt
#include <getopt.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/dsmhip.h"

// metaserver.cpp:472-484: the lines are formatted on the GPU (dsm_formatter_*: the bytes of the reference's printf loop)
static dsm_formatter* g_fmt = nullptr;
static int g_fmt_device = 0;
static int print_batch(void*, const dsm_tuple_batch* b) {
    if (!g_fmt && dsm_formatter_create(g_fmt_device, &g_fmt)) return 1;
    const char* text = nullptr;
    size_t len = 0;
    if (dsm_formatter_format(g_fmt, b, &text, &len)) return 1;
    return fwrite(text, 1, len, stdout) != len;
}

int main(int argc, char** argv) {
    if (argc <= 1) { std::cerr << "usage: " << argv[0] << " [options] < names.txt" << std::endl; return 1; }
    dsm_params p;
    dsm_params_default(&p);
    int port = 54666, device = 0, prefix_len = -1, unit_extra = 1;  // metaserver.cpp:515
    bool verbose = false;
    static option long_options[] = {{"pmin", required_argument, 0, 'P'},     {"pmax", required_argument, 0, 258},
                                    {"port", required_argument, 0, 'p'},     {"mindepth", required_argument, 0, 'm'},
                                    {"emin", required_argument, 0, 'e'},     {"emax", required_argument, 0, 'E'},
                                    {"verbose", no_argument, 0, 'v'},        {"debug", no_argument, 0, 256},
                                    {"device", required_argument, 0, 257},   {"prefix-len", required_argument, 0, 259},
                                    {"unit-extra", required_argument, 0, 260},
                                    {0, 0, 0, 0}};
    int c, oi = 0;
    while ((c = getopt_long(argc, argv, "P:p:m:e:E:F:T:vA", long_options, &oi)) != -1) {
        switch (c) {
            case 'P': p.pmin = (unsigned)atoi(optarg); break;
            case 258: p.pmax = (unsigned)atoi(optarg); break;
            case 'p': port = atoi(optarg); if (port < 1024) { std::cerr << argv[0] << ": argument of -p, --port must be >= 1024" << std::endl; return 1; } break;
            case 'm': p.mindepth = (unsigned)atoi(optarg); break;
            case 'e': p.emin = atof(optarg); break;
            case 'E': p.emax = atof(optarg); break;
            case 'v': verbose = true; break;
            case 256: case 'F': case 'T': case 'A': break;  // progress options of the reference: accepted, no effect
            case 257: device = atoi(optarg); break;
            case 259: prefix_len = atoi(optarg); if (prefix_len < 0 || prefix_len > 32) { std::cerr << argv[0] << ": argument of --prefix-len must be 0..32" << std::endl; return 1; } break;
            case 260: unit_extra = atoi(optarg); if (unit_extra < 0 || unit_extra > 4) { std::cerr << argv[0] << ": argument of --unit-extra must be 0..4" << std::endl; return 1; } break;
            default: std::cerr << "usage: " << argv[0] << " [options] < names.txt" << std::endl; return 1;
        }
    }
    if (p.emax < 0) { std::cerr << argv[0] << ": error: expecting parameter --emax" << std::endl; return 1; }  // metaserver.cpp:582-586
    if (p.emin > p.emax) { std::cerr << argv[0] << ": error: -e <double> must be smaller than or equal to -E <double>" << std::endl; return 1; }

    std::map<std::string, int> libtoid;  // metaserver.cpp:606-653
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::string name = line.substr(0, line.find_first_of('\t'));
        if (libtoid.count(name)) { std::cerr << "DUPLICATE CLIENT NAME IN stdin! name = " << name << std::endl; return 1; }
        int id = (int)libtoid.size();
        libtoid[name] = id;
    }
    const size_t d = libtoid.size();
    if (d == 0) { std::cerr << "no expected inputs" << std::endl; return 1; }
    if (d > 273) { std::cerr << "Too many input readers requested! MAX_READERS was 273" << std::endl; return 1; }

    int sock = socket(AF_INET, SOCK_STREAM, 0);  // ServerSocket::init, ServerSocket.cpp:13-48
    int yes = 1;
    setsockopt(sock, SOL_SOCKET, SO_REUSEADDR, &yes, sizeof yes);
    sockaddr_in addr;
    memset(&addr, 0, sizeof addr);
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = INADDR_ANY;
    addr.sin_port = htons(port);
    if (bind(sock, (sockaddr*)&addr, sizeof addr) < 0 || listen(sock, 256) < 0) { std::cerr << "ERROR on binding" << std::endl; return 1; }

    std::vector<bool> seen(d, false);
    std::vector<std::thread> readers;
    dsm_server* srv = nullptr;
    g_fmt_device = device;
    if (dsm_server_create((int)d, device, prefix_len, unit_extra, &p, print_batch, nullptr, &srv)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
    std::string perr;
    std::mutex err_mu;
    size_t pending = d;
    // An error inside the accept loop ends the process the way the reference's exit(1) does; reader threads of earlier
    // connections may still be running, and returning from main with joinable threads would abort instead.
    auto fatal = [] { std::cerr.flush(); _exit(1); };
    while (pending) {  // metaserver.cpp:682-728: handshake 'S' name '.', then the node stream until EOF
        int fd = accept(sock, nullptr, nullptr);
        if (fd < 0) { std::cerr << "ERROR on accept" << std::endl; fatal(); }
        uint8_t ch = 0;
        if (recv(fd, &ch, 1, MSG_WAITALL) != 1 || ch != 'S') { std::cerr << "received invalid start byte: " << (int)ch << std::endl; fatal(); }
        std::string name;
        for (;;) {
            if (recv(fd, &ch, 1, MSG_WAITALL) != 1) { std::cerr << "connection closed inside the handshake" << std::endl; fatal(); }
            if (ch == '.') break;
            name += (char)ch;
        }
        auto f = libtoid.find(name);
        if (f == libtoid.end()) { std::cerr << "received invalid libname: \"" << name << "\"" << std::endl; fatal(); }
        const int id = f->second;
        if (seen[id]) { std::cerr << "DUPLICATE CONNECTING CLIENT! id = " << id << ", name = " << name << std::endl; fatal(); }
        seen[id] = true;
        --pending;
        if (verbose) std::cerr << "new connection id = " << id << ", name = " << name << " (" << pending << " pending)" << std::endl;
        readers.emplace_back([fd, id, srv, &perr, &err_mu] {
            // decode while receiving: the streams of the samples are parsed side by side (the decoder is single-threaded per stream)
            int rc = 0;
            std::vector<uint8_t> buf(1 << 20);
            for (;;) {
                ssize_t r = recv(fd, buf.data(), buf.size(), 0);
                if (r <= 0) break;
                if (!rc) rc = dsm_server_feed(srv, id, buf.data(), (size_t)r);  // (after an error the connection is drained, like a reader that gave up)
            }
            close(fd);
            if (!rc) rc = dsm_server_end(srv, id);
            if (rc) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (perr.empty()) perr = std::string(dsm_last_error()) + " (reader " + std::to_string(id) + ")";
            }
        });
    }
    for (auto& t : readers) t.join();
    close(sock);
    if (!perr.empty()) { std::cerr << "error: " << perr << std::endl; return 1; }
    dsm_stats st;
    if (dsm_server_finish(srv, &st)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
    fflush(stdout);
    if (verbose)
        std::cerr << "Number of paths: " << st.union_nodes << std::endl << "Number of reported paths: " << st.tuples << std::endl
                  << "Number of reported occs: " << st.pairs << std::endl;
    if (verbose && prefix_len >= 0) {
        uint64_t peak = 0;
        const uint64_t units = dsm_server_units(srv, &peak);
        std::cerr << "Subtrees merged while receiving: " << units << " (largest: " << peak << " nodes over all samples)" << std::endl;
    }
    dsm_server_destroy(srv);
    return 0;
}
