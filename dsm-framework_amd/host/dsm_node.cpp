// dsm_node -- fused driver for one node: all samples, enumeration + merge + entropy filter on the GPUs, reference-format
// tuples out (what N metaenumerate clients + one metaserver per prefix produce, metaserver.cpp:467-485).
// Sample ids follow the order of the index files (the server's names order).
//   dsm_node -E emax [-e emin] [-P pmin] [--pmax N] [-m mindepth] [-f fmin] [-M maxdepth]
//            [--device D | --devices D0,D1,... [--exchange allgather|owner]] [--out-prefix path.] -p PREFIX[,PREFIX...] a.fmi b.fmi ...
// One device: one miner holds every index.  Several devices (--devices): the reference's fan-out -- one metaenumerate per
// sample (wrapper-SLURM/example-client.sh:27-30), one socket and one server per prefix (metaenumerate.cpp:268-309) -- becomes
// one thread + HIP stream per GPU inside this process, the samples dealt out in blocks (sample k lives on device k / (d / G)),
// and ONE ncclAllGather (RCCL over xGMI) per frontier level, issued on the engine's stream from the library's exchange
// callback.  Prefix k is filtered, ordered and emitted by device k mod G only; the outputs are written in prefix order.
// --exchange owner: the reference's own partition -- prefix k is merged by device k mod G ALONE (one metaserver per prefix,
// wrapper-SLURM/example-server.sh:27-41); the other devices send it their columns and get the union's child masks back
// (dsm_params.owner_mode).  One lane per owner: G x G threads, lane j of every device works on the prefixes j, j + G, ..., so every
// device is the server of one lane and a client in the others; the lanes' collectives are ordered by a gate per device.
// With --out-prefix every prefix goes to <out-prefix><PREFIX>.txt (server-wrapper.sh:35 naming), else to stdout.
#include <getopt.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <ctime>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsmhip.h"

// stdout carries the tuples.  Libraries underneath (the collective library prints a version banner) must not write into them:
// the tuples go to a private copy of the descriptor, and descriptor 1 is pointed at stderr.
static FILE* g_out = nullptr;

static const char* USAGE = "usage: dsm_node -E emax [options] [--devices D0,D1,..] -p PREFIX[,PREFIX..] a.fmi b.fmi ...";

// The lines of metaserver.cpp:472-484 come from the GPU (dsm_formatter_*: byte for byte the reference's printf loop); a formatter
// belongs to the sink that uses it (one emitter thread per miner calls a sink).
struct Fmt {
    dsm_formatter* f = nullptr;
    int device = 0;
    int text(const dsm_tuple_batch* b, const char** t, size_t* n) {
        if (!f && dsm_formatter_create(device, &f)) return 1;
        return dsm_formatter_format(f, b, t, n);
    }
    ~Fmt() { if (f) dsm_formatter_destroy(f); }
};
static Fmt g_fmt;  // the single-device run's

static int to_file(void* ctx, const dsm_tuple_batch* b) {
    const char* text = nullptr;
    size_t len = 0;
    if (g_fmt.text(b, &text, &len)) return 1;
    return fwrite(text, 1, len, (FILE*)ctx) != len;
}

// ---- several devices ---------------------------------------------------------------------------------------------------
// The rank threads meet once between opening their indexes and creating their miners (whose first step is a collective): a rank
// that could not open an index tells the others there, and nobody enters a collective that a missing rank would leave hanging.
struct Rendezvous {
    std::mutex mu;
    std::condition_variable cv;
    int expected = 0, arrived = 0;
    std::atomic<int> failed{0};
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (++arrived == expected) cv.notify_all();
        else cv.wait(lk, [&] { return arrived == expected; });
    }
};

// Before any communicator, stream or thread exists: every device ordinal is one of this node's cards and every sample is an index
// this library can open (dsm_index_probe: header, code table and tree shape, host work only).  What can still fail later (device
// memory) fails inside the rank threads, whose rendezvous keeps the others out of the collectives.
static int validate_inputs(const std::vector<int>& devs, const std::vector<std::string>& files) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { std::cerr << "dsm_node: no HIP device" << std::endl; return 1; }
    for (int d : devs)
        if (d < 0 || d >= ndev) { std::cerr << "dsm_node: --devices: no device " << d << " (this node has " << ndev << ")" << std::endl; return 1; }
    int bad = 0;
    for (const std::string& f : files)
        if (dsm_index_probe(f.c_str(), nullptr)) { std::cerr << "dsm_node: " << f << ": " << dsm_last_error() << std::endl; bad = 1; }
    return bad;
}

struct Rank {
    Fmt fmt;
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::vector<std::string> files;          // this rank's samples, in id order
    const std::vector<std::string>* prefixes = nullptr;
    std::vector<std::string>* out = nullptr; // one text per prefix, filled by the prefix's owner
    dsm_params p;
    std::string err;
    dsm_stats st;
    Rendezvous* meet = nullptr;
};

// dsm_allgather_fn: every rank contributes `bytes` bytes, receives world * bytes, rank-major; ordered on the engine's stream
static int rccl_allgather(void* ctx, const void* send, void* recv, size_t bytes, void* stream) {
    Rank* r = (Rank*)ctx;
    return ncclAllGather(send, recv, bytes, ncclUint8, r->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

// The tuples of a prefix start with the prefix: a batch goes to the output of the (longest) prefix its first path starts with.
static int to_prefix_text(void* ctx, const dsm_tuple_batch* b) {
    Rank* r = (Rank*)ctx;
    if (!b->ntuples) return 0;
    const char* path = b->path_bytes + b->path_off[0];
    const size_t plen = b->path_off[1] - b->path_off[0];
    int best = -1;
    for (size_t k = 0; k < r->prefixes->size(); ++k) {
        const std::string& pre = (*r->prefixes)[k];
        if ((int)(k % r->world) != r->rank) continue;  // only prefixes this rank owns arrive here
        if (pre.size() <= plen && memcmp(pre.data(), path, pre.size()) == 0 && (best < 0 || pre.size() > (*r->prefixes)[best].size())) best = (int)k;
    }
    if (best < 0) return 1;
    const char* text = nullptr;
    size_t len = 0;
    if (r->fmt.text(b, &text, &len)) return 1;
    (*r->out)[best].append(text, len);
    return 0;
}

static void run_rank(Rank* r) {
    auto fail = [&](const std::string& m) { r->err = m; };
    if (hipSetDevice(r->device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    std::vector<dsm_index*> idx;
    for (const std::string& f : r->files) {
        dsm_index* ix = nullptr;
        if (dsm_index_open(f.c_str(), r->device, &ix)) { fail(f + ": " + dsm_last_error()); break; }
        idx.push_back(ix);
    }
    if (!r->err.empty()) r->meet->failed.store(1);
    r->meet->arrive_and_wait();
    // Past this point every rank is inside collectives (miner creation agrees on capacities, mining gathers every level): a rank
    // that fails alone cannot be waited for by the others, so such a failure ends the process (the message goes out first).
    auto fatal = [&](const std::string& msg) {
        std::cerr << "dsm_node: device " << r->device << ": " << msg << std::endl;
        fflush(nullptr);
        _exit(1);
    };
    dsm_miner* m = nullptr;
    if (!r->meet->failed.load()) {
        dsm_params p = r->p;
        p.world_size = (uint32_t)r->world;
        p.rank = (uint32_t)r->rank;
        p.allgather = rccl_allgather;
        p.allgather_ctx = r;
        p.emit_owner_only = 1;
        p.stream = r->stream;
        if (dsm_miner_create(idx.data(), (int)idx.size(), &p, 0, &m)) fatal(std::string("miner: ") + dsm_last_error());
    }
    if (m) {
        std::vector<const char*> pre;
        for (const std::string& s : *r->prefixes) pre.push_back(s.c_str());
        if (dsm_miner_mine_many(m, pre.data(), (int)pre.size(), to_prefix_text, r, &r->st)) fatal(std::string("mine: ") + dsm_last_error());
        dsm_miner_destroy(m);
    }
    for (auto* ix : idx) dsm_index_close(ix);
    (void)hipStreamDestroy(r->stream);
}

static int run_devices(const std::vector<int>& devs, const dsm_params& p, const std::vector<std::string>& prefixes,
                       const std::vector<std::string>& files, const std::string& outprefix) {
    const int G = (int)devs.size();
    if (files.size() % G) { std::cerr << "dsm_node: the number of samples must be a multiple of the number of devices" << std::endl; return 1; }
    const size_t nlocal = files.size() / G;
    // A batch is routed to its prefix by its first path (to_prefix_text), which is only unambiguous when no prefix continues
    // another one; the reference runs one server per prefix of one length (wrapper-SLURM/example-server.sh:27-41).
    for (size_t a = 0; a < prefixes.size(); ++a)
        for (size_t b = 0; b < prefixes.size(); ++b)
            if (a != b && prefixes[b].compare(0, prefixes[a].size(), prefixes[a]) == 0) {
                std::cerr << "dsm_node: --devices needs prefixes none of which starts with another (" << prefixes[a] << ", " << prefixes[b] << ")" << std::endl;
                return 1;
            }
    if (int rc = validate_inputs(devs, files)) return rc;
    std::vector<ncclComm_t> comms(G);
    if (ncclCommInitAll(comms.data(), G, devs.data()) != ncclSuccess) { std::cerr << "dsm_node: ncclCommInitAll failed" << std::endl; return 1; }
    std::vector<std::string> out(prefixes.size());
    std::vector<Rank> ranks(G);
    Rendezvous meet;
    meet.expected = G;
    for (int r = 0; r < G; ++r) {
        ranks[r].meet = &meet;
        ranks[r].rank = r; ranks[r].world = G; ranks[r].device = devs[r]; ranks[r].fmt.device = devs[r]; ranks[r].comm = comms[r];
        ranks[r].files.assign(files.begin() + r * nlocal, files.begin() + (r + 1) * nlocal);
        ranks[r].prefixes = &prefixes; ranks[r].out = &out; ranks[r].p = p;
    }
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r) th.emplace_back(run_rank, &ranks[r]);
    for (auto& t : th) t.join();
    for (int r = 0; r < G; ++r) (void)ncclCommDestroy(comms[r]);
    int rc = 0;
    for (int r = 0; r < G; ++r)
        if (!ranks[r].err.empty()) { std::cerr << "dsm_node: device " << devs[r] << ": " << ranks[r].err << std::endl; rc = 1; }
    if (rc) return rc;
    for (size_t k = 0; k < prefixes.size(); ++k) {
        FILE* f = g_out;
        if (!outprefix.empty()) {
            f = fopen((outprefix + prefixes[k] + ".txt").c_str(), "w");
            if (!f) { std::cerr << "cannot open output for prefix " << prefixes[k] << std::endl; return 1; }
        }
        fwrite(out[k].data(), 1, out[k].size(), f);
        if (f != g_out) fclose(f);
    }
    uint64_t nodes = 0, tuples = 0;
    for (int r = 0; r < G; ++r) { nodes += ranks[r].st.reported; tuples += ranks[r].st.tuples; }
    fflush(g_out);
    std::cerr << G << " device(s): " << nodes << " nodes, " << ranks[0].st.union_nodes << " paths, " << tuples << " reported" << std::endl;
    return 0;
}

// ---- several devices, owner mode ------------------------------------------------------------------------------------------
struct LaneRank {
    Fmt fmt;
    int rank = 0, lane = 0, world = 1, device = 0;
    dsm_rccl* comm = nullptr;
    dsm_rccl_gate* gate = nullptr;
    hipStream_t stream = nullptr;
    dsm_miner* miner = nullptr;
    std::vector<std::string> prefixes;       // of this lane
    std::vector<std::string>* out = nullptr; // one text per prefix of this lane (filled by the lane's owner)
    dsm_stats st;
};
struct LaneSinkCtx { LaneRank* lr; };
static int to_lane_text(void* ctx, const dsm_tuple_batch* b) {
    LaneRank* r = (LaneRank*)ctx;
    if (!b->ntuples) return 0;
    const char* path = b->path_bytes + b->path_off[0];
    const size_t plen = b->path_off[1] - b->path_off[0];
    int best = -1;
    for (size_t k = 0; k < r->prefixes.size(); ++k) {
        const std::string& pre = r->prefixes[k];
        if (pre.size() <= plen && memcmp(pre.data(), path, pre.size()) == 0) best = (int)k;
    }
    if (best < 0) return 1;
    const char* text = nullptr;
    size_t len = 0;
    if (r->fmt.text(b, &text, &len)) return 1;
    (*r->out)[best].append(text, len);
    return 0;
}

static int run_devices_owner(const std::vector<int>& devs, const dsm_params& p0, const std::vector<std::string>& prefixes,
                             const std::vector<std::string>& files, const std::string& outprefix) {
    const int G = (int)devs.size();
    if (files.size() % G) { std::cerr << "dsm_node: the number of samples must be a multiple of the number of devices" << std::endl; return 1; }
    const size_t nlocal = files.size() / G;
    for (size_t a = 0; a < prefixes.size(); ++a)
        for (size_t b = 0; b < prefixes.size(); ++b)
            if (a != b && prefixes[b].compare(0, prefixes[a].size(), prefixes[a]) == 0) {
                std::cerr << "dsm_node: --devices needs prefixes none of which starts with another (" << prefixes[a] << ", " << prefixes[b] << ")" << std::endl;
                return 1;
            }
    if (int rc = validate_inputs(devs, files)) return rc;
    std::vector<std::vector<uint8_t>> ids(G, std::vector<uint8_t>(DSM_RCCL_ID_BYTES));
    for (int j = 0; j < G; ++j)
        if (dsm_rccl_unique_id(ids[j].data())) { std::cerr << "dsm_node: " << dsm_last_error() << std::endl; return 1; }
    std::vector<std::vector<LaneRank>> lr(G, std::vector<LaneRank>(G));       // [rank][lane]
    std::vector<std::vector<std::string>> outs(G);                            // [lane][prefix of the lane]
    for (int j = 0; j < G; ++j) {
        std::vector<std::string> mine;
        for (size_t k = j; k < prefixes.size(); k += G) mine.push_back(prefixes[k]);
        outs[j].resize(mine.size());
        for (int r = 0; r < G; ++r) {
            lr[r][j].rank = r; lr[r][j].lane = j; lr[r][j].world = G; lr[r][j].device = devs[r]; lr[r][j].fmt.device = devs[r];
            lr[r][j].prefixes = mine; lr[r][j].out = &outs[j];
        }
    }
    // per device: indexes, then lane by lane a communicator and a miner (creation runs collectives: every device thread goes
    // through the lanes in the same order), then the gate; a failure before the lanes run ends the process
    std::vector<std::vector<dsm_index*>> idx(G);
    auto fatal = [](int device, const std::string& msg) {
        std::cerr << "dsm_node: device " << device << ": " << msg << std::endl;
        fflush(nullptr);
        _exit(1);
    };
    {
        std::vector<std::thread> th;
        for (int r = 0; r < G; ++r)
            th.emplace_back([&, r] {
                if (hipSetDevice(devs[r]) != hipSuccess) fatal(devs[r], "hipSetDevice failed");
                for (size_t k = 0; k < nlocal; ++k) {
                    dsm_index* ix = nullptr;
                    if (dsm_index_open(files[r * nlocal + k].c_str(), devs[r], &ix)) fatal(devs[r], files[r * nlocal + k] + ": " + dsm_last_error());
                    idx[r].push_back(ix);
                }
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                dsm_rccl_gate* gate = nullptr;
                if (dsm_rccl_gate_create(G, &gate)) fatal(devs[r], dsm_last_error());
                for (int j = 0; j < G; ++j) {
                    LaneRank& L = lr[r][j];
                    L.gate = gate;
                    if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) fatal(devs[r], "hipStreamCreate failed");
                    if (dsm_rccl_create(ids[j].data(), G, r, devs[r], &L.comm)) fatal(devs[r], dsm_last_error());
                    dsm_params p = p0;
                    p.world_size = (uint32_t)G; p.rank = (uint32_t)r;
                    p.allgather = dsm_rccl_allgather; p.allgather_ctx = L.comm;
                    p.owner_mode = 1; p.owner_rank = (uint32_t)j;
                    p.gather = dsm_rccl_gather; p.bcast = dsm_rccl_bcast; p.owner_ctx = L.comm;
                    p.stream = L.stream;
                    p.arena_bytes = (uint64_t)(free_b * 0.7 / G);
                    if (dsm_miner_create(idx[r].data(), (int)idx[r].size(), &p, 0, &L.miner)) fatal(devs[r], std::string("miner: ") + dsm_last_error());
                }
                for (int j = 0; j < G; ++j) (void)dsm_rccl_attach_gate(lr[r][j].comm, gate, j);
            });
        for (auto& t : th) t.join();
    }
    {
        std::vector<std::thread> th;
        for (int r = 0; r < G; ++r)
            for (int j = 0; j < G; ++j)
                th.emplace_back([&, r, j] {
                    LaneRank& L = lr[r][j];
                    (void)hipSetDevice(L.device);
                    dsm_rccl_gate_begin(L.gate, j);
                    std::vector<const char*> pre;
                    for (const std::string& s_ : L.prefixes) pre.push_back(s_.c_str());
                    if (dsm_miner_mine_many(L.miner, pre.data(), (int)pre.size(), to_lane_text, &L, &L.st)) fatal(L.device, std::string("mine: ") + dsm_last_error());
                    dsm_rccl_gate_retire(L.gate, j);
                });
        for (auto& t : th) t.join();
    }
    uint64_t nodes = 0, tuples = 0, sent = 0, recvd = 0;
    for (int r = 0; r < G; ++r) {
        for (int j = 0; j < G; ++j) {
            LaneRank& L = lr[r][j];
            nodes += L.st.reported; tuples += L.st.tuples; sent += L.st.exchange_bytes_sent; recvd += L.st.exchange_bytes_received;
            dsm_miner_destroy(L.miner);
            dsm_rccl_destroy(L.comm);
            (void)hipStreamDestroy(L.stream);
        }
        dsm_rccl_gate_destroy(lr[r][0].gate);
        for (auto* ix : idx[r]) dsm_index_close(ix);
    }
    for (size_t k = 0; k < prefixes.size(); ++k) {
        FILE* f = g_out;
        if (!outprefix.empty()) {
            f = fopen((outprefix + prefixes[k] + ".txt").c_str(), "w");
            if (!f) { std::cerr << "cannot open output for prefix " << prefixes[k] << std::endl; return 1; }
        }
        const std::string& text = outs[k % G][k / G];
        fwrite(text.data(), 1, text.size(), f);
        if (f != g_out) fclose(f);
    }
    fflush(g_out);
    std::cerr << G << " device(s), owner mode: " << nodes << " nodes, " << tuples << " reported, " << sent << " bytes sent, " << recvd << " received" << std::endl;
    return 0;
}

int main(int argc, char** argv) {
    fflush(stdout);
    g_out = fdopen(dup(1), "w");
    if (!g_out || dup2(2, 1) < 0) { std::cerr << "dsm_node: cannot set up the output" << std::endl; return 1; }
    dsm_params p;
    dsm_params_default(&p);
    std::string prefixes = "", outprefix = "", devlist = "", exchange = "allgather";
    int device = 0;
    static option long_options[] = {{"pmin", required_argument, 0, 'P'},     {"pmax", required_argument, 0, 258},
                                    {"mindepth", required_argument, 0, 'm'}, {"emin", required_argument, 0, 'e'},
                                    {"emax", required_argument, 0, 'E'},     {"fmin", required_argument, 0, 'f'},
                                    {"maxdepth", required_argument, 0, 'M'}, {"prefix", required_argument, 0, 'p'},
                                    {"device", required_argument, 0, 257},   {"out-prefix", required_argument, 0, 259},
                                    {"devices", required_argument, 0, 260},  {"exchange", required_argument, 0, 261},
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
            case 260: devlist = optarg; break;
            case 261: exchange = optarg; break;
            default: std::cerr << USAGE << std::endl; return 1;
        }
    }
    if (p.emax < 0) { std::cerr << argv[0] << ": error: expecting parameter --emax" << std::endl; return 1; }  // metaserver.cpp:582-586
    if (p.emin > p.emax) { std::cerr << argv[0] << ": error: -e <double> must be smaller than or equal to -E <double>" << std::endl; return 1; }
    if (prefixes.empty() || optind >= argc) { std::cerr << USAGE << std::endl; return 1; }
    std::vector<std::string> pre;
    {
        std::stringstream ss(prefixes);
        std::string one;
        while (std::getline(ss, one, ',')) pre.push_back(one);
    }
    if (!devlist.empty()) {
        std::vector<int> devs;
        std::stringstream ss(devlist);
        std::string one;
        while (std::getline(ss, one, ',')) devs.push_back(atoi(one.c_str()));
        std::vector<std::string> files(argv + optind, argv + argc);
        if (devs.empty() || (exchange != "allgather" && exchange != "owner")) { std::cerr << USAGE << std::endl; return 1; }
        if (exchange == "owner") return run_devices_owner(devs, p, pre, files, outprefix);
        return run_devices(devs, p, pre, files, outprefix);
    }
    std::vector<dsm_index*> idx;
    for (int k = optind; k < argc; ++k) {
        dsm_index* ix = nullptr;
        if (dsm_index_open(argv[k], device, &ix)) { std::cerr << argv[k] << ": " << dsm_last_error() << std::endl; return 1; }
        idx.push_back(ix);
    }
    dsm_miner* m = nullptr;
    if (dsm_miner_create(idx.data(), (int)idx.size(), &p, 0, &m)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
    g_fmt.device = device;
    if (outprefix.empty()) {  // everything to one stream: one call, so that the tuples of a prefix leave while the next one is expanded
        std::vector<const char*> ps;
        for (const std::string& one : pre) ps.push_back(one.c_str());
        dsm_stats st;
        // (the lines are formatted on the card and only the text crosses the bus: dsm_miner_mine_text)
        auto text_to_file = [](void* ctx, const char* text, size_t len) -> int { return fwrite(text, 1, len, (FILE*)ctx) != len; };
        const int passes = getenv("DSM_NODE_PASSES") ? atoi(getenv("DSM_NODE_PASSES")) : 1;  // (timing aid: the same prefixes again; the output repeats)
        for (int pass = 0; pass < passes; ++pass) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            if (dsm_miner_mine_text(m, ps.data(), (int)ps.size(), text_to_file, g_out, &st)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
            fflush(g_out);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            std::cerr << pre.size() << " prefix(es): " << st.reported << " nodes, " << st.union_nodes << " paths, " << st.tuples << " reported in "
                      << (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 << " ms" << std::endl;
        }
        pre.clear();
    }
    for (const std::string& one : pre) {
        FILE* out = g_out;
        if (!outprefix.empty()) {
            out = fopen((outprefix + one + ".txt").c_str(), "w");
            if (!out) { std::cerr << "cannot open output for prefix " << one << std::endl; return 1; }
        }
        dsm_stats st;
        if (dsm_miner_mine(m, one.c_str(), to_file, out, &st)) { std::cerr << "error: " << dsm_last_error() << std::endl; return 1; }
        if (out != g_out) fclose(out);
        std::cerr << "prefix " << one << ": " << st.reported << " nodes, " << st.union_nodes << " paths, " << st.tuples << " reported" << std::endl;
    }
    fflush(g_out);
    dsm_miner_destroy(m);
    for (auto* ix : idx) dsm_index_close(ix);
    return 0;
}
