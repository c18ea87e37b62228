// exchange.hip -- a ready-made dsm_allgather_fn over RCCL for hosts that run one process per GPU (bench.py under
// torch.distributed.run, an MPI job): ncclAllGather on the engine's stream, straight from the library's per-level callback --
// no interpreter, no lock of the host language on the data path.  librccl is loaded at run time (dlopen), so the library
// itself has no link dependency on it; the host only has to carry the 128-byte id from rank 0 to the other ranks.
// Several prefix lanes of one process (each with its own communicator, so that one lane's collective overlaps the other lane's
// kernels) must enqueue their collectives in the same order on every rank: dsm_rccl_gate hands out turns round-robin among the
// lanes that are still running, which is the same order everywhere because every rank walks the same union trie.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dsmhip.h"

namespace dsm {
int fail(int code, const std::string& msg);
}

namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err_string = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        get_id = (decltype(get_id))dlsym(lib, "ncclGetUniqueId");
        init_rank = (decltype(init_rank))dlsym(lib, "ncclCommInitRank");
        all_gather = (decltype(all_gather))dlsym(lib, "ncclAllGather");
        send = (decltype(send))dlsym(lib, "ncclSend");
        recv = (decltype(recv))dlsym(lib, "ncclRecv");
        group_start = (decltype(group_start))dlsym(lib, "ncclGroupStart");
        group_end = (decltype(group_end))dlsym(lib, "ncclGroupEnd");
        broadcast = (decltype(broadcast))dlsym(lib, "ncclBroadcast");
        destroy = (decltype(destroy))dlsym(lib, "ncclCommDestroy");
        err_string = (decltype(err_string))dlsym(lib, "ncclGetErrorString");
        if (!get_id || !init_rank || !all_gather || !destroy || !err_string || !send || !recv || !group_start || !group_end || !broadcast) { err = "librccl lacks an expected symbol"; return false; }
        return true;
    }
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
bool rccl_ready() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return g_rccl.load();
}
}  // namespace

struct dsm_rccl_gate {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> active;
    int turn = 0;
    void advance() {  // next lane that is still running
        const int n = (int)active.size();
        for (int k = 1; k <= n; ++k) {
            const int j = (turn + k) % n;
            if (active[j]) { turn = j; return; }
        }
    }
};
struct dsm_rccl {
    ncclComm_t comm = nullptr;
    int device = 0;
    int world = 1, rank = 0;
    dsm_rccl_gate* gate = nullptr;
    int lane = 0;
};

static_assert(DSM_RCCL_ID_BYTES == sizeof(ncclUniqueId), "the id crosses the C ABI as plain bytes");

int dsm_rccl_unique_id(uint8_t* id) {
    if (!id) return dsm::fail(DSM_E_INVAL, "dsm_rccl_unique_id: null argument");
    if (!rccl_ready()) return dsm::fail(DSM_E_UNSUPPORTED, g_rccl.err);
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.get_id(&u);
    if (r != ncclSuccess) return dsm::fail(DSM_E_HIP, std::string("ncclGetUniqueId: ") + g_rccl.err_string(r));
    memcpy(id, &u, sizeof u);
    return DSM_OK;
}

int dsm_rccl_create(const uint8_t* id, int world_size, int rank, int device, dsm_rccl** out) {
    if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) return dsm::fail(DSM_E_INVAL, "dsm_rccl_create: bad arguments");
    *out = nullptr;
    if (!rccl_ready()) return dsm::fail(DSM_E_UNSUPPORTED, g_rccl.err);
    if (hipSetDevice(device) != hipSuccess) return dsm::fail(DSM_E_NODEV, "dsm_rccl_create: bad device ordinal");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.init_rank(&comm, world_size, u, rank);  // collective: every rank of the communicator calls it
    if (r != ncclSuccess) return dsm::fail(DSM_E_HIP, std::string("ncclCommInitRank: ") + g_rccl.err_string(r));
    dsm_rccl* c = new dsm_rccl();
    c->comm = comm;
    c->device = device;
    c->world = world_size;
    c->rank = rank;
    *out = c;
    return DSM_OK;
}

// From here on the communicator's collectives wait for their lane's turn.  Attach the gate after the miners exist: their
// creation runs collectives lane by lane on one thread, which no turn-taking of concurrently running lanes describes.
int dsm_rccl_attach_gate(dsm_rccl* c, dsm_rccl_gate* gate, int lane) {
    if (!c || (gate && (lane < 0 || lane >= (int)gate->active.size()))) return dsm::fail(DSM_E_INVAL, "dsm_rccl_attach_gate: bad arguments");
    c->gate = gate;
    c->lane = lane;
    return DSM_OK;
}

// dsm_allgather_fn with ctx = the dsm_rccl*: enqueued on the engine's stream, i.e. ordered after the kernels that filled sendbuf
// and before the ones that read recvbuf
int dsm_rccl_allgather(void* ctx, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, void* stream) {
    dsm_rccl* c = (dsm_rccl*)ctx;
    if (!c || !c->comm) return 1;
    dsm_rccl_gate* g = c->gate;
    if (g) {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return g->turn == c->lane; });
    }
    const ncclResult_t r = g_rccl.all_gather(sendbuf, recvbuf, bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream);
    if (g) {
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->advance();
        }
        g->cv.notify_all();
    }
    return r == ncclSuccess ? 0 : 1;
}

// a lane's turn among the lanes of the process (see dsm_rccl_gate)
namespace {
struct Turn {
    dsm_rccl_gate* g;
    Turn(dsm_rccl* c) : g(c->gate) {
        if (!g) return;
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return g->turn == c->lane; });
    }
    ~Turn() {
        if (!g) return;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->advance();
        }
        g->cv.notify_all();
    }
};
}  // namespace

// dsm_gather_fn: every rank but the root sends its bytes to the root, which receives them rank-major (one ncclRecv per peer in a
// group: the peers' messages arrive over their own links at the same time) and copies its own part in place.  One server per
// prefix, one connection per client and prefix (metaenumerate.cpp:268-309, metaserver.cpp:682-728) -- between GPUs.
int dsm_rccl_gather(void* ctx, int root, const void* sendbuf, void* recvbuf, size_t bytes, void* stream) {
    dsm_rccl* c = (dsm_rccl*)ctx;
    if (!c || !c->comm || root < 0 || root >= c->world) return 1;
    Turn turn(c);
    hipStream_t st = (hipStream_t)stream;
    if (c->rank != root) return g_rccl.send(sendbuf, bytes, ncclUint8, root, c->comm, st) == ncclSuccess ? 0 : 1;
    if (hipMemcpyAsync((uint8_t*)recvbuf + (size_t)root * bytes, sendbuf, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
    if (c->world == 1) return 0;
    bool ok = g_rccl.group_start() == ncclSuccess;
    for (int r = 0; r < c->world && ok; ++r)
        if (r != root) ok = g_rccl.recv((uint8_t*)recvbuf + (size_t)r * bytes, bytes, ncclUint8, r, c->comm, st) == ncclSuccess;
    return (g_rccl.group_end() == ncclSuccess && ok) ? 0 : 1;
}
// dsm_bcast_fn
int dsm_rccl_bcast(void* ctx, int root, void* buf, size_t bytes, void* stream) {
    dsm_rccl* c = (dsm_rccl*)ctx;
    if (!c || !c->comm || root < 0 || root >= c->world) return 1;
    Turn turn(c);
    if (c->world == 1) return 0;
    return g_rccl.broadcast(buf, buf, bytes, ncclUint8, root, c->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

int dsm_copy_from_device(void* host_dst, const void* device_src, size_t bytes, void* stream) {
    if (hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return dsm::fail(DSM_E_HIP, "dsm_copy_from_device failed");
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? DSM_OK : dsm::fail(DSM_E_HIP, "dsm_copy_from_device failed");
}
int dsm_copy_to_device(void* device_dst, const void* host_src, size_t bytes, void* stream) {
    if (hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return dsm::fail(DSM_E_HIP, "dsm_copy_to_device failed");
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? DSM_OK : dsm::fail(DSM_E_HIP, "dsm_copy_to_device failed");
}

void dsm_rccl_destroy(dsm_rccl* c) {
    if (!c) return;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.destroy(c->comm);
    }
    delete c;
}

int dsm_rccl_gate_create(int nlanes, dsm_rccl_gate** out) {
    if (!out || nlanes < 1) return dsm::fail(DSM_E_INVAL, "dsm_rccl_gate_create: bad arguments");
    dsm_rccl_gate* g = new dsm_rccl_gate();
    g->active.assign((size_t)nlanes, 1);
    *out = g;
    return DSM_OK;
}
void dsm_rccl_gate_begin(dsm_rccl_gate* g, int lane) {
    if (!g || lane < 0 || lane >= (int)g->active.size()) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->active[lane] = 1;
}
void dsm_rccl_gate_retire(dsm_rccl_gate* g, int lane) {
    if (!g || lane < 0 || lane >= (int)g->active.size()) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->active[lane] = 0;
        if (g->turn == lane) g->advance();
    }
    g->cv.notify_all();
}
void dsm_rccl_gate_reset(dsm_rccl_gate* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        for (auto& a : g->active) a = 1;
        g->turn = 0;
    }
    g->cv.notify_all();
}
void dsm_rccl_gate_destroy(dsm_rccl_gate* g) { delete g; }
