// abi.hip -- the C entry points of the enumeration / mining path (include/dsmhip.h): argument checks, then the engine (engine_api.h).
#include <cstring>
#include <string>

#include "common.h"
#include "engine_api.h"

using namespace dsm;

extern "C" {

void dsm_params_default(dsm_params* p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->prefix = "";
    p->fmin = 10;            // metaenumerate.cpp:141
    p->maxdepth = ~0u;       // metaenumerate.cpp:142
    p->pmin = 2;             // metaserver.cpp:126
    p->pmax = 0;
    p->mindepth = 0;
    p->emin = 0.0;
    p->emax = -1.0;          // mandatory in the reference CLI (metaserver.cpp:582-586)
    p->world_size = 1;
}

int dsm_enumerate(const dsm_index* idx, const char* prefix, uint32_t fmin, uint32_t maxdepth, dsm_byte_sink sink, void* ctx, dsm_stats* stats) {
    if (!idx) return fail(DSM_E_INVAL, "dsm_enumerate: null index");
    return enumerate_once(idx, prefix, fmin, maxdepth, sink, ctx, stats);
}

int dsm_mine(dsm_index* const* idx, int nlocal, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (!idx || nlocal <= 0 || !p) return fail(DSM_E_INVAL, "dsm_mine: bad arguments");
    for (int k = 0; k < nlocal; ++k)
        if (!idx[k]) return fail(DSM_E_INVAL, "dsm_mine: null index");
    return mine_once(idx, nlocal, p, sink, ctx, stats);
}

int dsm_miner_create(dsm_index* const* idx, int nlocal, const dsm_params* p, int stream_mode, dsm_miner** out) {
    if (!idx || nlocal <= 0 || !p || !out) return fail(DSM_E_INVAL, "dsm_miner_create: bad arguments");
    for (int k = 0; k < nlocal; ++k)
        if (!idx[k]) return fail(DSM_E_INVAL, "dsm_miner_create: null index");
    if (stream_mode && (nlocal != 1 || p->world_size > 1)) return fail(DSM_E_INVAL, "stream mode takes exactly one local index");
    *out = nullptr;
    int rc = 0;
    MinerBase* m = miner_create(idx, nlocal, *p, stream_mode != 0, &rc);
    if (!m) return rc;
    *out = reinterpret_cast<dsm_miner*>(m);
    return DSM_OK;
}
// a miner is created for tuples or for the wire stream (dsm_miner_create's stream_mode): the other kind of entry point is refused
static int mode_check(dsm_miner* m, bool want_stream, const char* fn) {
    if (!m) return fail(DSM_E_INVAL, std::string(fn) + ": null miner");
    if (reinterpret_cast<MinerBase*>(m)->stream_mode() != want_stream)
        return fail(DSM_E_INVAL, std::string(fn) + (want_stream ? ": the miner was not created with stream_mode" : ": the miner was created with stream_mode"));
    return 0;
}
int dsm_miner_mine(dsm_miner* m, const char* prefix, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (int rc = mode_check(m, false, "dsm_miner_mine")) return rc;
    return reinterpret_cast<MinerBase*>(m)->run(prefix, sink, nullptr, ctx, stats);
}
int dsm_miner_enumerate(dsm_miner* m, const char* prefix, dsm_byte_sink sink, void* ctx, dsm_stats* stats) {
    if (int rc = mode_check(m, true, "dsm_miner_enumerate")) return rc;
    return reinterpret_cast<MinerBase*>(m)->run(prefix, nullptr, sink, ctx, stats);
}
int dsm_miner_enumerate_many(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_prefix_byte_sink sink, void* ctx, dsm_stats* stats) {
    if (!m || !prefixes || nprefix < 0) return fail(DSM_E_INVAL, "dsm_miner_enumerate_many: bad arguments");
    if (int rc = mode_check(m, true, "dsm_miner_enumerate_many")) return rc;
    return reinterpret_cast<MinerBase*>(m)->run_many(prefixes, nprefix, nullptr, nullptr, ctx, stats, sink);
}
int dsm_miner_mine_many(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_tuple_sink sink, void* ctx, dsm_stats* stats) {
    if (!m || !prefixes || nprefix < 0) return fail(DSM_E_INVAL, "dsm_miner_mine_many: bad arguments");
    if (int rc = mode_check(m, false, "dsm_miner_mine_many")) return rc;
    return reinterpret_cast<MinerBase*>(m)->run_many(prefixes, nprefix, sink, nullptr, ctx, stats);
}
int dsm_miner_mine_text(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_text_sink sink, void* ctx, dsm_stats* stats) {
    if (!m || !prefixes || nprefix < 0 || !sink) return fail(DSM_E_INVAL, "dsm_miner_mine_text: bad arguments");
    if (int rc = mode_check(m, false, "dsm_miner_mine_text")) return rc;
    return reinterpret_cast<MinerBase*>(m)->run_many(prefixes, nprefix, nullptr, nullptr, ctx, stats, nullptr, sink);
}
void dsm_miner_destroy(dsm_miner* m) { delete reinterpret_cast<MinerBase*>(m); }


}  // extern "C"
