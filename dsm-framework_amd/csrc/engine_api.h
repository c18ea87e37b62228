// engine_api.h -- what the C entry points (abi.hip) and the server side (server.hip) see of the frontier engine (engine.hip).
#pragma once
#include <string>
#include <vector>

#include "common.h"

// A sample given as an already enumerated trie: the byte stream an (unmodified reference) client sent to the server, parsed with
// TrieReader's token rules (TrieReader.h:32-106) into level arrays on the device, in the order the nodes appear = path order inside a level.
struct dsm_trie {
    int device = 0;
    dsm::u64 nodes = 0, maxfreq = 0;
    std::vector<dsm::u64> level_off;  // level l holds nodes [level_off[l], level_off[l+1]); level 0 = the root
    dsm::u64* d_freq = nullptr;       // per node
    dsm::u8* d_pl = nullptr;          // bits 0-3 children present, bits 4-6 left-char code
    dsm::u32* d_fc = nullptr;         // index of the first child inside the next level
};

namespace dsm {

// a miner behind dsm_miner*: positions are 32 or 64 bits wide inside (MinerT<P>, engine.hip)
struct MinerBase {
    virtual ~MinerBase() {}
    virtual bool stream_mode() const = 0;
    virtual int run(const char* prefix, dsm_tuple_sink ts, dsm_byte_sink bs, void* ctx, dsm_stats* out) = 0;
    virtual int run_many(const char* const* prefixes, int n, dsm_tuple_sink ts, dsm_byte_sink bs, void* ctx, dsm_stats* out,
                         dsm_prefix_byte_sink ps = nullptr, dsm_text_sink xs = nullptr) = 0;
};
bool need_wide(dsm_index* const* idx, int n, const dsm_params* p);
MinerBase* miner_create(dsm_index* const* idx, int n, const dsm_params& p, bool stream_mode, int* rc);
int mine_once(dsm_index* const* idx, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);
int enumerate_once(const dsm_index* idx, const char* prefix, u32 fmin, u32 maxdepth, dsm_byte_sink sink, void* ctx, dsm_stats* stats);
int merge_once(bool wide, dsm_trie* const* tr, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);

// ---- the server side's engine runs (dsm_server, server.hip) ----
struct ServerOrder {  // Engine<P>::NodeOrder without the position type
    u32 depth = 0;
    std::vector<u32> sym;
    std::vector<std::vector<u16>> ord;
};
struct ServerEngines;  // the engines a server keeps between its runs (engine.hip)
ServerEngines* server_engines_create();
void server_engines_destroy(ServerEngines* e);
// One engine run over the given tries (sample id = position): capture (shallow pass, nothing emitted), a unit (a unit that does not fit
// the buffers splits like any prefix) or the closing pass over the depths lo..hi.  which: 0 = an engine for this run only, 1 = the kept
// engine for the passes over the tops of the streams, 2 = the kept engine for units.
int server_run(bool wide, dsm_trie* const* tr, int n, const dsm_params& q, const std::string& prefix, dsm_tuple_sink sink, void* ctx, bool emit, u32 lo,
               u32 hi, u32 expand_cap, const ServerOrder* seed, ServerOrder* capture, dsm_stats* out, ServerEngines* keep = nullptr, int which = 0);

}  // namespace dsm
