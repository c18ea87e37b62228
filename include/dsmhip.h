/* dsmhip.h -- C ABI of libdsmhip.so: the MI355X-native substring-enumeration hot path of the
 * Distributed String Mining framework (FM-index LF-mapping enumeration + cross-sample merge).
 *
 * The reference has no FFI; the path sits behind (1) the .fmi v17 file, (2) the C++ virtual
 * TextCollection interface, (3) the client->server wire protocol, (4) metaserver's stdout.
 * Every entry point below names the reference interface it replaces (paths relative to the
 * reference tree).  No C++ or torch type crosses this boundary: plain pointers and sizes only.
 *
 * Conventions
 *   - return value 0 = success, negative = -errno style failure; dsm_last_error() (thread local)
 *     has the message.  The library never exits the process (the reference CLIs exit(1)/abort();
 *     that behaviour belongs to the CLI wrappers, not the library).
 *   - handles are immutable after open; query entry points are re-entrant.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - a GPU is mandatory: there is no CPU fallback anywhere behind this header.
 */
#ifndef DSMHIP_H_
#define DSMHIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_ABI_VERSION 3

/* error codes (negative errno values) */
#define DSM_OK 0
#define DSM_E_INVAL (-22)      /* bad argument */
#define DSM_E_NOENT (-2)       /* file not found */
#define DSM_E_IO (-5)          /* truncated / unreadable file */
#define DSM_E_FORMAT (-74)     /* not a .fmi v14..v17 (reference throws std::runtime_error, FMIndex.cpp:267-268) */
#define DSM_E_NOMEM (-12)
#define DSM_E_NODEV (-19)      /* no usable HIP device */
#define DSM_E_UNSUPPORTED (-95)
#define DSM_E_CAPACITY (-28)   /* a frontier level does not fit the arena: use a longer prefix or a larger arena */
#define DSM_E_HIP (-71)        /* a HIP runtime call or kernel failed */
#define DSM_E_SINK (-125)      /* the caller's sink returned non-zero */

const char* dsm_last_error(void);
int dsm_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Index object.  Replaces TextCollection::load + class FMIndex (TextCollection.cpp:27-62,
 * FMIndex.cpp:245-357): reads a .fmi written by the unmodified reference builder and makes it
 * resident in HBM in the device layout (DESIGN.md "Data layout").
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsm_index dsm_index;

/* HuffWT::TCodeEntry, HuffWT.h:13-46 */
typedef struct dsm_code {
    uint64_t count;
    uint32_t bits;
    uint32_t code;
} dsm_code;

#define DSM_OPEN_KEEP_WT 1u /* keep the file's Huffman wavelet tree (reference 3-array layout) in HBM too */

int dsm_index_open(const char* fmi_path, int device, dsm_index** out);
int dsm_index_open_ex(const char* fmi_path, int device, unsigned flags, dsm_index** out);
/* The checks of dsm_index_open without a device: header, code table and wavelet-tree shape of the file (FMIndex.cpp:305-372,
 * metaenumerate.cpp:243-247: version, counts that add up to n, not color coded, every bit vector inside the file).  Host work only --
 * a driver validates every sample with it before it creates communicators or threads that a late failure would leave hanging.
 * *n (may be NULL) = BWT length. */
int dsm_index_probe(const char* fmi_path, uint64_t* n);
void dsm_index_close(dsm_index* idx);
/* TextCollection::getLength(), FMIndex.h:68-69 */
uint64_t dsm_index_length(const dsm_index* idx);
/* C[] of FMIndex.h:84-90 and the Huffman code table of HuffWT.h:49-54 */
int dsm_index_meta(const dsm_index* idx, uint64_t C[256], dsm_code codes[256]);
/* sample name the server sees: basename up to the first '.', metaenumerate.cpp:79-88 */
const char* dsm_index_name(const dsm_index* idx);
int dsm_index_device(const dsm_index* idx);
/* bytes of HBM held by the handle */
uint64_t dsm_index_device_bytes(const dsm_index* idx);
/* Residency (BASELINE configs[4]: more indexes than should stay in HBM).  dsm_index_offload gives the index blocks'
 * HBM back and keeps them in pinned host memory (the copy is made once: an index is immutable); queries and miners
 * fail with DSM_E_INVAL until dsm_index_reload, which allocates HBM again and queues ONE asynchronous host-to-device
 * copy on `stream` (work queued on that stream afterwards sees the index; other streams must wait for it).
 * The caller makes sure no query or miner is running on the index while it is offloaded or reloaded. */
int dsm_index_offload(dsm_index* idx);
int dsm_index_reload(dsm_index* idx, void* stream);
int dsm_index_resident(const dsm_index* idx);

/* LF(c,i) = C[c] + rank_c(BWT, i) -- TextCollection::LF, FMIndex.h:84-90 (HuffWT::rank HuffWT.h:66-83,
 * BitRank::rank BitRank.cpp:191-195).  i = UINT64_MAX is legal (rank(-1) = 0).  Host pointers. */
int dsm_lf_batch(const dsm_index* idx, const uint8_t* c, const uint64_t* i, uint64_t* out, size_t k, void* stream);
/* Same on device-resident arrays, asynchronous on `stream`.
 * layout: DSM_LAYOUT_PLANES = interleaved bit-plane blocks (the enumeration layout);
 *         DSM_LAYOUT_WT     = the file's wavelet tree walked node by node (needs DSM_OPEN_KEEP_WT). */
#define DSM_LAYOUT_PLANES 0u
#define DSM_LAYOUT_WT 1u
int dsm_lf_batch_dev(const dsm_index* idx, const uint8_t* d_c, const uint64_t* d_i, uint64_t* d_out, size_t k,
                     unsigned layout, void* stream);
/* getL(i) = BWT[i] -- TextCollection::getL, FMIndex.h:99-102.  Host pointers. */
int dsm_getl_batch(const dsm_index* idx, const uint64_t* i, uint8_t* out, size_t k, void* stream);
/* metaenumerate --check (metaenumerate.cpp:93-127): sum over c of |LF-interval(c)|; *total == n when sane. */
int dsm_index_check(const dsm_index* idx, uint64_t* total);

/* ------------------------------------------------------------------------------------------------
 * Counters every enumeration exports: work as the REFERENCE would have executed it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsm_stats {
    uint64_t reported;       /* emitted trie nodes = '(' tokens (EnumerateQuery.cpp:209), summed over local samples */
    uint64_t lf_steps;       /* TextCollection::LF calls the reference would make */
    uint64_t rank_ops;       /* BitRank::rank calls the reference would make (17 algorithmic bytes each) */
    uint64_t union_nodes;    /* merged trie nodes (metaserver total_paths) */
    uint64_t tuples;         /* printed tuples (metaserver total_output) */
    uint64_t pairs;          /* printed id:freq pairs (metaserver total_occs) */
    uint64_t candidates;     /* tuples handed to the host for the exact entropy test */
    uint64_t levels;         /* frontier levels processed */
    uint64_t max_frontier;   /* widest level */
    uint64_t expand_launches;
    double expand_ms;        /* HIP-event time summed over expand (LF-step) kernel launches */
    double device_ms;        /* HIP-event time of everything enqueued for the call */
    double host_ms;          /* host post-processing (exact entropy, tuple assembly) */
    uint64_t pair_order_exact; /* 1 when id:freq order and FP summation order follow the reference bit for bit */
    uint64_t splits;         /* prefixes that were split into longer ones because a level did not fit the device buffers */
    /* exact memory work of the LF-step (expand) kernel, counted by the kernel itself (ABI version 2) */
    uint64_t index_lines;    /* 64-byte index blocks it fetched */
    uint64_t records_read;   /* frontier records it read (nodes present in the sample) */
    uint64_t record_bytes;   /* bytes of frontier records it read and wrote (compact 16-byte words, or the wide fields) */
    uint64_t expand_slots;   /* threads it ran = frontier nodes x local samples (absent nodes only cost their handle and column entry) */
    uint64_t expand_column_bytes; /* bytes of exchange column it wrote */
    /* bytes this rank put on / took from the links in the per-level exchange (all-gather: its column out, the others' in;
     * owner mode: columns to the owner, child masks back) */
    uint64_t exchange_bytes_sent;
    uint64_t exchange_bytes_received;
} dsm_stats;

/* ------------------------------------------------------------------------------------------------
 * Single-sample enumeration.  Replaces EnumerateQuery::enumerate (EnumerateQuery.cpp:9-290) plus the
 * ClientSocket encoders (ClientSocket.h:12-46): the sink receives the exact bytes the reference client
 * writes to its socket AFTER the 'S' name '.' handshake, i.e. the node grammar
 *     node := '(' sym node* varint(freq) ['R' varint(count)]{depth<=6} leftchar ')'
 * in stream order, in one or more consecutive pieces.
 * ---------------------------------------------------------------------------------------------- */
typedef int (*dsm_byte_sink)(void* ctx, const uint8_t* bytes, size_t n);

int dsm_enumerate(const dsm_index* idx, const char* prefix, uint32_t fmin, uint32_t maxdepth,
                  dsm_byte_sink sink, void* ctx, dsm_stats* stats);

/* ------------------------------------------------------------------------------------------------
 * Multi-sample mining = enumerate + merge + entropy filter.  Replaces, for one prefix, every client's
 * EnumerateQuery plus metaserver's traverse() (metaserver.cpp:269-486).  Collective across ranks.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsm_tuple_batch {
    uint64_t ntuples;
    const uint32_t* path_off;  /* ntuples+1 offsets into path_bytes */
    const char* path_bytes;    /* substrings, A/C/G/T, not NUL terminated */
    const double* entropy;     /* exactly metaserver.cpp:389 (host double, glibc log) */
    const uint32_t* pair_off;  /* ntuples+1 offsets into ids/freqs */
    const uint32_t* ids;       /* sample ids in the reference's print order */
    const uint64_t* freqs;
} dsm_tuple_batch;

/* Receives tuples in the reference's output order (post-order, metaserver.cpp:467-485): batches of consecutive tuples, offsets starting
 * at 0, as many calls per prefix as the emitter finds convenient (a prefix leaves in chunks of about a million tuples or more; a chunk
 * in which a few tuples failed the entropy test on the host leaves as the runs between them).  The arrays are valid during the call. */
typedef int (*dsm_tuple_sink)(void* ctx, const dsm_tuple_batch* batch);

/* One exchange per frontier level (per frontier node and local sample: the node's frequency in 2 bytes -- 4 or 8 on the
 * few top levels where frequencies reach 65535 -- and one byte with the four "child survives fmin" bits and its left-char
 * code, behind a 16-byte header per rank): every rank contributes bytes_per_rank bytes at sendbuf and must end
 * up with world_size * bytes_per_rank bytes at recvbuf, rank-major (ncclAllGather semantics).  Device
 * pointers; must be ordered after prior work on `stream` and before later work on it. */
typedef int (*dsm_allgather_fn)(void* ctx, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, void* stream);

/* A ready-made dsm_allgather_fn over RCCL for hosts with one process per GPU (bench.py under torch.distributed.run, MPI):
 * ncclAllGather on the engine's stream straight from the library's per-level callback (metaenumerate.cpp:268-309 fans out over
 * sockets; this is its replacement between GPUs).  librccl is loaded at run time; the host carries the id from rank 0 to the
 * others (any channel) and passes dsm_rccl_allgather with the dsm_rccl* as allgather_ctx.  Several prefix lanes of a process
 * take one communicator each and share a gate, which makes them enqueue their collectives in the same order on every rank:
 * begin(lane) before and retire(lane) after a lane's run, reset between runs.  (A miner's creation runs collectives too: create
 * the miners one after the other, in the same order on every rank, and attach the gate afterwards.) */
#define DSM_RCCL_ID_BYTES 128
typedef struct dsm_rccl dsm_rccl;
typedef struct dsm_rccl_gate dsm_rccl_gate;
int dsm_rccl_unique_id(uint8_t* id /* DSM_RCCL_ID_BYTES */);
int dsm_rccl_create(const uint8_t* id, int world_size, int rank, int device, dsm_rccl** out);
int dsm_rccl_attach_gate(dsm_rccl* c, dsm_rccl_gate* gate /* NULL detaches */, int lane);  /* after the miners of all lanes exist */
int dsm_rccl_allgather(void* comm, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, void* stream);
void dsm_rccl_destroy(dsm_rccl* c);
int dsm_rccl_gate_create(int nlanes, dsm_rccl_gate** out);
void dsm_rccl_gate_begin(dsm_rccl_gate* g, int lane);
void dsm_rccl_gate_retire(dsm_rccl_gate* g, int lane);
void dsm_rccl_gate_reset(dsm_rccl_gate* g);
void dsm_rccl_gate_destroy(dsm_rccl_gate* g);

/* The reference's own partition between GPUs: ONE server per k-mer prefix merges every sample's stream
 * (wrapper-SLURM/example-server.sh:27-41, metaserver.cpp:682-739), every client has one connection per prefix
 * (metaenumerate.cpp:268-309).  With owner_mode the prefix of a miner is merged by rank owner_rank alone: per frontier level the
 * other ranks SEND it their columns (gather) and get back only the union's child masks, 4 bits per node, with the level's width
 * and frequency class (broadcast); reduce, scan, advance, reader-set orders, output predicates and candidate store run on the
 * owner only.  A host keeps one miner per owner alive (lane j: owner_rank j, prefixes j, j + world, ...), so every rank is the
 * server of one lane and a client in the others.
 *   gather: rank root must end up with world_size * bytes_per_rank bytes at recvbuf, rank-major, its own part included (sendbuf
 *           on root is its own contribution); the other ranks only send.
 *   bcast:  bytes at buf on root reach buf on every rank.
 * Device pointers, ordered on `stream` like dsm_allgather_fn.
 * Failures.  An owner that has to give a prefix up after it has taken a level's columns -- a failed call, a sink that refuses, a
 * limit -- still answers: its next broadcast starts with an abort word, every client of the prefix returns DSM_E_SINK ("the prefix's
 * owner failed"), the owner returns its own error; an arena that is too small is announced the same way and makes every rank split the
 * prefix (DSM_E_CAPACITY inside, invisible outside).  A CLIENT that fails between two levels cannot tell the owner, which then waits
 * in its gather: the host of a rank whose dsm_miner_* call returned an error other than DSM_E_CAPACITY in owner mode must tear down the
 * communicator (or the process), as with any collective library. */
typedef int (*dsm_gather_fn)(void* ctx, int root, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, void* stream);
typedef int (*dsm_bcast_fn)(void* ctx, int root, void* buf, size_t bytes, void* stream);
/* the same over RCCL (ncclSend / ncclRecv in one group; ncclBroadcast), ctx = the dsm_rccl* */
/* for hosts that stage a device buffer of the library through host memory (backends without device-to-device transport):
 * synchronous copies ordered after / before the work on `stream` */
int dsm_copy_from_device(void* host_dst, const void* device_src, size_t bytes, void* stream);
int dsm_copy_to_device(void* device_dst, const void* host_src, size_t bytes, void* stream);
int dsm_rccl_gather(void* comm, int root, const void* sendbuf, void* recvbuf, size_t bytes_per_rank, void* stream);
int dsm_rccl_bcast(void* comm, int root, void* buf, size_t bytes, void* stream);

typedef struct dsm_params {
    const char* prefix;      /* enforced path (hostinfo third column, metaenumerate.cpp:216); "" = whole trie */
    uint32_t fmin;           /* metaenumerate --fmin (default 10, metaenumerate.cpp:141) */
    uint32_t maxdepth;       /* metaenumerate --maxdepth (default ~0u) */
    uint32_t pmin;           /* metaserver -P (default 2, metaserver.cpp:126) */
    uint32_t pmax;           /* metaserver --pmax (0 = no limit) */
    uint32_t mindepth;       /* metaserver -m */
    double emin;             /* metaserver -e */
    double emax;             /* metaserver -E (mandatory there; <= 0 disables the entropy test, metaserver.cpp:413) */
    /* distribution: sample id of local index j is rank*nlocal + j; all ranks pass the same nlocal */
    uint32_t world_size;     /* 0 or 1 = single process */
    uint32_t rank;
    dsm_allgather_fn allgather;
    void* allgather_ctx;
    void* exchange_send;     /* optional caller-owned device buffers (e.g. torch tensors) of exchange_bytes */
    void* exchange_recv;     /*   recv must hold 2 * world_size * exchange_bytes (two halves, levels alternate) */
    uint64_t exchange_bytes;
    uint64_t arena_bytes;    /* device scratch budget; 0 = pick from free memory */
    uint32_t wide;           /* 1 = force 64-bit positions (all ranks must agree); 0 = from local index sizes */
    uint32_t emit_owner_only;/* multi-rank dsm_miner_mine_many: prefix k is filtered and emitted only by rank k % world_size
                                (every rank still expands its own samples for every prefix); 0 = every rank emits everything */
    void* stream;
    /* owner mode (see dsm_gather_fn): all ranks pass the same owner_rank; allgather is still needed (creation-time agreement) */
    uint32_t owner_mode;
    uint32_t owner_rank;
    dsm_gather_fn gather;
    dsm_bcast_fn bcast;
    void* owner_ctx;         /* ctx of gather and bcast */
} dsm_params;

void dsm_params_default(dsm_params* p);

int dsm_mine(dsm_index* const* idx, int nlocal, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);

/* ------------------------------------------------------------------------------------------------
 * Server side of the wire protocol.  Replaces TrieReader (TrieReader.h:32-106) + metaserver's traverse() for streams
 * that were produced elsewhere, e.g. by unmodified reference clients over TCP: dsm_trie_parse checks one connection's
 * bytes (everything after the 'S' name '.' handshake) with the reference's token rules and R checksums and uploads
 * the trie; dsm_merge merges d of them (sample id = position in the array) with the same kernels dsm_mine uses and
 * delivers the tuples the reference server would print.  p->fmin / maxdepth / prefix / world_size are ignored.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsm_trie dsm_trie;
int dsm_trie_parse(const uint8_t* bytes, size_t n, int device, dsm_trie** out);
void dsm_trie_free(dsm_trie* t);
/* The same decode fed piece by piece, as the bytes of the connection arrive (the reference's server reads its sockets token by
 * token while merging, metaserver.cpp:682-728, TrieReader.h:32-106): the library keeps a window of every level on the host and
 * moves what can no longer change to the card, so the caller never holds a whole stream.  _end checks that the stream is
 * complete, returns the trie and releases the handle (also when it fails); _abort releases it without a trie. */
typedef struct dsm_trie_stream dsm_trie_stream;
int dsm_trie_stream_begin(int device, dsm_trie_stream** out);
int dsm_trie_stream_feed(dsm_trie_stream* s, const uint8_t* bytes, size_t n);
int dsm_trie_stream_end(dsm_trie_stream* s, dsm_trie** out);
void dsm_trie_stream_abort(dsm_trie_stream* s);
uint64_t dsm_trie_nodes(const dsm_trie* t);
int dsm_merge(dsm_trie* const* tries, int n, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);
/* One metaserver: nsamples connections whose streams are merged WHILE they arrive (metaserver.cpp:682-739: the reference reads its
 * sockets token by token inside traverse(), holds no stream and prints a node as soon as every client has closed it).  Every client
 * of a server enforces the same prefix (metaenumerate.cpp:268-309, EnumerateQuery.cpp:240-290); prefix_len is its length (the server
 * wrapper knows it: one server per prefix, wrapper-SLURM/example-server.sh:27-41).  The subtree of a node of depth prefix_len + 1 +
 * unit_extra (unit_extra = 0..4: 4, 16, .. 1024 units at most) is merged and its tuples delivered as soon as every connection has
 * closed it, gone past it or ended -- while the later subtrees are still being received -- and leaves the device; a node between the
 * prefix and the units is delivered when its last child has been; the nodes of the prefix itself, which close last, follow at the end:
 * the sink sees the tuples in the reference's order, the card holds the subtrees in flight instead of nsamples complete streams.  A
 * stream that is not a single path down to depth prefix_len is refused (DSM_E_FORMAT).  prefix_len < 0: the streams are kept and merged by dsm_server_finish
 * (dsm_trie_stream_* + dsm_merge).  _feed / _end: one thread per sample (the connection's reader), any number of samples side by
 * side; the sink is called from a thread of the library.  sample = the id the reference gives the connection (position of its name
 * in the server's list).  _finish: after every connection has ended; waits for what is left and returns the totals. */
typedef struct dsm_server dsm_server;
int dsm_server_create(int nsamples, int device, int prefix_len, int unit_extra, const dsm_params* p, dsm_tuple_sink sink, void* ctx, dsm_server** out);
int dsm_server_feed(dsm_server* s, int sample, const uint8_t* bytes, size_t n);
int dsm_server_end(dsm_server* s, int sample);
int dsm_server_finish(dsm_server* s, dsm_stats* stats);
/* subtrees merged before the last connection ended ... so far; *peak_unit_nodes: nodes (all samples) of the largest one */
uint64_t dsm_server_units(const dsm_server* s, uint64_t* peak_unit_nodes);
void dsm_server_destroy(dsm_server* s);

/* Persistent form of the two calls above: device buffers are allocated once and reused for every
 * prefix (the reference keeps one EnumerateQuery + socket per prefix alive for the whole run,
 * metaenumerate.cpp:268-309).  p->prefix is ignored at creation; fmin/maxdepth/pmin/... are fixed.
 * stream_mode != 0: single local index, dsm_miner_enumerate delivers wire bytes;
 * stream_mode == 0: dsm_miner_mine delivers tuples. */
typedef struct dsm_miner dsm_miner;
int dsm_miner_create(dsm_index* const* idx, int nlocal, const dsm_params* p, int stream_mode, dsm_miner** out);
int dsm_miner_mine(dsm_miner* m, const char* prefix, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);
/* Several prefixes in one call: tuples arrive prefix by prefix in the given order (each prefix in the reference's
 * post-order); the host-side exact-entropy pass of prefix k overlaps the GPU expansion of prefix k+1.  The sink is
 * called from a library thread.  stats are summed over the prefixes. */
int dsm_miner_mine_many(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_tuple_sink sink, void* ctx, dsm_stats* stats);
/* The same, with the tuples as the reference server prints them (metaserver.cpp:472-484: "path %f id:freq ...\n"): the sink receives
 * consecutive pieces of that text, in order, from a library thread.  The exact entropy (metaserver.cpp:366-389, from the host's own
 * libm tables; frequencies beyond the tables are computed by the host), the emin / emax test, the lines' lengths and the lines are
 * computed on the card from the tuple arrays; only the text crosses the bus (about 0.6 of the binary batches' bytes).  The bytes are
 * those of dsm_format_batch over dsm_miner_mine_many's batches. */
typedef int (*dsm_text_sink)(void* ctx, const char* text, size_t len);
int dsm_miner_mine_text(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_text_sink sink, void* ctx, dsm_stats* stats);
int dsm_miner_enumerate(dsm_miner* m, const char* prefix, dsm_byte_sink sink, void* ctx, dsm_stats* stats);
/* Several prefixes in one call (the reference client walks its prefixes one after the other, one connection each,
 * metaenumerate.cpp:268-309): the sink receives (index of the prefix, bytes, n) for consecutive pieces of that prefix's stream
 * and one call with bytes == NULL, n == 0 when the prefix is complete; prefixes arrive in the given order.  The bytes of
 * prefix k cross PCIe while the GPU works on prefix k+1.  The sink is called from a library thread. */
typedef int (*dsm_prefix_byte_sink)(void* ctx, int prefix_index, const uint8_t* bytes, size_t n);
int dsm_miner_enumerate_many(dsm_miner* m, const char* const* prefixes, int nprefix, dsm_prefix_byte_sink sink, void* ctx, dsm_stats* stats);
void dsm_miner_destroy(dsm_miner* m);

/* metaserver's printf formatting of a batch (metaserver.cpp:472-484): "path %f id:freq ...\n".
 * *text is malloc'd; release with dsm_free. */
int dsm_format_batch(const dsm_tuple_batch* batch, char** text, size_t* len);
void dsm_free(void* p);
/* The same bytes from the GPU (metaserver.cpp:472-484 is 5x the mining pass when one host thread per core does it): one thread per
 * tuple measures its line, a scan places the lines, one thread per tuple writes.  "%f" is exact: the double's significand times 10^6
 * as a 128-bit integer, shifted by the binary exponent, rounded half to even -- what glibc's printf prints; a batch with a value of
 * 2^40 or more, an infinity or a NaN in its entropy column (never an entropy) is formatted by dsm_format_batch instead.
 * A formatter keeps its device and pinned buffers between calls; *text stays valid until the next call on the formatter (or its
 * destruction) and is NUL-terminated.  dsm_format_batch_dev: one-shot form, *text malloc'd, release with dsm_free. */
typedef struct dsm_formatter dsm_formatter;
int dsm_formatter_create(int device, dsm_formatter** out);
int dsm_formatter_format(dsm_formatter* f, const dsm_tuple_batch* batch, const char** text, size_t* len);
void dsm_formatter_destroy(dsm_formatter* f);
int dsm_format_batch_dev(const dsm_tuple_batch* batch, int device, char** text, size_t* len);

/* ------------------------------------------------------------------------------------------------
 * Index construction: the multi-string BWT the reference builder computes with incbwt
 * (builder.cpp:183-285, TextCollectionBuilder.cpp:65-152, incbwt/rlcsa_builder.cpp:35-78,165-179).  d_text: n bytes on the
 * device, the strings one after the other, each ending in a 0 byte; string k's terminator sorts as $_k with
 * $_0 < $_1 < ... < every other byte (incbwt/misc/utils.cpp:362-367).  d_bwt receives the n BWT bytes (0 where the suffix
 * starts a string).  At most 15 distinct non-zero bytes.  Sized for n beyond 2^32 (bounded-depth radix sort in batches).
 * pydsm.builder writes the .fmi v17 file from it (Huffman tree and serialisation identical to the reference's).
 * ---------------------------------------------------------------------------------------------- */
int dsm_bwt_build(const uint8_t* d_text, uint64_t n, uint8_t* d_bwt, int device, void* stream);

/* The rest of the reference `builder` (builder.cpp:329-472) behind the same boundary.
 * dsm_fmi_write: a BWT on the device -> <path> as .fmi version 17: the Huffman code table with the tie order of the reference's
 *   std::priority_queue (HuffWT.cpp:133-171), the pre-order wavelet tree with one BitRank per internal node (HuffWT.cpp:5-86:
 *   bit vector, Rs per 256 bits, Rb per word, BitRank.cpp:134-187) and FMIndex::save's container (FMIndex.cpp:155-217; no
 *   samples, names or text storage -- enumeration reads none).  Byte-identical to the reference builder's file of the same reads.
 * dsm_build_fasta: the whole tool: FASTA records (builder.cpp:203-262), per read upper-case / N-normalised (:60-104),
 *   text = reverse(read + '-' + revcomp(read)) + '\0' (:183-201), BWT by dsm_bwt_build, file by dsm_fmi_write.
 *   out_path is the file itself (the CLI appends ".fmi" like TextCollection::save). */
typedef struct dsm_build_info {
    uint64_t n;                /* BWT symbols */
    uint64_t number_of_texts;  /* reads */
    uint64_t max_text_length;  /* longest text with its terminator */
} dsm_build_info;
int dsm_fmi_write(const uint8_t* d_bwt, uint64_t n, uint32_t number_of_texts, uint64_t max_text_length, uint32_t samplerate, int device,
                  const char* path);
int dsm_build_fasta(const char* fasta_path, const char* out_path, uint32_t samplerate, int device, dsm_build_info* info);

/* ------------------------------------------------------------------------------------------------
 * Distance matrices of the tuple stream: the accumulation of wrapper-distance-matrix/smtxt2entropy.c
 * on the GPU.
 *   per tuple: normalised entropy (smtxt2entropy.c:128-145, evaluated on the host with the reference's expression
 *   and libm so that the bucket choice of :690-703 is exact), then for the chosen <max_entropy> bucket the pair
 *   counts and the three squared-distance sums of add() (:167-197) -- accumulated on the device.
 * The matrices returned by dsm_distmat_finish are cumulative exactly as the tool prints them (:722-752, :230-242).
 * The double sums add the same terms in a different order than the tool (tolerance: 1e-9 relative; counts exact).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dsm_distmat dsm_distmat;
int dsm_distmat_create(int device, uint32_t samples, const double* maxent, uint32_t nmaxent, uint32_t minfreq, dsm_distmat** out);
/* The tool's -S,--samplefile (run_to_sample[runs]: sample of every run id that may appear in the input; the matrices then
 * have max + 1 = `samples` rows; smtxt2entropy.c:101-104,385-423) and -N,--normalize (sizes[samples]: dataset sizes;
 * normalized_entropy :147-165, add_normalized :199-228 -- its lgamma matrix stays zero).  Either may be NULL. */
int dsm_distmat_create_ex(int device, uint32_t samples, const double* maxent, uint32_t nmaxent, uint32_t minfreq,
                          const int32_t* run_to_sample, uint32_t runs, const double* sizes, dsm_distmat** out);
void dsm_distmat_destroy(dsm_distmat* m);
/* -e,--entstep list of the tool (smtxt2entropy.c:258-285); returns the number of values written (<= cap) or < 0 */
int dsm_distmat_steps(double step, double* out, int cap);
/* a batch exactly as a dsm_tuple_sink receives it (paths and the entropy column are not used) */
int dsm_distmat_add(dsm_distmat* m, const dsm_tuple_batch* batch);
/* metaserver output lines "path entropy id:freq ...\n" (smtxt2entropy.c:84-125,656-680) */
int dsm_distmat_add_text(dsm_distmat* m, const char* text, size_t len);
/* maxent_sorted[nmaxent] (descending, the tool's matrix order), noutput[nmaxent], and four [nmaxent][samples][samples]
 * arrays; any pointer may be NULL.  May be called once. */
int dsm_distmat_finish(dsm_distmat* m, double* maxent_sorted, uint32_t* noutput, uint32_t* count, double* mlog, double* msqrt,
                       double* mlgamma);
/* the tool's four output files (count, log, sqrt, lgamma) as text; each malloc'd, release with dsm_free */
int dsm_distmat_format(uint32_t samples, uint32_t nmaxent, const double* maxent_sorted, const uint32_t* noutput, const uint32_t* count,
                       const double* mlog, const double* msqrt, const double* mlgamma, char* text[4]);

#ifdef __cplusplus
}
#endif
#endif /* DSMHIP_H_ */
