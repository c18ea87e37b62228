"""pydsm -- thin ctypes binding of libdsmhip.so (include/dsmhip.h).

Host-side mirror of the reference's interfaces for the substring-enumeration path:
  Index            ~ TextCollection::load / FMIndex (LF, getL, getLength)      FMIndex.h:68-102
  Index.enumerate  ~ EnumerateQuery::enumerate + ClientSocket encoders         EnumerateQuery.cpp:9-290
  mine             ~ all clients + metaserver traverse() for one prefix        metaserver.cpp:269-486
There is no CPU fallback: every call runs HIP kernels and raises DsmError when the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSM_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libdsmhip.so")  # (override: profiling variants)

LAYOUT_PLANES = 0
LAYOUT_WT = 1
OPEN_KEEP_WT = 1
MAXDEPTH_NONE = 0xFFFFFFFF


class DsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dsmhip error %d: %s" % (code, msg))
        self.code = code


class Code(C.Structure):
    _fields_ = [("count", C.c_uint64), ("bits", C.c_uint32), ("code", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("reported", C.c_uint64), ("lf_steps", C.c_uint64), ("rank_ops", C.c_uint64), ("union_nodes", C.c_uint64),
                ("tuples", C.c_uint64), ("pairs", C.c_uint64), ("candidates", C.c_uint64), ("levels", C.c_uint64),
                ("max_frontier", C.c_uint64), ("expand_launches", C.c_uint64), ("expand_ms", C.c_double),
                ("device_ms", C.c_double), ("host_ms", C.c_double), ("pair_order_exact", C.c_uint64), ("splits", C.c_uint64),
                ("index_lines", C.c_uint64), ("records_read", C.c_uint64), ("record_bytes", C.c_uint64), ("expand_slots", C.c_uint64),
                ("expand_column_bytes", C.c_uint64), ("exchange_bytes_sent", C.c_uint64), ("exchange_bytes_received", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TupleBatch(C.Structure):
    _fields_ = [("ntuples", C.c_uint64), ("path_off", C.POINTER(C.c_uint32)), ("path_bytes", C.c_void_p),
                ("entropy", C.POINTER(C.c_double)), ("pair_off", C.POINTER(C.c_uint32)), ("ids", C.POINTER(C.c_uint32)),
                ("freqs", C.POINTER(C.c_uint64))]


BYTE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
PREFIX_BYTE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
TUPLE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TupleBatch))
TEXT_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)


class Params(C.Structure):
    _fields_ = [("prefix", C.c_char_p), ("fmin", C.c_uint32), ("maxdepth", C.c_uint32), ("pmin", C.c_uint32),
                ("pmax", C.c_uint32), ("mindepth", C.c_uint32), ("emin", C.c_double), ("emax", C.c_double),
                ("world_size", C.c_uint32), ("rank", C.c_uint32), ("allgather", ALLGATHER), ("allgather_ctx", C.c_void_p),
                ("exchange_send", C.c_void_p), ("exchange_recv", C.c_void_p), ("exchange_bytes", C.c_uint64),
                ("arena_bytes", C.c_uint64), ("wide", C.c_uint32), ("emit_owner_only", C.c_uint32), ("stream", C.c_void_p),
                ("owner_mode", C.c_uint32), ("owner_rank", C.c_uint32), ("gather", GATHER), ("bcast", BCAST), ("owner_ctx", C.c_void_p)]


_lib = None


def lib():
    """Load libdsmhip.so (fails loudly if it was not built: `make -C dsm-framework_amd`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DsmError(-2, "libdsmhip.so not built (run `make -C dsm-framework_amd` or __graft_entry__.build())")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
        # Importing torch first makes libdsmhip.so bind to the copy torch uses, so device pointers, streams
        # and torch.distributed (RCCL) all live in the same runtime.  torch is plumbing here, not compute.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.dsm_last_error.restype = C.c_char_p
        L.dsm_abi_version.restype = C.c_int
        L.dsm_copy_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_rccl_gather.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_rccl_bcast.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_index_open_ex.argtypes = [C.c_char_p, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
        L.dsm_index_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_index_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]
        L.dsm_index_close.argtypes = [C.c_void_p]
        L.dsm_index_length.restype = C.c_uint64
        L.dsm_index_length.argtypes = [C.c_void_p]
        L.dsm_index_meta.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dsm_index_name.restype = C.c_char_p
        L.dsm_index_name.argtypes = [C.c_void_p]
        L.dsm_index_device.argtypes = [C.c_void_p]
        L.dsm_index_device_bytes.restype = C.c_uint64
        L.dsm_index_device_bytes.argtypes = [C.c_void_p]
        L.dsm_lf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_lf_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
        L.dsm_getl_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dsm_index_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.dsm_enumerate.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, BYTE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_params_default.argtypes = [C.POINTER(Params)]
        L.dsm_mine.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), TUPLE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_miner_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_miner_mine.argtypes = [C.c_void_p, C.c_char_p, TUPLE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_miner_mine_many.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, TUPLE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_miner_enumerate.argtypes = [C.c_void_p, C.c_char_p, BYTE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_miner_enumerate_many.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, PREFIX_BYTE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_miner_destroy.argtypes = [C.c_void_p]
        L.dsm_trie_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_trie_free.argtypes = [C.c_void_p]
        L.dsm_rccl_unique_id.argtypes = [C.c_void_p]
        L.dsm_rccl_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_rccl_attach_gate.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.dsm_rccl_destroy.argtypes = [C.c_void_p]
        L.dsm_rccl_destroy.restype = None
        L.dsm_rccl_gate_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        for fn in (L.dsm_rccl_gate_begin, L.dsm_rccl_gate_retire):
            fn.argtypes = [C.c_void_p, C.c_int]
            fn.restype = None
        for fn in (L.dsm_rccl_gate_reset, L.dsm_rccl_gate_destroy):
            fn.argtypes = [C.c_void_p]
            fn.restype = None
        L.dsm_trie_stream_begin.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_trie_stream_feed.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.dsm_trie_stream_end.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.dsm_trie_stream_abort.argtypes = [C.c_void_p]
        L.dsm_trie_stream_abort.restype = None
        L.dsm_trie_nodes.restype = C.c_uint64
        L.dsm_trie_nodes.argtypes = [C.c_void_p]
        L.dsm_merge.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(Params), TUPLE_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_server_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Params), TUPLE_SINK, C.c_void_p, C.POINTER(C.c_void_p)]
        L.dsm_server_feed.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        L.dsm_server_end.argtypes = [C.c_void_p, C.c_int]
        L.dsm_server_finish.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.dsm_server_units.restype = C.c_uint64
        L.dsm_server_units.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.dsm_server_destroy.argtypes = [C.c_void_p]
        L.dsm_server_destroy.restype = None
        L.dsm_format_batch.argtypes = [C.POINTER(TupleBatch), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.dsm_format_batch_dev.argtypes = [C.POINTER(TupleBatch), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.dsm_formatter_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.dsm_formatter_format.argtypes = [C.c_void_p, C.POINTER(TupleBatch), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.dsm_formatter_destroy.argtypes = [C.c_void_p]
        L.dsm_formatter_destroy.restype = None
        L.dsm_miner_mine_text.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, TEXT_SINK, C.c_void_p, C.POINTER(Stats)]
        L.dsm_free.argtypes = [C.c_void_p]
        L.dsm_index_offload.argtypes = [C.c_void_p]
        L.dsm_index_reload.argtypes = [C.c_void_p, C.c_void_p]
        L.dsm_index_resident.argtypes = [C.c_void_p]
        L.dsm_distmat_create.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.dsm_distmat_create_ex.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.POINTER(C.c_void_p)]
        L.dsm_distmat_destroy.argtypes = [C.c_void_p]
        L.dsm_distmat_steps.argtypes = [C.c_double, C.c_void_p, C.c_int]
        L.dsm_distmat_add.argtypes = [C.c_void_p, C.POINTER(TupleBatch)]
        L.dsm_distmat_add_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.dsm_distmat_finish.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.dsm_distmat_format.argtypes = [C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise DsmError(rc, lib().dsm_last_error().decode(errors="replace"))


def probe(path):
    """dsm_index_probe: the checks of opening an index (header, code table, tree shape) on the host alone; returns the BWT length"""
    n = C.c_uint64(0)
    _check(lib().dsm_index_probe(os.fsencode(path), C.byref(n)))
    return int(n.value)


class Index:
    """HBM-resident FM-index opened from a reference-format .fmi file."""

    def __init__(self, path, device=0, keep_wt=False):
        self.h = C.c_void_p()
        _check(lib().dsm_index_open_ex(os.fsencode(path), device, OPEN_KEEP_WT if keep_wt else 0, C.byref(self.h)))
        self.device = int(device)
        self.n = lib().dsm_index_length(self.h)
        self.name = lib().dsm_index_name(self.h).decode()
        self.device = device

    def close(self):
        if self.h:
            lib().dsm_index_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def meta(self):
        Cc = np.zeros(256, np.uint64)
        codes = (Code * 256)()
        _check(lib().dsm_index_meta(self.h, Cc.ctypes.data, codes))
        return Cc, codes

    def device_bytes(self):
        return lib().dsm_index_device_bytes(self.h)

    def offload(self):
        """Give the blocks' HBM back; they stay in pinned host memory (dsm_index_offload)."""
        _check(lib().dsm_index_offload(self.h))

    def reload(self, stream=None):
        _check(lib().dsm_index_reload(self.h, stream))

    @property
    def resident(self):
        return bool(lib().dsm_index_resident(self.h))

    def lf_batch(self, c, i):
        """LF(c[k], i[k]) for host arrays (FMIndex.h:84-90)."""
        c = np.ascontiguousarray(c, np.uint8)
        i = np.ascontiguousarray(i, np.uint64)
        out = np.zeros(len(c), np.uint64)
        _check(lib().dsm_lf_batch(self.h, c.ctypes.data, i.ctypes.data, out.ctypes.data, len(c), None))
        return out

    def lf_batch_dev(self, d_c, d_i, d_out, k, layout=LAYOUT_PLANES, stream=None):
        _check(lib().dsm_lf_batch_dev(self.h, d_c, d_i, d_out, k, layout, stream))

    def getl_batch(self, i):
        i = np.ascontiguousarray(i, np.uint64)
        out = np.zeros(len(i), np.uint8)
        _check(lib().dsm_getl_batch(self.h, i.ctypes.data, out.ctypes.data, len(i), None))
        return out

    def check(self):
        """metaenumerate --check (metaenumerate.cpp:93-127): returns the interval-size total; == n when sane."""
        t = C.c_uint64(0)
        _check(lib().dsm_index_check(self.h, C.byref(t)))
        return t.value

    def enumerate(self, prefix, fmin=10, maxdepth=MAXDEPTH_NONE, with_header=True):
        """Wire bytes of one client connection: b'S' name b'.' + node grammar (EnumerateQuery.cpp).  -> (bytes, stats)"""
        chunks = []

        err = []

        def sink(ctx, p, n):
            try:
                chunks.append(C.string_at(p, n))
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                err.append(e)
                return 1

        cb = BYTE_SINK(sink)
        st = Stats()
        _check_sink(lib().dsm_enumerate(self.h, prefix.encode(), fmin, maxdepth, cb, None, C.byref(st)), err)
        body = b"".join(chunks)
        if with_header:
            body = b"S" + self.name.encode() + b"." + body  # metaenumerate.cpp:285-286
        return body, st


def default_params():
    p = Params()
    lib().dsm_params_default(C.byref(p))
    return p


def _make_params(fmin, maxdepth, pmin, pmax, mindepth, emin, emax, world_size, rank, allgather, exchange, arena_bytes, wide,
                 stream, keep, emit_owner_only=0, owner_rank=None, owner_exchange=None):
    """owner_rank (with owner_exchange): the prefixes of this miner are merged by that rank alone (dsm_params.owner_mode);
    owner_exchange = an RcclComm (the library's ncclSend/Recv + ncclBroadcast) or an object with .gather(root, send_ptr, recv_ptr,
    nbytes, stream) and .bcast(root, ptr, nbytes, stream)."""
    p = default_params()
    p.fmin, p.maxdepth, p.pmin, p.pmax, p.mindepth = fmin, maxdepth, pmin, pmax, mindepth
    p.emin, p.emax = emin, emax
    p.world_size, p.rank = world_size, rank
    p.arena_bytes = arena_bytes
    p.wide = wide
    p.emit_owner_only = emit_owner_only
    p.stream = stream
    if isinstance(allgather, RcclComm):  # the library's own RCCL exchange: a C function and its context, no Python per level
        p.allgather = C.cast(lib().dsm_rccl_allgather, ALLGATHER)
        p.allgather_ctx = allgather.h
        keep.append(allgather)
    elif allgather is not None:
        def _ag(ctx, s, r, n, st):
            try:
                allgather(s, r, n, st)
                return 0
            except Exception:  # noqa: BLE001 - must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        cb_ag = ALLGATHER(_ag)
        keep.append(cb_ag)
        p.allgather = cb_ag
    if exchange is not None:
        p.exchange_send, p.exchange_recv, p.exchange_bytes = exchange
    if owner_rank is not None:
        p.owner_mode, p.owner_rank = 1, int(owner_rank)
        if isinstance(owner_exchange, RcclComm):
            p.gather = C.cast(lib().dsm_rccl_gather, GATHER)
            p.bcast = C.cast(lib().dsm_rccl_bcast, BCAST)
            p.owner_ctx = owner_exchange.h
            keep.append(owner_exchange)
        else:
            def _ga(ctx, root, s_, r_, n, st):
                try:
                    owner_exchange.gather(root, s_, r_, n, st)
                    return 0
                except Exception:  # noqa: BLE001 - must not unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1

            def _bc(ctx, root, b, n, st):
                try:
                    owner_exchange.bcast(root, b, n, st)
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            cb_ga, cb_bc = GATHER(_ga), BCAST(_bc)
            keep.extend([cb_ga, cb_bc])
            p.gather, p.bcast = cb_ga, cb_bc
    return p


class RcclGate:
    """Turn-taking of the prefix lanes of one process (dsm_rccl_gate): their collectives are enqueued in the same order on
    every rank.  begin(lane) / retire(lane) around a lane's run, reset() between runs."""

    def __init__(self, nlanes):
        self.h = C.c_void_p()
        _check(lib().dsm_rccl_gate_create(nlanes, C.byref(self.h)))

    def begin(self, lane):
        lib().dsm_rccl_gate_begin(self.h, lane)

    def retire(self, lane):
        lib().dsm_rccl_gate_retire(self.h, lane)

    def reset(self):
        lib().dsm_rccl_gate_reset(self.h)

    def close(self):
        if self.h:
            lib().dsm_rccl_gate_destroy(self.h)
            self.h = C.c_void_p()


class RcclComm:
    """One RCCL communicator of the library's native exchange (dsm_rccl_*): pass it as `allgather=` to Miner / mine.
    Rank 0 makes the id with RcclComm.unique_id() and the host carries it to the other ranks (128 bytes, any channel);
    creating the communicator is collective over its ranks."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib().dsm_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid, world_size, rank, device=0):
        if len(uid) != 128:
            raise ValueError("an RCCL id has 128 bytes")
        self.h = C.c_void_p()
        self.gate = None
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _check(lib().dsm_rccl_create(buf, world_size, rank, device, C.byref(self.h)))

    def attach_gate(self, gate, lane):
        """From now on this communicator's collectives wait for their lane's turn (call it once the miners of all lanes exist:
        their creation runs collectives one lane after the other)."""
        _check(lib().dsm_rccl_attach_gate(self.h, gate.h if gate is not None else None, lane))
        self.gate = gate

    def close(self):
        if self.h:
            lib().dsm_rccl_destroy(self.h)
            self.h = C.c_void_p()


class Formatter:
    """metaserver's output text of tuple batches from the GPU (dsm_formatter_*): the bytes of dsm_format_batch's snprintf loop."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _check(lib().dsm_formatter_create(device, C.byref(self.h)))

    def format(self, batch):
        """batch: a TupleBatch (or a pointer to one, as a sink receives it) -> bytes"""
        t = C.c_void_p()
        n = C.c_size_t(0)
        _check(lib().dsm_formatter_format(self.h, batch, C.byref(t), C.byref(n)))
        return C.string_at(t, n.value) if n.value else b""

    def close(self):
        if self.h:
            lib().dsm_formatter_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


TEXT_ON_HOST = bool(os.environ.get("DSM_TEXT_HOST"))  # the tuple text of mine()/merge()/Server from the host's snprintf loop instead of the GPU


def _tuple_sink(out, text, on_batch, err, device=0):
    """err: a list that receives the exception a callback raised.  An exception must not unwind through C (ctypes would print
    it and report success): the sink returns 1 instead, the library fails with DSM_E_SINK and the caller re-raises.
    The text comes from the device formatter (dsm_formatter_format), one formatter per sink, created at the first batch."""
    fmt = []

    def sink(ctx, b):
        try:
            if on_batch is not None:
                on_batch(b.contents)
            if text:
                if TEXT_ON_HOST:
                    t = C.c_void_p()
                    n = C.c_size_t(0)
                    if lib().dsm_format_batch(b, C.byref(t), C.byref(n)) != 0:
                        return 1
                    out.append(C.string_at(t, n.value))
                    lib().dsm_free(t)
                else:
                    if not fmt:
                        fmt.append(Formatter(device))
                    out.append(fmt[0].format(b))
            return 0
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            return 1
    cb = TUPLE_SINK(sink)
    cb._dsm_formatters = fmt  # (closed by _close_sink after the call that used the sink)
    return cb


def _close_sink(cb):
    for f in getattr(cb, "_dsm_formatters", []):
        f.close()


def _check_sink(rc, err):
    if err:
        raise err[0]
    _check(rc)


class Miner:
    """Persistent enumeration state over a fixed set of local indexes (device buffers allocated once).

    stream_mode=False: .mine(prefix) -> (tuple text, Stats);  stream_mode=True (one index): .enumerate(prefix)."""

    def __init__(self, indexes, fmin=10, maxdepth=MAXDEPTH_NONE, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0,
                 world_size=1, rank=0, allgather=None, exchange=None, arena_bytes=0, wide=0, stream=None, stream_mode=False,
                 emit_owner_only=False, owner_rank=None, owner_exchange=None):
        self._keep = []
        self.indexes = list(indexes)
        p = _make_params(fmin, maxdepth, pmin, pmax, mindepth, emin, emax, world_size, rank, allgather, exchange, arena_bytes,
                         wide, stream, self._keep, 1 if emit_owner_only else 0, owner_rank, owner_exchange)
        hs = (C.c_void_p * len(indexes))(*[ix.h for ix in indexes])
        self.h = C.c_void_p()
        _check(lib().dsm_miner_create(hs, len(indexes), C.byref(p), 1 if stream_mode else 0, C.byref(self.h)))

    def _device(self):
        return int(getattr(self.indexes[0], "device", 0)) if self.indexes else 0

    def mine_text(self, prefixes, on_text=None, on_raw=None):
        """dsm_miner_mine_text: the reference server's stdout for the prefixes, formatted on the card; on_text(bytes) receives the
        pieces as they arrive (default: collected and returned); on_raw(address, length) instead sees the library's buffer without
        a copy (valid during the call only).  -> (bytes or None, Stats)"""
        out, err = [], []

        def sink(ctx, p, n):
            try:
                if on_raw is not None:
                    on_raw(p, n)
                    return 0
                piece = C.string_at(p, n)
                if on_text is not None:
                    on_text(piece)
                else:
                    out.append(piece)
                return 0
            except BaseException as e:  # noqa: BLE001
                err.append(e)
                return 1
        cb = TEXT_SINK(sink)
        st = Stats()
        arr = (C.c_char_p * len(prefixes))(*[p.encode() for p in prefixes])
        _check_sink(lib().dsm_miner_mine_text(self.h, arr, len(prefixes), cb, None, C.byref(st)), err)
        return (b"".join(out) if on_text is None and on_raw is None else None), st

    def mine(self, prefix, text=True, on_batch=None):
        if text and on_batch is None and not TEXT_ON_HOST:
            return self.mine_text([prefix])
        out, err = [], []
        cb = _tuple_sink(out, text, on_batch, err, self._device())
        st = Stats()
        try:
            _check_sink(lib().dsm_miner_mine(self.h, prefix.encode(), cb, None, C.byref(st)), err)
        finally:
            _close_sink(cb)
        return (b"".join(out) if text else None), st

    def mine_many(self, prefixes, text=True, on_batch=None):
        """All prefixes in one call (host emission of prefix k overlaps GPU work on prefix k+1)."""
        if text and on_batch is None and not TEXT_ON_HOST:
            return self.mine_text(list(prefixes))
        out, err = [], []
        cb = _tuple_sink(out, text, on_batch, err, self._device())
        st = Stats()
        arr = (C.c_char_p * len(prefixes))(*[p.encode() for p in prefixes])
        try:
            _check_sink(lib().dsm_miner_mine_many(self.h, arr, len(prefixes), cb, None, C.byref(st)), err)
        finally:
            _close_sink(cb)
        return (b"".join(out) if text else None), st

    def enumerate(self, prefix, with_header=True, discard=False):
        chunks = []
        nbytes = [0]

        err = []

        def sink(ctx, p, n):
            try:
                nbytes[0] += n
                if not discard:
                    chunks.append(C.string_at(p, n))
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                err.append(e)
                return 1

        cb = BYTE_SINK(sink)
        st = Stats()
        _check_sink(lib().dsm_miner_enumerate(self.h, prefix.encode(), cb, None, C.byref(st)), err)
        if discard:
            return nbytes[0], st
        body = b"".join(chunks)
        if with_header:
            body = b"S" + self.indexes[0].name.encode() + b"." + body
        return body, st

    def enumerate_many(self, prefixes, with_header=True, discard=False, on_piece=None):
        """dsm_miner_enumerate_many: the wire streams of several prefixes, the bytes of one crossing PCIe while the GPU works
        on the next.  -> ([bytes per prefix] or [byte counts] with discard, stats summed over the prefixes)"""
        chunks = [[] for _ in prefixes]
        nbytes = [0] * len(prefixes)
        done = []
        err = []

        def sink(ctx, k, p, n):
            try:
                if not p:
                    done.append(k)
                    return 0
                nbytes[k] += n
                if on_piece is not None:
                    on_piece(k, C.string_at(p, n))
                elif not discard:
                    chunks[k].append(C.string_at(p, n))
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through C
                err.append(e)
                return 1

        cb = PREFIX_BYTE_SINK(sink)
        arr = (C.c_char_p * len(prefixes))(*[q.encode() for q in prefixes])
        st = Stats()
        _check_sink(lib().dsm_miner_enumerate_many(self.h, arr, len(prefixes), cb, None, C.byref(st)), err)
        if done != list(range(len(prefixes))):
            raise DsmError("dsm_miner_enumerate_many: prefixes completed as %r" % (done,))
        if discard or on_piece is not None:
            return nbytes, st
        head = (b"S" + self.indexes[0].name.encode() + b".") if with_header else b""
        return [head + b"".join(c) for c in chunks], st

    def close(self):
        if self.h:
            lib().dsm_miner_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def mine(indexes, prefix, fmin=10, maxdepth=MAXDEPTH_NONE, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0,
         world_size=1, rank=0, allgather=None, exchange=None, arena_bytes=0, wide=0, stream=None, text=True,
         on_batch=None):
    """Enumerate + merge + entropy filter for one prefix over the local indexes (collective when world_size > 1).

    Returns (metaserver-format text bytes or None, Stats).  `allgather(send_ptr, recv_ptr, nbytes, stream)` is
    called once per frontier level when world_size > 1.  `exchange` = (send_ptr, recv_ptr, nbytes) lets the caller
    own the exchange buffers (e.g. torch tensors handed to torch.distributed)."""
    keep = []
    p = _make_params(fmin, maxdepth, pmin, pmax, mindepth, emin, emax, world_size, rank, allgather, exchange, arena_bytes, wide,
                     stream, keep)
    p.prefix = prefix.encode()
    out, err = [], []
    cb = _tuple_sink(out, text, on_batch, err)
    hs = (C.c_void_p * len(indexes))(*[ix.h for ix in indexes])
    st = Stats()
    try:
        _check_sink(lib().dsm_mine(hs, len(indexes), C.byref(p), cb, None, C.byref(st)), err)
    finally:
        _close_sink(cb)
    return (b"".join(out) if text else None), st


class Trie:
    """One client connection's byte stream (what a reference metaenumerate sends), checked and uploaded (TrieReader.h:32-106).
    `stream` may include the b'S' name b'.' handshake; the sample name is then available as .name."""

    def __init__(self, stream, device=0, pieces=None):
        """pieces: feed the body through dsm_trie_stream_* in pieces of these sizes (cycled) instead of dsm_trie_parse."""
        self.name = None
        body = stream
        if stream[:1] == b"S" and b"." in stream:
            dot = stream.index(b".")
            self.name = stream[1:dot].decode(errors="replace")
            body = stream[dot + 1:]
        self.h = C.c_void_p()
        if pieces:
            ts = C.c_void_p()
            _check(lib().dsm_trie_stream_begin(device, C.byref(ts)))
            pos, k = 0, 0
            while pos < len(body):
                n = pieces[k % len(pieces)]
                rc = lib().dsm_trie_stream_feed(ts, body[pos:pos + n], min(n, len(body) - pos))
                if rc:
                    lib().dsm_trie_stream_abort(ts)
                    _check(rc)
                pos += n
                k += 1
            _check(lib().dsm_trie_stream_end(ts, C.byref(self.h)))  # (releases the handle, also when it fails)
        else:
            _check(lib().dsm_trie_parse(body, len(body), device, C.byref(self.h)))
        self.nodes = lib().dsm_trie_nodes(self.h)

    def close(self):
        if self.h:
            lib().dsm_trie_free(self.h)
            self.h = C.c_void_p()


def merge(tries, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0, arena_bytes=0, text=True, on_batch=None):
    """metaserver's traverse() over parsed client streams; tries[k] is sample id k.  -> (tuple text, Stats)"""
    keep = []
    p = _make_params(0, MAXDEPTH_NONE, pmin, pmax, mindepth, emin, emax, 1, 0, None, None, arena_bytes, 0, None, keep)
    out, err = [], []
    cb = _tuple_sink(out, text, on_batch, err)
    hs = (C.c_void_p * len(tries))(*[t.h for t in tries])
    st = Stats()
    try:
        _check_sink(lib().dsm_merge(hs, len(tries), C.byref(p), cb, None, C.byref(st)), err)
    finally:
        _close_sink(cb)
    return (b"".join(out) if text else None), st


class Server:
    """One metaserver (dsm_server_*): the streams of `nsamples` connections, merged while they arrive when prefix_len (the length of
    the prefix the clients enforce) is given -- the subtree of every node of depth prefix_len + 1 + unit_extra is merged and delivered
    as soon as all connections are past it -- or kept and merged by finish() (prefix_len=None).  feed(sample, bytes) takes the bytes of
    a connection after its handshake; tuples arrive in the reference's order."""

    def __init__(self, nsamples, prefix_len=None, unit_extra=0, device=0, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0, arena_bytes=0,
                 text=True, on_batch=None):
        self._keep = []
        p = _make_params(0, MAXDEPTH_NONE, pmin, pmax, mindepth, emin, emax, 1, 0, None, None, arena_bytes, 0, None, self._keep)
        self.out, self._err = [], []
        self._cb = _tuple_sink(self.out, text, on_batch, self._err)
        self.h = C.c_void_p()
        _check(lib().dsm_server_create(int(nsamples), device, -1 if prefix_len is None else int(prefix_len), int(unit_extra), C.byref(p),
                                       self._cb, None, C.byref(self.h)))

    def feed(self, sample, data):
        _check_sink(lib().dsm_server_feed(self.h, int(sample), data, len(data)), self._err)

    def end(self, sample):
        _check_sink(lib().dsm_server_end(self.h, int(sample)), self._err)

    def units(self):
        peak = C.c_uint64(0)
        n = lib().dsm_server_units(self.h, C.byref(peak))
        return int(n), int(peak.value)

    def finish(self):
        st = Stats()
        _check_sink(lib().dsm_server_finish(self.h, C.byref(st)), self._err)
        return b"".join(self.out), st

    def close(self):
        if self.h:
            lib().dsm_server_destroy(self.h)
            self.h = C.c_void_p()
            _close_sink(self._cb)


class DistMat:
    """Distance matrices of a tuple stream (wrapper-distance-matrix/smtxt2entropy.c), accumulated on the GPU.
    add(batch) takes the batches a tuple sink receives (use as on_batch=dm.add), add_text() takes metaserver output lines."""

    def __init__(self, samples, maxent=None, entstep=None, minfreq=0, device=0, run_to_sample=None, sizes=None):
        if (maxent is None) == (entstep is None):
            raise ValueError("give either maxent (list) or entstep")
        if entstep is not None:
            buf = (C.c_double * 1024)()
            n = lib().dsm_distmat_steps(float(entstep), buf, 1024)
            if n < 0:
                _check(n)
            maxent = list(buf[:n])
        self.samples, self.nm = int(samples), len(maxent)
        me = (C.c_double * self.nm)(*maxent)
        self.h = C.c_void_p()
        mp = (C.c_int32 * len(run_to_sample))(*run_to_sample) if run_to_sample else None   # the tool's -S file
        sz = (C.c_double * len(sizes))(*sizes) if sizes else None                         # the tool's -N file
        _check(lib().dsm_distmat_create_ex(device, self.samples, me, self.nm, minfreq, mp, len(run_to_sample) if run_to_sample else 0, sz,
                                           C.byref(self.h)))

    def add(self, batch):
        _check(lib().dsm_distmat_add(self.h, C.byref(batch)))

    def add_text(self, text):
        _check(lib().dsm_distmat_add_text(self.h, text, len(text)))

    def finish(self):
        """-> dict(maxent, noutput, count, log, sqrt, lgamma): cumulative matrices in the tool's print order."""
        nm, s = self.nm, self.samples
        me = np.zeros(nm, np.float64)
        nout = np.zeros(nm, np.uint32)
        cnt = np.zeros((nm, s, s), np.uint32)
        mats = [np.zeros((nm, s, s), np.float64) for _ in range(3)]
        _check(lib().dsm_distmat_finish(self.h, me.ctypes.data, nout.ctypes.data, cnt.ctypes.data, *[m.ctypes.data for m in mats]))
        return dict(maxent=me, noutput=nout, count=cnt, log=mats[0], sqrt=mats[1], lgamma=mats[2])

    @staticmethod
    def format(res):
        """The tool's four output files (count, log, sqrt, lgamma) as bytes."""
        nm, s = res["count"].shape[0], res["count"].shape[1]
        texts = (C.c_void_p * 4)()
        arrs = [np.ascontiguousarray(res[k]) for k in ("maxent", "noutput", "count", "log", "sqrt", "lgamma")]
        _check(lib().dsm_distmat_format(s, nm, *[a.ctypes.data for a in arrs], texts))
        out = [C.string_at(t) for t in texts]
        for t in texts:
            lib().dsm_free(t)
        return out

    def close(self):
        if self.h:
            lib().dsm_distmat_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
