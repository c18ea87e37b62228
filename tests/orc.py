"""ctypes binding of oracle/_build/liboracle.so -- the CPU oracle (test infrastructure only)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "_build", "liboracle.so")],
                           check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_last_error.restype = C.c_char_p
        L.orc_index_load.restype = C.c_void_p
        L.orc_index_load.argtypes = [C.c_char_p]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_index_length.restype = C.c_uint64
        L.orc_index_length.argtypes = [C.c_void_p]
        L.orc_index_meta.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_lf.restype = C.c_uint64
        L.orc_lf.argtypes = [C.c_void_p, C.c_uint, C.c_uint64]
        L.orc_lf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_getL.restype = C.c_uint
        L.orc_getL.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_bwt.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_enumerate.restype = C.c_void_p
        L.orc_enumerate.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_server.restype = C.c_void_p
        L.orc_server.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint,
                                 C.c_double, C.c_double, C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_mine.restype = C.c_void_p
        L.orc_mine.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                               C.c_uint, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_size_t), C.c_void_p]
        L.orc_enumerate_prefixes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _strs(xs):
    arr = (C.c_char_p * len(xs))(*[x.encode() if isinstance(x, str) else x for x in xs])
    return arr


class Index:
    def __init__(self, path):
        self.h = lib().orc_index_load(path.encode())
        if not self.h:
            raise RuntimeError(lib().orc_last_error().decode())
        self.n = lib().orc_index_length(self.h)

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    def meta(self):
        import numpy as np
        Cc = np.zeros(256, np.uint64)
        cnt = np.zeros(256, np.uint64)
        bits = np.zeros(256, np.uint32)
        code = np.zeros(256, np.uint32)
        lib().orc_index_meta(self.h, Cc.ctypes.data, cnt.ctypes.data, bits.ctypes.data, code.ctypes.data)
        return Cc, cnt, bits, code

    def lf(self, c, i):
        return lib().orc_lf(self.h, c, i & 0xFFFFFFFFFFFFFFFF)

    def lf_batch(self, c, i):
        import numpy as np
        c = np.ascontiguousarray(c, np.uint8)
        i = np.ascontiguousarray(i, np.uint64)
        out = np.zeros(len(c), np.uint64)
        lib().orc_lf_batch(self.h, c.ctypes.data, i.ctypes.data, out.ctypes.data, len(c))
        return out

    def getL(self, i):
        return lib().orc_getL(self.h, i)

    def bwt(self):
        import numpy as np
        out = np.zeros(self.n, np.uint8)
        lib().orc_bwt(self.h, out.ctypes.data)
        return out

    def enumerate(self, name, prefix, fmin=10, maxdepth=0xFFFFFFFF):
        """-> (stream bytes incl. 'S name .' header, (reported, lf_steps, rank_ops))"""
        n = C.c_size_t(0)
        st = (C.c_uint64 * 3)()
        p = lib().orc_enumerate(self.h, name.encode() if name is not None else None, prefix.encode(), fmin, maxdepth,
                                C.byref(n), st)
        data = C.string_at(p, n.value)
        lib().orc_free(p)
        return data, tuple(st)

    def enumerate_prefixes(self, prefixes, fmin=10, maxdepth=0xFFFFFFFF, threads=1):
        st = (C.c_uint64 * 3)()
        nb = C.c_uint64(0)
        lib().orc_enumerate_prefixes(self.h, len(prefixes), _strs(prefixes), fmin, maxdepth, threads, st, C.byref(nb))
        return tuple(st), nb.value


def server(names, streams, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0):
    """metaserver restatement: -> (stdout bytes, (total_paths, total_output, total_occs))"""
    n = C.c_size_t(0)
    st = (C.c_uint64 * 3)()
    bufs = (C.c_char_p * len(streams))(*streams)
    lens = (C.c_size_t * len(streams))(*[len(s) for s in streams])
    p = lib().orc_server(len(names), _strs(names), len(streams), bufs, lens, pmin, pmax, mindepth, emin, emax, C.byref(n), st)
    if not p:
        raise RuntimeError(lib().orc_last_error().decode())
    data = C.string_at(p, n.value)
    lib().orc_free(p)
    return data, tuple(st)


def mine(indexes, names, prefixes, fmin=10, maxdepth=0xFFFFFFFF, pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0, threads=1,
         discard=False):
    n = C.c_size_t(0)
    st = (C.c_uint64 * 6)()
    hs = (C.c_void_p * len(indexes))(*[ix.h for ix in indexes])
    p = lib().orc_mine(len(indexes), hs, _strs(names), len(prefixes), _strs(prefixes), fmin, maxdepth, pmin, pmax, mindepth,
                       emin, emax, threads, None if discard else C.byref(n), st)
    if not p:
        raise RuntimeError(lib().orc_last_error().decode())
    data = C.string_at(p, n.value)
    lib().orc_free(p)
    return data, tuple(st)
