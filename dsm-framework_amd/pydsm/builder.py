"""builder.py -- format-compatible .fmi v17 writer (SURVEY step 1b / f1).

Data preparation, NOT part of the timed hot path: it exists because only this repo travels to the GPU
box, so the multi-GB indexes of the benchmark configurations have to be produced there.  It follows
the reference builder's semantics:
  * per read: upper-case / N-normalise, text = reverse(read + '-' + revcomp(read))   builder.cpp:60-104,183-201
  * one '\\0' terminator per text, terminators ordered by insertion order ($_0 < $_1 < ...)
                                                             TextCollectionBuilder.cpp:65-92, incbwt/misc/utils.cpp:362-367
  * BWT of the collection -> Huffman-shaped wavelet tree -> BitRank per node           HuffWT.cpp:5-55,133-171
  * FMIndex::save layout, version 17                                                  FMIndex.cpp:155-217
The suffix sort is prefix doubling on torch tensors (GPU when available): tooling around the path, the
path itself never touches torch kernels.  Gate (tests/test_builder.py): the BWT equals the one decoded
from the reference builder's .fmi of the same FASTA, and the unmodified reference metaenumerate produces
identical streams from our file.
"""
import os
import struct

import numpy as np
import torch

SAMPLERATE = 124  # TextCollectionBuilder.h:30 (stored, unused by the enumeration path)


# ------------------------------------------------------------------------------------------------
# input handling
# ------------------------------------------------------------------------------------------------
def read_fasta(path_or_text):
    """FASTA records -> list of sequences (multi-line records joined), builder.cpp:203-262."""
    if "\n" in path_or_text or path_or_text.startswith(">"):
        lines = path_or_text.splitlines()
    else:
        with open(path_or_text) as f:
            lines = f.read().splitlines()
    reads, cur, seen = [], [], False
    for row in lines:
        if row[:1] == ">":
            if seen and cur:
                reads.append("".join(cur))
            cur, seen = [], True
        else:
            cur.append(row)
    if cur:
        reads.append("".join(cur))
    return [r for r in reads if r]


_NORM = np.full(256, ord("N"), np.uint8)
for _c in "ACGTN":
    _NORM[ord(_c)] = ord(_c)
    _NORM[ord(_c.lower())] = ord(_c)
for _c in "0123.":
    _NORM[ord(_c)] = ord(_c)  # colour-space symbols pass through normalize() (builder.cpp:86-91)
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip("ACGT", "TGCA"):
    _COMP[ord(_a)] = ord(_b)


def texts_from_reads(reads):
    """list of str -> (flat uint8 symbols with one 0 terminator per text, int64 start offsets [R+1])."""
    parts, offs = [], [0]
    for r in reads:
        a = _NORM[np.frombuffer(r.encode("latin-1"), np.uint8)]
        t = np.concatenate([_COMP[a], np.array([ord("-")], np.uint8), a[::-1], np.zeros(1, np.uint8)])
        parts.append(t)
        offs.append(offs[-1] + len(t))
    return np.concatenate(parts), np.array(offs, np.int64)


def texts_from_codes(codes):
    """[R, L] tensor of base codes 0..3 (A,C,G,T) -> flat symbols [R*(2L+2)] (torch uint8) of equal-length texts."""
    R, L = codes.shape
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=codes.device)
    comp = lut[(3 - codes).long()]
    fwd_rev = lut[codes.flip(1).long()]
    dash = torch.full((R, 1), 45, dtype=torch.uint8, device=codes.device)
    zero = torch.zeros((R, 1), dtype=torch.uint8, device=codes.device)
    return torch.cat([comp, dash, fwd_rev, zero], dim=1).reshape(-1)


def synth_reads(seed, nreads, rlen, genome_len, sub_rate=0.005, device="cpu", private_frac=0.0, genome_seed=1234):
    """Seeded synthetic read set: uniform random genome (shared by every sample via genome_seed), reads from
    both strands, substitution errors; optionally a sample-private sequence (SURVEY 8d cfg 2/3)."""
    gg = torch.Generator(device="cpu").manual_seed(genome_seed)
    genome = torch.randint(0, 4, (genome_len,), generator=gg, dtype=torch.uint8).to(device)
    g = torch.Generator(device=device).manual_seed(seed)
    if private_frac > 0:
        plen = max(rlen + 1, int(genome_len * private_frac))
        private = torch.randint(0, 4, (plen,), generator=g, dtype=torch.uint8, device=device)
        genome = torch.cat([genome, private])
    glen = genome.numel()
    start = torch.randint(0, glen - rlen + 1, (nreads,), generator=g, device=device)
    idx = start[:, None] + torch.arange(rlen, device=device)[None, :]
    reads = genome[idx]
    strand = torch.rand(nreads, generator=g, device=device) < 0.5
    rc = (3 - reads).flip(1)
    reads = torch.where(strand[:, None], rc, reads)
    err = torch.rand(reads.shape, generator=g, device=device) < sub_rate
    shift = torch.randint(1, 4, reads.shape, generator=g, dtype=torch.uint8, device=device)
    reads = torch.where(err, (reads + shift) % 4, reads)
    return reads.contiguous()


# ------------------------------------------------------------------------------------------------
# BWT of a string collection by prefix doubling
# ------------------------------------------------------------------------------------------------
def bwt_collection_hip(sym):
    """The same BWT from libdsmhip's dsm_bwt_build (csrc/bwt.hip: bounded-depth radix sort of the suffixes on the GPU, no
    2^31 limit).  sym: flat uint8 CUDA tensor whose 0 bytes are exactly the terminators."""
    import ctypes as C
    from . import lib, _check
    L = lib()
    L.dsm_bwt_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
    L.dsm_bwt_build.restype = C.c_int
    sym = sym.contiguous()
    bwt = torch.empty_like(sym)
    torch.cuda.synchronize(sym.device)
    _check(L.dsm_bwt_build(sym.data_ptr(), sym.numel(), bwt.data_ptr(), sym.device.index or 0, None))
    return bwt


def bwt_collection(sym, starts=None, width=None, hip=None):
    """sym: flat uint8 tensor, every text ends with 0.  Either `starts` (int64 [R+1], variable length texts)
    or `width` (equal-length texts).  Terminators sort by text order.  Returns the BWT (uint8 tensor).
    hip: use the library's GPU suffix sort (default: whenever sym lives on the GPU); False = the torch prefix doubling below,
    which also runs on the CPU (n < 2^31)."""
    if hip is None:
        hip = sym.is_cuda
    if hip:
        return bwt_collection_hip(sym)
    dev = sym.device
    n = sym.numel()
    pos = torch.arange(n, device=dev, dtype=torch.int64)
    if width is not None:
        R = n // width
        tid = pos // width
        end = tid * width + (width - 1)  # position of the text's terminator
        first = tid * width
    else:
        starts = starts.to(dev)
        R = starts.numel() - 1
        lens = starts[1:] - starts[:-1]
        tid = torch.repeat_interleave(torch.arange(R, device=dev, dtype=torch.int64), lens)
        end = starts[1:][tid] - 1
        first = starts[:-1][tid]
    is_term = pos == end
    # initial rank: terminators 0..R-1 by text order, other symbols R + byte value
    rank = torch.where(is_term, tid, sym.to(torch.int64) + R)
    del tid
    h = 1
    while True:
        nxt = pos + h
        valid = nxt <= end
        nxt = torch.where(valid, nxt, pos)
        second = torch.where(valid, rank[nxt] + 1, torch.zeros_like(rank))
        del nxt, valid
        key = rank * (n + R + 258) + second
        del second
        uniq, inv = torch.unique(key, return_inverse=True)
        del key
        nu = uniq.numel()
        del uniq
        rank = inv
        del inv
        if nu == n:
            break
        h *= 2
        if h > 4 * n:
            raise RuntimeError("prefix doubling did not converge")
    prev = torch.where(pos == first, torch.zeros_like(sym), sym[torch.clamp(pos - 1, min=0)])
    bwt = torch.empty_like(sym)
    bwt[rank] = prev
    return bwt


# ------------------------------------------------------------------------------------------------
# Huffman code exactly as node::makecodetable (HuffWT.cpp:133-171), including libstdc++'s heap order
# ------------------------------------------------------------------------------------------------
class _Heap:
    """std::priority_queue<node, vector<node>, greater<node>> (libstdc++ push_heap / pop_heap)."""

    def __init__(self):
        self.v = []

    @staticmethod
    def _comp(a, b):  # greater<node>: a.weight > b.weight
        return a[0] > b[0]

    def _push_heap(self, hole, top, value):
        v = self.v
        parent = (hole - 1) // 2
        while hole > top and self._comp(v[parent], value):
            v[hole] = v[parent]
            hole = parent
            parent = (hole - 1) // 2
        v[hole] = value

    def push(self, x):
        self.v.append(x)
        self._push_heap(len(self.v) - 1, 0, x)

    def pop(self):
        v = self.v
        top = v[0]
        value = v[-1]
        ln = len(v) - 1
        if ln > 0:
            hole, second = 0, 0
            while second < (ln - 1) // 2:
                second = 2 * (second + 1)
                if self._comp(v[second], v[second - 1]):
                    second -= 1
                v[hole] = v[second]
                hole = second
            if (ln & 1) == 0 and second == (ln - 2) // 2:
                second = 2 * (second + 1)
                v[hole] = v[second - 1]
                hole = second - 1
            v.pop()
            self._push_heap(hole, 0, value)
        else:
            v.pop()
        return top


def huffman_codes(counts):
    """counts[256] -> (bits[256], code[256]); code bits are consumed LSB first (HuffWT.h:66-83)."""
    q = _Heap()
    for i in range(256):
        if counts[i]:
            q.push((int(counts[i]), ("leaf", i)))
    while len(q.v) > 1:
        c0 = q.pop()
        c1 = q.pop()
        q.push((c0[0] + c1[0], ("node", c0, c1)))
    bits = np.zeros(256, np.uint32)
    code = np.zeros(256, np.uint32)

    def walk(nd, c, b):
        kind = nd[1]
        if kind[0] == "node":
            walk(kind[1], c, b + 1)                 # SetBit(code, bits, 0)
            walk(kind[2], c | (1 << b), b + 1)      # SetBit(code, bits, 1)
        else:
            code[kind[1]] = c
            bits[kind[1]] = b

    walk(q.v[0], 0, 0)
    return bits, code


# ------------------------------------------------------------------------------------------------
# wavelet tree + file
# ------------------------------------------------------------------------------------------------
def _bitrank_bytes(bits_t):
    """bool/uint8 tensor of n bits -> BitRank::save bytes (BitRank.cpp:134-151, BuildRank :154-187)."""
    n = int(bits_t.numel())
    integers = (n + 1 + 63) // 64
    pad = integers * 64 - n
    b = torch.cat([bits_t.to(torch.uint8), torch.zeros(pad, dtype=torch.uint8, device=bits_t.device)])
    b8 = b.view(-1, 8)
    by = b8[:, 0].clone()
    for k in range(1, 8):
        by |= b8[:, k] << k                                      # bit k of a word is bit k%8 of byte k/8 (LSB first)
    wordpop = b.view(-1, 64).sum(1, dtype=torch.int64)           # popcount per 64-bit word
    del b
    cum = torch.cat([torch.zeros(1, dtype=torch.int64, device=wordpop.device), torch.cumsum(wordpop, 0)])
    nRs = n // 256 + 1
    nRb = n // 64 + 1
    # Rs[j] = ones in words [0, 4j); words beyond `integers` count as 0
    j4 = torch.clamp(torch.arange(nRs, device=cum.device) * 4, max=integers)
    Rs = cum[j4]
    k = torch.arange(nRb, device=cum.device)
    lo = torch.clamp((k // 4) * 4, max=integers)
    hi = torch.clamp((k // 4) * 4 + (k % 4), max=integers)
    Rb = (cum[hi] - cum[lo]).to(torch.uint8)
    out = [struct.pack("<QQII", n, integers, 64, 256), by.cpu().numpy().tobytes(), Rs.cpu().numpy().astype("<u8").tobytes(),
           Rb.cpu().numpy().tobytes()]
    return out


def _wt_node(seq, code_lut, level, out):
    """HuffWT::HuffWT(uchar*, n, codetable, level) + HuffWT::save (HuffWT.cpp:5-55,73-86), pre-order."""
    ch = int(seq[0])
    lut = ((code_lut >> level) & 1).to(torch.uint8)
    bit = torch.empty_like(seq)
    step = 1 << 28                                 # bounded temporaries: indexes beyond 2^32 symbols are built this way too
    for o in range(0, int(seq.numel()), step):
        bit[o:o + step] = lut[seq[o:o + step].int()]
    s = int(bit.sum(dtype=torch.int64))
    n = int(seq.numel())
    if s == 0 or s == n:
        out.append(struct.pack("<BB", 1, ch))
        return
    out.append(struct.pack("<BB", 0, ch))
    out.extend(_bitrank_bytes(bit))
    m = bit.bool()
    step = 1 << 30
    left = torch.cat([seq[o:o + step][~m[o:o + step]] for o in range(0, n, step)])
    right = torch.cat([seq[o:o + step][m[o:o + step]] for o in range(0, n, step)])
    del bit, m
    _wt_node(left, code_lut, level + 1, out)
    del left
    _wt_node(right, code_lut, level + 1, out)


def write_fmi_hip(bwt, path, number_of_texts, max_text_length):
    """The same file from libdsmhip's dsm_fmi_write (csrc/fmiwrite.hip: wavelet-tree bit vectors and rank directories on the GPU)."""
    import ctypes as C
    from . import lib, _check
    L = lib()
    L.dsm_fmi_write.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p]
    L.dsm_fmi_write.restype = C.c_int
    bwt = bwt.contiguous()
    torch.cuda.synchronize(bwt.device)
    _check(L.dsm_fmi_write(bwt.data_ptr(), bwt.numel(), int(number_of_texts), int(max_text_length), SAMPLERATE, bwt.device.index or 0,
                           os.fsencode(path)))
    return {"n": int(bwt.numel())}


def build_fasta_hip(fasta_path, out_path, device=0):
    """FASTA file -> .fmi entirely inside the library (dsm_build_fasta; what host/builder_hip calls)."""
    import ctypes as C
    from . import lib, _check

    class Info(C.Structure):
        _fields_ = [("n", C.c_uint64), ("number_of_texts", C.c_uint64), ("max_text_length", C.c_uint64)]
    L = lib()
    L.dsm_build_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(Info)]
    L.dsm_build_fasta.restype = C.c_int
    info = Info()
    _check(L.dsm_build_fasta(os.fsencode(fasta_path), os.fsencode(out_path), SAMPLERATE, device, C.byref(info)))
    return {"n": info.n, "number_of_texts": info.number_of_texts, "max_text_length": info.max_text_length}


def write_fmi(bwt, path, number_of_texts, max_text_length, hip=None):
    """FMIndex::save (FMIndex.cpp:155-217), version 17, no samples / names / text storage.
    hip: write through the library (default: whenever the BWT lives on the GPU); False = the torch tooling below (also CPU)."""
    if hip is None:
        hip = bwt.is_cuda
    if hip:
        return write_fmi_hip(bwt, path, number_of_texts, max_text_length)
    n = int(bwt.numel())
    counts = torch.bincount(bwt.long(), minlength=256).cpu().numpy().astype(np.uint64)
    C = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype("<u8")     # FMIndex::makewavelet, FMIndex.cpp:395-410
    bits, code = huffman_codes(counts)
    out = [struct.pack("<BQI", 17, n, SAMPLERATE), C.tobytes(), struct.pack("<Q", 0)]
    for i in range(256):
        out.append(struct.pack("<QII", int(counts[i]), int(bits[i]), int(code[i])))
    code_lut = torch.tensor(code.astype(np.int64), device=bwt.device)
    _wt_node(bwt, code_lut, 0, out)
    out.append(struct.pack("<IQBBBI", number_of_texts, max_text_length, 0, 0, 0, 0))
    with open(path, "wb") as f:
        for b in out:
            f.write(b)
    return {"n": n, "counts": counts, "bits": bits, "code": code}


def build_from_fasta(fasta, out_path, device="cpu"):
    """FASTA path or text -> <out_path> (.fmi).  Mirrors `builder in.fasta` (builder.cpp:329)."""
    reads = read_fasta(fasta)
    sym, starts = texts_from_reads(reads)
    symt = torch.from_numpy(sym).to(device)
    bwt = bwt_collection(symt, starts=torch.from_numpy(starts))
    lens = starts[1:] - starts[:-1]
    return write_fmi(bwt, out_path, len(reads), int(lens.max()))


def build_from_codes(codes, out_path):
    """[R, L] base-code tensor (on any device) -> .fmi; equal-length texts, arithmetic text boundaries."""
    R, L = codes.shape
    sym = texts_from_codes(codes)
    bwt = bwt_collection(sym, width=2 * L + 2)
    del sym
    return write_fmi(bwt, out_path, R, 2 * L + 2)


def codes_to_fasta(codes):
    lut = np.frombuffer(b"ACGT", np.uint8)
    a = lut[codes.cpu().numpy()]
    return "".join(">r%d\n%s\n" % (i, row.tobytes().decode()) for i, row in enumerate(a))
