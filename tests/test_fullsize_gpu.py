"""Parity at BASELINE.json's full size (configs[1]: 10^7 reads x 100 bp, n = 2.02e9), through properties that do not
need the oracle to enumerate the whole trie:
  * --check: the LF intervals of all symbols tile [0, n)                       (metaenumerate.cpp:93-127)
  * random 8-mer prefixes of the FULL index: GPU tuples / wire bytes == oracle's, byte for byte
  * a pass over one-letter prefixes: every emitted node is a union node (d = 1), tuples <= candidates,
    counters per node match the reference's cost model, and stream mode reports the same node count
Set DSM_FULLSIZE_READS to shrink it (default 10000000)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def big():
    import torch
    import pydsm
    from pydsm import builder
    reads = int(os.environ.get("DSM_FULLSIZE_READS", "10000000"))
    genome = reads * 5
    d = os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "sample-0.s42_r%d_l100_g%d_e0.005.fmi" % (reads, genome))
    if not os.path.exists(path):
        codes = builder.synth_reads(42, reads, 100, genome, 0.005, device="cuda")
        builder.build_from_codes(codes, path + ".tmp")
        del codes
        torch.cuda.empty_cache()
        os.replace(path + ".tmp", path)
    ix = pydsm.Index(path)
    yield pydsm, ix, path, reads
    ix.close()


def test_check_and_random_prefixes_against_oracle(big):
    import orc
    pydsm, ix, path, reads = big
    assert ix.n == reads * 202
    assert ix.check() == ix.n
    o = orc.Index(path)
    rng = np.random.default_rng(2026)
    # LF on random positions, every live symbol
    Cc, cnt, bits, code = o.meta()
    syms = [int(s) for s in np.nonzero(cnt)[0]]
    pos = np.concatenate([np.array([0xFFFFFFFFFFFFFFFF, 0, ix.n - 1], np.uint64), rng.integers(0, ix.n, 200000).astype(np.uint64)])
    cs = rng.choice(syms, len(pos)).astype(np.uint8)
    assert (ix.lf_batch(cs, pos) == o.lf_batch(cs, pos)).all()
    with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m, pydsm.Miner([ix], fmin=10, stream_mode=True) as sm:
        for _ in range(6):
            p = "".join(rng.choice(list("ACGT"), 8))
            got, st = m.mine(p)
            want, ost = orc.mine([o], [ix.name], [p], fmin=10, pmin=1, emax=2.0)
            assert got == want, p
            assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples) == ost[:5], p
            wire, sst = sm.enumerate(p)
            owire, _ = o.enumerate(ix.name, p, fmin=10)
            assert wire == owire, p
            assert sst.reported == st.reported
    o.close()


def test_whole_pass_invariants(big):
    pydsm, ix, path, reads = big
    seen = {"tuples": 0, "last": None, "ok": True}

    def on_batch(b):
        # post-order inside a prefix: a tuple's path is never a proper prefix of the NEXT tuple's path unless ... (children first)
        n = int(b.ntuples)
        seen["tuples"] += n
        po = np.ctypeslib.as_array(b.path_off, shape=(n + 1,))
        ent = np.ctypeslib.as_array(b.entropy, shape=(n,))
        fr = np.ctypeslib.as_array(b.freqs, shape=(int(np.ctypeslib.as_array(b.pair_off, shape=(n + 1,))[-1]),))
        seen["ok"] = seen["ok"] and bool((ent >= 0).all() and (ent <= 2.0).all() and (fr >= 10).all() and (np.diff(po) >= 1).all())

    with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m:
        _, st = m.mine_many(["A", "C", "G", "T"], text=False, on_batch=on_batch)
    assert seen["ok"]
    assert st.union_nodes == st.reported                      # d = 1: every union node is a node of the one sample
    assert seen["tuples"] == st.tuples <= st.candidates
    assert 10.0 < st.lf_steps / st.reported < 11.5            # reference cost model: ~10.8 LF and ~24.3 rank-ops per node
    assert 23.0 < st.rank_ops / st.reported < 25.5
    # the same pass as wire streams reports the same number of nodes
    with pydsm.Miner([ix], fmin=10, stream_mode=True) as sm:
        nb, sst = sm.enumerate("G", discard=True)
    with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m:
        _, gst = m.mine("G", text=False)
    assert sst.reported == gst.reported and nb > 4 * sst.reported


def _host_cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def test_uncut_one_letter_prefix_equals_oracle(big):
    """ONE whole one-letter prefix of the configs[1] index, uncut (0.7e9 nodes, 1.7e7 tuples), against the oracle: the sha256 of
    the reference-format tuple text.  The oracle runs one thread per prefix (like metaenumerate.cpp:268), so it enumerates the
    1024 six-letter prefixes below "G" in parallel: their outputs, concatenated in prefix order, are the uncut run's text minus
    its lines for the nodes above depth 6 (post-order keeps every subtree contiguous; with one sample the nodes of an enforced
    path have one child and are never printed, metaserver.cpp:416-417).  Those few shallow lines are compared with a depth-cut
    oracle run of "G" (a node's line depends on its children only, which a cut one level below leaves complete)."""
    import hashlib
    import itertools
    import orc
    pydsm, ix, path, reads = big
    kw = dict(fmin=10, pmin=1, emax=2.0)
    with pydsm.Miner([ix], **kw) as m:
        text, st = m.mine("G")
    assert st.splits == 0
    buf = np.frombuffer(text, np.uint8)
    nl = np.flatnonzero(buf == 10)
    assert len(nl) == st.tuples and nl[-1] == len(buf) - 1
    starts = np.concatenate((np.zeros(1, np.int64), nl[:-1] + 1))
    shallow = np.zeros(len(starts), bool)
    for k in range(1, 6):  # a space among the first six bytes: a path shorter than six symbols
        shallow |= buf[np.minimum(starts + k, len(buf) - 1)] == 32
    deep, top = hashlib.sha256(), []
    pos = 0
    for j in np.flatnonzero(shallow):
        deep.update(memoryview(text)[pos:starts[j]])
        top.append(bytes(text[starts[j]:nl[j] + 1]))
        pos = int(nl[j]) + 1
    deep.update(memoryview(text)[pos:])
    o = orc.Index(path)
    threads = min(orc.lib().orc_max_threads(), _host_cores())
    six = ["G" + "".join(q) for q in itertools.product("ACGT", repeat=5)]
    want, ost = orc.mine([o], [ix.name], six, threads=threads, **kw)
    cut, _ = orc.mine([o], [ix.name], ["G"], maxdepth=6, **kw)
    o.close()
    assert hashlib.sha256(want).digest() == deep.digest(), "tuple text of the uncut prefix differs from the oracle's"
    assert len(text) - sum(len(t) for t in top) == len(want)
    assert top == [ln + b"\n" for ln in cut.split(b"\n") if ln and ln.find(b" ") < 6]
    # every node of the prefix was counted: the oracle's runs count their six enforced nodes each
    assert st.reported == ost[0] - 6 * len(six) + sum(4 ** k for k in range(6))
    print("uncut prefix G: %d nodes, %d tuples, %.2f GB of text, sha256 %s" % (st.reported, st.tuples, len(text) / 1e9, hashlib.sha256(text).hexdigest()[:16]))


def test_split_wire_stream_at_full_size(big):
    """Stream mode with buffers far too small for a one-letter prefix of the full-size index: the prefix goes out as slices of
    its sub-prefixes' streams (several levels of splitting) and the bytes equal the unsplit run's -- which, restricted to the
    same depth, equal the oracle's."""
    import hashlib
    import orc
    pydsm, ix, path, reads = big
    with pydsm.Miner([ix], fmin=10, maxdepth=13, stream_mode=True) as sm:
        whole, st0 = sm.enumerate("C")
    assert st0.splits == 0 and len(whole) > 50_000_000
    with pydsm.Miner([ix], fmin=10, maxdepth=13, stream_mode=True, arena_bytes=96 << 20) as sm:
        parts, st1 = sm.enumerate("C")
        many, _ = sm.enumerate_many(["T", "C"])
    assert st1.splits >= 4
    assert hashlib.sha256(parts).digest() == hashlib.sha256(whole).digest() and many[1] == whole
    o = orc.Index(path)
    want, _ = o.enumerate(ix.name, "C", fmin=10, maxdepth=13)
    o.close()
    assert whole == want


def test_three_midsize_samples_against_oracle():
    """d = 3 at a size where the top levels need wide frequency columns and the rest 16-bit ones (the toy fixtures never
    leave the 16-bit regime): GPU tuples == oracle's on whole one-letter prefixes restricted by maxdepth, and on deep 7-mers."""
    import torch
    import orc
    import pydsm
    from pydsm import builder
    d = os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench")
    os.makedirs(d, exist_ok=True)
    paths = []
    for s in range(3):
        p = os.path.join(d, "mid-%d.fmi" % s)
        if not os.path.exists(p):
            codes = builder.synth_reads(100 + s, 200000, 100, 1000000, 0.005, device="cuda", private_frac=0.05)
            builder.build_from_codes(codes, p + ".tmp")
            os.replace(p + ".tmp", p)
        paths.append(p)
    torch.cuda.empty_cache()
    idx = [pydsm.Index(p) for p in paths]
    oidx = [orc.Index(p) for p in paths]
    names = [ix.name for ix in idx]
    rng = np.random.default_rng(7)
    cases = [("A", dict(fmin=10, maxdepth=9, pmin=2, emax=2.0)), ("", dict(fmin=10, maxdepth=6, pmin=1, emax=0.0)),
             ("".join(rng.choice(list("ACGT"), 7)), dict(fmin=10, pmin=2, emax=2.0)),
             ("".join(rng.choice(list("ACGT"), 7)), dict(fmin=3, pmin=1, pmax=2, emax=1.5, emin=0.2, mindepth=12))]
    for p, kw in cases:
        got, st = pydsm.mine(idx, p, **kw)
        want, ost = orc.mine(oidx, names, [p], threads=4, **kw)
        assert got == want, (p, kw)
        assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (p, kw)
    for ix in idx + oidx:
        ix.close()


def test_positions_beyond_2_32():
    """configs[3]'s distinguishing feature is the position width (n = 8.08e9 > 2^32).  No BWT of that size can be sorted
    here, but rank / LF / the enumeration are defined for ANY symbol string, so: a seeded pseudo-BWT of n = 2^32 + 3e8
    symbols with DNA-like symbol frequencies is written as a v17 .fmi, and the device index (3 superblocks, 64-bit
    positions, wide frequency columns) must answer exactly like the oracle on the same file."""
    import torch
    import orc
    import pydsm
    from pydsm import builder
    n = int(os.environ.get("DSM_WIDE_N", str((1 << 32) + 300_000_000)))
    d = os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "pseudo-%d.fmi" % n)
    if not os.path.exists(path):
        g = torch.Generator(device="cuda").manual_seed(4242)
        syms = torch.tensor([0, ord("-"), ord("A"), ord("C"), ord("G"), ord("N"), ord("T")], dtype=torch.uint8, device="cuda")
        cum = torch.tensor([0.005, 0.010, 0.258, 0.505, 0.750, 0.752], device="cuda")
        bwt = torch.empty(n, dtype=torch.uint8, device="cuda")
        step = 1 << 28
        for o in range(0, n, step):
            m = min(step, n - o)
            bwt[o:o + m] = syms[torch.bucketize(torch.rand(m, device="cuda", generator=g), cum)]
        builder.write_fmi(bwt, path + ".tmp", 1, 0)
        del bwt
        torch.cuda.empty_cache()
        os.replace(path + ".tmp", path)
    o = orc.Index(path)
    with pydsm.Index(path) as ix:
        assert ix.n == o.n == n
        rng = np.random.default_rng(99)
        Cc, cnt, bits, code = o.meta()
        syms = [int(s) for s in np.nonzero(cnt)[0]]
        edges = [0xFFFFFFFFFFFFFFFF, 0, n - 1]
        for e in (1 << 31, 1 << 32, 2 << 31, 3 << 31):
            edges += [x for x in (e - 129, e - 128, e - 2, e - 1, e, e + 1, e + 127, e + 128) if 0 <= x < n]
        pos = np.concatenate([np.array(edges, np.uint64), rng.integers(0, n, 300000).astype(np.uint64)])
        for s in syms:      # every symbol at every edge, then a random mix
            cs = np.full(len(edges), s, np.uint8)
            assert (ix.lf_batch(cs, pos[:len(edges)]) == o.lf_batch(cs, pos[:len(edges)])).all()
        cs = rng.choice(syms, len(pos)).astype(np.uint8)
        assert (ix.lf_batch(cs, pos) == o.lf_batch(cs, pos)).all()
        assert (ix.getl_batch(pos[1:3000]) == np.array([o.getL(int(p)) for p in pos[1:3000]], np.uint8)).all()
        with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m, pydsm.Miner([ix], fmin=10, stream_mode=True) as sm:
            for p in ("ACGTAC", "TTTTTT", "GATTACA"):
                got, st = m.mine(p)
                want, ost = orc.mine([o], [ix.name], [p], fmin=10, pmin=1, emax=2.0)
                assert got == want, p
                assert st.reported > 1000 and (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples) == ost[:5], p
                wire, sst = sm.enumerate(p)
                owire, _ = o.enumerate(ix.name, p, fmin=10)
                assert wire == owire, p
    o.close()


def test_reload_time_of_a_full_size_index(big):
    """f4 measurement: bringing the 1 GB block array of the configs[1] index back from pinned host memory."""
    import time
    import torch
    pydsm, ix, path, reads = big
    ix.offload()
    t0 = time.time()
    ix.reload()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("reload of %.2f GB: %.1f ms = %.1f GB/s" % (ix.device_bytes() / 1e9, dt * 1e3, ix.device_bytes() / dt / 1e9))
    assert ix.check() == ix.n
    assert dt < 2.0


def test_two_full_size_samples_against_oracle(big):
    """d = 2 at BASELINE's per-sample size (two 10^7-read sets on one card, the second with 5 % private sequence): random
    8-mer prefixes and a whole one-letter prefix restricted by maxdepth, tuples and counters equal the oracle's."""
    import torch
    import orc
    from pydsm import builder
    pydsm, ix0, path0, reads = big
    d = os.path.dirname(path0)
    path1 = os.path.join(d, "sample-1.s43_r%d_l100_g%d_e0.005_p0.05.fmi" % (reads, reads * 5))
    if not os.path.exists(path1):
        codes = builder.synth_reads(43, reads, 100, reads * 5, 0.005, device="cuda", private_frac=0.05)
        builder.build_from_codes(codes, path1 + ".tmp")
        del codes
        torch.cuda.empty_cache()
        os.replace(path1 + ".tmp", path1)
    ix1 = pydsm.Index(path1)
    o0, o1 = orc.Index(path0), orc.Index(path1)
    names = [ix0.name, ix1.name]
    rng = np.random.default_rng(77)
    with pydsm.Miner([ix0, ix1], fmin=10, pmin=1, emax=2.0) as m1, pydsm.Miner([ix0, ix1], fmin=10, pmin=2, emax=1.0, emin=0.05) as m2:
        for _ in range(4):
            p = "".join(rng.choice(list("ACGT"), 8))
            for m, kw in ((m1, dict(fmin=10, pmin=1, emax=2.0)), (m2, dict(fmin=10, pmin=2, emax=1.0, emin=0.05))):
                got, st = m.mine(p)
                want, ost = orc.mine([o0, o1], names, [p], threads=4, **kw)
                assert got == want, (p, kw)
                assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (p, kw)
    got, st = pydsm.mine([ix0, ix1], "G", fmin=10, maxdepth=11, pmin=1, emax=2.0)
    want, ost = orc.mine([o0, o1], names, ["G"], fmin=10, maxdepth=11, pmin=1, emax=2.0, threads=8)
    assert got == want and st.tuples == ost[4] and st.max_frontier > 1000000
    ix1.close(); o0.close(); o1.close()


def test_eight_full_size_samples_on_one_card(big):
    """The one-GPU share of BASELINE configs[2], [3] and [4]: eight 10^7-read samples (seeds 42..49, 5 % private sequence
    each, n = 2.02e9 per index) resident on one card, d = 8, with the three configurations' filters -- the reference default
    (-P 2), -P 2 --pmax 8 with 64-bit positions, and -P 1 --pmax 1 (sample-specific substrings; no reader-set order needed)
    -- on random 8-mers and on a one-letter prefix cut at depth 10: tuples and counters equal the oracle's.  Then configs[4]'s
    residency cycle with two groups of eight handles: one group is offloaded while the other mines, and comes back by an
    asynchronous copy on a side stream (metaserver.cpp:406-419 for the predicates)."""
    import torch
    import orc
    from pydsm import builder
    pydsm, ix0, path0, reads = big
    d = os.path.dirname(path0)
    paths = []
    for s in range(8):
        p = os.path.join(d, "sample-%d.s%d_r%d_l100_g%d_e0.005_p0.05.fmi" % (s, 42 + s, reads, reads * 5))
        if not os.path.exists(p):
            codes = builder.synth_reads(42 + s, reads, 100, reads * 5, 0.005, device="cuda", private_frac=0.05)
            builder.build_from_codes(codes, p + ".tmp")
            del codes
            torch.cuda.empty_cache()
            os.replace(p + ".tmp", p)
        paths.append(p)
    torch.cuda.empty_cache()
    A = [pydsm.Index(p) for p in paths]
    O = [orc.Index(p) for p in paths]
    names = [ix.name for ix in A]
    assert len(set(names)) == 8 and all(ix.n == reads * 202 for ix in A)
    rng = np.random.default_rng(2027)
    kmers = ["".join(rng.choice(list("ACGT"), 8)) for _ in range(3)]
    cfgs = [dict(fmin=10, pmin=2, emax=2.0), dict(fmin=10, pmin=2, pmax=8, emax=2.0, wide=1), dict(fmin=10, pmin=1, pmax=1, emax=2.0)]
    first = {}
    for kw in cfgs:
        okw = {k: v for k, v in kw.items() if k != "wide"}
        with pydsm.Miner(A, **kw) as m:
            for p in kmers:
                got, st = m.mine(p)
                want, ost = orc.mine(O, names, [p], threads=8, **okw)
                assert got == want, (p, kw)
                assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (p, kw)
                first.setdefault((p, kw.get("pmax", 0)), got)
        got, st = pydsm.mine(A, "C", maxdepth=10, **kw)
        want, ost = orc.mine(O, names, ["C"], maxdepth=10, threads=8, **okw)
        assert got == want and (st.reported, st.union_nodes, st.tuples, st.pairs) == (ost[0], ost[3], ost[4], ost[5]), kw
        assert st.max_frontier > 200000 and st.pair_order_exact == 1
    for o in O:
        o.close()
    # ---- two groups of eight: B is a second set of handles on the same files
    B = [pydsm.Index(p) for p in paths]
    kw = cfgs[2]
    p = kmers[0]
    for ix in A:
        ix.offload()
    assert not any(ix.resident for ix in A)
    with pytest.raises(pydsm.DsmError):
        pydsm.mine(A, p, **kw)                      # an offloaded group is refused, not read
    side = torch.cuda.Stream()
    with pydsm.Miner(B, **kw) as mb:
        for ix in A:                                # group A comes back on the side stream while group B mines
            ix.reload(side.cuda_stream)
        got, _ = mb.mine(p)
        assert got == first[(p, 1)]
    side.synchronize()
    for ix in B:
        ix.offload()
    assert all(ix.resident for ix in A) and not any(ix.resident for ix in B)
    got, _ = pydsm.mine(A, p, **kw)
    assert got == first[(p, 1)]
    assert A[3].check() == A[3].n
    for ix in A + B:
        ix.close()


def test_real_bwt_beyond_2_32():
    """REAL indexes at BASELINE configs[3]'s size (round 1 could only fake one): 4e7 reads of 100 bp = 4 Gbases per sample
    (n = 8.08e9 > 2^32) are suffix-sorted on the GPU by dsm_bwt_build (csrc/bwt.hip), written as .fmi v17, opened with 64-bit
    positions and four superblocks, and must pass --check and answer LF, tuples, counters and wire bytes exactly like the
    oracle on the same file; then two such samples with configs[3]'s filter (-P 2 --pmax 8).  DSM_BIGBWT_READS shrinks it
    (n must stay above 2^32: at least 21300000)."""
    import time
    import torch
    import orc
    import pydsm
    from pydsm import builder
    reads = int(os.environ.get("DSM_BIGBWT_READS", "40000000"))
    d = os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "real-%d.fmi" % reads)
    if not os.path.exists(path):
        t0 = time.time()
        codes = builder.synth_reads(4242, reads, 100, reads * 5, 0.005, device="cuda")
        sym = builder.texts_from_codes(codes)
        del codes
        torch.cuda.empty_cache()
        t1 = time.time()
        bwt = builder.bwt_collection(sym, hip=True)
        torch.cuda.synchronize()
        t2 = time.time()
        del sym
        torch.cuda.empty_cache()
        builder.write_fmi(bwt, path + ".tmp", reads, 202)
        del bwt
        torch.cuda.empty_cache()
        os.replace(path + ".tmp", path)
        print("reads %.1f s, BWT of n = %d: %.1f s (%.1f Msymbols/s), .fmi %.1f s" % (t1 - t0, reads * 202, t2 - t1, reads * 202 / (t2 - t1) / 1e6,
                                                                                    time.time() - t2))
    n = reads * 202
    assert n > (1 << 32)
    o = orc.Index(path)
    with pydsm.Index(path) as ix:
        assert ix.n == o.n == n
        assert ix.check() == n                      # the LF intervals of all symbols tile [0, n): the BWT is a permutation-consistent one
        rng = np.random.default_rng(123)
        Cc, cnt, bits, code = o.meta()
        syms = [int(s) for s in np.nonzero(cnt)[0]]
        assert cnt[0] == reads and cnt[ord("-")] == reads and cnt[ord("A")] + cnt[ord("C")] + cnt[ord("G")] + cnt[ord("T")] == reads * 200
        pos = np.concatenate([np.array([0xFFFFFFFFFFFFFFFF, 0, n - 1, (1 << 32) - 1, 1 << 32, (1 << 32) + 1], np.uint64),
                              rng.integers(0, n, 200000).astype(np.uint64)])
        cs = rng.choice(syms, len(pos)).astype(np.uint8)
        assert (ix.lf_batch(cs, pos) == o.lf_batch(cs, pos)).all()
        with pydsm.Miner([ix], fmin=10, pmin=1, emax=2.0) as m, pydsm.Miner([ix], fmin=10, stream_mode=True) as sm:
            for _ in range(4):
                p = "".join(rng.choice(list("ACGT"), 8))
                got, st = m.mine(p)
                want, ost = orc.mine([o], [ix.name], [p], fmin=10, pmin=1, emax=2.0)
                assert got == want, p
                assert st.reported > 1000 and (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples) == ost[:5], p
                wire, sst = sm.enumerate(p)
                owire, _ = o.enumerate(ix.name, p, fmin=10)
                assert wire == owire, p
        # a real BWT, unlike the pseudo one: every read's reverse complement is in the collection, so a k-mer and its reverse
        # complement have the same frequency -- check it through LF-interval sizes of a few random 12-mers
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        with pydsm.Miner([ix], fmin=1, maxdepth=12, pmin=1, emax=0.0) as m1:
            for _ in range(3):
                p = "".join(rng.choice(list("ACGT"), 12))
                rc = "".join(comp[c] for c in reversed(p))
                a, _ = m1.mine(p)
                b, _ = m1.mine(rc)
                fa = [ln.split()[-1].split(":")[1] for ln in a.decode().splitlines() if ln.split()[0] == p]
                fb = [ln.split()[-1].split(":")[1] for ln in b.decode().splitlines() if ln.split()[0] == rc]
                assert fa == fb, (p, rc, fa, fb)
    # ---- a second 4-Gbase sample (5 % private sequence) and configs[3]'s filter with d = 2 ----
    path2 = os.path.join(d, "real-%d-b.fmi" % reads)
    if not os.path.exists(path2):
        codes = builder.synth_reads(4243, reads, 100, reads * 5, 0.005, device="cuda", private_frac=0.05)  # same genome as the first sample
        builder.build_from_codes(codes, path2 + ".tmp")
        del codes
        torch.cuda.empty_cache()
        os.replace(path2 + ".tmp", path2)
    o2 = orc.Index(path2)
    with pydsm.Index(path) as ixa, pydsm.Index(path2) as ixb:
        names = [ixa.name, ixb.name]
        assert names[0] != names[1]
        with pydsm.Miner([ixa, ixb], fmin=10, pmin=2, pmax=8, emax=2.0) as m:
            for _ in range(3):
                p = "".join(rng.choice(list("ACGT"), 8))
                got, st = m.mine(p)
                want, ost = orc.mine([o, o2], names, [p], fmin=10, pmin=2, pmax=8, emax=2.0, threads=4)
                assert got == want, p
                assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, p
    # ---- configs[3] at its one-card share: EIGHT 4-Gbase samples resident (8 x 4 GB of index), d = 8, -P 2 --pmax 8, 64-bit
    # positions (metaserver.cpp:406-419 for the predicates, EnumerateQuery.cpp:151-238 for the enumeration).  DSM_BIGBWT_SAMPLES
    # shrinks the number of samples (default 8).
    nsamp = int(os.environ.get("DSM_BIGBWT_SAMPLES", "8"))
    paths = [path, path2]
    for k in range(2, nsamp):
        pk = os.path.join(d, "real-%d-%d.fmi" % (reads, k))
        if not os.path.exists(pk):
            t0 = time.time()
            codes = builder.synth_reads(4242 + k, reads, 100, reads * 5, 0.005, device="cuda", private_frac=0.05)
            builder.build_from_codes(codes, pk + ".tmp")
            del codes
            torch.cuda.empty_cache()
            os.replace(pk + ".tmp", pk)
            print("sample %d of %d built in %.1f s" % (k, nsamp, time.time() - t0), flush=True)
        paths.append(pk)
    paths = paths[:nsamp]
    O = [o, o2] + [orc.Index(pk) for pk in paths[2:]]
    A = [pydsm.Index(pk) for pk in paths]
    names = [ix.name for ix in A]
    assert len(set(names)) == nsamp and all(ix.n == n for ix in A)
    kw = dict(fmin=10, pmin=2, pmax=8, emax=2.0)
    with pydsm.Miner(A, **kw) as m:
        for _ in range(3):
            p = "".join(rng.choice(list("ACGT"), 8))
            got, st = m.mine(p)
            want, ost = orc.mine(O[:nsamp], names, [p], threads=8, **kw)
            assert got == want, p
            assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, p
    got, st = pydsm.mine(A, "C", maxdepth=10, **kw)
    want, ost = orc.mine(O[:nsamp], names, ["C"], maxdepth=10, threads=8, **kw)
    assert got == want and (st.reported, st.union_nodes, st.tuples, st.pairs) == (ost[0], ost[3], ost[4], ost[5])
    print("configs[3] share: %d samples of n = %d, prefix C cut at depth 10: %d nodes, %d tuples, splits %d, max_frontier %d" % (
        nsamp, n, st.reported, st.tuples, st.splits, st.max_frontier), flush=True)
    for ix in A:
        ix.close()
    for ox in O:
        ox.close()
