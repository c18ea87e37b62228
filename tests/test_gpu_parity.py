"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same inputs and against the committed reference goldens.  Bit-exact (integer/byte work; the
entropy is IEEE double computed as metaserver.cpp:379,389 -- compared through its %f text AND raw bits
of the oracle's server output)."""
import os

import numpy as np
import pytest

import orc
from goldenlib import server_args_to_kw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pydsm_mod():
    import pydsm
    pydsm.lib()
    return pydsm


@pytest.mark.parametrize("setname,name", [("toy3", "toy-1"), ("toyN", "toyN"), ("five", "five-3")])
def test_lf_batch_both_layouts_match_oracle(golden, pydsm_mod, setname, name):
    import torch
    path = golden.fmi(setname, name)
    o = orc.Index(path)
    g = pydsm_mod.Index(path, keep_wt=True)
    assert g.n == o.n and g.name == name
    Cc, cnt, bits, code = o.meta()
    gC, gcodes = g.meta()
    assert (gC == Cc).all()
    assert [(c.count, c.bits, c.code) for c in gcodes] == list(zip(cnt.tolist(), bits.tolist(), code.tolist()))
    rng = np.random.default_rng(7)
    n = o.n
    edge = np.array([0xFFFFFFFFFFFFFFFF, 0, 1, 126, 127, 128, 129, n - 2, n - 1], np.uint64)
    pos = np.concatenate([edge, rng.integers(0, n, 20000).astype(np.uint64)])
    syms = [int(s) for s in np.nonzero(cnt)[0]] + [ord("Z"), 1, 255]
    cs = np.concatenate([np.full(len(pos), s, np.uint8) for s in syms])
    ps = np.tile(pos, len(syms))
    want = o.lf_batch(cs, ps)
    got = g.lf_batch(cs, ps)
    assert (got == want).all()
    # device-resident arrays, both layouts
    dc = torch.from_numpy(cs).cuda()
    dp = torch.from_numpy(ps.view(np.int64)).cuda()
    for layout in (pydsm_mod.LAYOUT_PLANES, pydsm_mod.LAYOUT_WT):
        dout = torch.zeros(len(cs), dtype=torch.int64, device="cuda")
        g.lf_batch_dev(dc.data_ptr(), dp.data_ptr(), dout.data_ptr(), len(cs), layout)
        torch.cuda.synchronize()
        assert (dout.cpu().numpy().view(np.uint64) == want).all(), layout
    # getL over the whole BWT, and the --check sum
    allpos = np.arange(n, dtype=np.uint64)
    assert (g.getl_batch(allpos) == o.bwt()).all()
    assert g.check() == n
    g.close()
    o.close()


def test_lf_wt_layout_needs_keep_flag(golden, pydsm_mod):
    g = pydsm_mod.Index(golden.fmi("toy3", "toy-1"))
    with pytest.raises(pydsm_mod.DsmError):
        g.lf_batch_dev(0, 0, 0, 1, pydsm_mod.LAYOUT_WT)
    g.close()


@pytest.mark.parametrize("name", ["toy-1", "toy-2", "toy-3"])
def test_stream_byte_identical_to_reference_client(golden, pydsm_mod, name):
    path = golden.fmi("toy3", name)
    o = orc.Index(path)
    with pydsm_mod.Index(path) as g:
        for prefix in ["A", "C", "G", "T", "AC", "GT", "TTG", "ACGTACGTACGT"]:
            got, st = g.enumerate(prefix, fmin=2)
            assert got == golden.stream("toy3", name, prefix), prefix
            _, (rep, lf, ranks) = o.enumerate(name, prefix, fmin=2)
            assert (st.reported, st.lf_steps, st.rank_ops) == (rep, lf, ranks), prefix
    o.close()


def test_stream_fmin1_maxdepth_and_N(golden, pydsm_mod):
    with pydsm_mod.Index(golden.fmi("toy3", "toy-1")) as g:
        for prefix in "ACGT":
            got, _ = g.enumerate(prefix, fmin=1, maxdepth=40)
            assert got == golden.stream("toy3", "toy-1", prefix, "fmin1.M40"), prefix
    with pydsm_mod.Index(golden.fmi("toyN", "toyN")) as g:
        for prefix in "ACGT":
            got, _ = g.enumerate(prefix, fmin=2)
            assert got == golden.stream("toyN", "toyN", prefix), prefix


def test_stream_edge_cases(golden, pydsm_mod):
    path = golden.fmi("toy3", "toy-2")
    o = orc.Index(path)
    with pydsm_mod.Index(path) as g:
        # empty prefix = whole trie from the root (enforcepath.empty(), EnumerateQuery.cpp:31-34)
        got, st = g.enumerate("", fmin=3, maxdepth=9)
        want, (rep, lf, ranks) = o.enumerate("toy-2", "", fmin=3, maxdepth=9)
        assert got == want and st.reported == rep and st.lf_steps == lf
        # absent prefix: only the handshake goes out
        got, st = g.enumerate("ACGTTTTTTTTTTTTTTTTTTTTTTGGGGGGGGGGGGGGGGGGGGGGG", fmin=2)
        want, _ = o.enumerate("toy-2", "ACGTTTTTTTTTTTTTTTTTTTTTTGGGGGGGGGGGGGGGGGGGGGGG", fmin=2)
        assert got == want
        # huge fmin prunes everything; maxdepth 1
        for kw in (dict(fmin=10 ** 9), dict(fmin=2, maxdepth=1), dict(fmin=50, maxdepth=3)):
            got, _ = g.enumerate("G", **kw)
            want, _ = o.enumerate("toy-2", "G", **kw)
            assert got == want, kw
        with pytest.raises(pydsm_mod.DsmError):
            g.enumerate("AXG", fmin=2)
    o.close()


def test_mine_matches_reference_server_output(golden, pydsm_mod):
    m = golden.manifest["sets"]["toy3"]
    names = m["names"]
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    for cfg, args in m["server_cfgs"].items():
        kw = server_args_to_kw(args)
        for p in ["A", "C", "G", "T", "AC", "GT"] + (["TTG"] if cfg == "default" else []):
            got, st = pydsm_mod.mine(idx, p, fmin=2, **kw)
            want = golden.server_out("toy3", cfg, p)
            assert got == want, (cfg, p)
            assert st.tuples == want.count(b"\n") and st.pair_order_exact == 1
    got, _ = pydsm_mod.mine(idx, "A", fmin=1, maxdepth=24, pmin=1, pmax=1, emax=2.0)
    assert got == golden.server_out("toy3", "p1_fmin1_M24", "A")
    for ix in idx:
        ix.close()


def test_mine_five_samples_default_fmin(golden, pydsm_mod):
    m = golden.manifest["sets"]["five"]
    idx = [pydsm_mod.Index(golden.fmi("five", n)) for n in m["names"]]
    oidx = [orc.Index(golden.fmi("five", n)) for n in m["names"]]
    for p in "ACGT":
        got, st = pydsm_mod.mine(idx, p, fmin=10, emax=2.0)
        assert got == golden.server_out("five", "default", p), p
        _, ost = orc.mine(oidx, m["names"], [p], fmin=10, emax=2.0)
        assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost
    # whole trie in one call == concatenation over the four one-letter prefixes? (no: the root is shared) -- check vs oracle
    got, _ = pydsm_mod.mine(idx, "", fmin=10, emax=2.0, pmin=1)
    want, _ = orc.mine(oidx, m["names"], [""], fmin=10, emax=2.0, pmin=1)
    assert got == want
    for ix in idx + oidx:
        ix.close()


def test_mine_single_sample_fp_noise_filter(golden, pydsm_mod):
    """d = 1: entropy is FP noise around 0 and emin = 0 suppresses the negative ones."""
    path = golden.fmi("toy3", "toy-1")
    o = orc.Index(path)
    with pydsm_mod.Index(path) as g:
        for p in "ACGT":
            got, st = pydsm_mod.mine([g], p, fmin=2, pmin=1, emax=2.0)
            want, ost = orc.mine([o], ["toy-1"], [p], fmin=2, pmin=1, emax=2.0)
            assert got == want
            assert st.tuples == ost[4] and 0 < st.tuples < st.candidates
    o.close()


def test_mine_random_parameter_sweep_against_oracle(golden, pydsm_mod):
    names = golden.manifest["sets"]["toy3"]["names"]
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    oidx = [orc.Index(golden.fmi("toy3", n)) for n in names]
    rng = np.random.default_rng(11)
    for _ in range(12):
        k = int(rng.integers(1, 4))
        sub = sorted(rng.choice(3, k, replace=False).tolist())
        kw = dict(fmin=int(rng.integers(1, 6)), maxdepth=int(rng.integers(3, 30)), pmin=1, pmax=int(rng.integers(0, 3)),
                  mindepth=int(rng.integers(0, 8)), emin=float(rng.choice([0.0, 0.3])), emax=float(rng.choice([0.0, 1.0, 2.0])))
        if kw["emin"] > kw["emax"] and kw["emax"] > 0:
            kw["emin"] = 0.0
        p = "".join(rng.choice(list("ACGT"), int(rng.integers(0, 4))))
        got, _ = pydsm_mod.mine([idx[s] for s in sub], p, **kw)
        want, _ = orc.mine([oidx[s] for s in sub], [names[s] for s in sub], [p], **kw)
        assert got == want, (sub, p, kw)
    for ix in idx + oidx:
        ix.close()


def test_persistent_miner_reuses_buffers(golden, pydsm_mod):
    names = golden.manifest["sets"]["toy3"]["names"]
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    with pydsm_mod.Miner(idx, fmin=2, emax=2.0) as m:
        for p in ["T", "A", "GT", "A"]:
            got, st = m.mine(p)
            assert got == golden.server_out("toy3", "default", p)
        ps = ["A", "C", "G", "T", "AC", "GT", "TTG"]
        got, st = m.mine_many(ps)
        assert got == b"".join(golden.server_out("toy3", "default", p) for p in ps)
        assert st.tuples == got.count(b"\n")
    with pydsm_mod.Miner([idx[1]], fmin=2, stream_mode=True) as m:
        for p in ["C", "TTG", "C"]:
            got, st = m.enumerate(p)
            assert got == golden.stream("toy3", names[1], p)
        # several prefixes in one call (their bytes leave the card while the next is enumerated), one with nothing below the root
        ps = ["A", "C", "G", "T", "AC", "ACGTTTTTTTTTTTTTTTTTTTTTTGGGGGGGGGGGGGGGGGGGGGGG", "GT", "TTG", "ACGTACGTACGT", "C"]
        got, st = m.enumerate_many(ps)
        head = b"S" + names[1].encode() + b"."
        o = orc.Index(golden.fmi("toy3", names[1]))
        for p, g in zip(ps, got):
            assert g == (golden.stream("toy3", names[1], p) if len(p) < 20 else o.enumerate(names[1], p, fmin=2)[0]), p
        o.close()
        sizes, _ = m.enumerate_many(ps, discard=True)
        assert sizes == [len(g) - len(head) for g in got]
        assert m.enumerate_many([])[0] == []
        # a sink that fails: the call reports it (the remaining prefixes are dropped) and the miner stays usable
        seen = []

        def failing(k, piece):
            seen.append(k)
            if k == 1:
                raise RuntimeError("sink full")
        with pytest.raises(RuntimeError):
            m.enumerate_many(["A", "C", "G", "T"], on_piece=failing)
        assert seen and max(seen) == 1
        got, _ = m.enumerate_many(["G", "T"])
        assert got == [golden.stream("toy3", names[1], p) for p in ("G", "T")]
    with pytest.raises(pydsm_mod.DsmError):
        pydsm_mod.Miner(idx, stream_mode=True)
    # an arena too small for the frontier buffers is refused at creation
    with pytest.raises(pydsm_mod.DsmError) as e:
        pydsm_mod.mine(idx, "A", fmin=2, emax=2.0, arena_bytes=1 << 20)
    assert e.value.code in (-28, -12)
    # a small one makes levels overflow: the prefix is split into longer ones and the output does not change
    total_splits = 0
    for arena in (3 << 20, 6 << 20):
        for p in ("A", "GT", "T"):
            got, st = pydsm_mod.mine(idx, p, fmin=2, emax=2.0, arena_bytes=arena)
            assert got == golden.server_out("toy3", "default", p), (arena, p)
            total_splits += st.splits
    assert total_splits > 0
    # stream mode splits as well: the stream goes out as slices of the sub-prefixes' streams (R counts carried across them)
    o = orc.Index(golden.fmi("toy3", names[0]))
    stream_splits = 0
    for arena in (2 << 20, 3 << 20, 5 << 20):
        with pydsm_mod.Miner([idx[0]], fmin=2, stream_mode=True, arena_bytes=arena) as m:
            for p in ("A", "C", "GT", "TTG", ""):
                got, st = m.enumerate(p)
                want = golden.stream("toy3", names[0], p) if p else o.enumerate(names[0], "", fmin=2)[0]
                assert got == want, (arena, p)
                stream_splits += st.splits
            got, st = m.enumerate_many(["T", "G", "C", "A"])
            assert got == [golden.stream("toy3", names[0], p) for p in ("T", "G", "C", "A")], arena
    assert stream_splits > 0
    with pydsm_mod.Miner([idx[0]], fmin=1, maxdepth=40, stream_mode=True, arena_bytes=3 << 20) as m:
        for p in "ACGT":
            got, st = m.enumerate(p)
            assert got == golden.stream("toy3", names[0], p, "fmin1.M40"), p
            stream_splits += st.splits
    o.close()
    for ix in idx:
        ix.close()


def test_64bit_position_path(golden, pydsm_mod):
    """cfg 4 needs u64 positions (n > 2^32); the same kernels instantiated for u64 must give the same bytes."""
    names = golden.manifest["sets"]["toy3"]["names"]
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    for p in ["A", "GT"]:
        got, st = pydsm_mod.mine(idx, p, fmin=2, emax=2.0, wide=1)
        assert got == golden.server_out("toy3", "default", p)
    with pydsm_mod.Miner([idx[0]], fmin=2, stream_mode=True, wide=1) as m:
        for p in ["C", "ACGTACGTACGT"]:
            got, _ = m.enumerate(p)
            assert got == golden.stream("toy3", names[0], p)
    with pydsm_mod.Miner([idx[0]], fmin=1, maxdepth=40, stream_mode=True, wide=1) as m:
        got, _ = m.enumerate("G")
        assert got == golden.stream("toy3", names[0], "G", "fmin1.M40")
    for ix in idx:
        ix.close()


def test_thirty_samples_exact_pair_order(golden, pydsm_mod):
    """d = 30 > 13: the reference's reader sets rehash and ids share buckets; pair order and FP summation order must still match."""
    m = golden.manifest["sets"]["many30"]
    idx = [pydsm_mod.Index(golden.fmi("many30", n)) for n in m["names"]]
    for cfg, args in m["server_cfgs"].items():
        for p in m["prefixes"]:
            got, st = pydsm_mod.mine(idx, p, fmin=m["fmin"], maxdepth=m["maxdepth"], **server_args_to_kw(args))
            assert got == golden.server_out("many30", cfg, p), (cfg, p)
            assert st.pair_order_exact == 1
    for ix in idx:
        ix.close()


def test_dense_sweeps_on_every_compact_level(golden, pydsm_mod, monkeypatch):
    """Several samples: on wide levels a sample's own nodes are packed into full tiles (expand.hip, DENSE) -- handles of the children from
    runs of lanes of one union tile plus what the previous dense tile held of it, planes put together in LDS, column entries of the
    absent nodes cleared by the producer.  DSM_DENSE_MIN=0 sends every compact level of the small golden sets through it (items of
    four union tiles): tuples against the reference server's stdout and the oracle, all six counters against the oracle, with 32- and
    64-bit positions, d = 3, 5 and 30, also with budgets small enough to split prefixes."""
    monkeypatch.setenv("DSM_DENSE_MIN", "0")
    for setname, fmin in (("toy3", 2), ("five", 10)):
        m = golden.manifest["sets"][setname]
        idx = [pydsm_mod.Index(golden.fmi(setname, n)) for n in m["names"]]
        oidx = [orc.Index(golden.fmi(setname, n)) for n in m["names"]]
        for wide in (0, 1):
            for p in ["", "A", "C", "G", "T", "GT"]:
                kw = dict(fmin=fmin, emax=2.0, pmin=1 if p == "" else 2)
                got, st = pydsm_mod.mine(idx, p, wide=wide, **kw)
                want, ost = orc.mine(oidx, m["names"], [p], **kw)
                assert got == want, (setname, wide, p)
                assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (setname, wide, p)
                if len(p) == 1:
                    assert got == golden.server_out(setname, "default", p)
        got, st = pydsm_mod.mine(idx, "A", fmin=fmin, emax=2.0, arena_bytes=3 << 20)   # forced prefix splits
        want, _ = orc.mine(oidx, m["names"], ["A"], fmin=fmin, emax=2.0)
        assert got == want and (setname == "toy3" or st.splits > 0)
        for ix in idx + oidx:
            ix.close()
    m = golden.manifest["sets"]["many30"]
    idx = [pydsm_mod.Index(golden.fmi("many30", n)) for n in m["names"]]
    for cfg, args in m["server_cfgs"].items():
        for p in m["prefixes"]:
            got, st = pydsm_mod.mine(idx, p, fmin=m["fmin"], maxdepth=m["maxdepth"], **server_args_to_kw(args))
            assert got == golden.server_out("many30", cfg, p), (cfg, p)
    for ix in idx:
        ix.close()


def test_dense_sweep_of_a_thin_sample(pydsm_mod, tmp_path, monkeypatch):
    """Dense sweeps when a sample holds a few dozen nodes among tens of thousands of the union level (40 reads beside 30 000: its dense
    tiles span hundreds of union tiles, most items hold none of its nodes), and a third sample shares nothing with the others.
    Tuples and counters against the oracle, both position widths."""
    from pydsm import builder
    import torch
    monkeypatch.setenv("DSM_DENSE_MIN", "0")
    rng = np.random.default_rng(2024)
    genome = rng.integers(0, 4, 60000)
    other = rng.integers(0, 4, 3000)

    def reads(g, k, ln=60):
        st = rng.integers(0, len(g) - ln, k)
        return np.stack([g[a:a + ln] for a in st]).astype(np.uint8)
    sets = [reads(genome, 30000), reads(genome, 40), reads(other, 400)]
    paths = []
    for k, codes in enumerate(sets):
        p = tmp_path / ("thin%d.fasta.fmi" % k)
        builder.build_from_codes(torch.from_numpy(codes), str(p))
        paths.append(str(p))
    idx = [pydsm_mod.Index(p) for p in paths]
    oidx = [orc.Index(p) for p in paths]
    names = [ix.name for ix in idx]
    for wide in (0, 1):
        for p, kw in (("", dict(fmin=1, maxdepth=14, pmin=1, emax=9.0)), ("G", dict(fmin=2, pmin=1, emax=9.0)), ("T", dict(fmin=1, maxdepth=20, pmin=1, pmax=2, emax=9.0))):
            # (pmin 1 throughout: the reference server loses its place in the streams on single-reader nodes above depth 7 otherwise.
            # fmin 1 with a depth cut: a node of one occurrence AT the cut costs the reference one getL when it was reached inside
            # followOneBranch and nothing when nextSymbol returned at once -- the records carry that bit, EnumerateQuery.cpp:105-153)
            got, st = pydsm_mod.mine(idx, p, wide=wide, **kw)
            want, ost = orc.mine(oidx, names, [p], threads=4, **kw)
            assert got == want, (wide, p)
            assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (wide, p)
    assert st.max_frontier > 10000
    for ix in idx + oidx:
        ix.close()


def test_candidate_block_overflow_falls_back(pydsm_mod, tmp_path, monkeypatch):
    """One sample: the advance sweep stores a level's candidates in a block of their own arena, taken before their number is known.  A
    block too small for a level (DSM_CAND_ARENA: 16 KB = 1024 records, then nothing at all) makes that level store them the old way
    (scan of the words, cand_store_kernel); levels that fit and levels that do not alternate within one prefix."""
    from pydsm import builder
    import torch
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 40000)
    st = rng.integers(0, len(genome) - 60, 20000)
    codes = np.stack([genome[a:a + 60] for a in st]).astype(np.uint8)
    p = tmp_path / "one.fasta.fmi"
    builder.build_from_codes(torch.from_numpy(codes), str(p))
    o = orc.Index(str(p))
    with pydsm_mod.Index(str(p)) as g:
        want = {}
        for pre in ("", "C"):
            want[pre] = orc.mine([o], [g.name], [pre], fmin=2, pmin=1, emax=2.0)
        for arena in (None, "16384", "256"):
            if arena:
                monkeypatch.setenv("DSM_CAND_ARENA", arena)
            for pre in ("", "C"):
                got, st_ = pydsm_mod.mine([g], pre, fmin=2, pmin=1, emax=2.0)
                w, ost = want[pre]
                assert got == w, (arena, pre)
                assert (st_.reported, st_.lf_steps, st_.rank_ops, st_.union_nodes, st_.tuples, st_.pairs) == ost, (arena, pre)
                assert st_.max_frontier > 5000 and st_.tuples > 2000
    o.close()


def _downgrade_fmi(raw, ver):
    """Rewrite a v17 .fmi as v16 / v15 / v14 (FMIndex.cpp:267-290, HuffWT.h:21-37): v<16 stores code counts as u32, v14 stores C[] as u32."""
    import struct
    pos = 1
    n, sr = struct.unpack_from("<QI", raw, pos)
    pos += 12
    C = struct.unpack_from("<256Q", raw, pos)
    pos += 2048
    (bwt_end,) = struct.unpack_from("<Q", raw, pos)
    pos += 8
    codes = [struct.unpack_from("<QII", raw, pos + 16 * i) for i in range(256)]
    pos += 16 * 256
    out = bytes([ver]) + struct.pack("<QI", n, sr)
    out += struct.pack("<256I", *C) if ver == 14 else struct.pack("<256Q", *C)
    out += struct.pack("<Q", bwt_end)
    for cnt, bits, code in codes:
        out += struct.pack("<III", cnt, bits, code) if ver < 16 else struct.pack("<QII", cnt, bits, code)
    return out + raw[pos:]


def test_older_fmi_versions_and_degenerate_alphabets(golden, pydsm_mod, tmp_path):
    raw = golden.read("toy3/toy-2.fasta.fmi.gz")
    ref = pydsm_mod.Index(golden.fmi("toy3", "toy-2"))
    want, _ = ref.enumerate("GT", fmin=2)
    for ver in (16, 15, 14):
        p = tmp_path / ("v%d.toy-2.fasta.fmi" % ver)
        p.write_bytes(_downgrade_fmi(raw, ver))
        o = orc.Index(str(p))
        with pydsm_mod.Index(str(p)) as g:
            assert g.n == ref.n == o.n
            got, _ = g.enumerate("GT", fmin=2)
            assert got[got.index(b"."):] == want[want.index(b"."):]     # same node stream (the sample name differs)
            assert got == o.enumerate(g.name, "GT", fmin=2)[0]
        o.close()
    # a version-14 file whose 32-bit C[] wrapped (FMIndex.cpp:346-356: the reference recounts it; here the wrap is played at 6 bits on
    # the toy index): the loader repairs it from the code table, the oracle as the reference does, and both give the pristine stream
    import struct
    v14 = bytearray(_downgrade_fmi(raw, 14))
    C = list(struct.unpack_from("<256I", v14, 13))
    assert any(c >= 64 for c in C)
    struct.pack_into("<256I", v14, 13, *[c % 64 for c in C])
    p = tmp_path / "wrapped.toy-2.fasta.fmi"
    p.write_bytes(bytes(v14))
    o = orc.Index(str(p))
    with pydsm_mod.Index(str(p)) as g:
        got, _ = g.enumerate("GT", fmin=2)
        assert got[got.index(b"."):] == want[want.index(b"."):]
        assert got == o.enumerate(g.name, "GT", fmin=2)[0]
        assert g.check() == g.n
    o.close()
    ref.close()
    # reads without any A,C,G,T: nothing to enumerate, LF still answers for the symbols that exist
    from pydsm import builder
    p = tmp_path / "onlyN.fasta.fmi"
    builder.build_from_fasta(">a\nNNNNNN\n>b\nNNN\n", str(p))
    o = orc.Index(str(p))
    with pydsm_mod.Index(str(p)) as g:
        assert g.check() == g.n == o.n
        for prefix in ("A", ""):
            got, st = g.enumerate(prefix, fmin=1)
            assert got == o.enumerate(g.name, prefix, fmin=1)[0] and st.reported == 0
        pos = np.arange(g.n, dtype=np.uint64)
        for c in (0, ord("-"), ord("N"), ord("A")):
            assert (g.lf_batch(np.full(g.n, c, np.uint8), pos) == o.lf_batch(np.full(g.n, c, np.uint8), pos)).all()
        text, _ = pydsm_mod.mine([g], "", fmin=1, pmin=1, emax=2.0)
        assert text == b""
    o.close()
    # a single read: every node has frequency 1 or 2, deep unary chains (followOneBranch territory), depth == read length
    p = tmp_path / "one.fasta.fmi"
    builder.build_from_fasta(">a\nACGTTGCAACGGATTACAGATTACA\n", str(p))
    o = orc.Index(str(p))
    with pydsm_mod.Index(str(p)) as g:
        for kw in (dict(fmin=1), dict(fmin=2), dict(fmin=1, maxdepth=7)):
            for prefix in ("", "A", "GATTACA"):
                got, st = g.enumerate(prefix, **kw)
                want2, (rep, lf, ranks) = o.enumerate(g.name, prefix, **kw)
                assert got == want2, (kw, prefix)
                assert st.reported == rep
        got, _ = pydsm_mod.mine([g], "", fmin=1, pmin=1, emax=0.0)
        want3, _ = orc.mine([o], [g.name], [""], fmin=1, pmin=1, emax=0.0)
        assert got == want3
    o.close()


def test_index_residency_offload_and_reload(golden, pydsm_mod):
    """an index gives its HBM back and comes back with one async copy; queries and miners refuse an offloaded
    index; results after the reload are unchanged (a miner created before survives the cycle)."""
    names = golden.manifest["sets"]["toy3"]["names"]
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    pos = np.arange(0, idx[0].n, 7, dtype=np.uint64)
    cs = np.full(len(pos), ord("C"), np.uint8)
    before = idx[0].lf_batch(cs, pos)
    full = idx[0].device_bytes()
    with pydsm_mod.Miner(idx, fmin=2, emax=2.0) as m:
        want = golden.server_out("toy3", "default", "GT")
        assert m.mine("GT")[0] == want
        for ix in idx:
            ix.offload()
            assert not ix.resident and ix.device_bytes() < full // 4
            ix.offload()                                   # idempotent
        with pytest.raises(pydsm_mod.DsmError):
            idx[0].lf_batch(cs, pos)
        with pytest.raises(pydsm_mod.DsmError):
            m.mine("GT")
        for ix in idx:
            ix.reload()
            assert ix.resident and ix.device_bytes() == full
        assert (idx[0].lf_batch(cs, pos) == before).all()
        assert m.mine("GT")[0] == want
        idx[1].offload(); idx[1].reload()                 # second cycle reuses the pinned copy
        assert m.mine("A")[0] == golden.server_out("toy3", "default", "A")
    k = pydsm_mod.Index(golden.fmi("toy3", names[0]), keep_wt=True)
    with pytest.raises(pydsm_mod.DsmError):
        k.offload()
    k.close()
    for ix in idx:
        ix.close()


def test_seventy_and_273_samples_against_oracle(pydsm_mod, tmp_path):
    """More than 64 samples take the widest order kernel (reader sets rehash up to 541 buckets) and, for the distance matrices,
    the global-atomic path; 273 is the reference's MAX_READERS.  Small synthetic samples from one genome, oracle as judge."""
    from pydsm import builder
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, 1500)
    for d, fmin, pfx in ((70, 2, ["A", "GT"]), (273, 2, ["C"])):
        paths = []
        for s in range(d):
            starts = rng.integers(0, len(genome) - 40, 60)
            codes = np.stack([genome[a:a + 40] for a in starts]).astype(np.uint8)
            flip = rng.random(codes.shape) < 0.01
            codes = np.where(flip, (codes + rng.integers(1, 4, codes.shape)) % 4, codes).astype(np.uint8)
            p = tmp_path / ("s%03d_%d.fasta.fmi" % (s, d))
            import torch
            builder.build_from_codes(torch.from_numpy(codes), str(p))
            paths.append(str(p))
        idx = [pydsm_mod.Index(p) for p in paths]
        oidx = [orc.Index(p) for p in paths]
        names = [ix.name for ix in idx]
        kw = dict(fmin=fmin, maxdepth=12, pmin=1, emax=9.0)  # pmin 1: the reference server desynchronises on single-reader nodes above depth 7 otherwise
        for p in pfx:
            got, st = pydsm_mod.mine(idx, p, **kw)
            want, ost = orc.mine(oidx, names, [p], threads=4, **kw)
            assert got == want, (d, p)
            assert st.pair_order_exact == 1 and st.tuples == ost[4] and st.tuples > 100
            kw1 = dict(kw, pmax=1)      # sample-specific substrings: the order kernels are skipped (single readers need no order)
            got, st = pydsm_mod.mine(idx, p, **kw1)
            want, ost = orc.mine(oidx, names, [p], threads=4, **kw1)
            assert got == want and st.tuples == ost[4], (d, p, "pmax=1")
        for ix in idx + oidx:
            ix.close()


def test_glibc_of_the_gpu_box_is_recent_enough():
    """The exact entropy (metaserver.cpp:379,389) and the keep table of the LF-step kernel are computed with the box's libm: glibc's
    log() is the correctly rounded-in-practice implementation since 2.28 (the goldens were made with 2.35, tests/golden/MANIFEST.json);
    an older libm could print a different last digit.  bench.py records the version in `detail.glibc`."""
    v = os.confstr("CS_GNU_LIBC_VERSION")
    assert v and v.startswith("glibc "), v
    major, minor = (int(x) for x in v.split()[1].split(".")[:2])
    assert (major, minor) >= (2, 28), v
