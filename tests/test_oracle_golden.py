"""Pins the CPU oracle (oracle/dsm_oracle.cpp) against outputs of the unmodified reference:
.fmi files from its builder, raw client streams from its metaenumerate, stdout of its metaserver."""
import numpy as np
import pytest

import orc
from goldenlib import server_args_to_kw


def fasta_reads(text):
    reads, cur = [], []
    for line in text.splitlines():
        if line.startswith(">"):
            if cur:
                reads.append("".join(cur))
            cur = []
        else:
            cur.append(line)
    if cur:
        reads.append("".join(cur))
    return reads


def transform(read):
    """builder.cpp:60-104,183-201: normalise, text = reverse(read + '-' + revcomp(read))."""
    norm = []
    for ch in read:
        u = ch.upper()
        norm.append(u if u in "ACGTN" else "N")
    r = "".join(norm)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = "".join(comp[c] for c in reversed(r))
    return (r + "-" + rc)[::-1]


@pytest.mark.parametrize("setname,name", [("toy3", "toy-1"), ("toy3", "toy-3"), ("toyN", "toyN"), ("five", "five-2")])
def test_fmi_reader_and_lf_against_naive_counts(golden, setname, name):
    ix = orc.Index(golden.fmi(setname, name))
    reads = fasta_reads(golden.fasta(setname, name))
    texts = [transform(r) for r in reads]
    n = sum(len(t) + 1 for t in texts)
    assert ix.n == n
    bwt = ix.bwt()
    # symbol histogram of the BWT == histogram of the indexed texts (+ one terminator per text)
    allsyms = np.frombuffer(("".join(texts)).encode(), np.uint8)
    hist = np.bincount(allsyms, minlength=256).astype(np.uint64)
    hist[0] = len(texts)
    assert (np.bincount(bwt, minlength=256).astype(np.uint64) == hist).all()
    Cc, cnt, bits, code = ix.meta()
    assert (cnt == hist).all()
    assert (Cc == np.concatenate([[0], np.cumsum(hist)[:-1]]).astype(np.uint64)).all()
    # LF(c,i) == C[c] + #c in bwt[0..i] for every live symbol, incl. i = -1 and n-1
    rng = np.random.default_rng(1)
    pos = np.concatenate([np.array([0xFFFFFFFFFFFFFFFF, 0, n - 1, n - 2], np.uint64), rng.integers(0, n, 3000).astype(np.uint64)])
    for c in np.nonzero(hist)[0]:
        occ = np.concatenate([[0], np.cumsum(bwt == c)]).astype(np.uint64)
        with np.errstate(over="ignore"):
            want = Cc[c] + occ[(pos + np.uint64(1)).astype(np.int64)]
        got = ix.lf_batch(np.full(len(pos), c, np.uint8), pos)
        assert (got == want).all()
    # absent symbol: LF returns C[c]  (FMIndex.h:86-87)
    assert ix.lf(ord("Z"), 5) == int(Cc[ord("Z")])
    ix.close()


def test_backward_search_counts_match_text(golden):
    """freq of a pattern from LF steps == naive count over reads and their reverse complements."""
    ix = orc.Index(golden.fmi("toy3", "toy-2"))
    reads = fasta_reads(golden.fasta("toy3", "toy-2"))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    both = reads + ["".join(comp[c] for c in reversed(r)) for r in reads]
    Cc, _, _, _ = ix.meta()
    rng = np.random.default_rng(3)
    for _ in range(40):
        r = reads[int(rng.integers(0, len(reads)))]
        k = int(rng.integers(1, 12))
        p0 = int(rng.integers(0, len(r) - k))
        pat = r[p0:p0 + k]
        sp, ep = 0, ix.n - 1
        for ch in pat:  # Query.h:37-45: pushing c appends it on the right of the substring
            sp = ix.lf(ord(ch), sp - 1)
            ep = ix.lf(ord(ch), ep) - 1
        naive = sum(sum(1 for i in range(len(t) - k + 1) if t[i:i + k] == pat) for t in both)
        assert ep - sp + 1 == naive
    ix.close()


@pytest.mark.parametrize("name", ["toy-1", "toy-2", "toy-3"])
@pytest.mark.parametrize("prefix", ["A", "C", "G", "T", "AC", "GT", "TTG", "ACGTACGTACGT"])
def test_client_stream_byte_identical(golden, name, prefix):
    ix = orc.Index(golden.fmi("toy3", name))
    got, (reported, lf, ranks) = ix.enumerate(name, prefix, fmin=2)
    want = golden.stream("toy3", name, prefix)
    assert got == want
    assert reported > 0 and lf > reported and ranks > lf
    ix.close()


@pytest.mark.parametrize("prefix", ["A", "C", "G", "T"])
def test_client_stream_fmin1_maxdepth(golden, prefix):
    ix = orc.Index(golden.fmi("toy3", "toy-1"))
    got, _ = ix.enumerate("toy-1", prefix, fmin=1, maxdepth=40)
    assert got == golden.stream("toy3", "toy-1", prefix, "fmin1.M40")
    ix.close()


@pytest.mark.parametrize("prefix", ["A", "C", "G", "T"])
def test_client_stream_with_N_symbols(golden, prefix):
    ix = orc.Index(golden.fmi("toyN", "toyN"))
    got, _ = ix.enumerate("toyN", prefix, fmin=2)
    assert got == golden.stream("toyN", "toyN", prefix)
    ix.close()


def _toy3_cfgs(golden):
    out = []
    for cfg, args in golden.manifest["sets"]["toy3"]["server_cfgs"].items():
        for p in ["A", "C", "G", "T", "AC", "GT"] + (["TTG"] if cfg == "default" else []):
            out.append((cfg, args, p))
    return out


def test_server_output_byte_identical_from_golden_streams(golden):
    names = golden.manifest["sets"]["toy3"]["names"]
    for cfg, args, p in _toy3_cfgs(golden):
        streams = [golden.stream("toy3", n, p) for n in names]
        got, stats = orc.server(names, streams, **server_args_to_kw(args))
        want = golden.server_out("toy3", cfg, p)
        assert got == want, (cfg, p)
        assert stats[1] == want.count(b"\n")
    # connection order must not matter (ids come from the names list)
    streams = [golden.stream("toy3", n, "A") for n in names]
    got, _ = orc.server(names, streams[::-1], emax=2.0)
    assert got == golden.server_out("toy3", "default", "A")


def test_mine_end_to_end_matches_reference_pipeline(golden):
    for setname, cfgs in (("five", ["default"]), ("toy3", ["default", "p1", "emin_m"])):
        m = golden.manifest["sets"][setname]
        names = m["names"]
        idx = [orc.Index(golden.fmi(setname, n)) for n in names]
        for cfg in cfgs:
            kw = server_args_to_kw(m["server_cfgs"][cfg])
            got, stats = orc.mine(idx, names, ["A", "C", "G", "T"], fmin=m["fmin"], threads=2, **kw)
            want = b"".join(golden.server_out(setname, cfg, p) for p in "ACGT")
            assert got == want, (setname, cfg)
        for ix in idx:
            ix.close()


def test_mine_fmin1_maxdepth_p1(golden):
    names = golden.manifest["sets"]["toy3"]["names"]
    idx = [orc.Index(golden.fmi("toy3", n)) for n in names]
    got, _ = orc.mine(idx, names, ["A", "C", "G", "T"], fmin=1, maxdepth=24, pmin=1, pmax=1, emax=2.0)
    want = b"".join(golden.server_out("toy3", "p1_fmin1_M24", p) for p in "ACGT")
    assert got == want


def test_server_rejects_bad_streams(golden):
    names = golden.manifest["sets"]["toy3"]["names"]
    streams = [golden.stream("toy3", n, "A") for n in names]
    with pytest.raises(RuntimeError):
        orc.server(names, [b"X" + streams[0][1:]] + streams[1:], emax=2.0)      # bad start byte
    with pytest.raises(RuntimeError):
        orc.server(["a", "b", "c"], streams, emax=2.0)                          # unknown name
    with pytest.raises(RuntimeError):
        orc.server(names, [streams[0], streams[0], streams[2]], emax=2.0)       # duplicate client
    with pytest.raises(RuntimeError):
        bad = bytearray(streams[1])
        bad[len(bad) // 2] ^= 0x55
        orc.server(names, [streams[0], bytes(bad), streams[2]], emax=2.0)       # corrupted stream / R checksum


def test_unsupported_version(tmp_path, golden):
    raw = golden.read("toy3/toy-1.fasta.fmi.gz")
    p = tmp_path / "bad.fasta.fmi"
    p.write_bytes(bytes([13]) + raw[1:])
    with pytest.raises(RuntimeError):
        orc.Index(str(p))
    p.write_bytes(raw[: len(raw) // 2])
    with pytest.raises(RuntimeError):
        orc.Index(str(p))
