"""Own .fmi writer (data preparation for the benchmark configurations) against the reference builder:
same BWT, same header fields, and -- when oracle/_ref is present -- the unmodified reference
metaenumerate reads our file and emits the same streams."""
import os
import socket
import subprocess
import threading

import numpy as np
import pytest
import torch

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture(scope="module")
def builder():
    from pydsm import builder as b
    return b


@pytest.mark.parametrize("setname,name", [("toy3", "toy-2"), ("toyN", "toyN"), ("five", "five-1")])
def test_same_bwt_and_tables_as_reference_builder(golden, builder, tmp_path, setname, name):
    out = str(tmp_path / (name + ".fasta.fmi"))
    info = builder.build_from_fasta(golden.fasta(setname, name), out)
    ref = orc.Index(golden.fmi(setname, name))
    own = orc.Index(out)
    assert own.n == ref.n == info["n"]
    assert (own.bwt() == ref.bwt()).all()
    for a, b in zip(own.meta(), ref.meta()):
        assert (a == b).all()
    # Huffman tie order follows libstdc++'s heap, so the file itself is byte-identical
    assert open(out, "rb").read() == open(golden.fmi(setname, name), "rb").read()
    # and the oracle enumerates the same stream from either file
    s1, _ = own.enumerate(name, "AC", fmin=2)
    s2, _ = ref.enumerate(name, "AC", fmin=2)
    assert s1 == s2
    own.close()
    ref.close()


def test_codes_path_equals_fasta_path(builder, tmp_path):
    codes = builder.synth_reads(seed=7, nreads=300, rlen=40, genome_len=2000, sub_rate=0.01)
    a = str(tmp_path / "a.fmi")
    b = str(tmp_path / "b.fmi")
    builder.build_from_codes(codes, a)
    builder.build_from_fasta(builder.codes_to_fasta(codes), b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # synthetic sets are seeded: same seed, same reads; samples share the genome
    again = builder.synth_reads(seed=7, nreads=300, rlen=40, genome_len=2000, sub_rate=0.01)
    assert torch.equal(codes, again)


def test_variable_length_and_edge_reads(builder, tmp_path):
    fasta = ">a\nACGTNNAC\n>b\nA\n>c\nacgtRYacgtacgtacgtt\n>d\nTTTTTTTTTTTT\n>e\nTTTTTTTTTTTT\n"
    out = str(tmp_path / "v.fasta.fmi")
    builder.build_from_fasta(fasta, out)
    if os.path.exists(os.path.join(REF, "builder")):
        fa = tmp_path / "r.fasta"
        fa.write_text(fasta)
        subprocess.run([os.path.join(REF, "builder"), str(fa)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == open(str(fa) + ".fmi", "rb").read()
    ix = orc.Index(out)
    assert ix.n == sum(2 * len(r) + 2 for r in ["ACGTNNAC", "A", "acgtRYacgtacgtacgtt", "T" * 12, "T" * 12])
    ix.close()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "metaenumerate")), reason="needs oracle/_ref (build container only)")
def test_reference_client_reads_our_file(golden, builder, tmp_path):
    name = "toy-3"
    out = str(tmp_path / (name + ".fasta.fmi"))
    builder.build_from_fasta(golden.fasta("toy3", name), out)
    srv = socket.socket()
    srv.bind(("127.0.0.1", 0))
    srv.listen(1)
    got = []

    def sink():
        c, _ = srv.accept()
        buf = []
        while True:
            b = c.recv(1 << 16)
            if not b:
                break
            buf.append(b)
        got.append(b"".join(buf))

    t = threading.Thread(target=sink)
    t.start()
    subprocess.run([os.path.join(REF, "metaenumerate"), "--fmin", "2", out], input=("127.0.0.1 %d GT\n" % srv.getsockname()[1]).encode(),
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t.join()
    assert got[0] == golden.stream("toy3", name, "GT")


# ---- the GPU suffix sort of the writer (csrc/bwt.hip, dsm_bwt_build) ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("setname,name", [("toy3", "toy-1"), ("toyN", "toyN"), ("five", "five-4")])
def test_hip_bwt_writes_the_reference_builders_file(golden, builder, tmp_path, setname, name):
    out = str(tmp_path / (name + ".fasta.fmi"))
    builder.build_from_fasta(golden.fasta(setname, name), out, device="cuda")
    assert open(out, "rb").read() == open(golden.fmi(setname, name), "rb").read()


@pytest.mark.gpu
def test_hip_bwt_equals_prefix_doubling_on_awkward_collections(builder, monkeypatch):
    """Variable lengths, duplicate strings (ties resolved by text order), homopolymers, strings that are suffixes of others, N and
    '-' symbols, single-symbol strings; also with tiny batches so that every bucket boundary is crossed."""
    rng = np.random.default_rng(5)
    reads = ["A", "T", "ACGT", "ACGT", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", "TTTT", "NNNN", "ACGTNACGT",
             "GATTACA" * 9, "GATTACA" * 9 + "G", "C" * 70, "C" * 71]
    for _ in range(400):
        ln = int(rng.integers(1, 120))
        reads.append("".join(rng.choice(list("ACGT"), ln)))
    for _ in range(60):
        reads.append(reads[int(rng.integers(0, len(reads)))])  # exact duplicates
    sym, starts = builder.texts_from_reads(reads)
    want = builder.bwt_collection(torch.from_numpy(sym), starts=torch.from_numpy(starts), hip=False)
    for batch in (None, "1000", "37"):
        if batch is None:
            monkeypatch.delenv("DSM_BWT_BATCH", raising=False)
        else:
            monkeypatch.setenv("DSM_BWT_BATCH", batch)
        got = builder.bwt_collection(torch.from_numpy(sym).cuda(), hip=True)
        assert torch.equal(got.cpu(), want), batch


@pytest.mark.gpu
def test_hip_bwt_equals_prefix_doubling_on_a_read_set(builder):
    codes = builder.synth_reads(seed=11, nreads=200000, rlen=100, genome_len=500000, sub_rate=0.005, device="cuda", private_frac=0.05)
    sym = builder.texts_from_codes(codes)
    want = builder.bwt_collection(sym, width=202, hip=False)
    got = builder.bwt_collection(sym, hip=True)
    assert torch.equal(got, want)
    # many small batches give the same order
    os.environ["DSM_BWT_BATCH"] = "3000000"
    try:
        assert torch.equal(builder.bwt_collection(sym, hip=True), want)
    finally:
        del os.environ["DSM_BWT_BATCH"]


# ---- the library's builder (csrc/fmiwrite.hip, host/builder_hip): the reference `builder` as a C++ drop-in -------------------
@pytest.mark.gpu
@pytest.mark.parametrize("setname,name", [("toy3", "toy-1"), ("toyN", "toyN"), ("five", "five-3")])
def test_builder_hip_cli_writes_the_reference_builders_file(golden, tmp_path, setname, name):
    """host/builder_hip <fasta> (FASTA parse, normalisation, BWT, Huffman shape, bit vectors, rank directories, container -- all
    behind the C ABI) writes byte for byte the file the reference builder wrote for the same FASTA (builder.cpp:329-472)."""
    exe = os.path.join(ROOT, "dsm-framework_amd", "host", "builder_hip")
    fa = tmp_path / (name + ".fasta")
    fa.write_text(golden.fasta(setname, name))
    r = subprocess.run([exe, "-v", str(fa)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert open(str(fa) + ".fmi", "rb").read() == open(golden.fmi(setname, name), "rb").read()
    # an output name: <output>.fmi (builder.cpp:391-393, 425-426)
    r = subprocess.run([exe, str(fa), str(tmp_path / "other")], capture_output=True)
    assert r.returncode == 0 and open(str(tmp_path / "other.fmi"), "rb").read() == open(golden.fmi(setname, name), "rb").read()
    r = subprocess.run([exe, str(tmp_path / "missing.fasta")], capture_output=True)
    assert r.returncode == 1 and b"unable to read input file" in r.stderr


@pytest.mark.gpu
def test_library_writer_equals_torch_writer(builder, tmp_path):
    """dsm_fmi_write against the torch tooling on awkward collections: variable lengths, N and lower case, a read set whose
    bit vectors are not multiples of 64 / 256 bits, and a larger seeded set."""
    fasta = ">a\nACGTNNAC\n>b\nA\n>c\nacgtRYacgtacgtacgtt\n>d\nTTTTTTTTTTTT\n>e\nTTTTTTTTTTTT\n"
    fa = tmp_path / "v.fasta"
    fa.write_text(fasta)
    builder.build_fasta_hip(str(fa), str(tmp_path / "v.lib.fmi"))
    builder.build_from_fasta(fasta, str(tmp_path / "v.torch.fmi"))
    assert open(str(tmp_path / "v.lib.fmi"), "rb").read() == open(str(tmp_path / "v.torch.fmi"), "rb").read()
    codes = builder.synth_reads(seed=11, nreads=20011, rlen=37, genome_len=50000, sub_rate=0.01, device="cuda")
    sym = builder.texts_from_codes(codes)
    bwt = builder.bwt_collection(sym, hip=True)
    builder.write_fmi(bwt, str(tmp_path / "s.lib.fmi"), 20011, 76, hip=True)
    builder.write_fmi(bwt.cpu(), str(tmp_path / "s.torch.fmi"), 20011, 76, hip=False)
    assert open(str(tmp_path / "s.lib.fmi"), "rb").read() == open(str(tmp_path / "s.torch.fmi"), "rb").read()
    # a file without a trailing newline: the reference reads rows with getline(...).good() and drops the unterminated one
    fb = tmp_path / "w.fasta"
    fb.write_text(">a\nACGT\n>b\nGGGTTT")
    info = builder.build_fasta_hip(str(fb), str(tmp_path / "w.fmi"))
    assert info["number_of_texts"] == 1 and info["n"] == 10
