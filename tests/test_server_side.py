"""Server side of the wire protocol on the GPU box: streams captured from the UNMODIFIED reference client (goldens) go
through dsm_trie_parse + dsm_merge and must reproduce the reference server's stdout; the metaserver_hip executable is fed
by unmodified reference clients over TCP (oracle/_ref/metaenumerate travels with the repo)."""
import os
import socket
import subprocess
import time

import pytest

from goldenlib import server_args_to_kw, wait_listen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsm-framework_amd", "host")
REF = os.path.join(ROOT, "oracle", "_ref")


@pytest.fixture(scope="module")
def pydsm_mod():
    import pydsm
    pydsm.lib()
    return pydsm


def test_merge_of_reference_streams_matches_reference_server(golden, pydsm_mod):
    m = golden.manifest["sets"]["toy3"]
    names = m["names"]
    for p in ["A", "C", "G", "T", "AC", "GT", "TTG"]:
        tries = [pydsm_mod.Trie(golden.stream("toy3", n, p)) for n in names]
        assert [t.name for t in tries] == names
        for cfg, args in m["server_cfgs"].items():
            if p == "TTG" and cfg != "default":
                continue
            got, st = pydsm_mod.merge(tries, **server_args_to_kw(args))
            assert got == golden.server_out("toy3", cfg, p), (cfg, p)
            assert st.tuples == got.count(b"\n")
        for t in tries:
            t.close()
    # connection order does not matter: ids are positions in the names list -> merge wants them in id order
    tries = [pydsm_mod.Trie(golden.stream("toy3", n, "A")) for n in reversed(names)]
    got, _ = pydsm_mod.merge(list(reversed(tries)), emax=2.0)
    assert got == golden.server_out("toy3", "default", "A")


def test_merge_fmin1_streams_and_absent_sample(golden, pydsm_mod):
    names = golden.manifest["sets"]["toy3"]["names"]
    # our own client's bytes for the other two samples (byte-identical to the reference client's, see test_gpu_parity)
    idx = [pydsm_mod.Index(golden.fmi("toy3", n)) for n in names]
    streams = [golden.stream("toy3", names[0], "A", "fmin1.M40")] + [ix.enumerate("A", fmin=1, maxdepth=40)[0] for ix in idx[1:]]
    tries = [pydsm_mod.Trie(s) for s in streams]
    got, _ = pydsm_mod.merge(tries, pmin=1, pmax=1, emax=2.0)
    want, _ = pydsm_mod.mine(idx, "A", fmin=1, maxdepth=40, pmin=1, pmax=1, emax=2.0)
    assert got == want and len(got) > 0
    # a sample that sent only its handshake (prefix absent) is simply absent from every node
    empty = pydsm_mod.Trie(b"Stoy-9.")
    assert empty.nodes == 0
    got2, _ = pydsm_mod.merge([tries[0], empty], pmin=1, emax=0.0)
    only, _ = pydsm_mod.merge([tries[0]], pmin=1, emax=0.0)
    strip = lambda t: [(ln.split()[0], ln.split()[2:]) for ln in t.splitlines()]  # entropy differs: sumN starts at d (metaserver.cpp:357)
    assert strip(got2) == strip(only) and got2
    for ix in idx:
        ix.close()


def test_thirty_streams(golden, pydsm_mod):
    m = golden.manifest["sets"]["many30"]
    idx = [pydsm_mod.Index(golden.fmi("many30", n)) for n in m["names"]]
    for p in m["prefixes"]:
        tries = [pydsm_mod.Trie(ix.enumerate(p, fmin=m["fmin"], maxdepth=m["maxdepth"])[0]) for ix in idx]
        for cfg, args in m["server_cfgs"].items():
            got, _ = pydsm_mod.merge(tries, **server_args_to_kw(args))
            assert got == golden.server_out("many30", cfg, p), (cfg, p)
        for t in tries:
            t.close()
    for ix in idx:
        ix.close()


def test_streams_fed_piece_by_piece_give_the_same_merge(golden, pydsm_mod):
    """dsm_trie_stream_begin/feed/end (what metaserver_hip's reader threads call while the bytes arrive): pieces of one byte, of
    sizes around the decoder's look-ahead and of sizes that make windows of a level go to the card before the stream ends."""
    m = golden.manifest["sets"]["toy3"]
    names = m["names"]
    for p, pieces in (("A", [1]), ("C", [23, 24, 25, 7]), ("G", [4096]), ("T", [100000, 3])):
        tries = [pydsm_mod.Trie(golden.stream("toy3", n, p), pieces=pieces) for n in names]
        whole = [pydsm_mod.Trie(golden.stream("toy3", n, p)) for n in names]
        assert [t.nodes for t in tries] == [t.nodes for t in whole]
        got, st = pydsm_mod.merge(tries, emax=2.0)
        assert got == golden.server_out("toy3", "default", p), (p, pieces)
        for t in tries + whole:
            t.close()
    empty = pydsm_mod.Trie(b"Stoy-9.", pieces=[5])
    assert empty.nodes == 0
    empty.close()
    # small upload windows, so that parts of a level leave the host (and the card's level buffers grow) while the stream comes in
    os.environ["DSM_TRIE_WINDOW"] = "48"
    try:
        for p, pieces in (("A", [1000]), ("GT", [37])):
            tries = [pydsm_mod.Trie(golden.stream("toy3", n, p), pieces=pieces) for n in names]
            got, _ = pydsm_mod.merge(tries, emax=2.0)
            assert got == golden.server_out("toy3", "default", p), (p, pieces)
            for t in tries:
                t.close()
    finally:
        del os.environ["DSM_TRIE_WINDOW"]
    s = golden.stream("toy3", "toy-1", "C")
    body = s[s.index(b".") + 1:]
    for bad in (body[:-1], body.replace(b"(C(A", b"(C(X", 1), body + b")", bytes([body[0], body[1]]) + body[4:]):
        for pieces in ([1], [24], [5000]):
            with pytest.raises(pydsm_mod.DsmError):
                pydsm_mod.Trie(bad, pieces=pieces)


def test_corrupt_streams_are_rejected(golden, pydsm_mod):
    s = golden.stream("toy3", "toy-1", "C")
    body = s[s.index(b".") + 1:]
    bad = [body[:-1],                                    # truncated
           body.replace(b"(C(A", b"(C(X", 1),            # not a DNA symbol
           body[:-2] + b"Q)",                            # bad left char
           body + b")",                                  # unbalanced
           bytes([body[0], body[1]]) + body[4:]]         # dropped child open: R checksums no longer agree
    for b in bad:
        with pytest.raises(pydsm_mod.DsmError):
            pydsm_mod.Trie(b)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "metaenumerate")), reason="oracle/_ref/metaenumerate not present")
def test_reference_clients_feed_our_server(golden, tmp_path):
    names = golden.manifest["sets"]["toy3"]["names"]
    for prefix, cfg, sargs in [("G", "default", ["-E", "2.0"]), ("AC", "emin_m", ["-E", "1.4", "-e", "0.5", "-m", "8"])]:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = open(tmp_path / "out.txt", "wb")
        srv = subprocess.Popen([os.path.join(HOST, "metaserver_hip"), "-p", str(port)] + sargs, stdin=subprocess.PIPE, stdout=out,
                               stderr=subprocess.PIPE)
        srv.stdin.write(("\n".join(names) + "\n").encode())
        srv.stdin.close()
        assert wait_listen(port, srv), "metaserver_hip did not start listening"
        clients = []
        for n in names:
            c = subprocess.Popen([os.path.join(REF, "metaenumerate"), "--fmin", "2", golden.fmi("toy3", n)], stdin=subprocess.PIPE,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            c.stdin.write(("127.0.0.1 %d %s\n" % (port, prefix)).encode())
            c.stdin.close()
            clients.append(c)
        for c in clients:
            assert c.wait(timeout=120) == 0
        assert srv.wait(timeout=120) == 0, srv.stderr.read()
        out.close()
        assert open(tmp_path / "out.txt", "rb").read() == golden.server_out("toy3", cfg, prefix), (prefix, cfg)
