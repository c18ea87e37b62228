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


def _body(stream):
    return stream[stream.index(b".") + 1:]


def _feed_round_robin(srv, bodies, pieces):
    pos = [0] * len(bodies)
    k = 0
    while any(pos[i] < len(b) for i, b in enumerate(bodies)):
        for i, b in enumerate(bodies):
            if pos[i] < len(b):
                n = pieces[k % len(pieces)]
                srv.feed(i, b[pos[i]:pos[i] + n])
                pos[i] += n
                k += 1
    for i in range(len(bodies)):
        srv.end(i)


def test_server_merges_subtrees_while_the_streams_arrive(golden, pydsm_mod):
    """dsm_server_* with the prefix length given (metaserver.cpp:682-739: the reference merges while it reads its sockets): the
    subtree of every node one level below the enforced prefix is merged as soon as all connections are past it.  Output = the
    reference server's, whatever the interleaving of the connections."""
    import threading
    m = golden.manifest["sets"]["toy3"]
    names = m["names"]
    for p in ["A", "C", "G", "T", "AC", "GT", "TTG"]:
        bodies = [_body(golden.stream("toy3", n, p)) for n in names]
        for ci, (cfg, args) in enumerate(m["server_cfgs"].items()):
            if p == "TTG" and cfg != "default":
                continue
            # units one, two or three levels below the prefix (4, 16, 64 subtrees at most; the nodes between are printed between them)
            for extra in ((0, 1, 2) if cfg == "default" else ((ci + len(p)) % 3,)):
                srv = pydsm_mod.Server(len(names), prefix_len=len(p), unit_extra=extra, **server_args_to_kw(args))
                _feed_round_robin(srv, bodies, [4096, 1000, 77])
                got, st = srv.finish()
                units, peak = srv.units()
                srv.close()
                assert got == golden.server_out("toy3", cfg, p), (cfg, p, extra)
                assert st.tuples == got.count(b"\n")
                assert 1 <= units <= 4 ** (extra + 1) and peak > 0
    want = golden.server_out("toy3", "default", "A")
    os.environ["DSM_TRIE_WINDOW"] = "64"   # windows of the levels go to the card (and leave it with a unit) all the time
    try:
        bodies = [_body(golden.stream("toy3", n, "A")) for n in names]
        # one connection far ahead of the others, one far behind: a unit waits for the slowest stream
        srv = pydsm_mod.Server(len(names), prefix_len=1, unit_extra=1, emax=2.0)
        srv.feed(0, bodies[0]); srv.end(0)
        for o in range(0, len(bodies[1]), 513):
            srv.feed(1, bodies[1][o:o + 513])
        srv.end(1)
        for o in range(0, len(bodies[2]), 9999):
            srv.feed(2, bodies[2][o:o + 9999])
        srv.end(2)
        got, _ = srv.finish()
        assert srv.units()[0] == 16
        srv.close()
        assert got == want
        # the connections' reader threads side by side (what metaserver_hip does)
        srv = pydsm_mod.Server(len(names), prefix_len=1, unit_extra=2, emax=2.0)
        def reader(i):
            b = bodies[i]
            for o in range(0, len(b), 1777 + 100 * i):
                srv.feed(i, b[o:o + 1777 + 100 * i])
            srv.end(i)
        ths = [threading.Thread(target=reader, args=(i,)) for i in range(len(names))]
        for t in ths: t.start()
        for t in ths: t.join()
        got, _ = srv.finish()
        srv.close()
        assert got == want
    finally:
        del os.environ["DSM_TRIE_WINDOW"]
    # a hint shorter than the prefix: one unit (the whole subtree), same output; no hint: merged by finish()
    bodies = [_body(golden.stream("toy3", n, "AC")) for n in names]
    for plen in (1, 0, None):
        srv = pydsm_mod.Server(len(names), prefix_len=plen, unit_extra=1, emax=2.0)
        _feed_round_robin(srv, bodies, [30000])
        got, _ = srv.finish()
        srv.close()
        assert got == golden.server_out("toy3", "default", "AC"), plen
    # a hint longer than the prefix: the stream is not a single path above the unit depth -> refused, not merged wrongly
    srv = pydsm_mod.Server(len(names), prefix_len=3, emax=2.0)
    with pytest.raises(pydsm_mod.DsmError):
        _feed_round_robin(srv, bodies, [30000])
        srv.finish()
    srv.close()
    # a sample whose stream is empty (prefix absent), and a sample that lacks one of the subtrees: equal to the merge of whole streams
    t0 = pydsm_mod.Trie(golden.stream("toy3", names[0], "A"))
    t1 = pydsm_mod.Trie(golden.stream("toy3", names[1], "AC"))   # only the subtree AC of A
    empty = pydsm_mod.Trie(b"Stoy-9.")
    want2, _ = pydsm_mod.merge([t0, empty, t1], pmin=1, emax=0.0)
    srv = pydsm_mod.Server(3, prefix_len=1, unit_extra=1, pmin=1, emax=0.0)
    _feed_round_robin(srv, [_body(golden.stream("toy3", names[0], "A")), b"", _body(golden.stream("toy3", names[1], "AC"))], [5000])
    got2, _ = srv.finish()
    srv.close()
    assert got2 == want2 and got2
    for t in (t0, t1, empty):
        t.close()


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
    for prefix, cfg, sargs in [("G", "default", ["-E", "2.0"]), ("AC", "emin_m", ["-E", "1.4", "-e", "0.5", "-m", "8"]),
                               ("G", "default", ["-E", "2.0", "--prefix-len", "1"]), ("AC", "emin_m", ["-E", "1.4", "-e", "0.5", "-m", "8", "--prefix-len", "2"])]:
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
