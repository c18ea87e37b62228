"""Host logic without a GPU: the server-side decoder of client wire streams (csrc/stream_parse.h: TrieReader's token rules,
ServerSocket's varint, the R checksums) on the committed reference streams, and its rejection of damaged ones."""
import os
import subprocess

import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("native") / "stream_parse_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-o", out,
                    os.path.join(ROOT, "tests", "native", "stream_parse_check.cpp")], check=True)
    return out


def _run(exe, tmp_path, body, piece=0):
    p = tmp_path / "s.bin"
    p.write_bytes(body)
    r = subprocess.run([exe, str(p)] + ([str(piece)] if piece else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode().strip(), r.stderr.decode()


def _body(stream):
    return stream[stream.index(b".") + 1:]      # drop 'S' name '.'


def test_reference_streams_parse_and_count(golden, exe, tmp_path):
    names = golden.manifest["sets"]["toy3"]["names"]
    sigs = set()
    for name in names:
        o = orc.Index(golden.fmi("toy3", name))
        for prefix in ("A", "GT", "TTG", "ACGTACGTACGT"):
            stream = golden.stream("toy3", name, prefix)
            rc, out, err = _run(exe, tmp_path, _body(stream))
            assert rc == 0 and out.startswith("ok "), (out, err)
            fields = dict(kv.split("=") for kv in out.split()[1:])
            _, (reported, _, _) = o.enumerate(name, prefix, fmin=2)
            assert int(fields["nodes"]) == reported                                 # the client's own count of '(' tokens
            assert int(fields["stored"]) == reported + 1                       # plus the root
            sigs.add(fields["sig"])
        o.close()
    assert len(sigs) > 6
    rc, out, _ = _run(exe, tmp_path, b"")                                      # a client that found nothing
    assert rc == 0 and "nodes=0" in out


def test_damaged_streams_are_rejected(golden, exe, tmp_path):
    name = golden.manifest["sets"]["toy3"]["names"][0]
    body = _body(golden.stream("toy3", name, "GT"))
    assert _run(exe, tmp_path, body)[0] == 0
    i = body.index(b"R")
    for bad in (body[:-1],                                   # truncated: the last ')' is missing
                body[: len(body) // 2],                      # cut in the middle
                body[:i + 1] + bytes([body[i + 1] ^ 1]) + body[i + 2:],   # wrong R checksum (TrieReader.h:86-95)
                body.replace(b"(G", b"(X", 1),               # not a DNA byte (TrieReader.h:58-63)
                b")" + body,
                body + b"(A"):
        rc, out, err = _run(exe, tmp_path, bad)
        assert rc == 1 and out.startswith("error:"), (out, err)
        assert "AddressSanitizer" not in err and "runtime error" not in err


def test_incremental_decode_equals_the_one_shot_parse(golden, exe, tmp_path):
    """The decoder fed in pieces (1 byte, sizes around its 24-byte look-ahead, larger ones), its final entries taken away
    after every piece (what dsm_trie_stream does on the way to the card): same level arrays as parsing the whole stream."""
    names = golden.manifest["sets"]["toy3"]["names"]
    for name, prefix in ((names[0], "A"), (names[1], "GT"), (names[2], "ACGTACGTACGT")):
        body = _body(golden.stream("toy3", name, prefix))
        rc, whole, err = _run(exe, tmp_path, body)
        assert rc == 0, err
        for piece in (1, 7, 23, 24, 25, 64, 1000, 65536):
            rc, out, err = _run(exe, tmp_path, body, piece)
            assert rc == 0 and out == whole, (name, prefix, piece, out, err)
    body = _body(golden.stream("toy3", names[0], "GT"))
    i = body.index(b"R")
    for bad in (body[:-1], body[: len(body) // 2], body[:i + 1] + bytes([body[i + 1] ^ 1]) + body[i + 2:], body.replace(b"(G", b"(X", 1)):
        for piece in (1, 24, 4096):
            rc, out, _ = _run(exe, tmp_path, bad, piece)
            assert rc == 1 and out.startswith("error:"), (piece, out)


def _tokens(body):
    """An independent reading of the wire grammar (ClientSocket.h:12-46): yields ('open', sym) / ('close', freq)."""
    i, depth = 0, 0
    def varint():
        nonlocal i
        c = body[i]; i += 1
        if c >= 0x80:
            return c ^ 0x80
        v = int.from_bytes(body[i:i + c], "little"); i += c
        return v
    while i < len(body):
        if body[i:i + 1] == b"(":
            depth += 1
            yield ("open", chr(body[i + 1]), depth)
            i += 2
        else:
            f = varint()
            if depth <= 6:
                assert body[i:i + 1] == b"R"; i += 1
                varint()
            i += 2
            yield ("close", f, depth)
            depth -= 1


def test_unit_marks_of_the_decoder(golden, exe, tmp_path):
    """dsm_server's units of merging-while-receiving: when a node `extra` + 1 levels below the enforced prefix closes, the decoder notes
    how far every deeper level has grown; the subtree sizes that follow from the marks, the closes of the nodes between the prefix and
    the units and their order equal an independent reading of the token stream."""
    names = golden.manifest["sets"]["toy3"]["names"]
    for name, prefix in ((names[0], "A"), (names[1], "GT"), (names[2], "TTG")):
        body = _body(golden.stream("toy3", name, prefix))
        K = len(prefix)
        for extra in (0, 1, 2):
            U = K + 1 + extra
            want, path, sizes = [], [], {}
            for kind, val, depth in _tokens(body):
                if kind == "open":
                    if depth > K:
                        path = path[:depth - K - 1] + [val]
                    if depth == U:
                        sizes["".join(path[:U - K])] = 0
                    if depth >= U:
                        sizes["".join(path[:U - K])] += 1
                elif K < depth <= U:
                    key = "".join(path[:depth - K])
                    want.append("%s:%d:%d" % (key, sizes[key] if depth == U else 0, val))
            for piece in (1, 24, 1000, 1 << 20):
                rc, out, err = _run_units(exe, tmp_path, body, piece, U, K)
                assert rc == 0, (out, err)
                fields = dict(kv.split("=") for kv in out.split()[1:])
                assert fields["chain"] == prefix and fields["units"] == ",".join(want), (name, prefix, extra, piece)
        # a prefix length that is too long for the stream is refused
        rc, out, _ = _run_units(exe, tmp_path, body, 4096, K + 2, K + 1)
        assert rc == 1 and "enforced path" in out


def _run_units(exe, tmp_path, body, piece, unit_depth, chain_len):
    p = tmp_path / "s.bin"
    p.write_bytes(body)
    r = subprocess.run([exe, str(p), str(piece), str(unit_depth), str(chain_len)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode().strip(), r.stderr.decode()


def test_server_takes_the_subtrees_in_union_post_order_and_never_early(golden, tmp_path_factory, tmp_path):
    """The merger of dsm_server (merging while receiving): which subtree is due next is host logic over the streams' decoders
    (dsm::pick_event).  The three reference streams of a prefix are fed in random pieces and interleavings; every event is taken
    once, in the post-order of the union trie, and none before every stream is past it (checked against a complete first parse)."""
    exe = str(tmp_path_factory.mktemp("native") / "server_sched_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "native", "server_sched_check.cpp")], check=True)
    names = golden.manifest["sets"]["toy3"]["names"]
    for prefix in ("A", "GT", "TTG"):
        files = []
        for n in names:
            f = tmp_path / ("%s.%s.bin" % (n, prefix))
            f.write_bytes(_body(golden.stream("toy3", n, prefix)))
            files.append(str(f))
        K = len(prefix)
        for extra in (0, 1, 2):
            for seed in (1, 2, 3):
                r = subprocess.run([exe, str(K), str(K + 1 + extra), str(seed)] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                out = r.stdout.decode().strip()
                assert r.returncode == 0 and out.startswith("ok "), (prefix, extra, seed, out, r.stderr.decode()[-500:])
                f = dict(kv.split("=") for kv in out.split()[1:])
                assert int(f["units"]) >= 1 and int(f["events"]) >= int(f["units"])
                if prefix == "A":
                    assert int(f["early"]) >= 1, out      # something is merged before the last stream has ended
    # a sample without the prefix (empty stream) and one that holds a single subtree of it
    files = []
    for n, p in ((names[0], "A"), (names[1], "AC")):
        f = tmp_path / ("%s.%s.mix.bin" % (n, p))
        f.write_bytes(_body(golden.stream("toy3", n, p)))
        files.append(str(f))
    e = tmp_path / "empty.bin"
    e.write_bytes(b"")
    r = subprocess.run([exe, "1", "3", "7", files[0], str(e), files[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.decode().startswith("ok "), (r.stdout, r.stderr.decode()[-500:])
