"""Drop-in executables on the GPU box: our client against the UNMODIFIED reference metaserver (oracle/_ref binary,
which travels with the repo), and the fused dsm_node driver against the reference server's golden stdout."""
import os
import socket
import subprocess
import time

import pytest

from goldenlib import wait_listen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dsm-framework_amd", "host")
REF = os.path.join(ROOT, "oracle", "_ref")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_dsm_node_prints_reference_tuples(golden):
    names = golden.manifest["sets"]["toy3"]["names"]
    fmis = [golden.fmi("toy3", n) for n in names]
    out = subprocess.run([os.path.join(HOST, "dsm_node"), "-E", "2.0", "-f", "2", "-p", "A,GT,TTG"] + fmis, check=True,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout
    assert out == b"".join(golden.server_out("toy3", "default", p) for p in ["A", "GT", "TTG"])
    # mandatory -E like the reference server (metaserver.cpp:582-586)
    r = subprocess.run([os.path.join(HOST, "dsm_node"), "-p", "A"] + fmis, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"expecting parameter --emax" in r.stderr


def test_dsm_node_device_list_runs_the_exchange_on_rccl(golden, tmp_path):
    """--devices: one thread + stream per GPU, ncclAllGather from the library's exchange callback (no Python on the path).  One
    card is all a test box has, so the world is one rank -- DSM_FORCE_EXCHANGE sends every level through the collective anyway
    (send buffer, RCCL all-gather, status words, owner-only emission)."""
    names = golden.manifest["sets"]["toy3"]["names"]
    fmis = [golden.fmi("toy3", n) for n in names]
    ps = ["A", "C", "GT", "TTG"]
    want = b"".join(golden.server_out("toy3", "default", p) for p in ps)
    for env in (dict(os.environ, DSM_FORCE_EXCHANGE="1"), dict(os.environ)):
        r = subprocess.run([os.path.join(HOST, "dsm_node"), "--devices", "0", "-E", "2.0", "-f", "2", "-p", ",".join(ps)] + fmis,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout == want
    r = subprocess.run([os.path.join(HOST, "dsm_node"), "--devices", "0", "-E", "2.0", "-f", "2", "--out-prefix", str(tmp_path / "o."), "-p", "A,GT"] + fmis,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    for p in ("A", "GT"):
        assert open(tmp_path / ("o." + p + ".txt"), "rb").read() == golden.server_out("toy3", "default", p)
    # owner mode (--exchange owner: prefix k merged by device k mod G alone; lanes, gate, ncclSend/Recv + ncclBroadcast callbacks),
    # with the world of one rank a test box allows: plain, and with every level forced through the gather / broadcast
    for env in (dict(os.environ, DSM_FORCE_EXCHANGE="1"), dict(os.environ)):
        r = subprocess.run([os.path.join(HOST, "dsm_node"), "--devices", "0", "--exchange", "owner", "-E", "2.0", "-f", "2", "-p", ",".join(ps)] + fmis,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout == want and b"owner mode" in r.stderr
    # samples must divide evenly among the devices
    r = subprocess.run([os.path.join(HOST, "dsm_node"), "--devices", "0,0", "-E", "2.0", "-p", "A"] + fmis, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"multiple of the number of devices" in r.stderr


def test_dsm_node_device_list_validates_every_input_first(golden, tmp_path):
    """--devices, both exchanges: every sample is validated (dsm_index_probe) and every device ordinal checked BEFORE a communicator,
    stream or rank thread exists -- a bad sample among good ones is named and the process leaves with 1 instead of leaving ranks in a
    collective (the reference's client stops in main() before it connects, metaenumerate.cpp:226-247)."""
    names = golden.manifest["sets"]["toy3"]["names"]
    fmis = [golden.fmi("toy3", n) for n in names]
    raw = open(fmis[1], "rb").read()
    cut = tmp_path / "cut.fmi"
    cut.write_bytes(raw[:len(raw) // 2])
    ver = tmp_path / "version.fmi"
    ver.write_bytes(b"\x0d" + raw[1:])
    for ex in ("allgather", "owner"):
        base = [os.path.join(HOST, "dsm_node"), "--devices", "0", "--exchange", ex, "-E", "2.0", "-f", "2", "-p", "A,C"]
        r = subprocess.run(base + [fmis[0], str(cut), str(ver)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and r.stdout == b""
        assert b"cut.fmi: truncated or corrupt .fmi" in r.stderr and b"version.fmi: FMIndex: invalid save file version" in r.stderr
        assert b"device 0:" not in r.stderr     # (reported by main(), not by a rank thread)
        r = subprocess.run(base + [fmis[0], str(tmp_path / "missing.fmi"), fmis[2]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and b"missing.fmi: file not found" in r.stderr
        r = subprocess.run([os.path.join(HOST, "dsm_node"), "--devices", "0,97,98", "--exchange", ex, "-E", "2.0", "-p", "A"] + fmis,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1 and b"no device 97" in r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "metaenumerate_patched")), reason="oracle/_ref/metaenumerate_patched not present")
def test_patched_reference_client_sends_the_reference_streams(golden):
    """INTEGRATION.md section 2, compiled: the reference metaenumerate with its EnumerateQuery::enumerate call replaced by
    dsm_enumerate (oracle/integration_s2.patch, built by oracle/Makefile.ref) sends, prefix by prefix, exactly the bytes the
    unmodified client sends (the committed golden streams)."""
    import threading
    names = golden.manifest["sets"]["toy3"]["names"]
    prefixes = ["A", "C", "G", "T"]
    for name in names[:2]:
        got = {}
        socks = []
        hosts = ""
        for p in prefixes:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            s.listen(1)
            socks.append((p, s))
            hosts += "127.0.0.1 %d %s\n" % (s.getsockname()[1], p)

        def serve(p, s):
            c, _ = s.accept()
            buf = b""
            while True:
                b = c.recv(1 << 16)
                if not b:
                    break
                buf += b
            c.close()
            s.close()
            got[p] = buf

        ths = [threading.Thread(target=serve, args=ps) for ps in socks]
        for t in ths:
            t.start()
        r = subprocess.run([os.path.join(REF, "metaenumerate_patched"), "--fmin", "2", golden.fmi("toy3", name)], input=hosts.encode(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        for t in ths:
            t.join(timeout=60)
        assert r.returncode == 0, r.stderr
        for p in prefixes:
            assert got[p] == golden.stream("toy3", name, p), (name, p)


def test_check_mode(golden):
    r = subprocess.run([os.path.join(HOST, "metaenumerate_hip"), "--check", golden.fmi("toy3", "toy-1")], input=b"localhost 5000 A\n",
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"OK" in r.stderr and b"n = 102000, total = 102000" in r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "metaserver")), reason="oracle/_ref/metaserver not present")
def test_our_client_feeds_the_unmodified_reference_server(golden, tmp_path):
    names = golden.manifest["sets"]["toy3"]["names"]
    prefixes = ["C", "AC", "T"]
    procs, hosts = [], ""
    for p in prefixes:
        port = _free_port()
        out = open(tmp_path / ("out." + p), "wb")
        pr = subprocess.Popen([os.path.join(REF, "metaserver"), "-p", str(port), "-E", "2.0"], stdin=subprocess.PIPE, stdout=out,
                              stderr=subprocess.DEVNULL)
        pr.stdin.write(("\n".join(names) + "\n").encode())
        pr.stdin.close()
        procs.append((p, pr, out))
        hosts += "127.0.0.1 %d %s\n" % (port, p)
    ports = [int(l.split()[1]) for l in hosts.splitlines()]
    for (_, pr, _), pt in zip(procs, ports):
        assert wait_listen(pt, pr), "metaserver did not start listening"
    clients = []
    for n in names:
        c = subprocess.Popen([os.path.join(HOST, "metaenumerate_hip"), "--fmin", "2", golden.fmi("toy3", n)], stdin=subprocess.PIPE,
                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        c.stdin.write(hosts.encode())
        c.stdin.close()
        clients.append(c)
    for c in clients:
        assert c.wait(timeout=300) == 0, c.stderr.read()
    for p, pr, out in procs:
        assert pr.wait(timeout=120) == 0
        out.close()
        assert open(tmp_path / ("out." + p), "rb").read() == golden.server_out("toy3", "default", p), p
