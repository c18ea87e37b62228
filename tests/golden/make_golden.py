#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the unmodified reference.

Run in the build container only (needs oracle/_ref/{builder,metaserver,metaenumerate},
built from /root/reference by `make -f oracle/Makefile.ref`).  The outputs committed next
to this script are data only: seeded synthetic FASTA inputs, the reference-built .fmi files,
raw client byte streams captured with a TCP sink, and metaserver stdout.

    python tests/golden/make_golden.py            # regenerates everything

Fixture sets (see tests/golden/MANIFEST.json, written by this script):
  toy3   3 samples x 1000 reads x 50 bp, 6.6 kbp genome (order-6 de Bruijn + random); fmin 2; prefixes A C G T and AC GT TTG
  toyN   1 sample with N / lower-case / IUPAC symbols (normalisation + 7-symbol alphabet)
  deep1  toy3 sample 1, --fmin 1 -M 40 (followOneBranch path), prefixes A C G T
  many30 30 tiny samples, --fmin 3 -M 14 (more than 13 readers: std::unordered_set rehashes, ids share buckets)
  five   5 samples x 1600..2200 reads x 60 bp, 4.7 kbp genome, default --fmin 10 (cfg-1 plumbing, scaled down)
"""
import gzip
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def synth_reads(rng, genome, nreads, rlen, sub_rate, private=None, private_frac=0.0):
    reads = []
    for _ in range(nreads):
        src = genome
        if private is not None and rng.random() < private_frac:
            src = private
        p = int(rng.integers(0, len(src) - rlen + 1))
        r = src[p:p + rlen].copy()
        if rng.random() < 0.5:  # sample either strand
            r = (3 - r)[::-1]
        m = rng.random(rlen) < sub_rate
        r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) % 4
        reads.append(BASES[r].tobytes().decode())
    return reads


def de_bruijn(k):
    """Order-k de Bruijn sequence over 4 symbols, linearised (every k-mer occurs): keeps every
    depth<=6 trie node present in every sample, which the reference server needs when pmin>1
    (traverseOne does not consume the R token, metaserver.cpp:211-226)."""
    a = [0] * (4 * k)
    seq = []

    def db(t, p):
        if t > k:
            if k % p == 0:
                seq.extend(a[1:p + 1])
        else:
            a[t] = a[t - p]
            db(t + 1, p)
            for j in range(a[t - p] + 1, 4):
                a[t] = j
                db(t + 1, t)
    db(1, 1)
    seq = seq + seq[:k - 1]
    return np.array(seq, dtype=np.uint8)


def write_fasta(path, reads, width=0):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(">r%d\n" % i)
            if width:
                for k in range(0, len(r), width):
                    f.write(r[k:k + width] + "\n")
            else:
                f.write(r + "\n")


def run_builder(fasta):
    subprocess.run([os.path.join(REF, "builder"), fasta], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return fasta + ".fmi"


class Sink(threading.Thread):
    """Accept one connection and keep every byte it sends (the raw client stream)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.s = socket.socket()
        self.s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.s.bind(("127.0.0.1", 0))
        self.s.listen(1)
        self.port = self.s.getsockname()[1]
        self.data = b""

    def run(self):
        c, _ = self.s.accept()
        chunks = []
        while True:
            b = c.recv(1 << 16)
            if not b:
                break
            chunks.append(b)
        self.data = b"".join(chunks)
        c.close()
        self.s.close()


def capture_streams(fmi, prefixes, args):
    """One reference client, one sink per prefix -> {prefix: bytes}."""
    sinks = [Sink() for _ in prefixes]
    for s in sinks:
        s.start()
    hosts = "".join("127.0.0.1 %d %s\n" % (s.port, p) for s, p in zip(sinks, prefixes))
    subprocess.run([os.path.join(REF, "metaenumerate")] + args + [fmi], input=hosts.encode(),
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for s in sinks:
        s.join()
    return {p: s.data for s, p in zip(sinks, prefixes)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_servers(names, fmis, prefixes, client_args, server_args):
    """Full reference pipeline on loopback -> {prefix: server stdout bytes}."""
    tmp = tempfile.mkdtemp()
    procs, outs, ports = [], {}, {}
    for p in prefixes:
        ports[p] = free_port()
        out = open(os.path.join(tmp, "out.%s" % p), "wb")
        pr = subprocess.Popen([os.path.join(REF, "metaserver"), "-p", str(ports[p])] + server_args,
                              stdin=subprocess.PIPE, stdout=out, stderr=subprocess.DEVNULL)
        pr.stdin.write(("\n".join(names) + "\n").encode())
        pr.stdin.close()
        procs.append((p, pr, out))
    import time
    time.sleep(0.5)
    hosts = "".join("127.0.0.1 %d %s\n" % (ports[p], p) for p in prefixes)
    clients = []
    for fmi in fmis:
        c = subprocess.Popen([os.path.join(REF, "metaenumerate")] + client_args + [fmi],
                             stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        c.stdin.write(hosts.encode())
        c.stdin.close()
        clients.append(c)
    for c in clients:
        assert c.wait() == 0
    for p, pr, out in procs:
        rc = pr.wait(timeout=120)
        out.close()
        assert rc == 0, ("metaserver failed", p, rc)
        outs[p] = open(os.path.join(tmp, "out.%s" % p), "rb").read()
    shutil.rmtree(tmp)
    return outs


def put(path, data):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    if path.endswith(".gz"):
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def main():
    assert os.path.exists(os.path.join(REF, "builder")), "build oracle/_ref first"
    manifest = {"sets": {}}
    work = tempfile.mkdtemp()

    # ---------------- toy3 ----------------
    rng = np.random.default_rng(20261003)
    genome = np.concatenate([de_bruijn(6), rng.integers(0, 4, 2500).astype(np.uint8)])
    names, fmis = [], []
    for s in range(3):
        private = rng.integers(0, 4, 1500).astype(np.uint8)
        reads = synth_reads(rng, genome, 1000, 50, 0.01, private, 0.05)
        name = "toy-%d" % (s + 1)
        fa = os.path.join(work, name + ".fasta")
        write_fasta(fa, reads)
        fmi = run_builder(fa)
        names.append(name)
        fmis.append(fmi)
        put(os.path.join(HERE, "toy3", name + ".fasta.gz"), open(fa, "rb").read())
        put(os.path.join(HERE, "toy3", name + ".fasta.fmi.gz"), open(fmi, "rb").read())
    prefixes = ["A", "C", "G", "T"]
    deep_prefixes = ["AC", "GT", "TTG", "ACGTACGTACGT"]
    streams = {}
    for name, fmi in zip(names, fmis):
        st = capture_streams(fmi, prefixes + deep_prefixes, ["--fmin", "2"])
        for p, d in st.items():
            put(os.path.join(HERE, "toy3", "stream.%s.%s.fmin2.bin.gz" % (name, p)), d)
            streams[(name, p)] = len(d)
    server_cfgs = {
        "default": ["-E", "2.0"],
        "pmax2": ["-E", "2.0", "-P", "2", "--pmax", "2"],
        "p1": ["-E", "2.0", "-P", "1", "--pmax", "1"],
        "emin_m": ["-E", "1.4", "-e", "0.5", "-m", "8"],
        "noent": ["-E", "0", "-P", "1"],
    }
    for cfg, sargs in server_cfgs.items():
        # -P 1 variants are safe; pmin>1 needs every depth<=6 node in >=2 samples
        outs = run_servers(names, fmis, prefixes + (["AC", "GT"] if cfg != "default" else ["AC", "GT", "TTG"]),
                           ["--fmin", "2"], sargs)
        for p, d in outs.items():
            put(os.path.join(HERE, "toy3", "server.%s.%s.txt.gz" % (cfg, p)), d)
    manifest["sets"]["toy3"] = {"names": names, "fmin": 2, "prefixes": prefixes + deep_prefixes,
                                "server_cfgs": server_cfgs,
                                "stream_bytes": {"%s/%s" % k: v for k, v in streams.items()}}

    # ---------------- deep1: fmin 1, maxdepth 40 (followOneBranch) ----------------
    st = capture_streams(fmis[0], prefixes, ["--fmin", "1", "-M", "40"])
    for p, d in st.items():
        put(os.path.join(HERE, "toy3", "stream.%s.%s.fmin1.M40.bin.gz" % (names[0], p)), d)
    outs = run_servers(names, fmis, prefixes, ["--fmin", "1", "-M", "24"], ["-E", "2.0", "-P", "1", "--pmax", "1"])
    for p, d in outs.items():
        put(os.path.join(HERE, "toy3", "server.p1_fmin1_M24.%s.txt.gz" % p), d)
    manifest["sets"]["deep1"] = {"sample": names[0], "fmin": 1, "maxdepth": 40}

    # ---------------- toyN: N, lower case, IUPAC, multi-line FASTA ----------------
    reads = synth_reads(rng, genome, 300, 50, 0.01)
    out = []
    for i, r in enumerate(reads):
        r = list(r)
        if i % 7 == 0:
            r[int(rng.integers(0, 50))] = "N"
        if i % 11 == 0:
            r[int(rng.integers(0, 50))] = "R"   # becomes N (builder.cpp:60-104)
        if i % 5 == 0:
            r = [c.lower() for c in r]
        out.append("".join(r))
    fa = os.path.join(work, "toyN.fasta")
    write_fasta(fa, out, width=30)
    fmiN = run_builder(fa)
    put(os.path.join(HERE, "toyN", "toyN.fasta.gz"), open(fa, "rb").read())
    put(os.path.join(HERE, "toyN", "toyN.fasta.fmi.gz"), open(fmiN, "rb").read())
    st = capture_streams(fmiN, prefixes, ["--fmin", "2"])
    for p, d in st.items():
        put(os.path.join(HERE, "toyN", "stream.toyN.%s.fmin2.bin.gz" % p), d)
    manifest["sets"]["toyN"] = {"names": ["toyN"], "fmin": 2, "prefixes": prefixes}

    # ---------------- five: default fmin 10 ----------------
    rng = np.random.default_rng(5)
    genome5 = np.concatenate([de_bruijn(6), rng.integers(0, 4, 600).astype(np.uint8)])
    names5, fmis5 = [], []
    for s in range(5):
        reads = synth_reads(rng, genome5, 1600 + 150 * s, 60, 0.01)
        name = "five-%d" % (s + 1)
        fa = os.path.join(work, name + ".fasta")
        write_fasta(fa, reads)
        fmi = run_builder(fa)
        names5.append(name)
        fmis5.append(fmi)
        put(os.path.join(HERE, "five", name + ".fasta.gz"), open(fa, "rb").read())
        put(os.path.join(HERE, "five", name + ".fasta.fmi.gz"), open(fmi, "rb").read())
    outs = run_servers(names5, fmis5, prefixes, [], ["-E", "2.0"])
    for p, d in outs.items():
        put(os.path.join(HERE, "five", "server.default.%s.txt.gz" % p), d)
    manifest["sets"]["five"] = {"names": names5, "fmin": 10, "prefixes": prefixes,
                                "server_cfgs": {"default": ["-E", "2.0"]}}

    # ---------------- many30: 30 samples (reader sets rehash 13 -> 29 -> 59 buckets) ----------------
    rng = np.random.default_rng(30)
    genome30 = np.concatenate([de_bruijn(6), rng.integers(0, 4, 400).astype(np.uint8)])
    names30, fmis30 = [], []
    for s in range(30):
        reads = synth_reads(rng, genome30, 700 + 10 * s, 40, 0.01)
        name = "m%02d" % s
        fa = os.path.join(work, name + ".fasta")
        write_fasta(fa, reads)
        fmi = run_builder(fa)
        names30.append(name)
        fmis30.append(fmi)
        put(os.path.join(HERE, "many30", name + ".fasta.fmi.gz"), open(fmi, "rb").read())
    cfgs30 = {"default": ["-E", "5.0"], "p3": ["-E", "4.9", "-e", "1.0", "-P", "3", "--pmax", "20", "-m", "7"]}
    for cfg, sargs in cfgs30.items():
        outs = run_servers(names30, fmis30, ["AC", "G"], ["--fmin", "3", "-M", "14"], sargs)
        for p, d in outs.items():
            put(os.path.join(HERE, "many30", "server.%s.%s.txt.gz" % (cfg, p)), d)
    manifest["sets"]["many30"] = {"names": names30, "fmin": 3, "maxdepth": 14, "prefixes": ["AC", "G"], "server_cfgs": cfgs30}

    manifest["glibc"] = os.confstr("CS_GNU_LIBC_VERSION")
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(work)
    total = 0
    for d, _, fs in os.walk(HERE):
        for fn in fs:
            total += os.path.getsize(os.path.join(d, fn))
    print("golden fixtures written, %.1f KB total" % (total / 1024))


if __name__ == "__main__":
    sys.exit(main())
