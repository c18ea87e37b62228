"""Access to the committed golden fixtures (tests/golden/, produced by make_golden.py)."""
import gzip
import json
import os
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def server_args_to_kw(args):
    """['-E','2.0','-P','1','--pmax','1','-e','0.5','-m','8'] -> kwargs of the server restatements."""
    kw = dict(pmin=2, pmax=0, mindepth=0, emin=0.0, emax=-1.0)
    it = iter(args)
    for a in it:
        v = next(it)
        if a == "-E":
            kw["emax"] = float(v)
        elif a == "-e":
            kw["emin"] = float(v)
        elif a == "-P":
            kw["pmin"] = int(v)
        elif a == "--pmax":
            kw["pmax"] = int(v)
        elif a == "-m":
            kw["mindepth"] = int(v)
        else:
            raise ValueError(a)
    return kw


class Golden:
    def __init__(self):
        self.manifest = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
        self.tmp = tempfile.mkdtemp(prefix="dsmgold")
        self._fmi = {}

    def read(self, rel):
        with gzip.open(os.path.join(GOLD, rel), "rb") as f:
            return f.read()

    def fmi(self, setname, name):
        """Path of the (decompressed) reference-built index <name>.fasta.fmi."""
        key = (setname, name)
        if key not in self._fmi:
            d = os.path.join(self.tmp, setname)
            os.makedirs(d, exist_ok=True)
            p = os.path.join(d, name + ".fasta.fmi")
            with open(p, "wb") as f:
                f.write(self.read("%s/%s.fasta.fmi.gz" % (setname, name)))
            self._fmi[key] = p
        return self._fmi[key]

    def fasta(self, setname, name):
        return self.read("%s/%s.fasta.gz" % (setname, name)).decode()

    def stream(self, setname, name, prefix, tag="fmin2"):
        return self.read("%s/stream.%s.%s.%s.bin.gz" % (setname, name, prefix, tag))

    def server_out(self, setname, cfg, prefix):
        return self.read("%s/server.%s.%s.txt.gz" % (setname, cfg, prefix))


def wait_listen(port, proc=None, timeout=30.0):
    """Block until some process listens on TCP `port` (from /proc/net/tcp: no probe connection that a server would take
    for a client).  Returns False when `proc` exits first or the time is up."""
    import time
    want = "%04X" % port
    t0 = time.time()
    while time.time() - t0 < timeout:
        for f in ("/proc/net/tcp", "/proc/net/tcp6"):
            try:
                for line in open(f).read().splitlines()[1:]:
                    col = line.split()
                    if col[1].rsplit(":", 1)[1] == want and col[3] == "0A":
                        return True
            except OSError:
                pass
        if proc is not None and proc.poll() is not None:
            return False
        time.sleep(0.02)
    return False
