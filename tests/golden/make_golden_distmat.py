#!/usr/bin/env python3
"""Distance-matrix goldens: the UNMODIFIED reference smtxt2entropy (oracle/_ref/smtxt2entropy, built by
oracle/Makefile.ref from wrapper-distance-matrix/smtxt2entropy.c) run on the committed reference-server outputs.
Writes tests/golden/<set>/distmat.<case>.{count,log,sqrt,lgamma}.gz and the case table into MANIFEST.json.
Run in the build container only (needs /root/reference through oracle/_ref)."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
TOOL = os.path.join(ROOT, "oracle", "_ref", "smtxt2entropy")

CASES = {
    # set: {case: (server cfg, prefixes, tool arguments without -s / -F)}
    "toy3": {"m4": ("default", ["A", "C", "G", "T"], ["-m", "0.25,0.5,0.75,1.0"]),
             "step": ("default", ["A", "C"], ["-e", "0.3"])},
    "five": {"m3": ("default", ["A", "C", "G", "T"], ["-m", "0.4,0.8,1.0"]),
             "minfreq": ("default", ["A", "C", "G", "T"], ["-m", "0.5,1.0", "-M", "12"])},
    "many30": {"m2": ("default", ["AC", "G"], ["-m", "0.7,1.0"])},
}
# cases with the -S run-to-sample file and / or the -N dataset-size file: {set: {case: (cfg, prefixes, args, mapping, sizes)}}
FILE_CASES = {
    "five": {"smap": ("default", ["A", "C", "G", "T"], ["-m", "0.5,1.0"], [0, 1, 1, 2, 0], None),
             "norm": ("default", ["A", "C", "G", "T"], ["-m", "0.4,0.8,1.0"], None, [1.0, 2.5, 0.5, 4.0, 1.5]),
             "smap_norm": ("default", ["A", "C"], ["-e", "0.5"], [2, 0, 1, 1, 0], [3.0, 1.0, 2.0])},
    "many30": {"smap": ("default", ["AC", "G"], ["-m", "0.6,1.0", "-M", "4"], [i % 7 for i in range(30)], None)},
}


def main():
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], check=True)
    man_path = os.path.join(HERE, "MANIFEST.json")
    man = json.load(open(man_path))
    out_cases = {}
    for setname, cases in CASES.items():
        names = man["sets"][setname]["names"]
        for case, (cfg, prefixes, targs) in cases.items():
            if cfg not in man["sets"][setname]["server_cfgs"]:
                continue
            prefixes = prefixes or man["sets"][setname]["prefixes"]
            text = b""
            for p in prefixes:
                f = os.path.join(HERE, setname, "server.%s.%s.txt.gz" % (cfg, p))
                if os.path.exists(f):
                    text += gzip.open(f, "rb").read()
            if not text:
                continue
            with tempfile.TemporaryDirectory() as td:
                subprocess.run([TOOL, "-s", str(len(names)), "-F", "out"] + targs, input=text, cwd=td, check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                for kind in ("count", "log", "sqrt", "lgamma"):
                    data = open(os.path.join(td, "%s.out" % kind), "rb").read()
                    with gzip.GzipFile(os.path.join(HERE, setname, "distmat.%s.%s.gz" % (case, kind)), "wb", mtime=0) as g:
                        g.write(data)
            out_cases.setdefault(setname, {})[case] = {"server_cfg": cfg, "prefixes": prefixes, "args": targs, "lines": text.count(b"\n")}
            print(setname, case, text.count(b"\n"), "lines")
    for setname, cases in FILE_CASES.items():
        for case, (cfg, prefixes, targs, mapping, sizes) in cases.items():
            text = b"".join(gzip.open(os.path.join(HERE, setname, "server.%s.%s.txt.gz" % (cfg, p)), "rb").read() for p in prefixes)
            with tempfile.TemporaryDirectory() as td:
                args = [TOOL, "-F", "out"] + targs
                if mapping is not None:
                    open(os.path.join(td, "map.txt"), "w").write("".join("%d\n" % x for x in mapping))
                    args += ["-S", "map.txt"]
                    smpls = max(mapping) + 1
                else:
                    smpls = len(man["sets"][setname]["names"])
                    args += ["-s", str(smpls)]
                if sizes is not None:
                    open(os.path.join(td, "sizes.txt"), "w").write("".join("d%d\t%r\n" % (i, x) for i, x in enumerate(sizes)))
                    args += ["-N", "sizes.txt"]
                subprocess.run(args, input=text, cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                for kind in ("count", "log", "sqrt", "lgamma"):
                    data = open(os.path.join(td, "%s.out" % kind), "rb").read()
                    with gzip.GzipFile(os.path.join(HERE, setname, "distmat.%s.%s.gz" % (case, kind)), "wb", mtime=0) as g:
                        g.write(data)
            out_cases.setdefault(setname, {})[case] = {"server_cfg": cfg, "prefixes": prefixes, "args": targs, "lines": text.count(b"\n"),
                                                        "mapping": mapping, "sizes": sizes}
            print(setname, case, text.count(b"\n"), "lines")
    man["distmat"] = out_cases
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
