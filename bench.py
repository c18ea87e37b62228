#!/usr/bin/env python3
"""bench.py -- enumerated substrings/s of the MI355X substring-enumeration hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] at N=1, configs[2] at N>1): one synthetic read set per GPU,
10^7 reads x 100 bp from a shared 50 Mbp random genome (20x, 0.5 % substitutions; seeds 42+rank), indexed as
the reference builder would (n = 2.02e9 BWT symbols, .fmi v17 written by pydsm.builder and loaded through
the C ABI), enumerated with --fmin 10, merged and filtered with -E 2.0 (-P 1 for a single sample, the reference
default -P 2 otherwise).  A step = one pass of enumerate+merge+filter over every k-mer prefix for every sample.
Index construction and loading are outside the timed region; inputs are HBM resident when it starts.

One JSON line on rank 0; `value` = emitted trie nodes ('(' tokens, EnumerateQuery.cpp:209) summed over all
samples and steps / max-over-ranks wall time.
"""
import argparse
import ctypes
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALG_BYTES_PER_RANK = 17    # 8 B superblock + 1 B block + 8 B word per BitRank::rank


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--rlen", type=int, default=100)
    ap.add_argument("--genome", type=int, default=50_000_000)
    ap.add_argument("--sub-rate", type=float, default=0.005)
    ap.add_argument("--prefix-len", type=int, default=-1, help="k-mer prefix length of the chunks; default 1")
    ap.add_argument("--fmin", type=int, default=10)
    ap.add_argument("--emax", type=float, default=2.0)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline leg (rank 0, N=1)")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal at N=1: drive the whole multi-rank exchange path (send buffer, "
                    "one RCCL all-gather per level through torch.distributed, lanes, status words) with a world of one")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra records of the default N=1 run (64-bit positions; eight samples on one GPU)")
    ap.add_argument("--extras", default=os.environ.get("DSM_BENCH_EXTRAS", "all"),
                    help="which extra records: all, or a comma list of u64,stream,server,d8,d64,big (big = configs[3]: 4-Gbase samples)")
    ap.add_argument("--pmin", type=int, default=-1, help="metaserver -P; default 1 for a single sample, else 2")
    ap.add_argument("--pmax", type=int, default=0, help="metaserver --pmax (BASELINE configs[3]: 8, configs[4]: 1)")
    ap.add_argument("--wide", action="store_true", help="force 64-bit positions (the code path of n > 2^32, BASELINE configs[3])")
    ap.add_argument("--stream-mode", action="store_true", help="time the wire-stream path (dsm_enumerate, the TCP drop-in client) instead of mining")
    ap.add_argument("--lanes", type=int, default=0, help="independent prefix lanes (own stream, own communicator) that overlap one lane's "
                    "all-gather with the other's kernels; default 1 (N=1) / 2 (N>1)")
    ap.add_argument("--nlocal", type=int, default=1, help="samples per GPU (BASELINE configs[4]: 8 per GPU); default 1")
    ap.add_argument("--exchange", choices=["allgather", "owner", "both"], default=os.environ.get("DSM_BENCH_MODE", "allgather"),
                    help="N > 1: 'allgather' = every rank merges every prefix from one all-gather per level; 'owner' = the reference's "
                         "partition, prefix k is merged by rank k %% N alone (columns gathered to it, the union's child masks broadcast "
                         "back), one lane per owner so that every rank is the server of one lane and a client in the others; 'both' = one "
                         "after the other on the same indexes, both values in the line, the faster one as the headline")
    ap.add_argument("--workdir", default=os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench"))
    return ap.parse_args()


def build_index(args, rank, dev):
    import torch
    from pydsm import builder
    os.makedirs(args.workdir, exist_ok=True)
    private = 0.05 if args.gpus * args.nlocal > 1 else 0.0  # SURVEY 8d cfg 3: sample-specific 5 % when there are several samples
    tag = "s%d_r%d_l%d_g%d_e%g%s" % (42 + rank, args.reads, args.rlen, args.genome, args.sub_rate, "_p%g" % private if private else "")
    path = os.path.join(args.workdir, "sample-%d.%s.fmi" % (rank, tag))
    t0 = time.time()
    if not os.path.exists(path):
        codes = builder.synth_reads(42 + rank, args.reads, args.rlen, args.genome, args.sub_rate, device=dev,
                                    private_frac=private)
        builder.build_from_codes(codes, path + ".tmp")
        del codes
        torch.cuda.empty_cache()
        os.replace(path + ".tmp", path)
    return path, time.time() - t0


def lf_kernel_sha():
    """first 16 hex digits of the sha256 of the LF-step kernel's sources: ties profiles/traffic.json to the build it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("expand.hip", "lfstep.h", "common.h"):
        try:
            h.update(open(os.path.join(ROOT, "dsm-framework_amd", "csrc", f), "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


def host_cores():
    """CPU share of this job: affinity mask, capped by the cgroup quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(path, args, pmin, seconds=None):
    """The oracle (our restatement of the reference client+server, same 3-array layout, one OpenMP thread per
    prefix like metaenumerate.cpp:268, one in-process server per prefix) timed on a bounded sample of the same
    workload: 6-mer prefixes of the same index, in random order, until the time budget is used."""
    import random
    import orc
    ix = orc.Index(path)
    threads = min(orc.lib().orc_max_threads(), host_cores())
    allp = ["".join(p) for p in itertools.product("ACGT", repeat=6)]
    random.Random(1).shuffle(allp)
    done, nodes, t0 = 0, 0, time.time()
    budget = args.cpu_seconds if seconds is None else seconds
    while done < len(allp) and time.time() - t0 < budget:
        batch = allp[done:done + threads]
        _, st = orc.mine([ix], ["sample-0"], batch, fmin=args.fmin, pmin=pmin, emax=args.emax, threads=threads, discard=True)
        nodes += st[0]
        done += len(batch)
    dt = time.time() - t0
    ix.close()
    return {"value": nodes / dt, "unit": "substrings/s", "cores": threads, "kind": "port",
            "sample": "%d random 6-mer prefixes of 4096 of the same index (client enumeration + in-process merge/filter, "
                      "one OpenMP thread per prefix), %d nodes in %.1f s" % (done, nodes, dt)}


def cpu_baseline_reference(path, args, ix, pydsm):
    """The UNMODIFIED reference client (oracle/_ref/metaenumerate, built from /root/reference by oracle/Makefile.ref; it travels to
    the GPU box as a binary) on a bounded sample of the same workload: one thread and one TCP connection per 6-mer prefix
    (metaenumerate.cpp:268-309), the connections drained by local sinks.  Timed from the first connection to the last close, i.e.
    without the index load; the node count is taken from the bytes the reference sent (parsed after the timed region)."""
    import random
    import socket
    import subprocess
    import threading
    exe = os.path.join(ROOT, "oracle", "_ref", "metaenumerate")
    if not os.path.exists(exe):
        return None
    cores = host_cores()
    nprefix = max(cores, int(cores * args.cpu_seconds / 1.3))  # about 1.3 core-seconds per 6-mer prefix of this index
    allp = ["".join(p) for p in itertools.product("ACGT", repeat=6)]
    random.Random(1).shuffle(allp)
    prefixes = allp[:min(len(allp), nprefix)]
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", 0))
    srv.listen(len(prefixes) + 8)
    port = srv.getsockname()[1]
    tm = {"first": None, "last": None, "bytes": 0}
    lock = threading.Lock()
    streams = []  # what the reference client sent, connection by connection (parsed after the clock has stopped)

    def drain(c):
        parts = []
        while True:
            b = c.recv(1 << 20)
            if not b:
                break
            parts.append(b)
        c.close()
        with lock:
            tm["last"] = time.time()
            tm["bytes"] += sum(len(b) for b in parts)
            streams.append(parts)

    def acceptor():
        ths = []
        for _ in prefixes:
            c, _a = srv.accept()
            with lock:
                if tm["first"] is None:
                    tm["first"] = time.time()
            t = threading.Thread(target=drain, args=(c,), daemon=True)
            t.start()
            ths.append(t)
        for t in ths:
            t.join()

    acc = threading.Thread(target=acceptor, daemon=True)
    acc.start()
    hostinfo = "".join("127.0.0.1 %d %s\n" % (port, p) for p in prefixes)
    r = subprocess.run([exe, "--fmin", str(args.fmin), path], input=hostinfo.encode(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=600)
    acc.join(timeout=60)
    srv.close()
    if r.returncode != 0 or tm["first"] is None:
        return None
    dt = tm["last"] - tm["first"]
    # the node count comes from the reference's OWN bytes: every connection's stream goes through the library's stream parser
    # (TrieReader's token rules, 'R' checksums verified), which counts the '(' tokens (EnumerateQuery.cpp:207-209)
    nodes = 0
    for parts in streams:
        t = pydsm.Trie(b"".join(parts), device=ix.device if hasattr(ix, "device") else 0)
        nodes += t.nodes
        t.close()
    return {"value": nodes / dt, "unit": "substrings/s", "cores": cores, "kind": "reference", "host_cpus": os.cpu_count(),
            "sample": "the unmodified reference metaenumerate (oracle/_ref) on %d random 6-mer prefixes of 4096 of the same index, one thread "
                      "and one TCP connection per prefix into local sinks, %d nodes (%.0f MB of stream) in %.1f s from first connection to "
                      "last close; %d of the box's %d CPUs are this job's share" % (len(prefixes), nodes, tm["bytes"] / 1e6, dt, cores, os.cpu_count())}


def extra_records(args, dev, local, pydsm, ix, path, prefixes, pmin):
    """Two more driver-visible measurements of the default N=1 run (each one pass after one warm-up pass, same prefixes):
    the 64-bit position path on the same index (BASELINE configs[3]'s code path) and eight full-size samples resident on this
    one GPU with d = 8 (the one-GPU share of configs[2] / [4])."""
    import torch
    out = []

    def one(ixs, label, **kw):
        with pydsm.Miner(ixs, fmin=args.fmin, emax=args.emax, stream=torch.cuda.current_stream().cuda_stream, **kw) as m:
            m.mine_many(prefixes, text=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = m.mine_many(prefixes, text=False)[1]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        return {"record": label, "value": st.reported / dt, "unit": "substrings/s", "ms_per_step": dt * 1e3, "nodes": st.reported,
                "tuples": st.tuples, "union_nodes": st.union_nodes, "expand_ms": st.expand_ms, "device_ms": st.device_ms,
                "splits": st.splits}

    def note(msg):
        print("bench: extra records: " + msg, file=sys.stderr, flush=True)

    class Skip(Exception):
        pass

    def want(key):
        if args.extras != "all" and key not in args.extras.split(","):
            raise Skip()

    try:
        want("u64")
        out.append(dict(one([ix], "same workload with 64-bit positions (wide=1)", pmin=pmin, wide=1), dtype="u64"))
        note("64-bit positions done")
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "64-bit positions", "error": repr(e)})
    try:  # the literal metaenumerate replacement: the wire stream of every prefix, through PCIe into a sink (EnumerateQuery.cpp:207-222)
        want("stream")
        with pydsm.Miner([ix], fmin=args.fmin, stream=torch.cuda.current_stream().cuda_stream, stream_mode=True) as m:
            m.enumerate_many(prefixes, discard=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nbs, st = m.enumerate_many(prefixes, discard=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out.append({"record": "same workload as wire streams (stream mode: the client's bytes to a host sink)", "value": st.reported / dt,
                    "unit": "substrings/s", "ms_per_step": dt * 1e3, "nodes": st.reported, "wire_bytes": int(sum(nbs)),
                    "expand_ms": st.expand_ms, "device_ms": st.device_ms, "splits": st.splits, "dtype": "u32"})
        note("wire-stream mode done")
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "wire-stream mode", "error": repr(e)})
    try:  # f2, the server side (metaserver.cpp:682-739): four clients' streams of prefix A, fed by four threads, merged WHILE they arrive
        want("server")
        out.append(server_record(args, dev, local, pydsm))
        note("server side done")
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "server side (merge while receiving)", "error": repr(e)})
    try:
        want("d8")
        a8 = argparse.Namespace(**vars(args))
        a8.gpus, a8.nlocal = 1, 8
        t0 = time.time()
        paths = [build_index(a8, j, dev)[0] for j in range(8)]
        build_s = time.time() - t0
        ixs = [pydsm.Index(pth, device=local) for pth in paths]
        for pm, px, lab in ((2, 0, "configs[2] share"), (1, 1, "configs[4] share (sample-specific substrings)")):
            rec = one(ixs, "8 read sets of %d x %d bp resident on one GPU, d=8, pmin=%d pmax=%d: %s" % (args.reads, args.rlen, pm, px, lab),
                      pmin=pm, pmax=px)
            rec.update(dtype="u32", index_build_s=build_s)
            out.append(rec)
        for x in ixs:
            x.close()
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "8 samples on one GPU", "error": repr(e)})
    try:  # BASELINE configs[4] at its real d: 64 read sets resident on the one card, -P 1 --pmax 1 (sample-specific substrings).  A tenth of
        # the default sample size, so that the 64 index builds keep the default run within minutes (the files
        # tests/test_many_samples_gpu.py builds: shared through DSM_BENCH_DIR)
        want("d64")
        a64 = argparse.Namespace(**vars(args))
        a64.gpus, a64.nlocal = 1, 64
        a64.reads, a64.genome = max(1000, args.reads // 10), max(5000, args.genome // 10)
        t0 = time.time()
        paths = [build_index(a64, j, dev)[0] for j in range(64)]
        build_s = time.time() - t0
        ixs = [pydsm.Index(pth, device=local) for pth in paths]
        for pm, px, lab in ((1, 1, "configs[4]: sample-specific substrings"), (2, 0, "the reference's default filter: reader-set orders past 13 samples")):
            rec = one(ixs, "64 read sets of %d x %d bp (n=%d each) resident on one GPU, d=64, pmin=%d pmax=%d: %s" % (a64.reads, args.rlen, ixs[0].n, pm, px, lab),
                      pmin=pm, pmax=px)
            rec.update(dtype="u32", index_build_s=build_s, index_hbm_bytes=sum(x.device_bytes() for x in ixs))
            out.append(rec)
        for x in ixs:
            x.close()
        note("64 samples done")
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "64 samples on one GPU (configs[4])", "error": repr(e)})
    try:  # BASELINE configs[3]: 4-Gbase read sets (n = 8.08e9 > 2^32, real BWTs built here by dsm_bwt_build), first one alone, then
        # its one-card share: EIGHT of them resident (8 x 4 GB of index), d = 8, -P 2 --pmax 8
        want("big")
        from pydsm import builder
        big = 4 * args.reads
        nbig = int(os.environ.get("DSM_BENCH_BIG_SAMPLES", "8"))
        bpaths = []
        t0 = time.time()
        for k in range(nbig):  # (the files tests/test_fullsize_gpu.py::test_real_bwt_beyond_2_32 builds: shared when both run on one box)
            pth = os.path.join(args.workdir, "real-%d%s.fmi" % (big, "" if k == 0 else ("-b" if k == 1 else "-%d" % k)))
            if not os.path.exists(pth):
                codes = builder.synth_reads(4242 + k, big, args.rlen, 5 * big, args.sub_rate, device=dev, private_frac=0.05 if k else 0.0)
                builder.build_from_codes(codes, pth + ".tmp")
                del codes
                torch.cuda.empty_cache()
                os.replace(pth + ".tmp", pth)
                note("4-Gbase sample %d of %d built (%.0f s so far)" % (k + 1, nbig, time.time() - t0))
            bpaths.append(pth)
        build_s = time.time() - t0
        bixs = [pydsm.Index(pth, device=local) for pth in bpaths]
        rec = one(bixs[:1], "1 read set of %d x %d bp (n=%d > 2^32, 64-bit positions), configs[3]'s sample size, pmin=1" % (big, args.rlen, bixs[0].n),
                  pmin=1)
        rec.update(dtype="u64")
        out.append(rec)
        note("one 4-Gbase sample done")
        if nbig > 1:
            with pydsm.Miner(bixs, fmin=args.fmin, emax=args.emax, pmin=2, pmax=8, stream=torch.cuda.current_stream().cuda_stream) as m:
                m.mine_many(prefixes[:1], text=False)  # warm-up: the first prefix only (allocations, pinned buffers)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st = m.mine_many(prefixes, text=False)[1]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            out.append({"record": "configs[3] one-card share: %d read sets of %d x %d bp (n=%d each) resident on one GPU, d=%d, pmin=2 pmax=8, "
                                  "64-bit positions" % (nbig, big, args.rlen, bixs[0].n, nbig), "value": st.reported / dt, "unit": "substrings/s",
                        "ms_per_step": dt * 1e3, "nodes": st.reported, "tuples": st.tuples, "union_nodes": st.union_nodes, "expand_ms": st.expand_ms,
                        "device_ms": st.device_ms, "splits": st.splits, "max_frontier": st.max_frontier, "dtype": "u64", "index_build_s": build_s,
                        "index_hbm_bytes": sum(x.device_bytes() for x in bixs)})
        for x in bixs:
            x.close()
    except Skip:
        pass
    except Exception as e:  # noqa: BLE001
        out.append({"record": "4-Gbase samples (configs[3])", "error": repr(e)})
    return out


def server_record(args, dev, local, pydsm):
    """Four read sets of a tenth of the benchmark's size each; their clients' wire streams for prefix A (our own client: byte-identical to
    the reference's) are fed in 1 MB pieces by four threads into dsm_server -- once merged after the last stream has ended (what
    round 2's server did), once merged while they arrive in 16 subtrees.  Tuples are identical; the record is the second run."""
    import threading
    import torch
    from pydsm import builder
    reads = max(100_000, args.reads // 10)
    streams = []
    for s_ in range(4):
        pth = os.path.join(args.workdir, "srv-%d.s%d_r%d_l%d.fmi" % (s_, 200 + s_, reads, args.rlen))
        if not os.path.exists(pth):
            codes = builder.synth_reads(200 + s_, reads, args.rlen, reads * 5, args.sub_rate, device=dev, private_frac=0.05)
            builder.build_from_codes(codes, pth + ".tmp")
            del codes
            os.replace(pth + ".tmp", pth)
        ixs = pydsm.Index(pth, device=local)
        with pydsm.Miner([ixs], fmin=args.fmin, stream_mode=True) as m:
            streams.append(m.enumerate("A", with_header=False)[0])
        ixs.close()
    torch.cuda.empty_cache()

    def run(prefix_len, unit_extra):
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        low, stop = [free0], [False]

        def sampler():
            while not stop[0]:
                low[0] = min(low[0], torch.cuda.mem_get_info()[0])
                time.sleep(0.002)
        srv = pydsm.Server(len(streams), prefix_len=prefix_len, unit_extra=unit_extra, pmin=2, emax=args.emax, text=False)
        ts = threading.Thread(target=sampler)
        ts.start()
        t0 = time.perf_counter()

        def reader(i):
            b = streams[i]
            for o in range(0, len(b), 1 << 20):
                srv.feed(i, b[o:o + (1 << 20)])
            srv.end(i)
        ths = [threading.Thread(target=reader, args=(i,)) for i in range(len(streams))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t1 = time.perf_counter()
        _, st = srv.finish()
        t2 = time.perf_counter()
        stop[0] = True
        ts.join()
        units = srv.units()[0]
        srv.close()
        return {"feed_s": t1 - t0, "after_last_byte_s": t2 - t1, "tuples": st.tuples, "union_nodes": st.union_nodes, "units": units,
                "device_peak_mb": (free0 - low[0]) / 1e6}
    at_end = run(None, 0)
    live = run(1, 1)
    tot = live["feed_s"] + live["after_last_byte_s"]
    return {"record": "server side: 4 client streams of %.0f MB each (prefix A of 4 read sets of %d x %d bp) fed by 4 threads into dsm_server, merged "
                      "while they arrive (16 subtrees)" % (len(streams[0]) / 1e6, reads, args.rlen),
            "value": live["union_nodes"] / tot, "unit": "merged nodes/s", "ms_per_step": tot * 1e3, "union_nodes": live["union_nodes"],
            "tuples": live["tuples"], "subtrees_merged_while_receiving": live["units"], "seconds_after_the_last_byte": live["after_last_byte_s"],
            "device_memory_peak_mb": live["device_peak_mb"], "device_memory_peak_mb_when_merged_at_the_end": at_end["device_peak_mb"],
            "tuples_when_merged_at_the_end": at_end["tuples"], "stream_bytes": int(sum(len(b) for b in streams)), "dtype": "u32"}


def main():
    args = parse()
    if os.environ.get("DSM_BENCH_WATCHDOG"):  # rehearsal aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DSM_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (one rank per GPU) before
        # anything here has touched the GPU, and leave with its exit code.  (Never exec from a process that has initialised HIP.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # under a launcher its world size counts
        args.gpus = world
    ndev = torch.cuda.device_count()
    local = local % max(1, ndev)  # rehearsals with more ranks than cards (DSM_BENCH_BACKEND=gloo) share a card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    forced = bool(args.force_exchange) and world == 1
    if forced:
        os.environ["DSM_FORCE_EXCHANGE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # any free port: this world of one talks to nobody
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(os.environ.get("DSM_BENCH_BACKEND", "nccl"), rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DSM_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import pydsm

    paths, build_s = [], 0.0
    for j in range(args.nlocal):
        pth, bs = build_index(args, rank * args.nlocal + j, dev)
        paths.append(pth)
        build_s += bs
    path = paths[0]
    ixs = [pydsm.Index(pth, device=local) for pth in paths]
    ix = ixs[0]
    # One-letter prefixes at every N: a level costs a fixed ~50 us of launches (plus one all-gather when N > 1), and 16
    # two-letter prefixes have four times the levels (measured at N=1: 290 vs 353 ms per pass).  With more ranks than
    # prefixes some ranks emit nothing; that costs less than the extra levels.
    def run_mode(mode):
        """one measurement (lanes, miners, warm-up, K timed steps) with the given exchange; the miners stay open (the caller closes them)"""
        owner_mode = mode == "owner" and (world > 1 or forced)
        if owner_mode and world * args.nlocal < 2:
            # (the owner partition lives in the several-sample engine: handles derived in the LF-step kernel)
            raise SystemExit("bench: --exchange owner/both needs at least two samples in the job (--force-exchange at N=1: add --nlocal 2)")
        # owner mode wants at least as many prefixes as ranks (every rank the server of some prefix): two letters beyond four ranks
        plen = args.prefix_len if args.prefix_len >= 0 else (2 if owner_mode and world > 4 else 1)
        prefixes = ["".join(p) for p in itertools.product("ACGT", repeat=plen)] if plen > 0 else [""]
        pmin = args.pmin if args.pmin > 0 else (1 if world * args.nlocal == 1 else 2)

        # exchange buffers owned by torch so that torch.distributed (RCCL over xGMI) all-gathers them device to device:
        # one collective per frontier level, nothing else on the data path.  A lane = one miner with its own HIP stream and
        # (multi-rank) its own communicator; lanes take the prefixes round-robin and run concurrently, so one lane's
        # collective overlaps the other lane's kernels.
        # Multi-rank runs use two lanes so that one lane's all-gather overlaps the other lane's kernels.  Both lanes use the ONE
        # default communicator; pydsm.dist.TurnGate makes them enqueue their collectives in strict alternation, i.e. in the same
        # order on every rank (each lane issues the same number of collectives everywhere because every rank walks the same
        # union trie).  DSM_BENCH_LANES / --lanes override.
        nlanes = args.lanes if args.lanes > 0 else int(os.environ.get("DSM_BENCH_LANES", "1" if world == 1 and not forced else "2"))
        if owner_mode:
            nlanes = world  # lane j: the prefixes rank j serves
        nlanes = max(1, min(nlanes, len(prefixes))) if not owner_mode else nlanes
        lanes = []
        gate = None
        # The exchange itself: by default the library's own (dsm_rccl_*: one RCCL communicator per lane, ncclAllGather from the C
        # callback on the lane's stream, no Python per level; torch.distributed only carries the ids and the timing barrier).
        # DSM_BENCH_EXCHANGE=torch (and every non-nccl backend, e.g. gloo rehearsals with several ranks on one card) goes through
        # pydsm.dist.Exchange = torch.distributed.all_gather_into_tensor from a Python callback.
        native = (world > 1 or forced) and dist.get_backend() == "nccl" and os.environ.get("DSM_BENCH_EXCHANGE", "rccl") == "rccl"
        if (world > 1 or forced) and nlanes > 1:
            if native:
                gate = pydsm.RcclGate(nlanes)
            else:
                from pydsm.dist import TurnGate
                gate = TurnGate(nlanes)  # same enqueue order of the lanes' collectives on every rank
        for j in range(nlanes):
            lane = {"prefixes": prefixes[j::nlanes], "stream": torch.cuda.Stream(device=dev) if nlanes > 1 else torch.cuda.current_stream()}
            allgather = exchange = None
            if native:
                try:
                    ids = [pydsm.RcclComm.unique_id() if rank == 0 else None]
                except pydsm.DsmError as e:  # librccl could not be loaded: the same on every rank of a node
                    ids = [None]
                    print("bench.py: native exchange unavailable (%s); using torch.distributed" % e, file=sys.stderr)
                if world > 1:
                    dist.broadcast_object_list(ids, src=0)
                if ids[0] is None:
                    native = False
                    if gate is not None:
                        from pydsm.dist import TurnGate
                        gate = TurnGate(nlanes)
                else:
                    lane["rccl"] = pydsm.RcclComm(ids[0], world, rank, local)  # collective over the ranks
                    allgather = lane["rccl"]
            if native:
                pass
            elif world > 1 or forced:
                from pydsm.dist import Exchange
                group = None  # the default communicator, shared by the lanes
                lane["ex"] = Exchange(int(os.environ.get("DSM_BENCH_XBYTES", str(1 << 30))) // nlanes, world, dev, group=group, stream=lane["stream"], lane=j)
                allgather, exchange = lane["ex"].allgather, lane["ex"].params()
            arena = 0
            if nlanes > 1:  # split what is left between the lanes still to be created (and the ranks sharing this card)
                free_b, _ = torch.cuda.mem_get_info(dev)
                sharing = (world + ndev - 1) // max(1, ndev)
                want = (256 << 20) + 64 * sum(x.n for x in ixs) * world
                arena = int(min(want, free_b * 0.7 / (nlanes - j) / sharing))
            lane["miner"] = pydsm.Miner(ixs, fmin=args.fmin, pmin=pmin, pmax=args.pmax, emax=args.emax, world_size=world, rank=rank,
                                        allgather=allgather, exchange=exchange, stream=lane["stream"].cuda_stream,
                                        emit_owner_only=world > 1 or forced, arena_bytes=arena, wide=1 if args.wide else 0,
                                        stream_mode=args.stream_mode, owner_rank=j if owner_mode else None,
                                        owner_exchange=(lane.get("rccl") or lane.get("ex")) if owner_mode else None)
            lanes.append(lane)
        if gate is not None:  # creation-time collectives ran lane by lane on the main thread; from here on lanes take turns
            for j, ln in enumerate(lanes):
                if native:
                    ln["rccl"].attach_gate(gate, j)
                else:
                    ln["ex"].gate = gate

        tot = {"reported": 0, "rank_ops": 0, "lf_steps": 0, "expand_ms": 0.0, "launches": 0, "tuples": 0, "union": 0,
               "device_ms": 0.0, "host_ms": 0.0, "cand": 0, "index_lines": 0, "records_read": 0, "record_bytes": 0, "slots": 0, "colbytes": 0,
               "xsent": 0, "xrecv": 0}
        import threading
        tot_lock = threading.Lock()

        def lane_step(lane, record, errs):
            try:
                if gate is not None:
                    gate.begin(lanes.index(lane))
                with torch.cuda.stream(lane["stream"]):
                    if args.stream_mode:
                        nbs, st = lane["miner"].enumerate_many(lane["prefixes"], discard=True)
                        tot["wire_bytes"] = tot.get("wire_bytes", 0) + (sum(nbs) if record else 0)
                    else:
                        st = lane["miner"].mine_many(lane["prefixes"], text=False)[1]
                if record:
                    with tot_lock:
                        tot["reported"] += st.reported
                        tot["rank_ops"] += st.rank_ops
                        tot["lf_steps"] += st.lf_steps
                        tot["expand_ms"] += st.expand_ms
                        tot["launches"] += st.expand_launches
                        tot["tuples"] += st.tuples
                        tot["union"] += st.union_nodes
                        tot["device_ms"] += st.device_ms
                        tot["host_ms"] += st.host_ms
                        tot["cand"] += st.candidates
                        tot["index_lines"] += st.index_lines
                        tot["records_read"] += st.records_read
                        tot["record_bytes"] += st.record_bytes
                        tot["slots"] += st.expand_slots
                        tot["colbytes"] += st.expand_column_bytes
                        tot["xsent"] += st.exchange_bytes_sent
                        tot["xrecv"] += st.exchange_bytes_received
                        tot["splits"] = tot.get("splits", 0) + st.splits
                        tot["levels"] = tot.get("levels", 0) + st.levels
                        tot["max_frontier"] = max(tot.get("max_frontier", 0), st.max_frontier)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
                if world > 1:  # the other ranks are blocked in a collective: fail the whole job instead of hanging it
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
                    os._exit(13)
            finally:
                if gate is not None:
                    gate.retire(lanes.index(lane))

        def step(record):
            errs = []
            if gate is not None:
                gate.reset()
            if len(lanes) == 1:
                lane_step(lanes[0], record, errs)
            else:
                ths = [threading.Thread(target=lane_step, args=(ln, record, errs)) for ln in lanes]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
            if errs:
                raise errs[0]

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        if os.environ.get("DSM_TRACE_EXCHANGE") and world > 1 and gate is not None and not native:
            def dump_order():
                with open(os.path.join(ROOT, "gpurun_out", "order_rank%d.txt" % rank), "w") as f:
                    for it in gate.log:
                        f.write("%d %d\n" % it)
            import atexit
            atexit.register(dump_order)
            import threading as _th
            _t = _th.Timer(100, dump_order)
            _t.daemon = True
            _t.start()
        if os.environ.get("DSM_TRACE_EXCHANGE") and world > 1 and not native:
            import atexit
            atexit.register(lambda: print("TRACE rank %d: %s" % (rank, lanes[0]["ex"].trace[:60]), file=sys.stderr, flush=True))
        for _ in range(args.warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt, float(tot["reported"])], dtype=torch.float64, device=dev if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
        if world > 1:
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dt = float(tmax[0])
            reported_all = float(tsum[1])
        else:
            reported_all = float(tot["reported"])
        return {"owner_mode": owner_mode, "plen": plen, "prefixes": prefixes, "pmin": pmin, "nlanes": nlanes, "lanes": lanes, "native": native,
                "tot": tot, "dt": dt, "reported_all": reported_all, "mode": mode}

    # --exchange both (N > 1): the all-gather and the owner partition on the same indexes, one after the other; the line reports both and
    # carries the faster one as its value -- the first run on a multi-GPU node then says which partition to keep
    both = None
    if args.exchange == "both" and (world > 1 or forced):
        ra = run_mode("allgather")
        for ln in ra["lanes"]:
            ln["miner"].close()
        rb = run_mode("owner")
        va, vb = ra["reported_all"] / ra["dt"], rb["reported_all"] / rb["dt"]
        both = {"allgather": {"value": va, "ms_per_step": ra["dt"] * 1e3 / args.steps, "exchange_bytes_sent_per_step_rank0": ra["tot"]["xsent"] / max(1, args.steps),
                              "exchange_bytes_received_per_step_rank0": ra["tot"]["xrecv"] / max(1, args.steps), "lanes": ra["nlanes"],
                              "collectives_per_step_rank0": ra["tot"].get("levels", 0) / max(1, args.steps)},
                "owner": {"value": vb, "ms_per_step": rb["dt"] * 1e3 / args.steps, "exchange_bytes_sent_per_step_rank0": rb["tot"]["xsent"] / max(1, args.steps),
                          "exchange_bytes_received_per_step_rank0": rb["tot"]["xrecv"] / max(1, args.steps), "lanes": rb["nlanes"],
                          "collectives_per_step_rank0": 2 * rb["tot"].get("levels", 0) / max(1, args.steps)}}
        both["headline"] = "allgather" if va > vb else "owner"
        res = ra if va > vb else rb   # the line describes the faster mode
        open_lanes = rb["lanes"]      # (the owner run's miners are the ones still open)
    else:
        res = run_mode("owner" if args.exchange == "owner" else "allgather")
        open_lanes = res["lanes"]
    owner_mode, plen, prefixes, pmin, nlanes, lanes, native = (res[k] for k in ("owner_mode", "plen", "prefixes", "pmin", "nlanes", "lanes", "native"))
    tot, dt, reported_all = res["tot"], res["dt"], res["reported_all"]
    # Text (metaserver.cpp:472-484's printf loop) is outside the timed region of `value` (the sink receives binary batches).  One more
    # pass with a sink that formats every batch on the GPU (dsm_formatter_format, the drop-in's way since round 4), timed twice: the
    # calls alone (`format_ms_per_step`) and the whole pass (`text_pass_ms`: what a user of the text surface waits for; the formatter's
    # transfers share the bus with the next prefix's tuples).  `format_host_ms_per_step`: the same text from the host's snprintf loop,
    # on a tenth of the batches, scaled -- the round-3 path, kept as the checker.
    format_ms, format_bytes, text_pass_ms, format_host_ms, text_pass_formatter_ms, text_mode_bytes = None, 0, None, None, None, None
    if rank == 0 and world == 1 and not args.stream_mode and not forced and not args.no_cpu:  # (--no-cpu: profiling runs, exactly steps + warmup passes)
        import ctypes as C
        acc = {"s": 0.0, "bytes": 0, "host_s": 0.0, "host_bytes": 0, "k": 0}
        fm = pydsm.Formatter(local)

        def fmt(b):
            t = C.c_void_p()
            n = C.c_size_t(0)
            t1 = time.perf_counter()
            rc = pydsm.lib().dsm_formatter_format(fm.h, C.byref(b), C.byref(t), C.byref(n))
            acc["s"] += time.perf_counter() - t1
            if rc == 0:
                acc["bytes"] += n.value
            acc["k"] += 1
            if acc["k"] % 10 == 1:
                t1 = time.perf_counter()
                rc = pydsm.lib().dsm_format_batch(C.byref(b), C.byref(t), C.byref(n))
                acc["host_s"] += time.perf_counter() - t1
                if rc == 0:
                    acc["host_bytes"] += n.value
                    pydsm.lib().dsm_free(t)

        with torch.cuda.stream(lanes[0]["stream"]):
            lanes[0]["miner"].mine_many(lanes[0]["prefixes"], text=False, on_batch=fmt)
        format_ms, format_bytes = acc["s"] * 1e3, acc["bytes"]
        if acc["host_bytes"]:
            format_host_ms = acc["host_s"] * 1e3 * acc["bytes"] / acc["host_bytes"]
        acc2 = {"bytes": 0}

        def fmt_only(b):
            t = C.c_void_p()
            n = C.c_size_t(0)
            if pydsm.lib().dsm_formatter_format(fm.h, C.byref(b), C.byref(t), C.byref(n)) == 0:
                acc2["bytes"] += n.value

        with torch.cuda.stream(lanes[0]["stream"]):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lanes[0]["miner"].mine_many(lanes[0]["prefixes"], text=False, on_batch=fmt_only)
            torch.cuda.synchronize()
            text_pass_formatter_ms = (time.perf_counter() - t1) * 1e3
            # ... and the text mode of the emitter (dsm_miner_mine_text: entropy, filter and lines on the card, only text over the bus):
            # what dsm_node does
            acc3 = {"bytes": 0}

            def count(ptr, n):  # (a sink that takes the bytes where the library left them, as fwrite would)
                acc3["bytes"] += n
            lanes[0]["miner"].mine_text(lanes[0]["prefixes"], on_raw=count)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lanes[0]["miner"].mine_text(lanes[0]["prefixes"], on_raw=count)
            torch.cuda.synchronize()
            text_pass_ms = (time.perf_counter() - t1) * 1e3
            text_mode_bytes = acc3["bytes"] // 2
        fm.close()
    for ln in open_lanes:
        ln["miner"].close()

    if rank == 0:
        # Bytes the LF-step kernel moves, from its own exact counters: 64 B per distinct index block of a
        # tile, per thread a 4-byte record handle, its column entry and 0.5 B of child planes, and the records it read and wrote
        # (16-byte compact words in all but the top levels of a prefix).
        kernel_bytes = 64 * tot["index_lines"] + 4 * tot["slots"] + tot["slots"] // 2 + tot["colbytes"] + tot["record_bytes"]
        esec = tot["expand_ms"] * 1e-3
        ach = kernel_bytes / esec / 1e9 if esec > 0 else 0.0
        ref_equiv = tot["rank_ops"] * ALG_BYTES_PER_RANK / esec / 1e9 if esec > 0 else 0.0
        traffic, tinfo = None, {}
        tj = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from rocprofv3 --pmc (see profiles/README)
        kernel_sha = lf_kernel_sha()
        if os.path.exists(tj):
            try:
                tinfo = json.load(open(tj))
                # the offline counters describe ONE build of the kernel: they are quoted only next to the source they were taken from
                if (tinfo.get("reads") == args.reads and tinfo.get("prefix_len") == plen and tinfo.get("gpus") == args.gpus and args.nlocal == 1
                        and not args.wide and not args.stream_mode and tinfo.get("kernel_sha16") == kernel_sha):
                    traffic = tinfo.get("bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        traffic_gbs = (traffic / (tot["expand_ms"] / max(1, tot["launches"]) * 1e-3) / 1e9) if traffic and tot["expand_ms"] > 0 else None
        out = {
            "metric": "enumerated substrings/sec (Emax=2.0)", "value": reported_all / dt, "unit": "substrings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64" if (args.wide or ix.n >= 0xFFFFFFF0) else "u32",
            "data": "synthetic",
            "config": {"workload": "%d synthetic read set(s) of %d x %d bp (n=%d BWT symbols each), %d FM-index(es) per GPU, "
                                   "fmin=%d Emax=%g pmin=%d pmax=%d%s%s, %d prefixes" % (world * args.nlocal, args.reads, args.rlen, ix.n, args.nlocal, args.fmin, args.emax, pmin,
                                                                              args.pmax, ", 64-bit positions" if args.wide else "",
                                                                              ", wire-stream mode" if args.stream_mode else "", len(prefixes)),
                       "parallelism": ("sample-per-gpu x%d, prefix k merged by rank k %% %d (columns gathered to it, child masks broadcast back), "
                                       "%d lanes per GPU" % (world, world, nlanes)) if owner_mode else
                                      "sample-per-gpu x%d, one all-gather per frontier level, %d prefix lane(s) per GPU" % (world, nlanes),
                       "exchange": ("none (single process)" if not (world > 1 or forced) else
                                    "dsm_rccl: ncclAllGather from the library's callback, one communicator per lane" if native else
                                    "torch.distributed all_gather_into_tensor (%s) from a Python callback" % dist.get_backend())},
            # `achieved` / `frac`: live and self-counted -- the bytes the kernel itself counted in THIS run (64 B per distinct index block
            # of a tile, handles, column entries, planes, records: the algorithmic bytes of DESIGN.md section 4) / the HIP-event time of
            # this run's launches / 8 TB/s.  `traffic` / `pmc_*`: HBM bytes per launch as the PMC counters saw them (rocprofv3 --pmc
            # FETCH_SIZE / WRITE_SIZE passes of this command, tools/profiling/r04_final.sh -> profiles/traffic.json), an OFFLINE figure
            # that is quoted only when the profile was taken from this very kernel source (`kernel_sha16`), never mixed into `frac`.
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "frac_source": "kernel byte counters of this run / live HIP-event time",
                         "traffic": traffic, "traffic_source": "profiles/traffic.json (offline rocprofv3 --pmc of the same command and kernel source)" if traffic else None,
                         "pmc_gbs": traffic_gbs, "pmc_frac": traffic_gbs / HBM_PEAK_GBS if traffic_gbs is not None else None,
                         "kernel_sha16": kernel_sha,
                         "model_gbs": ach, "model_frac": ach / HBM_PEAK_GBS,
                         "kernel": "expand_kernel (LF-step)", "launches": tot["launches"],
                         "avg_launch_ms": tot["expand_ms"] / max(1, tot["launches"]),                     # HIP events, this run
                         "rocprof_avg_launch_ms": tinfo.get("rocprof_avg_launch_ms") if traffic else None,  # kernel trace of the profiled run
                         "rocprof_expand_ms_per_step": tinfo.get("rocprof_expand_ms_per_step") if traffic else None,
                         # what the kernel is bound by: vector-instruction issue -- SQ counters of the same offline passes
                         "valu_busy_of_resident_wave_time": (tinfo.get("valu") or {}).get("busy_of_resident_wave_time") if traffic else None,
                         "valu_insts_per_node": (tinfo.get("valu") or {}).get("insts_per_node") if traffic else None,
                         "bytes_per_launch": kernel_bytes / max(1, tot["launches"]),
                         "bytes_per_node": kernel_bytes / max(1, tot["reported"]),
                         "index_lines_per_node": tot["index_lines"] / max(1, tot["reported"]),
                         # SURVEY 8d's accounting (17 B per BitRank::rank the REFERENCE would execute): measures the data-structure
                         # win, not kernel efficiency -- one 64-byte block answers many reference rank-ops, so it may exceed the peak
                         "reference_equivalent_gbs": ref_equiv,
                         # with more than one lane the expand launches of the lanes share the device: their durations
                         # (and so `achieved`) describe two kernels running side by side, not one kernel alone
                         "concurrent_lanes": nlanes},
            "detail": {"rank0_nodes_per_step": tot["reported"] / max(1, args.steps), "rank_ops_per_node": tot["rank_ops"] / max(1, tot["reported"]),
                       "lf_per_node": tot["lf_steps"] / max(1, tot["reported"]), "tuples_per_step": tot["tuples"] / max(1, args.steps),
                       "prefix_splits_per_step": tot.get("splits", 0) / max(1, args.steps), "levels_per_step": tot.get("levels", 0) / max(1, args.steps),
                       "max_frontier": tot.get("max_frontier", 0),
                       "union_nodes_per_step": tot["union"] / max(1, args.steps), "candidates_per_step": tot["cand"] / max(1, args.steps),
                       "expand_ms_per_step": tot["expand_ms"] / max(1, args.steps), "device_ms_per_step": tot["device_ms"] / max(1, args.steps),
                       "host_ms_per_step": tot["host_ms"] / max(1, args.steps), "index_build_s": build_s,
                       "format_ms_per_step": format_ms, "format_text_bytes_per_step": format_bytes, "text_pass_ms": text_pass_ms, "text_pass_bytes": text_mode_bytes, "text_pass_with_formatter_sink_ms": text_pass_formatter_ms,
                       "format_host_ms_per_step": format_host_ms,
                       "exchange_bytes_sent_per_step_rank0": tot["xsent"] / max(1, args.steps),
                       "exchange_bytes_received_per_step_rank0": tot["xrecv"] / max(1, args.steps),
                       # collectives rank 0 issued per pass: one all-gather per level, or (owner mode) a gather and a broadcast per level,
                       # over all its lanes
                       "collectives_per_step_rank0": ((2 if owner_mode else 1) * tot.get("levels", 0) / max(1, args.steps)) if (world > 1 or forced) else 0,
                       "lanes": nlanes,
                       "index_hbm_bytes": ix.device_bytes(), "wire_bytes_per_step": tot.get("wire_bytes", 0) / max(1, args.steps),
                       # the exact entropy (metaserver.cpp:379,389) is the host's libm: its version goes next to every result
                       "glibc": os.confstr("CS_GNU_LIBC_VERSION") if hasattr(os, "confstr") else None},
        }
        if both is not None:
            out["exchange_modes"] = both
        print("bench: gpu leg done: %.3e substrings/s, %.1f ms/step" % (out["value"], out["ms_per_step"]), file=sys.stderr, flush=True)
        default_cfg = world == 1 and args.nlocal == 1 and not args.stream_mode and not forced
        if default_cfg and not args.no_cpu:
            ref = None
            try:
                ref = cpu_baseline_reference(path, args, ix, pydsm)
            except Exception as e:  # noqa: BLE001
                print("bench: reference CPU baseline failed: %r" % (e,), file=sys.stderr, flush=True)
            port = cpu_baseline(path, args, pmin, seconds=args.cpu_seconds / 2 if ref is not None else None)
            if ref is not None:
                ref["port_substrings_per_s"] = port["value"]  # our CPU restatement of the same path, same cores, for comparison
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
        if default_cfg and not args.no_extras and not args.no_cpu and not args.wide and args.reads >= 1_000_000:
            out["extra_records"] = extra_records(args, dev, local, pydsm, ix, path, prefixes, pmin)
        print(json.dumps(out), flush=True)
    for x in ixs:
        x.close()
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
