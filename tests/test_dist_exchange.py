"""N>1 path, host side: the per-level all-gather plumbing with 2 gloo ranks on CPU (layout, halves, errors),
and -- on the GPU box -- two ranks sharing the one GPU, each owning one sample, against the reference golden."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
    _init(rank, world, port)
    from pydsm.dist import Exchange
    ex = Exchange(1024, world, "cpu")
    ok = True
    for level, nbytes in enumerate([16, 1024, 40, 0, 8]):
        ex.send[:nbytes] = torch.arange(nbytes, dtype=torch.uint8) + 17 * rank + level
        half = (level & 1) * 1024 * world
        ex.allgather(ex.send.data_ptr(), ex.recv.data_ptr() + half, nbytes)
        for r in range(world):
            want = (torch.arange(nbytes, dtype=torch.uint8) + 17 * r + level)
            got = ex.recv[half + r * nbytes: half + (r + 1) * nbytes]
            ok = ok and torch.equal(got, want)
    # previous level's half is untouched by the next exchange
    for bad in (lambda: ex.allgather(ex.send.data_ptr() + 1, ex.recv.data_ptr(), 8),
                lambda: ex.allgather(ex.send.data_ptr(), ex.recv.data_ptr() + 8, 8),
                lambda: ex.allgather(ex.send.data_ptr(), ex.recv.data_ptr(), 4096)):
        try:
            bad()
            ok = False
        except ValueError:
            pass
    q.put((rank, ok, ex.calls))
    dist.destroy_process_group()


def test_exchange_layout_two_gloo_ranks_cpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True, 5), (1, True, 5)]


def _gpu_worker(rank, world, port, q, fmis, prefixes, kw):
    sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
    _init(rank, world, port)
    import pydsm
    from pydsm.dist import Exchange
    torch.cuda.set_device(0)
    ex = Exchange(1 << 22, world, "cuda:0")
    nloc = len(fmis) // world
    idx = [pydsm.Index(f, device=0) for f in fmis[rank * nloc:(rank + 1) * nloc]]
    out = []
    with pydsm.Miner(idx, world_size=world, rank=rank, allgather=ex.allgather, exchange=ex.params(), **kw) as m:
        for p in prefixes:
            text, st = m.mine(p)
            out.append((p, text, st.reported, st.union_nodes, st.pair_order_exact))
    # owner-only emission: prefix k is emitted by rank k % world, everybody expands everything
    with pydsm.Miner(idx, world_size=world, rank=rank, allgather=ex.allgather, exchange=ex.params(), emit_owner_only=True, **kw) as m:
        parts = []

        def on_batch(b):
            parts.append(int(b.ntuples))
        text, st = m.mine_many(prefixes, on_batch=on_batch)
        out.append(("owner", text, st.reported, st.union_nodes, st.pair_order_exact))
    # a budget so small that levels / the owner's candidate store overflow: every rank must split the same prefixes
    with pydsm.Miner(idx, world_size=world, rank=rank, allgather=ex.allgather, exchange=ex.params(), emit_owner_only=True,
                     arena_bytes=(5 << 20) * nloc, **kw) as m:
        text, st = m.mine_many(prefixes)
        out.append(("owner-split", text, st.splits, st.union_nodes, st.pair_order_exact))
    q.put((rank, out, ex.calls))
    for ix in idx:
        ix.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("setname,world,cfg,dense", [("toy3", 3, "default", 0), ("five", 5, "default", 1), ("toy3", 3, "pmax2", 1),
                                                      ("many30", 5, "p3", 0), ("many30", 5, "p3", 1)])   # 5 ranks x 6 samples: several samples per rank, d = 30
def test_one_sample_per_rank_matches_reference_server(golden, setname, world, cfg, dense, monkeypatch):
    import orc
    if dense:   # every compact level through the dense LF-step sweep (the workers inherit the environment)
        monkeypatch.setenv("DSM_DENSE_MIN", "0")
    from goldenlib import server_args_to_kw
    m = golden.manifest["sets"][setname]
    names = m["names"]
    fmis = [golden.fmi(setname, n) for n in names]
    kw = server_args_to_kw(m["server_cfgs"][cfg])
    kw["fmin"] = m["fmin"]
    if "maxdepth" in m:
        kw["maxdepth"] = m["maxdepth"]
    prefixes = ["A", "GT"] if setname == "toy3" else (["AC", "G"] if setname == "many30" else ["C"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, fmis, prefixes, kw)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    # every rank computes the same union trie; tuples must equal the reference server's stdout
    reported = {}
    splits = []
    for rank, out, calls in res:
        assert calls > 0
        for p, text, rep, union, exact in out:
            assert exact == 1
            if p in ("owner", "owner-split"):
                mine = b"".join(golden.server_out(setname, cfg, q) for k, q in enumerate(prefixes) if k % world == rank)
                assert text == mine, (rank, p)
                if p == "owner-split":
                    splits.append(rep)
                continue
            assert text == golden.server_out(setname, cfg, p), (rank, p)
            reported.setdefault(p, []).append(rep)
    # per-rank `reported` is that rank's own sample: the sum equals the oracle's client total
    assert len(set(splits)) == 1  # the same number of splits on every rank
    if setname == "toy3":
        assert splits[0] > 0, "the tiny budget was meant to force prefix splits"
    oidx = [orc.Index(f) for f in fmis]
    for p in prefixes:
        want = sum(ix.enumerate(n, p, fmin=m["fmin"], maxdepth=m.get("maxdepth", 0xFFFFFFFF))[1][0] for ix, n in zip(oidx, names))
        assert sum(reported[p]) == want


def _owner_worker(rank, world, port, q, fmis, prefixes, kw, arena):
    """Owner mode as a host drives it: one miner (lane) per owner, every rank the server of lane `rank` and a client in the others;
    the lanes run on their own threads and streams, their collectives ordered by a TurnGate (same order on every rank)."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
    _init(rank, world, port)
    import pydsm
    from pydsm.dist import Exchange, TurnGate
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    nloc = len(fmis) // world
    idx = [pydsm.Index(f, device=0) for f in fmis[rank * nloc:(rank + 1) * nloc]]
    gate = TurnGate(world)
    lanes = []
    for j in range(world):  # creation runs collectives: lane by lane, in the same order on every rank
        st = torch.cuda.Stream(device=dev)
        ex = Exchange(1 << 22, world, dev, stream=st, lane=j)
        m = pydsm.Miner(idx, world_size=world, rank=rank, allgather=ex.allgather, exchange=ex.params(), stream=st.cuda_stream,
                        owner_rank=j, owner_exchange=ex, arena_bytes=arena, **kw)
        lanes.append((st, ex, m))
    for st, ex, m in lanes:
        ex.gate = gate
    res = [None] * world
    errs = []

    def run(j):
        st, ex, m = lanes[j]
        gate.begin(j)
        try:
            with torch.cuda.stream(st):
                text, stt = m.mine_many(prefixes[j::world])
            res[j] = (text, stt.reported, stt.union_nodes, stt.splits, stt.exchange_bytes_sent, stt.exchange_bytes_received, stt.tuples)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
        finally:
            gate.retire(j)
    ths = [threading.Thread(target=run, args=(j,)) for j in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(240)
    q.put((rank, res, errs))
    if errs:
        q.close()
        q.join_thread()  # (the queue's feeder thread must have written the result before the process goes)
        os._exit(1)  # (the other ranks may be blocked in a collective)
    for st, ex, m in lanes:
        m.close()
    for ix in idx:
        ix.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("setname,world,cfg,arena", [("toy3", 3, "default", 0), ("toy3", 3, "default", 5 << 20), ("five", 5, "default", 0),
                                                      ("many30", 5, "p3", 0)])   # (fewer prefixes than ranks: some lanes stay idle)
def test_owner_mode_one_server_per_prefix(golden, setname, world, cfg, arena, monkeypatch):
    if setname != "toy3" or arena:   # (these cases: the dense LF-step sweep on every compact level, owner and clients alike)
        monkeypatch.setenv("DSM_DENSE_MIN", "0")
    """The reference's partition (one metaserver per prefix, wrapper-SLURM/example-server.sh:27-41; one client connection per
    prefix, metaenumerate.cpp:268-309) between ranks: prefix k is merged by rank k % world alone, the others send it their
    columns and get the union's child masks back.  Tuples of every prefix equal the reference server's stdout; a client receives
    far fewer bytes than it sends; a budget of a few MiB makes owner and clients split the same prefixes."""
    from goldenlib import server_args_to_kw
    m = golden.manifest["sets"][setname]
    fmis = [golden.fmi(setname, n) for n in m["names"]]
    kw = server_args_to_kw(m["server_cfgs"][cfg])
    kw["fmin"] = m["fmin"]
    if "maxdepth" in m:
        kw["maxdepth"] = m["maxdepth"]
    prefixes = {"toy3": ["A", "GT", "C", "T", "G", "AC", "TTG"], "five": ["A", "C", "G", "T"], "many30": ["AC", "G"]}[setname]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    nloc = len(fmis) // world
    ps = [ctx.Process(target=_owner_worker, args=(r, world, port, q, fmis, prefixes, kw, arena * nloc)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    splits = {}
    for rank, lanes, errs in res:
        assert not errs, errs
        for j, (text, reported, union, nsplit, sent, received, tuples) in enumerate(lanes):
            mine = b"".join(golden.server_out(setname, cfg, p) for p in prefixes[j::world])
            if not prefixes[j::world]:
                assert text == b"" and sent == received == 0
            elif j == rank:  # the server of this lane
                assert text == mine, (rank, j)
                assert received > sent > 0
            else:          # a client: it emits nothing, sends its columns and receives the masks
                assert text == b"" and tuples == 0
                assert sent > received > 0
            splits.setdefault(j, set()).add(nsplit)
    assert all(len(v) == 1 for v in splits.values())  # owner and clients split the same prefixes
    if arena:
        assert sum(next(iter(v)) for v in splits.values()) > 0, "the small budget was meant to force prefix splits"


def _owner_abort_worker(rank, world, port, q, fmis, prefixes, kw, fail_depth):
    os.environ["DSM_TEST_OWNER_FAIL"] = str(fail_depth)  # the library's test hook: the owner of a prefix gives up at this depth, after the gather
    _owner_worker(rank, world, port, q, fmis, prefixes, kw, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("fail_depth", [0, 3])
def test_owner_failure_reaches_every_rank(golden, fail_depth):
    """an owner that fails between a level's gather and its broadcast must not leave the clients waiting.  Every owner is
    made to fail (DSM_TEST_OWNER_FAIL) at a level after it has taken the columns: all three ranks come back, the owners with the
    injected error, the clients with "the prefix's owner failed" -- nobody hangs in a collective."""
    from goldenlib import server_args_to_kw
    m = golden.manifest["sets"]["toy3"]
    fmis = [golden.fmi("toy3", n) for n in m["names"]]
    kw = server_args_to_kw(m["server_cfgs"]["default"])
    kw["fmin"] = m["fmin"]
    prefixes = ["A", "C", "G"]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_owner_abort_worker, args=(r, world, port, q, fmis, prefixes, kw, fail_depth)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)  # (a hang shows as a timeout here)
    for p in ps:
        p.join(60)
    assert len(res) == world
    for rank, lanes, errs in res:
        assert len(errs) == world, (rank, errs)  # its own lane (owner) and the two lanes it is a client of
        assert sum("injected owner failure" in e for e in errs) == 1, errs
        assert sum("owner failed" in e for e in errs) == world - 1, errs


def test_turn_gate_interleaves_lanes_deterministically():
    """Two lane threads with different numbers of collectives: the global order is strict round-robin while both are active."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
    from pydsm.dist import TurnGate
    for trial in range(20):
        gate = TurnGate(2)
        log = []

        def lane(j, n):
            import random
            import time
            gate.begin(j)
            for k in range(n):
                time.sleep(random.random() * 0.002)
                gate.acquire(j)
                log.append((j, k))
                gate.release(j)
            gate.retire(j)
        ts = [threading.Thread(target=lane, args=(0, 5)), threading.Thread(target=lane, args=(1, 9))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(10)
        assert log == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2), (0, 3), (1, 3), (0, 4), (1, 4), (1, 5), (1, 6), (1, 7), (1, 8)]
        gate.reset()


def _nccl_worker(port, q):
    """One RCCL rank (two ranks cannot share a card under RCCL): the exchange exactly as bench.py drives it at N > 1 --
    two lane threads, each with its own HIP stream, collectives of the ONE default communicator ordered by a TurnGate."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "dsm-framework_amd"))
    from pydsm.dist import Exchange, TurnGate
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        gate = TurnGate(2)
        lanes = []
        for j in range(2):
            st = torch.cuda.Stream(device=dev)
            lanes.append((st, Exchange(1 << 16, 1, dev, stream=st, gate=gate, lane=j)))
        ok = [True, True]

        def run(j):
            st, ex = lanes[j]
            gate.begin(j)
            try:
                with torch.cuda.stream(st):
                    for level, nbytes in enumerate([16, 4096, 48, 65536, 32] * 8):
                        ex.send[:nbytes] = (level * 7 + j) % 251
                        half = (level & 1) * ex.nbytes * ex.world
                        ex.allgather(ex.send.data_ptr(), ex.recv.data_ptr() + half, nbytes, st.cuda_stream)
                        st.synchronize()
                        got = ex.recv[half: half + nbytes]
                        ok[j] = ok[j] and bool((got == (level * 7 + j) % 251).all())
            finally:
                gate.retire(j)
        ths = [threading.Thread(target=run, args=(j,)) for j in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        calls = [ex.calls for _, ex in lanes]
        dist.destroy_process_group()
        q.put(("ok", ok, calls))
    except Exception as e:  # noqa: BLE001
        q.put(("err", repr(e), None))


@pytest.mark.gpu
def test_rccl_exchange_from_two_lane_threads_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=60)
    assert res[0] == "ok", res
    assert res[1] == [True, True] and res[2] == [40, 40]


@pytest.mark.gpu
def test_bench_forced_exchange_rehearsal_on_rccl(tmp_path):
    """bench.py --force-exchange: one rank drives the complete N > 1 path (send buffer, one RCCL all-gather per level, two lane
    threads under the turn gate, capacity agreement, per-prefix status words, owner-only emission) with both exchanges and must
    report exactly the nodes and tuples of the plain single-rank run."""
    import json
    import subprocess
    env = dict(os.environ, DSM_BENCH_DIR=str(tmp_path))
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", "--reads", "200000", "--genome", "1000000"]
    outs = []
    # plain; the library's own RCCL exchange (dsm_rccl_*: ncclAllGather from the C callback, two communicators under the native
    # gate); the torch.distributed exchange (Python callback, one communicator under the TurnGate)
    for extra, ex in (([], None), (["--force-exchange"], "rccl"), (["--force-exchange"], "torch")):
        r = subprocess.run(base + extra, env=dict(env, DSM_BENCH_EXCHANGE=ex) if ex else env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b, c = outs
    assert b["config"]["exchange"].startswith("dsm_rccl") and c["config"]["exchange"].startswith("torch.distributed")
    for k in ("rank0_nodes_per_step", "tuples_per_step", "union_nodes_per_step", "candidates_per_step", "rank_ops_per_node"):
        assert a["detail"][k] == c["detail"][k], k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):            # the driver's contract for the JSON line
        assert k in a, k
    assert a["n_gpus"] == 1 and a["scaling"] == "weak" and a["higher_is_better"] is True and a["vs_baseline"] is None
    assert "workload" in a["config"] and a["dtype"] in ("u32", "u64") and a["data"] == "synthetic"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in a["roofline"], k
    assert a["roofline"]["bound"] == "hbm" and abs(a["roofline"]["frac"] - a["roofline"]["achieved"] / a["roofline"]["peak"]) < 1e-9
    assert b["config"]["parallelism"].endswith("2 prefix lane(s) per GPU")
    for k in ("rank0_nodes_per_step", "tuples_per_step", "union_nodes_per_step", "candidates_per_step", "rank_ops_per_node"):
        assert a["detail"][k] == b["detail"][k], k


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_card_over_gloo(tmp_path):
    """The driver's N > 1 launch line on the one card a test box has: `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2` with DSM_BENCH_BACKEND=gloo (two RCCL ranks cannot share a card): two samples, one per rank, one all-gather per level,
    two prefix lanes per rank.  Rank 0 prints one JSON line; its aggregate node count is the sum of both ranks' own nodes."""
    import json
    import subprocess
    env = dict(os.environ, DSM_BENCH_DIR=str(tmp_path), DSM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    def launch(extra):
        for attempt in range(3):  # (a port picked here can be taken again before the launcher binds it)
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu",
                   "--reads", "200000", "--genome", "1000000"] + extra
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
            if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
                break
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    d = launch([])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["exchange"].startswith("torch.distributed") and "2 prefix lane(s)" in d["config"]["parallelism"]
    assert d["detail"]["union_nodes_per_step"] > d["detail"]["rank0_nodes_per_step"] > 0   # two samples: the union trie is larger than one sample's
    # (the arena is sized from the miner's own allocations, not from the card's free memory, which moves with the other rank of the card)
    assert d["detail"]["prefix_splits_per_step"] == 0
    # the same launch line in owner mode: prefix k merged by rank k % 2 alone, one lane per owner; same nodes, same tuples
    o = launch(["--exchange", "owner"])
    assert "merged by rank k % 2" in o["config"]["parallelism"] and o["n_gpus"] == 2
    assert o["detail"]["rank0_nodes_per_step"] == d["detail"]["rank0_nodes_per_step"]  # (union_nodes depends on how prefixes were split)
    assert o["value"] > 0 and o["detail"]["exchange_bytes_sent_per_step_rank0"] > 0
