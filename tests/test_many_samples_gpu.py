"""BASELINE.json configs[4] at its real d: 64 read sets mined together, `-P 1 --pmax 1` (sample-specific substrings: a
substring is printed iff exactly one of the 64 samples holds it, metaserver.cpp:406-419; MAX_READERS, metaserver.cpp:19).

64 samples of DSM_MANY_READS (default 5 x 10^5) reads x 100 bp (n = 1.01e8 each, seeds 42..105, 5 % private sequence per sample;
bench.py's 64-sample record runs the same with 10^6 reads) are built and kept resident on the one card, d = 64: eight batched LF-step launches of eight samples
per level, reader sets past libstdc++'s first rehash (13 -> 29 -> 59 -> 127 buckets: order_big_kernel, setorder.h).
Tuples and all six counters must equal the oracle's for
  * -P 1 --pmax 1 -E 2.0 (configs[4] to the letter: with 64 samples the smoothed entropy of a substring that one sample holds 10-40
                          times is 5-6 bits, so nothing is printed at this coverage -- the counters still have to agree),
  * -P 1 --pmax 1 -E 7.0 (the same filter with the entropy bound out of the way: the sample-specific substrings themselves;
                          no reader-set order is needed, the order kernels are skipped),
  * -P 2 -E 7.0          (the reference's default filter: exercises order_big_kernel at scale)
on three random 8-mers and on a one-letter prefix cut at depth 10."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _paths(reads, n):
    import torch
    from pydsm import builder
    d = os.environ.get("DSM_BENCH_DIR", "/tmp/dsm_bench")
    os.makedirs(d, exist_ok=True)
    out = []
    for s in range(n):
        # (bench.py --nlocal 64 --reads ... names its files the same way: they are shared through DSM_BENCH_DIR)
        p = os.path.join(d, "sample-%d.s%d_r%d_l100_g%d_e0.005_p0.05.fmi" % (s, 42 + s, reads, reads * 5))
        if not os.path.exists(p):
            codes = builder.synth_reads(42 + s, reads, 100, reads * 5, 0.005, device="cuda", private_frac=0.05)
            builder.build_from_codes(codes, p + ".tmp")
            del codes
            os.replace(p + ".tmp", p)
        out.append(p)
    torch.cuda.empty_cache()
    return out


def test_sixty_four_samples_against_oracle():
    import orc
    import pydsm
    reads = int(os.environ.get("DSM_MANY_READS", "500000"))
    nsamples = int(os.environ.get("DSM_MANY_SAMPLES", "64"))
    paths = _paths(reads, nsamples)
    A = [pydsm.Index(p) for p in paths]
    O = [orc.Index(p) for p in paths]
    names = [ix.name for ix in A]
    assert len(set(names)) == nsamples and all(ix.n == reads * 202 for ix in A)
    rng = np.random.default_rng(4064)
    kmers = ["".join(rng.choice(list("ACGT"), 8)) for _ in range(3)]
    cfgs = [dict(fmin=10, pmin=1, pmax=1, emax=2.0), dict(fmin=10, pmin=1, pmax=1, emax=7.0), dict(fmin=10, pmin=2, emax=7.0)]
    threads = min(16, os.cpu_count() or 1)
    total = [0, 0, 0]
    for ci, kw in enumerate(cfgs):
        with pydsm.Miner(A, **kw) as m:
            for p in kmers:
                got, st = m.mine(p)
                want, ost = orc.mine(O, names, [p], threads=threads, **kw)
                assert got == want, (p, kw)
                assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, (p, kw)
                total[ci] += st.tuples
        got, st = pydsm.mine(A, "G", maxdepth=10, **kw)
        want, ost = orc.mine(O, names, ["G"], maxdepth=10, threads=threads, **kw)
        assert got == want, kw
        assert (st.reported, st.lf_steps, st.rank_ops, st.union_nodes, st.tuples, st.pairs) == ost, kw
        assert st.pair_order_exact == 1 and st.max_frontier > 100000
        total[ci] += st.tuples
    assert total[1] > 100 and total[2] > 10000, total  # (sample-specific substrings are rare: 5 % private sequence)
    for o in O:
        o.close()
    for ix in A:
        ix.close()
