"""Roofline evidence of the LF-step kernel (rounds 3 and 4).
  make_traffic.py trace <kernel_trace.csv> STEPS WARMUP
      -> per-pass total of ALL expand_kernel instantiations over the timed passes and the average per REAL launch
         (the launches queued ahead that found another frequency class return at once: they are counted, not averaged in)
  make_traffic.py traffic <pmc_agg lines> <bench json> <trace summary json>
      -> profiles/traffic.json: HBM bytes per LF-step launch = FETCH_SIZE + 1/2 x (coalesced record and handle reads, known
         exactly from the kernel's counters) + WRITE_SIZE (MI355X_MICROARCH.md: FETCH_SIZE counts a 128-byte request as 64 bytes;
         calibrated in round 1 with tools/gather_calib: random 64-byte reads factor 1.00, streaming reads 0.50)"""
import csv
import json
import sys


def trace(path, steps, warm):
    rows = [r for r in csv.DictReader(open(path)) if "expand_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
    passes = steps + warm
    if len(sys.argv) > 5:
        passes += int(sys.argv[5])  # further untimed passes of the profiled command (they come last)
    if len(dur) % passes != 0:  # every pass issues the same launches: anything else means the command ran something this script does not know
        raise SystemExit("make_traffic: %d LF-step launches do not divide into %d passes" % (len(dur), passes))
    per = len(dur) // passes
    timed = dur[per * warm: per * (warm + steps)]
    # A launch queued ahead that finds another frequency class only compares two words and returns (a microsecond or two; under the
    # tracer about as long as the sweep of a one-node level).  How many launches of a pass really swept a level is known exactly -- the
    # engine counts them (bench.py: roofline.launches / steps, the last argument) -- so the others are the shortest ones of the pass.
    real_per_pass = int(sys.argv[6]) if len(sys.argv) > 6 else None
    noop, real = [], []
    for p_ in range(warm, warm + steps):
        one = sorted(dur[per * p_: per * (p_ + 1)])
        k = per - real_per_pass if real_per_pass is not None else sum(1 for d in one if d < 2500)
        if k < 0:
            raise SystemExit("make_traffic: %d launches in a pass, fewer than the %d the engine counted" % (per, real_per_pass))
        noop += one[:k]
        real += one[k:]
    out = {"expand_launches_per_pass_total": per, "noop_launches_per_pass": len(noop) / steps, "real_launches_per_pass": len(real) / steps,
           "rocprof_expand_ms_per_step": sum(timed) / steps / 1e6, "rocprof_avg_launch_ms": sum(real) / max(1, len(real)) / 1e6,
           "steps": steps, "warmup": warm}
    print(json.dumps(out))


def traffic(pmc_path, bench_path, trace_path):
    agg = {}
    for line in open(pmc_path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        for k, v in json.loads(line).items():
            name, counter = k.rsplit("|", 1)
            if "expand_kernel" in name:
                agg.setdefault(counter, [0.0, 0])
                agg[counter][0] += v["sum"]
                agg[counter][1] += v["n"]
    bench = json.loads([ln for ln in open(bench_path) if ln.startswith("{")][-1])
    tr = json.loads(open(trace_path).read())
    launches = bench["roofline"]["launches"] // bench["steps"]
    nodes = bench["detail"]["rank0_nodes_per_step"]
    # passes the profiled command ran: launches the counters saw / launches of a pass (the real ones plus the dozen queued ahead
    # that returned at once); bench.py --steps 1 --warmup 0 is one pass (round 3's run had a second, untimed one: the text-format pass)
    passes = max(1, round(agg["FETCH_SIZE"][1] / (launches + 12.0)))
    fetch = agg["FETCH_SIZE"][0] * 1024.0 / passes   # KiB
    write = agg["WRITE_SIZE"][0] * 1024.0 / passes
    known = 20.0 * nodes                    # 16-byte compact record + 4-byte handle per node read, coalesced
    per_step = fetch + 0.5 * known + write
    hit, miss = agg.get("TCC_HIT_sum", [0, 0])[0], agg.get("TCC_MISS_sum", [0, 0])[0]
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    h = hashlib.sha256()
    for f in ("expand.hip", "lfstep.h", "common.h"):
        h.update(open(os.path.join(root, "dsm-framework_amd", "csrc", f), "rb").read())
    out = {"round": 4, "kernel_sha16": h.hexdigest()[:16], "reads": 10000000, "prefix_len": 1, "gpus": 1, "kernel": "expand_kernel<u32,...> (all record-format variants)",
           "launches": launches, "FETCH_SIZE_bytes": fetch, "WRITE_SIZE_bytes": write, "coalesced_read_bytes_known": known,
           "correction": "gfx950 FETCH_SIZE counts 128-B coalesced requests at 64 B (calibrated in round 1 with tools/gather_calib: random "
                         "64-B block reads factor 1.00, streaming reads factor 0.50); traffic = FETCH + 0.5*known coalesced reads (16-byte "
                         "compact record + 4-byte handle per node) + WRITE",
           "bytes_per_step": per_step, "bytes_per_launch": per_step / launches,
           "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
           "rocprof_avg_launch_ms": tr["rocprof_avg_launch_ms"], "rocprof_expand_ms_per_step": tr["rocprof_expand_ms_per_step"],
           "bench_avg_launch_ms_same_box": bench["roofline"]["avg_launch_ms"],
           "bench_expand_ms_per_step_same_box": bench["detail"]["expand_ms_per_step"],
           "pmc_launches_seen": agg["FETCH_SIZE"][1], "pmc_passes": passes,
           "sq": {k: v[0] for k, v in agg.items() if k.startswith("SQ_")},
           "source": "tools/profiling/r04_final.sh on one box: bench.py --steps 3 --warmup 1 (JSON line and kernel trace), then one "
                     "rocprofv3 --pmc pass per counter group of bench.py --steps 1 --warmup 0 --no-cpu --no-extras"}
    sq = out["sq"]
    if "SQ_INSTS_VALU" in sq and "SQ_WAVE_CYCLES" in sq and sq["SQ_WAVE_CYCLES"]:
        clock_hz = 2.34e9   # measured inside the kernel (s_memtime against s_memrealtime, tools/profiling/clock_probe.sh)
        kernel_s = bench["detail"]["expand_ms_per_step"] * 1e-3
        out["valu"] = {"insts_per_pass": sq["SQ_INSTS_VALU"] / passes, "insts_per_node": sq["SQ_INSTS_VALU"] / passes / nodes,
                       # SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count quad-cycles per wave; four waves share a SIMD
                       "busy_of_resident_wave_time": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
                       "busy_of_kernel_time": sq["SQ_INSTS_VALU"] / passes * 4.0 / (1024 * kernel_s * clock_hz),
                       "resident_waves_avg": sq["SQ_WAVE_CYCLES"] / passes * 4.0 / (kernel_s * clock_hz), "clock_hz": clock_hz}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
