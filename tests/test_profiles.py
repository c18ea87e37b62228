"""The committed evidence under profiles/ hangs together (CPU only): traffic.json's arithmetic, the bench lines' contract, and that the
PMC traffic bench.py would quote belongs to the kernel sources in the tree."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _line(name):
    return json.loads([ln for ln in open(os.path.join(PROF, name)) if ln.startswith("{")][-1])


def test_traffic_json_arithmetic_and_kernel_hash():
    t = json.load(open(os.path.join(PROF, "traffic.json")))
    total = t["FETCH_SIZE_bytes"] + 0.5 * t["coalesced_read_bytes_known"] + t["WRITE_SIZE_bytes"]
    assert abs(total - t["bytes_per_step"]) <= 1e-6 * total
    assert abs(t["bytes_per_step"] / t["launches"] - t["bytes_per_launch"]) <= 1e-6 * t["bytes_per_launch"]
    assert 0.2 < t["bytes_per_launch"] / (t["bench_avg_launch_ms_same_box"] * 1e-3) / 8e12 < 0.6   # the PMC fraction of the HBM peak
    # tracer and HIP events of the same box agree to within the tracer's overhead on these launches
    assert 1.0 <= t["rocprof_avg_launch_ms"] / t["bench_avg_launch_ms_same_box"] < 1.25
    h = hashlib.sha256()
    for f in ("expand.hip", "lfstep.h", "common.h"):
        h.update(open(os.path.join(ROOT, "dsm-framework_amd", "csrc", f), "rb").read())
    assert t["kernel_sha16"] == h.hexdigest()[:16], "profiles/traffic.json was measured on other LF-step kernel sources: rerun tools/profiling/r04_final.sh"


def test_bench_lines_keep_the_contract():
    for name in ("r04_bench_noextras.json", "r04_bench_full.json"):
        j = _line(name)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline", "cpu_baseline"):
            assert k in j, (name, k)
        r = j["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert j["n_gpus"] == 1 and j["vs_baseline"] is None and j["dtype"] == "u32" and "workload" in j["config"]
        assert j["cpu_baseline"]["kind"] == "reference" and j["cpu_baseline"]["cores"] >= 1
    full = _line("r04_bench_full.json")
    recs = full["extra_records"]
    assert len(recs) >= 9 and not any("error" in r for r in recs)
    share = [r for r in recs if r.get("record", "").startswith("configs[3] one-card share")][0]
    assert share["splits"] <= 4 and share["ms_per_step"] < 6000
